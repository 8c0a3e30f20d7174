"""Print the headline and the extra_workloads of a bench.py JSON line (file argument)."""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d.get("timing"))
if "roofline" in d:
    r = d["roofline"]
    print("roofline", r["kernel"], r["frac"], r.get("per_layer_us"))
    for k, v in r["other_kernels"].items():
        print("   ", k, v.get("us", v.get("us_per_step")), v["frac"])
for e in d.get("extra_workloads", []):
    print(e["workload"][:60], e.get("ms_per_step"), e.get("dominant_kernel", {}).get("frac"),
          {k: v["us"] for k, v in e.get("k1", {}).items()})
