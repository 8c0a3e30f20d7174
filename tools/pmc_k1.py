"""usage: pmc_k1.py <out.json> -- aggregate FETCH_SIZE / WRITE_SIZE (KiB) per K1 kernel launch
from the two --pmc passes round_artifacts.sh makes over tools/k1_only.py."""
import collections
import csv
import json
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402

NAMES = {"render_wave_kernel": "render_fwd_kernel", "render_fwd_kernel": "render_fwd_kernel",
         "logprob_wave_kernel": "logprob_fwd_kernel", "logprob_fwd_kernel": "logprob_fwd_kernel",
         "bwd_cell_kernel": "render_bwd_kernel", "render_bwd_kernel": "render_bwd_kernel",
         "render_bwd_gather_kernel": "render_bwd_kernel"}
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"gpurun_out/k1pmc_{c}/k1_counter_collection.csv")))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] != c:
            continue
        for pat, key in NAMES.items():
            if pat + "<" in r["Kernel_Name"] or pat + "(" in r["Kernel_Name"]:
                agg[key].append(float(r["Counter_Value"]))
                res[key]["kernel"] = pat
    for k, v in agg.items():
        res[k][c + "_KB_per_launch"] = round(sum(v) / len(v), 1)
        res[k]["launches_sampled"] = len(v)
cfg = bench.CONFIGS["mnist_24_24_bs128"]
alg = bench.k1_algorithmic_bytes(cfg)
out = {"how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on "
              "tools/k1_only.py: the three K1 kernels at cfg-2 (B=128, M=24, C=1, 40x40, 11x11 templates); "
              "counter unit KiB; 'fetch_x2' applies the gfx950 FETCH_SIZE x2 correction of "
              "MI355X_MICROARCH.md; algorithmic bytes: DESIGN.md section 4",
       "kernels": {}}
for k, d in res.items():
    f, w = d.get("FETCH_SIZE_KB_per_launch", 0) * 1024, d.get("WRITE_SIZE_KB_per_launch", 0) * 1024
    d["hbm_bytes_per_launch_raw"] = int(f + w)
    d["hbm_bytes_per_launch_fetch_x2"] = int(2 * f + w)
    d["algorithmic_bytes_per_launch"] = alg[k] * cfg["batch"]
    out["kernels"][k] = d
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
