"""usage: pmc_k8.py <out.json> -- aggregate FETCH_SIZE / WRITE_SIZE (KiB) per K8 kernel launch."""
import csv, json, collections, sys
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"gpurun_out/k8pmc_{c}/k8_counter_collection.csv")))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] != c:
            continue
        n = r["Kernel_Name"]
        for key, pats in (("conv_fwd_kernel", ("conv_fwd_kernel", "conv_fwd_pipe_kernel")),
                          ("conv_dgrad_kernel", ("conv_dgrad_kernel", "conv_dgrad_pipe_kernel")),
                          ("conv_wgrad_kernel", ("conv_wgrad_kernel", "conv_wgrad_pipe_kernel")),
                          ("conv_bwd_pair_kernel", ("conv_bwd_pair_kernel", "conv_bwd_pair_pipe_kernel",
                                                    "conv_bwd_pair_mixed_kernel"))):
            if any(p + "<" in n or p + "(" in n for p in pats):
                agg[key].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k][c + "_KB_per_launch"] = round(sum(v) / len(v), 1)
        res[k]["launches_sampled"] = len(v)
out = {"how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on "
              "tools/conv_pmc.py: the K8 launches of one cfg-2 step (B=128, 128-channel layers 19x19/s2, "
              "9x9/s1, 7x7/s1); averages over the three layers; counter unit KiB; 'fetch_x2' applies the "
              "gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md (these kernels read 16 B/lane)",
       "kernels": {}}
for k, d in res.items():
    f, w = d.get("FETCH_SIZE_KB_per_launch", 0) * 1024, d.get("WRITE_SIZE_KB_per_launch", 0) * 1024
    d["hbm_bytes_per_launch_raw"] = int(f + w)
    d["hbm_bytes_per_launch_fetch_x2"] = int(2 * f + w)
    out["kernels"][k] = d
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
