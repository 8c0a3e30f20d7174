import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
gaps = {"after prologue": [], "rmsprop -> prologue": [], "rmsprop -> graph": []}
for a, b in zip(rows, rows[1:]):
    gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if "rmsprop" in a["Kernel_Name"]:
        gaps["rmsprop -> prologue" if "prologue" in b["Kernel_Name"] else "rmsprop -> graph"].append(gap)
    elif "prologue" in a["Kernel_Name"]:
        gaps["after prologue"].append(gap)
for k, v in gaps.items():
    v = sorted(v)
    if v:
        print(f"{k:22s} n {len(v):3d}  median {v[len(v)//2]:6.2f} us  min {v[0]:6.2f}  max {v[-1]:8.2f}")
