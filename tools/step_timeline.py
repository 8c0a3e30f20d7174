"""usage: step_timeline.py <kernel_trace.csv> -- one replayed step as a
timeline: start offset, duration and the gap in front of every kernel."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'stage_batch_kernel' in n or 'step_prologue_kernel' in n]
pairs = [(a, b) for a, b in zip(idx, idx[1:]) if b - a > 15]
# (the bench also times the step without its optimiser launch: take a full one)
full = [(a, b) for a, b in pairs if any('rmsprop' in n for n in names[a:b])]
pairs = full if len(full) >= 3 else pairs
a, b = pairs[-3]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
nxt = int(rows[b]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3
print(f"kernels/step {len(step)} period us {(nxt - t0) / 1e3:.1f} busy us {busy:.1f}")
prev = t0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('void ', '')[:70]
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.2f} gap {(s - prev) / 1e3:6.2f}  {n}")
    prev = e
print(f"{(nxt - t0) / 1e3:8.1f} next step, gap {(nxt - prev) / 1e3:.2f}")
