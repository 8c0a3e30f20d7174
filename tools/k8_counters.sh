#!/bin/bash
# usage: tools/k8_counters.sh <out.txt> -- per K8 launch (kernel x grid = layer) SQ / TCP / TCC
# counters of the cfg-2 encoder layers: which unit is busy while the MFMA pipe is not.
R=$PWD; OUT=$R/${1:-gpurun_out/k8_counters.txt}; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/k8c
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_LEVEL_WAVES" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM SQ_BUSY_CYCLES" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_BUSY_avr TCC_TAG_STALL_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/k8c/$tag -o s -- python3 $R/tools/conv_pmc.py > /dev/null 2>&1
done
python3 - > $OUT <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/k8c/**/s_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'conv_' not in n: continue
        m = re.search(r'(conv_\w+)', n)
        key = (m.group(1), int(r.get('Grid_Size', 0) or 0), int(r.get('Workgroup_Size', 0) or 0))
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/k8c/**/s_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'conv_' not in n: continue
        m = re.search(r'(conv_\w+)', n)
        key = (m.group(1), int(r.get('Grid_Size', 0) or 0), int(r.get('Workgroup_Size', 0) or 0))
        dur[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print("K8 launches of one cfg-2 step (B=128; tools/conv_pmc.py), rocprofv3 --pmc, one counter group per pass;")
print("per launch averages; SQ_* cycle counters are summed over the chip's SEs / CUs as rocprofv3 reports them")
for key in sorted(agg):
    d = agg[key]
    us = sorted(dur.get(key, [0]))[len(dur.get(key, [0])) // 2]
    print(f"\n{key[0]}  grid {key[1]} x {key[2]} threads   (under profiler: {us:.1f} us)")
    for c in sorted(d):
        print(f"   {c:34s} {sum(d[c]) / len(d[c]):16.0f}   (n={len(d[c])})")
    g = lambda c: (sum(d[c]) / len(d[c])) if c in d else float('nan')
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles, summed over the 1024 SIMDs; GRBM_GUI_ACTIVE: cycles,
    # summed over the 8 XCDs; SQ_WAVE_CYCLES / SQ_WAIT_*: quad-cycles summed over waves
    dur_cyc = g('GRBM_GUI_ACTIVE') / 8
    print(f"   -> kernel {dur_cyc / 2.4e3:.1f} us @2.4 GHz; MFMA pipe busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / max(dur_cyc, 1):.3f} of the"
          f" kernel's cycles (per SIMD); waves waiting {g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):.3f} of their cycles,"
          f" on LDS {g('SQ_WAIT_INST_LDS') / max(g('SQ_WAVE_CYCLES'), 1):.3f};"
          f" LDS bank conflicts {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.3f} of LDS-active cycles;"
          f" L2 hit rate {g('TCC_HIT_sum') / max(g('TCC_HIT_sum') + g('TCC_MISS_sum'), 1):.3f};"
          f" vector-L1 accesses per L2 read request {g('TCP_TOTAL_CACHE_ACCESSES_sum') / max(g('TCP_TCC_READ_REQ_sum'), 1):.2f};"
          f" HBM fetch {g('FETCH_SIZE') / 1024:.1f} MiB write {g('WRITE_SIZE') / 1024:.1f} MiB (raw counters, KiB units);"
          f" resident waves per SIMD ~{g('SQ_WAVE_CYCLES') * 4 / 1024 / max(dur_cyc, 1):.2f};"
          f" mean latency of a vector-memory instruction (LDS-DMA pieces incl.) ~{g('SQ_INST_LEVEL_VMEM') / max(g('SQ_INSTS_VMEM'), 1):.0f} cycles,"
          f" of an LDS instruction ~{g('SQ_INST_LEVEL_LDS') / max(g('SQ_INSTS_LDS'), 1):.0f}")
PY
rm -rf /tmp/k8c
