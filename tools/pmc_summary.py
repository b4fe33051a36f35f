#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel (mean over dispatches)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "render_mega" not in k and "trace_batch" not in k and "reduce_stage" not in k: continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        meta[k] = (row["Grid_Size"], row["Workgroup_Size"], row["LDS_Block_Size"], row["VGPR_Count"], row["SGPR_Count"], row["Scratch_Size"])
for k in acc:
    print(k[:90]); print("  grid,wg,lds,vgpr,sgpr,scratch =", meta[k])
    c = {n: sum(v)/len(v) for n, v in acc[k].items()}
    for n in sorted(c): print(f"  {n:28s} {c[n]:18.1f}  (n={len(acc[k][n])})")
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_THREAD_CYCLES_VALU" in c:
        print(f"  -> active-lane fraction (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64)) = {c['SQ_THREAD_CYCLES_VALU']/(c['SQ_ACTIVE_INST_VALU']*64):.3f}")
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
        print(f"  -> ACTIVE_INST_VALU*4 / BUSY_CYCLES = {c['SQ_ACTIVE_INST_VALU']*4/c['SQ_BUSY_CYCLES']:.3f}")
    if "SQ_WAVE_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
        print(f"  -> ACTIVE_INST_VALU / WAVE_CYCLES = {c['SQ_ACTIVE_INST_VALU']/c['SQ_WAVE_CYCLES']:.3f}")
