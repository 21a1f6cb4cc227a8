#!/usr/bin/env python
"""Idle time between the kernels of the wav2vec2 phase (eager launches) in a rocprofv3 kernel trace of bench.py: for every step, from
its conv0_kernel to the last pool_silu_kernel - kernel time, gaps, and the distribution of the gaps."""
import csv, glob, sys
rows = [r for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
starts = [i for i, x in enumerate(iv) if "conv0_kernel" in x[2]]
for a in starts[-3:]:
    b = next(i for i in range(a, len(iv)) if "pool_silu" in iv[i][2])
    seg = iv[a:b + 1]
    busy = sum(e - s for s, e, _ in seg) / 1e3
    gaps = [(seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(len(seg) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"step: {len(seg)} kernels, span {(seg[-1][1] - seg[0][0]) / 1e3:.0f} us, kernel time {busy:.0f} us, gaps {sum(pos):.0f} us "
          f"(median {sorted(pos)[len(pos) // 2]:.2f} us, max {max(pos):.1f} us, > 5 us: {sum(1 for g in pos if g > 5)})")
