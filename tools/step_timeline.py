#!/usr/bin/env python
"""Idle gaps of the GPU inside the timed steps of a bench run: reads the kernel trace (and the memory-copy trace, if present) of
`rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 bench.py ...`, merges all kernel intervals
(every stream) and lists the gaps longer than GAP_US between them with the kernels on either side and the copies that ran inside."""
import csv, glob, sys
d = sys.argv[1]
gap_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
rows = [r for f in kt for r in csv.DictReader(open(f))]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]) for r in rows)
copies = []
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", ""), r.get("Bytes", r.get("Size", ""))))
copies.sort()
t0 = iv[0][0]
end, last = iv[0][1], iv[0][2]
idle = 0.0
# the timed region: from the first conv0_kernel after the last big gap (> 0.3 s: warmup / weight load) to the end
print(f"{len(iv)} kernels, {len(copies)} copies")
for s, e, n in iv[1:]:
    if s > end:
        g = (s - end) / 1e3
        if g >= gap_us:
            inside = [c for c in copies if c[0] < s and c[1] > end]
            print(f"t={(end - t0) / 1e6:9.3f} ms  idle {g:9.1f} us  after {last:45s} before {n:45s} copies inside: "
                  + "; ".join(f"{c[2]} {c[3]}B {(c[1] - c[0]) / 1e3:.0f}us" for c in inside[:6]) + (" ..." if len(inside) > 6 else ""))
        idle += g
    if e > end:
        end, last = e, n
print(f"total span {(end - t0) / 1e6:.1f} ms, idle {idle / 1e3:.1f} ms")
# the launch sequence across one batch boundary (from the last body kernels of a batch to the first conv0_kernel of the next), with queues
full = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r["Kernel_Name"].split("(")[0][:80]) for r in rows))
conv = [i for i, x in enumerate(full) if "conv0_kernel" in x[4]]
if len(conv) >= 3:
    i1 = conv[-2]
    i0 = max(0, i1 - 40)
    print("\nboundary before the second-to-last conv0_kernel:")
    prev_end = None
    for s, e, q, st, n in full[i0:i1 + 3]:
        print(f"  t={(s - t0) / 1e6:9.3f} ms dur {(e - s) / 1e3:7.1f} us  since prev end {((s - prev_end) / 1e3 if prev_end else 0):7.1f} us  queue {q} stream {st}  {n}")
        prev_end = e if prev_end is None else max(prev_end, e)
# join bubbles of the AR/VAE body: at every copy of a chunk index's motion (copyBufferRectAligned, issued behind the join of the two
# clip groups' graphs) the end of the last kernel of either group's stream before it - the difference is time one group ran alone
kern = [x for x in full if "rocclr" not in x[4] and "at::native" not in x[4]]
joins = [x for x in full if "copyBufferRectAligned" in x[4]]
print("\njoin bubbles (us): end of group A's last kernel, of group B's, before each motion copy")
import bisect
starts = [x[0] for x in kern]
for jx in joins[-9:]:
    i = bisect.bisect_left(starts, jx[0])
    last = {}
    for x in reversed(kern[max(0, i - 400):i]):
        if x[3] not in last:
            last[x[3]] = x
        if len(last) == 2:
            break
    if len(last) == 2:
        (sa, xa), (sb, xb) = sorted(last.items())
        print(f"  t={(jx[0] - t0) / 1e6:9.3f} ms  stream {sa} ended {(jx[0] - xa[1]) / 1e3:8.1f} us before the copy ({xa[4][:40]}), stream {sb} {(jx[0] - xb[1]) / 1e3:8.1f} us ({xb[4][:40]})")
