#!/usr/bin/env python
"""Condense rocprofv3 output directories (gpurun_out/...) into the small summaries kept under profiles/.

  summarize_profile.py stats  <dir> <out.csv>            kernel-trace --stats: the per-kernel table
  summarize_profile.py pmc    <fetch_dir> <write_dir> <out.json>   FETCH_SIZE / WRITE_SIZE passes -> bytes per launch
  summarize_profile.py shapes <dir> <out.csv>            kernel-trace: time per (kernel, grid size), the launch-shape view
  summarize_profile.py levels <dir> <out.csv>            kernel-trace of a --branches 1 run: the AR/VAE body per scale step

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
exactly half of a wide (16 B/lane) coalesced read stream, so reads = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import os
import sys


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    assert f, (d, pat)
    return f[0]


def short(name):
    name = name.replace("void ", "").replace("artalk::", "")
    return name.split("(")[0][:70]


def stats(d, out):
    rows = list(csv.DictReader(open(find(d, "*kernel_stats.csv"))))
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct\n")
        for r in rows[:24]:
            f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.3f},{float(r['AverageNs']) / 1e3:.2f},"
                    f"{float(r['MinNs']) / 1e3:.2f},{float(r['MaxNs']) / 1e3:.2f},{float(r['Percentage']):.3f}\n")
    print(open(out).read())


def shapes(d, out):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(find(d, "*kernel_trace.csv"))):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in agg.values())
    with open(out, "w") as f:
        f.write("kernel,blocks_x,grid_y,grid_z,calls,total_ms,avg_us,pct\n")
        for k, v in sorted(agg.items(), key=lambda t: -t[1][1])[:60]:
            f.write(f"\"{k[0]}\",{k[1]},{k[2]},{k[3]},{v[0]},{v[1] / 1e3:.3f},{v[1] / v[0]:.2f},{100 * v[1] / tot:.2f}\n")
    print(open(out).read())


def levels(d, out):
    """Kernel trace of a run with ONE clip group (bench.py --branches 1): the AR/VAE body of a chunk index is a fixed kernel
    sequence; cut it at ar_begin / ar_bits / dec_input / bsq_history and report, per segment, the summed kernel time, the wall
    span (first start to last end) and the launches, averaged over the bodies of the trace, plus the top kernels per segment."""
    rows = [r for r in csv.DictReader(open(find(d, "*kernel_trace.csv")))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seg_names = ["hist_kv", "level0", "level1", "level2", "level3", "level4", "vae_decode", "reencode"]
    segs = {n: dict(kernel_us=0.0, wall_us=0.0, launches=0, n=0, kern=collections.defaultdict(lambda: [0, 0.0])) for n in seg_names}
    cur, lvl, start, last_end, bodies = None, 0, None, None, 0
    pending = []     # launches since the last boundary that may belong to hist_kv (they precede ar_begin)

    def close(name, items):
        if not items:
            return
        sg = segs[name]
        sg["n"] += 1
        sg["launches"] += len(items)
        sg["wall_us"] += (max(int(r["End_Timestamp"]) for r in items) - min(int(r["Start_Timestamp"]) for r in items)) / 1e3
        for r in items:
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            sg["kernel_us"] += dur
            k = sg["kern"][short(r["Kernel_Name"])]
            k[0] += 1
            k[1] += dur

    items = []
    state = "outside"
    for r in rows:
        name = short(r["Kernel_Name"])
        if state == "outside":
            # the history K/V GEMMs (12 launches + a pack) directly precede ar_begin: keep a short window
            if name.startswith("ar_begin_kernel"):
                close("hist_kv", pending[-13:])
                pending = []
                state, lvl, items = "level", 0, [r]
                bodies += 1
            else:
                pending.append(r)
            continue
        items.append(r)
        if state == "level" and name.startswith("ar_bits_kernel"):
            close(f"level{lvl}", items)
            items = []
            lvl += 1
            if lvl == 5:
                state = "vae"
        elif state == "vae" and name.startswith("dec_finish_kernel"):
            close("vae_decode", items)
            items, state = [], "reenc"
        elif state == "reenc" and name.startswith("vq_embed_kernel"):
            close("reencode", items)
            items, state, pending = [], "outside", []
    with open(out, "w") as f:
        f.write("segment,bodies,launches_per_body,kernel_us_per_body,wall_us_per_body,top_kernels(us per body)\n")
        for n in seg_names:
            sg = segs[n]
            if not sg["n"]:
                continue
            top = sorted(sg["kern"].items(), key=lambda t: -t[1][1])[:6]
            tops = "; ".join(f"{k} x{v[0] // sg['n']} {v[1] / sg['n']:.0f}" for k, v in top)
            f.write(f"{n},{sg['n']},{sg['launches'] / sg['n']:.0f},{sg['kernel_us'] / sg['n']:.0f},{sg['wall_us'] / sg['n']:.0f},\"{tops}\"\n")
    print(open(out).read())


def sequence(d, out, which=3):
    """The launch list of ONE body (the `which`-th ar_begin ... vq_embed span of the trace): start offset, duration, gap to the end
    of the previous launch, queue, grid, workgroup, LDS and name - to see where a scale step's wall time goes between its kernels."""
    rows = [r for r in csv.DictReader(open(find(d, "*kernel_trace.csv")))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    begins = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("ar_begin_kernel")]
    i0 = begins[min(which, len(begins) - 1)]
    t0, prev_end, n_bits = int(rows[i0]["Start_Timestamp"]), None, 0
    with open(out, "w") as f:
        f.write("start_us,dur_us,gap_us,queue,grid,wg,lds,kernel\n")
        for r in rows[i0:]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
            f.write(f"{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{gap:.1f},{r.get('Queue_Id', '')},{r.get('Grid_Size_X', '')},{r.get('Workgroup_Size_X', '')},"
                    f"{r.get('LDS_Block_Size', '')},{short(r['Kernel_Name'])}\n")
            prev_end = max(prev_end or 0, e)
            n_bits += 1
            if n_bits > 1 and (short(r["Kernel_Name"]).startswith("ar_begin_kernel") or n_bits > 900):      # the next body starts
                break


def markers(d, out):
    """rocprofv3 --marker-trace: host-side roctx ranges of the path (artalk.* : enqueue time of each kernel group)."""
    agg = collections.defaultdict(lambda: [0, 0.0])
    f = find(d, "*marker_api_trace.csv")
    for r in csv.DictReader(open(f)):
        name = r.get("Function") or r.get("Marker_Name") or r.get("Name") or "?"
        agg[name][0] += 1
        agg[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    with open(out, "w") as fo:
        fo.write("range,count,total_host_ms,avg_host_us\n")
        for k, v in sorted(agg.items(), key=lambda t: -t[1][1]):
            fo.write(f"\"{k}\",{v[0]},{v[1] / 1e3:.3f},{v[1] / v[0]:.1f}\n")
    print(open(out).read())


def pmc(fd, wd, out):
    res = collections.defaultdict(dict)
    for tag, d in (("FETCH_SIZE", fd), ("WRITE_SIZE", wd)):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(find(d, "*counter_collection.csv"))):
            if r["Counter_Name"] != tag:
                continue
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        for k, (n, v) in agg.items():
            res[k][tag + "_KiB_per_launch"] = v / n
            res[k]["launches_" + tag] = n
    outd = {}
    for k, v in res.items():
        if "FETCH_SIZE_KiB_per_launch" in v and "WRITE_SIZE_KiB_per_launch" in v:
            v["hbm_read_bytes_per_launch"] = 2.0 * v["FETCH_SIZE_KiB_per_launch"] * 1024
            v["hbm_write_bytes_per_launch"] = v["WRITE_SIZE_KiB_per_launch"] * 1024
            v["hbm_bytes_per_launch"] = v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"]
            outd[k] = v
    top = dict(sorted(outd.items(), key=lambda t: -t[1]["hbm_bytes_per_launch"] * t[1]["launches_FETCH_SIZE"])[:12])
    json.dump(top, open(out, "w"), indent=1)
    for k, v in top.items():
        print(f"{k:70s} launches={v['launches_FETCH_SIZE']:5d} read={v['hbm_read_bytes_per_launch'] / 1e6:9.1f} MB write={v['hbm_write_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    if sys.argv[1] == "markers":
        markers(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "levels":
        levels(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "sequence":
        sequence(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "shapes":
        shapes(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
