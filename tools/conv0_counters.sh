# What bounds conv0_kernel (VERDICT r4 next #4a): hardware counters of the kernel inside the real step, each counter in its own rocprofv3
# pass (--pmc with --kernel-trace only).  Output: gpurun_out/r05/conv0_counters.log
#   VALU duty   = SQ_ACTIVE_INST_VALU / (4 * SQ_BUSY_CYCLES)   (cycles a SIMD's vector ALU executes / SIMD-cycles the SQs were busy; approx.)
#   issue rate  = SQ_INSTS_VALU / waves / duration
#   store rate  = WRITE_SIZE (KB, L2 -> fabric) / duration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
for C in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR WRITE_SIZE GRBM_GUI_ACTIVE; do
  rm -rf $O/pmc_c0_$C
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_c0_$C -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-mode > $O/pmc_c0_$C.log 2>&1 || echo "counter $C failed"
done
python3 - <<'PY' | tee gpurun_out/r05/conv0_counters.log
import csv, glob, collections
O="gpurun_out/r05"
med=lambda a: sorted(a)[len(a)//2]
vals={}; dur=[]
for d in glob.glob(f"{O}/pmc_c0_*/"):
    c=d.rstrip("/").split("pmc_c0_")[1]
    v=collections.defaultdict(float)
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv0_kernel" in r["Kernel_Name"] and r["Counter_Name"]==c:
                v[r["Dispatch_Id"]]+=float(r["Counter_Value"])
    if v: vals[c]=med(list(v.values()))
    for f in glob.glob(d+"/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv0_kernel" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
d=med(dur)
print(f"conv0_kernel: median duration {d:.1f} us over {len(dur)} launches (96 chunks x 12800 frames x 512 channels: 2.52 GB of P8 written, 24.6 MB read)")
for k,v in sorted(vals.items()): print(f"  {k:24s} {v:.4g}")
g=vals.get("GRBM_GUI_ACTIVE"); 
if g: print(f"  shader clock            {g/8/d/1e3:.2f} GHz")
if "SQ_ACTIVE_INST_VALU" in vals and "SQ_BUSY_CYCLES" in vals: print(f"  VALU-active / SQ-busy cycles        {vals['SQ_ACTIVE_INST_VALU']/vals['SQ_BUSY_CYCLES']:.3f}")
if "SQ_ACTIVE_INST_VALU" in vals and g: print(f"  VALU-active cycles per SIMD-cycle   {vals['SQ_ACTIVE_INST_VALU']/ (g/8*1024):.3f}   (SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles))")
if "SQ_INSTS_VALU" in vals and g: print(f"  VALU instructions per SIMD-cycle    {vals['SQ_INSTS_VALU']/(g/8*1024):.3f}")
if "SQ_WAIT_INST_ANY" in vals and "SQ_WAVE_CYCLES" in vals: print(f"  waves waiting on an instruction     {vals['SQ_WAIT_INST_ANY']/vals['SQ_WAVE_CYCLES']:.3f} of wave-cycles")
if "WRITE_SIZE" in vals: print(f"  fabric writes           {vals['WRITE_SIZE']/1e6:.2f} GB (WRITE_SIZE in KB) -> {vals['WRITE_SIZE']*1e3/d/1e6:.2f} TB/s of 8 TB/s HBM peak")
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
