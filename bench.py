#!/usr/bin/env python
"""Headline benchmark: motion frames/s of the audio->motion path at batch 32 x 10 s clips per GPU (BASELINE.json).

One "step" = one pass of the whole hot path (wav2vec2 -> 5-scale AR decode with hipGraph replay -> VAE decode ->
re-encode) over one batch of 32 synthetic 10-second 16 kHz clips that are already resident in HBM.  N > 1 is
launched by ``python -m torch.distributed.run`` (one rank per GPU): every rank takes its own 32 clips (weak scaling,
clips are independent) and the step ends with the RCCL all-gather that collects the FLAME codes of all ranks.

Prints ONE JSON line (rank 0).  Besides the contract keys it carries ``roofline`` (the dominant kernel: the 128x128-tile
fp32 MFMA GEMM, timed with HIP events around every eager launch of it inside the timed region) and ``cpu_baseline``
(the CPU oracle, a restatement pinned bit-exact to the reference, timed on the host cores for a bounded sample).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GFLOP_PER_FRAME = 2.159          # algorithmic work with KV cache, BASELINE.md section 3
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak
MODES = {
    # precision -> (dtype string, dominant kernel symbol prefix in profiles/*pmc*.json, description, peak TF/s of ALGORITHMIC flops, note)
    "f32": ("f32", "gemm_f32_kernel<128, 128, 2, 2, 16, 0, 0>",
            "gemm_f32_kernel<128,128,2,2,16,0,0> (v_mfma_f32_32x32x2_f32): every launch of it (wav2vec2 conv/encoder GEMMs, AdaLN table)",
            PEAK_F32_MFMA_TFLOPS, "fp32 MFMA peak"),
    "f16x3": ("f32 (operands split into 2 fp16, 3 fp16 MFMA products per fp32 product, fp32 accumulate)", "gemm_p8_2wgp_kernel<0>",
              "gemm_p8_2wgp_kernel<0> (128x128 tiles, two persistent workgroups per CU, LDS-DMA staged, deferred epilogue, v_mfma_f32_32x32x16_f16 x3 per k-block): every launch of it (the wav2vec2 encoder GEMMs)",
              PEAK_F16_MFMA_TFLOPS / 3.0, "dense fp16 MFMA peak 2500 TF/s / 3 MFMA products per algorithmic product"),
}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU cores this process may really use: min(affinity, cgroup quota).  Oversubscribing torch's intra-op pool on a
    box that exposes more hardware threads than its quota makes the CPU baseline crawl."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return min(n, 32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--config", default="full")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3"],
                    help="GEMM arithmetic: exact fp32 MFMA, or fp16 operand-split MFMA with fp32-class accuracy")
    ap.add_argument("--branches", type=int, default=0, help="concurrent clip groups of the AR body (0 = auto)")
    ap.add_argument("--splitk", default="0,0", help="tuning: split-K tile threshold,target workgroups (0 = keep)")
    ap.add_argument("--overlap", action="store_true", help="overlapped schedule: wav2vec2 of chunk index j+1 beside the AR/VAE body of j")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-mode", action="store_true", help="skip the extra (untimed) f32-mode measurement")
    ap.add_argument("--cpu-clips", type=int, default=12)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with: python -m torch.distributed.run --nnodes=1 "
                     "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from artalk_amd.config import ARTalkConfig
    from artalk_amd.model import BitwiseARModel
    from artalk_amd.synth import synth_audio
    from artalk_amd.weights import generate_state_dict

    cfg = ARTalkConfig.by_name(args.config)
    t0 = time.time()
    sd = generate_state_dict(cfg)
    model = BitwiseARModel(cfg).eval().to(dev)
    model.load_state_dict(sd, strict=True)
    model.set_precision(args.precision)
    sk = [int(x) for x in args.splitk.split(",")]
    model.set_graphs(True, args.branches, sk[0], sk[1])
    t_load = time.time() - t0
    log(f"rank {rank}: weights generated + loaded in {t_load:.1f} s")

    B = args.batch
    seeds = [rank * B + i for i in range(B)]
    audios = [torch.from_numpy(synth_audio(s, args.seconds)).to(dev) for s in seeds]     # resident in HBM
    frames_per_clip = model.seq_length(audios[0].shape[0])
    chunks = sum(model.n_chunks(a.shape[0]) for a in audios)
    model.reserve(B, chunks)
    if args.overlap:
        model.set_overlap(True)
    gathered = torch.empty(world * B, frames_per_clip, cfg.motion_dim, device=dev) if world > 1 else None

    def step():
        outs = model.inference_batch(audios)
        if world > 1:
            dist.all_gather_into_tensor(gathered, torch.stack(outs))     # result collection over xGMI (RCCL)
        return outs

    for _ in range(args.warmup):
        outs = step()
    torch.cuda.synchronize()
    log("warmup done")
    model.set_profiling(1)       # light: hipGraphs stay on; HIP events bracket the eager launches of the dominant kernel
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    log(f"timed region done: {dt:.3f} s for {args.steps} steps")
    prof = model.get_profile()       # numbers of the last timed step
    model.set_profiling(0)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_frames = args.steps * B * frames_per_clip * world
    value = total_frames / dt
    # parity of the benchmarked configuration against the reference golden (clip seed 0 == tests/golden/full_10s_s0)
    parity = None
    gpath = os.path.join(REPO, "tests", "golden", "full_10s_s0.npz")
    if args.config == "full" and args.seconds == 10.0 and os.path.exists(gpath):
        g = np.load(gpath)
        parity = {"case": "full_10s_s0", "flame_max_abs_err": float(np.abs(outs[0].cpu().numpy() - g["out"]).max())}

    dom_tflops = prof["dom_flop"] / (prof["dom_ms"] * 1e-3) / 1e12 if prof["dom_ms"] > 0 else 0.0
    # HBM traffic per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of
    # this same command, separate runs; profiles/r01_pmc_traffic.json, corrected as MI355X_MICROARCH.md prescribes)
    traffic = None
    tpath = os.path.join(REPO, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tpath):
        for k, v in json.load(open(tpath)).items():
            if isinstance(v, dict) and k.startswith(MODES[args.precision][1]):
                traffic = round(v["hbm_bytes_per_launch"])
    dtype, _, kdesc, peak, peak_note = MODES[args.precision]
    roofline = {
        "bound": "mfma", "achieved": round(dom_tflops, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
        "frac": round(dom_tflops / peak, 4), "traffic": traffic, "kernel": kdesc, "peak_note": peak_note,
        "launches": int(prof["dom_launches"]), "avg_launch_ms": round(prof["dom_ms"] / max(prof["dom_launches"], 1), 4),
        "share_of_step_ms": round(prof["dom_ms"], 2),
    }
    result = {
        "metric": "motion frames/sec (25 fps clips) at batch=32 per GPU", "value": round(value, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"configs[2]: batch={B} synthetic {args.seconds:g} s 16 kHz clips per GPU, {chunks} 4-s chunks, "
                               "hipGraph decode loop, deterministic synthetic weights (489.5 M params)",
                   "clips_per_gpu": B, "frames_per_clip": frames_per_clip, "precision": args.precision,
                   "model_config": args.config},
        "fps_per_clip": round(value / (B * world), 1),
        "algorithmic_tflops": round(value * GFLOP_PER_FRAME / 1e3, 2),
        # overlapped schedule: the wav2vec2 buckets are measured on their own stream and run BESIDE ada/ar of the previous chunk
        # index, so the buckets add up to more than total_ms; ada_ms includes waiting for the features of its chunk index
        "schedule": "overlapped (wav2vec2 of chunk index j+1 beside the AR/VAE body of j)" if (args.overlap and B >= 8) else "sequential",
        "stages_ms": {k: round(prof[k], 2) for k in ("style_ms", "w2v_conv_ms", "w2v_encoder_ms", "ada_ms", "ar_ms", "vae_ms", "total_ms")},
        "roofline": roofline,
        "parity": parity,
        "load_s": round(t_load, 1),
    }

    # the same workload in the other GEMM mode (exact fp32 MFMA), outside the timed region: the f32 kernel is MFMA-bound and
    # sits much closer to its (6x lower) roofline, the f16x3 kernel is faster in absolute terms
    if world == 1 and args.precision == "f16x3" and not args.no_alt_mode:
        model.set_precision("f32")
        step(); torch.cuda.synchronize()
        model.set_profiling(1)
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        dt32 = (time.perf_counter() - t1) / 2
        p32 = model.get_profile()
        model.set_profiling(0)
        model.set_precision("f16x3")
        tf32 = p32["dom_flop"] / (p32["dom_ms"] * 1e-3) / 1e12 if p32["dom_ms"] > 0 else 0.0
        result["f32_mode"] = {"value": round(B * frames_per_clip / dt32, 1), "ms_per_step": round(dt32 * 1e3, 2),
                              "roofline": {"bound": "mfma", "achieved": round(tf32, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                           "frac": round(tf32 / PEAK_F32_MFMA_TFLOPS, 4), "kernel": MODES["f32"][2]}}

    if not args.no_cpu_baseline and world == 1:
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        from artalk_oracle import ARTalkOracle
        cores = host_cores()
        torch.set_num_threads(cores)
        log(f"cpu baseline on {cores} threads")
        oracle = ARTalkOracle(cfg, sd)
        n_done, t_cpu, frames = 0, 0.0, 0
        for i in range(args.cpu_clips):
            a = torch.from_numpy(synth_audio(i, args.seconds))
            t1 = time.perf_counter()
            o = oracle.inference({"audio": a[None], "style_motion": None})
            t_cpu += time.perf_counter() - t1
            frames += o.shape[1]
            n_done += 1
            log(f"cpu baseline clip {i}: {t_cpu:.1f} s so far")
            if t_cpu > 12.0:
                break
        result["cpu_baseline"] = {
            "value": round(frames / t_cpu, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_done} of the same {args.seconds:g} s clips, batch 1 sequentially as the reference runs "
                      f"(no KV cache, fp32 torch-CPU), {t_cpu:.1f} s of CPU work",
        }
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
