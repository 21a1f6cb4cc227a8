"""Data-parallel clip sharding + result collection (artalk_amd/dist.py) with world_size 2 on the gloo backend.
The GPU path uses the same functions with backend 'nccl' (RCCL); here infer_fn is a stand-in that makes the
result a pure function of the clip so the gathered order can be checked."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from artalk_amd.dist import gather_clips, run_sharded, seq_length, shard_range


def _fake_infer(audios, styles):
    outs = []
    for a in audios:
        T = seq_length(a.shape[0])
        outs.append(a[:1].repeat(T, 106) + torch.arange(T, dtype=torch.float32)[:, None])
    return outs


def _worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    audios = [torch.full((16000 + 640 * i,), float(i + 1)) for i in range(n_clips)]
    res = run_sharded(_fake_infer, audios, None, gather=True)
    want = _fake_infer(audios, None)
    ok = len(res) == n_clips and all(torch.equal(r, w) for r, w in zip(res, want))
    q.put((rank, ok))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(n_clips, worker=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker or _worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]


class _LazyClips:
    """What bench.py passes: len() + indexing, clips built on demand (only the local shard should ever be built)."""
    def __init__(self, n):
        self.n, self.built = n, []

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        self.built.append(i)
        return torch.full((16000 + 640 * i,), float(i + 1))


def _worker_lazy(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    clips = _LazyClips(n_clips)
    # what bench.py passes: the frame count of every clip of the job, known on the host -> ONE collective and no read-back
    calls, reads = [], []
    orig, orig_cpu = dist.all_gather_into_tensor, torch.Tensor.cpu
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    torch.Tensor.cpu = lambda self, *a, **k: (reads.append(1), orig_cpu(self, *a, **k))[1]
    res = run_sharded(_fake_infer, clips, None, gather=True, lengths=[seq_length(16000 + 640 * i) for i in range(n_clips)])
    dist.all_gather_into_tensor, torch.Tensor.cpu = orig, orig_cpu
    if len(calls) != 1 or reads:
        q.put((rank, False))
        return
    # only max_frames known: the lengths are exchanged too and read back (the synchronising fallback)
    clips2 = _LazyClips(n_clips)
    res2 = run_sharded(_fake_infer, clips2, None, gather=True, max_frames=seq_length(16000 + 640 * (n_clips - 1)))
    if not all(torch.equal(a, b) for a, b in zip(res, res2)):
        q.put((rank, False))
        return
    want = _fake_infer([torch.full((16000 + 640 * i,), float(i + 1)) for i in range(n_clips)], None)
    ok = len(res) == n_clips and all(torch.equal(r, w) for r, w in zip(res, want))
    ok = ok and sorted(clips.built) == list(shard_range(n_clips, rank, world))       # nothing but the local shard was built
    q.put((rank, ok))
    dist.destroy_process_group()


def _worker_single(rank, world, port, n_clips, q):
    from artalk_amd.dist import init_single_process_group
    os.environ["MASTER_PORT"] = str(port)
    init_single_process_group("gloo")
    audios = [torch.full((16000 + 640 * i,), float(i + 1)) for i in range(n_clips)]
    calls = []
    orig = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    res = run_sharded(_fake_infer, audios, None, gather=True, force_collective=True)      # lengths derived from the sample counts
    dist.all_gather_into_tensor = orig
    want = _fake_infer(audios, None)
    ok = len(calls) == 1 and all(torch.equal(r, w) for r, w in zip(res, want))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_lazy_clip_list_world2():
    _run(7, _worker_lazy)


def test_force_collective_single_rank():
    """bench.py --force-collective: a world-size-1 group still runs the all-gather (here on gloo; the -m gpu test runs
    the same code on RCCL)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_single, args=(0, 1, _free_port(), 3, q))
    p.start()
    assert q.get(timeout=120) == (0, True)
    p.join(timeout=60)


def test_seq_length_is_the_models_formula():
    """The gather buffer is sized with the float formula of app/models.py:66, not an integer ceil (they differ by one frame
    where N/16000*25 rounds up in floating point)."""
    import math
    for n in (1, 639, 640, 641, 64000, 64640, 160000, 221184, 16000 * 13 + 7):
        assert seq_length(n) == math.ceil(n / 16000 * 25.0)


def test_shard_range_partitions():
    for n in (1, 5, 32, 33, 256):
        for w in (1, 2, 4, 8):
            idx = [i for r in range(w) for i in shard_range(n, r, w)]
            assert idx == list(range(n))
            sizes = [len(shard_range(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def test_gather_world2_even():
    _run(6)


def test_gather_world2_ragged():
    _run(5)     # shards of 2 and 3 clips with different lengths


def test_gather_single_process_is_identity():
    clips = [torch.ones(3, 106), torch.zeros(5, 106)]
    out = gather_clips(clips, 5)
    assert all(torch.equal(a, b) for a, b in zip(out, clips))


def test_bench_self_launch_relays_children_failure():
    """``python bench.py --gpus 2`` from a GPU-free parent: the parent must not touch the GPU, start the two ranks through
    ``torch.distributed.run`` on 127.0.0.1 and relay their return code.  Where there are not two GPUs (this container has none, a
    one-GPU box has one) the ranks fail at device selection: the parent's exit status must be non-zero and its log must show the
    launch line - a parent that swallowed the failure, or tried to initialise the GPU itself, would not."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs present: the ranks would run the real benchmark")
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                           "--no-alt-mode"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode != 0
    assert "torch.distributed.run" in proc.stderr and "--nproc-per-node=2" in proc.stderr and "127.0.0.1" in proc.stderr
    assert '"metric"' not in proc.stdout


def _short_infer(audios, styles):
    """An infer_fn that returns one frame less than the host formula says (a bug the gather must not hide)."""
    return [o[:-1] for o in _fake_infer(audios, styles)]


def _worker_mismatch(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    audios = [torch.full((16000 + 640 * i,), float(i + 1)) for i in range(n_clips)]
    ok = False
    try:
        # run_sharded derives every clip's frame count from its sample count on the host; gather_clips slices the gathered buffer by those
        # counts - if infer_fn returned something else, rows would be truncated or zero rows returned without this check
        run_sharded(_short_infer, audios, None, gather=True)
    except AssertionError as e:
        ok = "host-derived length" in str(e)
    q.put((rank, ok))
    dist.destroy_process_group()


def test_gather_rejects_lengths_that_differ_from_the_host_formula():
    _run(5, _worker_mismatch)


def test_as_rank_walks_the_shards_of_a_larger_job_locally():
    """run_sharded(as_rank=(r, W)): the shard rank r of a W-rank job computes, through the same local path, without a process group (what
    tests/test_configs_gpu.py::test_config3_workload_on_one_gpu does with the 8 x 32 clips of BASELINE configs[3] on one GPU)."""
    audios = [torch.full((16000 + 640 * i,), float(i + 1)) for i in range(11)]
    want = _fake_infer(audios, None)
    seen = []
    for r in range(4):
        mine = shard_range(11, r, 4)
        outs = run_sharded(_fake_infer, audios, None, gather=False, as_rank=(r, 4))
        assert len(outs) == len(mine)
        for k, i in enumerate(mine):
            assert torch.equal(outs[k], want[i])
        seen += list(mine)
    assert seen == list(range(11))
    import pytest
    with pytest.raises(AssertionError):
        run_sharded(_fake_infer, audios, None, gather=True, as_rank=(0, 4))


# ---- 8-rank readiness without hardware: BASELINE configs[3]'s partition on gloo, world size 8 ----
class _Cfg3Clips:
    """configs[3]: clips of equal length (the real ones are 160 000 samples; 1 600 here keep the eight CPU ranks within seconds - the
    partition, the padding and the collective do not depend on the length).  Lazy: only the local shard may be built."""
    def __init__(self, n, ragged=False):
        self.n, self.ragged, self.built = n, ragged, []

    def __len__(self):
        return self.n

    def n_samples(self, i):
        return 1600 + (640 * (i % 5) if self.ragged else 0)

    def __getitem__(self, i):
        self.built.append(i)
        return torch.full((self.n_samples(i),), float(i + 1))


def _worker8(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    try:
        for ragged in (False, True):
            clips = _Cfg3Clips(n_clips, ragged)
            lengths = [seq_length(clips.n_samples(i)) for i in range(n_clips)]
            calls, reads = [], []
            orig, orig_cpu = dist.all_gather_into_tensor, torch.Tensor.cpu
            dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
            torch.Tensor.cpu = lambda self, *a, **k: (reads.append(1), orig_cpu(self, *a, **k))[1]
            res = run_sharded(_fake_infer, clips, None, gather=True, lengths=lengths)
            dist.all_gather_into_tensor, torch.Tensor.cpu = orig, orig_cpu
            ok = ok and len(calls) == 1 and not reads                                      # ONE collective, nothing read back
            ok = ok and sorted(clips.built) == list(shard_range(n_clips, rank, world))      # only the local shard was built
            ok = ok and len(res) == n_clips
            for i in range(n_clips):                                                          # input order, every clip once, right length
                T = lengths[i]
                want0 = float(i + 1)
                ok = ok and tuple(res[i].shape) == (T, 106) and float(res[i][0, 0]) == want0 and float(res[i][T - 1, 105]) == want0 + T - 1
        # the length assert must fire on EVERY rank of an 8-rank job too (a rank that passed it alone would hang the others in the collective)
        clips = _Cfg3Clips(n_clips, True)
        try:
            run_sharded(_short_infer, clips, None, gather=True, lengths=[seq_length(clips.n_samples(i)) for i in range(n_clips)])
            ok = False
        except AssertionError as e:
            ok = ok and "host-derived length" in str(e)
    except Exception as e:      # noqa: BLE001 - reported through the queue
        ok = False
        print(f"rank {rank}: {type(e).__name__}: {e}", flush=True)
    q.put((rank, ok))
    dist.destroy_process_group()


def _run8(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, n_clips, q)) for r in range(8)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, True) for r in range(8)]


def test_config3_partition_world8():
    """BASELINE configs[3]: 256 clips over 8 ranks = 8 shards of 32, rank-major order = input order, one all-gather, no read-back."""
    _run8(256)


def test_config3_ragged_250_clips_world8():
    """250 clips over 8 ranks: shards of 31 and 32 (empty-slot padding in the gather buffer), ragged lengths."""
    sizes = [len(shard_range(250, r, 8)) for r in range(8)]
    assert sorted(set(sizes)) == [31, 32] and sum(sizes) == 250
    _run8(250)


def test_bench_self_launch_relays_failure_at_8_ranks():
    """The same GPU-free-parent relay as above at ``--gpus 8`` (the driver's SCALE run): eight ranks are started on 127.0.0.1 and, where
    there are not eight GPUs, their failure comes back as a non-zero exit status with no result line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("eight GPUs present: the ranks would run the real benchmark")
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                           "--no-alt-mode"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert proc.returncode != 0
    assert "torch.distributed.run" in proc.stderr and "--nproc-per-node=8" in proc.stderr and "127.0.0.1" in proc.stderr
    assert '"metric"' not in proc.stdout
