"""Data-parallel sharding of independent clips over the GPUs of one node.

Clips share nothing but read-only weights and chunks inside a clip are sequential, so the path shards by clip:
one process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm), weights replicated
(2 GB of 288 GB), no data-path collective.  The only exchange is the optional collection of results: one
all-gather of the padded FLAME codes (the frame counts follow from the sample counts on the host, so nothing is
read back and the host never waits for the device at N > 1 either).  The same code runs on the gloo backend with
CPU tensors (tests/test_dist_gloo.py) and is what ``bench.py`` runs for N > 1.
"""
from __future__ import annotations

import math
import os
import socket
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def seq_length(n_samples: int) -> int:
    """Frames of a clip, the float formula of app/models.py:66 (== BitwiseARModel.seq_length)."""
    return math.ceil(n_samples / 16000 * 25.0)


def shard_range(n_items: int, rank: int, world_size: int) -> range:
    """Contiguous block partition: rank r gets items [r*n/W, (r+1)*n/W) (sizes differ by at most one)."""
    lo = (n_items * rank) // world_size
    hi = (n_items * (rank + 1)) // world_size
    return range(lo, hi)


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_single_process_group(backend: str = "nccl", device: Optional[torch.device] = None):
    """A world-size-1 process group (used by ``bench.py --force-collective`` and the GPU test of the RCCL path: the same
    init + all-gather code as N > 1, on one GPU).  Rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, rank=0, world_size=1, **kw)


def gather_clips(local: Sequence[torch.Tensor], max_frames: int, group=None, force_collective: bool = False,
                 all_lengths: Optional[Sequence[int]] = None) -> List[torch.Tensor]:
    """All-gather per-clip results ``(T_i, D)`` of every rank; returns the clips of all ranks in rank order on
    every rank.  Every rank must pass the same number of clips (pad the shard with empty ``(0, D)`` tensors).
    With one rank nothing is exchanged unless ``force_collective`` (then the collective runs over the 1-rank group).

    ``all_lengths`` = frames of all ``world * n`` (padded) slots, known on the host (a clip's frame count follows from its sample
    count): then ONE collective runs and nothing is read back, so the host never waits for the device here and a serving loop
    keeps its batches in flight at N > 1 as at N = 1.  Without it the lengths are all-gathered too and read back (one host
    synchronisation per call)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):
        return list(local)
    n = len(local)
    D = local[0].shape[1]
    dev = local[0].device
    buf = torch.zeros(n, max_frames, D, dtype=local[0].dtype, device=dev)
    for i, t in enumerate(local):
        assert t.shape[0] <= max_frames, f"clip of {t.shape[0]} frames does not fit the gather buffer ({max_frames})"
        buf[i, : t.shape[0]] = t
    all_buf = torch.empty(world * n, max_frames, D, dtype=buf.dtype, device=dev)
    dist.all_gather_into_tensor(all_buf, buf, group=group)
    if all_lengths is not None:
        assert len(all_lengths) == world * n, "all_lengths must cover every (padded) slot of every rank"
        all_len = [int(v) for v in all_lengths]
        # the host-derived counts slice the gathered buffer: they must be what infer_fn really returned for this rank's slots,
        # else rows would be truncated (or zero rows returned) silently.  Host ints on both sides: no synchronisation.
        rank = dist.get_rank(group)
        for i, t in enumerate(local):
            assert int(t.shape[0]) == all_len[rank * n + i], (
                f"rank {rank} slot {i}: infer_fn returned {int(t.shape[0])} frames but the host-derived length is {all_len[rank * n + i]}")
    else:
        lens = torch.tensor([t.shape[0] for t in local], dtype=torch.int64).to(dev)
        all_len = torch.empty(world * n, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(all_len, lens, group=group)
        all_len = all_len.cpu().tolist()         # host synchronisation: pass all_lengths to avoid it
    return [all_buf[i, : all_len[i]] for i in range(world * n)]


def run_sharded(infer_fn, audios: Sequence[torch.Tensor], styles=None, gather: bool = True, group=None,
                max_frames: Optional[int] = None, force_collective: bool = False, lengths: Optional[Sequence[int]] = None,
                as_rank: Optional[tuple] = None):
    """Shard ``audios`` over the ranks, run ``infer_fn(list_of_audio, list_of_style)`` on the local shard and
    (optionally) all-gather.  Result order equals input order; shards are padded to equal size with empties.
    ``audios`` only needs ``len()`` and indexing (a lazy sequence may build just the local clips); pass ``lengths`` (frames of
    EVERY clip of the job, e.g. ``seq_length(n_samples)``) or at least ``max_frames`` in that case.  With ``lengths`` - given, or
    derived here from every clip's sample count when neither is given - the gather is one collective and the host never waits
    for the device (``gather_clips``); with only ``max_frames`` the lengths are exchanged and read back."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if as_rank is not None:
        # ``as_rank = (r, W)``: compute exactly what rank r of a W-rank job computes - its shard through the same local path - without a
        # process group (no exchange: ``gather`` must be False).  One GPU can walk a multi-GPU workload shard by shard this way
        # (tests/test_configs_gpu.py::test_config3_workload_on_one_gpu); it measures nothing about scaling.
        assert not gather, "as_rank runs one shard locally; there is nothing to gather from"
        rank, world = int(as_rank[0]), int(as_rank[1])
        assert 0 <= rank < world
    n = len(audios)
    mine = shard_range(n, rank, world)
    loc_a = [audios[i] for i in mine]
    loc_s = [styles[i] for i in mine] if styles is not None else None
    out = infer_fn(loc_a, loc_s) if loc_a else []
    if not gather or (world == 1 and not force_collective):
        return out
    per = max(len(shard_range(n, r, world)) for r in range(world))
    D = out[0].shape[1] if out else 106
    if out:
        dev = out[0].device
    elif dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device("cpu")
    padded = list(out) + [torch.zeros(0, D, device=dev)] * (per - len(out))
    if lengths is None and max_frames is None:
        lengths = [seq_length(int(audios[i].shape[-1])) for i in range(n)]
    if lengths is not None:
        assert len(lengths) == n
        if max_frames is None:
            max_frames = max(lengths)
    slots = None
    if lengths is not None:      # frames of every padded slot, rank-major (empty slots pad the shorter shards)
        slots = []
        for r in range(world):
            rr = shard_range(n, r, world)
            slots += [int(lengths[i]) for i in rr] + [0] * (per - len(rr))
    allc = gather_clips(padded, max_frames, group, force_collective, slots)
    res = []
    for r in range(world):
        k = len(shard_range(n, r, world))
        res.extend(allc[r * per: r * per + k])
    return res
