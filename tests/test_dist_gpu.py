"""The N > 1 code path of bench.py on ONE GPU: ``bench.py --force-collective`` creates the RCCL process group
(``init_process_group("nccl")``, world size 1) and runs the same ``run_sharded`` + ``all_gather_into_tensor`` result
collection a multi-GPU run uses.  Run as a child process, as the driver runs bench.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_force_collective_runs_rccl_all_gather():
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--force-collective", "--config", "tiny", "--batch", "3",
           "--seconds", "6.3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-alt-mode"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["value"] > 0
    assert "RCCL all_gather_into_tensor" in r["config"]["collective"]
    assert r["config"]["clips_per_gpu"] == 3 and r["config"]["frames_per_clip"] == 158


def test_bench_gpus_mismatch_is_refused():
    """Under torchrun (WORLD_SIZE set) --gpus must equal the world size."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "does not match WORLD_SIZE" in p.stderr
