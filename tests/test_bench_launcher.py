"""CPU: `bench.py --gpus N` without an outer launcher starts N rank processes itself (one per GPU on the GPU box; the reference does the
same through torch.distributed.launch: scripts/run_SI.sh:6, tools/train_incremental.py:406-409).  `--rendezvous-only` exercises exactly
that machinery -- child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, process-group formation over 127.0.0.1, one
all-reduce of a 1 per rank, rank 0's JSON line, the children's exit status -- with the gloo backend when there is no GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n", [2, 3])
def test_bench_starts_its_own_ranks(n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--rendezvous-only"], env=env, capture_output=True,
                         text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout            # rank 0 only
    d = json.loads(lines[0])
    assert d["rendezvous_only"] is True and d["ranks"] == n


@pytest.mark.timeout(120)
def test_bench_under_an_outer_launcher_does_not_spawn():
    """WORLD_SIZE already set (torch.distributed.run / the driver's launcher): bench.py must NOT start processes; with WORLD_SIZE=1 and
    --gpus 1 the rendezvous self-test is a plain single-process run"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rendezvous-only"], env=env, capture_output=True,
                         text=True, timeout=100)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["ranks"] == 1 and d["backend"] is None
