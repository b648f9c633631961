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


def _run_bench(args, timeout):
    import subprocess
    import sys
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    return r, time.time() - t0


def test_bench_launcher_forms_an_8_rank_group():
    """`python bench.py --gpus 8` without an outer launcher (what the driver's 8-GPU run may do): eight fresh rank processes, one
    rendezvous on 127.0.0.1, every rank counted (gloo here; RCCL when eight GPUs are visible)."""
    import json
    r, _ = _run_bench(["--gpus", "8", "--rendezvous-only"], 600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"rendezvous_only": True, "ranks": 8, "backend": "gloo"}


def test_bench_launcher_reaps_siblings_when_a_rank_dies_after_rendezvous():
    """A rank that raises after the process group exists leaves the others inside a collective: launch_ranks must notice the dead child,
    end the survivors and return non-zero, instead of waiting with them (tools/train_incremental.py:406-411 relies on the launcher too)."""
    r, took = _run_bench(["--gpus", "3", "--rendezvous-only", "--inject-failure", "1"], 300)
    assert r.returncode != 0
    assert "injected failure on rank 1" in r.stderr
    assert took < 120, took   # ended by the launcher, not by a collective timeout (gloo's default is 30 minutes)
    import subprocess
    left = subprocess.run(["pgrep", "-f", "bench.py --gpus 3 --rendezvous-only --inject-failure"], capture_output=True, text=True).stdout.split()
    assert not left, left   # no orphaned rank processes
