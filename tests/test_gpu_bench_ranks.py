"""bench.py's N > 1 path, rehearsed: two fresh rank processes (launched by torch.distributed.run exactly as the driver
launches the scaling runs) share the test box's one GPU, with gloo standing in for RCCL as the transport of the gradient
all-reduce.  Everything else is the real thing: the per-rank batches, the captured / pipelined schedules with the
collective between sub-graphs, barrier + max-over-ranks timing, the JSON line.  What this cannot cover is RCCL itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
           "--warmup", "1", "--batch", "8", "--no-cpu-baseline"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]   # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["default", "teacher_ahead_1", "eager"])
def test_bench_two_ranks_over_gloo(mode):
    extra = {"default": [], "teacher_ahead_1": ["--teacher-ahead", "1"], "eager": ["--no-graph"]}[mode]
    r = _run(extra, 29300 + os.getpid() % 500 + {"default": 0, "teacher_ahead_1": 1, "eager": 2}[mode])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2
    assert r["scaling"] == "weak" and r["config"]["per_gpu_batch"] == 8 and r["config"]["global_batch"] == 16
    assert r["config"]["parallelism"] == "dp2" and "gloo" in r["backend"]
    assert r["value"] > 0 and abs(r["value"] - 16 * 3 / (r["ms_per_step"] * 3e-3)) < 1e-6 * r["value"]
    assert r["allreduce_ms"] > 0
    sched = r["config"]["schedule"]
    assert "fell back" not in sched, sched
    if mode == "eager":
        assert sched == "eager"
    else:
        assert "hipGraph replay" in sched and "gradient all-reduce between sub-graphs (never captured)" in sched
        assert ("teacher two batches ahead" in sched) == (mode == "default")
    assert 0 < r["student_bpd"] < 20
    for k in ("roofline", "cpu_baseline", "oracle_check", "f32_parity_mode"):   # rank-0, N=1-only objects
        assert k not in r
