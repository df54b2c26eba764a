"""bench.py's output contract, on a reduced workload: one JSON line with the fields the driver reads (metric, value, unit,
n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload, roofline,
cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--n", "3072",
           "--m", "256", "--cpu-sample-n", "512"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    sb = d["step_breakdown_ms"]  # one untimed step with events around every launch: where a step's time goes
    assert sb["syrk"] > 0 and sb["potf2"] > 0 and sb["gemm_panel"] > 0 and sb["comm"] == 0.0
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0


def test_bench_multi_process_setup_with_one_rank():
    """FRIEDRICH_BENCH_FORCE_DIST=1: the N > 1 set-up of bench.py -- torch.distributed process group over RCCL, ncclUniqueId
    hand-over, both communicators of the context, the collective self-test (broadcast, all-gather, fan-out, bulk-stream
    all-gather), barrier + MAX-over-ranks timing -- with the single rank a 1-GPU box has.  A multi-GPU node runs the same code
    with WORLD_SIZE > 1 (tests/test_gpu_rccl_multi.py covers the sharded factor there)."""
    env = dict(os.environ, FRIEDRICH_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--n", "3072",
           "--m", "256", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "strong"
