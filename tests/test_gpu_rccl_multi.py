"""The sharded factorisation over REAL RCCL with one process per GPU.  Skipped unless at least two GPUs are visible: the
builder's boxes have one (the thread-rank transport of test_gpu_dist.py / test_gpu_dist_guard.py and the gloo replay cover
the schedules there), so this is the file a multi-GPU node runs first:
  * the three schedules at a small size against the oracle,
  * BASELINE configs[3] at FULL size -- N = 32768, d = 16, RBF, 512-column panels over ALL visible GPUs, the chain-first
    schedule -- every rank's factor against a single-rank factor of the same rows on its own GPU, split queries,
  * a rank that goes missing in the middle of a factorisation: the library's time-out, the rebuilt communicator, the
    conservative schedule.
FRIEDRICH_TEST_RCCL_WORLD=1 runs the workers with a single rank (a 1-GPU box: the code of the workers, not the transport)."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, rand_inputs, rel_err

pytestmark = pytest.mark.gpu


def _world():
    torch = pytest.importorskip("torch")
    forced = int(os.environ.get("FRIEDRICH_TEST_RCCL_WORLD", "0"))
    if forced > 0:
        return forced
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one process per GPU over RCCL)")
    return torch.cuda.device_count()


def _setup(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime

    import torch
    import torch.distributed as dist

    from friedrich_amd import sharding
    from friedrich_amd.device import Context

    torch.cuda.set_device(rank)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", rank))
    ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=600))
    ctx = Context(rank)
    link = sharding.TorchLink(dist, rank, world, ctl)
    link.attach(ctx)
    return torch, dist, ctx, link


def _teardown(dist, ctx):
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, n, nb, split, out_dir):
    torch, dist, ctx, link = _setup(rank, world, port)
    ctx.set_option("nb", nb)
    ctx.set_option("dist_schedule", split)
    ctx.set_option("comm_timeout_ms", 60000)
    k = ("matern2", 0.7, 1.2)
    X = rand_inputs(n, 5, n)
    chol = ctx.cholesky_from_inputs(k, X, 0.1)
    chol.refactor(k, 0.1)  # back to back: the buffers and event rings of one factorisation reused by the next
    np.save(os.path.join(out_dir, f"L{rank}.npy"), chol.l())
    chol.free()
    _teardown(dist, ctx)


@pytest.mark.parametrize("split", [0, 1, 2], ids=["bcast", "split", "chain"])
def test_rccl_processes_sharded_factor(tmp_path, split):
    world = min(_world(), 4)
    import torch.multiprocessing as mp

    from oracle import oracle as O

    n, nb = 2100, 256
    port = 33500 + (os.getpid() % 2000) + split
    mp.spawn(_worker, args=(world, port, n, nb, split, str(tmp_path)), nprocs=world, join=True)
    X = rand_inputs(n, 5, n)
    st, L_o, _ = O.make_cholesky_cov_matrix(("matern2", 0.7, 1.2), X, 0.1)
    for r in range(world):
        assert rel_err(np.load(tmp_path / f"L{r}.npy"), np.tril(L_o)) < 1e-9


def _worker_full(rank, world, port, n, schedule, out_dir):
    """configs[3] as it is sharded; every rank compares ON ITS GPU with a single-rank factor of the same build (which
    tests/test_gpu_fullsize_oracle.py compares with the oracle at this size)"""
    import ctypes

    torch, dist, ctx, link = _setup(rank, world, port)
    from friedrich_amd import sharding, synth
    from friedrich_amd.device import Context

    d, m = 16, 512
    X, y, Xq = synth.make_problem(n, d, cfg=3, m=m)
    ref_ctx = Context(rank)
    ls = ref_ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    dev = torch.device("cuda", rank)

    def factor_on_device(chol):
        buf = torch.empty((n, n), dtype=torch.float64, device=dev).t()
        chol.ctx.check(chol.lib.fr_chol_download_l(chol.h, ctypes.c_void_p(buf.data_ptr()), n, 0))
        chol.ctx.synchronize()
        return buf

    ref = ref_ctx.cholesky_from_inputs(k, X, hp["noise"])
    Lref = factor_on_device(ref)
    yres = y - hp["prior"]
    lo, hi = sharding.query_slice(m, rank, world)
    mean_ref = ref.predict_mean(k, yres, Xq[lo:hi], np.full(hi - lo, hp["prior"]))
    ref.free()
    ctx.set_option("nb", 512)
    ctx.set_option("dist_schedule", schedule)
    ctx.set_option("comm_timeout_ms", 120000)
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"])
    chol.refactor(k, hp["noise"])
    L = factor_on_device(chol)
    err = float((L - Lref).abs().max() / Lref.abs().max())
    mean = chol.predict_mean(k, yres, Xq[lo:hi], np.full(hi - lo, hp["prior"]))
    info = chol.info()
    with open(os.path.join(out_dir, f"r{rank}.json"), "w") as f:
        json.dump({"err": err, "mean_err": rel_err(mean, mean_ref), "info": info, "timeouts": ctx.counter("comm_timeouts")}, f)
    chol.free()
    ref_ctx.close()
    _teardown(dist, ctx)


@pytest.mark.parametrize("schedule", [2, 1], ids=["chain", "split"])
def test_rccl_config3_full_size_over_all_gpus(tmp_path, schedule):
    world = _world()
    import torch.multiprocessing as mp

    n = int(os.environ.get("FRIEDRICH_TEST_RCCL_N", "32768"))
    port = 35500 + (os.getpid() % 2000) + schedule
    mp.spawn(_worker_full, args=(world, port, n, schedule, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = json.load(open(tmp_path / f"r{r}.json"))
        assert res["err"] < 1e-12, res
        assert res["mean_err"] < 1e-10, res
        assert res["info"]["n_subst"] == 0 and res["info"]["fail_col"] == -1 and res["timeouts"] == 0, res


def _worker_hang(rank, world, port, out_dir):
    torch, dist, ctx, link = _setup(rank, world, port)
    from friedrich_amd import sharding
    from friedrich_amd.device import Context

    ref = Context(rank)
    schedule, reasons, took = sharding.guarded_schedule(ctx, link, lambda s: sharding.preflight_fit(ctx, ref, n=4096), timeout_ms=5000)
    ref.close()
    with open(os.path.join(out_dir, f"r{rank}.json"), "w") as f:
        json.dump({"schedule": schedule, "reasons": reasons, "timeouts": ctx.counter("comm_timeouts")}, f)
    _teardown(dist, ctx)


def test_rccl_missing_rank_falls_back(tmp_path, monkeypatch):
    """the last rank skips a collective of the schedule-2 preflight: its peers' collectives wait on the device, the polled
    synchronisation runs out, the communicators are aborted and rebuilt, schedule 1 passes"""
    world = _world()
    if world < 2:
        pytest.skip("a missing peer needs a peer")
    import torch.multiprocessing as mp

    monkeypatch.setenv("FRIEDRICH_AMD_TEST_COMM_HANG", f"2,{world - 1},9")
    mp.spawn(_worker_hang, args=(world, 36500 + (os.getpid() % 2000), str(tmp_path)), nprocs=world, join=True)
    res = [json.load(open(tmp_path / f"r{r}.json")) for r in range(world)]
    assert all(r["schedule"] == 1 for r in res), res
    assert all(len(r["reasons"]) == 1 and "schedule 2" in r["reasons"][0] for r in res), res
    assert sum(r["timeouts"] for r in res) >= 1
