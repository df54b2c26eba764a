"""The sharded factorisation over the RCCL branch of comm.hip with peers.

With at least two GPUs visible: REAL RCCL, one process per GPU -- the file a multi-GPU node runs first.  On a 1-GPU box (every box
the builder has had) RCCL refuses two ranks on one device, so the same workers run as THREAD-ranks of one child process whose
`FRIEDRICH_AMD_RCCL_PATH` names tests/mock_rccl/librccl_mock.so -- a test-only stand-in that implements the dozen nccl* entry
points comm.hip resolves (rendezvous in ncclCommInitRank, collectives as stream work ordered by events, host calls that block
until the peer calls or the communicator is aborted, optional device-side waits: rccl_mock.hip).  What that executes with peers
for the first time: CallGuard::enter / leave around every call, GroupGuard, init_rank_bounded with world > 1, ensure_comm2's
hand-shake, the grouped fan-out / scatter, comm_stream_sync's polling, the watchdog's abort of a blocked call.
  * the three schedules at a small size against the oracle,
  * BASELINE configs[3] at FULL size -- N = 32768, d = 16, RBF, 512-column panels, the chain-first and the split schedule -- every
    rank's factor against a single-rank factor of the same rows on its GPU, split queries,
  * a rank that goes missing in the middle of a factorisation: the library's time-out (host-side: the watchdog aborts the blocked
    call; device-side: the polled synchronisation runs out), the rebuilt communicator, the conservative schedule,
  * a rank that never reaches ncclCommInitRank: the bounded rendezvous.
FRIEDRICH_TEST_RCCL_WORLD=1 runs the process workers with a single rank (the code of the workers, not the transport)."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import ROOT, rand_inputs, rel_err

pytestmark = pytest.mark.gpu


def _gpus():
    torch = pytest.importorskip("torch")
    return torch.cuda.device_count()


def _mode():
    """-> ("process", world) with >= 2 GPUs (or FRIEDRICH_TEST_RCCL_WORLD forced), else ("mock", None)"""
    forced = int(os.environ.get("FRIEDRICH_TEST_RCCL_WORLD", "0"))
    if forced > 0:
        return "process", forced
    n = _gpus()
    return ("process", n) if n >= 2 else ("mock", None)


# ---- set-up of one rank: a process over real RCCL, or a thread over the mock ------------------------------------------------------
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
    return ctx, link, rank, lambda: (ctx.close(), dist.barrier(), dist.destroy_process_group())


class _MockLink:
    """thread-rank whose communicator goes through fr_ctx_comm_init (the RCCL branch) -- the mock supplies the peers"""

    def __new__(cls, shared, rank):
        from friedrich_amd import sharding

        class Link(sharding.ThreadLink):
            def attach(self, ctx):
                ids = self.gather(ctx.comm_unique_id() if self.rank == 0 else None)
                ctx.comm_init(self.rank, self.world, ids[0])
                ctx.comm_selftest()

        return Link(shared, rank)


def _setup_mock(rank, shared, timeout_ms=None):
    from friedrich_amd.device import Context

    ctx = Context(0)
    if timeout_ms is not None:
        ctx.set_option("comm_timeout_ms", timeout_ms)
    link = _MockLink(shared, rank)
    link.attach(ctx)
    return ctx, link, 0, ctx.close


def _launch(worker, world_wanted, args, env=None, mock_world=None):
    """run worker(rank, world, setup, *args) on every rank: processes over real RCCL, or a child process of thread-ranks over the mock"""
    mode, gpus = _mode()
    if mode == "process":
        import torch.multiprocessing as mp

        world = min(gpus, world_wanted) if world_wanted else gpus
        port = 33500 + (os.getpid() % 2000) + (hash(worker.__name__) % 500)
        for k, v in (env or {}).items():
            os.environ[k] = v.replace("{last}", str(world - 1))
        try:
            mp.spawn(_process_entry, args=(world, port, worker.__name__, args), nprocs=world, join=True)
        finally:
            for k in (env or {}):
                os.environ.pop(k, None)
        return world
    sys.path.insert(0, os.path.join(ROOT, "tests", "mock_rccl"))
    import build_mock

    lib = build_mock.build()
    world = mock_world or world_wanted or 4
    child_env = dict(os.environ, FRIEDRICH_AMD_RCCL_PATH=lib, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    for k, v in (env or {}).items():
        child_env[k] = v.replace("{last}", str(world - 1))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--mock-worker", worker.__name__, str(world), json.dumps(args)],
                       env=child_env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"mock-RCCL child failed ({r.returncode}):\n{r.stdout[-3000:]}\n{r.stderr[-6000:]}"
    return world


def _process_entry(rank, world, port, name, args):
    globals()[name](rank, world, lambda timeout_ms=None: _setup(rank, world, port), *args)


def _mock_main(name, world, args):
    from friedrich_amd import sharding

    shared = sharding.ThreadShared(world)
    errors = []

    def run(r):
        try:
            globals()[name](r, world, lambda timeout_ms=None: _setup_mock(r, shared, timeout_ms), *args)
        except BaseException as e:  # noqa: BLE001
            import traceback

            traceback.print_exc()
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(1 if errors else 0)  # (helper threads abandoned inside a bounded rendezvous must not meet the interpreter's teardown)


def _worker(rank, world, setup, n, nb, split, out_dir):
    ctx, link, dev_index, teardown = setup()
    ctx.set_option("nb", nb)
    ctx.set_option("dist_schedule", split)
    ctx.set_option("comm_timeout_ms", 60000)
    k = ("matern2", 0.7, 1.2)
    X = rand_inputs(n, 5, n)
    chol = ctx.cholesky_from_inputs(k, X, 0.1)
    chol.refactor(k, 0.1)  # back to back: the buffers and event rings of one factorisation reused by the next
    np.save(os.path.join(out_dir, f"L{rank}.npy"), chol.l())
    with open(os.path.join(out_dir, f"c{rank}.json"), "w") as f:
        json.dump({"world": ctx.comm_info()["world"] if hasattr(ctx, "comm_info") else world, "timeouts": ctx.counter("comm_timeouts")}, f)
    chol.free()
    teardown()


@pytest.mark.parametrize("split", [0, 1, 2], ids=["bcast", "split", "chain"])
def test_rccl_processes_sharded_factor(tmp_path, split):
    from oracle import oracle as O

    n, nb = 2100, 256
    world = _launch(_worker, 4, [n, nb, split, str(tmp_path)], mock_world=3)
    X = rand_inputs(n, 5, n)
    st, L_o, _ = O.make_cholesky_cov_matrix(("matern2", 0.7, 1.2), X, 0.1)
    for r in range(world):
        assert rel_err(np.load(tmp_path / f"L{r}.npy"), np.tril(L_o)) < 1e-9
        assert json.load(open(tmp_path / f"c{r}.json"))["timeouts"] == 0


def _worker_full(rank, world, setup, n, schedule, out_dir):
    """configs[3] as it is sharded; every rank compares ON ITS GPU with a single-rank factor of the same build (which
    tests/test_gpu_fullsize_oracle.py compares with the oracle at this size)"""
    import ctypes

    import torch

    ctx, link, dev_index, teardown = setup()
    from friedrich_amd import sharding, synth
    from friedrich_amd.device import Context

    d, m = 16, 512
    X, y, Xq = synth.make_problem(n, d, cfg=3, m=m)
    ref_ctx = Context(dev_index)
    ls = ref_ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    dev = torch.device("cuda", dev_index)

    def factor_on_device(chol):
        buf = torch.empty((n, n), dtype=torch.float64, device=dev).t()
        chol.ctx.check(chol.lib.fr_chol_download_l(chol.h, ctypes.c_void_p(buf.data_ptr()), n, 0))
        chol.ctx.synchronize()
        return buf

    ref_ctx.set_option("nb", 512)
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
    # (column blocks: thread-ranks share ONE GPU in mock mode -- eight ranks x three 8 GiB matrices leave no room for 8 GiB temporaries)
    diff = scale = 0.0
    for c0 in range(0, n, 2048):
        a, b = L[:, c0:c0 + 2048], Lref[:, c0:c0 + 2048]
        diff = max(diff, float((a - b).abs().max()))
        scale = max(scale, float(b.abs().max()))
    err = diff / scale
    mean = chol.predict_mean(k, yres, Xq[lo:hi], np.full(hi - lo, hp["prior"]))
    info = chol.info()
    with open(os.path.join(out_dir, f"r{rank}.json"), "w") as f:
        json.dump({"err": err, "mean_err": rel_err(mean, mean_ref), "info": info, "timeouts": ctx.counter("comm_timeouts")}, f)
    chol.free()
    del L, Lref
    ref_ctx.close()
    teardown()


@pytest.mark.parametrize("schedule", [2, 1], ids=["chain", "split"])
def test_rccl_config3_full_size_over_all_gpus(tmp_path, schedule):
    n = int(os.environ.get("FRIEDRICH_TEST_RCCL_N", "32768"))
    world = _launch(_worker_full, 0, [n, schedule, str(tmp_path)], mock_world=8)
    for r in range(world):
        res = json.load(open(tmp_path / f"r{r}.json"))
        assert res["err"] < 1e-12, res
        assert res["mean_err"] < 1e-10, res
        assert res["info"]["n_subst"] == 0 and res["info"]["fail_col"] == -1 and res["timeouts"] == 0, res


def _worker_hang(rank, world, setup, out_dir):
    ctx, link, dev_index, teardown = setup()
    from friedrich_amd import sharding
    from friedrich_amd.device import Context

    ref = Context(dev_index)
    schedule, reasons, took = sharding.guarded_schedule(ctx, link, lambda s: sharding.preflight_fit(ctx, ref, n=4096), timeout_ms=5000)
    ref.close()
    with open(os.path.join(out_dir, f"r{rank}.json"), "w") as f:
        json.dump({"schedule": schedule, "reasons": reasons, "timeouts": ctx.counter("comm_timeouts")}, f)
    teardown()


@pytest.mark.parametrize("where", ["host", "device"])
def test_rccl_missing_rank_falls_back(tmp_path, where):
    """the last rank skips a collective of the schedule-2 preflight.  "host": its peers' matching RCCL calls do not return -- the
    watchdog aborts them through CallGuard; "device" (mock only: RCCL_MOCK_RENDEZVOUS_MS): their collectives wait on the device,
    the polled synchronisation (comm_stream_sync) runs out.  Either way the communicators are aborted and rebuilt, schedule 1 passes"""
    mode, gpus = _mode()
    if mode == "process" and gpus < 2:
        pytest.skip("a missing peer needs a peer")
    if mode == "process" and where == "device":
        pytest.skip("real RCCL decides by itself where a call with a missing peer waits: one variant")
    env = {"FRIEDRICH_AMD_TEST_COMM_HANG": "2,{last},9"}
    if mode == "mock" and where == "device":
        env["RCCL_MOCK_RENDEZVOUS_MS"] = "50"
    world = _launch(_worker_hang, 0, [str(tmp_path)], env=env, mock_world=3)
    res = [json.load(open(tmp_path / f"r{r}.json")) for r in range(world)]
    assert all(r["schedule"] == 1 for r in res), res
    assert all(len(r["reasons"]) == 1 and "schedule 2" in r["reasons"][0] for r in res), res
    assert sum(r["timeouts"] for r in res) >= 1


def _worker_init_absent(rank, world, setup, out_dir):
    from friedrich_amd.device import FriedrichError

    import time

    t0 = time.perf_counter()
    try:
        ctx, link, dev_index, teardown = setup(timeout_ms=3000)
        outcome = "attached"
        teardown()
    except (FriedrichError, threading.BrokenBarrierError) as e:
        outcome = f"{type(e).__name__}: {e}"
    with open(os.path.join(out_dir, f"r{rank}.json"), "w") as f:
        json.dump({"outcome": outcome, "seconds": time.perf_counter() - t0}, f)


def test_rccl_init_rendezvous_is_bounded(tmp_path):
    """one rank never completes ncclCommInitRank (mock: RCCL_MOCK_INIT_ABSENT_RANK): no ncclCommAbort can reach a communicator that
    does not exist yet, so init_rank_bounded's deadline has to end the peers' wait -- FR_RCCL_ERROR after comm_timeout_ms, not a hang"""
    mode, _ = _mode()
    if mode != "mock":
        pytest.skip("needs the mock's absent-rank switch")
    world = _launch(_worker_init_absent, 0, [str(tmp_path)], env={"RCCL_MOCK_INIT_ABSENT_RANK": "{last}"}, mock_world=3)
    res = [json.load(open(tmp_path / f"r{r}.json")) for r in range(world)]
    assert all("attached" not in r["outcome"] for r in res), res
    assert all(r["seconds"] < 25 for r in res), res
    assert any("FriedrichError" in r["outcome"] for r in res), res


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[1] == "--mock-worker":
        _mock_main(sys.argv[2], int(sys.argv[3]), json.loads(sys.argv[4]))
