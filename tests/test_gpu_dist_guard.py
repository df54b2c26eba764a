"""The guards around a sharded run: bounded waits inside the library (option comm_timeout_ms -> FR_RCCL_ERROR instead of a
hang), recovery of the context (fr_ctx_comm_finalize + a fresh communicator), the schedule fall-back of
friedrich_amd.sharding.guarded_schedule, and bench.py's N > 1 path -- preflight, fall-back 2 -> 1 -> 0, per-rank breakdown --
driven with thread-ranks on the one GPU of a test box.  The missing rank is injected with FRIEDRICH_AMD_TEST_COMM_HANG
("schedule,rank,nth": that rank skips its nth collective under that schedule and stalls, like a crashed peer)."""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

from conftest import ROOT, rand_inputs, rel_err
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def run_thread_ranks(world, fn):
    from friedrich_amd import sharding
    from friedrich_amd.device import Context

    shared = sharding.ThreadShared(world)
    results, errors = [None] * world, [None] * world

    def worker(rank):
        ctx = None
        try:
            ctx = Context()
            link = sharding.ThreadLink(shared, rank)
            link.attach(ctx)
            results[rank] = fn(ctx, link)
        except BaseException as e:  # noqa: BLE001
            errors[rank] = e
            shared.barrier.abort()
        finally:
            if ctx is not None:
                ctx.close()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    for e in errors:
        if e is not None:
            raise e
    return results


@pytest.mark.parametrize("sync", ["0", "1"], ids=["event-ordered", "host-synchronised"])
@pytest.mark.parametrize("schedule", [0, 1, 2])
def test_local_transport_modes_agree_with_oracle(monkeypatch, schedule, sync):
    """the in-process transport orders its copies by events on the ranks' streams (default) -- a missing dependency between the
    chain / bulk / main streams of one rank would show as a wrong factor -- or synchronises around every collective"""
    monkeypatch.setenv("FRIEDRICH_AMD_LOCAL_SYNC", sync)
    n, nb, world = 2300, 256, 3
    k = ("matern2", 0.7, 1.2)
    X = rand_inputs(n, 5, 11)
    _, L_o, _ = O.make_cholesky_cov_matrix(k, X, 0.1)

    def fn(ctx, link):
        ctx.set_option("nb", nb)
        ctx.set_option("dist_schedule", schedule)
        out = []
        chol = ctx.cholesky_from_inputs(k, X, 0.1)
        for _ in range(3):  # back-to-back factorisations: buffers and events of one reused by the next
            chol.refactor(k, 0.1)
        out = chol.l()
        chol.free()
        return out

    for L in run_thread_ranks(world, fn):
        assert rel_err(L, np.tril(L_o)) < 1e-9


def test_missing_rank_times_out_and_the_context_recovers(monkeypatch):
    """rank 1 skips a collective in the middle of a schedule-2 factorisation: every rank gets FR_RCCL_ERROR within the
    time-out instead of waiting for ever, the contexts are 'lost' until finalized, and with a fresh communicator (and the
    conservative schedule) the same contexts factor correctly"""
    from friedrich_amd import _capi
    from friedrich_amd.device import FriedrichError

    monkeypatch.setenv("FRIEDRICH_AMD_TEST_COMM_HANG", "2,1,6")
    n, nb, world = 1500, 128, 3
    k = ("squared_exp", 0.9, 1.1)
    X = rand_inputs(n, 4, 21)
    _, L_o, _ = O.make_cholesky_cov_matrix(k, X, 0.1)

    def fn(ctx, link):
        ctx.set_option("nb", nb)
        ctx.set_option("dist_schedule", 2)
        ctx.set_option("comm_timeout_ms", 1500)
        t0 = time.perf_counter()
        with pytest.raises(FriedrichError) as ei:
            ctx.cholesky_from_inputs(k, X, 0.1)
        waited = time.perf_counter() - t0
        assert ei.value.status == _capi.FR_RCCL_ERROR
        # lost: nothing sharded works on this context until it is finalized -- and it fails at once, not after a time-out
        t0 = time.perf_counter()
        with pytest.raises(FriedrichError):
            ctx.cholesky_from_inputs(k, X, 0.1)
        assert time.perf_counter() - t0 < 1.0
        timeouts = ctx.counter("comm_timeouts")
        link.barrier()
        ctx.comm_finalize(abort=True)
        # single-rank again: the context works on its own
        solo = ctx.cholesky_from_inputs(k, X, 0.1)
        L_solo = solo.l()
        solo.free()
        link.attach(ctx)
        ctx.set_option("dist_schedule", 1)
        chol = ctx.cholesky_from_inputs(k, X, 0.1)
        L = chol.l()
        chol.free()
        return waited, timeouts, L_solo, L

    res = run_thread_ranks(world, fn)
    assert all(w < 10.0 for w, _, _, _ in res), [r[0] for r in res]
    assert sum(t for _, t, _, _ in res) >= 1
    for _, _, L_solo, L in res:
        assert rel_err(L_solo, np.tril(L_o)) < 1e-9
        assert rel_err(L, np.tril(L_o)) < 1e-9


@pytest.mark.parametrize("hang,expect", [("2,0,9", 1), ("", 2)], ids=["schedule 2 hangs", "nothing hangs"])
def test_guarded_schedule_falls_back(monkeypatch, hang, expect):
    from friedrich_amd import sharding
    from friedrich_amd.device import Context

    if hang:
        monkeypatch.setenv("FRIEDRICH_AMD_TEST_COMM_HANG", hang)
    world = 4

    def fn(ctx, link):
        ref = Context()
        try:
            return sharding.guarded_schedule(ctx, link, lambda s: sharding.preflight_fit(ctx, ref, n=1536, d=4), timeout_ms=2000)
        finally:
            ref.close()

    for schedule, reasons, took in run_thread_ranks(world, fn):
        assert schedule == expect
        assert (len(reasons) == 1 and "schedule 2" in reasons[0]) if hang else reasons == []
        assert expect in took


def _run_bench(extra_args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--n", "3072", "--m", "256",
           "--no-cpu-baseline", "--no-extras", "--verify", "--preflight-n", "1536", "--preflight-timeout-ms", "3000"] + extra_args
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0]), out.stderr


def test_bench_thread_ranks_reach_the_sharded_schedule():
    """bench.py's N > 1 path (round 3 ran it with WORLD_SIZE = 1 only, where the factorisation is the single-GPU one): four
    thread-ranks, schedule 2 preflighted and used, one line with the per-rank breakdown"""
    d, _ = _run_bench(["--local-ranks", "4"])
    assert d["schedule_used"] == 2 and d["fallback_reason"] is None
    assert d["value"] > 0 and d["scaling"] == "strong" and "thread-ranks" in d["config"]["parallelism"]
    pr = d["per_rank"]
    for key in ("comm_ms", "potf2_ms", "panel_ms", "syrk_ms", "fit_ms", "predict_ms"):
        assert len(pr[key]) == 4, key
    assert all(c > 0 for c in pr["comm_calls"]) and all(t == 0 for t in pr["comm_timeouts"])
    assert all(e is not None and e < 1e-10 for e in d["verify_rel_err"]), d["verify_rel_err"]
    assert "2" in d["preflight"]["ms_by_schedule"]


def test_bench_falls_back_when_a_rank_goes_missing():
    """a rank skips a collective of the schedule-2 preflight: the watchdog's time-out fires on its peers, every rank rebuilds
    its communicator, schedule 1 is preflighted and measured, and the line says so"""
    d, err = _run_bench(["--local-ranks", "4"], {"FRIEDRICH_AMD_TEST_COMM_HANG": "2,2,8"})
    assert d["schedule_used"] == 1
    assert d["fallback_reason"] and "schedule 2" in d["fallback_reason"]
    assert "falling back" in err
    assert sum(d["per_rank"]["comm_timeouts"]) >= 1
    assert all(e is not None and e < 1e-10 for e in d["verify_rel_err"]), d["verify_rel_err"]
    assert d["value"] > 0


def test_bench_runs_as_replicas_when_no_schedule_works():
    """every schedule loses a rank (three injections cannot be expressed with one knob: the hang is placed in the first
    collective any sharded factorisation issues, the status agreement, which all schedules share) -> replicas, still one
    valid line"""
    d, err = _run_bench(["--local-ranks", "3", "--dist-schedule", "0"], {"FRIEDRICH_AMD_TEST_COMM_HANG": "0,1,1"})
    assert d["schedule_used"] == -1 and "schedule 0" in d["fallback_reason"]
    assert "replicated" in d["config"]["parallelism"]
    assert all(e is not None and e < 1e-10 for e in d["verify_rel_err"]), d["verify_rel_err"]
