"""N > 1 path on CPU: a 2-rank gloo group replays friedrich_amd.sharding.panel_schedule -- the same ownership map
and panel-broadcast order as chol.hip -- with numpy tiles (scipy for the tile factorisation) and checks that every
rank ends with the oracle's factor, and that bench.py's query split covers every row exactly once."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, rand_inputs, rel_err


def _worker_split(rank, world, port, n, nb, out_dir):
    """the split variant (option dist_schedule = 1): diagonal block by the owner + broadcast, scatter of the rows below, every
    rank solves its slice, all-gather"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scipy.linalg as sl
    import torch
    import torch.distributed as dist

    from friedrich_amd import sharding
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k = ("matern2", 0.7, 1.2)
    X = rand_inputs(n, 3, 42)
    A = np.full((n, n), np.nan)
    for j in range(0, n, nb):
        if sharding.owner_of(j, nb, world) == rank:
            w = min(nb, n - j)
            A[j:, j:j + w] = O.make_covariance_matrix(k, X[j:], X[j:j + w])
            A[j:j + w, j:j + w] += 0.01 * np.eye(w)
    for step in sharding.panel_schedule(n, nb, world):
        kk, kb, owner = step["k"], step["kb"], step["owner"]
        head = np.zeros((kb, kb))
        if rank == owner:
            head = sl.cholesky(A[kk:kk + kb, kk:kk + kb], lower=True)
        t = torch.from_numpy(np.ascontiguousarray(head))
        dist.broadcast(t, src=owner)
        L11 = t.numpy()
        A[kk:kk + kb, kk:kk + kb] = L11
        slice_rows, slices = sharding.split_slices(n, kk, kb, world)
        # scatter: the owner holds the updated, unsolved rows below
        bufs = []
        for r, (lo, rows) in enumerate(slices):
            b = np.zeros((slice_rows, kb))
            if rank == owner and rows > 0:
                b[:rows] = A[lo:lo + rows, kk:kk + kb]
            bufs.append(torch.from_numpy(b))
        mine = torch.zeros((slice_rows, kb), dtype=torch.float64)
        dist.scatter(mine, bufs if rank == owner else None, src=owner)
        lo, rows = slices[rank]
        solved = np.zeros((slice_rows, kb))
        if rows > 0:
            solved[:rows] = sl.solve_triangular(L11, mine.numpy()[:rows].T, lower=True).T
        gathered = [torch.zeros((slice_rows, kb), dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(solved))
        for r, (lo, rows) in enumerate(slices):
            if rows > 0:
                A[lo:lo + rows, kk:kk + kb] = gathered[r].numpy()[:rows]
        P = A[kk + kb:, kk:kk + kb]
        for j in step["updates"][rank]:
            w = min(nb, n - j)
            A[j:, j:j + w] -= P[j - kk - kb:] @ P[j - kk - kb:j - kk - kb + w].T
    np.save(os.path.join(out_dir, f"L{rank}.npy"), np.tril(A))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, n, nb, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scipy.linalg as sl
    import torch
    import torch.distributed as dist

    from friedrich_amd import sharding
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k = ("matern2", 0.7, 1.2)
    X = rand_inputs(n, 3, 42)
    A = np.full((n, n), np.nan)
    # Gram: owned block columns only
    for j in range(0, n, nb):
        if sharding.owner_of(j, nb, world) == rank:
            w = min(nb, n - j)
            A[j:, j:j + w] = O.make_covariance_matrix(k, X[j:], X[j:j + w])
            A[j:j + w, j:j + w] += 0.01 * np.eye(w)
    moved = 0
    for step in sharding.panel_schedule(n, nb, world):
        kk, kb, owner = step["k"], step["kb"], step["owner"]
        panel = np.zeros((n - kk, kb))
        if rank == owner:
            L11 = sl.cholesky(A[kk:kk + kb, kk:kk + kb], lower=True)
            A[kk:kk + kb, kk:kk + kb] = L11
            if kk + kb < n:
                A[kk + kb:, kk:kk + kb] = sl.solve_triangular(L11, A[kk + kb:, kk:kk + kb].T, lower=True).T
            panel = np.tril(A[kk:, kk:kk + kb], 0) if False else A[kk:, kk:kk + kb].copy()
        t = torch.from_numpy(np.ascontiguousarray(panel))
        dist.broadcast(t, src=owner)
        moved += sharding.panel_bytes(n, kk, kb)
        A[kk:, kk:kk + kb] = t.numpy()
        P = A[kk + kb:, kk:kk + kb]
        for j in step["updates"][rank]:
            w = min(nb, n - j)
            A[j:, j:j + w] -= P[j - kk - kb:] @ P[j - kk - kb:j - kk - kb + w].T
    np.save(os.path.join(out_dir, f"L{rank}.npy"), np.tril(A))
    lo, hi = sharding.query_slice(37, rank, world)
    np.save(os.path.join(out_dir, f"q{rank}.npy"), np.arange(lo, hi))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,nb", [(300, 64), (257, 128)])
def test_two_rank_panel_schedule(tmp_path, n, nb):
    import torch.multiprocessing as tmp_mp

    from oracle import oracle as O

    world = 2
    port = 29500 + (os.getpid() % 2000) + n % 7
    tmp_mp.spawn(_worker, args=(world, port, n, nb, str(tmp_path)), nprocs=world, join=True)
    X = rand_inputs(n, 3, 42)
    st, L_o, _ = O.make_cholesky_cov_matrix(("matern2", 0.7, 1.2), X, 0.1)
    for r in range(world):
        L = np.load(tmp_path / f"L{r}.npy")
        assert rel_err(L, np.tril(L_o)) < 1e-11
    q = np.concatenate([np.load(tmp_path / f"q{r}.npy") for r in range(world)])
    assert q.tolist() == list(range(37))


def test_schedule_covers_every_block_once():
    from friedrich_amd import sharding

    for n, nb, world in [(1000, 128, 3), (4096, 512, 8), (100, 256, 4)]:
        steps = list(sharding.panel_schedule(n, nb, world))
        assert [s["k"] for s in steps] == list(range(0, n, nb))
        for s in steps:
            assert s["owner"] == (s["k"] // nb) % world
            cols = sorted(c for r in s["updates"].values() for c in r)
            assert cols == list(range(s["k"] + s["kb"], n, nb))
            for r, cs in s["updates"].items():
                assert all(sharding.owner_of(c, nb, world) == r for c in cs)


@pytest.mark.parametrize("n,nb", [(300, 64), (257, 128)])
def test_two_rank_split_panel_schedule(tmp_path, n, nb):
    """the split variant of the panel step (scatter + per-rank solves + all-gather) over a 2-rank gloo group"""
    import torch.multiprocessing as tmp_mp

    from oracle import oracle as O

    world = 2
    port = 31500 + (os.getpid() % 2000) + n % 7
    tmp_mp.spawn(_worker_split, args=(world, port, n, nb, str(tmp_path)), nprocs=world, join=True)
    X = rand_inputs(n, 3, 42)
    st, L_o, _ = O.make_cholesky_cov_matrix(("matern2", 0.7, 1.2), X, 0.1)
    for r in range(world):
        assert rel_err(np.load(tmp_path / f"L{r}.npy"), np.tril(L_o)) < 1e-11


def test_split_slices_cover_the_rows_below_once():
    from friedrich_amd import sharding

    for n, k, kb, world in [(1000, 0, 128, 3), (4096, 512, 512, 8), (700, 640, 60, 4), (32768, 1024, 512, 8)]:
        slice_rows, slices = sharding.split_slices(n, k, kb, world)
        assert slice_rows % 128 == 0 and slice_rows * world >= n
        rows = [r for lo, cnt in slices for r in range(lo, lo + cnt)]
        assert rows == list(range(k + kb, n))


def _worker_chain(rank, world, port, n, nb, out_dir):
    """schedule 2 (dist_schedule = 2): diagonal chain first -- H_p and R1_p fanned out, u1 on the next owner, the bulk rows by
    scatter / per-rank solves / all-gather, trailing updates with the nearest owned column first; in the host issue order
    of potrf_dist_chain (friedrich_amd.sharding.chain_rounds)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scipy.linalg as sl
    import torch
    import torch.distributed as dist

    from friedrich_amd import sharding
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k = ("matern2", 0.7, 1.2)
    X = rand_inputs(n, 3, 42)
    P = -(-n // nb)
    kof = lambda q: min(q * nb, n)
    A = np.full((n, n), np.nan)
    for q in range(P):  # Gram: owned block columns only
        if q % world == rank:
            j, w = kof(q), kof(q + 1) - kof(q)
            A[j:, j:j + w] = O.make_covariance_matrix(k, X[j:], X[j:j + w])
            A[j:j + w, j:j + w] += 0.01 * np.eye(w)

    def bcast(block, src):
        t = torch.from_numpy(np.ascontiguousarray(block))
        dist.broadcast(t, src=src)
        return t.numpy()

    for r in sharding.chain_rounds(n, nb, world):
        kk, k1, k2, k3, owner = r["k"], r["k1"], r["k2"], r["k3"], r["owner"]
        kb = k1 - kk
        # 1: D_p on its owner
        D = np.zeros((kb, kb))
        if rank == owner:
            D = sl.cholesky(A[kk:k1, kk:k1], lower=True)
            A[kk:k1, kk:k1] = D
        # 2: R1_p solved by the owner, fanned out (the one message of the chain); u1 on the next owner
        if k2 > k1:
            R1 = np.zeros((k2 - k1, kb))
            if rank == owner:
                R1 = sl.solve_triangular(D, A[k1:k2, kk:k1].T, lower=True).T
            R1 = bcast(R1, owner)
            A[k1:k2, kk:k1] = R1
            if rank == r["next"]:
                A[k1:k2, k1:k2] -= R1 @ R1.T
        # 3: bulk stream: D_p to every rank, then the rows below R1_p
        D = bcast(D, owner)
        A[kk:k1, kk:k1] = D
        sr, slices = r["bulk"]
        if sr > 0:
            bufs = []
            for q, (lo, rows) in enumerate(slices):
                b = np.zeros((sr, kb))
                if rank == owner and rows > 0:
                    b[:rows] = A[lo:lo + rows, kk:k1]
                bufs.append(torch.from_numpy(b))
            mine = torch.zeros((sr, kb), dtype=torch.float64)
            dist.scatter(mine, bufs if rank == owner else None, src=owner)
            lo, rows = slices[rank]
            solved = np.zeros((sr, kb))
            if rows > 0:
                solved[:rows] = sl.solve_triangular(D, mine.numpy()[:rows].T, lower=True).T
            gathered = [torch.zeros((sr, kb), dtype=torch.float64) for _ in range(world)]
            dist.all_gather(gathered, torch.from_numpy(solved))
            for q, (lo, rows) in enumerate(slices):
                if rows > 0:
                    A[lo:lo + rows, kk:k1] = gathered[q].numpy()[:rows]
        # 5: trailing updates, nearest owned column first
        if k2 > k1:
            Lp = A[:, kk:k1]
            q = r["near"][rank]
            if q == r["p"] + 1:
                if k3 > k2:
                    A[k2:k3, k1:k2] -= Lp[k2:k3] @ Lp[k1:k2].T
                if n > k3:
                    A[k3:, k1:k2] -= Lp[k3:] @ Lp[k1:k2].T
            elif q < P:
                kq, wq = kof(q), kof(q + 1) - kof(q)
                A[kq:, kq:kq + wq] -= Lp[kq:] @ Lp[kq:kq + wq].T
            for j in range(q + 1, P):
                if j % world == rank:
                    kj, wj = kof(j), kof(j + 1) - kof(j)
                    A[kj:, kj:kj + wj] -= Lp[kj:] @ Lp[kj:kj + wj].T
    np.save(os.path.join(out_dir, f"L{rank}.npy"), np.tril(A))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,nb", [(300, 64), (257, 128), (100, 128), (200, 128)])
def test_two_rank_chain_schedule(tmp_path, n, nb):
    """the chain-first schedule (the default of a sharded factorisation) over a 2-rank gloo group"""
    import torch.multiprocessing as tmp_mp

    from oracle import oracle as O

    world = 2
    port = 32500 + (os.getpid() % 2000) + n % 7
    tmp_mp.spawn(_worker_chain, args=(world, port, n, nb, str(tmp_path)), nprocs=world, join=True)
    X = rand_inputs(n, 3, 42)
    st, L_o, _ = O.make_cholesky_cov_matrix(("matern2", 0.7, 1.2), X, 0.1)
    for r in range(world):
        assert rel_err(np.load(tmp_path / f"L{r}.npy"), np.tril(L_o)) < 1e-11


def test_chain_rounds_cover_every_tile_once():
    """every block column p is split into head / R1 / bulk slices that cover its rows exactly once, and every (panel, later
    block column) pair is updated by exactly one rank: the owner of the column, either as its nearest column or in the rest"""
    from friedrich_amd import sharding

    for n, nb, world in [(1000, 128, 3), (4096, 512, 8), (100, 256, 4), (32768, 512, 8), (700, 128, 2)]:
        P = -(-n // nb)
        rounds = list(sharding.chain_rounds(n, nb, world))
        assert [r["k"] for r in rounds] == list(range(0, n, nb))
        for r in rounds:
            rows = list(range(*r["head"])) + list(range(*r["r1"])) + [x for lo, cnt in r["bulk"][1] for x in range(lo, lo + cnt)]
            assert rows == list(range(r["k"], n))
            assert r["owner"] == r["p"] % world and r["next"] == (r["p"] + 1) % world
            for rank, q in r["near"].items():
                assert q > r["p"] and q % world == rank and all(j % world != rank for j in range(r["p"] + 1, q))


# ---- the guarded start of a sharded run over a REAL torch.distributed group (gloo, 2 processes) ----------------------------
# friedrich_amd.sharding.TorchLink / guarded_schedule / reattach are what bench.py runs between the ranks of an N > 1 launch;
# the GPU tests drive the same functions with thread-ranks (ThreadLink).  Here the link is the one bench.py uses -- object
# all-gather and broadcast over a gloo group -- and the library context is a stand-in that records what the policy asks of it.
class _FakeCtx:
    def __init__(self, rank, fail_attach_at=None):
        self.rank, self.log, self.attached, self.opts = rank, [], 0, {}
        self.fail_attach_at = fail_attach_at

    def set_option(self, name, value):
        self.opts[name] = value

    def comm_unique_id(self):
        return bytes([self.attached % 251]) * 128

    def comm_init(self, rank, world, uid):
        from friedrich_amd.device import FriedrichError

        assert len(uid) == 128 and rank == self.rank
        self.attached += 1
        self.log.append(("init", uid[0]))
        if self.fail_attach_at is not None and self.attached == self.fail_attach_at and rank == 1:
            raise FriedrichError(8, "injected: communicator init failed on rank 1")

    def comm_selftest(self):
        self.log.append(("selftest",))

    def comm_finalize(self, abort=False):
        self.log.append(("finalize", bool(abort)))


def _worker_guard(rank, world, port, out_dir, scenario):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import datetime
    import json

    import torch.distributed as dist

    from friedrich_amd import sharding
    from friedrich_amd.device import FriedrichError

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=120))
    link = sharding.TorchLink(dist, rank, world, ctl)
    ctx = _FakeCtx(rank, fail_attach_at=2 if scenario == "attach_fails" else None)
    link.attach(ctx)

    def preflight(s):
        if scenario == "all_pass":
            return None
        if s == 2:  # wrong factor on rank 1 only: rank 0 must learn it
            return "factor deviates" if rank == 1 else None
        if s == 1 and scenario != "attach_fails":  # a time-out inside the library on rank 0 only
            if rank == 0:
                raise FriedrichError(8, "sharded factorisation timed out")
            return None
        return None

    schedule, reasons, took = sharding.guarded_schedule(ctx, link, preflight, timeout_ms=1234)
    assert link.gather(rank) == list(range(world))
    link.barrier()
    with open(os.path.join(out_dir, f"g{rank}.json"), "w") as f:
        json.dump({"schedule": schedule, "reasons": reasons, "took": sorted(took), "log": ctx.log, "opts": ctx.opts}, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["all_pass", "two_fall_backs", "attach_fails"])
def test_guarded_schedule_over_gloo(tmp_path, scenario):
    import json

    import torch.multiprocessing as tmp_mp

    world = 2
    port = 33100 + (os.getpid() % 2000) + len(scenario)
    tmp_mp.spawn(_worker_guard, args=(world, port, str(tmp_path), scenario), nprocs=world, join=True)
    res = [json.load(open(tmp_path / f"g{r}.json")) for r in range(world)]
    # both ranks reach the same decision with the same reasons, whoever saw the failure
    assert res[0]["schedule"] == res[1]["schedule"] and res[0]["reasons"] == res[1]["reasons"]
    for r in res:
        assert r["opts"]["comm_timeout_ms"] == 1234
    if scenario == "all_pass":
        assert res[0]["schedule"] == 2 and res[0]["reasons"] == [] and res[0]["took"] == [2]
        assert [e[0] for e in res[0]["log"]] == ["init", "selftest"]
    elif scenario == "two_fall_backs":
        assert res[0]["schedule"] == 0 and len(res[0]["reasons"]) == 2
        assert "schedule 2: rank 1: factor deviates" in res[0]["reasons"][0] and "schedule 1: rank 0" in res[0]["reasons"][1]
        for r in res:
            # after every failure: communicator dropped WITHOUT waiting for the peer, a fresh one attached and self-tested
            assert [e[0] for e in r["log"]] == ["init", "selftest", "finalize", "init", "selftest", "finalize", "init", "selftest"]
            assert all(e[1] for e in r["log"] if e[0] == "finalize")
            assert r["opts"]["dist_schedule"] == 0
        # the fresh ids came from rank 0 and reached rank 1
        assert [e[1] for e in res[1]["log"] if e[0] == "init"][1:] == [e[1] for e in res[0]["log"] if e[0] == "init"][1:]
    else:
        # schedule 2 fails its preflight; re-attaching fails on rank 1: both ranks end without communicator (replicas)
        assert res[0]["schedule"] == -1 and any("attaching a fresh communicator failed" in x for x in res[0]["reasons"])
        for r in res:
            assert r["log"][-1] == ["finalize", True]


# ---- add_samples and the gradient terms as they are dealt to the ranks ---------------------------------
def _worker_8e(rank, world, port, n0, nb_new, out_dir):
    """numpy replay of the two deals over a gloo group: a rank solves ITS slice of L21^T and one all-gather returns the whole;
    a rank forms the rows of L^-1 of ITS chunks (backward solve on the leading block), accumulates its partial K^-1 and reduces
    it -- tr(K^-1 G), tr(K^-1) are linear in K^-1, so the partial scalars add up"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scipy.linalg as sl
    import torch
    import torch.distributed as dist

    from friedrich_amd import sharding
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k = ("squared_exp", 0.8, 1.3)
    n = n0 + nb_new
    X = rand_inputs(n, 3, 42)
    K = O.make_covariance_matrix(k, X, X) + 0.04 * np.eye(n)
    L11 = sl.cholesky(K[:n0, :n0], lower=True)
    # (b) add_rows: slices of the right-hand sides
    width, slices = sharding.add_rows_slices(nb_new, world)
    lo, rows = slices[rank]
    mine = np.zeros((width, n0))
    if rows > 0:
        mine[:rows] = sl.solve_triangular(L11, K[:n0, n0 + lo:n0 + lo + rows], lower=True).T
    gathered = [torch.zeros((width, n0), dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(mine))
    L21 = np.concatenate([g.numpy() for g in gathered])[:nb_new]
    S = K[n0:, n0:] - L21 @ L21.T  # replicated on every rank
    L = np.zeros((n, n))
    L[:n0, :n0], L[n0:, :n0], L[n0:, n0:] = L11, L21, sl.cholesky(S, lower=True)
    np.save(os.path.join(out_dir, f"L{rank}.npy"), L)
    # (c) gradient terms: rows of W = L^-1 in chunks
    rng = np.random.default_rng(5)
    G = rng.standard_normal((n, n))
    G = G + G.T
    Kinv_part = np.zeros((n, n))
    for k0, k1, owner in sharding.grad_chunks(n, world):
        if owner != rank:
            continue
        E = np.zeros((k1, k1 - k0))
        E[k0:, :] = np.eye(k1 - k0)
        Xc = sl.solve_triangular(L[:k1, :k1], E, lower=True, trans="T")  # = W[k0:k1, :k1]^T
        Kinv_part[:k1, :k1] += Xc @ Xc.T
    part = torch.tensor([float(np.sum(Kinv_part * G)), float(np.trace(Kinv_part))], dtype=torch.float64)
    allp = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(allp, part)
    tot = sum(p.numpy() for p in allp)  # (rank order)
    np.save(os.path.join(out_dir, f"g{rank}.npy"), np.array([tot[0], tot[1], float(np.sum(np.linalg.inv(K) * G)), float(np.trace(np.linalg.inv(K)))]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n0,nb_new", [(700, 300), (1100, 1)])
def test_two_rank_add_rows_and_grad_chunks(tmp_path, n0, nb_new):
    import torch.multiprocessing as tmp_mp

    from oracle import oracle as O

    world = 2
    port = 35100 + (os.getpid() % 2000) + nb_new % 7
    tmp_mp.spawn(_worker_8e, args=(world, port, n0, nb_new, str(tmp_path)), nprocs=world, join=True)
    n = n0 + nb_new
    X = rand_inputs(n, 3, 42)
    st, L_o, _ = O.make_cholesky_cov_matrix(("squared_exp", 0.8, 1.3), X, 0.2)
    for r in range(world):
        assert rel_err(np.load(tmp_path / f"L{r}.npy"), np.tril(L_o)) < 1e-10
        g = np.load(tmp_path / f"g{r}.npy")
        assert abs(g[0] - g[2]) < 1e-9 * (abs(g[2]) + 1.0) and abs(g[1] / g[3] - 1.0) < 1e-10


def test_grad_chunks_and_add_rows_slices_cover_everything_once():
    from friedrich_amd import sharding

    for n in (1024, 1300, 8192, 8192 + 640, 32768):
        for world in (2, 3, 4, 8):
            ch = sharding.grad_chunks(n, world)
            rows = sorted((k0, k1) for k0, k1, _ in ch)
            assert rows[0][0] == 0 and rows[-1][1] == n and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            assert all(k0 % sharding.grad_chunk_rows(n) == 0 for k0, _, _ in ch)
            # the snake keeps the ranks' work level: cost of a chunk ~ (where it ends)^2 x its rows
            cost = [0.0] * world
            for k0, k1, o in ch:
                cost[o] += float(k1) ** 2 * (k1 - k0)
            if len(ch) >= 2 * world:
                assert max(cost) < 1.6 * (sum(cost) / world), (n, world, cost)
    for nb_new in (1, 2, 7, 512, 515):
        for world in (2, 3, 8):
            width, sl_ = sharding.add_rows_slices(nb_new, world)
            got = [r for lo, rows in sl_ for r in range(lo, lo + rows)]
            assert got == list(range(nb_new)) and all(rows <= width for _, rows in sl_)


# ---- one rendezvous primitive on the control plane (round-4 advisor finding: barrier against all_gather_object) ----------------
def _worker_rendezvous(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import datetime
    import json

    import torch.distributed as dist

    from friedrich_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=60))
    link = sharding.TorchLink(dist, rank, world, ctl)
    log = []
    # round 1: rank 0's step fails before the rendezvous of the timed region; rank 1 is already waiting in it
    if rank == 0:
        log.append(list(sharding.agree(link, False, "step failed")))
    else:
        try:
            sharding.sync_point(link)
            log.append("passed")
        except sharding.PeerFailure as e:
            log.append(["peer", str(e)])
    # round 2: both healthy -- a plain barrier on one side meets a sync_point on the other (same primitive underneath)
    if rank == 0:
        link.barrier()
        log.append("barrier")
    else:
        sharding.sync_point(link)
        log.append("sync")
    # round 3: both report, one of them a failure
    log.append(list(sharding.agree(link, rank == 0, None if rank == 0 else "late failure")))
    with open(os.path.join(out_dir, f"r{rank}.json"), "w") as f:
        json.dump(log, f)
    dist.barrier()
    dist.destroy_process_group()


def test_control_plane_rendezvous_is_one_primitive(tmp_path):
    import json

    import torch.multiprocessing as tmp_mp

    world = 2
    port = 36100 + (os.getpid() % 2000)
    tmp_mp.spawn(_worker_rendezvous, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (json.load(open(tmp_path / f"r{r}.json")) for r in range(world))
    assert r0[0] == [False, "rank 0: step failed"] and r1[0] == ["peer", "rank 0: step failed"]
    assert r0[1] == "barrier" and r1[1] == "sync"
    assert r0[2] == [False, "rank 1: late failure"] and r1[2] == [False, "rank 1: late failure"]
