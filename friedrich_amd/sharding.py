"""Host-side description of how the hot path is sharded across the GPUs of one node (mirrors chol.hip).

  * Gram + Cholesky: block column b (width nb) of the n x n matrix belongs to rank b % world (`owner_of`).  The owner
    assembles, updates and factors it; after factoring, the panel (rows k..n of the block column plus its 128 x 128
    inverse blocks) is broadcast to every rank, so all ranks end with the complete factor.
  * predict family: the m query rows are split into contiguous slices (`query_slice`), no communication.

`panel_schedule` enumerates, per outer step, who factors / broadcasts and which block columns each rank updates;
tests/test_sharding_gloo.py replays it with numpy tiles over a 2-rank gloo group and bench.py uses `query_slice`.
"""


def owner_of(col, nb, world):
    return (col // nb) % world


def query_slice(m, rank, world):
    return (m * rank) // world, (m * (rank + 1)) // world


def panel_bytes(n, k, kb, inv_block=128):
    """bytes one panel broadcast moves: (n-k) x kb factor rows + the inverse blocks of the kb columns"""
    return 8 * ((n - k) * kb + ((kb + inv_block - 1) // inv_block) * inv_block * inv_block)


def panel_schedule(n, nb, world):
    """yield dicts {k, kb, owner, updates: {rank: [block column starts it updates with this panel]}}"""
    k = 0
    while k < n:
        kb = min(nb, n - k)
        updates = {r: [] for r in range(world)}
        j = k + kb
        while j < n:
            updates[owner_of(j, nb, world)].append(j)
            j += nb
        yield {"k": k, "kb": kb, "owner": owner_of(k, nb, world), "updates": updates}
        k += kb


def split_slices(n, k, kb, world, block=128):
    """option dist_schedule = 1: the rows below the diagonal block of the panel at k, cut into `world` slices of whole 128-row
    blocks -> (slice_rows, [(first row, rows) per rank])  (mirrors split_panel in chol.hip)"""
    slice_rows = -(-(-(-n // world)) // block) * block
    below0 = k + kb
    out = []
    for r in range(world):
        lo = below0 + r * slice_rows
        out.append((lo, max(0, min(slice_rows, n - lo))))
    return slice_rows, out


# ---- schedule 2 (option dist_schedule = 2, the default): the chain of diagonal blocks first (potrf_dist_chain in chol.hip) ----
def nearest_owned(p, rank, world):
    """index of the first panel after p that `rank` owns (may lie beyond the last panel)"""
    return p + 1 + ((rank - (p + 1)) % world)


def chain_rounds(n, nb, world):
    """One dict per panel p, in the host issue order of potrf_dist_chain:
         k, k1, k2, k3   first column of panels p, p + 1, p + 2, p + 3 (clamped to n)
         owner, next     ranks owning panels p and p + 1
         r1              (k1, k2): rows of R1_p = L[panel p + 1's rows, p], solved by the owner, fanned out  [chain stream, comm 0];
                         the next owner applies them to its diagonal block at once (u1) -- the chain's only message
         head            (k, k1): the diagonal block D_p, fanned out with its inverse blocks           [bulk stream, comm 1]
         bulk            slice_rows and [(first row, rows)] per rank for the rows from k2 on: scatter, per-rank solves,
                         all-gather                                                                    [bulk stream, comm 1]
         near[rank]      the panel whose block column `rank` updates first with panel p (its nearest owned one)"""
    P = -(-n // nb)
    kof = lambda q: min(q * nb, n)
    for p in range(P):
        k, k1, k2, k3 = kof(p), kof(p + 1), kof(p + 2), kof(p + 3)
        below = n - k2
        sr = -(-(-(-below // world)) // 128) * 128 if below > 0 else 0
        yield {
            "p": p, "k": k, "k1": k1, "k2": k2, "k3": k3, "owner": p % world, "next": (p + 1) % world,
            "head": (k, k1), "r1": (k1, k2),
            "bulk": (sr, [(k2 + r * sr, max(0, min(sr, below - r * sr))) for r in range(world)]),
            "near": {r: nearest_owned(p, r, world) for r in range(world)},
        }


# ---- add_samples and the gradient terms (fr_chol_add_rows in chol.hip, grad_terms_impl in grad.hip) -------
def add_rows_slices(nb_new, world):
    """fr_chol_add_rows: the nb_new right-hand sides of  L21^T = L11^-1 K12  in `world` equal slices
    -> (slice width, [(first new row, rows) per rank]); one all-gather of slice x n_old doubles per rank returns L21^T"""
    width = -(-nb_new // world)
    out = []
    for r in range(world):
        lo = min(nb_new, r * width)
        out.append((lo, min(nb_new, lo + width) - lo))
    return width, out


def grad_chunk_rows(n):
    """rows per chunk of the sharded gradient terms"""
    return 2048 if n >= 8192 else 512


def grad_chunks(n, world):
    """fr_grad_terms: the rows of W = L^-1 in chunks, dealt in a snake over DESCENDING row ranges (the cost of a chunk -- a backward
    solve on the leading block that ends with it and that block's share of W^T W -- grows with the square of where it ends)
    -> [(k0, k1, owner)], largest first.  A rank accumulates  sum over its chunks of  W[k0:k1, :k1]^T W[k0:k1, :k1]  into the
    leading k1 x k1 corner of its PARTIAL K^-1; the reductions are linear in K^-1, so the ranks' p + 2 scalars just add up."""
    cs = grad_chunk_rows(n)
    nc = -(-n // cs)
    out = []
    for t in range(nc):
        owner = (world - 1 - t % world) if (t // world) & 1 else t % world
        j = nc - 1 - t
        out.append((j * cs, min(n, (j + 1) * cs), owner))
    return out


# ---- guarded start of a sharded run: preflight, watchdog, schedule fall-back (bench.py, tests/test_gpu_dist_guard.py) ---------
# The library bounds every wait for a collective (option "comm_timeout_ms": comm.hip) and reports FR_RCCL_ERROR instead of
# hanging; what to do then is the host's decision.  These helpers are that decision for bench.py and the tests: try the
# schedules from the most aggressive to the most conservative, each behind a small sharded fit that is compared with a
# single-rank fit of the same rows; after a failure EVERY rank drops its communicator (aborted where the wait ran out),
# attaches a fresh one and tries the next schedule; when none works the ranks continue as independent replicas (world 1).
# Agreement between the ranks travels OUT OF BAND -- a `link` object (gloo group of torch.distributed in bench.py, a
# threading.Barrier for thread-ranks) -- never over the communicator under test.
import itertools
import threading
import time

SCHEDULE_NAMES = {
    2: "diagonal chain first: block to the next owner, then scatter / all-gather of the rows below",
    1: "diagonal block broadcast + scatter / all-gather of the rows below",
    0: "one broadcast per panel",
    -1: "replicated: every rank factors the whole matrix (no working sharded schedule), queries sharded",
}

_thread_group_ids = itertools.count(500000)


class ThreadShared:
    """state shared by the thread-ranks of one process (in-process transport of the library)"""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.base = next(_thread_group_ids) * 64


class ThreadLink:
    """out-of-band link of one thread-rank: rendezvous through a threading.Barrier"""

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world
        self.generation = 0

    def gather(self, obj):
        self.shared.slots[self.rank] = obj
        self.shared.barrier.wait(timeout=900)
        out = list(self.shared.slots)
        self.shared.barrier.wait(timeout=900)
        return out

    def barrier(self):
        self.gather(None)  # (every rendezvous of the control plane is the SAME primitive: see agree)

    def abort(self):
        """this rank is about to die of an exception: peers waiting in a rendezvous fail at once instead of timing out"""
        self.shared.barrier.abort()

    def attach(self, ctx):
        """collective: a fresh communicator on every rank's context"""
        self.generation += 1
        ctx.comm_init_local(self.shared.base + self.generation, self.rank, self.world)
        ctx.comm_selftest()


class TorchLink:
    """out-of-band link of one process-rank: a gloo group of torch.distributed (CPU side: independent of RCCL and of the GPU)"""

    def __init__(self, dist, rank, world, control_group):
        self.dist, self.rank, self.world, self.ctl = dist, rank, world, control_group

    def gather(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj, group=self.ctl)
        return out

    def barrier(self):
        self.gather(None)  # (every rendezvous of the control plane is the SAME primitive: see agree)

    def abort(self):
        pass  # (a process-rank that dies takes the job down: torch.distributed.run ends its peers)

    def attach(self, ctx):
        ids = [ctx.comm_unique_id() if self.rank == 0 else None]
        self.dist.broadcast_object_list(ids, src=0, group=self.ctl)
        ctx.comm_init(self.rank, self.world, ids[0])
        ctx.comm_selftest()


class SoloLink:
    """a single rank without any peer"""
    rank, world = 0, 1

    def gather(self, obj):
        return [obj]

    def barrier(self):
        pass

    def abort(self):
        pass

    def attach(self, ctx):
        pass


def preflight_fit(ctx, ref_ctx, n=4096, d=8, tol=1e-11):
    """collective: a sharded fit of n synthetic rows on `ctx` (its communicator, its dist_schedule) compared with the
    single-rank fit of the same rows on `ref_ctx` (a context without communicator on the same GPU).
    -> None, or a string saying what is wrong"""
    import numpy as np

    from . import synth

    X, _, _ = synth.make_problem(n, d, cfg=1, m=1)
    k = ("squared_exp", 1.1, 1.0)
    ref = ref_ctx.cholesky_from_inputs(k, X, 0.1)
    Lref = ref.l()
    ref.free()
    chol = ctx.cholesky_from_inputs(k, X, 0.1)
    L = chol.l()
    info = chol.info()
    chol.free()
    err = float(np.max(np.abs(L - Lref)) / np.max(np.abs(Lref)))
    if not np.isfinite(err) or err > tol:
        return f"preflight factor (N={n}) deviates from the single-rank factor by {err:.2e} (tolerance {tol:.0e})"
    if info["n_subst"] != 0 or info["fail_col"] != -1:
        return f"preflight factor (N={n}) reports substitutions / a failing column: {info}"
    return None


class PeerFailure(RuntimeError):
    """raised by sync_point on a healthy rank: a peer reported a failure in the same rendezvous"""


def agree(link, ok, why=None):
    """-> (every rank ok?, the first failing rank's reason).  EVERY rendezvous of the control plane -- this one, sync_point,
    link.barrier -- is one link.gather, so a rank that fails early and reports it here meets its peers wherever they are (in a
    sync_point of the measurement, in their own agree behind it): the ranks can never sit in two different collectives of
    the out-of-band link (round-4 advisor finding: barrier against all_gather_object).  A plain barrier contributes None = ok."""
    flags = link.gather((bool(ok), why))
    bad = [(r, f[1]) for r, f in enumerate(flags) if f is not None and not f[0]]
    if not bad:
        return True, None
    return False, f"rank {bad[0][0]}: {bad[0][1]}"


def sync_point(link):
    """rendezvous inside a guarded measurement: returns when every rank is here, raises PeerFailure when one of them reported
    a failure instead (it is in `agree` after catching its error) -- the caller then goes to the fall-back WITHOUT another
    agree: this rendezvous was it"""
    ok, why = agree(link, True)
    if not ok:
        raise PeerFailure(why)


def reattach(ctx, link):
    """every rank drops its communicator -- without waiting for peers: it may be the one that timed out -- and attaches a
    fresh one.  -> None or the reason it failed (then the ranks are left without communicator: replicas)"""
    from .device import FriedrichError

    ctx.comm_finalize(abort=True)
    try:
        link.attach(ctx)
        ok, why = True, None
    except FriedrichError as e:
        ok, why = False, f"attaching a fresh communicator failed: {e}"
    ok, why = agree(link, ok, why)
    if not ok:
        ctx.comm_finalize(abort=True)
    return why


def guarded_schedule(ctx, link, preflight, schedules=(2, 1, 0), timeout_ms=20000, log=None):
    """Pick the first schedule of `schedules` whose preflight passes on EVERY rank.  preflight(schedule) is collective and
    returns None / a reason, or raises FriedrichError (a time-out inside the library).  The context must have a communicator
    attached; on return it has a working one and the option dist_schedule set -- or none at all (-1: replicas).
    -> (schedule, [reasons of the schedules that failed], {schedule: preflight milliseconds})"""
    from .device import FriedrichError

    reasons, took = [], {}
    ctx.set_option("comm_timeout_ms", timeout_ms)
    for i, s in enumerate(schedules):
        ctx.set_option("dist_schedule", s)
        t0 = time.perf_counter()
        try:
            why = preflight(s)
            ok = why is None
        except FriedrichError as e:
            ok, why = False, str(e)
        except BaseException:
            link.abort()  # (not a library error: this rank is going down; do not leave the peers in a 10-minute wait)
            raise
        took[s] = 1e3 * (time.perf_counter() - t0)
        ok, why = agree(link, ok, why)
        if ok:
            return s, reasons, took
        reasons.append(f"schedule {s}: {why}")
        if log:
            log(f"sharded schedule {s} failed its preflight ({why}); falling back")
        if i + 1 < len(schedules):
            why = reattach(ctx, link)
            if why is not None:
                reasons.append(why)
                return -1, reasons, took
    ctx.comm_finalize(abort=True)
    return -1, reasons, took
