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
