"""The sharded factorisation over REAL RCCL with one process per GPU.  Skipped unless at least two GPUs are visible: the
builder's boxes have one (the thread-rank transport of test_gpu_dist.py and the gloo replay cover the schedule there), so
this is the test a multi-GPU node runs first."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, rand_inputs, rel_err

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, n, nb, split, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    from friedrich_amd.device import Context

    torch.cuda.set_device(rank)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", rank))
    ctx = Context(rank)
    ids = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    ctx.comm_init(rank, world, ids[0])
    ctx.comm_selftest()
    ctx.set_option("nb", nb)
    ctx.set_option("dist_schedule", split)
    k = ("matern2", 0.7, 1.2)
    X = rand_inputs(n, 5, n)
    chol = ctx.cholesky_from_inputs(k, X, 0.1)
    np.save(os.path.join(out_dir, f"L{rank}.npy"), chol.l())
    chol.free()
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("split", [0, 1, 2], ids=["bcast", "split", "chain"])
def test_rccl_two_processes_sharded_factor(tmp_path, split):
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one process per GPU over RCCL)")
    import torch.multiprocessing as mp

    from oracle import oracle as O

    world = min(torch.cuda.device_count(), 4)
    n, nb = 2100, 256
    port = 33500 + (os.getpid() % 2000) + split
    mp.spawn(_worker, args=(world, port, n, nb, split, str(tmp_path)), nprocs=world, join=True)
    X = rand_inputs(n, 5, n)
    st, L_o, _ = O.make_cholesky_cov_matrix(("matern2", 0.7, 1.2), X, 0.1)
    for r in range(world):
        assert rel_err(np.load(tmp_path / f"L{r}.npy"), np.tril(L_o)) < 1e-9
