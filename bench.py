#!/usr/bin/env python3
"""bench.py -- GP fit + predict on synthetic N x d f64 data through the C ABI (libfriedrich_amd.so).

One "step" = one pass of the hot path over one batch of synthetic input:
    fit      Gram assembly (lower + noise^2) + blocked Cholesky      fr_chol_refactor   (algebra/mod.rs:59-92)
    predict  cross-Gram + K^-1 K* solve + mean epilogue, m queries   fr_predict_mean    (mod.rs:226-244)
Workload (BASELINE.json configs[3], the configuration the metric is quoted on; it fits one MI355X):
N = 32768, d = 16, RBF kernel with friedrich's default hyper-parameters, m = 4096 query rows.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 3 --warmup 1

N > 1: one process per GPU.  The factorisation is sharded (block-cyclic column panels; the chain of diagonal
blocks travels by point-to-point fan-out, the rows below by scatter / per-rank solves / all-gather over xGMI,
every rank ends with the full factor: DESIGN.md section 6); the m query rows are split across ranks.  Total work
is fixed => "scaling": "strong".  Inputs (X, y, X*) are resident in HBM before the timed region starts.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the FP64-MFMA trailing SYRK update) with
HIP events recorded inside the library on the stream the kernel runs on; `cpu_baseline` times the CPU oracle
(oracle/, the restatement of the reference's nalgebra path) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PMC_NB = 1024  # block size of the run profiles/r03/fit32k_counters.json was collected with
PMC_FILE = os.path.join("profiles", "r03", "fit32k_counters.json")
PEAK_F64_MFMA_TFLOPS = 78.6  # MI355X datasheet FP64 matrix peak; scripts/mfma_f64_peak measures 77.0-77.6 on the box


def flops_fit(n, d):
    return n ** 3 / 3.0 + 0.5 * n * (n + 1) * (3.0 * d + 20.0)


def flops_predict(n, m, d):
    return n * m * (3.0 * d + 20.0) + 2.0 * n * n * m + 2.0 * n * m


def host_description():
    model, cores = "unknown CPU", os.cpu_count() or 1
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, cores


def strong_cpu_line(n):
    """A LAPACK blocked multi-thread Cholesky (scipy / OpenBLAS dpotrf) -- NOT the reference's algorithm and NOT a tuned CPU
    baseline: an untuned reference point on the cores of ONE socket (the process is pinned to the first 32 hardware threads
    for the call: round 2 ran it unpinned on 64 threads of a two-socket host and got less than an 8-vCPU container does)."""
    try:
        import scipy.linalg as sl
    except Exception:
        return None
    rng = np.random.default_rng(0)
    Q = rng.standard_normal((n, 64))
    A = Q @ Q.T + n * np.eye(n)
    threads = min(os.cpu_count() or 1, 32)
    old_aff = None
    try:
        old_aff = os.sched_getaffinity(0)
        os.sched_setaffinity(0, set(sorted(old_aff)[:threads]))
    except (AttributeError, OSError):
        old_aff = None
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=threads)
    except Exception:
        limit, threads = None, os.cpu_count() or 1
    try:
        sl.cholesky(A[:1024, :1024], lower=True)
        t0 = time.perf_counter()
        sl.cholesky(A, lower=True, overwrite_a=True, check_finite=False)
        dt = time.perf_counter() - t0
    finally:
        if limit is not None:
            limit.restore_original_limits()
        if old_aff is not None:
            os.sched_setaffinity(0, old_aff)
    return {"what": f"UNTUNED reference point: scipy.linalg.cholesky (OpenBLAS dpotrf, {threads} threads pinned to the first {threads} "
                    f"hardware threads) of a {n} x {n} matrix", "seconds": dt, "GFLOP/s": n ** 3 / 3.0 / dt / 1e9, "threads": threads}


def cpu_baseline(n, d, m, cfg):
    """The reference's CPU path (oracle restatement, one thread) on a bounded sample of the workload."""
    from friedrich_amd import synth
    from oracle import oracle as O

    O.set_threads(1)  # nalgebra is single-threaded: so is the baseline
    X, y, Xq = synth.make_problem(n, d, cfg=cfg, m=m)
    ls = O.fit_bandwidth_mean(X[:1024])  # heuristic on a sub-sample: only conditions the sample problem
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    t0 = time.perf_counter()
    gp = O.OracleGP(O.ConstantPrior(hp["prior"]), k, hp["noise"], None, X, y)
    t1 = time.perf_counter()
    gp.predict(Xq)
    t2 = time.perf_counter()
    fl = flops_fit(n, d) + flops_predict(n, m, d)
    model, cores = host_description()
    return {
        "value": fl / (t2 - t0) / 1e9,
        "unit": "GFLOP/s",
        "cores": 1,
        "kind": "port",
        "sample": f"N={n} d={d} m={m} same generator/kernel, oracle fit {t1 - t0:.1f}s + predict {t2 - t1:.1f}s, 1 thread "
                  f"(the reference's nalgebra path is single-threaded) on a host with {cores} hardware threads: {model}",
        "host": {"cpu_model": model, "hardware_threads": cores},
        "strong_cpu": strong_cpu_line(6144),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--d", type=int, default=16)
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--nb", type=int, default=0, help="outer Cholesky block; 0 = the library's choice (1024 on one GPU at this size, 512 sharded)")
    ap.add_argument("--dist-schedule", type=int, default=-1, help="N > 1, how a panel step travels: 0 = one broadcast per panel; 1 = diagonal block broadcast + rows scattered / solved per rank / all-gathered; 2 = as 1 with the diagonal chain running ahead; -1 = the library's default (2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-n", type=int, default=5120)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)

    import torch
    import torch.distributed as dist

    from friedrich_amd import synth
    from friedrich_amd.device import Context

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # FRIEDRICH_BENCH_FORCE_DIST=1 takes the multi-process set-up (process group, ncclUniqueId hand-over, communicator
    # self-test) with a single rank too: the only way to exercise it on a 1-GPU box
    use_dist = world > 1 or os.environ.get("FRIEDRICH_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)

    ctx = Context(local_rank)
    ctx.set_option("nb", args.nb)
    if args.dist_schedule >= 0:
        ctx.set_option("dist_schedule", args.dist_schedule)
    dist_sched = args.dist_schedule if args.dist_schedule >= 0 else int(os.environ.get("FRIEDRICH_AMD_DIST_SCHEDULE", "2"))
    nb_eff = args.nb if args.nb > 0 else (1024 if (world == 1 and args.n >= 24576) else 512)
    if use_dist:
        ids = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.comm_init(rank, world, ids[0])
        ctx.comm_selftest()

    n, d, m = args.n, args.d, args.m
    cfg = 4
    X, y, Xq = synth.make_problem(n, d, cfg=cfg, m=m)
    # friedrich's builder defaults (builder.rs:73, kernel.rs:594-600); the bandwidth heuristic itself runs on the GPU
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    kernel = ("squared_exp", hp["ls"], hp["ampl"])
    noise = hp["noise"]

    # query rows are sharded across ranks; everything the timed region touches is resident in HBM
    lo, hi = (m * rank) // world, (m * (rank + 1)) // world
    m_loc = hi - lo
    Xq_d = torch.from_numpy(np.ascontiguousarray(Xq[lo:hi].T)).to(dev).t() if m_loc > 0 else torch.empty((0, d), dtype=torch.float64, device=dev)
    y_d = torch.from_numpy(y - hp["prior"]).to(dev)
    prior_d = torch.full((m_loc,), hp["prior"], dtype=torch.float64, device=dev)
    mean_d = torch.empty((m_loc,), dtype=torch.float64, device=dev)
    X_d = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev).t()
    chol = ctx.cholesky_from_inputs(kernel, X_d, noise, capacity_hint=n)  # allocates + first (untimed) factorisation

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    fit_ms, pred_ms = [], []

    def step(record):
        t0 = time.perf_counter()
        chol.refactor(kernel, noise)  # Gram + Cholesky (host returns after the status read-back)
        t1 = time.perf_counter()
        if m_loc > 0:
            chol.predict_mean(kernel, y_d, Xq_d, prior_d, out=mean_d)
        ctx.synchronize()
        t2 = time.perf_counter()
        if record:
            fit_ms.append(1e3 * (t1 - t0))
            pred_ms.append(1e3 * (t2 - t1))

    for _ in range(args.warmup):
        step(False)
    ctx.profile_reset()
    ctx.profile_enable(True, classes=["syrk"])
    sync()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync()
    elapsed = time.perf_counter() - t_start
    prof = ctx.profile()
    ctx.profile_enable(False)

    # outside the timed region: the same predict associated as K*^T (K^-1 y) (option predict_assoc = 1; two n x 1 solves
    # instead of two n x m ones) -- reported next to the reference's association, not part of `value`
    alpha_ms = None
    if m_loc > 0:
        ctx.set_option("predict_assoc", 1)
        chol.predict_mean(kernel, y_d, Xq_d, prior_d, out=mean_d)
        ctx.synchronize()
        t0 = time.perf_counter()
        chol.predict_mean(kernel, y_d, Xq_d, prior_d, out=mean_d)
        ctx.synchronize()
        alpha_ms = 1e3 * (time.perf_counter() - t0)
        ctx.set_option("predict_assoc", 0)

    # more numbers outside the timed region (rank 0's shard): single-point latency (the Bayesian-optimisation inner loop of
    # readme.md:7), a handful of points, likelihood, the cached-alpha predict, and the fit with the host -> device staging of
    # the training inputs inside the clock
    extras = {}
    if m_loc > 0 and rank == 0:
        def best_ms(fn, reps=3):
            fn()
            ctx.synchronize()
            b = 1e30
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                ctx.synchronize()
                b = min(b, time.perf_counter() - t0)
            return 1e3 * b

        for mm in (1, 16, 32, 64):
            if mm > m_loc:
                continue
            q = Xq_d[:mm]
            if mm <= 16:
                extras[f"predict_m{mm}_ms"] = best_ms(lambda: chol.predict_mean(kernel, y_d, q, prior_d[:mm], out=mean_d[:mm]))
            var_d = torch.empty((mm,), dtype=torch.float64, device=dev)
            extras[f"predict_variance_m{mm}_ms"] = best_ms(lambda: chol.predict_variance(kernel, q, out=var_d))
        extras["likelihood_ms"] = best_ms(lambda: chol.likelihood(kernel, y_d, noise))
        chol.set_targets(y_d)
        extras["predict_ms_cached_alpha"] = best_ms(lambda: chol.predict_mean(kernel, None, Xq_d, prior_d, out=mean_d))
        if world == 1:
            # one optimizer iteration's gradient terms (optimizer.rs:159-203: K^-1 = W^T W and the fused reductions); the
            # refactor that precedes it in fit_parameters is `fit_ms` without the Gram heuristics
            npar = 2
            extras["grad_terms_ms"] = best_ms(lambda: chol.grad_terms(kernel, y_d, noise, True, npar), reps=1)
            Xrow = np.ascontiguousarray(X)  # the caller's row-major samples (ndarray / Vec<Vec<f64>> of the reference)

            def fit_from_host():
                xs = ctx.inputs_to_device(Xrow, "rowmajor")
                c2 = ctx.cholesky_from_inputs(kernel, xs, noise, capacity_hint=n)
                c2.free()
                xs.free()

            extras["fit_ms_h2d_inclusive"] = best_ms(fit_from_host, reps=2)

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    info = chol.info()
    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(args.steps, 1)
        total_flops = flops_fit(n, d) + flops_predict(n, m, d)
        syrk = prof["syrk"]
        achieved = syrk["flops"] / max(syrk["ms"], 1e-9) / 1e9  # TFLOP/s
        # PMC counters cannot be read from inside the process: the figure is the committed rocprofv3 --pmc summary of the
        # same workload (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 fetch correction); null for any other size
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, PMC_FILE)
        if os.path.exists(pmc_path) and (n, d, nb_eff, world) == (32768, 16, PMC_NB, 1):
            with open(pmc_path) as f:
                traffic = json.load(f)["kernels"]["fr::syrk_lower_f64_kernel"]["total_bytes_per_launch"]  # same kernel code as this build: regenerate (scripts/profile_r03.sh) whenever gemm_f64.hip / gemm_tile.hpp change
            traffic_src = PMC_FILE
        out = {
            "metric": "gp_fit_predict_gflops",
            "value": total_flops / (ms_per_step * 1e-3) / 1e9,
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"GP fit (Gram + Cholesky) + predict, N={n} d={d} RBF, m={m} queries, friedrich default hyper-parameters",
                "n": n, "d": d, "m": m, "kernel": "squared_exp", "nb": nb_eff,
                "parallelism": "1 GPU" if world == 1 else f"block-cyclic column panels over {world} GPUs (RCCL, " + ["one broadcast per panel", "diagonal block broadcast + scatter / all-gather of the rows below", "diagonal chain first: block to the next owner, then scatter / all-gather of the rows below"][dist_sched] + "), queries sharded",
            },
            "fit_ms": float(np.mean(fit_ms)),
            "predict_ms": float(np.mean(pred_ms)),
            "predict_ms_alpha_assoc": alpha_ms,
            **extras,
            "cholesky_tflops": (n ** 3 / 3.0) / (np.mean(fit_ms) * 1e-3) / 1e12,
            "n_substitutions": info["n_subst"],
            "roofline": {
                "kernel": "syrk_lower_f64_kernel (trailing SYRK update of the blocked Cholesky, v_mfma_f64_16x16x4_f64)",
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_F64_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_F64_MFMA_TFLOPS,
                "traffic": traffic,
                "traffic_unit": "bytes per launch (L2 memory-side requests, Infinity-Cache hits included)",
                "traffic_source": traffic_src,
                "launches": syrk["launches"],
                "avg_launch_ms": syrk["ms"] / max(syrk["launches"], 1),
                "flops_per_launch": syrk["flops"] / max(syrk["launches"], 1),
            },
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed at N = 1 only (rank 0's host cores)
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_n, d, 256, cfg)
        print(json.dumps(out), flush=True)

    chol.free()
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
