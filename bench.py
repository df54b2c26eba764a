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
Before the timed region every candidate schedule (2 -> 1 -> 0) has to get a sharded fit of 4096 rows right -- compared
with a single-rank fit on the same GPU -- under the library's collective time-out; a schedule that fails or times out is
dropped, the communicators are rebuilt, the next one is tried, and if none works the ranks run as replicas.  The line
says which (`schedule_used`, `fallback_reason`) and carries one untimed step's per-class times of every rank
(`per_rank`: comm / potf2 / panel / syrk) next to the critical-path model of DESIGN.md section 6.

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

PMC_NB = 1024  # block size of the run profiles/r06/fit32k_counters.json was collected with
PMC_FILE = os.path.join("profiles", "r06", "fit32k_counters.json")
PEAK_F64_MFMA_TFLOPS = 78.6  # MI355X datasheet FP64 matrix peak; measured_mfma_peak() times the instruction loop on the box (77.0-77.6)
MFMA_PEAK_PROBE = os.path.join(ROOT, "friedrich_amd", "lib", "mfma_f64_peak")  # scripts/mfma_f64_peak.hip, built by friedrich_amd/build.py


def measured_mfma_peak():
    """"measured MFMA-loop TFLOP/s on the box AND datasheet; both reported".  Runs the prebuilt instruction
    loop (v_mfma_f64_16x16x4_f64 back to back, the chip filled at several occupancies) as a process of its own, before the timed
    region, and returns the best chip-filling rate in TFLOP/s -- None when the helper is missing or fails (never a guess)."""
    import re
    import subprocess

    if not os.path.exists(MFMA_PEAK_PROBE):
        return None
    try:
        r = subprocess.run([MFMA_PEAK_PROBE], capture_output=True, text=True, timeout=60)
    except (OSError, subprocess.TimeoutExpired):
        return None
    rates = [float(v) for b, v in re.findall(r"blocks=\s*(\d+) waves/block=\d+:\s*([0-9.]+) TFLOP/s", r.stdout) if int(b) >= 256]
    rates = [v for v in rates if 1.0 < v < 1.25 * PEAK_F64_MFMA_TFLOPS]  # (a launch that failed reads as an absurd rate)
    return max(rates) if rates else None


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


BENCH_N, BENCH_M = 32768, 4096  # the workload the line is quoted on (what the CPU sample is extrapolated to)


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
    # "time it fully for N <= 8192 and extrapolate ~ n^3 beyond (label as extrapolated)": the wall time the
    # same single-thread path would need for the bench's own workload, next to the GPU's step time
    fit_x = (t1 - t0) * (BENCH_N / n) ** 3
    pred_x = (t2 - t1) * (BENCH_N / n) ** 2 * (BENCH_M / m)
    return {
        "extrapolated_s_at_n32768": {
            "fit": fit_x, "predict": pred_x, "step": fit_x + pred_x, "label": "EXTRAPOLATED, not measured",
            "method": f"fit time x ({BENCH_N}/{n})^3 (n^3/3 dominates), predict time x ({BENCH_N}/{n})^2 x ({BENCH_M}/{m}) (2 n^2 m dominates), "
                      f"from the timed N={n}, m={m} sample; the N = {BENCH_N} run itself would take hours on one core",
        },
        "value": fl / (t2 - t0) / 1e9,
        "unit": "GFLOP/s",
        "cores": 1,
        "kind": "port",
        "sample": f"N={n} d={d} m={m} same generator/kernel, oracle fit {t1 - t0:.1f}s + predict {t2 - t1:.1f}s, 1 thread "
                  f"(the reference's nalgebra path is single-threaded) on a host with {cores} hardware threads: {model}",
        "host": {"cpu_model": model, "hardware_threads": cores},
        "strong_cpu": strong_cpu_line(6144),
    }


# the critical-path model of the chain-first schedule at N = 32768, nb = 512, W = 8 (microseconds per
# panel; compute terms measured on one idle GPU, transfers priced at 45 GB/s per xGMI link + 20 us per RCCL call) -- printed
# next to the measured per-class times of an N > 1 run so that the first scaling curve says where its time went
MODEL_TERMS_US = {"diag_block_D_p": 156, "R1_solve": 52, "M_p_fanout": 65, "u1": 34, "chain_step": 320,  # profiles/r06/dist_model.txt (D_p: ONE resident launch)
                  "bulk_latency_first": 980, "bulk_latency_last": 390, "rank_share_of_update_k0": 1070,
                  "predicted_fit_ms_8gpus": "35-38 (schedule 2), ~60 (schedule 1), 100-130 (schedule 0)"}


def run_rank(args, link, device_index, emit, mode):
    """One rank of the bench: mode "process" (one process per GPU, RCCL), "threads" (thread-ranks sharing ONE GPU over the
    library's in-process transport: exercises every line of the N > 1 path on a 1-GPU box, not a measurement) or "solo"."""
    import torch

    from friedrich_amd import sharding, synth
    from friedrich_amd.device import Context, FriedrichError

    rank, world = link.rank, link.world
    dev = torch.device("cuda", device_index)
    log = (lambda msg: print(f"bench.py: {msg}", file=sys.stderr, flush=True)) if rank == 0 else (lambda msg: None)
    ctx = Context(device_index)
    ctx.set_option("nb", args.nb)
    nb_eff = args.nb if args.nb > 0 else (1024 if (world == 1 and args.n >= 18432) else 512)
    sharded = world > 1 or mode == "process_forced"

    # ---- N > 1: pick the schedule behind a preflight, under the library's watchdog -------------------------------------
    schedule, fallback, preflight_ms = None, [], {}
    if sharded:
        link.attach(ctx)  # communicator + collective self-test
        first = args.dist_schedule if args.dist_schedule >= 0 else int(os.environ.get("FRIEDRICH_BENCH_FIRST_SCHEDULE", "2"))
        candidates = [s for s in (2, 1, 0) if s <= first]
        if world > 1:
            ref_ctx = Context(device_index)
            try:
                schedule, fallback, preflight_ms = sharding.guarded_schedule(
                    ctx, link, lambda s: sharding.preflight_fit(ctx, ref_ctx, n=args.preflight_n), schedules=candidates,
                    timeout_ms=args.preflight_timeout_ms, log=log)
            finally:
                ref_ctx.close()
        else:
            schedule = candidates[0]
            ctx.set_option("dist_schedule", schedule)
        ctx.set_option("comm_timeout_ms", args.comm_timeout_ms)

    peak_measured = measured_mfma_peak() if (rank == 0 and mode == "solo") else None
    n, d, m = args.n, args.d, args.m
    cfg = 4
    X, y, Xq = synth.make_problem(n, d, cfg=cfg, m=m)
    # friedrich's builder defaults (builder.rs:73, kernel.rs:594-600); the bandwidth heuristic itself runs on the GPU
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    kernel = ("squared_exp", hp["ls"], hp["ampl"])
    noise = hp["noise"]

    # query rows are sharded across ranks; everything the timed region touches is resident in HBM
    lo, hi = sharding.query_slice(m, rank, world)
    m_loc = hi - lo
    Xq_d = torch.from_numpy(np.ascontiguousarray(Xq[lo:hi].T)).to(dev).t() if m_loc > 0 else torch.empty((0, d), dtype=torch.float64, device=dev)
    y_d = torch.from_numpy(y - hp["prior"]).to(dev)
    prior_d = torch.full((m_loc,), hp["prior"], dtype=torch.float64, device=dev)
    mean_d = torch.empty((m_loc,), dtype=torch.float64, device=dev)
    # (--verify's reference output is allocated HERE, with everything else and before anything is freed: a block that torch's
    # caching allocator hands out again is only safe in the order of TORCH's stream, and the library writes on streams of its
    # own -- with thread-ranks sharing one allocator, a buffer allocated late could be a block whose previous owner, a
    # temporary of another thread, still had a kernel pending: one rank in six runs "deviated" by 0.6 that way in round 5)
    want_d = torch.empty((m_loc,), dtype=torch.float64, device=dev) if args.verify else None
    X_d = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev).t()

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize(dev)
        sharding.sync_point(link)  # (raises PeerFailure when a peer reports a failed step instead of arriving)
        if mode.startswith("process") and sharded:
            import torch.distributed as dist
            dist.barrier()  # (the contract's barrier over RCCL; link.barrier is the control plane's)

    def measure():
        """first (untimed) factorisation, warm-up, the timed steps, one profiled step -> dict; raises FriedrichError"""
        chol = ctx.cholesky_from_inputs(kernel, X_d, noise, capacity_hint=n)  # allocates + first (untimed) factorisation
        try:
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
            ctx.profile_enable(True, classes=["syrk", "syrk_chain"])
            sync()
            t_start = time.perf_counter()
            for _ in range(args.steps):
                step(True)
            sync()
            elapsed = time.perf_counter() - t_start
            prof = ctx.profile()
            # one more, untimed step with every class timed (HIP events around every launch: they perturb the step, hence
            # outside the clock): where the time of a step goes -- collectives (incl. the wait for the peers), diagonal-block
            # kernels, panel products, trailing updates
            ctx.profile_reset()
            ctx.profile_enable(True)
            t0 = time.perf_counter()
            step(False)
            prof_step_ms = 1e3 * (time.perf_counter() - t0)
            classes = ctx.profile()
            ctx.profile_enable(False)
            return {"chol": chol, "elapsed": elapsed, "fit_ms": fit_ms, "pred_ms": pred_ms, "prof": prof, "classes": classes,
                    "prof_step_ms": prof_step_ms}
        except BaseException:
            chol.free()
            raise

    res = None
    while res is None:
        agreed = False
        try:
            res = measure()
            ok, why = True, None
        except sharding.PeerFailure as e:
            ok, why, agreed = False, str(e), True  # (the rendezvous inside measure() WAS the agreement)
        except FriedrichError as e:
            if not sharded or world == 1 or schedule is None or schedule < 0:
                raise
            ok, why = False, str(e)
        except BaseException:
            link.abort()
            raise
        if sharded and world > 1 and schedule is not None and schedule >= 0:
            if not agreed:
                ok, why = sharding.agree(link, ok, why)
            if not ok:
                if res is not None:
                    res["chol"].free()
                    res = None
                fallback.append(f"schedule {schedule} at N={n}: {why}")
                log(f"sharded schedule {schedule} failed on the bench workload ({why}); falling back")
                lower = [s for s in (1, 0) if s < schedule]
                err = sharding.reattach(ctx, link) if lower else "no schedule left"
                if err is None:
                    ctx.set_option("comm_timeout_ms", args.comm_timeout_ms)
                    schedule = lower[0]
                    ctx.set_option("dist_schedule", schedule)
                else:
                    if lower:
                        fallback.append(err)
                    ctx.comm_finalize(abort=True)
                    schedule = -1
    chol, elapsed, fit_ms, pred_ms, prof = res["chol"], res["elapsed"], res["fit_ms"], res["pred_ms"], res["prof"]

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
    if m_loc > 0 and rank == 0 and not args.no_extras:
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

        for mm in (1, 16, 32, 64, 128, 256):
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
        if world == 1 and mode == "solo":
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

    # sharded: one optimizer iteration's gradient terms with the rows of L^-1 dealt to the ranks (collective: every rank calls;
    # outside the timed region) -- per rank in `per_rank`
    grad_ms = None
    if sharded and world > 1 and schedule is not None and schedule >= 0 and not args.no_extras:
        try:
            chol.grad_terms(kernel, y_d, noise, True, 2)
            ctx.synchronize()
            t0 = time.perf_counter()
            chol.grad_terms(kernel, y_d, noise, True, 2)
            ctx.synchronize()
            grad_ms = 1e3 * (time.perf_counter() - t0)
        except FriedrichError as e:
            log(f"sharded grad_terms failed: {e}")

    # --verify: this rank's share of the predictions against a single-rank fit of the same rows on the same GPU (tests)
    verify_err = None
    if args.verify and m_loc > 0:
        ref_ctx = Context(device_index)
        ref = ref_ctx.cholesky_from_inputs(kernel, X_d, noise)
        ref.predict_mean(kernel, y_d, Xq_d, prior_d, out=want_d)
        chol.predict_mean(kernel, y_d, Xq_d, prior_d, out=mean_d)
        ref_ctx.synchronize()
        ctx.synchronize()
        got_h, want_h = mean_d.cpu().numpy(), want_d.cpu().numpy()  # (compared on the host: no device temporaries)
        verify_err = float(np.max(np.abs(got_h - want_h)) / np.max(np.abs(want_h)))
        ref.free()
        ref_ctx.close()

    # MAX over ranks of the timed region; every rank's per-class times of the profiled step
    per_rank = link.gather({"elapsed": elapsed, "fit_ms": float(np.mean(fit_ms)), "pred_ms": float(np.mean(pred_ms)),
                            "classes": {k: round(v["ms"], 3) for k, v in res["classes"].items()},
                            "launches": {k: v["launches"] for k, v in res["classes"].items()},
                            "prof_step_ms": res["prof_step_ms"], "comm_timeouts": ctx.counter("comm_timeouts"), "verify": verify_err,
                            "grad_terms_ms": grad_ms})
    elapsed = max(r["elapsed"] for r in per_rank)

    info = chol.info()
    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(args.steps, 1)
        total_flops = flops_fit(n, d) + flops_predict(n, m, d)
        syrk = prof["syrk"]
        achieved = syrk["flops"] / max(syrk["ms"], 1e-9) / 1e9  # TFLOP/s
        # the trailing updates of the chain-bound tail of a fit (at most 16384 trailing rows, CUs set aside for the panel chain) are
        # launches of another kernel symbol -- resident workgroups that claim tiles, syrk_lower_persist_f64_kernel -- and a profile
        # class of their own, so that `roofline` stays "the launches of syrk_lower_f64_kernel" (what rocprofv3's per-kernel
        # average counts); they are reported next to it, and so is the rate over ALL trailing-update launches of a fit
        chain = prof.get("syrk_chain", {"flops": 0.0, "ms": 0.0, "launches": 0})
        all_ms, all_flops = syrk["ms"] + chain["ms"], syrk["flops"] + chain["flops"]
        # PMC counters cannot be read from inside the process: the figure is the committed rocprofv3 --pmc summary of the
        # same workload (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 fetch correction); null for any other size
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, PMC_FILE)
        if os.path.exists(pmc_path) and (n, d, nb_eff, world) == (32768, 16, PMC_NB, 1):
            with open(pmc_path) as f:
                traffic = json.load(f)["kernels"]["fr::syrk_lower_f64_kernel"]["total_bytes_per_launch"]  # same kernel code as this build: tests/test_profiles_fresh.py fails when gemm_f64.hip / gemm_tile.hpp changed after the passes (scripts/profile_r06.sh)
            traffic_src = PMC_FILE
        if not sharded:
            parallelism = "1 GPU"
        else:
            what = f"{world} thread-ranks on ONE GPU over the in-process transport (TEST MODE: exercises the N > 1 path, not a measurement)" \
                if mode == "threads" else f"{world} GPUs (RCCL)"
            parallelism = (f"block-cyclic column panels over {what}, {sharding.SCHEDULE_NAMES[schedule]}, queries sharded" if schedule >= 0
                           else f"{what}: {sharding.SCHEDULE_NAMES[-1]}")
        out = {
            "metric": "gp_fit_predict_gflops",
            "value": total_flops / (ms_per_step * 1e-3) / 1e9,
            "unit": "GFLOP/s",
            "n_gpus": 1 if mode == "threads" else world,
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
                "panel_widths": ("2048 while more than 22528 rows remain, 1024 down to 16384, then 512" if (args.nb == 0 and nb_eff == 1024)
                                 else f"{nb_eff}" + (" down to 16384 rows, then 512" if (nb_eff > 512 and world == 1) else "")),
                "parallelism": parallelism,
            },
            "fit_ms": float(np.mean(fit_ms)),
            "predict_ms": float(np.mean(pred_ms)),
            "predict_ms_alpha_assoc": alpha_ms,
            **extras,
            "cholesky_tflops": (n ** 3 / 3.0) / (np.mean(fit_ms) * 1e-3) / 1e12,
            "n_substitutions": info["n_subst"],
            # one untimed step with HIP events around every launch of every class (rank 0; per_rank below has all of them)
            "step_breakdown_ms": {"step_with_events": res["prof_step_ms"], **per_rank[0]["classes"]},
            "roofline": {
                "kernel": "syrk_lower_f64_kernel (trailing SYRK update of the blocked Cholesky, v_mfma_f64_16x16x4_f64)",
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_F64_MFMA_TFLOPS,
                "peak_source": "datasheet (MI355X FP64 matrix, dense)",
                "peak_measured": peak_measured,
                "peak_measured_source": "v_mfma_f64_16x16x4_f64 instruction loop on this box before the timed region (scripts/mfma_f64_peak.hip)",
                "frac_of_measured_peak": (achieved / peak_measured) if peak_measured else None,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_F64_MFMA_TFLOPS,
                "traffic": traffic,
                "traffic_unit": "bytes per launch (L2 memory-side requests, Infinity-Cache hits included)",
                "traffic_source": traffic_src,
                "launches": syrk["launches"],
                "avg_launch_ms": syrk["ms"] / max(syrk["launches"], 1),
                "flops_per_launch": syrk["flops"] / max(syrk["launches"], 1),
                "chain_phase_launches": {
                    "kernel": "syrk_lower_persist_f64_kernel (the same tile code as resident workgroups; chain-bound tail of the fit)",
                    "launches": chain["launches"], "avg_launch_ms": chain["ms"] / max(chain["launches"], 1),
                    "achieved": chain["flops"] / max(chain["ms"], 1e-9) / 1e9,
                    "frac": chain["flops"] / max(chain["ms"], 1e-9) / 1e9 / PEAK_F64_MFMA_TFLOPS,
                },
                "all_trailing_updates": {"launches": syrk["launches"] + chain["launches"], "achieved": all_flops / max(all_ms, 1e-9) / 1e9,
                                         "frac": all_flops / max(all_ms, 1e-9) / 1e9 / PEAK_F64_MFMA_TFLOPS},
            },
        }
        if sharded:
            out["schedule_used"] = schedule
            out["fallback_reason"] = "; ".join(fallback) if fallback else None
            out["preflight"] = {"n": args.preflight_n, "timeout_ms": args.preflight_timeout_ms,
                                "ms_by_schedule": {str(k): round(v, 1) for k, v in preflight_ms.items()}}
            out["per_rank"] = {
                "fit_ms": [round(r["fit_ms"], 3) for r in per_rank], "predict_ms": [round(r["pred_ms"], 3) for r in per_rank],
                "comm_ms": [r["classes"]["comm"] for r in per_rank], "potf2_ms": [r["classes"]["potf2"] for r in per_rank],
                "panel_ms": [r["classes"]["gemm_panel"] for r in per_rank], "syrk_ms": [r["classes"]["syrk"] for r in per_rank],
                "gram_ms": [r["classes"]["gram"] for r in per_rank], "solve_ms": [r["classes"]["gemm_solve"] for r in per_rank],
                "comm_calls": [r["launches"]["comm"] for r in per_rank], "step_with_events_ms": [round(r["prof_step_ms"], 3) for r in per_rank],
                "comm_timeouts": [r["comm_timeouts"] for r in per_rank],
                "grad_terms_ms": [None if r["grad_terms_ms"] is None else round(r["grad_terms_ms"], 3) for r in per_rank],
                "note": "one untimed step with HIP events around every launch; comm_ms includes the wait for the peers; classes on "
                        "different streams overlap, so they do not add up to the step",
            }
            out["model_terms_us_n32768_w8"] = MODEL_TERMS_US
        if args.verify:
            out["verify_rel_err"] = [r["verify"] for r in per_rank]
        if not args.no_cpu_baseline and world == 1 and mode == "solo":  # the CPU leg is timed at N = 1 only (rank 0's host cores)
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_n, d, 256, cfg)
        emit(json.dumps(out))

    chol.free()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--d", type=int, default=16)
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--nb", type=int, default=0, help="outer Cholesky block; 0 = the library's choice (1024 on one GPU at this size, 512 sharded)")
    ap.add_argument("--dist-schedule", type=int, default=-1, help="N > 1, the FIRST schedule tried (the preflight falls back to the lower ones): 0 = one broadcast per panel; 1 = diagonal block broadcast + rows scattered / solved per rank / all-gathered; 2 = as 1 with the diagonal chain running ahead; -1 = 2")
    ap.add_argument("--preflight-n", type=int, default=4096, help="N > 1: rows of the sharded fit every candidate schedule has to get right first")
    ap.add_argument("--preflight-timeout-ms", type=int, default=20000, help="N > 1: the library's collective time-out during the preflight")
    ap.add_argument("--comm-timeout-ms", type=int, default=60000, help="N > 1: the library's collective time-out during the measurement")
    ap.add_argument("--local-ranks", type=int, default=0, help="TEST MODE: that many thread-ranks on ONE GPU over the library's in-process transport instead of one process per GPU")
    ap.add_argument("--verify", action="store_true", help="after the measurement, compare every rank's predictions with a single-rank fit on the same GPU (verify_rel_err)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra measurements (latencies of small predicts, gradient terms)")
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

    from friedrich_amd import sharding

    emit = lambda line: print(line, flush=True)
    if args.local_ranks > 1:
        # thread-ranks: the whole N > 1 path of this file (preflight, fall-back, per-rank breakdown) on the one GPU of a test box
        import threading

        torch.cuda.set_device(local_rank)
        shared = sharding.ThreadShared(args.local_ranks)
        errors = []

        def worker(r):
            try:
                run_rank(args, sharding.ThreadLink(shared, r), local_rank, emit, "threads")
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
                shared.barrier.abort()

        threads = [threading.Thread(target=worker, args=(r,)) for r in range(args.local_ranks)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return

    torch.cuda.set_device(local_rank)
    # FRIEDRICH_BENCH_FORCE_DIST=1 takes the multi-process set-up (process group, ncclUniqueId hand-over, communicator
    # self-test) with a single rank too: the only way to exercise it on a 1-GPU box
    use_dist = world > 1 or os.environ.get("FRIEDRICH_BENCH_FORCE_DIST") == "1"
    if not use_dist:
        run_rank(args, sharding.SoloLink(), local_rank, emit, "solo")
        return
    import datetime

    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    # control plane: a gloo group (CPU side) -- agreement between the ranks must not depend on the communicator under test
    ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=600))
    try:
        run_rank(args, sharding.TorchLink(dist, rank, world, ctl), local_rank, emit, "process" if world > 1 else "process_forced")
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
