#!/usr/bin/env python3
"""CAVIaR iterations/s on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg5|nwd] [--no-cpu-baseline]

`--workload nwd` benchmarks the upstream NWD demixer forward pass instead (its own metric, traces/s; see bench_nwd).

A "step" is one complete `caviar()` run of one circuit per GPU: `iters` (50) CAVIaR iterations plus the
false-negative scan and final phi update (reference circuitmap/optimise/caviar.py:79-98), on synthetic data of the
BASELINE shape (N=1000 neurons, K=2000 trials, T=900 samples; seeded restatement of the reference's `simulate`, generated on the
GPU by default -- include/sim_hip.h -- or on the host with --host-sim).
Inputs (PSC integrals + encoded stimulus) are resident in HBM before the timed region; results stay in HBM.
value = (circuits x iterations) / wall time, whole job, max over ranks.  For N > 1 GPUs every rank fits its own
independent circuit (N circuits, seeds 2000 .. 2000 + N - 1, rotated over the ranks step by step so that every GPU sees the same
mix): embarrassingly parallel, no data-path collective ("weak" scaling).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, live HIP-event timing from the library) and
`cpu_baseline` (the literal float64 oracle timed on this host, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "cfg2": dict(N=300, K=1000, T=900, H=10),
    "cfg3": dict(N=1000, K=2000, T=900, H=10),       # the configuration BASELINE.json's metric is quoted on
    "cfg5": dict(N=5000, K=10000, T=900, H=10),
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


_REAL_STDOUT = None


def quiet_stdout():
    """Native libraries print to fd 1 too (RCCL's version banner under NCCL_DEBUG=VERSION, rocm warnings): route fd 1 to stderr for the
    whole run and keep the real stdout for the ONE JSON line."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (line + "\n").encode())


def circuit_for(rank, step, world):
    """which of the `world` circuits rank `rank` fits at step `step` (see main): a Latin-square rotation"""
    return (rank + step) % world


def default_priors(N):
    import numpy as np
    return dict(mu=np.zeros(N), beta=1e1 * np.ones(N), shape=1.0, rate=1e-1,
                phi=np.c_[1e-1 * np.ones(N), 5e0 * np.ones(N)],
                phi_cov=np.array([np.array([[1e-1, 0], [0, 1e0]]) for _ in range(N)]))


def cpu_baseline(sim, pri, wl):
    """The literal float64 oracle (operation-for-operation restatement of the reference, O(N^2 K) sweep) timed on
    this host for ONE iteration (iteration 0: full active set, the most expensive one), plus the running-residual
    formulation for a few iterations as a stronger baseline.  Bounded to roughly 30-40 s of CPU work."""
    from oracle import caviar_oracle as co
    args = (sim["psc"], sim["stim_matrix"], pri["mu"], pri["beta"], pri["shape"], pri["rate"], pri["phi"], pri["phi_cov"])
    try:
        import threadpoolctl
        nthreads = max(i["num_threads"] for i in threadpoolctl.threadpool_info()) if threadpoolctl.threadpool_info() else 1
    except Exception:
        nthreads = os.cpu_count()
    out = {}
    if wl["N"] <= 1000:
        t = time.time()
        co.caviar(*args, iters=1, seed=0, literal=True, fn_scan=False)
        dt = time.time() - t
        out.update(value=1.0 / dt, unit="iterations/s", cores=int(nthreads), kind="port",
                   sample="literal O(N^2 K) oracle, 1 iteration (it=0) of 50, %.1f s; cores = BLAS threads, sweep loop is single-threaded NumPy" % dt)
    n_fast = 6 if wl["N"] <= 1000 else 1
    t = time.time()
    co.caviar(*args, iters=n_fast, seed=0, literal=False, fn_scan=False)
    dtf = time.time() - t
    out["fast_formulation"] = dict(value=n_fast / dtf, unit="iterations/s",
                                   sample="running-residual oracle, first %d iterations, %.1f s" % (n_fast, dtf))
    if "value" not in out:
        out.update(value=n_fast / dtf, unit="iterations/s", cores=int(nthreads), kind="port",
                   sample="running-residual oracle (literal form too slow at this size), first %d iterations, %.1f s" % (n_fast, dtf))
    return out


def nwd_flops_per_trace(T):
    """multiply-adds x 2 of the reference network for one trace of T samples (neural_waveform_demixing.py:258-268), as dense
    PyTorch would count them: Conv1d = out_len * cout * cin * k; ConvTranspose1d = in_len * cin * cout * k."""
    pool = lambda L: (L - 3) // 2 + 1
    P0 = pool(T); E1 = P0 - 62; P1 = pool(E1); E2 = P1 - 31; P2 = pool(E2); E3 = P2 - 15; P3 = pool(E3); E4 = P3 - 15
    mac = (E1 * 16 * 1 * 32 + E2 * 16 * 16 * 32 + E3 * 32 * 16 * 16 + E4 * 32 * 32 * 16
           + E4 * 32 * 16 * 16 + E3 * 48 * 16 * 16 + E2 * 32 * 16 * 32 + E1 * 32 * 4 * 32 + T * 1 * 4 * 256)
    return 2 * mac


def bench_nwd(a):
    """One step = NeuralDemixer.__call__ on K = 2000 traces of T = 900 samples (the trace batch of cfg3), traces and result
    resident in HBM (float64 in, float64 out), seeded random parameters of the reference architecture.  The kernel is
    compute-bound on the fp32 matrix cores (157.3 TFLOP/s dense, MI355X_MICROARCH.md); its launch is timed live with HIP
    events on the library's stream."""
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    from circuitmap_amd.neural_waveform_demixing import NeuralDemixer
    from oracle import nwd_oracle as no          # parameter generator + the CPU baseline (checker / baseline only)
    K, T = 2000, 900
    state = no.random_state(seed=0)
    rng = np.random.default_rng(3000 + rank)
    traces = rng.normal(0, 0.05, (K, T)) + np.exp(-np.abs(np.arange(T)[None, :] - rng.uniform(100, 500, (K, 1))) / 60.0) * rng.uniform(0.5, 4, (K, 1))
    dm = NeuralDemixer(state=state, device=local_rank)
    ctx = dm.ctx
    d_in, d_out = ctx.alloc(traces.nbytes), ctx.alloc(traces.nbytes)
    ctx.upload(d_in, traces)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(a.warmup):
        dm.demix_device(d_in, K, T, d_out)
    barrier()
    t0 = time.perf_counter()
    kms = []
    for _ in range(a.steps):
        kms.append(dm.demix_device(d_in, K, T, d_out))
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        flops = nwd_flops_per_trace(T) * K
        k_ms = float(np.mean(kms))
        achieved = flops / (k_ms * 1e-3) / 1e12
        out = {"metric": "NWD demixer traces/sec (K=%d, T=%d)" % (K, T), "value": world * a.steps * K / dt, "unit": "traces/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "nwd: NeuralDemixer forward (U-Net + rescale + monotone filter) on %d traces x %d samples per GPU, "
                                      "seeded random parameters" % (K, T)},
               "roofline": {"bound": "mfma", "kernel": "k_nwd_forward", "achieved": achieved, "peak": 157.3, "unit": "TFLOP/s",
                            "frac": achieved / 157.3,
                            # HBM bytes per launch from the PMC passes in profiles/r01_f_nwd_pmc.csv (FETCH_SIZE 8558.6 KB doubled per the
                            # MI355X guide + WRITE_SIZE 14062.5 KB); algorithmic: 28.8 MB of float64 traces in and out
                            "traffic": int((2 * 8558.6 + 14062.5) * 1024), "avg_launch_ms": k_ms,
                            "algorithmic_flops_per_launch": int(flops),
                            "hbm_bytes_per_launch": int(K * T * 16)}}
        if world == 1 and not a.no_cpu_baseline:
            nb = 256
            nthreads = torch.get_num_threads()
            t = time.time()
            ref = no.demix(traces[:nb], state)
            dtc = time.time() - t
            got = ctx.download(d_out, np.empty_like(traces))[:nb]
            out["cpu_baseline"] = {"value": nb / dtc, "unit": "traces/s", "cores": int(nthreads), "kind": "port",
                                   "sample": "float32 PyTorch-CPU oracle on the first %d traces, %.2f s" % (nb, dtc),
                                   "max_abs_diff_vs_gpu_over_scale": float(np.max(np.abs(got - ref)) / max(np.max(np.abs(ref)), 1e-30))}
        emit(json.dumps(out))
    ctx.free(d_in); ctx.free(d_out)
    dm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS) + ["nwd"])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phases", action="store_true", help="stamp every phase of every iteration (more host API calls)")
    ap.add_argument("--circuit-seed", type=int, default=2000, help="seed of rank 0's synthetic circuit (rank r uses seed + r)")
    ap.add_argument("--host-sim", action="store_true", help="generate the synthetic experiments with the NumPy restatement on the host "
                    "(default: on the GPU, include/sim_hip.h -- psc and stim_matrix never exist on the host)")
    a = ap.parse_args()
    quiet_stdout()
    if a.workload == "nwd":
        return bench_nwd(a)

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or ("RANK" in os.environ and os.environ.get("CAVIAR_BENCH_FORCE_DIST") == "1"):
        # one process per GPU under torch.distributed.run; backend "nccl" is RCCL on ROCm.  The data path has no collective
        # (independent circuits): RCCL only carries the barriers and the max-over-ranks of the elapsed time.
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    from circuitmap_amd import _lib
    from circuitmap_amd.optimise.caviar import Problem
    from circuitmap_amd.simulation import simulate

    wl = WORKLOADS[a.workload]
    N, K, T, H = wl["N"], wl["K"], wl["T"], wl["H"]
    pri = default_priors(N)
    # One independent circuit per GPU and step.  With N ranks there are N circuits (seeds circuit_seed .. circuit_seed + N - 1), every
    # rank holds all of them in HBM and fits circuit (rank + step) mod N at step `step`: at any moment the N GPUs work on N different
    # circuits, and over the run every GPU sees the same mix, so the max-over-ranks time measures the machine, not which rank drew
    # the circuit with the slowest active-set trajectory (fit time varies +-4 % between these seeds).  N = 1: one circuit.
    gen_ms, dsims, sims, probs = None, [], [], []
    ctx = _lib.Context(local_rank) if a.host_sim else None
    for c in range(world):
        if a.host_sim:
            sim = simulate(N=N, T=T, H=H, trials=K, seed=a.circuit_seed + c, lite=(N > 2000))
            sims.append(sim)
            probs.append(Problem(ctx, sim["psc"], sim["stim_matrix"]))        # H2D + one-off prologue: outside the timed region
        else:
            from circuitmap_amd.simulation import simulate_device
            d = simulate_device(N=N, T=T, H=H, trials=K, seed=a.circuit_seed + c, device_id=local_rank, keep_on_device=True)
            ctx, gen_ms = d["ctx"], d["device_ms"]
            dsims.append(d)
            probs.append(Problem(ctx, None, None, device_ptrs=(N, K, T, d["psc_ptr"], d["stim_ptr"])))   # one-off prologue: outside the timed region
    prob = probs[0]
    nnz, nvals = prob.nnz
    run_args = (pri["mu"], pri["beta"], pri["shape"], pri["rate"], pri["phi"], pri["phi_cov"])
    opts = dict(iters=a.iters, seed=0)
    # Per-kernel HIP events cost a launch boundary each (~8 % of a fit with four per iteration), so the library records them only on
    # request.  The timed region asks for them on every 4th step: the kernel durations in `roofline` are
    # measured live inside the timed region, on a sample of its steps; `value` pays for that sample (about 2 %).
    stamped = [(s % 4 == 3) or (a.steps < 4 and s == a.steps - 1) for s in range(a.steps)]
    timing_level = 2 if a.phases else 1

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for w in range(a.warmup):
        probs[circuit_for(rank, w, world)].run(*run_args, **opts)
    for c in range(a.warmup, world):                       # every circuit of the rotation has run once before the clock starts
        probs[circuit_for(rank, c, world)].run(*run_args, **opts)
    barrier()
    t0 = time.perf_counter()
    tms = []
    for step in range(a.steps):
        tms.append(probs[circuit_for(rank, step, world)].run(*run_args, timing=timing_level if (stamped[step] or a.phases) else 0, **opts).as_dict())
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        its = a.iters
        value = world * a.steps * its / dt
        # --- dominant kernel (rocprof, profiles/): k_iter_mid, one launch per iteration, timed live by HIP events that the
        # library records on its stream around every launch (cm_timings.mid_ms).  Algorithmic bytes of one launch:
        #   block 0   y, R (8 B each), colany, mask (1 B each) per trial; mu, beta (8 B), row stats (32 B) per neuron; z written
        #   waves     per neuron: its lam row (8 + 1 B per targeted trial), counts of V stimulus values (4 B each), phi priors
        #             (48 B), phi / phi_cov written (48 B); next MC logit: inverse order (4 B), key (8 B), ndtri table row
        #             (2 S x 8 B), P logits written (8 B each)
        S = 100
        tms_k = [t for t, st_ in zip(tms, stamped) if st_ or a.phases]           # the steps that carried per-kernel events
        mid_raw_ms = float(np.mean([t["mid_ms"] for t in tms_k])) / its          # event pair around the launch, ms
        # the event pair also contains one dependent-launch boundary; measure it the same way around an empty one-wavefront kernel
        # (its own 0.6 us of execution is the only part of the subtraction that is not boundary)
        ov = C.c_float(0.0)
        _lib.check(_lib.load().cm_k_event_pair_overhead(ctx.handle, 200, 1, 64, C.byref(ov)))
        boundary_ms = float(ov.value)
        mid_ms = max(mid_raw_ms - boundary_ms, 1e-6)                             # average kernel duration, ms
        mid_bytes = K * (8 + 8 + 1 + 1 + 8) + N * (8 + 8 + 32) + nnz * 9 + N * (8 + 4 * nvals + 48 + 48 + 4 + 8 + 2 * S * 8 + 8 * (nvals - 1))
        achieved = mid_bytes / (mid_ms * 1e-3) / 1e9
        # HBM traffic of the same kernel from rocprofv3 PMC passes (profiles/r01_h_cfg3_pmc_summary.csv: FETCH_SIZE 1227.6 KB, doubled as the
        # MI355X guide prescribes for gfx950, + WRITE_SIZE 102.1 KB per launch); measured for cfg3 only
        traffic = {"cfg3": int((2 * 1227.6 + 102.1) * 1024)}.get(a.workload)
        # --- the Gauss-Seidel chain kernel (k_sweep_par), second opinion: its launch time and the records it actually streamed
        seq_ms = float(np.mean([t["sweep_seq_ms"] for t in tms_k])) / its
        recs = float(np.mean([t["sum_seq_records"] for t in tms])) / its         # 32-B records per launch (average over the fit)
        rows = float(np.mean([t["sum_seq_rows"] for t in tms])) / its
        chain_bytes = recs * 32 + (recs - rows) * 8 + K * 12                      # records + lam writes + R and ticket vectors
        dense_equiv_bytes = 12 * N * K                                             # I + lam read + lam write, fp32 dense
        phase = {k: float(np.mean([t[k] for t in tms])) for k in ("total_ms", "iters_ms", "fn_scan_ms")}
        phase.update({k: float(np.mean([t[k] for t in tms_k])) for k in ("solve_ms", "sweep_indep_ms", "sweep_seq_ms", "mid_ms", "post_ms")})
        phase["steps_with_kernel_events"] = int(sum(1 for st_ in stamped if st_ or a.phases))
        out = {
            "metric": "CAVIaR iterations/sec (N=%d, K=%d, T=%d)" % (N, K, T),
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if gen_ms is None else "synthetic (generated on the GPU in %.2f ms)" % gen_ms,
            "config": {"workload": "%s: N=%d neurons, K=%d trials, T=%d, H=%d targets/trial, %d iterations + fn_scan per step, "
                                   "1 circuit per GPU and step%s" % (a.workload, N, K, T, H, its,
                                                                      "" if world == 1 else ", %d circuits rotated over the ranks" % world),
                       "nnz_targeted": int(nnz), "distinct_powers": int(nvals) - 1,
                       "mean_active_set": float(np.mean([t["sum_active"] for t in tms])) / max(its, 1)},
            "roofline": {"bound": "hbm", "kernel": "k_iter_mid", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": mid_ms, "event_pair_ms": mid_raw_ms, "event_pair_boundary_ms": boundary_ms,
                         "algorithmic_bytes_per_launch": int(mid_bytes),
                         "timing_note": "avg_launch_ms = event_pair_ms - event_pair_boundary_ms (boundary measured around an empty kernel, whose own "
                                        "~2 us of dispatch a profiler attributes to every kernel): rocprof's average for this kernel lies between "
                                        "avg_launch_ms and event_pair_ms (profiles/README.md)",
                         "note": "latency-bound: 10 dependent Newton steps of fp64 exp/log per neuron; every kernel of this "
                                 "workload is latency- or dependency-bound, none is bandwidth-bound (DESIGN.md section 5)",
                         "chain_kernel": {"kernel": "k_sweep_par", "avg_launch_ms": seq_ms, "algorithmic_bytes_per_launch": int(chain_bytes),
                                          "achieved": chain_bytes / (seq_ms * 1e-3) / 1e9, "chain_rows_per_launch": rows,
                                          "dense_equiv_bytes_per_launch": int(dense_equiv_bytes),
                                          "dense_equiv_achieved": dense_equiv_bytes / (seq_ms * 1e-3) / 1e9}},
            "phase_ms_per_step": phase,
        }
        if world == 1 and not a.no_cpu_baseline:
            if sims:
                sim = sims[0]
            else:                                              # the oracle needs host copies of what the GPU generated
                sim = {"psc": ctx.download(dsims[0]["psc_ptr"], np.empty((K, T))),
                       "stim_matrix": ctx.download(dsims[0]["stim_ptr"], np.empty((N, K)))}
            out["cpu_baseline"] = cpu_baseline(sim, pri, wl)
        emit(json.dumps(out))
    for q in probs:
        q.close()
    for d in dsims:
        ctx.free(d["psc_ptr"]); ctx.free(d["stim_ptr"])
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
