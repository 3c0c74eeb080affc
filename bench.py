#!/usr/bin/env python3
"""CAVIaR iterations/s on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg5|demo|nwd] [--circuits-per-gpu B] [--no-cpu-baseline]

`--workload nwd` benchmarks the upstream NWD demixer forward pass instead (its own metric, traces/s; see bench_nwd).

A "step" is one pass of the hot path over one batch of synthetic input: B independent circuits of the BASELINE shape (N=1000
neurons, K=2000 trials, T=900 samples; default B = 512) fitted per GPU as ONE lock-step batch -- `iters` (50) CAVIaR iterations each plus
the false-negative scan and final phi update (reference circuitmap/optimise/caviar.py:79-98) -- through the C ABI's
cm_caviar_run_batch: the circuit is the outer grid dimension of every kernel launch.  (The library's own default for a batch, cm_options.interleave = 0,
runs it as two halves in flight on two streams, so that the serial phases of one overlap the wide phases of the other: ~5 % more
throughput, reported by the default run as the extra leg `interleaved` and timed instead of the headline with `--interleave`; the
headline keeps one lock-step batch per step so that every per-kernel figure -- HIP events, rocprof, PMC -- describes a kernel that had
the GPU to itself.)  A single fit is a latency chain (N sequential
Gauss-Seidel steps, 31 dependent Cholesky panels, 10 dependent Newton steps) that occupies a few percent of an MI355X; the batch is
what fills it, and it is the reference's own scale-out unit (independent circuits, one process each:
scripts/generate_loho_cv_slurm_scripts.py:8-36).  Every circuit of a batch is bit-identical to its own single fit
(tests/test_gpu_parity.py).  `--circuits-per-gpu 1` measures the latency form (one circuit at a time); on one GPU the default run reports it
beside the batch (`single_circuit`).  Synthetic data: seeded restatement of the reference's `simulate`, generated on the GPU by
default (include/sim_hip.h) or on the host with --host-sim.
Inputs (PSC integrals + encoded stimulus) are resident in HBM before the timed region; results stay in HBM.
value = (circuits x iterations) / wall time, whole job, max over ranks.  With N > 1 GPUs there are N sets of B circuits (seeds 2000 ...),
rotated over the ranks step by step so that every GPU sees the same mix: embarrassingly parallel, no data-path collective ("weak"
scaling).  `python bench.py --gpus N` started as one process launches its own N ranks (torch.distributed.run) before touching a GPU.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel by measured total, live HIP-event timing from the library) and
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
    # the one workload the reference publishes a CAVIaR rate for: examples/demo_compressive_connectivity_mapping.ipynb:172-175 (N = 200 neurons,
    # 2 min at 30 Hz = K ~ 3600 trials, H = 10 targets, connection_prob 0.1), msrmp 0.4 (:512), 50 iterations in 25 s = 1.97 it/s (:538) on an
    # unnamed Colab GPU -- ~180 trials per cell, i.e. every chain row spans three 64-entry record segments
    "demo": dict(N=200, K=3600, T=900, H=10, sim=dict(connection_prob=0.1), fit=dict(msrmp=0.4)),
}
PUBLISHED = {"demo": dict(value=1.97, unit="iterations/s", source="reference examples/demo_compressive_connectivity_mapping.ipynb:538 (50 it in 25 s)",
                          hardware="Google Colab, accelerator GPU (model unnamed), JAX <= 0.3.15",
                          note="one circuit; context, not a same-node comparison (different, unnamed hardware; the notebook's data were unseeded)")}
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


def cpu_baseline(sim, pri, wl, fit=None):
    """The CPU path timed on this host (the reference's own jitted JAX-CPU path cannot run here: jax is absent from the reference tree and the image).
    `value`: the COMPILED port (oracle/caviar_port.cpp, C++17 / OpenMP, running-residual formulation -- the formulation a tuned CPU
    implementation would use; held to the reference-source fixtures by tests/test_caviar_port.py) on ONE circuit of the workload, all iterations +
    fn_scan, on all host threads; the same on one thread (`one_thread`: what one core of a one-circuit-per-core sweep would do) and the NumPy
    restatements of earlier rounds (`numpy_literal`: operation-for-operation O(N^2 K) form, 1 iteration; `numpy_fast`) for continuity.
    One circuit: never to be divided into the batch aggregate `value` of the GPU line -- `single_circuit` is the like-for-like figure."""
    from oracle import caviar_oracle as co
    from oracle import caviar_port as cport
    args = (sim["psc"], sim["stim_matrix"], pri["mu"], pri["beta"], pri["shape"], pri["rate"], pri["phi"], pri["phi_cov"])
    fit = dict(fit or {})
    # (threads: the box's CPU share for one GPU is 16 cores whatever os.cpu_count() says -- 256 OpenMP threads on it took 58 s per fit, one thread 0.9 s)
    ncpu = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    out = {}
    iters = 50 if wl["N"] <= 2000 else 3
    cport.caviar(*args, iters=1, seed=0, n_threads=ncpu, **fit)                 # warm-up (thread pool, page faults)
    reps, t = 0, time.time()
    while reps < 3 or (time.time() - t < 3.0 and reps < 10):
        cport.caviar(*args, iters=iters, seed=0, n_threads=ncpu, **fit)
        reps += 1
    dt = (time.time() - t) / reps
    out.update(value=iters / dt, unit="iterations/s", cores=int(ncpu), kind="port",
               implementation="compiled C++17 / OpenMP restatement (oracle/caviar_port.cpp, g++ -O3 -march=x86-64-v3 -fopenmp), running-residual formulation",
               sample="one circuit of the workload, %d iterations + fn_scan, %d host threads, mean of %d fits of %.2f s" % (iters, ncpu, reps, dt))
    t = time.time()
    cport.caviar(*args, iters=iters, seed=0, n_threads=1, **fit)
    dt1 = time.time() - t
    out["one_thread"] = dict(value=iters / dt1, unit="iterations/s", cores=1, sample="the same fit on one thread, %.2f s" % dt1)
    if wl["N"] <= 1000:
        t = time.time()
        co.caviar(*args, iters=1, seed=0, literal=True, fn_scan=False, **fit)
        dtl = time.time() - t
        out["numpy_literal"] = dict(value=1.0 / dtl, unit="iterations/s", sample="literal O(N^2 K) NumPy oracle, 1 iteration (it=0) of 50, %.1f s" % dtl)
    n_fast = 3 if wl["N"] <= 1000 else 1
    t = time.time()
    co.caviar(*args, iters=n_fast, seed=0, literal=False, fn_scan=False, **fit)
    dtf = time.time() - t
    out["numpy_fast"] = dict(value=n_fast / dtf, unit="iterations/s", sample="running-residual NumPy oracle, first %d iterations, %.1f s" % (n_fast, dtf))
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
    from circuitmap_amd.neural_waveform_demixing import NeuralDemixer, random_parameters
    K, T = 2000, 900
    state = random_parameters(seed=0)
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
            from oracle import nwd_oracle as no      # the CPU baseline leg only (checker / baseline)
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


# algorithmic bytes / flops per CAVIaR iteration, dense formulation
def dense_equiv(N, K):
    return 12.0 * N * K + 8.0 * N * N, 2.0 * N * N * K + (2.0 / 3.0) * N ** 3


def launcher_argv(gpus, port, argv):
    """`python bench.py --gpus N` started as ONE process: the command that starts the N ranks (one per GPU, RCCL over xGMI) --
    the same line the driver uses."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(a, argv):
    """Never re-exec a process that has touched the GPU: this runs before torch / HIP are imported, starts the ranks as CHILD
    processes, relays their one JSON line and exits with their status."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(launcher_argv(a.gpus, port, argv), env=env, stdout=subprocess.PIPE)
    sys.stdout.buffer.write(proc.stdout)
    sys.stdout.flush()
    return proc.returncode


def dry_run_cpu(a):
    """Rehearsal of the multi-rank skeleton without a GPU (gloo): rendezvous, barriers, max-over-ranks of the elapsed time, one
    JSON line from rank 0.  No CAVIaR work is done and the line says so (`dry_run`); used by tests/test_bench_launcher.py."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    for step in range(a.steps):
        circuit_for(rank, step, world)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0 + 1e-3 * (rank + 1)
    seen = world
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        seen = dist.get_world_size()
    if rank == 0:
        emit(json.dumps({"metric": "dry run (no GPU work)", "value": 0.0, "unit": "iterations/s", "n_gpus": a.gpus, "n_ranks_seen": seen,
                         "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / max(a.steps, 1) * 1e3, "dry_run": True}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def measure_batch(ctx, local_rank, wl, pri, iters, B, seed0, steps, warmup, sync):
    """Lock-step batch of B circuits on this GPU (cm_caviar_run_batch): B fresh circuits generated on the device, `steps` timed batches."""
    from circuitmap_amd.optimise.caviar import Problem, run_batch
    from circuitmap_amd.simulation import simulate_device
    N, K, T, H = wl["N"], wl["K"], wl["T"], wl["H"]
    dsims, probs = [], []
    for c in range(B):
        d = simulate_device(N=N, T=T, H=H, trials=K, seed=seed0 + c, device_id=local_rank, keep_on_device=True, ctx=ctx, **wl.get("sim", {}))
        dsims.append(d)
        probs.append(Problem(ctx, None, None, device_ptrs=(N, K, T, d["psc_ptr"], d["stim_ptr"])))
        ctx.free(d["psc_ptr"]); ctx.free(d["stim_ptr"])             # the problem keeps its own encoding
    opts = dict(iters=iters, seed=0, **wl.get("fit", {}))
    for _ in range(warmup):
        run_batch(probs, pri, **opts)
    sync()
    t0 = time.perf_counter()
    tms = []
    for s_ in range(steps):
        tms.append(run_batch(probs, pri, timing=1 if s_ == steps - 1 else 0, **opts)[0].as_dict())
    sync()
    dt = time.perf_counter() - t0
    for q in probs:
        q.close()
    return dt, tms


def end_to_end(_lib, ctx, local_rank, wl, pri, a, B):
    """The whole call, not only the fit (the reference's `model.time` wraps everything, circuitmap/model.py:105-148): the same B circuits
    (a) from inputs already in HBM -- cm_problem_create(CM_MEM_DEVICE) x B, one lock-step cm_caviar_run_batch, cm_caviar_fetch of
    mu / beta / phi / phi_cov / z, cm_problem_destroy -- and (b) from host arrays through the C entry point cm_caviar_fit_batch (upload over
    PCIe included; the dense N x K lam is not requested).  Never `value`: the headline keeps inputs resident."""
    import numpy as np
    from circuitmap_amd.optimise.caviar import Problem, run_batch, fetch_batch, make_options, _priors_struct
    from circuitmap_amd.simulation import simulate_device
    N, K, T, H = wl["N"], wl["K"], wl["T"], wl["H"]
    opts = dict(iters=a.iters, seed=0, **wl.get("fit", {}))
    devs = [simulate_device(N=N, T=T, H=H, trials=K, seed=a.circuit_seed + j, device_id=local_rank, keep_on_device=True, ctx=ctx, **wl.get("sim", {})) for j in range(B)]
    res = {}

    def dev_pass():
        t = [time.perf_counter()]
        probs = Problem.create_batch(ctx, [("device", N, K, T, d["psc_ptr"], d["stim_ptr"]) for d in devs])
        ctx.synchronize(); t.append(time.perf_counter())
        run_batch(probs, pri, **opts)
        ctx.synchronize(); t.append(time.perf_counter())
        outs = fetch_batch(probs)
        t.append(time.perf_counter())
        for q in probs:
            q.close()
        t.append(time.perf_counter())
        return [(t[i + 1] - t[i]) * 1e3 for i in range(4)], outs[0][0]
    dev_pass()
    best = None
    for _ in range(3):
        ms, mu0 = dev_pass()
        if best is None or sum(ms) < sum(best):
            best = ms
    tot = sum(best)
    res["device_resident"] = {"ms": tot, "value": B * a.iters / tot * 1e3, "unit": "iterations/s", "create_ms": best[0], "create_ms_per_circuit": best[0] / B,
                              "fit_ms": best[1], "fetch_ms": best[2], "destroy_ms": best[3],
                              "note": "inputs in HBM: cm_problem_create_batch(CM_MEM_DEVICE) (the prologue kernels once per group of 64, memory from the context's pool), "
                                      "cm_caviar_run_batch with its default scheduling, cm_caviar_fetch_batch (mu / beta / phi / phi_cov / z / shape / rate of all "
                                      "circuits with one device-to-host copy), destroy (memory back to the pool); best of 3"}
    # (b) host arrays through the C entry point
    hosts = [(ctx.download(d["psc_ptr"], np.empty((K, T))), ctx.download(d["stim_ptr"], np.empty((N, K)))) for d in devs]
    for d in devs:
        ctx.free(d["psc_ptr"]); ctx.free(d["stim_ptr"])
    lib = ctx.lib
    o = make_options(lib, **opts)
    keep = []
    prs = (_lib.CmPriors * B)(*[_priors_struct(N, pri["mu"], pri["beta"], pri["shape"], pri["rate"], pri["phi"], pri["phi_cov"], keep) for _ in range(B)])
    resv = (_lib.CmResult * B)()
    bufs = []
    for c in range(B):
        ob = dict(mu=np.empty(N), beta=np.empty(N), phi=np.empty((N, 2)), phi_cov=np.empty((N, 2, 2)), z=np.empty(K))
        for k_, v in ob.items():
            setattr(resv[c], k_, _lib.dptr(v))
        bufs.append(ob)
    i32 = lambda v: (C.c_int32 * B)(*v)
    dpp = lambda arrs: (C.POINTER(C.c_double) * B)(*[_lib.dptr(x) for x in arrs])
    args = (B, i32([N] * B), i32([K] * B), i32([T] * B), dpp([h[0] for h in hosts]), dpp([h[1] for h in hosts]), prs, C.byref(o),
            (C.c_int32 * 1)(local_rank), 1, resv, None)
    _lib.check(lib.cm_caviar_fit_batch(*args))
    hbest = None
    for _ in range(2):
        t0 = time.perf_counter()
        _lib.check(lib.cm_caviar_fit_batch(*args))
        dt = (time.perf_counter() - t0) * 1e3
        hbest = dt if hbest is None else min(hbest, dt)
    res["host_arrays"] = {"ms": hbest, "value": B * a.iters / hbest * 1e3, "unit": "iterations/s", "upload_bytes": int(B * (K * T + N * K) * 8),
                          "same_mu_as_device_resident": bool(np.array_equal(bufs[0]["mu"], mu0)),
                          "note": "cm_caviar_fit_batch: upload of psc + stim (pageable host memory staged through pinned slots by four copy threads, PCIe-bound), "
                                  "create, fit, fetch (no dense lam), destroy -- as two halves, the second uploaded under the first's fit; the context and its memory pool are kept between calls; best of 2"}
    # (c) the same circuits as COMPACT host inputs (cm_caviar_fit_batch_ex): psc as float32, the stimulus matrix as its non-zero (trial, neuron, power)
    # entries -- built here, outside the clock, from float32-rounded traces (a caller that records in float32 holds them like this; the dense float64
    # call on the same rounded values is fitted once for the bitwise comparison)
    keep2 = []
    hosts32 = [(np.asarray(h[0], dtype=np.float32), h[1]) for h in hosts]
    ins = (_lib.CmInputs * B)(*[_lib.compact_inputs(p32, st, keep2) for p32, st in hosts32])
    mu_c = [np.empty(N) for _ in range(B)]
    for c in range(B):
        resv[c].mu = _lib.dptr(mu_c[c])
    args_c = (B, i32([N] * B), i32([K] * B), i32([T] * B), ins, prs, C.byref(o), (C.c_int32 * 1)(local_rank), 1, resv, None)
    _lib.check(lib.cm_caviar_fit_batch_ex(*args_c))
    cbest = None
    for _ in range(2):
        t0 = time.perf_counter()
        _lib.check(lib.cm_caviar_fit_batch_ex(*args_c))
        dt = (time.perf_counter() - t0) * 1e3
        cbest = dt if cbest is None else min(cbest, dt)
    mu_compact0 = mu_c[0].copy()
    d64 = hosts32[0][0].astype(np.float64)
    one = (B, i32([N] * B), i32([K] * B), i32([T] * B), dpp([d64] * B), dpp([hosts[0][1]] * B), prs, C.byref(o), (C.c_int32 * 1)(local_rank), 1, resv, None)
    _lib.check(lib.cm_caviar_fit_batch(*((1,) + one[1:])))
    res["host_arrays_compact"] = {"ms": cbest, "value": B * a.iters / cbest * 1e3, "unit": "iterations/s",
                                  "upload_bytes": int(sum(p.nbytes for p in keep2)),
                                  "bitwise_equal_to_dense_float64_call": bool(np.array_equal(mu_compact0, mu_c[0])),
                                  "note": "cm_caviar_fit_batch_ex: psc as float32, stimulus matrix as (trial, neuron, power) triples of its non-zero entries, expanded on "
                                          "the device to the dense float64 arrays of the plain call; otherwise the same sequence; best of 2"}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS) + ["nwd"])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--circuits-per-gpu", type=int, default=None, help="B: every step fits B independent circuits per GPU as ONE lock-step batch "
                    "(cm_caviar_run_batch: the circuit is the outer grid dimension of every launch); value counts all of their iterations.  "
                    "Default 512 for cfg3 (a step costs about 5 ms + 0.137 ms per circuit: 283 / 322 / 339 / 350 k it/s at 128 / 256 / 512 / 1024), 256 for cfg2 and demo, 1 for cfg5 "
                    "(one N=5000 circuit already fills the GPU); 1 = one circuit at a time (the latency form)")
    ap.add_argument("--interleave", action="store_true", help="timed region with the library's default scheduling of a batch (cm_options.interleave = 0: two "
                    "half-batches at a time on two streams, ~5 %% more throughput) instead of one lock-step batch per step; per-kernel durations -- and "
                    "with them `roofline` -- are then those of kernels that share the GPU with the other half's")
    ap.add_argument("--batch-leg", type=int, default=16, help="with --circuits-per-gpu 1 on one GPU: also time a lock-step batch of this many "
                    "circuits after the main measurement and report it under `batch_throughput` (0 = skip)")
    ap.add_argument("--batch-streams", type=int, default=1, help="G (default 1): the B circuits of a step run as G lock-step batches of B / G circuits, each on its "
                    "own context (stream + mailboxes) driven by its own host thread, so that the latency-bound phases of one batch overlap the "
                    "wide phases of another")
    ap.add_argument("--streams-leg", type=int, default=None, help="0: skip the `interleaved` leg (on one GPU with one lock-step batch per step the same B "
                    "circuits are otherwise also timed with the library's default scheduling, two half-batches at a time on two streams)")
    ap.add_argument("--no-single-leg", action="store_true", help="with B > 1 on one GPU: skip the one-circuit-at-a-time measurement that is "
                    "otherwise reported beside the batch under `single_circuit`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the `end_to_end` leg (problem creation + fit + fetch of the same B circuits, from "
                    "device-resident inputs and from host arrays through cm_caviar_fit_batch)")
    ap.add_argument("--phases", action="store_true", help="stamp every phase of every iteration (more host API calls)")
    ap.add_argument("--circuit-seed", type=int, default=2000, help="seed of rank 0's synthetic circuit (rank r uses seed + r)")
    ap.add_argument("--host-sim", action="store_true", help="generate the synthetic experiments with the NumPy restatement on the host "
                    "(default: on the GPU, include/sim_hip.h -- psc and stim_matrix never exist on the host)")
    ap.add_argument("--dry-run-cpu", action="store_true", help="rehearse the multi-rank skeleton on the CPU (gloo), no GPU work")
    a = ap.parse_args()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a, sys.argv[1:]))            # not under torch.distributed.run: start the ranks ourselves (before any GPU call)
    quiet_stdout()
    if a.dry_run_cpu:
        return dry_run_cpu(a)
    if a.workload == "nwd":
        return bench_nwd(a)

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (the HIP path has no CPU fallback)")
    if os.environ.get("CAVIAR_BENCH_SHARE_GPU") == "1":    # rehearsal of the multi-rank path on a one-GPU box: every rank uses device 0
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist, control_plane = None, None
    if world > 1 or ("RANK" in os.environ and os.environ.get("CAVIAR_BENCH_FORCE_DIST") == "1"):
        # one process per GPU under torch.distributed.run; backend "nccl" is RCCL on ROCm.  The data path has no collective
        # (independent circuits): the process group only carries the barriers and the max-over-ranks of the elapsed time.
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        control_plane = os.environ.get("CAVIAR_BENCH_BACKEND", "nccl")
        if control_plane == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:                                              # CAVIAR_BENCH_BACKEND=gloo: the same two calls over gloo (rehearsals on a one-GPU box)
            control_plane = "gloo"
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    from circuitmap_amd import _lib
    from circuitmap_amd.optimise.caviar import Problem, run_batch
    from circuitmap_amd.simulation import simulate

    wl = WORKLOADS[a.workload]
    N, K, T, H = wl["N"], wl["K"], wl["T"], wl["H"]
    if a.circuits_per_gpu is None:
        a.circuits_per_gpu = 1 if a.workload == "cfg5" else (512 if a.workload == "cfg3" else 256)
    B = max(1, a.circuits_per_gpu)
    if a.streams_leg is None:
        a.streams_leg = 2 if (B >= 32 and a.batch_streams == 1 and not a.interleave) else 0
    pri = default_priors(N)
    # `world` sets of B independent circuits (seeds circuit_seed ...); every rank holds all sets in HBM and fits set (rank + step) mod
    # world at step `step`: at any moment the GPUs work on different circuits, and over the run every GPU sees the same mix, so the
    # max-over-ranks time measures the machine, not which rank drew the circuits with the slowest active-set trajectories (fit time
    # varies +-4 % between these seeds).  One GPU: one set.
    gen_ms, sims, sets = None, [], []
    G = max(1, min(a.batch_streams, B))
    ctxs = [_lib.Context(local_rank) for _ in range(G)]
    ctx = ctxs[0]
    for c in range(world):
        probs = []
        for j in range(B):
            ctx = ctxs[j % G]                              # circuit j of a set lives on context j mod G
            seed = a.circuit_seed + c * B + j
            if a.host_sim:
                sim = simulate(N=N, T=T, H=H, trials=K, seed=seed, lite=(N > 2000), **wl.get("sim", {}))
                if not sims:
                    sims.append(sim)
                probs.append(Problem(ctx, sim["psc"], sim["stim_matrix"]))        # H2D + one-off prologue: outside the timed region
            else:
                from circuitmap_amd.simulation import simulate_device
                d = simulate_device(N=N, T=T, H=H, trials=K, seed=seed, device_id=local_rank, keep_on_device=True, ctx=ctx, **wl.get("sim", {}))
                gen_ms = d["device_ms"]
                probs.append(Problem(ctx, None, None, device_ptrs=(N, K, T, d["psc_ptr"], d["stim_ptr"])))   # one-off prologue: outside the timed region
                if c == 0 and j == 0 and world == 1 and not a.no_cpu_baseline:
                    sims.append({"psc": ctx.download(d["psc_ptr"], np.empty((K, T))), "stim_matrix": ctx.download(d["stim_ptr"], np.empty((N, K)))})
                ctx.free(d["psc_ptr"]); ctx.free(d["stim_ptr"])
        sets.append(probs)
    ctx = ctxs[0]
    pool = None
    if G > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=G)
    nnz, nvals = sets[0][0].nnz
    run_args = (pri["mu"], pri["beta"], pri["shape"], pri["rate"], pri["phi"], pri["phi_cov"])
    opts = dict(iters=a.iters, seed=0, **wl.get("fit", {}))
    # Per-kernel HIP events cost a launch boundary each (~8 % of a fit with four per iteration), so the library records them only on
    # request.  The timed region asks for them on every 4th step: the kernel durations in `roofline` are
    # measured live inside the timed region, on a sample of its steps; `value` pays for that sample (about 2 %).
    stamped = [(s % 4 == 3) or (a.steps < 4 and s == a.steps - 1) for s in range(a.steps)]
    timing_level = 2 if a.phases else 1

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        for cx in ctxs:
            cx.synchronize()

    def step_fn(step, timing):
        probs = sets[circuit_for(rank, step, world)]
        if B == 1:
            return probs[0].run(*run_args, timing=timing, **opts).as_dict()
        if G == 1:
            return run_batch(probs, pri, timing=timing, interleave=a.interleave, **opts)[0].as_dict()
        futs = [pool.submit(run_batch, probs[g::G], pri, timing=timing, interleave=a.interleave, **opts) for g in range(G)]
        res = [f.result()[0].as_dict() for f in futs]
        return res[0]                                      # per-kernel times and counters: those of the first group (B / G circuits per launch)

    for w in range(max(a.warmup, world)):                  # every set of the rotation has run once before the clock starts
        step_fn(w, 0)
    import gc
    gc.collect()
    gc.disable()                 # a generation-2 collection of this process (torch is loaded) is a 50 ms pause: keep it out of the timed steps
    barrier()
    t0 = time.perf_counter()
    tms, step_wall = [], []
    for step in range(a.steps):
        ts_ = time.perf_counter()
        tms.append(step_fn(step, timing_level if (stamped[step] or a.phases) else 0))
        step_wall.append((time.perf_counter() - ts_) * 1e3)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if os.environ.get("CAVIAR_BENCH_TRACE_STEPS"):
        print("step wall ms:", " ".join("%.2f" % v for v in step_wall), file=sys.stderr)
    n_ranks_seen = 1
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if control_plane == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        n_ranks_seen = dist.get_world_size()

    exit_code = 0
    if rank == 0:
        its = a.iters
        value = world * B * a.steps * its / dt
        S = 100
        tms_k = [t for t, st_ in zip(tms, stamped) if st_ or a.phases]           # the steps that carried per-kernel events
        mean_k = lambda key: float(np.mean([t[key] for t in tms_k]))
        B_iter, F_iter = dense_equiv(N, K)
        # ---- candidates for "the dominant kernel", each timed live by HIP events the library records on its stream
        # (1) k_iter_mid, one launch per iteration (event pair around every launch; the pair includes one dependent-launch boundary,
        #     ~2-3 us, which a profiler attributes elsewhere -- avg_launch_ms is therefore an upper bound of the rocprof average).
        #     Algorithmic bytes of one launch and circuit:
        #       block 0   y, R (8 B each), colany, mask (1 B each) per trial; mu, beta (8 B), row stats (32 B) per neuron; z written
        #       waves     per neuron: its lam row (8 + 1 B per targeted trial), counts of V stimulus values (4 B each), phi priors
        #                 (48 B), phi / phi_cov written (48 B); next MC logit: inverse order (4 B), key (8 B), P logits written (8 B each)
        #     The ndtri(u) table row the MC logit reads (2 S x 8 B per neuron) is listed SEPARATELY (`table_bytes_per_launch`): the
        #     table is an artefact of this implementation (it replaces S quantile evaluations per neuron), not compulsory traffic.
        mid_ms = mean_k("mid_ms") / its
        Bl = int(tms_k[0].get("batch", B // G if G > 1 else B)) if tms_k else B      # circuits per launch (the library runs a batch as two interleaved halves by default)
        mid_bytes = Bl * (K * (8 + 8 + 1 + 1 + 8) + N * (8 + 8 + 32) + nnz * 9 + N * (8 + 4 * nvals + 48 + 48 + 4 + 8 + 8 * (nvals - 1)))
        mid_table = Bl * N * 2 * S * 16
        # (2) the Gauss-Seidel chain kernel (k_sweep_par), one launch per iteration: the records it actually streamed
        seq_ms = mean_k("sweep_seq_ms") / its
        recs = float(np.mean([t["sum_seq_records"] for t in tms])) / its         # 32-B records per launch and circuit (average over the fit)
        rows = float(np.mean([t["sum_seq_rows"] for t in tms])) / its
        chain_bytes = Bl * (recs * 32 + (recs - rows) * 8 + K * 12)               # records + lam writes + R and ticket vectors
        # (3) the passes of the tiled Cholesky over the trailing matrix: k_chol_panel (one panel per launch; one circuit up to 48 tile rows), or
        #     panel-pair steps thin k_chol_bulk (incl. the panel solve of its column) + k_chol_trsm + rank-64 k_chol_bulk (batches, larger matrices), timed as a
        #     group by one event pair per factorisation.  Algorithmic bytes of a pass: every trailing 32 x 32 fp64 tile
        #     read and rewritten once (16 KB), plus the 8 KB tiles no blocking of this width avoids, each counted once per launch however many
        #     workgroups share it -- panel tiles read, solved panel tiles written, the diagonal tile's inverse
        fl = float(np.mean([t["fact_launches"] for t in tms_k]))
        ft = float(np.mean([t["fact_tile_updates"] for t in tms_k]))
        fp = float(np.mean([t["fact_panel_tiles"] for t in tms_k]))
        fact_total = mean_k("fact_ms")
        mid_name = "k_iter_mid"
        # batches: the all-zero rows (95 % in the steady state) run in their own lean kernel k_mc_dead, timed separately; their share of the bytes:
        # phi / phi_cov constants in, phi / phi_cov out, inverse order, logits out (the shared table is listed separately, as above)
        dead_ms = mean_k("mc_dead_ms") / its
        dead_frac = 1.0 - float(np.mean([t["sum_active"] for t in tms])) / max(its, 1) / N
        dead_bytes = Bl * dead_frac * N * (8 + 64 + 48 + 4 + 8 * (nvals - 1))
        if Bl > 1:
            mid_bytes -= dead_bytes
        cands = {
            mid_name: dict(total_ms=mid_ms * its, launches=its, bytes=mid_bytes, note="latency-bound: 10 dependent Newton steps of fp64 exp/log per neuron, "
                               "then the next iteration's Monte-Carlo logit; block 0: sigma / z / fused small solve"),
            "k_sweep_par": dict(total_ms=seq_ms * its, launches=its, bytes=chain_bytes, note="dependency-bound: Gauss-Seidel chain, one workgroup per circuit"),
        }
        if Bl > 1:
            cands["k_mc_dead"] = dict(total_ms=dead_ms * its, launches=its, bytes=dead_bytes, note="fp64 VALU issue: the next Monte-Carlo logit of every all-zero row "
                                      "(S x P sigmoid-logit evaluations per row), one wavefront per row")
        if fl > 0:
            cands["k_chol_wide+k_chol_gpanel" if B > 1 else ("k_chol_bulk+k_chol_trsm" if N > 48 * 32 else "k_chol_panel")] = dict(
                total_ms=fact_total, launches=fl, bytes=(ft * 16384.0 + fp * 8192.0) / fl,
                note="passes of the blocked fp64 Cholesky over the trailing matrix (batches: four panels per pass, k_chol_gpanel + k_chol_wide on 64 x 64 "
                     "super-tiles; one circuit: one panel per k_chol_panel launch up to 48 tile rows, panel pairs k_chol_bulk + k_chol_trsm above): bound by "
                     "the CU's fetch path and the matrix cores together while the trailing matrix is large, launch-chain-bound in the last panels and for one circuit")
        # (4) the tiled block_update_mu as a whole -- precision-matrix build, Cholesky passes, recursive-doubling inverse, mu / beta reductions --
        #     timed by one event pair per solve.  Its work is GEMM-shaped on the fp64 matrix cores: the roof is the dense fp64 MFMA peak, achieved =
        #     tile-product flops (factor + inverse, 65 536 per 32 x 32 x 32 product) / time.
        tiled_total = mean_k("tiled_ms")
        n_tiled = float(np.mean([t["tiled_solves"] for t in tms_k]))
        tile_flops = (float(np.mean([t["fact_products"] for t in tms_k])) + float(np.mean([t["inv_products"] for t in tms_k]))) * 65536.0
        if n_tiled > 0 and tiled_total > 0:
            cands["tiled solve (k_build_precision, k_chol_*, k_inv_*, column sums)"] = dict(
                total_ms=tiled_total, launches=n_tiled, bound="mfma", flops=tile_flops / n_tiled,
                # its algorithmic bytes: the compulsory ones -- every active-set lower triangle read once, factor and inverse written once
                bytes=float(np.mean([t["fact_tri_tiles"] for t in tms_k])) * 3 * 8192 / n_tiled,
                note="block_update_mu of the iterations whose active sets exceed the in-LDS solver (7 of 50 at this shape): sparse precision build, "
                     "grouped fp64-MFMA Cholesky (rank-128 passes on 64 x 64 super-tiles), recursive-doubling inverse, column reductions; "
                     "`roofline.solve` splits out the factor passes and sets them against the compulsory bytes")
        dom = max(cands, key=lambda k_: cands[k_]["total_ms"])
        for k_, c_ in cands.items():
            c_["avg_launch_ms"] = c_["total_ms"] / max(c_["launches"], 1)
            if c_.get("bound") == "mfma":
                c_["achieved"] = c_["flops"] / max(c_["avg_launch_ms"] * 1e-3, 1e-12) / 1e12        # TFLOP/s
            else:
                c_["achieved"] = c_["bytes"] / max(c_["avg_launch_ms"] * 1e-3, 1e-12) / 1e9          # GB/s
        d = cands[dom]
        # the fp64 matrix-core rate this device actually sustains (operands in registers, no memory traffic: cm_k_mfma_peak), next to the guide's
        # nominal dense peak that `peak` / `frac` are quoted against
        mfma_meas = None
        try:
            tf = C.c_double()
            _lib.check(ctx.lib.cm_k_mfma_peak(ctx.handle, 30000, C.byref(tf)))
            mfma_meas = float(tf.value)
        except Exception as e:                                   # (a measurement aid: never fails the line)
            print("bench.py: cm_k_mfma_peak failed: %s" % e, file=sys.stderr)
        # HBM traffic of that kernel from rocprofv3 PMC passes (profiles/, gfx950 correction FETCH_SIZE x 2 + WRITE_SIZE per the MI355X guide);
        # only quoted when a summary for THIS build, workload and batch size is committed -- otherwise null (see profiles/README.md)
        traffic, traffic_src = pmc_traffic(a.workload, Bl, dom)
        phase = {k: float(np.mean([t[k] for t in tms])) for k in ("total_ms", "iters_ms", "fn_scan_ms")}
        phase.update({k: mean_k(k) for k in ("solve_ms", "sweep_indep_ms", "sweep_seq_ms", "mid_ms", "mc_dead_ms", "post_ms", "fact_ms", "tiled_ms")})
        phase["steps_with_kernel_events"] = int(sum(1 for st_ in stamped if st_ or a.phases))
        out = {
            "metric": "CAVIaR iterations/sec (N=%d, K=%d, T=%d)" % (N, K, T),
            "value": value, "unit": "iterations/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "control_plane": control_plane, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if gen_ms is None else "synthetic (generated on the GPU in %.2f ms per circuit)" % gen_ms,
            "config": {"workload": "%s: N=%d neurons, K=%d trials, T=%d, H=%d targets/trial, %d iterations + fn_scan per step, "
                                   "%s per GPU and step%s" % (a.workload, N, K, T, H, its,
                                                              "1 circuit" if B == 1 else ("%d circuits as one lock-step batch" % B) if (G == 1 and not a.interleave) else ("%d circuits as two interleaved half-batches (library default)" % B) if G == 1 else ("%d circuits as %d lock-step batches on %d streams" % (B, G, G)),
                                                              "" if world == 1 else ", %d circuit sets rotated over the ranks" % world),
                       "circuits_per_gpu": B, "nnz_targeted": int(nnz), "distinct_powers": int(nvals) - 1,
                       "mean_active_set": float(np.mean([t["sum_active"] for t in tms])) / max(its, 1)},
            "roofline": {"bound": d.get("bound", "hbm"), "kernel": dom, "achieved": d["achieved"],
                         "peak": 78.6 if d.get("bound") == "mfma" else HBM_PEAK_GBS, "unit": "TFLOP/s" if d.get("bound") == "mfma" else "GB/s",
                         "frac": d["achieved"] / (78.6 if d.get("bound") == "mfma" else HBM_PEAK_GBS), "traffic": traffic,
                         "peak_measured": mfma_meas if d.get("bound") == "mfma" else None,
                         "frac_of_measured": (d["achieved"] / mfma_meas) if (d.get("bound") == "mfma" and mfma_meas) else None,
                         "peak_note": "peak = the guide's dense fp64 matrix peak (78.6 TFLOP/s); peak_measured = what v_mfma_f64_16x16x4_f64 sustains on this device "
                                      "with operands in registers and no memory traffic (cm_k_mfma_peak, 4 chains x 8 wavefronts per SIMD; tools/micro/mfma_peak.hip)",
                         "avg_launch_ms": d["avg_launch_ms"], "launches_per_step": d["launches"], "total_ms_per_step": d["total_ms"],
                         "algorithmic_bytes_per_launch": int(d["bytes"]), "algorithmic_flops_per_launch": int(d.get("flops", 0)), "note": d["note"],
                         "timing_note": "HIP event pair around each launch on the library's stream; the pair contains one dependent-launch "
                                        "boundary, so avg_launch_ms is an upper bound of the rocprof kernel average (profiles/README.md)",
                         "selection": "largest per-step total among the event-timed kernels",
                         "other_kernels": {k_: {"avg_launch_ms": c_["avg_launch_ms"], "total_ms_per_step": c_["total_ms"], "bound": c_.get("bound", "hbm"),
                                                "algorithmic_bytes_per_launch": int(c_["bytes"]), "achieved": c_["achieved"],
                                                "unit": "TFLOP/s" if c_.get("bound") == "mfma" else "GB/s",
                                                "frac": c_["achieved"] / (78.6 if c_.get("bound") == "mfma" else HBM_PEAK_GBS)}
                                           for k_, c_ in cands.items() if k_ != dom},
                         "pmc_source": traffic_src,
                         "k_iter_mid_table_bytes_per_launch": int(mid_table),
                         "k_iter_mid_table_note": "(ndtri(u), u) pairs the Monte-Carlo logit reads, 2 S x 16 B per neuron and launch; one table serves all "
                                                  "circuits of a batch (the streams are data independent), so its HBM share is 1 / B of this figure",
                         # whole-iteration dense-equivalent RATE: bytes the reference's dense formulation would move per
                         # iteration x iterations/s.  The compressed encoding never moves those bytes -- this is not traffic and not `achieved`.
                         "dense_equiv_rate": {"B_iter": B_iter, "F_iter": F_iter, "hbm_frac": B_iter * value / world / (HBM_PEAK_GBS * 1e9),
                                              "fp64_mfma_frac": F_iter * value / world / 78.6e12}},
            "phase_ms_per_step": phase,
        }
        if fl > 0 and fact_total > 0:
            # the tiled solve measured against what the PROBLEM needs, not what the pass structure moves: compulsory bytes = every active-set
            # lower triangle read once, factor and inverse written once (3 x 8 KB per 32 x 32 tile); matrix-core share = tile-product flops of the
            # factor passes / their time / the dense fp64 MFMA peak (78.6 TFLOP/s)
            tri = float(np.mean([t["fact_tri_tiles"] for t in tms_k]))
            prod = float(np.mean([t["fact_products"] for t in tms_k]))
            pass_bytes = ft * 16384.0 + fp * 8192.0
            out["roofline"]["solve"] = {
                "fact_ms_per_step": fact_total, "passes_per_step": fl,
                "compulsory_bytes_per_step": int(tri * 3 * 8192), "pass_bytes_per_step": int(pass_bytes),
                "restream_factor": pass_bytes / max(tri * 3 * 8192, 1.0),
                "hbm_frac_on_compulsory_bytes": tri * 3 * 8192 / (fact_total * 1e-3) / (HBM_PEAK_GBS * 1e9),
                "hbm_frac_on_pass_bytes": pass_bytes / (fact_total * 1e-3) / (HBM_PEAK_GBS * 1e9),
                "tile_product_flops_per_step": prod * 65536.0, "mfma_frac": prod * 65536.0 / (fact_total * 1e-3) / 78.6e12,
                "note": "fact_ms = first to last launch of the Cholesky factorisations of a step (event pairs); the inverse of the factor and the "
                        "precision-matrix build are not inside it"}
        if world == 1 and B == 1 and a.batch_leg > 1 and a.workload != "cfg5":
            # many circuits per GPU in one launch set: the throughput form of the same path
            bdt, btms = measure_batch(ctx, local_rank, wl, pri, its, a.batch_leg, a.circuit_seed + 100, 4, 2, barrier)
            bval = a.batch_leg * 4 * its / bdt
            out["batch_throughput"] = {"circuits_per_gpu": a.batch_leg, "value": bval, "unit": "iterations/s", "steps": 4, "warmup": 2,
                                       "ms_per_step": bdt / 4 * 1e3, "x_single_circuit": bval / value,
                                       "dense_equiv_hbm_frac": B_iter * bval / (HBM_PEAK_GBS * 1e9),
                                       "k_iter_mid_avg_launch_ms": btms[-1]["mid_ms"] / its, "k_sweep_par_avg_launch_ms": btms[-1]["sweep_seq_ms"] / its,
                                       "fact_ms_per_step": btms[-1]["fact_ms"],
                                       "note": "every circuit bit-identical to its single fit (tests/test_gpu_parity.py::test_fit_batch_baseline_config3)"}
        if a.workload in PUBLISHED:
            out["published_reference"] = PUBLISHED[a.workload]
            if B == 1:
                out["published_reference"] = dict(PUBLISHED[a.workload], x_published=value / PUBLISHED[a.workload]["value"])
        batch_res0 = None
        if B > 1:
            # self-check of the launch configuration that was just timed: circuit 0's results as the lock-step batch left them ...
            batch_res0 = sets[0][0].fetch()
        if world == 1 and B > 1 and not a.no_single_leg:
            # the same path one circuit at a time: the latency of a fit (what round 1's bench line reported)
            q = sets[0][0]
            for _ in range(2):
                q.run(*run_args, **opts)
            barrier()
            t1 = time.perf_counter()
            stm = [q.run(*run_args, timing=1 if i_ % 4 == 3 else 0, **opts).as_dict() for i_ in range(8)]
            barrier()
            sdt = time.perf_counter() - t1
            sk = [t for i_, t in enumerate(stm) if i_ % 4 == 3]
            out["single_circuit"] = {"value": 8 * its / sdt, "unit": "iterations/s", "ms_per_fit": sdt / 8 * 1e3, "steps": 8, "warmup": 2,
                                     "k_iter_mid_avg_launch_ms": float(np.mean([t["mid_ms"] for t in sk])) / its,
                                     "k_sweep_par_avg_launch_ms": float(np.mean([t["sweep_seq_ms"] for t in sk])) / its,
                                     "fact_ms_per_fit": float(np.mean([t["fact_ms"] for t in sk])),
                                     "batch_x_single": value / (8 * its / sdt),
                                     "note": "one circuit per step on the same GPU: a latency chain occupying a few percent of the chip"}
            if a.workload in PUBLISHED:
                out["single_circuit"]["x_published"] = out["single_circuit"]["value"] / PUBLISHED[a.workload]["value"]
        if batch_res0 is not None:
            # ... against the same problem object fitted alone (cm_caviar_run): bit for bit, all eight outputs (mu, beta, lam, shape,
            # rate, phi, phi_cov, z).  A mismatch makes the run fail (exit code 1 after the line is printed).
            q = sets[0][0]
            q.run(*run_args, **opts)
            single_res0 = q.fetch()
            out["batch_equals_single"] = bool(all(np.array_equal(np.asarray(x, dtype=float), np.asarray(y_, dtype=float), equal_nan=True)
                                                  for x, y_ in zip(batch_res0[:8], single_res0[:8])))
            out["batch_equals_single_note"] = ("circuit 0 of the timed lock-step batch vs its own single fit on the same problem object, "
                                               "bitwise over mu, beta, lam, shape, rate, phi, phi_cov, z")
            del batch_res0, single_res0
        if world == 1 and B > 1 and not a.no_end_to_end and a.workload != "cfg5":
            out["end_to_end"] = end_to_end(_lib, ctx, local_rank, wl, pri, a, min(B, 256))     # (256 circuits whatever the step's batch: 7.8 GB of dense host arrays)
        if world == 1 and B > 1 and a.streams_leg and G == 1:
            # the same B circuits with the library's default scheduling (cm_options.interleave = 0): two half-batches at a time, the second on the
            # context's twin stream from the library's helper thread
            probs = sets[0]
            for _ in range(2):
                run_batch(probs, pri, **opts)
            ctx.synchronize()
            gc.collect(); gc.disable()
            t1 = time.perf_counter()
            lwall = []
            for _ in range(6):
                ts_ = time.perf_counter()
                ltm = run_batch(probs, pri, **opts)
                lwall.append((time.perf_counter() - ts_) * 1e3)
            ctx.synchronize()
            ldt = time.perf_counter() - t1
            gc.enable()
            if os.environ.get("CAVIAR_BENCH_TRACE_STEPS"):
                print("interleaved leg step wall ms:", " ".join("%.2f" % v for v in lwall), file=sys.stderr)
            same = []
            for ci in (0, B - 1):                              # one circuit of each half against its own single fit
                got = probs[ci].fetch()
                probs[ci].run(*run_args, **opts)
                one = probs[ci].fetch()
                same.append(bool(all(np.array_equal(np.asarray(x, dtype=float), np.asarray(y_, dtype=float), equal_nan=True) for x, y_ in zip(got[:8], one[:8]))))
                del got, one
            lval = B * 6 * its / ldt
            out["interleaved"] = {"value": lval, "unit": "iterations/s", "steps": 6, "warmup": 2, "ms_per_step": ldt / 6 * 1e3, "x_one_batch": lval / value,
                                  "circuits_per_launch": sorted({int(t.batch) for t in ltm}), "equals_single": all(same),
                                  "note": "cm_caviar_run_batch with its default scheduling: the batch as two halves in flight on two streams of the context, so "
                                          "that the dependency-bound phases of one half (Gauss-Seidel chains, Newton steps, last Cholesky panels, false-negative "
                                          "scan) overlap the wide phases of the other.  Kernels of the two halves then share the GPU and per-kernel durations "
                                          "are no longer those of a kernel alone on the chip, which is why the timed region -- and `roofline` with it -- "
                                          "keeps one lock-step batch per step (`--interleave` times this form instead)"}
            if not all(same):
                out["batch_equals_single"] = False
        if world == 1 and not a.no_cpu_baseline and sims:
            out["cpu_baseline"] = cpu_baseline(sims[0], pri, wl, wl.get("fit"))
        emit(json.dumps(out))
        if out.get("batch_equals_single") is False:
            print("bench.py: circuit 0 of the batch differs from its single fit", file=sys.stderr)
            exit_code = 1
    for probs in sets:
        for q in probs:
            q.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return exit_code


# HBM bytes per launch from the committed rocprofv3 PMC summaries: profiles/pmc_traffic.json, generated from the per-kernel CSVs by
# tools/pmc_traffic.py (nothing copied by hand); traffic = 2 x FETCH_SIZE + WRITE_SIZE per the MI355X guide's gfx950 correction.
PMC_FAMILIES = {   # bench kernel / family -> (kernels whose traffic counts, the kernel whose dispatch count is the family's launch count)
    "k_iter_mid": (("k_iter_mid", "k_iter_listed_flat", "k_iter_overflow"), ("k_iter_mid", "k_iter_listed_flat")),
    "k_mc_dead": (("k_mc_dead",), ("k_mc_dead",)),
    "k_sweep_par": (("k_sweep_par",), ("k_sweep_par",)),
    "k_chol_wide+k_chol_gpanel": (("k_chol_diag", "k_chol_gpanel", "k_chol_wide"), ("k_chol_wide",)),
    "k_chol_panel": (("k_chol_first", "k_chol_panel"), ("k_chol_panel",)),
    "k_chol_bulk+k_chol_trsm": (("k_chol_first", "k_chol_bulk", "k_chol_trsm"), ("k_chol_bulk<false, true>", "k_chol_bulk<true, true>")),   # one pair step per rank-64 pass
    "tiled solve (k_build_precision, k_chol_*, k_inv_*, column sums)": (
        ("k_fill_prior", "k_build_precision", "k_chol_first", "k_chol_panel", "k_chol_bulk", "k_chol_trsm", "k_chol_diag", "k_chol_gpanel", "k_chol_wide", "k_inv_seed", "k_inv_T", "k_inv_X21", "k_inv_T2", "k_inv_X212",
         "k_bp_prepare", "k_mu_beta_rows", "k_tri_matvec", "k_colsum_partial", "k_colsum_final"), ("k_mu_beta_rows", "k_tri_matvec")),   # (one k_mu_beta_rows -- round 3: k_tri_matvec -- per solve; k_build_precision runs in row bands)
}


def pmc_traffic(workload, circuits_per_launch, family):
    """(bytes per launch, source) of a kernel family from profiles/pmc_traffic.json, or (None, None) when no profile of this workload /
    batch size is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path) or family not in PMC_FAMILIES:
        return None, None
    with open(path) as f:
        js = json.load(f)
    key = "%s/%d" % (workload, circuits_per_launch)
    ent = js.get("entries", {}).get(key)
    if not ent:
        return None, None
    flag = "<true" if circuits_per_launch > 1 else "<false"
    members, counters = PMC_FAMILIES[family]
    base = lambda k: k.split("<")[0]
    tot = sum(v["calls"] * (2.0 * v["fetch_kb"] + v["write_kb"]) * 1024 for k, v in ent.items() if base(k) in members and flag in k)
    n = sum(v["calls"] for k, v in ent.items() if any(k.startswith(c) and (len(c) > len(base(c)) or base(k) == c) for c in counters) and flag in k)
    if n == 0 or tot == 0:
        return None, None
    return int(tot / n), " + ".join(js.get("sources", {}).get(key, [])) + " (separate --pmc passes; tools/pmc_traffic.py)"


if __name__ == "__main__":
    sys.exit(main() or 0)
