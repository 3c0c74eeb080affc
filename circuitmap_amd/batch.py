"""Batch-of-circuits sweep: independent circuits sharded across the GPUs of one node.

CAVIaR does not shard within a circuit (the sweep is a sequential chain with one decision per neuron), but
circuits are independent, so the multi-GPU path is embarrassingly parallel: one process per GPU
(`torch.distributed`, backend "nccl" = RCCL over xGMI on MI355X, "gloo" in CPU tests), rank r fits circuits
r, r + W, r + 2W, ...  The only collectives are a broadcast of the shared priors / fit options from rank 0 before the
sweep and a gather of the per-circuit results to rank 0 after it; nothing is exchanged inside the iteration loop.
(The reference has no multi-device path at all: it scales out with SLURM job arrays of single-GPU processes,
reference scripts/generate_loho_cv_slurm_scripts.py:8-36.)
"""
import numpy as np

RESULT_KEYS = ("mu", "beta", "shape", "rate", "phi", "phi_cov", "z")


def shard_indices(n_circuits, rank, world_size):
    """Circuits owned by `rank`: round-robin, so any prefix of the batch is balanced."""
    return list(range(rank, n_circuits, world_size))


def _default_fit(psc, stim, priors, fit_options, device_id, ctx=None):
    """One circuit on one GPU.  `ctx` = a private `_lib.Context` (own stream and mailbox) when several fits share the GPU."""
    names = ["mu", "beta", "lam", "shape", "rate", "phi", "phi_cov", "z"]
    if ctx is None:
        from .optimise import caviar
        res = caviar(psc, stim, priors["mu"], priors["beta"], priors["shape"], priors["rate"], priors["phi"], priors["phi_cov"],
                     device_id=device_id, **fit_options)
        return dict(zip(names, res[:8]))
    from .optimise.caviar import Problem
    prob = Problem(ctx, psc, stim)
    try:
        prob.run(priors["mu"], priors["beta"], priors["shape"], priors["rate"], priors["phi"], priors["phi_cov"], **fit_options)
        res = prob.fetch()
    finally:
        prob.close()
    return dict(zip(names, res[:8]))


def _pack(result, N, K):
    """Flatten one circuit's result (without the N x K lam) into a float64 vector of fixed length."""
    parts = [np.asarray(result["mu"], float).reshape(N), np.asarray(result["beta"], float).reshape(N),
             np.asarray([result["shape"], result["rate"]], float), np.asarray(result["phi"], float).reshape(2 * N),
             np.asarray(result["phi_cov"], float).reshape(4 * N), np.asarray(result["z"], float).reshape(K)]
    return np.concatenate(parts)


def _unpack(vec, N, K):
    o = 0
    out = {}
    for key, n, shape in [("mu", N, (N,)), ("beta", N, (N,)), ("_sr", 2, (2,)), ("phi", 2 * N, (N, 2)),
                          ("phi_cov", 4 * N, (N, 2, 2)), ("z", K, (K,))]:
        out[key] = vec[o:o + n].reshape(shape).copy()
        o += n
    sr = out.pop("_sr")
    out["shape"], out["rate"] = float(sr[0]), float(sr[1])
    return out


def packed_length(N, K):
    return 8 * N + 2 + K


def fit_batch(circuits, priors=None, fit_options=None, fit_fn=None, device=None, group=None, per_gpu_concurrency=1):
    """Fit a list of circuits [(psc K x T, stim N x K), ...] (same N, K) across the ranks of the default process group.

    Rank 0's `priors` / `fit_options` are broadcast to every rank.  Returns, on rank 0, a list of result dicts
    (mu, beta, shape, rate, phi, phi_cov, z) in circuit order; on other ranks None.  Without an initialised process
    group it degenerates to a sequential loop on one GPU.  `fit_fn(psc, stim, priors, fit_options, device_id)` is
    injectable so that the sharding / collective logic can be exercised on CPU with the gloo backend.

    `per_gpu_concurrency` > 1 runs that many of this rank's fits at a time on its GPU, each from its own host thread with a private
    context (stream + mailbox).  A single fit is a latency chain that occupies ~2 % of an MI355X, so fits overlap: measured
    aggregate at N=1000, K=2000: 1.55x with 2, 2.1x with 3 (erratic beyond: the latency-critical chains start to queue behind each
    other's wide early-iteration kernels).  Results are identical to the serial loop (each fit is deterministic on its own stream).
    """
    import torch
    import torch.distributed as dist
    fit_fn = fit_fn or _default_fit
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    backend = dist.get_backend(group) if distributed else None
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    device_id = device.index if device.type == "cuda" else 0

    # ---- broadcast of the shared configuration (rank 0 -> all)
    if distributed:
        box = [(priors, fit_options, len(circuits))] if rank == 0 else [None]
        dist.broadcast_object_list(box, src=0, group=group, device=device if backend == "nccl" else None)
        priors, fit_options, n_total = box[0]
    else:
        n_total = len(circuits)
    fit_options = dict(fit_options or {})
    N, K = np.asarray(circuits[0][1]).shape if len(circuits) else (0, 0)
    if priors is None:
        from .model import Model
        priors = Model(N).priors

    # ---- local fits (no communication)
    mine = shard_indices(n_total, rank, world)
    L = packed_length(N, K)
    n_slots = (n_total + world - 1) // world                       # equal-size gather payloads
    local = torch.zeros((n_slots, L), dtype=torch.float64)
    if per_gpu_concurrency > 1 and fit_fn is _default_fit and len(mine) > 1:
        from concurrent.futures import ThreadPoolExecutor
        import threading
        from . import _lib
        tls = threading.local()

        def work(job):
            slot, ci = job
            if not hasattr(tls, "ctx"):
                tls.ctx = _lib.Context(device_id)                  # one private context per worker thread (closed by its finaliser)
            psc, stim = circuits[ci]
            return slot, _pack(_default_fit(psc, stim, priors, fit_options, device_id, ctx=tls.ctx), N, K)
        with ThreadPoolExecutor(max_workers=int(per_gpu_concurrency)) as ex:
            for slot, vec in ex.map(work, list(enumerate(mine))):
                local[slot] = torch.from_numpy(vec)
    else:
        for slot, ci in enumerate(mine):
            psc, stim = circuits[ci]
            local[slot] = torch.from_numpy(_pack(fit_fn(psc, stim, priors, fit_options, device_id), N, K))

    # ---- gather of the results (all -> rank 0)
    if not distributed:
        return [_unpack(local[s].numpy(), N, K) for s in range(len(mine))]
    local = local.to(device)
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, bufs, dst=0, group=group)
    if rank != 0:
        return None
    out = [None] * n_total
    for r in range(world):
        rows = bufs[r].cpu().numpy()
        for slot, ci in enumerate(shard_indices(n_total, r, world)):
            out[ci] = _unpack(rows[slot], N, K)
    return out
