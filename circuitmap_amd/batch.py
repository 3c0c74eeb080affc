"""Batch-of-circuits sweep: independent circuits sharded across the GPUs of one node.

CAVIaR does not shard within a circuit (the sweep is a sequential chain with one decision per neuron), but
circuits are independent, so the multi-GPU path is embarrassingly parallel: one process per GPU
(`torch.distributed`, backend "nccl" = RCCL over xGMI on MI355X, "gloo" in CPU tests), rank r fits circuits
r, r + W, r + 2W, ...  The only collectives are a broadcast of the shared priors / fit options from rank 0 before the
sweep, optionally a scatter of the circuits themselves when only rank 0 holds them (`scatter=True`), and a gather of the
per-circuit results to rank 0 after it; nothing is exchanged inside the iteration loop.
(The reference has no multi-device path at all: it scales out with SLURM job arrays of single-GPU processes,
reference scripts/generate_loho_cv_slurm_scripts.py:8-36.)
"""
import numpy as np

RESULT_KEYS = ("mu", "beta", "shape", "rate", "phi", "phi_cov", "z")


def shard_indices(n_circuits, rank, world_size):
    """Circuits owned by `rank`: round-robin, so any prefix of the batch is balanced."""
    return list(range(rank, n_circuits, world_size))


def _default_fit(psc, stim, priors, fit_options, device_id):
    """One circuit on one GPU through the drop-in `caviar()`."""
    names = ["mu", "beta", "lam", "shape", "rate", "phi", "phi_cov", "z"]
    from .optimise import caviar
    res = caviar(psc, stim, priors["mu"], priors["beta"], priors["shape"], priors["rate"], priors["phi"], priors["phi_cov"],
                 device_id=device_id, **fit_options)
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


def _fit_lockstep(circuits, mine, priors, fit_options, device_id, max_group, streams=1, compact=False):
    """This rank's circuits as lock-step batches on its GPU (cm_caviar_run_batch): the circuit is the outer grid dimension of every
    launch.  `streams` > 1: that many batches at a time, each on its own context (stream + mailboxes) from its own host thread --
    the latency-bound phases of one batch (its Gauss-Seidel chains, its Cholesky panels) then overlap the wide phases of the
    others.  Returns the result dicts in the order of `mine`."""
    from . import _lib
    from .optimise.caviar import Problem, fetch_batch, run_batch
    names = ["mu", "beta", "lam", "shape", "rate", "phi", "phi_cov", "z"]
    # A group must fit the GPU: a circuit holds three ld x ld float64 solve matrices (ld = N rounded up to 32: 24 MB at N = 1000, 0.6 GB at
    # N = 5000), the staged dense inputs of its encoding and ~100 bytes per targeted entry; the group is capped at 60 % of the free device
    # memory (256 circuits of N = 5000 would ask for 155 GB + staging).
    if mine:
        c0 = circuits[mine[0]]
        if len(c0) == 2:
            n0, k0, t0 = np.asarray(c0[1]).shape[0], np.asarray(c0[1]).shape[1], np.asarray(c0[0]).shape[1]
        else:
            n0, k0, t0 = c0[2], c0[3], c0[4]
        ld = (n0 + 31) // 32 * 32
        per = 3 * ld * ld * 8 + (n0 * k0 + k0 * t0) * 8 + 64 * n0 * 8 * 3 + 200 * 10 * k0 + (1 << 20)
        try:
            import torch
            free = torch.cuda.mem_get_info(device_id)[0]
            max_group = max(1, min(max_group, int(0.6 * free / streams) // per))
        except Exception:
            pass
    groups = [mine[g0:g0 + max_group] for g0 in range(0, len(mine), max_group)]
    out = {}

    def make_problems(ctx, idx):
        cs = [circuits[ci] for ci in idx]
        if len(cs[0]) == 2:
            return Problem.create_batch(ctx, cs, compact=compact)    # host arrays (psc, stim): one encoding pass for the group (compact: float32 traces + stimulus triples over PCIe)
        # ("device", float64 CUDA tensor [K*T + N*K], N, K, T): scattered circuits, never on this host
        return Problem.create_batch(ctx, [("device", N, K, T, t.data_ptr(), t.data_ptr() + K * T * 8) for _, t, N, K, T in cs])

    def fit_group(ctx, idx):
        probs = make_problems(ctx, idx)
        try:
            run_batch(probs, priors, **fit_options)
            for ci, res in zip(idx, fetch_batch(probs)):             # one device-to-host copy for the group
                out[ci] = dict(zip(names, res))
        finally:
            for q in probs:
                q.close()

    streams = max(1, min(int(streams), len(groups)))
    if streams == 1:
        ctx = _lib.default_context(device_id)
        for idx in groups:
            fit_group(ctx, idx)
        ctx.release_cache()          # the shared default context lives as long as the process: its pooled slabs go back to the driver (torch's allocator cannot see them)
    else:
        from concurrent.futures import ThreadPoolExecutor
        import queue
        ctxs = queue.Queue()
        for _ in range(streams):
            ctxs.put(_lib.Context(device_id))

        def work(idx):
            ctx = ctxs.get()
            try:
                fit_group(ctx, idx)
            finally:
                ctxs.put(ctx)
        with ThreadPoolExecutor(max_workers=streams) as ex:
            list(ex.map(work, groups))
        while not ctxs.empty():
            ctxs.get().close()
    return [out[ci] for ci in mine]


def _scatter_circuits(circuits, n_total, shape, rank, world, device, group):
    """Rank 0 holds every circuit's host arrays; rank r receives the circuits shard_indices(n_total, r, world) -- one `dist.scatter` per slot
    (RCCL over xGMI with the "nccl" backend: rank 0 stages a slot's `world` circuits in HBM, the receivers get device tensors that go
    straight into cm_problem_create(CM_MEM_DEVICE); gloo: CPU tensors).  Returns {circuit index: circuit} for this rank, rank 0 included."""
    import torch
    import torch.distributed as dist
    N, K, T = shape
    L = K * T + N * K
    mine = shard_indices(n_total, rank, world)
    n_slots = (n_total + world - 1) // world
    local = {}
    for slot in range(n_slots):
        recv = torch.empty(L, dtype=torch.float64, device=device)
        parts = None
        if rank == 0:
            parts = []
            for r in range(world):
                idx = shard_indices(n_total, r, world)
                if slot < len(idx):
                    psc, stim = circuits[idx[slot]]
                    flat = np.concatenate([np.ascontiguousarray(psc, dtype=np.float64).ravel(), np.ascontiguousarray(stim, dtype=np.float64).ravel()])
                    parts.append(torch.from_numpy(flat).to(device))
                else:
                    parts.append(torch.zeros(L, dtype=torch.float64, device=device))     # (equal-size payloads: a rank without a circuit in this slot)
        dist.scatter(recv, parts, src=0, group=group)
        if slot < len(mine):
            if device.type == "cuda":
                local[mine[slot]] = ("device", recv, N, K, T)
            else:
                a = recv.numpy()
                local[mine[slot]] = (a[:K * T].reshape(K, T), a[K * T:].reshape(N, K))
    if device.type == "cuda":
        torch.cuda.synchronize(device)                                # the library reads the tensors on its own stream
    return local


def fit_batch(circuits, priors=None, fit_options=None, fit_fn=None, device=None, group=None, circuits_per_launch=256, streams=1, scatter=False, compact=False):
    """Fit a list of circuits [(psc K x T, stim N x K), ...] across the ranks of the default process group.

    Rank 0's `priors` / `fit_options` are broadcast to every rank.  Returns, on rank 0, a list of result dicts
    (mu, beta, shape, rate, phi, phi_cov, z) in circuit order; on other ranks None.  Without an initialised process
    group it degenerates to one GPU.  `fit_fn(psc, stim, priors, fit_options, device_id)` is injectable so that the sharding /
    collective logic can be exercised on CPU with the gloo backend (then the circuits are fitted one by one).

    `scatter=False`: every rank already holds every circuit's arrays (e.g. read from shared storage) and fits its share.
    `scatter=True`: only rank 0 holds data (`circuits` may be None elsewhere); it sends each rank its share first (`dist.scatter`, one
    slot of `world` circuits at a time; on GPUs the receivers keep the circuits in HBM and never copy them to their hosts).  All
    circuits must then have one shape (N, K, T).

    On a GPU each rank fits its share as LOCK-STEP BATCHES of up to `circuits_per_launch` circuits (C ABI cm_caviar_run_batch): one
    set of kernel launches per phase for a whole batch, one Gauss-Seidel workgroup per circuit.  The library runs two batches at a time
    (cm_options.interleave: a single batch of 32 circuits or more as two halves on two streams of the context), the dependency-bound phases
    of one overlapping the wide phases of the other; `streams` > 1 adds that many contexts with a host thread each on top, which has not
    paid since the chains, the false-negative scan and the tiled solve were shortened (round 3).  A single fit is a latency chain that
    occupies a few percent of an MI355X; a batch of 256 delivers ~20x its iteration rate.  Every result is bit-identical to the circuit's
    own fit.  `compact=True` (host circuits): the traces cross PCIe as float32 and the stimulus matrix as its non-zero entries
    (cm_problem_create_batch_ex: a quarter of the bytes; bitwise the same circuits when the values are float32-representable).
    """
    import torch
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    backend = dist.get_backend(group) if distributed else None
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    device_id = device.index if device.type == "cuda" else 0

    # ---- broadcast of the shared configuration (rank 0 -> all)
    shape = None
    if distributed:
        if rank == 0:
            shape = (np.asarray(circuits[0][1]).shape + (np.asarray(circuits[0][0]).shape[1],)) if len(circuits) else (0, 0, 0)
            if scatter and any(np.asarray(c[1]).shape + (np.asarray(c[0]).shape[1],) != shape for c in circuits):
                raise ValueError("fit_batch(scatter=True) needs circuits of one shape")
        box = [(priors, fit_options, len(circuits), shape)] if rank == 0 else [None]
        dist.broadcast_object_list(box, src=0, group=group, device=device if backend == "nccl" else None)
        priors, fit_options, n_total, shape = box[0]
        if scatter and n_total > 0:
            circuits = _scatter_circuits(circuits, n_total, shape, rank, world, device, group)
    else:
        n_total = len(circuits)
    fit_options = dict(fit_options or {})
    if shape is not None:
        N, K = int(shape[0]), int(shape[1])
    else:
        N, K = np.asarray(circuits[0][1]).shape if len(circuits) else (0, 0)
    if priors is None:
        from .model import Model
        priors = Model(N).priors

    # ---- local fits (no communication)
    mine = shard_indices(n_total, rank, world)
    L = packed_length(N, K)
    n_slots = (n_total + world - 1) // world                       # equal-size gather payloads
    local = torch.zeros((n_slots, L), dtype=torch.float64)
    if fit_fn is None:
        for slot, res in enumerate(_fit_lockstep(circuits, mine, priors, fit_options, device_id, max(1, int(circuits_per_launch)), streams, compact=bool(compact) and not scatter)):
            local[slot] = torch.from_numpy(_pack(res, N, K))
    else:
        for slot, ci in enumerate(mine):
            c = circuits[ci]
            if len(c) != 2:                                           # a scattered circuit on a GPU: the injected fit wants host arrays
                _, t, n_, k_, t_ = c
                a = t.cpu().numpy()
                c = (a[:k_ * t_].reshape(k_, t_), a[k_ * t_:].reshape(n_, k_))
            psc, stim = c
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
