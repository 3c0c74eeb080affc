"""Batch-of-circuits sharding over 2 ranks (gloo, CPU): broadcast of the shared configuration, round-robin ownership,
local fits, gather to rank 0.  The fit itself is injected (the CPU oracle here; the HIP path on a GPU node), so this
covers exactly the N > 1 logic that bench.py --gpus N and circuitmap_amd.batch use."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_fit(psc, stim, priors, fit_options, device_id):
    from oracle import caviar_oracle as co
    res = co.caviar(psc, stim, priors["mu"], priors["beta"], priors["shape"], priors["rate"], priors["phi"], priors["phi_cov"],
                    literal=False, **fit_options)
    return dict(zip(["mu", "beta", "lam", "shape", "rate", "phi", "phi_cov", "z"], res[:8]))


def _make_circuits(n):
    from circuitmap_amd.simulation import simulate
    out = []
    for c in range(n):
        sim = simulate(N=12, T=60, H=3, trials=40, seed=100 + c, connection_prob=0.3, lite=True)
        out.append((sim["psc"], sim["stim_matrix"]))
    return out


def _worker(rank, world, port, n_circuits, q, scatter=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from circuitmap_amd import batch
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        # scatter=True: the data exists on rank 0 only and travels to the other rank through the scatter
        circuits = _make_circuits(n_circuits) if (rank == 0 or not scatter) else None
        # only rank 0 supplies the options: the others must receive them through the broadcast
        opts = dict(iters=4, num_mc_samples=5, seed=2) if rank == 0 else None
        res = batch.fit_batch(circuits, priors=None, fit_options=opts, fit_fn=_oracle_fit, scatter=scatter)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_circuits,scatter", [(5, False), (2, False), (5, True), (1, True)])
def test_fit_batch_two_ranks_equals_serial(n_circuits, scatter):
    """scatter=False: both ranks hold every circuit; scatter=True: only rank 0 does, rank 1 passes circuits=None and receives its share
    (5 circuits: ranks own 3 and 2 -- the last slot carries a padding payload; 1 circuit: rank 1 owns nothing)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_circuits, q, scatter)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] is None and len(got[0]) == n_circuits
    # serial reference on this process
    from circuitmap_amd import batch
    serial = batch.fit_batch(_make_circuits(n_circuits), fit_options=dict(iters=4, num_mc_samples=5, seed=2), fit_fn=_oracle_fit)
    for a, b in zip(got[0], serial):
        for k in batch.RESULT_KEYS:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True), k   # bit-identical per circuit


def test_shard_indices_partition():
    from circuitmap_amd.batch import shard_indices
    for n in (0, 1, 7, 8, 9):
        for w in (1, 2, 8):
            parts = [shard_indices(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_bench_circuit_rotation():
    """bench.py with N ranks: at every step the ranks fit N different circuits, and over any N consecutive steps every rank
    fits every circuit once (equal work per GPU by construction)."""
    import bench
    for world in (1, 2, 4, 8):
        for step in range(3 * world):
            assert sorted(bench.circuit_for(r, step, world) for r in range(world)) == list(range(world))
        for r in range(world):
            assert sorted(bench.circuit_for(r, s, world) for s in range(5, 5 + world)) == list(range(world))
