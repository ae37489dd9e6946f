"""Chains sharded over GPUs: one process per GPU, ``torch.distributed`` for the set-up traffic.

Chains are independent (reference ``gibbs/parallel.py:20-41`` runs each in its own process), so the
only communication is (1) ONE broadcast of the fixed design arrays from rank 0 -- RCCL over xGMI when
the backend is ``nccl``, each array moved as a device tensor -- and (2) a gather of the recorded
``(alpha, beta, tau)`` rows at the end.  There is no per-iteration collective.

``torch`` is plumbing here (process group, broadcast); the sampler itself is the HIP engine.  The
compute backend is injected (``engine_factory``) so that the sharding logic can be exercised on CPU
with ``gloo`` in the test-suite; the default factory is the HIP engine and raises without a GPU.
"""
import numpy as np

from ._problem import FlatProblem, chain_generators, default_start


def shard_chains(n_chains, world_size, rank):
    """Chain ids owned by ``rank``: chain c lives on rank ``c % world_size``."""
    return [c for c in range(n_chains) if c % world_size == rank]


def _dist():
    import torch.distributed as dist
    return dist


def broadcast_problem(prob, src=0, device=None):
    """Broadcast a :class:`FlatProblem` from rank ``src``; every rank returns an equal problem.

    ``prob`` is ignored on the other ranks (may be None).  With ``device`` (a ``torch.device`` of the
    local GPU) the arrays travel as device tensors, i.e. over RCCL/xGMI with the ``nccl`` backend.
    """
    import torch
    dist = _dist()
    rank = dist.get_rank()
    arrays = prob.to_arrays() if rank == src else None
    meta = [[(k, v.shape, str(v.dtype)) for k, v in arrays.items()]] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src)
    out = {}
    for name, shape, dtype in meta[0]:
        if rank == src:
            t = torch.from_numpy(arrays[name].reshape(-1).copy())
        else:
            t = torch.empty(int(np.prod(shape)), dtype=getattr(torch, np.dtype(dtype).name))
        if device is not None:
            t = t.to(device)
        dist.broadcast(t, src=src)
        out[name] = t.cpu().numpy().reshape(shape)
    return prob if rank == src else FlatProblem.from_arrays(out)


def _hip_engine_factory(prob, keys, device):
    from ._engine import Engine
    return Engine(prob, keys, device=device)


def run_sharded(prob, n_chains, size, burnin=0, random_state=None, start=None, device=0,
                engine_factory=None, gather=True):
    """Run ``n_chains`` chains of ``size`` iterations split over the ranks of the default process group.

    Every rank derives the same per-chain generators from ``random_state`` (chain k's generator is
    the one the reference would give its k-th copy), draws start values and Philox keys for ITS chains
    only, and runs them as one device batch.  Returns ``(alpha, beta, tau)`` with a leading chain axis
    in global chain order on every rank when ``gather`` (else only this rank's chains).
    """
    dist = _dist()
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = shard_chains(n_chains, world, rank)
    gens = chain_generators(random_state, n_chains)
    keep = size - burnin
    a = np.zeros((len(mine), keep, prob.q))
    b = np.zeros((len(mine), keep, prob.p))
    t = np.zeros((len(mine), keep))
    if mine:
        starts, keys = [], []
        for c in mine:
            st = dict(start) if start is not None else default_start(gens[c], prob)
            starts.append(st)
            keys.append(int(gens[c].bit_generator.random_raw()))
        eng = (engine_factory or _hip_engine_factory)(prob, keys, device)
        for i, st in enumerate(starts):
            eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
        a, b, t = eng.run(size, burnin)
        if hasattr(eng, 'close'):
            eng.close()
    if not gather:
        return a, b, t
    parts = [None] * world
    dist.all_gather_object(parts, (mine, a, b, t))
    A = np.zeros((n_chains, keep, prob.q))
    B = np.zeros((n_chains, keep, prob.p))
    T = np.zeros((n_chains, keep))
    for ids, pa, pb, pt in parts:
        for i, c in enumerate(ids):
            A[c], B[c], T[c] = pa[i], pb[i], pt[i]
    return A, B, T
