"""Worker of tests/test_distributed_cpu.py: one rank of a world_size-2 group on CPU.

The sharding driver is the product code (occuspytial_amd.distributed); the compute backend is the CPU oracle injected
as ``engine_factory`` (tests may use the oracle; the product default is the HIP engine).  Two communicators are
exercised: the product's own FileComm (``file``) and a gloo process group of torch.distributed wrapped in the same
four-method interface (``gloo``; torch appears in tests only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class OracleEngine:
    """Engine-shaped adapter: a batch of chains = a list of one-chain oracle samplers."""

    def __init__(self, prob, keys, device):
        from oracle.occ_oracle import OracleSampler
        self.chains = [OracleSampler(prob, k) for k in keys]
        self.n_chains = len(keys)

    def set_start(self, i, alpha, beta, tau, eta):
        self.chains[i].set_start(alpha, beta, tau, eta)

    def run(self, size, burnin):
        out = [c.run(size, burnin) for c in self.chains]
        return tuple(np.stack([o[j] for o in out]) for j in range(3))

    def close(self):
        pass


class GlooComm:
    """torch.distributed (gloo) behind the communicator interface of occuspytial_amd.distributed."""

    def __init__(self):
        import torch.distributed as dist
        dist.init_process_group(backend='gloo')
        self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()

    def allgather_obj(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def bcast_obj(self, obj, root=0):
        box = [obj]
        self.dist.broadcast_object_list(box, src=root)
        return box[0]

    def barrier(self):
        self.dist.barrier()

    def allreduce_max(self, x):
        return np.max(np.stack(self.allgather_obj(np.asarray(x, dtype=np.float64))), axis=0)

    def close(self):
        self.dist.barrier()
        self.dist.destroy_process_group()


def main():
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.distributed import FileComm, broadcast_problem, run_sharded, shard_chains
    from occuspytial_amd.utils import make_lattice_problem

    out_path, n_chains, kind = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    comm = GlooComm() if kind == 'gloo' else FileComm.from_env(timeout=120.0)
    rank, world = comm.rank, comm.world
    prob = None
    if rank == 0:
        Q, W, X, y, *_ = make_lattice_problem(8, 9, visits=3, p=2, q=2, random_state=4)
        del W[5], y[5]  # one not-surveyed site
        prob = FlatProblem(Q, W, X, y)
    prob = broadcast_problem(prob, comm, root=0)
    assert prob.n == 72 and prob.not_surveyed == [5]
    A, B, T = run_sharded(prob, n_chains, size=12, burnin=2, random_state=77, engine_factory=OracleEngine, comm=comm)
    mine = shard_chains(n_chains, world, rank)
    # the host-side collectives a benchmark uses
    mx = comm.allreduce_max(np.array([float(rank), -float(rank)]))
    assert mx.tolist() == [world - 1.0, 0.0]
    np.savez(out_path + f'.rank{rank}.npz', A=A, B=B, T=T, mine=np.array(mine))
    comm.close()


if __name__ == '__main__':
    main()
