"""Worker of tests/test_distributed_cpu.py: one rank of a world_size-2 gloo group on CPU.

The sharding driver is the product code (occuspytial_amd.distributed); the compute backend is the
CPU oracle injected as ``engine_factory`` (tests may use the oracle; the product default is the HIP
engine)."""
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

    def set_start(self, i, alpha, beta, tau, eta):
        self.chains[i].set_start(alpha, beta, tau, eta)

    def run(self, size, burnin):
        out = [c.run(size, burnin) for c in self.chains]
        return tuple(np.stack([o[j] for o in out]) for j in range(3))


def main():
    import torch.distributed as dist
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.distributed import broadcast_problem, run_sharded, shard_chains
    from occuspytial_amd.utils import make_lattice_problem

    out_path, n_chains = sys.argv[1], int(sys.argv[2])
    dist.init_process_group(backend='gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    prob = None
    if rank == 0:
        Q, W, X, y, *_ = make_lattice_problem(8, 9, visits=3, p=2, q=2, random_state=4)
        del W[5], y[5]  # one not-surveyed site
        prob = FlatProblem(Q, W, X, y)
    prob = broadcast_problem(prob, src=0)
    assert prob.n == 72 and prob.not_surveyed == [5]
    A, B, T = run_sharded(prob, n_chains, size=12, burnin=2, random_state=77, engine_factory=OracleEngine)
    mine = shard_chains(n_chains, world, rank)
    np.savez(out_path + f'.rank{rank}.npz', A=A, B=B, T=T, mine=np.array(mine))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
