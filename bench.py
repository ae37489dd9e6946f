#!/usr/bin/env python
"""Headline benchmark: Gibbs iterations/sec of LogitICARGibbs on a 100x100 ICAR lattice.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--chains-per-gpu C]

A "step" is ONE Gibbs iteration of every chain resident on a GPU (all seven conditional updates of
the reference's LogitICARGibbs.step(), logit.py:254-266, for C chains batched in each kernel).
Workload (BASELINE.json metric / SURVEY.md 8d): 100x100 queen lattice, 10 000 sites, 5 visits per
site, p = q = 2, synthetic data (data seed 0, sampler seed 10), default hyper-parameters, 4 chains
per GPU.  For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank runs its own
4 chains -- chains are the natural shard, weak scaling -- after ONE RCCL broadcast of the fixed design
arrays from rank 0; there is no per-iteration collective.  value = chain-iterations of all ranks /
max-over-ranks wall time of exactly K timed steps.

The JSON line also carries
  roofline     : the dominant kernel (k_iter: tau, right-hand side, the whole MINRES solve of eta, projection
                 and beta sums of one iteration, all chains): algorithmic bytes per launch / its mean launch
                 time, measured live right after the timed region with two HIP events around each of 200
                 k_iter launches on the engine's main stream while the chains keep running (occ_profile);
  cpu_baseline : the CPU oracle (C restatement of the reference loop, oracle/) timed on one host core
                 for a bounded number of iterations of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def minres_bytes_per_launch(prob, n_chains, sell_entries):
    """Algorithmic bytes of ONE MINRES step of the joint system for all chains (DESIGN.md "Roofline
    accounting"): what an implementation that keeps the vectors in HBM must move per step.

    per chain and site : reads g_{k-1}, p_{k-2}, p_{k-3}, w_{k-4}, w_{k-3}, x (6 x 16 B) + omega_b (8 B);
                         writes p_{k-1}, g_k, w_{k-2}, x (4 x 16 B)                   -> 168 B
    shared by the chains: Q diagonal (8 B/site) + SELL-64 off-diagonals (4 B index + 8 B value per
                         stored slot, padding included)
    Neighbour gathers hit lines already counted; partial sums are O(blocks).
    """
    return n_chains * prob.n * 168 + prob.n * 8 + sell_entries * 12


def iter_bytes_per_launch(prob, n_chains, sell_entries, steps_per_chain):
    """Algorithmic bytes of one k_iter launch: `steps_per_chain` MINRES steps per chain (mean MINRES
    iterations + 3, measured) at the per-step figure above, plus the one-off terms of SURVEY 8(d):
    right-hand side (omega_b, z, X, two noise vectors, warm start read; rhs written) and projection /
    beta sums (X read again; eta and x written).  k_iter keeps the vectors in registers between steps, so its
    real HBM traffic (roofline.traffic) is far below this figure: `achieved` is the rate an HBM-resident
    implementation would need to match its speed."""
    n, p = prob.n, prob.p
    per_step_chain = n * 168
    shared_per_step = n * 8 + sell_entries * 12
    rhs = n * (8 + 1 + 8 * p + 16 + 32 + 8)
    tail = n * (8 * p + 8 + 16)
    return int(n_chains * (steps_per_chain * per_step_chain + rhs + tail) + steps_per_chain * shared_per_step)


def pmc_traffic(kernel, workload_key):
    """HBM bytes per launch from the committed PMC pass (profiles/r01_pmc_hbm_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this script, gfx950 FETCH_SIZE correction applied).
    Counters cannot be read from inside this process; the value is only reported for the workload the
    pass was taken on, else null."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_hbm_traffic.json')
    try:
        d = json.load(open(path))
    except OSError:
        return None
    if d.get('workload') != workload_key:
        return None
    # template instances carry their arguments in the name (occ::k_iter<8, 1>)
    hits = [v for name, v in d['kernels'].items() if name == kernel or name.startswith(kernel + '<')]
    return hits[0]['hbm_bytes_per_launch'] if hits else None


def sell_entry_count(prob):
    """Stored SELL-64 off-diagonal slots (64 x max off-diagonal row length per 64-row slice)."""
    deg = np.diff(prob.Q.indptr) - (prob.Q.diagonal() != 0)
    n = prob.n
    pad = (-n) % 64
    d = np.concatenate([deg, np.zeros(pad, dtype=deg.dtype)]).reshape(-1, 64)
    return int(d.max(axis=1).sum() * 64)


def cpu_baseline(prob, target_seconds=12.0):
    """Oracle (CPU port) iterations/sec on one host core, bounded sample of the same workload."""
    from oracle.occ_oracle import OracleSampler
    from occuspytial_amd._problem import chain_generators, default_start
    gen = chain_generators(10, 1)[0]
    st = default_start(gen, prob)
    orc = OracleSampler(prob, int(gen.bit_generator.random_raw()))
    orc.set_start(st['alpha'], st['beta'], st['tau'], st['eta'])
    t0 = time.perf_counter()
    orc.run(5, 4)
    per = (time.perf_counter() - t0) / 5
    iters = int(max(20, min(5000, target_seconds / max(per, 1e-6))))
    t0 = time.perf_counter()
    orc.run(iters, iters - 1)
    dt = time.perf_counter() - t0
    return {'value': iters / dt, 'unit': 'iterations/s', 'cores': 1, 'kind': 'port',
            'sample': f'1 chain x {iters} iterations of the same 100x100 workload, oracle/occ_oracle.c '
                      f'(sequential C restatement of the reference loop), {dt:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--chains-per-gpu', type=int, default=4)
    ap.add_argument('--lattice', type=int, nargs=2, default=[100, 100])
    ap.add_argument('--visits', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N '
                             '--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...')
        args.gpus = world

    import torch
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
    from occuspytial_amd.utils import make_lattice_problem

    dist = None
    if world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ):
        # launched by torch.distributed.run: one rank per GPU, RCCL ("nccl" backend) for the set-up traffic
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on C-level stdout at communicator creation: keep stdout for the
        # one JSON line by pointing fd 1 at stderr until the first collective has run
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    # ---- inputs: rank 0 generates, every other rank receives them over RCCL -----------------------
    rows, cols = args.lattice
    prob = None
    if rank == 0:
        Q, W, X, y, *_ = make_lattice_problem(rows, cols, visits=args.visits, p=2, q=2, random_state=0)
        prob = FlatProblem(Q, W, X, y)
    if dist is not None:
        from occuspytial_amd.distributed import broadcast_problem
        prob = broadcast_problem(prob, src=0, device=torch.device('cuda', local_rank))

    C = args.chains_per_gpu
    total_chains = C * world
    gens = chain_generators(10, total_chains)
    mine = list(range(rank * C, rank * C + C))
    starts = [default_start(gens[c], prob) for c in mine]
    keys = [int(gens[c].bit_generator.random_raw()) for c in mine]
    eng = Engine(prob, keys, device=local_rank)
    for i, st in enumerate(starts):
        eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (includes the hipGraph capture), then K timed steps ------------------------------
    if args.warmup > 0:
        eng.run(args.warmup, args.warmup - 1)
    stats_warm = eng.stats()
    barrier()
    t0 = time.perf_counter()
    a, b, t = eng.run(args.steps, args.steps - 1)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=torch.device('cuda', local_rank))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(b)) and np.all(t > 0)
    stats = eng.stats()

    if rank == 0:
        # ---- roofline of the dominant kernel, live, HIP events on the engine's stream --------------
        prof = eng.profile(reps=200)
        st1 = eng.stats()
        sell = sell_entry_count(prob)
        fused = bool(stats['persistent_solve']) and prof['iter']['launches'] > 0
        if fused:
            # the timed region itself: every k_iter launch is clocked from inside (first workgroup in to last chain
            # out, constant-rate device wall clock), and its MINRES iterations are counted on the device
            d_solves = max(1, stats['solves'] - stats_warm['solves'])
            steps = (stats['krylov_total'] - stats_warm['krylov_total']) / d_solves + 3.0
            kname = 'k_iter'
            ka = {'avg_us': stats['iter_kernel_mean_us'], 'launches': stats['iter_kernel_launches']}
            bytes_launch = iter_bytes_per_launch(prob, C, sell, steps)
            per_iter = {'iter': 1, 'z_ob': 1}
            prof['iter_in_situ_hip_events'] = prof['iter']
            prof['iter'] = dict(ka, total_us=ka['avg_us'] * ka['launches'])
            timing = ('every k_iter launch of the TIMED REGION clocked inside the kernel (device wall clock, first '
                      'workgroup in to last chain out; HIP events cannot bracket a graph node without adding nodes); '
                      'cross-checks: avg_launch_us_by_kernel.iter_in_situ_hip_events = two HIP events on the main '
                      'stream around each of 200 further launches, launched one by one after the timed region (the '
                      'device idles between them: slower), and the rocprofv3 kernel_stats of the same command in '
                      'profiles/')
        else:
            steps = None
            kname, ka = 'k_minres', prof['minres']
            bytes_launch = minres_bytes_per_launch(prob, C, sell)
            per_iter = {'eta_init': 1, 'minres': stats['krylov_cap'] + 3, 'beta_partial': 1, 'z_ob': 1}
            timing = ('HIP events on the engine stream around 200 replays of a captured solve prefix '
                      '(k_eta_init + k_minres launches 1..8, every launch cache-cold as in the real solve), '
                      'k_eta_init subtracted; = kernel duration + one dependent-launch boundary')
        achieved = bytes_launch / (ka['avg_us'] * 1e-6) / 1e9 if ka['avg_us'] > 0 else 0.0
        total_us = sum(prof[k]['avg_us'] * per_iter[k] for k in per_iter)
        out = {
            'metric': 'Gibbs iterations/sec on 100x100 ICAR lattice, 4 chains; 1/2/4/8 GPUs',
            'value': total_chains * args.steps / elapsed,
            'unit': 'iterations/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': f'{rows}x{cols} queen ICAR lattice, {prob.n} sites, {args.visits} visits/site, '
                            f'p=q=2, {C} chains per GPU batched in every kernel (BASELINE configs[1] data, '
                            'the metric\'s 4 chains)',
                'chains_per_gpu': C, 'total_chains': total_chains, 'sites': prob.n, 'visit_rows': prob.R,
                'parallelism': f'chains sharded {C}/GPU, no data-path collective',
                'krylov_iterations_mean': round(stats['krylov_mean'], 2),
                'krylov_cap': stats['krylov_cap'], 'stalls': stats['stalls'],
                'fused_iteration_kernel': bool(stats['persistent_solve']), 'main_stream_cus': stats['main_stream_cus'],
                'threads_per_block': stats['threads_per_block'],
                'device_ms_last_run': round(stats['last_run_ms'], 3),
            },
            'roofline': {
                'bound': 'hbm', 'kernel': kname,
                'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4),
                'traffic': pmc_traffic('occ::' + kname, f'{rows}x{cols} queen lattice, {C} chains') if args.visits == 5 else None,
                'traffic_source': 'profiles/r01_pmc_hbm_traffic.json (separate rocprofv3 --pmc passes, bytes per launch)',
                'bytes_per_launch': bytes_launch, 'avg_launch_us': round(ka['avg_us'], 3),
                'launches_timed': ka['launches'],
                'timing': timing,
                'minres_steps_per_launch': round(steps, 2) if steps else None,
                'algorithmic_bytes_per_minres_step': minres_bytes_per_launch(prob, C, sell),
                'share_of_critical_path_launch_time': round(ka['avg_us'] * per_iter['iter' if fused else 'minres'] / total_us, 3) if total_us else None,
                'avg_launch_us_by_kernel': {k: round(v['avg_us'], 3) for k, v in prof.items()},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(prob)
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
