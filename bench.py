#!/usr/bin/env python
"""Headline benchmark: Gibbs iterations/sec of LogitICARGibbs on a 100x100 ICAR lattice.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--chains-per-gpu C]

A "step" is ONE Gibbs iteration of every chain resident on a GPU (all seven conditional updates of
the reference's LogitICARGibbs.step(), logit.py:254-266, for C chains batched in each kernel).
Workload (BASELINE.json metric / SURVEY.md 8d): 100x100 queen lattice, 10 000 sites, 5 visits per
site, p = q = 2, synthetic data (data seed 0, sampler seed 10), default hyper-parameters, 4 chains
per GPU.  For N > 1 (launched by torch.distributed.run -- used as a process launcher only: this script does not import
torch -- one rank per GPU) every rank runs its own 4 chains -- chains are the natural shard, weak scaling -- after ONE
RCCL broadcast (ncclCommInitRank / ncclBroadcast through the engine's C ABI) of the laid-out problem arrays from rank
0's device to the other devices; there is no per-iteration collective.  value = chain-iterations of all ranks /
max-over-ranks wall time of exactly K timed steps.  The same line carries `split_4_chains`: SURVEY 8(e)'s split of
the metric's 4 chains over the N GPUs (4 / 2 / 1 / <= 1 per GPU).  Without a launcher, `--gpus N` drives the N GPUs
from this one process, a host thread each (EngineGroup).

The JSON line also carries
  roofline     : the dominant kernel (k_iter: tau, right-hand side, the whole MINRES solve of eta, projection
                 and beta sums of one iteration, all chains): algorithmic bytes per launch / its mean DISPATCH
                 duration -- the start / stop events of hipExtLaunchKernel on the engine's main stream, i.e. the
                 timestamps rocprofv3 reports for the kernel -- over 200 launches that continue the chains right after
                 the timed region (occ_profile); roofline.in_kernel_clock: every launch of the timed region clocked
                 from inside the kernel;
  cpu_baseline : the CPU oracle (C restatement of the reference loop, oracle/) in reference-faithful mode (dense
                 eigenfactor prior draw, one thread per chain): 4 chains on 4 threads, and one chain per host core
                 (rank 0, N = 1 only; bounded samples).
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


def whole_iteration_bytes(prob, K, n_no_frac=0.4):
    """SURVEY 8(d), algorithmic bytes of one Gibbs iteration of ONE chain at K MINRES iterations (R_e = R)."""
    n, p, q, R = prob.n, prob.p, prob.q, prob.R
    m = prob.Q.nnz
    csr = 12 * m + 4 * (n + 1)
    n_no = n_no_frac * n
    R_no = n_no_frac * R
    terms = [8 * n * (p + 2),                               # omega_b
             8 * n + csr,                                   # tau
             8 * n * (p + 3) + 4 * m + 4 * (n + 1),         # eta rhs (edge-form prior term)
             K * (csr + 8 * n + 2 * 15 * 8 * n),            # eta Krylov
             24 * n,                                        # projection
             8 * n * (p + 3),                               # beta
             8 * R * q + 8 * R + n,                         # omega_a
             8 * R * q + 9 * R,                             # alpha
             8 * n_no * (p + 1) + 8 * R_no * q + n_no]      # z
    return int(sum(terms))


def pmc_traffic(kernel, workload_key):
    """HBM bytes per launch from the latest committed PMC pass for this workload (profiles/r*_pmc_hbm_traffic.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this script, gfx950 FETCH_SIZE correction applied;
    tools/profile_round.sh).  Counters cannot be read from inside this process; the value is only reported for a
    workload a pass was taken on, else null.  Returns (bytes per launch, file name)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*pmc_hbm_traffic.json')), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get('workload') != workload_key:
            continue
        # template instances carry their arguments in the name (occ::k_iter<8, 1, 1>)
        hits = [v for name, v in d['kernels'].items() if name == kernel or name.startswith(kernel + '<')]
        if hits:
            return hits[0]['hbm_bytes_per_launch'], os.path.relpath(path, ROOT)
    return None, None


def sell_entry_count(prob):
    """Stored SELL-64 off-diagonal slots (64 x max off-diagonal row length per 64-row slice)."""
    deg = np.diff(prob.Q.indptr) - (prob.Q.diagonal() != 0)
    n = prob.n
    pad = (-n) % 64
    d = np.concatenate([deg, np.zeros(pad, dtype=deg.dtype)]).reshape(-1, 64)
    return int(d.max(axis=1).sum() * 64)


def _oracle_chain(prob, gen, E=None):
    from oracle.occ_oracle import OracleSampler
    from occuspytial_amd._problem import default_start
    st = default_start(gen, prob)
    orc = OracleSampler(prob, int(gen.bit_generator.random_raw()))
    if E is not None:
        orc.set_dense_eigen(E)
    orc.set_start(st['alpha'], st['beta'], st['tau'], st['eta'])
    return orc


def _timed_threads(chains, iters):
    """`iters` iterations of every oracle chain, one OS thread per chain (ctypes releases the GIL): the reference's
    one-process-per-chain fan-out (gibbs/parallel.py:38-41).  Returns wall seconds."""
    import threading
    ths = [threading.Thread(target=c.run, args=(iters, iters - 1)) for c in chains]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return time.perf_counter() - t0


def cpu_baseline(prob, mode='auto', target_seconds=10.0):
    """the CPU restatement (oracle/, kind "port") in REFERENCE-FAITHFUL mode on the host cores of this box:
    dense eigenfactor prior draw (logit.py:64-67, 77: E from numpy eigh, n x (n-1), one dense matvec per iteration),
    joint MINRES rtol 1e-5, one OS thread per chain -- `value` = the metric's 4 chains on 4 threads; beside it the same
    with one chain per host core (a throughput ceiling) and the build's own edge-form prior on one core.  Bounded
    samples of the same workload (about `target_seconds` each); the eigh set-up is timed separately and is not part
    of any rate (the reference pays it at construction: 50.8 s at 100x100 in the survey container)."""
    from oracle.occ_oracle import dense_eigenfactor
    from occuspytial_amd._problem import chain_generators
    ncore = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    ncore = min(ncore, 16)   # a one-GPU box's CPU share (the host may show every core of a multi-GPU node)
    dense = mode == 'dense_eigen' or (mode == 'auto' and prob.n <= 10000)
    out = {'unit': 'iterations/s', 'kind': 'port'}

    def rate(n_threads, E, seed):
        chains = [_oracle_chain(prob, g, E) for g in chain_generators(seed, n_threads)]
        per = _timed_threads(chains, 3) / 3          # every thread runs 3 iterations: seconds per iteration under this load
        iters = int(max(5, min(5000, target_seconds / max(per, 1e-6))))
        dt = _timed_threads(chains, iters)
        return n_threads * iters / dt, iters, dt

    v1, it1, dt1 = rate(1, None, 10)
    out['edge_form_1_thread'] = {'value': v1, 'threads': 1, 'sample': f'1 chain x {it1} iterations, edge-form prior draw (the build\'s own algorithm), {dt1:.1f} s'}
    if not dense:
        out.update({'value': v1, 'cores': 1, 'threads': 1, 'mode': 'edge',
                    'sample': out['edge_form_1_thread']['sample'] + ' -- no reference-faithful baseline exists at this size '
                              '(the dense eigenfactor is O(n^2) memory: the reference cannot construct the problem)'})
        return out
    t0 = time.perf_counter()
    E = dense_eigenfactor(prob.Q)
    t_eigh = time.perf_counter() - t0
    v4, it4, dt4 = rate(4, E, 10)
    out.update({'value': v4, 'cores': 4, 'threads': 4, 'mode': 'dense_eigen',
                'sample': f'4 chains x {it4} iterations on 4 threads (one per chain), oracle/occ_oracle.c with the reference\'s dense '
                          f'eigenfactor prior draw ({E.nbytes / 1e6:.0f} MB matrix, one dense matvec per iteration), {dt4:.1f} s; '
                          f'eigh set-up {t_eigh:.1f} s on the host, not counted'})
    if ncore > 4:
        va, ita, dta = rate(ncore, E, 11)
        out['all_cores'] = {'value': va, 'threads': ncore, 'mode': 'dense_eigen',
                            'sample': f'{ncore} chains x {ita} iterations, one thread per core of this GPU\'s CPU share (at most 16), {dta:.1f} s'}
    out['eigh_setup_s'] = round(t_eigh, 1)
    return out


class _Solo:
    """The communicator of a one-process run."""
    rank, world = 0, 1

    def barrier(self):
        pass

    def allreduce_max(self, x):
        return x

    def bcast_obj(self, obj, root=0):
        return obj

    def allgather_obj(self, obj):
        return [obj]

    def close(self):
        pass


def timed_run(eng, comm, steps, warmup, checkpoint=False):
    """W untimed steps (graph capture included), then exactly K timed steps bracketed by a barrier and a device
    synchronisation on both sides; the elapsed time is the MAX over ranks.  Every timed iteration is recorded (burnin 0:
    nothing of the reference's loop is skipped inside the timed region).  `eng` may be None (a rank without chains in the
    4-chain split): it still takes part in the barriers.  Returns (elapsed, stats before, stats after, the last recorded
    (alpha, beta, tau) row of every local chain, the chains' state at the start of the timed region when asked for)."""
    stats_warm, ckpt = None, None
    if eng is not None and warmup > 0:
        eng.run(warmup, warmup - 1)
    if eng is not None:
        if checkpoint:
            ckpt = eng.checkpoint()
        stats_warm = eng.stats()
        eng.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    rec = eng.run(steps, 0) if eng is not None else None
    if eng is not None:
        eng.synchronize()
    comm.barrier()
    elapsed = float(comm.allreduce_max(time.perf_counter() - t0))
    last = []
    if rec is not None:
        a, b, t = rec
        assert a.shape[1] == steps and np.all(np.isfinite(a)) and np.all(np.isfinite(b)) and np.all(t > 0)
        last = [np.concatenate([a[c, -1], b[c, -1], t[c, -1:]]).tolist() for c in range(a.shape[0])]
    return elapsed, stats_warm, (eng.stats() if eng is not None else None), last, ckpt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--chains-per-gpu', type=int, default=4)
    ap.add_argument('--lattice', type=int, nargs=2, default=[100, 100])
    ap.add_argument('--visits', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-mode', choices=['auto', 'dense_eigen', 'edge'], default='auto')
    ap.add_argument('--no-split', action='store_true', help='skip the second measurement of N > 1 (the metric\'s 4 chains split over the GPUs)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    in_process = world == 1 and args.gpus > 1   # no launcher: ONE process drives the N GPUs, a host thread each
    from occuspytial_amd import _lib
    n_visible = max(1, _lib.load().occ_device_count())   # (counting devices does not initialise the GPU)
    local_rank %= n_visible   # one rank per GPU on a full node; lets a 2-rank rehearsal share a one-GPU box
    if world > 1:
        args.gpus = world

    from occuspytial_amd._engine import Engine, EngineGroup
    from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
    from occuspytial_amd.utils import make_lattice_problem

    # ---- process group: RCCL called directly (no PyTorch); launched by torch.distributed.run or any launcher that
    # sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT.  RCCL prints a banner on C-level stdout at communicator
    # creation: keep stdout for the one JSON line by pointing fd 1 at stderr meanwhile.
    comm, comm_note = _Solo(), 'single process'
    if world > 1:
        from occuspytial_amd.distributed import init_comm
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            comm, comm_note = init_comm(device=local_rank)
            comm.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    # ---- inputs: rank 0 generates them; the other ranks' DEVICES receive the laid-out arrays over RCCL ----------
    rows, cols = args.lattice
    prob = None
    if rank == 0:
        Q, W, X, y, *_ = make_lattice_problem(rows, cols, visits=args.visits, p=2, q=2, random_state=0)
        prob = FlatProblem(Q, W, X, y)

    def make_engine(chain_ids, gens, problem):
        """Engine of this rank for the global chains `chain_ids` (None when it has none)."""
        if world > 1:
            from occuspytial_amd.distributed import distributed_engine
            # (collective: every rank takes part in the broadcast even when it runs no chain of this split)
            keys = [int(gens[c].bit_generator.random_raw()) for c in chain_ids] or [1]
            eng, mine = distributed_engine(problem, comm, keys, device=local_rank)
            if not chain_ids:
                eng.close()
                return None, mine
        elif in_process:
            mine = problem
            keys = [int(gens[c].bit_generator.random_raw()) for c in chain_ids]
            eng = EngineGroup(problem, keys, [g % n_visible for g in range(args.gpus)])
        else:
            mine = problem
            keys = [int(gens[c].bit_generator.random_raw()) for c in chain_ids]
            eng = Engine(problem, keys, device=local_rank)
        return eng, mine

    def start_engine(eng, mine, chain_ids, gens, starts=None):
        for i, c in enumerate(chain_ids):
            st = starts[c] if starts else default_start(gens[c], mine)
            eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])

    C = args.chains_per_gpu
    n_dev = args.gpus
    total_chains = C * n_dev
    # chain c of the whole job owns the generator the reference would give its c-th copy (gibbs/base.py:293-306);
    # start values come from it first, then the Philox key (its next raw word)
    gens = chain_generators(10, total_chains)
    if world > 1:
        mine_ids = list(range(rank * C, rank * C + C))
        meta_prob = comm.bcast_obj(None if rank else {k: getattr(prob, k) for k in ('n', 'p', 'q', 'S', 'R', 'tau_rate', 'tau_shape', 'a_mu', 'a_prec', 'b_mu', 'b_prec')}, 0)
        from occuspytial_amd._engine import ProblemMeta
        host_prob = prob if rank == 0 else ProblemMeta(**meta_prob)
    else:
        mine_ids = list(range(total_chains))
        host_prob = prob
    starts = {c: default_start(gens[c], host_prob) for c in mine_ids}
    eng, mine = make_engine(mine_ids, gens, prob)
    start_engine(eng, mine, mine_ids, gens, starts)
    transport = eng.transport
    elapsed, stats_warm, stats, last_rows, ckpt0 = timed_run(eng, comm, args.steps, args.warmup, checkpoint=(rank == 0 and not in_process))
    # every chain's last recorded draw, by global chain number: the draws do not depend on how the chains are spread over
    # GPUs (chain c owns the generator the reference gives its c-th copy), so a multi-GPU line can be checked against a
    # one-GPU run of the same chains
    last_draws = {}
    for part in comm.allgather_obj(dict(zip(mine_ids, last_rows))):
        last_draws.update(part)

    # ---- N > 1: SURVEY 8(e)'s split of the metric's 4 chains (4 / 2 / 1 / <= 1 per GPU), same steps ---------------
    split = None
    if n_dev > 1 and not args.no_split:
        gens4 = chain_generators(10, 4)
        if world > 1:
            ids4 = [c for c in range(4) if c % world == rank]
        else:
            ids4 = list(range(4))
        st4 = {c: default_start(gens4[c], host_prob) for c in ids4}
        if in_process:
            keys4 = [int(gens4[c].bit_generator.random_raw()) for c in ids4]
            eng4, mine4 = EngineGroup(prob, keys4, [g % n_visible for g in range(min(n_dev, 4))]), prob
        else:
            eng4, mine4 = make_engine(ids4, gens4, prob)
        if eng4 is not None:
            start_engine(eng4, mine4, ids4, gens4, st4)
        el4, _, _, last4, _ = timed_run(eng4, comm, args.steps, args.warmup)
        split_draws = {}
        for part in comm.allgather_obj(dict(zip(ids4, last4))):
            split_draws.update(part)
        if eng4 is not None:
            eng4.close()
        per = [sum(1 for c in range(4) if c % n_dev == g) for g in range(n_dev)]
        split = {'value': 4 * args.steps / el4, 'unit': 'iterations/s', 'scaling': 'strong', 'total_chains': 4,
                 'chains_per_gpu': per, 'ms_per_step': 1e3 * el4 / args.steps,
                 'last_draws': [split_draws[c] for c in sorted(split_draws)],
                 'note': 'the metric\'s 4 chains, chain c on GPU c % N; one chain alone still pays the whole latency-bound '
                         'iteration (about 65 us at 100x100), so this split gains little over 1 GPU -- the weak line above '
                         '(4 chains on every GPU) is what more GPUs buy'}

    # ---- who took part (N > 1): every rank reports itself THROUGH the communicator the broadcast ran on -- with RCCL that
    # is ncclBroadcast between the ranks' devices, so a rank that RCCL does not reach cannot appear
    ranks_info = None
    if n_dev > 1:
        seen = comm.allgather_obj({'rank': rank, 'local_rank': local_rank, 'device': local_rank, 'chains': len(mine_ids), 'pid': os.getpid()})
        ranks_info = {'ranks_seen_by_rccl' if 'rccl' in comm_note.split(' ')[0] else 'ranks_seen_by_file_rendezvous': len(seen),
                      'distinct_devices': len({(r['device']) for r in seen}) if world > 1 else n_dev,
                      'communicator': comm_note, 'members': seen}

    if rank == 0:
        # ---- roofline of the dominant kernel, live, HIP events on the engine's stream --------------
        eng0 = eng.engines[0] if in_process else eng
        if in_process:
            stats, stats_warm = stats['per_device'][0], stats_warm['per_device'][0]
        # the profile pass REPLAYS the first 200 launches of the timed region (the chains are put back to where the region
        # started: same keys, same iterations, same solves), so the dispatch overhead it measures belongs to launches the
        # timed region made -- same MINRES steps per launch -- not to later ones
        stats_end = stats
        replay = ckpt0 is not None
        if replay:
            eng0.restore(ckpt0)
        stats_before_prof = eng0.stats()
        prof = eng0.profile(reps=min(200, args.steps))
        sell = sell_entry_count(prob)
        fused = bool(stats['persistent_solve']) and prof['iter']['launches'] > 0
        if fused:
            # the timed region itself: every k_iter launch is clocked from inside (first workgroup in to last chain out,
            # constant-rate device wall clock), and its MINRES iterations are counted on the device
            d_solves = max(1, stats['solves'] - stats_warm['solves'])
            steps = (stats['krylov_total'] - stats_warm['krylov_total']) / d_solves + 3.0
            kname = 'k_tiles' if stats['persistent_solve'] == 3 else 'k_iter'   # (k_tiles: the tile-looping form for large lattices, occ_tiles.hpp)
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
        # The number rocprofv3 reproduces (the in-kernel clock stops ~3 us short of what the profiler reports for
        # the same launch -- launch ramp and end-of-kernel release are outside a kernel's own view): the DISPATCH duration
        # of k_iter, from the start / stop events of hipExtLaunchKernel (the begin / end timestamps of the dispatch's
        # completion signal, the profiler's own basis), over the 200 in-situ launches that follow the timed region, with the
        # MINRES steps THOSE launches ran.  `frac` is quoted on this basis; the in-kernel clock of the timed region stays
        # beside it.
        in_kernel = {'achieved': round(achieved, 1), 'frac': round(achieved / HBM_PEAK_GBS, 4), 'avg_launch_us': round(ka['avg_us'], 3),
                     'launches_timed': ka['launches'], 'bytes_per_launch': bytes_launch,
                     'minres_steps_per_launch': round(steps, 2) if steps else None,
                     'basis': 'device wall clock read inside k_iter, first workgroup in to last chain out, every launch of the timed region'}
        stats_prof = eng0.stats()
        disp_us = stats_prof.get('profile_iter_dispatch_us', 0.0)
        dispatch = None
        if fused and disp_us > 0:
            # the same 200 launches by both clocks: what the dispatch adds around the kernel's own view of itself
            n0, m0 = stats_before_prof['iter_kernel_launches'], stats_before_prof['iter_kernel_mean_us']
            n1, m1 = stats_prof['iter_kernel_launches'], stats_prof['iter_kernel_mean_us']
            in_kernel_prof = (n1 * m1 - n0 * m0) / max(1, n1 - n0)
            overhead = max(0.0, disp_us - in_kernel_prof)
            m0 = stats_end['iter_kernel_mean_us']
            est = m0 + overhead                     # dispatch duration of the TIMED REGION's launches
            achieved = bytes_launch / (est * 1e-6) / 1e9
            dispatch = {'dispatch_overhead_us': round(overhead, 3), 'profile_pass_dispatch_us': round(disp_us, 3),
                        'profile_pass_in_kernel_us': round(in_kernel_prof, 3), 'profile_pass_launches': int(n1 - n0),
                        'profile_pass': ('a replay of the first launches of the timed region (chains restored to its start state)' if replay
                                         else 'further launches that continue the chains after the timed region'),
                        'profile_pass_minres_steps_per_launch': round(stats_prof['profile_minres_iterations'] + 3.0, 2)}
            ka = {'avg_us': est, 'launches': ka['launches']}
            timing = ('mean DISPATCH duration of the k_iter launches of the timed region = their mean by the in-kernel clock '
                      '(every launch; roofline.in_kernel_clock) + the dispatch overhead measured live on a REPLAY of the '
                      'region\'s first launches (chains restored to the region\'s start state: the same solves): '
                      'hipExtLaunchKernel start / stop events (the begin / end timestamps of the dispatch\'s completion signal = '
                      'what rocprofv3 --kernel-trace reports; launch ramp and end-of-kernel release included) minus the '
                      'in-kernel clock of the same launches (roofline.dispatch_basis).  HIP events cannot bracket a node of a '
                      'replayed graph without adding nodes.  Check: profiles/r04_bench_trace_region.json = the rocprofv3 kernel '
                      'trace of this command averaged over the timed region\'s launches')
        traffic, traffic_file = pmc_traffic('occ::' + kname, f'{rows}x{cols} queen lattice, {C} chains') if args.visits == 5 else (None, None)
        total_us = sum(prof[k]['avg_us'] * per_iter[k] for k in per_iter)
        # SURVEY 8(d)'s whole-iteration accounting beside the dominant kernel's: B_iter(measured K, R_e = R, n_no = 0.4 n)
        kmean = stats['krylov_mean']
        whole = whole_iteration_bytes(prob, kmean) * C
        out = {
            'metric': 'Gibbs iterations/sec on 100x100 ICAR lattice, 4 chains; 1/2/4/8 GPUs',
            'value': total_chains * args.steps / elapsed,
            'unit': 'iterations/s',
            'n_gpus': n_dev,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'last_draws': [last_draws[c] for c in sorted(last_draws)],
            'config': {
                'workload': f'{rows}x{cols} queen ICAR lattice, {prob.n} sites, {args.visits} visits/site, '
                            f'p=q=2, {C} chains per GPU batched in every kernel (BASELINE configs[1] data, '
                            'the metric\'s 4 chains)',
                'chains_per_gpu': C, 'total_chains': total_chains, 'sites': prob.n, 'visit_rows': prob.R,
                'parallelism': (f'chains sharded {C}/GPU, no data-path collective; '
                                + ('one process, one host thread per GPU' if in_process else 'one process per GPU' if world > 1 else 'one GPU')),
                'communicator': comm_note, 'problem_transport': transport,
                'krylov_iterations_mean': round(stats['krylov_mean'], 2),
                'krylov_cap': stats['krylov_cap'], 'stalls': stats['stalls'],
                'fused_iteration_kernel': bool(stats['persistent_solve']), 'solve_form': {0: 'one launch per MINRES step', 1: 'k_iter, any placement', 2: 'k_iter, one XCD per chain', 3: 'k_tiles'}[stats['persistent_solve']],
                'solve_workgroups_per_chain': stats['solve_workgroups'], 'main_stream_cus': stats['main_stream_cus'],
                'fused_fallbacks': stats['fused_fallbacks'],
                'threads_per_block': stats['threads_per_block'],
                'device_ms_last_run': round(stats['last_run_ms'], 3),
            },
            'roofline': {
                'bound': 'hbm', 'kernel': kname,
                'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4),
                'traffic': traffic,
                'traffic_source': (traffic_file or 'none for this workload') + ' (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch)',
                'bytes_per_launch': bytes_launch, 'avg_launch_us': round(ka['avg_us'], 3),
                'launches_timed': ka['launches'],
                'timing': timing,
                'minres_steps_per_launch': round(steps, 2) if steps else None,
                'in_kernel_clock': in_kernel,
                'dispatch_basis': dispatch,
                'algorithmic_bytes_per_minres_step': minres_bytes_per_launch(prob, C, sell),
                'share_of_critical_path_launch_time': round(ka['avg_us'] * per_iter['iter' if fused else 'minres'] / total_us, 3) if total_us else None,
                'avg_launch_us_by_kernel': {k: round(v['avg_us'], 3) for k, v in prof.items()},
                'whole_iteration': {
                    'note': 'SURVEY 8(d) accounting of ONE WHOLE Gibbs iteration (all seven conditionals, 15 vector passes per '
                            'MINRES iteration at the measured K) x chains per GPU / measured time per iteration on one GPU',
                    'bytes_per_iteration': whole, 'krylov_iterations_mean': round(kmean, 2),
                    'achieved': round(whole / (elapsed / args.steps) / 1e9, 1), 'unit': 'GB/s',
                    'frac': round(whole / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
            },
        }
        if n_dev > 1:
            # how to read a SCALE curve from these lines: `value` is WEAK scaling -- every GPU runs the metric's
            # 4 chains -- so value / n_gpus is what compares with the N = 1 line (BENCH); split_4_chains is the metric's
            # literal 4 chains divided over the GPUs (strong scaling: one chain alone still pays the latency-bound iteration)
            out['value_per_gpu'] = out['value'] / n_dev
            out['n1_equivalent'] = {'value': out['value'] / n_dev, 'unit': 'iterations/s',
                                    'note': 'value / n_gpus: each GPU runs the N = 1 workload (4 chains); compare with the N = 1 line'}
            out['scale_curve'] = 'read `value` (weak: 4 chains per GPU); efficiency = value(N) / (N * value(1))'
            out['ranks'] = ranks_info
        if split is not None:
            out['split_4_chains'] = split
        if n_dev == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(prob, args.cpu_baseline_mode)
        print(json.dumps(out), flush=True)
    eng.close()
    comm.close()


if __name__ == '__main__':
    main()
