"""Chains sharded over GPUs, one PROCESS per GPU -- without PyTorch.

Chains are independent (reference ``gibbs/parallel.py:20-41`` runs each in its own joblib process), so a multi-GPU run
is replicas plus ONE broadcast of the fixed problem arrays at set-up: the root rank lays the problem out
and uploads it, the other ranks receive the device arrays over RCCL / xGMI (``occ_create_distributed`` in the C ABI,
librccl opened with dlopen by the engine library).  There is no per-iteration collective; the recorded
``(alpha, beta, tau)`` rows are gathered at the end.

Two communicators with one interface (``rank``, ``world``, ``barrier``, ``allreduce_max``, ``bcast_obj``,
``allgather_obj``):

* :class:`FileComm` -- a rendezvous directory on the node (``/dev/shm`` or ``/tmp``): enough for everything the host
  side of a single-node launch needs, used for the one thing RCCL cannot do for itself (handing rank 0's
  ``ncclUniqueId`` to the other ranks), by the CPU tests, and as the fallback when librccl cannot be used;
* :class:`RcclComm` -- ``ncclCommInitRank`` through the C ABI (``occ_comm_*``): the communicator the device broadcast
  runs on; its host-side collectives are staged through a device buffer.

(The in-process alternative -- one process driving several GPUs with a host thread each -- is
``LogitICARGibbs(..., devices=[...])``; it needs none of this.)
"""
import ctypes as C
import os
import pickle
import shutil
import threading
import time

import numpy as np

from ._problem import chain_generators, default_start


def shard_chains(n_chains, world_size, rank):
    """Chain ids owned by ``rank``: chain c lives on rank ``c % world_size``."""
    return [c for c in range(n_chains) if c % world_size == rank]


# ---------------------------------------------------------------------------------------------------------------
class FileComm:
    """Collectives of a single-node process group through files in a rendezvous directory.

    Every collective is an all-gather of pickled objects: rank r writes ``<op>.<r>`` (atomically, by rename) and reads
    the files of the ranks it needs.  Operation numbers advance in lock step on all ranks (every rank must make the
    same calls in the same order, as with any communicator)."""

    def __init__(self, rank, world, path, timeout=300.0, session=False):
        self.rank, self.world, self.path, self.timeout = int(rank), int(world), path, float(timeout)
        self._op = 0
        self._nonce = ''
        os.makedirs(path, exist_ok=True)
        if session:
            self._open_session()

    def _open_session(self):
        """A directory name can outlive a launch (a worker group restarted by the same agent, two launches from one shell,
        a crashed run that never removed it): operation numbers restart at 0 and stale files -- an old ncclUniqueId, old
        results -- would be read as current.  Rank 0 therefore opens a SESSION: it removes what it finds, then publishes a
        nonce (its pid and start time) that prefixes every file name of this launch; the other ranks take a nonce only
        from a LIVE rank 0 that is their sibling (same parent process), so the leftover of a dead launch is never taken."""
        sess = os.path.join(self.path, 'session')
        if self.rank == 0:
            for f in os.listdir(self.path):
                try:
                    os.remove(os.path.join(self.path, f))
                except OSError:
                    pass
            self._nonce = '%d-%x' % (os.getpid(), time.time_ns())
            tmp = sess + '.tmp'
            with open(tmp, 'w') as fh:
                fh.write('%s %d' % (self._nonce, os.getppid()))
            os.replace(tmp, sess)
            return
        t0 = time.monotonic()
        while True:
            try:
                nonce, ppid = open(sess).read().split()
                pid = int(nonce.split('-')[0])
                alive = os.path.exists('/proc/%d' % pid) if os.path.isdir('/proc') else True
                if alive and int(ppid) == os.getppid():
                    self._nonce = nonce
                    return
            except (OSError, ValueError):
                pass
            if time.monotonic() - t0 > self.timeout:
                raise TimeoutError('rank %d found no live session of rank 0 in %s within %.0f s' % (self.rank, self.path, self.timeout))
            time.sleep(0.002)

    @classmethod
    def from_env(cls, timeout=300.0):
        """Rank / world size from the launcher's environment (``RANK``, ``WORLD_SIZE`` as torch.distributed.run and
        most MPI-style launchers set them).  The directory is keyed by the rendezvous port, the run id, the elastic
        restart count and the launcher's PID -- all ranks of one launch are children of the same agent process -- and
        rank 0 opens a session in it (``_open_session``), so files of an earlier launch or attempt are never read."""
        rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
        base = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else '/tmp'
        tag = '%s_%s_%s_%d' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'run'),
                               os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'), os.getppid())
        return cls(rank, world, os.path.join(base, 'occ_rdv_' + tag), timeout, session=True)

    def _file(self, op, r):
        return os.path.join(self.path, '%s%06d.%d' % (self._nonce + '.' if self._nonce else '', op, r))

    def _put(self, op, obj):
        tmp = self._file(op, self.rank) + '.tmp'
        with open(tmp, 'wb') as fh:
            pickle.dump(obj, fh, protocol=pickle.HIGHEST_PROTOCOL)
        os.replace(tmp, self._file(op, self.rank))

    def _get(self, op, r):
        f = self._file(op, r)
        t0 = time.monotonic()
        while not os.path.exists(f):
            if time.monotonic() - t0 > self.timeout:
                raise TimeoutError('rank %d waited %.0f s for rank %d (operation %d) in %s' % (self.rank, self.timeout, r, op, self.path))
            time.sleep(0.0005)
        with open(f, 'rb') as fh:
            return pickle.load(fh)

    def allgather_obj(self, obj):
        op, self._op = self._op, self._op + 1
        self._put(op, obj)
        return [obj if r == self.rank else self._get(op, r) for r in range(self.world)]

    def bcast_obj(self, obj, root=0):
        # (every rank writes a token so that the root cannot run ahead and be overtaken by a later operation's files)
        return self.allgather_obj(obj if self.rank == root else None)[root]

    def barrier(self):
        self.allgather_obj(None)

    def allreduce_max(self, x):
        return np.max(np.stack([np.asarray(v, dtype=np.float64) for v in self.allgather_obj(np.asarray(x, dtype=np.float64))]), axis=0)

    def close(self):
        """Collective, two-phase: after the last barrier every rank leaves a token saying it has read that barrier's files;
        rank 0 removes the directory only when it has seen every token (a fixed sleep let a slow rank lose the file it was
        still polling for)."""
        self.barrier()
        bye = lambda r: os.path.join(self.path, '%sbye.%d' % (self._nonce + '.' if self._nonce else '', r))
        if self.rank != 0:
            try:
                open(bye(self.rank), 'w').close()
            except OSError:
                pass
            return
        t0 = time.monotonic()
        for r in range(1, self.world):
            while not os.path.exists(bye(r)) and time.monotonic() - t0 < min(self.timeout, 30.0):
                time.sleep(0.001)
        shutil.rmtree(self.path, ignore_errors=True)


class RcclComm:
    """``ncclCommInitRank`` through the engine library (``occ_comm_*``).  ``side`` is any communicator that can
    broadcast a small object (the 128-byte unique id): a :class:`FileComm`."""

    def __init__(self, side, device, init_timeout=180.0):
        from . import _lib
        self._lib = _lib.load()
        self.rank, self.world, self.device, self.side = side.rank, side.world, int(device), side
        uid = (C.c_uint8 * 128)()
        ok = True
        if self.rank == 0:
            ok = self._lib.occ_comm_unique_id(uid) == 0
        # all ranks or none: a rank that cannot open librccl must not leave the others inside ncclCommInitRank
        word = side.bcast_obj(bytes(uid) if ok else None, 0)
        if word is None:
            raise RuntimeError('RCCL is not usable on rank 0: ' + self._err(None))
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(word)
        box = {}

        def init():
            box['code'] = self._lib.occ_comm_create(self.world, self.rank, buf, self.device, C.byref(h))

        th = threading.Thread(target=init, daemon=True)
        th.start()
        th.join(init_timeout)
        good = (not th.is_alive()) and box.get('code') == 0
        if not all(side.allgather_obj(bool(good))):
            raise RuntimeError('ncclCommInitRank did not succeed on every rank (this rank: %s)'
                               % ('ok' if good else ('timed out' if th.is_alive() else self._err(None))))
        self.handle = h

    def _err(self, h):
        msg = self._lib.occ_comm_last_error(h)
        return msg.decode() if msg else ''

    def _check(self, code):
        if code != 0:
            raise RuntimeError('RCCL communicator failure: ' + self._err(self.handle))

    def barrier(self):
        self._check(self._lib.occ_comm_barrier(self.handle))

    def allreduce_max(self, x):
        v = np.ascontiguousarray(np.atleast_1d(np.asarray(x, dtype=np.float64)))
        self._check(self._lib.occ_comm_allreduce_max(self.handle, C.c_void_p(v.ctypes.data), v.size))
        return v if np.ndim(x) else float(v[0])

    def bcast_bytes(self, data, root=0):
        n = np.array([len(data) if self.rank == root else 0], dtype=np.int64)
        self._check(self._lib.occ_comm_broadcast_host(self.handle, C.c_void_p(n.ctypes.data), 8, root))
        buf = np.frombuffer(data, dtype=np.uint8).copy() if self.rank == root else np.empty(int(n[0]), dtype=np.uint8)
        if buf.size:
            self._check(self._lib.occ_comm_broadcast_host(self.handle, C.c_void_p(buf.ctypes.data), buf.size, root))
        return buf.tobytes()

    def bcast_obj(self, obj, root=0):
        return pickle.loads(self.bcast_bytes(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL) if self.rank == root else b'', root))

    def allgather_obj(self, obj):
        return [self.bcast_obj(obj if self.rank == r else None, r) for r in range(self.world)]

    def close(self):
        if getattr(self, 'handle', None):
            self._lib.occ_comm_destroy(self.handle)
            self.handle = None
        self.side.close()


def init_comm(device=None, prefer_rccl=True, timeout=300.0):
    """The process group of this launch: an :class:`RcclComm` when ``device`` is given and librccl works on every rank,
    else the :class:`FileComm` (all ranks take the same branch).  Returns ``(comm, note)``."""
    side = FileComm.from_env(timeout)
    if device is None or not prefer_rccl:
        return side, 'file rendezvous (no device communicator requested)'
    try:
        return RcclComm(side, device), 'rccl (ncclCommInitRank over a file rendezvous of the unique id)'
    except Exception as exc:   # every rank raises or none does (see RcclComm.__init__)
        return side, 'file rendezvous (RCCL unusable: %s)' % exc


# ---------------------------------------------------------------------------------------------------------------
def broadcast_problem(prob, comm, root=0):
    """The full :class:`FlatProblem` on every rank, through the communicator's object broadcast (host data).  This is
    the generic route (CPU tests, engines that are not the HIP engine); the HIP engine does not need it:
    :func:`distributed_engine` moves the design arrays device to device and gives the other ranks' hosts only the sizes
    and hyper-parameters."""
    from ._problem import FlatProblem
    arrays = comm.bcast_obj(prob.to_arrays() if comm.rank == root else None, root)
    return prob if comm.rank == root else FlatProblem.from_arrays(arrays)


def distributed_engine(prob, comm, keys, root=0, device=None):
    """This rank's HIP engine for ``keys``; ``prob`` is needed on ``root`` only.  Returns ``(engine, meta)`` where
    ``meta`` (a :class:`ProblemMeta` everywhere but on the root, which keeps its problem) has what start values need."""
    from ._engine import Engine, ProblemMeta
    meta = comm.bcast_obj(ProblemMeta.of(prob).to_dict() if comm.rank == root else None, root)
    mine = prob if comm.rank == root else ProblemMeta(**meta)
    if isinstance(comm, RcclComm):
        return Engine.distributed(mine, comm, keys, root), mine
    # no device communicator: every rank builds its own engine from the host arrays
    full = broadcast_problem(prob, comm, root)
    if device is None:
        device = int(os.environ.get('LOCAL_RANK', '0'))
    return Engine(full, keys, device=device), full


def _hip_engine_factory(prob, keys, device):
    from ._engine import Engine
    return Engine(prob, keys, device=device)


def run_sharded(prob, n_chains, size, burnin=0, random_state=None, start=None, device=0,
                engine_factory=None, gather=True, comm=None):
    """Run ``n_chains`` chains of ``size`` iterations split over the ranks of ``comm``.

    Every rank derives the same per-chain generators from ``random_state`` (chain k's generator is the one the
    reference would give its k-th copy, ``gibbs/base.py:293-306``), draws start values and Philox keys for ITS chains
    only -- chain c lives on rank ``c % world`` -- and runs them as one device batch.  Returns ``(alpha, beta, tau)``
    with a leading chain axis in global chain order on every rank when ``gather`` (else only this rank's chains).
    ``engine_factory(prob, keys, device)`` builds the compute backend (default: the HIP engine)."""
    if comm is None:
        comm = FileComm.from_env()
    world, rank = comm.world, comm.rank
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
    parts = comm.allgather_obj((mine, a, b, t))
    A = np.zeros((n_chains, keep, prob.q))
    B = np.zeros((n_chains, keep, prob.p))
    T = np.zeros((n_chains, keep))
    for ids, pa, pb, pt in parts:
        for i, c in enumerate(ids):
            A[c], B[c], T[c] = pa[i], pb[i], pt[i]
    return A, B, T
