"""ctypes binding of ``libocc_gibbs.so`` (the C ABI declared in ``include/occ_gibbs.h``).

There is no CPU fallback: if the shared library is missing, or no MI355X is usable, every entry
point that would compute raises.  Build the library with ``python -c "import __graft_entry__ as g;
g.build()"`` or ``make -C occuspytial_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libocc_gibbs.so')

OCC_OK = 0
OCC_E_BADARG, OCC_E_HIP, OCC_E_MINRES, OCC_E_CHOLESKY, OCC_E_STATE = -1, -2, -3, -4, -5
N_KERNEL_KINDS = 9
KERNEL_KINDS = ('omega_b', 'noise', 'eta_init', 'minres', 'beta_partial', 'omega_a', 'alpha_draw', 'z_ob', 'iter')


class OccProblem(C.Structure):
    _fields_ = [
        ('n', C.c_int64), ('n_surveyed', C.c_int64), ('n_rows', C.c_int64),
        ('p', C.c_int32), ('q', C.c_int32),
        ('q_indptr', C.c_void_p), ('q_indices', C.c_void_p), ('q_data', C.c_void_p),
        ('X', C.c_void_p), ('site_id', C.c_void_p), ('site_ptr', C.c_void_p),
        ('W', C.c_void_p), ('y', C.c_void_p),
        ('a_mu', C.c_void_p), ('a_prec', C.c_void_p), ('b_mu', C.c_void_p), ('b_prec', C.c_void_p),
        ('tau_rate', C.c_double), ('tau_shape', C.c_double),
        ('rsr_dim', C.c_int32), ('rsr_K', C.c_void_p), ('rsr_Q', C.c_void_p), ('rsr_E', C.c_void_p),
        ('prior_factor', C.c_void_p), ('prior_factor_cols', C.c_int64),
    ]


class OccStats(C.Structure):
    _fields_ = [
        ('iterations', C.c_int64), ('graph_launches', C.c_int64), ('eager_iterations', C.c_int64),
        ('stalls', C.c_int64), ('krylov_cap', C.c_int32), ('krylov_last', C.c_int32),
        ('krylov_mean', C.c_double), ('krylov_total', C.c_int64), ('solves', C.c_int64), ('last_run_ms', C.c_double),
        ('n_blocks_sites', C.c_int32), ('n_blocks_rows', C.c_int32), ('threads_per_block', C.c_int32),
        ('n_chains', C.c_int32), ('persistent_solve', C.c_int32), ('solve_workgroups', C.c_int32),
        ('main_stream_cus', C.c_int32), ('fused_fallbacks', C.c_int32), ('profile_minres_iterations', C.c_double),
        ('iter_kernel_launches', C.c_int64), ('iter_kernel_mean_us', C.c_double),
        ('repromotions', C.c_int32), ('stream_probes', C.c_int32), ('handover_mode', C.c_int32),
        ('stream_pairs_masked', C.c_int32), ('stream_pairs_plain', C.c_int32), ('demoted', C.c_int32),
        ('profile_iter_dispatch_us', C.c_double),
        ('stream_pairs_idle', C.c_int32), ('stream_pairs_evicted', C.c_int32),
    ]


# every symbol include/occ_gibbs.h declares: (name, restype, argtypes)
SYMBOLS = (
    ('occ_abi_version', C.c_int32, []),
    ('occ_device_count', C.c_int32, []),
    ('occ_last_error', C.c_char_p, [C.c_void_p]),
    ('occ_create', C.c_int, [C.POINTER(OccProblem), C.c_int32, C.POINTER(C.c_uint64), C.c_int32,
                             C.POINTER(C.c_void_p)]),
    ('occ_destroy', C.c_int, [C.c_void_p]),
    ('occ_set_keys', C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ('occ_set_start', C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    ('occ_step', C.c_int, [C.c_void_p]),
    ('occ_run', C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('occ_get_state', C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_void_p, C.c_int64,
                                C.POINTER(C.c_int64)]),
    ('occ_set_state', C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_void_p, C.c_int64]),
    ('occ_get_stats', C.c_int, [C.c_void_p, C.POINTER(OccStats)]),
    ('occ_profile', C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    ('occ_create_group', C.c_int, [C.POINTER(OccProblem), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]),
    ('occ_comm_unique_id', C.c_int, [C.c_void_p]),
    ('occ_comm_create', C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    ('occ_comm_destroy', C.c_int, [C.c_void_p]),
    ('occ_comm_barrier', C.c_int, [C.c_void_p]),
    ('occ_comm_allreduce_max', C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    ('occ_comm_broadcast_host', C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    ('occ_comm_last_error', C.c_char_p, [C.c_void_p]),
    ('occ_create_distributed', C.c_int, [C.POINTER(OccProblem), C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_void_p)]),
    ('occ_group_transport', C.c_char_p, [C.c_void_p]),
    ('occ_synchronize', C.c_int, [C.c_void_p]),
    ('occ_cond_tau', C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.POINTER(C.c_double)]),
    ('occ_cond_eta', C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_int32)]),
    ('occ_cond_beta', C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('occ_cond_alpha', C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('occ_cond_z', C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    ('occ_draw', C.c_int, [C.c_int32, C.c_int32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, C.c_void_p, C.c_void_p]),
)

_lib = None


class EngineUnavailable(RuntimeError):
    """The HIP engine cannot be used (library not built, or no usable gfx950 device)."""


ABI_VERSION = 6  # OCC_ABI_VERSION of include/occ_gibbs.h this binding was written against


def load():
    """Load the shared library and declare its prototypes (no GPU call is made here)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineUnavailable(
                f'{LIB_PATH} is missing: the HIP engine has not been built '
                '(run __graft_entry__.build() or `make -C occuspytial_amd/csrc`). There is no CPU fallback.')
        lib = C.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.occ_abi_version() != ABI_VERSION:  # a stale build: struct layouts would not match
            raise EngineUnavailable(f'{LIB_PATH} has ABI version {lib.occ_abi_version()}, this binding needs {ABI_VERSION}: rebuild it '
                                    '(`make -C occuspytial_amd/csrc`)')
        _lib = lib
    return _lib


def error_text(handle=None):
    msg = load().occ_last_error(handle)
    return msg.decode() if msg else ''


def raise_for(code, handle=None):
    """Map a status code to the exception the reference raises for the same condition."""
    if code == OCC_OK:
        return
    text = error_text(handle)
    if code in (OCC_E_BADARG, OCC_E_STATE):
        raise ValueError(text or 'bad argument')
    if code == OCC_E_MINRES:
        raise RuntimeError('MINRES solver did not converge!')          # reference logit.py:91-92
    if code == OCC_E_CHOLESKY:
        raise RuntimeError('Cholesky factorization/solver failed!')    # reference distributions.pyx:21
    raise EngineUnavailable(f'HIP engine failure: {text}')
