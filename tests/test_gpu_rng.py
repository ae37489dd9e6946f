"""Known-answer and distributional tests of the DEVICE variate generators, through the C ABI (``occ_draw``).

The Polya-Gamma sampler stands where the reference calls the third-party ``polyagamma`` package
(``logit.py:191-193, 202-204``), which is absent from the reference tree and from the image, and the
reference holds no known-answer test at that boundary: PG parity is UNPINNED by the reference.  What pins the
device sampler instead: (1) the closed-form mean, variance and Laplace transform of PG(1, z) (Polson, Scott &
Windle 2013) on a million device draws per z; (2) a KS test against the defining infinite series; (3) agreement
with the CPU oracle's independent implementation of the same stream specification, draw by draw; (4) the draws the
sampler's own omega_b kernel makes equal ``occ_draw``'s, so (1)-(3) speak for the kernels of the iteration.
"""
import numpy as np
import pytest
from scipy import stats

pytestmark = pytest.mark.gpu

STREAM_OMEGA_B, STREAM_TAU, STREAM_ETA_SITE, STREAM_Z = 1, 2, 3, 8


def _pg_mean(z):
    return 0.25 if abs(z) < 1e-8 else np.tanh(z / 2) / (2 * z)


def _pg_var(z):
    return 1 / 24 if abs(z) < 1e-3 else (np.sinh(z) - z) / (4 * z ** 3 * np.cosh(z / 2) ** 2)


@pytest.mark.parametrize('z', [0.0, 0.3, 1.0, 1.5, 1.5625, 2.5, 5.0, 12.0, -3.0, 40.0])
def test_device_pg1_moments_and_laplace_transform(z):
    """tests/test_oracle_rng.py's theory checks on a million DEVICE draws (5x the oracle's sample: tighter)."""
    from occuspytial_amd._engine import device_draw
    N = 1_000_000
    x = device_draw('pg1', np.full(N, z), key=1234 + int(abs(z) * 16), it=7, stream=STREAM_OMEGA_B)
    assert np.all(x > 0) and np.all(np.isfinite(x))
    m, v = _pg_mean(z), _pg_var(z)
    assert abs(x.mean() - m) < 5 * np.sqrt(v / N)
    assert abs(x.var() - v) < 0.015 * v
    for t in (0.5, 2.0, 10.0):
        lt = np.cosh(z / 2) / np.cosh(np.sqrt((z * z / 2 + t) / 2))
        e = np.exp(-t * x)
        assert abs(e.mean() - lt) < 5 * e.std() / np.sqrt(N)


def test_device_pg1_ks_against_the_defining_series():
    from occuspytial_amd._engine import device_draw
    rng = np.random.default_rng(0)
    k = np.arange(1, 801) - 0.5
    for z in (0.0, 2.0, 6.0):
        x = device_draw('pg1', np.full(40000, z), key=5, it=1)
        g = rng.standard_exponential((40000, 800))
        ref = (g / (k ** 2 + (z / (2 * np.pi)) ** 2)).sum(axis=1) / (2 * np.pi ** 2)
        assert stats.ks_2samp(x, ref).pvalue > 1e-3


def test_device_draws_equal_the_oracle_draw_by_draw(oracle):
    """Same stream specification, two independent implementations (HIP device functions / C): PG(1, z) over a grid
    of z with every branch of the sampler (tail / truncated inverse Gaussian below and above 1/t, |z| up to 60),
    gamma variates over the shapes the tau conditional meets, normals, uniforms."""
    from occuspytial_amd._engine import device_draw
    z = np.concatenate([np.linspace(-8, 8, 4001), np.array([0.0, 1e-12, 25.0, -25.0, 60.0]), np.random.default_rng(1).normal(0, 3, 20000)])
    dev = device_draw('pg1', z, key=77, it=3, stream=STREAM_OMEGA_B)
    ref = oracle.pg1(z, key=77, it=3, stream=STREAM_OMEGA_B)
    assert np.abs(dev / ref - 1).max() < 1e-10            # same accept/reject path everywhere, libm-level differences
    L = oracle.lib()
    for shape in (0.3, 0.5, 1.0, 75.5, 5000.5):
        dev = device_draw('std_gamma', np.full(512, shape), key=9, it=4, stream=STREAM_TAU)
        # the oracle's gamma cursor sits at index 0 of the sub-stream: element 0 is the draw the tau kernel makes
        assert abs(dev[0] / L.orc_std_gamma_draw(9, 4, STREAM_TAU, shape) - 1) < 1e-12
    dn = device_draw('normal', n=4096, key=3, it=0, stream=STREAM_ETA_SITE)
    du = device_draw('uniform', n=4096, key=3, it=0, stream=STREAM_Z)
    rn = np.array([L.orc_block_normal(3, i, 0, 0, STREAM_ETA_SITE) for i in range(4096)])
    ru = np.array([L.orc_block_uniform(3, i, 0, 0, STREAM_Z) for i in range(4096)])
    assert np.abs(dn - rn).max() < 1e-13 and np.array_equal(du, ru)


@pytest.mark.parametrize('shape', [0.3, 1.0, 5.0, 75.0, 5000.0])
def test_device_gamma_distribution(shape):
    from occuspytial_amd._engine import device_draw
    x = device_draw('std_gamma', np.full(200_000, shape), key=11, it=2, stream=STREAM_TAU)
    assert stats.kstest(x, 'gamma', args=(shape,)).pvalue > 1e-3


def test_device_normal_and_uniform_distribution():
    from occuspytial_amd._engine import device_draw
    x = device_draw('normal', n=500_000, key=3, it=1, stream=STREAM_ETA_SITE)
    assert stats.kstest(x, 'norm').pvalue > 1e-3
    u = device_draw('uniform', n=500_000, key=3, it=1, stream=STREAM_Z)
    assert stats.kstest(u, 'uniform').pvalue > 1e-3 and u.min() > 0.0 and u.max() < 1.0


def test_the_samplers_omega_b_kernel_draws_what_occ_draw_draws():
    """beta = 0 and eta = a grid of z put z_i = x_i'beta + eta_i = eta_i in front of the omega_b kernel
    (logit.py:195-204): the omega_b it produces equals occ_draw's PG(1, z_i) from the same sub-streams BIT FOR BIT
    (so the theory tests above are tests of the iteration's own draws), and its moments match PG(1, z) theory."""
    from occuspytial_amd._engine import Engine, device_draw
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=2, p=2, q=2, random_state=3)
    prob = FlatProblem(Q, W, X, y)
    key = 0xC0FFEE1234567
    eng = Engine(prob, [key])
    grid = np.tile(np.array([0.0, 0.7, -1.9, 3.2, 6.5]), prob.n // 5)
    eng.set_start(0, np.zeros(2), np.zeros(2), 1.0, grid)
    it = int(eng.get('iter'))
    eng.step()                                            # its prologue draws omega_b(it) from the state just set
    om = eng.get('omega_b')
    assert np.array_equal(om, device_draw('pg1', grid, key=key, it=it, stream=STREAM_OMEGA_B))
    for j, z in enumerate((0.0, 0.7, -1.9, 3.2, 6.5)):
        x = om[j::5]
        assert abs(x.mean() - _pg_mean(z)) < 5 * np.sqrt(_pg_var(z) / x.size)
    eng.close()


def test_device_pg1_leaves_its_loops_on_arguments_no_chain_should_produce():
    """A wave that never finishes hangs the device.  The sampler's rejection loops end with probability one for any sane
    argument; for NaN, inf and for finite arguments past |z| ~ 1e100 -- where the alternating series' coefficients overflow to
    inf x 0 = NaN and every comparison is false -- pg1_draw returns NaN at once (occ_rng.hpp), which the Cholesky
    factorisation downstream reports.  Large sane arguments still draw."""
    from occuspytial_amd._engine import device_draw
    z = np.array([np.nan, np.inf, -np.inf, 1e101, -3e250, 1e99, 1e6, -4e3, 50.0, 0.0] * 32)
    out = device_draw('pg1', z, key=3, it=1)
    bad = ~np.isfinite(z) | (np.abs(z) >= 2e100)
    assert np.isnan(out[bad]).all()
    assert np.isfinite(out[~bad]).all() and (out[~bad] > 0).all()
    big = (~bad) & (np.abs(z) >= 4e3)
    assert np.allclose(out[big] * 2 * np.abs(z[big]), 1.0, rtol=0.2)   # PG(1, z) concentrates at 1 / (2 |z|)


def test_wave_sum_forms_agree_bit_for_bit():
    """The engine adds four quantities over a wave in one fixed order (wave_sum: DPP levels); the form k_iter and the
    per-slice readers use transposes the quantities over the lanes of a quad and moves one value instead of four.
    It must return the bits of the plain form and of four separate wave sums, whatever the data (magnitudes spread
    over 60 binary orders, mixed signs, zeros)."""
    from occuspytial_amd._engine import device_draw
    rng = np.random.default_rng(7)
    x = rng.standard_normal(64 * 512) * np.exp2(rng.integers(-30, 30, 64 * 512))
    x[rng.random(x.size) < 0.05] = 0.0
    out = device_draw('wave_sum_check', x)
    assert not np.isnan(out).any()
    # and the value is the sum it claims to be (quantity q of wave w: lanes with i % 4 == q report it)
    w = x.reshape(-1, 64)
    for q in range(4):
        ref = (w * (1.0 + 0.37 * q) + (w * w if q == 3 else 0.0)).sum(axis=1)
        got = out.reshape(-1, 64)[:, q]
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-9 * np.abs(w).max())
