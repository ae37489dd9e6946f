"""Posterior diagnostics in numpy (what the reference obtains from ``arviz.summary``,
``occuspytial/posterior.py:63-76``): mean, sd, 94 % HDI, MCSE, bulk/tail ESS and split R-hat, following
Vehtari, Gelman, Simpson, Carpenter & Buerkner (2021).  Also the distributional-parity tool of the
test-suite (posterior means within a few MCSE, split R-hat across reference/build chains).
"""
import numpy as np
from scipy import stats


def _split(x):
    """(chains, draws) -> (2*chains, draws//2)."""
    c, d = x.shape
    h = d // 2
    return np.concatenate([x[:, :h], x[:, d - h:]], axis=0)


def _rank_normalise(x):
    r = stats.rankdata(x.ravel(), method='average').reshape(x.shape)
    return stats.norm.ppf((r - 0.375) / (x.size + 0.25))


def _autocov(x):
    n = x.shape[-1]
    m = 1 << int(np.ceil(np.log2(2 * n)))
    xc = x - x.mean(axis=-1, keepdims=True)
    f = np.fft.rfft(xc, m, axis=-1)
    return np.fft.irfft(f * np.conj(f), m, axis=-1)[..., :n] / n


def rhat(x):
    """Split R-hat of a (chains, draws) array (rank-normalised, max of bulk and folded)."""
    def basic(y):
        y = _split(y)
        n = y.shape[1]
        w = y.var(axis=1, ddof=1).mean()
        b = n * y.mean(axis=1).var(ddof=1)
        return np.sqrt(((n - 1) / n * w + b / n) / w) if w > 0 else np.nan
    x = np.asarray(x, dtype=float)
    return max(basic(_rank_normalise(x)), basic(_rank_normalise(np.abs(x - np.median(x)))))


def ess(x):
    """Effective sample size of a (chains, draws) array (Geyer's initial monotone sequence)."""
    x = _split(np.asarray(x, dtype=float))
    m, n = x.shape
    if n < 4:
        return float(m * n)
    acov = _autocov(x)
    w = (acov[:, 0] * n / (n - 1)).mean()
    var_plus = w * (n - 1) / n + (x.mean(axis=1).var(ddof=1) if m > 1 else 0.0)
    if not var_plus > 0:
        return float(m * n)
    rho = 1.0 - (w - acov.mean(axis=0)) / var_plus
    rho[0] = 1.0
    t, tau = 1, -1.0
    pair_prev = np.inf
    while t + 1 < n:
        pair = rho[t - 1] + rho[t] if t > 1 else rho[0] + rho[1]
        if t == 1:
            pair = rho[0] + rho[1]
        if pair < 0:
            break
        pair = min(pair, pair_prev)
        tau += 2 * pair
        pair_prev = pair
        t += 2
    tau = max(tau, 1.0 / np.log10(m * n))
    return float(m * n / tau)


def ess_bulk(x):
    return ess(_rank_normalise(np.asarray(x, dtype=float)))


def ess_tail(x):
    x = np.asarray(x, dtype=float)
    lo, hi = np.quantile(x, [0.05, 0.95])
    return min(ess((x <= lo).astype(float)), ess((x <= hi).astype(float)))


def hdi(x, prob=0.94):
    s = np.sort(np.asarray(x, dtype=float).ravel())
    k = int(np.floor(prob * s.size))
    widths = s[k:] - s[:s.size - k]
    i = int(np.argmin(widths))
    return s[i], s[i + k]


def mcse_mean(x):
    x = np.asarray(x, dtype=float)
    return x.std(ddof=1) / np.sqrt(ess(x))


def summarise(x):
    """Row of diagnostics for one scalar quantity given as (chains, draws)."""
    x = np.asarray(x, dtype=float)
    lo, hi = hdi(x)
    sd = x.std(ddof=1)
    e = ess(x)
    e_sd = ess((x - x.mean()) ** 2)
    row = {
        'mean': x.mean(), 'sd': sd, 'hdi_3%': lo, 'hdi_97%': hi,
        'mcse_mean': sd / np.sqrt(e), 'mcse_sd': sd * np.sqrt(np.exp(1) * (1 - 1 / e_sd) ** (e_sd - 1) - 1) if e_sd > 1 else np.nan,
        'ess_bulk': ess_bulk(x), 'ess_tail': ess_tail(x),
        'r_hat': rhat(x) if x.shape[0] > 1 else np.nan,
    }
    return row


def summary(arrays):
    """Table of diagnostics: ``arrays`` maps name -> (chains, draws[, dim]); returns a pandas
    DataFrame when pandas is importable, else a dict of rows."""
    rows = {}
    for name, a in arrays.items():
        a = np.asarray(a)
        if a.ndim == 2:
            rows[name] = summarise(a)
        else:
            for j in range(a.shape[2]):
                rows[f'{name}[{j}]'] = summarise(a[:, :, j])
    try:
        import pandas as pd
        return pd.DataFrame.from_dict(rows, orient='index')
    except Exception:
        return rows
