"""Posterior container (API of the reference's ``occuspytial/posterior.py:31-104``).

The reference converts the stacked chains to an ``arviz`` ``InferenceData`` and forwards
``summary`` / ``plot_*`` to arviz.  ``arviz`` is an optional dependency here: when it is importable the
behaviour is the reference's; when it is not, ``data`` is a minimal name -> array mapping with the
same ``data[name].data`` access, ``summary`` is computed by :mod:`occuspytial_amd.diagnostics`, and the
plotting methods raise ``ImportError``.
"""
import numpy as np

try:  # pragma: no cover - depends on the environment
    import arviz as az
except Exception:  # arviz absent (or broken): numpy-only fallback
    az = None


class _Variable:
    """Stand-in for an xarray DataArray: ``.data`` / ``.values`` are ``(chain, draw[, dim])`` arrays."""

    def __init__(self, array):
        self.data = array
        self.values = array
        self.shape = array.shape


class _Dataset(dict):
    """Stand-in for ``InferenceData.posterior`` when arviz is not installed."""

    @property
    def data_vars(self):
        return list(self)


class PosteriorParameter:
    """Posterior samples of ``alpha``, ``beta``, ``tau`` from one or more chains.

    ``PosteriorParameter(*chains)`` takes :class:`~occuspytial_amd.chain.Chain` objects;
    ``post['alpha']`` is an ndarray ``(chains, draws, q)``, ``post['tau']`` is ``(chains, draws)``.
    """

    def __init__(self, *chains):
        self.data = self._create_inference_data(chains)

    @staticmethod
    def _stack(chains):
        return {name: np.stack([c[name] for c in chains]) for name in chains[0]._names}

    def _create_inference_data(self, chains):
        stacked = self._stack(chains)
        if az is not None:
            return az.convert_to_inference_data(stacked).posterior
        return _Dataset((name, _Variable(arr)) for name, arr in stacked.items())

    @property
    def summary(self):
        """mean, sd, hdi_3%, hdi_97%, mcse_mean, mcse_sd, ess_bulk, ess_tail, r_hat per parameter."""
        if az is not None:
            return az.summary(self.data)
        from .diagnostics import summary
        return summary({name: self[name] for name in self.data})

    def _plot(self, fn_name, **kwargs):
        if az is None:
            raise ImportError(f'arviz is required for {fn_name}')
        return getattr(az, fn_name)(self.data, **kwargs)

    def plot_trace(self, **kwargs):
        """``arviz.plot_trace`` of the posterior."""
        return self._plot('plot_trace', **kwargs)

    def plot_auto_corr(self, **kwargs):
        """``arviz.plot_autocorr`` of the posterior."""
        return self._plot('plot_autocorr', **kwargs)

    def plot_pair(self, **kwargs):
        """``arviz.plot_pair`` of the posterior."""
        return self._plot('plot_pair', **kwargs)

    def plot_density(self, **kwargs):
        """``arviz.plot_posterior`` of the posterior."""
        return self._plot('plot_posterior', **kwargs)

    def plot_ess(self, **kwargs):
        """``arviz.plot_ess`` of the posterior."""
        return self._plot('plot_ess', **kwargs)

    def __getitem__(self, name):
        return self.data[name].data
