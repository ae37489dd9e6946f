"""Seed handling and synthetic-input generators.

Mirrors the public helpers of the reference's ``occuspytial/utils.py``:

* :func:`get_generator`      -- reference ``utils.py:7-35`` (SFC64-backed numpy Generator).
* :func:`rand_precision_mat` -- reference ``utils.py:38-103``; the reference delegates the lattice
  adjacency to ``libpysal.weights.lat2SW``; here the queen/rook lattice is built directly.
* :func:`make_data`          -- reference ``utils.py:106-294`` (same argument checks and error
  messages, same sampling distributions).
* :func:`make_lattice_problem` / :func:`make_graph_problem` -- the benchmark inputs described in
  SURVEY.md section 8(d) (eta = 0 in the generator, so no dense pseudo-inverse is needed and the
  generator scales to 500x500).

Everything here runs on the host with numpy/scipy and is used for set-up only.
"""
import numpy as np
from scipy import sparse


def get_generator(random_state=None):
    """Return ``numpy.random.Generator(SFC64(random_state))`` (reference ``utils.py:34-35``)."""
    bitgenerator = np.random.SFC64(random_state)
    return np.random.default_rng(bitgenerator)


def lattice_adjacency(lat_row, lat_col, criterion='queen'):
    """Binary adjacency (CSR) of a ``lat_row`` x ``lat_col`` lattice, sites numbered row-major.

    ``criterion='rook'`` joins the 4 edge-sharing neighbours, ``'queen'`` adds the 4 diagonal ones.
    """
    n = lat_row * lat_col
    idx = np.arange(n).reshape(lat_row, lat_col)
    pairs = [(idx[:, :-1], idx[:, 1:]), (idx[:-1, :], idx[1:, :])]
    if criterion == 'queen':
        pairs += [(idx[:-1, :-1], idx[1:, 1:]), (idx[:-1, 1:], idx[1:, :-1])]
    elif criterion != 'rook':
        raise ValueError("criterion must be 'rook' or 'queen'")
    rows = np.concatenate([a.ravel() for a, _ in pairs] + [b.ravel() for _, b in pairs])
    cols = np.concatenate([b.ravel() for _, b in pairs] + [a.ravel() for a, _ in pairs])
    data = np.ones(rows.size, dtype=np.int64)
    return sparse.csr_matrix((data, (rows, cols)), shape=(n, n))


def rand_precision_mat(lat_row, lat_col, max_neighbors=8, rho=1):
    """Spatial (I)CAR precision matrix ``D - rho*W`` of a rectangular lattice (COO).

    Same contract as the reference (``utils.py:38-103``): ``max_neighbors`` in {4, 8}, integer
    entries when ``rho`` is an int, diagonal = neighbour counts, ``rho=1`` gives the singular ICAR
    precision.
    """
    if max_neighbors == 8:
        nn = 'queen'
    elif max_neighbors == 4:
        nn = 'rook'
    else:
        raise ValueError('Maximum number of neighbors should be one of {4, 8}')
    W = lattice_adjacency(lat_row, lat_col, nn).tocoo()
    D = np.asarray(W.sum(axis=1)).ravel()
    out = sparse.coo_matrix(
        (np.concatenate([-W.data * rho, D]),
         (np.concatenate([W.row, np.arange(D.size)]), np.concatenate([W.col, np.arange(D.size)]))),
        shape=W.shape,
    )
    return out


def _expit(x):
    return np.exp(-np.logaddexp(0, -x))


def _check_make_data_args(n, min_v, max_v, ns):
    """Argument rules of the reference generator (``utils.py:239-259``); same messages."""
    rules = (
        (n < 150, 'n cant be lower than 150'),
        (min_v is not None and min_v < 1, 'min_v needs to be at least 1'),
        (max_v is not None and max_v < 2, 'max_v is too small'),
        (max_v is not None and max_v > n, 'max_v cant be more than n'),
        (ns is not None and ns == 0, 'ns should be positive'),
        (ns is not None and ns > n, 'ns cant be more than n'),
    )
    for failed, message in rules:
        if failed:
            raise ValueError(message)
    return (2 if min_v is None else min_v,
            n // 10 if max_v is None else max_v,
            n // 2 if ns is None else ns)


def make_data(n=150, min_v=None, max_v=None, ns=None, p=3, q=3, tau_range=(0.25, 1.5),
              max_neighbors=8, random_state=None):
    """Random occupancy-survey data with an ICAR spatial effect (reference ``utils.py:106-294``).

    Returns ``Q, W, X, y, alpha, beta, tau, z`` like the reference generator, drawing from the
    SFC64 stream in the reference's order (sites, visit counts, alpha, beta, tau, lattice shape,
    eta, X, z, then per surveyed site W_i and y_i) so that a ``random_state`` picks the same
    surveyed sites, visit counts and coefficients.  ``eta ~ N(0, pinv(Q)/tau)`` goes through a
    dense eigendecomposition, so this generator is for small ``n`` only; benchmarks use
    :func:`make_lattice_problem`.
    """
    from scipy.linalg import pinvh

    min_v, max_v, ns = _check_make_data_args(n, min_v, max_v, ns)
    rng = get_generator(random_state)
    sites = rng.choice(range(n), size=ns, replace=False)
    visits = rng.integers(min_v, max_v, size=ns, endpoint=True)
    alpha, beta = rng.standard_normal(q), rng.standard_normal(p)
    tau = rng.uniform(*tau_range)

    lat_row = rng.choice([d for d in range(3, n) if n % d == 0])
    Q = rand_precision_mat(lat_row, n // lat_row, max_neighbors=max_neighbors).astype(float)
    try:
        cov = pinvh(Q.toarray(), rtol=1e-5) / tau
    except TypeError:  # scipy < 1.7 spells the cutoff ``cond`` (as reference utils.py:277 does)
        cov = pinvh(Q.toarray(), cond=1e-5) / tau
    eta = rng.multivariate_normal(np.zeros(n), cov, method='eigh')

    X = rng.uniform(-2, 2, n * p).reshape(n, -1)
    X[:, 0] = 1
    z = rng.binomial(1, p=_expit(X @ beta - eta), size=n)  # sign quirk of utils.py:283 kept

    W, y = {}, {}
    for site, v in zip(sites, visits):
        Wi = rng.uniform(-2, 2, size=v * q).reshape(v, -1)
        Wi[:, 0] = 1
        W[site] = Wi
        y[site] = rng.binomial(1, z[site] * _expit(Wi @ alpha))
    return Q, W, X, y, alpha, beta, tau, z


def _survey_from_design(rng, Q, n, visits, p, q, surveyed=None):
    """Common tail of the benchmark generators (distributions of reference utils.py:264-292)."""
    alpha = rng.standard_normal(q)
    beta = rng.standard_normal(p)
    X = rng.uniform(-2, 2, n * p).reshape(n, -1)
    X[:, 0] = 1
    z = rng.binomial(1, p=_expit(X @ beta), size=n)
    if surveyed is None:
        surveyed = np.arange(n)
    visits = np.broadcast_to(np.asarray(visits), (len(surveyed),))
    W, y = {}, {}
    for i, j in zip(surveyed, visits):
        i, j = int(i), int(j)
        _W = rng.uniform(-2, 2, size=j * q).reshape(j, -1)
        _W[:, 0] = 1
        W[i] = _W
        y[i] = rng.binomial(1, z[i] * _expit(_W @ alpha))
    return Q, W, X, y, alpha, beta, z


def make_lattice_problem(lat_row, lat_col, visits=5, p=2, q=2, max_neighbors=8, random_state=0):
    """Benchmark input of SURVEY.md 8(d): lattice ICAR, every site surveyed ``visits`` times.

    Returns ``Q (csr), W, X, y, alpha, beta, z``.
    """
    rng = get_generator(random_state)
    Q = rand_precision_mat(lat_row, lat_col, max_neighbors=max_neighbors).astype(float).tocsr()
    return _survey_from_design(rng, Q, lat_row * lat_col, visits, p, q)


def make_graph_problem(n=3000, k=6, visits=10, p=2, q=2, random_state=0):
    """Irregular areal adjacency (config 5): symmetrised k-nearest-neighbour graph of uniform points.

    The graph is made connected by linking consecutive points along a space-filling sort when
    the k-NN graph alone is not; mean degree is a little above ``k``.  Returns the same tuple as
    :func:`make_lattice_problem`.
    """
    from scipy.sparse.csgraph import connected_components
    from scipy.spatial import cKDTree

    rng = get_generator(random_state)
    pts = rng.uniform(0, 1, size=(n, 2))
    kk = max(2, k // 2 + 1)
    _, nbr = cKDTree(pts).query(pts, k=kk + 1)
    rows = np.repeat(np.arange(n), kk)
    cols = nbr[:, 1:].ravel()
    A = sparse.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(n, n))
    A = ((A + A.T) > 0).astype(float)
    ncomp, lab = connected_components(A, directed=False)
    if ncomp > 1:
        order = np.lexsort((pts[:, 1], pts[:, 0]))
        chain = sparse.csr_matrix((np.ones(n - 1), (order[:-1], order[1:])), shape=(n, n))
        A = ((A + chain + chain.T) > 0).astype(float)
    A = A.tocsr()
    D = np.asarray(A.sum(axis=1)).ravel()
    Q = (sparse.diags(D) - A).tocsr()
    return _survey_from_design(rng, Q, n, visits, p, q)
