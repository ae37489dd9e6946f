/*
 * occ_oracle_abi.c -- the CPU restatement behind the IDENTICAL C ABI (include/occ_gibbs.h), SURVEY 8(b).
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/.  It builds into oracle/liboccoracle_abi.so and is loaded by
 * tests that point occuspytial_amd._lib at it explicitly (tests/test_cpu_abi.py): the SAME ctypes binding, Engine /
 * EngineGroup wrappers and sampler classes that drive the HIP library then drive the oracle, so the host side of the
 * product (chain fan-out, start values, chunked runs, checkpoints, groups, the per-conditional entry points fed with
 * the reference's fixtures) is exercised on a machine without a GPU.  The product never loads this file: _lib.LIB_PATH
 * names occuspytial_amd/libocc_gibbs.so and there is no search path, no environment switch and no fallback.
 *
 * Every occ_* entry point of the header is exported.  A batch of chains is an array of one-chain oracle samplers; the
 * variate streams are the specification the kernels implement, so results agree with the device to
 * the tolerances of tests/test_gpu_parity.py.  Entry points that only make sense on a device (occ_profile, a
 * communicator of more than one rank) return OCC_E_HIP with a message.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/occ_gibbs.h"
#include "occ_oracle.h"

struct occ_sampler {
    int C, p, q, rsr_dim;
    long n, S, R;
    orc_sampler **ch;
    int64_t *indptr, *indices, *site_id, *site_ptr;
    double *qdata, *X, *W, *yrow, *a_prec, *b_prec, *a_pbm, *b_pbm, *prior_F;
    uint8_t *obs_site, *surveyed;
    double tau_rate, tau_shape;
    int64_t steps, runs;
    char err[256];
};
struct occ_comm { int world, rank; char err[128]; };

static char g_err[256];
static int fail(occ_sampler *s, int code, const char *msg)
{
    snprintf(s ? s->err : g_err, sizeof(g_err), "%s", msg);
    return code;
}
static void *dup(const void *p, size_t bytes)
{
    void *d = malloc(bytes ? bytes : 1);
    if (bytes) memcpy(d, p, bytes);
    return d;
}

int32_t occ_abi_version(void) { return OCC_ABI_VERSION; }
int32_t occ_device_count(void) { return 0; }
const char *occ_last_error(const occ_sampler *s) { return s ? s->err : g_err; }

int occ_destroy(occ_sampler *s)
{
    if (!s) return OCC_OK;
    for (int c = 0; s->ch && c < s->C; ++c)
        if (s->ch[c]) orc_destroy(s->ch[c]);
    free(s->ch); free(s->indptr); free(s->indices); free(s->site_id); free(s->site_ptr); free(s->qdata); free(s->X);
    free(s->W); free(s->yrow); free(s->a_prec); free(s->b_prec); free(s->a_pbm); free(s->b_pbm); free(s->prior_F);
    free(s->obs_site); free(s->surveyed);
    free(s);
    return OCC_OK;
}

/* the checks of the engine's build_layout (occ_gibbs.hip), same codes and messages */
static int check_problem(const occ_problem *pb)
{
    if (!pb) return fail(NULL, OCC_E_BADARG, "null problem");
    if (pb->n < 1 || pb->n > 0x7fffffff || pb->n_rows > 0x7fffffff || pb->n_surveyed > pb->n) return fail(NULL, OCC_E_BADARG, "problem sizes out of range");
    if (pb->p < 1 || pb->p > OCC_MAX_COVARIATES || pb->q < 1 || pb->q > OCC_MAX_COVARIATES) return fail(NULL, OCC_E_BADARG, "p and q must lie in [1, 32]");
    if (!(pb->tau_rate > 0.0) || !(pb->tau_shape > 0.0)) return fail(NULL, OCC_E_BADARG, "tau_rate and tau_shape must be positive");
    const long n = (long)pb->n;
    if (pb->q_indptr[0] != 0 || pb->q_indptr[n] < n) return fail(NULL, OCC_E_BADARG, "malformed Q indptr");
    double scale = 0.0;
    for (long i = 0; i < n; ++i) {
        double rowsum = 0.0, rowabs = 0.0;
        long last = -1;
        for (int32_t k = pb->q_indptr[i]; k < pb->q_indptr[i + 1]; ++k) {
            const long j = pb->q_indices[k];
            if (j < 0 || j >= n || j <= last) return fail(NULL, OCC_E_BADARG, "Q columns must be sorted, unique and in range");
            last = j;
            rowsum += pb->q_data[k];
            rowabs += fabs(pb->q_data[k]);
            if (j != i && pb->q_data[k] > 0.0 && !pb->prior_factor)
                return fail(NULL, OCC_E_BADARG, "Q must have non-positive off-diagonal entries (or come with a prior factor: occ_problem::prior_factor)");
        }
        if (rowabs > scale) scale = rowabs;
        if (!pb->prior_factor && fabs(rowsum) > 1e-10 * (rowabs > 1e-300 ? rowabs : 1e-300)) return fail(NULL, OCC_E_BADARG, "Spatial precision matrix Q must be singular.");
    }
    if (!(scale > 0.0)) return fail(NULL, OCC_E_BADARG, "Spatial precision matrix Q must be singular.");
    if (pb->prior_factor && pb->prior_factor_cols != pb->n - 1) return fail(NULL, OCC_E_BADARG, "the CPU restatement takes the reference's n x (n - 1) eigenfactor as prior factor");
    if (pb->rsr_dim < 0 || (pb->rsr_dim > 0 && (!pb->rsr_K || !pb->rsr_Q || !pb->rsr_E))) return fail(NULL, OCC_E_BADARG, "bad reduced-rank basis");
    return OCC_OK;
}

int occ_create(const occ_problem *pb, int32_t n_chains, const uint64_t *keys, int32_t device, occ_sampler **out)
{
    (void)device;
    if (!out) return OCC_E_BADARG;
    *out = NULL;
    if (!keys || n_chains < 1) return fail(NULL, OCC_E_BADARG, "bad keys / n_chains");
    int rc = check_problem(pb);
    if (rc) return rc;
    occ_sampler *s = (occ_sampler *)calloc(1, sizeof(*s));
    const long n = (long)pb->n, S = (long)pb->n_surveyed, R = (long)pb->n_rows, nnz = pb->q_indptr[n];
    s->C = n_chains; s->n = n; s->S = S; s->R = R; s->p = pb->p; s->q = pb->q; s->rsr_dim = pb->rsr_dim;
    s->tau_rate = pb->tau_rate; s->tau_shape = pb->tau_shape;
    s->indptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    s->indices = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz ? nnz : 1));
    for (long i = 0; i <= n; ++i) s->indptr[i] = pb->q_indptr[i];
    for (long k = 0; k < nnz; ++k) s->indices[k] = pb->q_indices[k];
    s->qdata = (double *)dup(pb->q_data, sizeof(double) * (size_t)nnz);
    s->X = (double *)dup(pb->X, sizeof(double) * (size_t)(n * s->p));
    s->W = (double *)dup(pb->W, sizeof(double) * (size_t)(R * s->q));
    s->yrow = (double *)dup(pb->y, sizeof(double) * (size_t)R);
    s->site_id = (int64_t *)malloc(sizeof(int64_t) * (size_t)(S ? S : 1));
    s->site_ptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(S + 1));
    for (long t = 0; t < S; ++t) s->site_id[t] = pb->site_id[t];
    for (long t = 0; t <= S; ++t) s->site_ptr[t] = pb->site_ptr[t];
    s->a_prec = (double *)dup(pb->a_prec, sizeof(double) * (size_t)(s->q * s->q));
    s->b_prec = (double *)dup(pb->b_prec, sizeof(double) * (size_t)(s->p * s->p));
    s->a_pbm = (double *)calloc((size_t)s->q, sizeof(double));
    s->b_pbm = (double *)calloc((size_t)s->p, sizeof(double));
    for (int a = 0; a < s->q; ++a) for (int b = 0; b < s->q; ++b) s->a_pbm[a] += pb->a_prec[a * s->q + b] * pb->a_mu[b];
    for (int a = 0; a < s->p; ++a) for (int b = 0; b < s->p; ++b) s->b_pbm[a] += pb->b_prec[a * s->p + b] * pb->b_mu[b];
    s->obs_site = (uint8_t *)calloc((size_t)(S ? S : 1), 1);
    s->surveyed = (uint8_t *)calloc((size_t)n, 1);
    for (long t = 0; t < S; ++t) {
        if (s->site_id[t] < 0 || s->site_id[t] >= n || s->surveyed[s->site_id[t]]) { occ_destroy(s); return fail(NULL, OCC_E_BADARG, "site_id entries must be unique and in [0, n)"); }
        s->surveyed[s->site_id[t]] = 1;
        for (int64_t r = s->site_ptr[t]; r < s->site_ptr[t + 1]; ++r) s->obs_site[t] |= (uint8_t)(s->yrow[r] != 0.0);
    }
    if (pb->prior_factor) s->prior_F = (double *)dup(pb->prior_factor, sizeof(double) * (size_t)(n * (n - 1)));
    s->ch = (orc_sampler **)calloc((size_t)n_chains, sizeof(*s->ch));
    for (int c = 0; c < n_chains; ++c) {
        s->ch[c] = orc_create(n, s->p, s->q, S, s->indptr, s->indices, s->qdata, s->X, s->site_id, s->site_ptr, s->W, s->yrow,
                              pb->a_mu, pb->a_prec, pb->b_mu, pb->b_prec, pb->tau_rate, pb->tau_shape, keys[c]);
        if (pb->rsr_dim > 0 && orc_set_rsr(s->ch[c], pb->rsr_dim, pb->rsr_K, pb->rsr_Q, pb->rsr_E)) { occ_destroy(s); return fail(NULL, OCC_E_BADARG, "bad reduced-rank basis"); }
        if (s->prior_F) orc_set_dense_eigen(s->ch[c], s->prior_F);
    }
    *out = s;
    return OCC_OK;
}

int occ_set_keys(occ_sampler *s, const uint64_t *keys)
{
    if (!s || !keys) return OCC_E_BADARG;
    for (int c = 0; c < s->C; ++c) orc_set_key(s->ch[c], keys[c]);
    return OCC_OK;
}

int occ_set_start(occ_sampler *s, int32_t chain, const double *alpha, const double *beta, double tau, const double *eta)
{
    if (!s) return OCC_E_BADARG;
    if (chain < 0 || chain >= s->C || !alpha || !beta || !eta) return fail(s, OCC_E_BADARG, "bad chain / null start pointer");
    orc_sampler *o = s->ch[chain];
    const double zero = 0.0;
    if (s->rsr_dim > 0) {  /* `eta` holds theta; the spatial effects follow (logit.py:457-460) */
        double *e0 = (double *)calloc((size_t)s->n, sizeof(double));
        orc_set_start(o, alpha, beta, tau, e0);
        free(e0);
        orc_set(o, "theta", eta, s->rsr_dim);
    } else {
        orc_set_start(o, alpha, beta, tau, eta);
    }
    double *x0 = (double *)calloc((size_t)(2 * s->n), sizeof(double));  /* x0 = None (logit.py:71) */
    orc_set(o, "xz", x0, 2 * s->n);
    free(x0);
    orc_set(o, "iter", &zero, 1);
    return OCC_OK;
}

static int map_err(occ_sampler *s, int e)
{
    if (e == ORC_ERR_MINRES) return fail(s, OCC_E_MINRES, "MINRES solver did not converge!");
    if (e == ORC_ERR_CHOLESKY) return fail(s, OCC_E_CHOLESKY, "Cholesky factorization/solver failed!");
    return e ? fail(s, OCC_E_HIP, "oracle error") : OCC_OK;
}

int occ_step(occ_sampler *s)
{
    if (!s) return OCC_E_BADARG;
    for (int c = 0; c < s->C; ++c) {
        const int e = orc_step(s->ch[c]);
        if (e) return map_err(s, e);
    }
    s->steps += 1;
    return OCC_OK;
}

int occ_run(occ_sampler *s, int64_t n_iter, int64_t burnin, double *out_alpha, double *out_beta, double *out_tau)
{
    if (!s) return OCC_E_BADARG;
    if (n_iter < 1 || burnin < 0 || burnin >= n_iter) return fail(s, OCC_E_BADARG, "burnin value cannot be larger than sample size");
    if (!out_alpha || !out_beta || !out_tau) return fail(s, OCC_E_BADARG, "null output buffer");
    const int64_t keep = n_iter - burnin;
    for (int c = 0; c < s->C; ++c) {
        const int e = orc_run(s->ch[c], (long)n_iter, (long)burnin, out_alpha + (size_t)c * keep * s->q, out_beta + (size_t)c * keep * s->p,
                              out_tau + (size_t)c * keep);
        if (e) return map_err(s, e);
    }
    s->runs += 1;
    return OCC_OK;
}

int occ_get_state(occ_sampler *s, int32_t chain, const char *name, double *out, int64_t cap, int64_t *len)
{
    if (!s || !name || !len) return OCC_E_BADARG;
    if (chain < 0 || chain >= s->C) return fail(s, OCC_E_BADARG, "bad chain index");
    orc_sampler *o = s->ch[chain];
    if (!strcmp(name, "exists")) {  /* the engine's meaning: from the CURRENT z (what the next omega_a update will use) */
        *len = s->S;
        if (out) {
            if (cap < s->S) return fail(s, OCC_E_STATE, "output buffer too small");
            double *z = (double *)malloc(sizeof(double) * (size_t)s->n);
            orc_get(o, "z", z, s->n);
            for (long t = 0; t < s->S; ++t) out[t] = (s->obs_site[t] || z[s->site_id[t]] != 0.0) ? 1.0 : 0.0;
            free(z);
        }
        return OCC_OK;
    }
    const long size = orc_get(o, name, NULL, 0);
    if (size < 0) return fail(s, OCC_E_STATE, "unknown state name");
    *len = size;
    if (out) {
        if (cap < size) return fail(s, OCC_E_STATE, "output buffer too small");
        orc_get(o, name, out, size);
    }
    return OCC_OK;
}

int occ_set_state(occ_sampler *s, int32_t chain, const char *name, const double *in, int64_t len)
{
    if (!s || !name || !in) return OCC_E_BADARG;
    if (chain < 0 || chain >= s->C) return fail(s, OCC_E_BADARG, "bad chain index");
    if (strcmp(name, "alpha") && strcmp(name, "beta") && strcmp(name, "tau") && strcmp(name, "eta") && strcmp(name, "z") && strcmp(name, "omega_a") &&
        strcmp(name, "xz") && strcmp(name, "iter") && strcmp(name, "theta"))
        return fail(s, OCC_E_STATE, "unknown state name or wrong length");
    if (orc_set(s->ch[chain], name, in, (long)len)) return fail(s, OCC_E_STATE, "unknown state name or wrong length");
    return OCC_OK;
}

int occ_get_stats(occ_sampler *s, occ_stats *out)
{
    if (!s || !out) return OCC_E_BADARG;
    memset(out, 0, sizeof(*out));
    double v = 0.0;
    orc_get(s->ch[0], "iter", &v, 1);
    out->iterations = (int64_t)v;
    orc_get(s->ch[0], "minres_itn", &v, 1);
    out->krylov_last = (int32_t)v;
    out->eager_iterations = s->steps;
    out->n_chains = s->C;
    return OCC_OK;
}

int occ_profile(occ_sampler *s, int32_t reps, int64_t counts[OCC_N_KERNEL_KINDS], double total_us[OCC_N_KERNEL_KINDS])
{
    (void)reps; (void)counts; (void)total_us;
    return fail(s, OCC_E_HIP, "occ_profile times HIP kernels: not available in the CPU restatement");
}

int occ_synchronize(occ_sampler *s) { return s ? OCC_OK : OCC_E_BADARG; }
const char *occ_group_transport(const occ_sampler *s) { return s ? "cpu restatement (no device)" : ""; }

/* ---- per-conditional entry points with injected variates: the oracle's reference pieces --------------------------- */
static int chain_ok(occ_sampler *s, int chain)
{
    if (chain < 0 || chain >= s->C) return fail(s, OCC_E_BADARG, "bad chain index");
    if (s->rsr_dim > 0) return fail(s, OCC_E_BADARG, "the per-conditional entry points cover the ICAR model");
    return OCC_OK;
}
static double *state(occ_sampler *s, int chain, const char *name, long n)
{
    double *v = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    orc_get(s->ch[chain], name, v, n);
    return v;
}

int occ_cond_tau(occ_sampler *s, int32_t chain, double gamma_variate, double *tau_out)
{
    if (!s) return OCC_E_BADARG;
    int rc = chain_ok(s, chain);
    if (rc) return rc;
    double *eta = state(s, chain, "eta", s->n);
    const double rate = orc_tau_rate(s->n, s->indptr, s->indices, s->qdata, eta, s->tau_rate);
    const double tau = (1.0 / rate) * gamma_variate;  /* rng.gamma(shape, 1 / rate) = standard_gamma(shape) * (1 / rate) */
    free(eta);
    orc_set(s->ch[chain], "tau", &tau, 1);
    if (tau_out) *tau_out = tau;
    return OCC_OK;
}

int occ_cond_eta(occ_sampler *s, int32_t chain, const double *omega_b, const double *eps_site, const double *prior_term, double *rhs_out,
                 double *xz_out, double *eta_out, int32_t *itn_out)
{
    if (!s || !omega_b || !eps_site || !prior_term) return OCC_E_BADARG;
    int rc = chain_ok(s, chain);
    if (rc) return rc;
    const long n = s->n;
    double *beta = state(s, chain, "beta", s->p), *k = state(s, chain, "k", n), *xz = state(s, chain, "xz", 2 * n);
    double tau = 0.0;
    orc_get(s->ch[chain], "tau", &tau, 1);
    double *rhs = (double *)malloc(sizeof(double) * (size_t)n), *eta = (double *)malloc(sizeof(double) * (size_t)n);
    const double st = sqrt(tau);
    for (long i = 0; i < n; ++i) {  /* logit.py:213, 76-78, in the order orc_update_eta evaluates it */
        double xb = 0.0;
        for (int a = 0; a < s->p; ++a) xb += s->X[i * s->p + a] * beta[a];
        const double b = k[i] - omega_b[i] * xb;
        rhs[i] = (b + sqrt(omega_b[i]) * eps_site[i]) + st * prior_term[i];
    }
    long itn = 0;
    int istop = 0;
    const long info = orc_minres_joint(n, s->indptr, s->indices, s->qdata, omega_b, tau, rhs, xz, 1e-5, 5 * 2 * n, &itn, &istop);
    if (!info) {
        orc_ensure_sums_to_zero(n, xz, xz + n, eta);
        orc_set(s->ch[chain], "xz", xz, 2 * n);
        orc_set(s->ch[chain], "eta", eta, n);
        orc_set(s->ch[chain], "omega_b", omega_b, n);
        if (rhs_out) memcpy(rhs_out, rhs, sizeof(double) * (size_t)n);
        if (xz_out) memcpy(xz_out, xz, sizeof(double) * (size_t)(2 * n));
        if (eta_out) memcpy(eta_out, eta, sizeof(double) * (size_t)n);
        if (itn_out) *itn_out = (int32_t)itn;
    }
    free(beta); free(k); free(xz); free(rhs); free(eta);
    return info ? fail(s, OCC_E_MINRES, "MINRES solver did not converge!") : OCC_OK;
}

int occ_cond_beta(occ_sampler *s, int32_t chain, const double *omega_b, const double *eps, double *beta_out)
{
    if (!s || !omega_b || !eps) return OCC_E_BADARG;
    int rc = chain_ok(s, chain);
    if (rc) return rc;
    const int p = s->p;
    double *k = state(s, chain, "k", s->n), *eta = state(s, chain, "eta", s->n);
    double *A = (double *)malloc(sizeof(double) * (size_t)(p * p)), *r = (double *)malloc(sizeof(double) * (size_t)p), *out = (double *)malloc(sizeof(double) * (size_t)p);
    orc_beta_system(s->n, p, s->X, omega_b, k, eta, s->b_prec, s->b_pbm, A, r);
    const int bad = orc_precision_mvnorm(p, r, A, eps, out);
    if (!bad) {
        orc_set(s->ch[chain], "beta", out, p);
        if (beta_out) memcpy(beta_out, out, sizeof(double) * (size_t)p);
    }
    free(k); free(eta); free(A); free(r); free(out);
    return bad ? fail(s, OCC_E_CHOLESKY, "Cholesky factorization/solver failed!") : OCC_OK;
}

int occ_cond_alpha(occ_sampler *s, int32_t chain, const double *omega_a, const double *eps, double *alpha_out)
{
    if (!s || !omega_a || !eps) return OCC_E_BADARG;
    int rc = chain_ok(s, chain);
    if (rc) return rc;
    const int q = s->q;
    double *z = state(s, chain, "z", s->n);
    uint8_t *ex = (uint8_t *)malloc((size_t)(s->S ? s->S : 1));
    for (long t = 0; t < s->S; ++t) ex[t] = (uint8_t)(s->obs_site[t] || z[s->site_id[t]] != 0.0);
    double *A = (double *)malloc(sizeof(double) * (size_t)(q * q)), *r = (double *)malloc(sizeof(double) * (size_t)q), *out = (double *)malloc(sizeof(double) * (size_t)q);
    orc_alpha_system(s->S, q, s->site_ptr, ex, s->W, s->yrow, omega_a, s->a_prec, s->a_pbm, A, r);
    const int bad = orc_precision_mvnorm(q, r, A, eps, out);
    if (!bad) {
        orc_set(s->ch[chain], "alpha", out, q);
        orc_set(s->ch[chain], "omega_a", omega_a, s->R);
        if (alpha_out) memcpy(alpha_out, out, sizeof(double) * (size_t)q);
    }
    free(z); free(ex); free(A); free(r); free(out);
    return bad ? fail(s, OCC_E_CHOLESKY, "Cholesky factorization/solver failed!") : OCC_OK;
}

int occ_cond_z(occ_sampler *s, int32_t chain, const double *u, double *z_out)
{
    if (!s || !u) return OCC_E_BADARG;
    int rc = chain_ok(s, chain);
    if (rc) return rc;
    double *z = state(s, chain, "z", s->n), *eta = state(s, chain, "eta", s->n), *alpha = state(s, chain, "alpha", s->q), *beta = state(s, chain, "beta", s->p);
    for (long t = 0; t < s->S; ++t) {  /* logit.py:241-248 */
        if (s->obs_site[t]) continue;
        const long i = (long)s->site_id[t];
        const double pr = orc_z_prob(s->p, s->q, s->X + i * s->p, beta, eta[i], (long)(s->site_ptr[t + 1] - s->site_ptr[t]), s->W + s->site_ptr[t] * s->q, alpha);
        z[i] = (u[i] < pr) ? 1.0 : 0.0;
    }
    for (long i = 0; i < s->n; ++i) {  /* logit.py:249-251 */
        if (s->surveyed[i]) continue;
        double xb = 0.0;
        for (int a = 0; a < s->p; ++a) xb += s->X[i * s->p + a] * beta[a];
        z[i] = (u[i] < orc_expit(xb + eta[i])) ? 1.0 : 0.0;
    }
    orc_set(s->ch[chain], "z", z, s->n);
    if (z_out) memcpy(z_out, z, sizeof(double) * (size_t)s->n);
    free(z); free(eta); free(alpha); free(beta);
    return OCC_OK;
}

int occ_draw(int32_t device, int32_t kind, uint64_t key, uint32_t iteration, uint32_t stream, int64_t n, const double *param, double *out)
{
    (void)device;
    if (n < 0 || !out || kind < 0 || kind > 3 || ((kind == 0 || kind == 1) && !param && n > 0)) return fail(NULL, OCC_E_BADARG, "occ_draw: bad arguments");
    if (kind == 0) orc_pg1_array(key, iteration, stream, (long)n, param, out);
    for (int64_t i = 0; i < n && kind != 0; ++i)
        out[i] = kind == 1 ? orc_std_gamma_draw_at(key, (uint32_t)i, iteration, stream, param[i])
               : kind == 2 ? orc_block_normal(key, (uint32_t)i, 0, iteration, stream) : orc_block_uniform(key, (uint32_t)i, 0, iteration, stream);
    return OCC_OK;
}

/* ---- groups: one sampler per "device"; a communicator of one rank ------------------------------------------------- */
int occ_create_group(const occ_problem *problem, int32_t n_devices, const int32_t *devices, const int32_t *chains_per_device, const uint64_t *keys,
                     occ_sampler **out)
{
    if (!out || !devices || !chains_per_device || !keys || n_devices < 1) return OCC_E_BADARG;
    size_t koff = 0;
    for (int g = 0; g < n_devices; ++g) out[g] = NULL;
    for (int g = 0; g < n_devices; ++g) {
        const int rc = occ_create(problem, chains_per_device[g], keys + koff, devices[g], &out[g]);
        if (rc) {
            for (int h = 0; h < g; ++h) { occ_destroy(out[h]); out[h] = NULL; }
            return rc;
        }
        koff += (size_t)chains_per_device[g];
    }
    return OCC_OK;
}
int occ_comm_unique_id(uint8_t id[128]) { if (!id) return OCC_E_BADARG; memset(id, 0, 128); return OCC_OK; }
int occ_comm_create(int32_t world, int32_t rank, const uint8_t id[128], int32_t device, occ_comm **out)
{
    (void)id; (void)device;
    if (!out || rank != 0) return OCC_E_BADARG;
    *out = NULL;
    if (world != 1) return fail(NULL, OCC_E_HIP, "the CPU restatement has no communicator of more than one rank");
    *out = (occ_comm *)calloc(1, sizeof(occ_comm));
    (*out)->world = 1;
    return OCC_OK;
}
int occ_comm_destroy(occ_comm *c) { free(c); return OCC_OK; }
int occ_comm_barrier(occ_comm *c) { return c ? OCC_OK : OCC_E_BADARG; }
int occ_comm_allreduce_max(occ_comm *c, double *inout, int32_t n) { (void)inout; (void)n; return c ? OCC_OK : OCC_E_BADARG; }
int occ_comm_broadcast_host(occ_comm *c, void *buf, int64_t bytes, int32_t root) { (void)buf; (void)bytes; return (c && root == 0) ? OCC_OK : OCC_E_BADARG; }
const char *occ_comm_last_error(const occ_comm *c) { return c ? c->err : g_err; }
int occ_create_distributed(const occ_problem *problem, occ_comm *comm, int32_t root, int32_t n_chains, const uint64_t *keys, occ_sampler **out)
{
    if (!comm || root != 0) return OCC_E_BADARG;
    return occ_create(problem, n_chains, keys, 0, out);
}
