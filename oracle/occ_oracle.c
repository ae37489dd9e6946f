/*
 * occ_oracle.c -- CPU restatement of the reference's LogitICARGibbs inner loop (see occ_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never shipped, never on the product path.
 *
 * Sequential, one chain, plain CSR, plain loops.  Each function cites the reference lines it
 * follows (paths relative to /root/reference/occuspytial/).
 */
#define _GNU_SOURCE
#include "occ_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_SQRT1_2
#define M_SQRT1_2 0.70710678118654752440
#endif
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ================================================================================================
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw 2011, "Parallel random numbers: as easy as 1, 2, 3")
 * ================================================================================================ */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void philox_words(uint64_t key, uint32_t c0, uint32_t c1, uint32_t iter, uint32_t stream,
                         uint64_t w[2])
{
    uint32_t ctr[4] = {c0, c1, iter, stream};
    uint32_t k[2] = {(uint32_t)key, (uint32_t)(key >> 32)};
    uint32_t o[4];
    orc_philox4x32_10(ctr, k, o);
    w[0] = ((uint64_t)o[1] << 32) | o[0];
    w[1] = ((uint64_t)o[3] << 32) | o[2];
}

/* 52 random bits, centred: (k + 1/2) 2^-52, k in [0, 2^52) -- exactly representable, never 0 or 1 */
double orc_u01(uint64_t w) { return ((double)(w >> 12) + 0.5) * 0x1.0p-52; }

static double box_muller(const uint64_t w[2])
{
    double u1 = orc_u01(w[0]), u2 = orc_u01(w[1]);
    return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

double orc_block_normal(uint64_t key, uint32_t c0, uint32_t c1, uint32_t iter, uint32_t stream)
{
    uint64_t w[2];
    philox_words(key, c0, c1, iter, stream, w);
    return box_muller(w);
}

double orc_block_uniform(uint64_t key, uint32_t c0, uint32_t c1, uint32_t iter, uint32_t stream)
{
    uint64_t w[2];
    philox_words(key, c0, c1, iter, stream, w);
    return orc_u01(w[0]);
}

/* sequential cursor over the sub-stream (index, iter, stream): words are handed out one at a time,
 * two per Philox block; a normal always takes a fresh block and drops a pending half block. */
typedef struct {
    uint64_t key;
    uint32_t index, iter, stream, sub;
    int have;
    uint64_t cached;
} cursor_t;

static cursor_t cursor_open(uint64_t key, uint32_t index, uint32_t iter, uint32_t stream)
{
    cursor_t c = {key, index, iter, stream, 0, 0, 0};
    return c;
}
static uint64_t cur_word(cursor_t *c)
{
    if (c->have) { c->have = 0; return c->cached; }
    uint64_t w[2];
    philox_words(c->key, c->index, c->sub++, c->iter, c->stream, w);
    c->cached = w[1];
    c->have = 1;
    return w[0];
}
static double cur_unif(cursor_t *c) { return orc_u01(cur_word(c)); }
static double cur_norm(cursor_t *c)
{
    uint64_t w[2];
    c->have = 0;
    philox_words(c->key, c->index, c->sub++, c->iter, c->stream, w);
    return box_muller(w);
}

/* ================================================================================================
 * PG(1, z): Polson, Scott & Windle (2013) sec. 4 / Devroye (2009): J*(1, z/2)/4 with truncation
 * point t = 0.64.  Stands where the reference calls polyagamma.random_polyagamma(1, b, ...)
 * (gibbs/logit.py:191-193, 202-204).  Distribution-exact; the reference's own draws come from a
 * third-party C library that is not available, so values are not comparable draw for draw.
 *
 * The specification (DESIGN.md "Variate streams"; round 4; csrc/occ_rng.hpp implements the same):
 * ONE rejection loop.  With Z = |z|/2, f = pi^2/8 + Z^2/2 and Devroye's alternating series
 * sum (-1)^n a_n(x), the target density is cosh(Z) exp(-Z^2 x/2) sum (-1)^n a_n(x).  Envelope, piecewise:
 *   x > t            cosh Z (pi/2) exp(-f x)                      mass p  = cosh Z (pi/2) exp(-f t) / f
 *   x <= t, Z <  1/t  cosh Z l(x), l = 2 Levy(0,1) density         mass q0 cosh Z, q0 = 4 Phi(-1/sqrt t)
 *                     (the factor exp(-Z^2 x/2) <= 1 is left to the acceptance test)
 *   x <= t, Z >= 1/t  cosh Z exp(-Z^2 x/2) l(x) = 2 cosh Z e^-Z IG(x; 1/Z, 1), proposed on ALL x > 0,
 *                     a proposal beyond t rejected                 mass 2 cosh Z e^-Z
 * so the right piece is proposed with probability 1 / (1 + k f exp(f t - s)), (k, s) = (2 q0 / pi, 0) below 1/t
 * and (4 / pi, Z) from 1/t on -- cosh Z cancels: one exp, no erfc.  Round r of a draw takes Philox block r of the
 * sub-stream, its four 32-bit words (x0, x1, x2, x3): Ux = u01 of the 64-bit word x1:x0 (52 bits: the proposal's variate),
 * Um = (x2 + 1/2) 2^-32 (which piece), Us = (x3 + 1/2) 2^-32 (the acceptance test) -- probabilities to within 2^-33 --
 * and U2 = (Um - ptail) / (1 - ptail), which is uniform given that the left piece was picked.
 *   right piece   X = t - log(Ux) / f
 *   left piece    N = -Phi^-1(Ux c) by Wichura's AS 241 with -log(Ux c) formed as -(log Ux + log c):
 *                 Z <  1/t: c = Phi(-1/sqrt t), N >= 1/sqrt t by inversion, X = min(1/N^2, t);
 *                 Z >= 1/t: c = 1/2, Y = N^2 is chi-square(1), Michael-Schucany-Haas root of IG(1/Z, 1) with U2
 *                 (the smaller root X kept iff U2 (mu + X) <= mu), round rejected when X > t
 *   acceptance    Us <= E (1 - r_1 + r_2 - ...), E = exp(-Z^2 X / 2) on the left piece below 1/t and 1 elsewhere,
 *                 r_n = a_n / a_0 = (2n+1) exp(-2 n (n+1) / X) for X <= t and (2n+1) exp(-n (n+1) pi^2 X / 2) beyond
 *                 (a_0 itself is never formed), decided by the alternating partial sums.
 * Rounds 1-3 evaluated the proposal's mass by two erfc and three exp, drew the truncated inverse Gaussian in a
 * rejection loop of its own and formed a_0, a_1 in full: about twice the arithmetic, in dependent chains.
 * ================================================================================================ */
#define PG_T 0.64
#define PG_P_LEVY 0.10564977366685526  /* Phi(-1/sqrt(t)) = Phi(-1.25) */
#define PG_LOG_P_LEVY (-2.2476256772143182) /* log of it */
#define PG_LOG_HALF (-0.69314718055994531)
#define PG_K_BELOW 0.26903493944991954 /* 8 Phi(-1.25) / pi = 2 q0 / pi */
#define PG_K_ABOVE 1.2732395447351628  /* 4 / pi */

static double horner8(const double *k, double x)
{
    double r = k[7];
    for (int i = 6; i >= 0; --i) r = r * x + k[i];
    return r;
}
/* -Phi^-1(p) for p in (0, 1/2], Wichura (1988) algorithm AS 241 (PPND16, relative accuracy ~1e-16), with
 * neg_log_p = -log(p) supplied by the caller */
static double pg_neg_quantile(double p, double neg_log_p)
{
    static const double a[8] = {3.3871328727963666080, 1.3314166789178437745e2, 1.9715909503065514427e3,
                                1.3731693765509461125e4, 4.5921953931549871457e4, 6.7265770927008700853e4,
                                3.3430575583588128105e4, 2.5090809287301226727e3};
    static const double b[8] = {1.0, 4.2313330701600911252e1, 6.8718700749205790830e2, 5.3941960214247511077e3,
                                2.1213794301586595867e4, 3.9307895800092710610e4, 2.8729085735721942674e4,
                                5.2264952788528545610e3};
    static const double cc[8] = {1.42343711074968357734, 4.63033784615654529590, 5.76949722146069140550,
                                 3.64784832476320460504, 1.27045825245236838258, 2.41780725177450611770e-1,
                                 2.27238449892691845833e-2, 7.74545014278341407640e-4};
    static const double d[8] = {1.0, 2.05319162663775882187, 1.67638483018380384940, 6.89767334985100004550e-1,
                                1.48103976427480074590e-1, 1.51986665636164571966e-2, 5.47593808499534494600e-4,
                                1.05075007164441684324e-9};
    static const double e[8] = {6.65790464350110377720, 5.46378491116411436990, 1.78482653991729133580,
                                2.96560571828504891230e-1, 2.65321895265761230930e-2, 1.24266094738807843860e-3,
                                2.71155556874348757815e-5, 2.01033439929228813265e-7};
    static const double f[8] = {1.0, 5.99832206555887937690e-1, 1.36929880922735805310e-1, 1.48753612908506148525e-2,
                                7.86869131145613259100e-4, 1.84631831751005468180e-5, 1.42151175831644588870e-7,
                                2.04426310338993978564e-15};
    const double q = p - 0.5;
    if (q >= -0.425) {
        const double r = 0.180625 - q * q;
        return -q * horner8(a, r) / horner8(b, r);
    }
    double r = sqrt(neg_log_p);
    if (r <= 5.0) { r -= 1.6; return horner8(cc, r) / horner8(d, r); }
    r -= 5.0;
    return horner8(e, r) / horner8(f, r);
}

static double pg1_draw_at(uint64_t key, uint32_t index, uint32_t iter, uint32_t stream, double z)
{
    const double Z = 0.5 * fabs(z);
    if (!(Z < 1.0e100)) return (z - z) * NAN;  /* (the device returns NaN for these too: occ_rng.hpp) */
    const double fz = 0.125 * M_PI * M_PI + 0.5 * Z * Z;
    const int below = Z < 1.0 / PG_T;
    const double ptail = 1.0 / (1.0 + (below ? PG_K_BELOW : PG_K_ABOVE) * fz * exp(fz * PG_T - (below ? 0.0 : Z)));
    const double rfz = 1.0 / fz, mu = 1.0 / (below ? 1.0 : Z), hm = 0.5 * mu, hzz = below ? 0.5 * Z * Z : 0.0;
    const double rq = 1.0 / (1.0 - ptail);
    for (uint32_t r = 0;; ++r) {
        uint64_t w[2];
        philox_words(key, index, r, iter, stream, w);
        const double Ux = orc_u01(w[0]), Um = ((double)(uint32_t)w[1] + 0.5) * 0x1.0p-32, Us = ((double)(uint32_t)(w[1] >> 32) + 0.5) * 0x1.0p-32;
        const double U2 = (Um - ptail) * rq;
        const int right = Um < ptail;
        const double lg = log(Ux);
        double X;
        int left_below = 0;
        if (right) {
            X = PG_T - lg * rfz;
        } else {
            const double N = pg_neg_quantile(Ux * (below ? PG_P_LEVY : 0.5), -(lg + (below ? PG_LOG_P_LEVY : PG_LOG_HALF)));
            const double Y = N * N;
            if (below) {
                X = 1.0 / Y;
                if (X > PG_T) X = PG_T;
                left_below = 1;
            } else {
                const double muY = mu * Y;
                X = mu + hm * muY - hm * sqrt(4.0 * muY + muY * muY);
                if (U2 * (mu + X) > mu) X = mu * mu / X;
                if (X > PG_T) continue;  /* beyond the truncation point: the round is rejected */
            }
        }
        if (!(X > 0.0)) continue; /* (never, for a finite Z: the device guards its series loop the same way) */
        const double E = left_below ? exp(-hzz * X) : 1.0;
        if (Us > E) continue;  /* above every partial sum */
        double S = 1.0;
        for (int n = 1; n < 32; ++n) { /* (decided within a few terms; bounded as on the device) */
            const double nn = (double)n * (double)(n + 1);
            const double rn = (double)(2 * n + 1) * (right ? exp(-0.5 * M_PI * M_PI * nn * X) : exp(-2.0 * nn / X));
            if (n & 1) {
                S -= rn;
                if (Us <= E * S) return 0.25 * X;
            } else {
                S += rn;
                if (Us > E * S) break;
            }
        }
    }
}
void orc_pg1_array(uint64_t key, uint32_t iter, uint32_t stream, long n, const double *z, double *out)
{
    for (long i = 0; i < n; ++i) out[i] = pg1_draw_at(key, (uint32_t)i, iter, stream, z[i]);
}

/* standard gamma, Marsaglia & Tsang (2000); shape < 1 by the U^(1/a) boost.  Stands where the
 * reference calls Generator.gamma (gibbs/logit.py:209); same distribution, different stream. */
static double std_gamma(cursor_t *c, double shape)
{
    double boost = 1.0, a = shape;
    if (a < 1.0) {
        boost = pow(cur_unif(c), 1.0 / a);
        a += 1.0;
    }
    double d = a - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x = cur_norm(c);
        double v = 1.0 + cc * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        double u = cur_unif(c);
        if (u < 1.0 - 0.0331 * (x * x) * (x * x)) return boost * d * v;
        if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return boost * d * v;
    }
}
double orc_std_gamma_draw(uint64_t key, uint32_t iter, uint32_t stream, double shape)
{
    cursor_t c = cursor_open(key, 0, iter, stream);
    return std_gamma(&c, shape);
}
/* ... from the sub-stream of `index` (the tau conditional uses index 0) */
double orc_std_gamma_draw_at(uint64_t key, uint32_t index, uint32_t iter, uint32_t stream, double shape)
{
    cursor_t c = cursor_open(key, index, iter, stream);
    return std_gamma(&c, shape);
}

/* ================================================================================================
 * Reference pieces
 * ================================================================================================ */
double orc_expit(double x)
{ /* scipy.special.expit as used at gibbs/logit.py:241-242,249 */
    if (x < 0.0) { double e = exp(x); return e / (1.0 + e); }
    return 1.0 / (1.0 + exp(-x));
}

/* gibbs/logit.py:208  rate = 0.5 * (eta @ Q @ eta) + tau_rate */
double orc_tau_rate(long n, const int64_t *indptr, const int64_t *indices, const double *qdata,
                    const double *eta, double tau_rate)
{
    double quad = 0.0;
    for (long i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) acc += qdata[k] * eta[indices[k]];
        quad += eta[i] * acc;
    }
    return 0.5 * quad + tau_rate;
}

/* distributions.pyx:24-39 */
void orc_ensure_sums_to_zero(long n, const double *x, const double *z, double *out)
{
    double xs = 0.0, zs = 0.0;
    for (long i = 0; i < n; ++i) { xs += x[i]; zs += z[i]; }
    double a = -xs / zs;
    for (long i = 0; i < n; ++i) out[i] = x[i] + a * z[i];
}

/* distributions.pyx:95-105: eps ~ N(0,I); U = chol(prec) upper (dpotrf 'U'); out = U'eps + b (dtrmv
 * 'U','T'); solve U'U out = out (dpotrs).  prec (row-major, symmetric) is overwritten: its upper
 * triangle holds U (the reference's test_distributions.py:15-16 relies on the overwrite). */
int orc_precision_mvnorm(int d, const double *b, double *prec, const double *eps, double *out)
{
    double *U = prec;
    for (int j = 0; j < d; ++j) {
        double s = U[j * d + j];
        for (int k = 0; k < j; ++k) s -= U[k * d + j] * U[k * d + j];
        if (!(s > 0.0)) return j + 1;
        double ujj = sqrt(s);
        U[j * d + j] = ujj;
        for (int i = j + 1; i < d; ++i) {
            double t = U[j * d + i];
            for (int k = 0; k < j; ++k) t -= U[k * d + j] * U[k * d + i];
            U[j * d + i] = t / ujj;
        }
    }
    /* out = U' eps + b */
    for (int i = d - 1; i >= 0; --i) {
        double t = 0.0;
        for (int k = 0; k <= i; ++k) t += U[k * d + i] * eps[k];
        out[i] = t + b[i];
    }
    /* U' y = out (forward), U x = y (backward) */
    for (int i = 0; i < d; ++i) {
        double t = out[i];
        for (int k = 0; k < i; ++k) t -= U[k * d + i] * out[k];
        out[i] = t / U[i * d + i];
    }
    for (int i = d - 1; i >= 0; --i) {
        double t = out[i];
        for (int k = i + 1; k < d; ++k) t -= U[i * d + k] * out[k];
        out[i] = t / U[i * d + i];
    }
    return 0;
}

/* gibbs/logit.py:229-231 */
void orc_beta_system(long n, int p, const double *X, const double *omega, const double *k,
                     const double *spat, const double *b_prec, const double *b_prec_by_mu, double *A,
                     double *r)
{
    for (int a = 0; a < p * p; ++a) A[a] = 0.0;
    for (int a = 0; a < p; ++a) r[a] = 0.0;
    for (long i = 0; i < n; ++i) {
        const double *x = X + i * p;
        double t = k[i] - omega[i] * spat[i];
        for (int a = 0; a < p; ++a) {
            double xo = x[a] * omega[i];
            for (int c = 0; c < p; ++c) A[a * p + c] += xo * x[c];
            r[a] += x[a] * t;
        }
    }
    for (int a = 0; a < p * p; ++a) A[a] += b_prec[a];
    for (int a = 0; a < p; ++a) r[a] += b_prec_by_mu[a];
}

/* gibbs/logit.py:187-190 (which rows exist) and 220-223 (the q x q system).  The reference stacks
 * the rows of the observed sites first and of the newly occupied sites after them; a sum does not
 * depend on that order beyond rounding, so rows are visited in flat order here. */
void orc_alpha_system(long S, int q, const int64_t *site_ptr, const uint8_t *exists_site,
                      const double *W, const double *yrow, const double *omega_a,
                      const double *a_prec, const double *a_prec_by_mu, double *A, double *r)
{
    for (int a = 0; a < q * q; ++a) A[a] = 0.0;
    for (int a = 0; a < q; ++a) r[a] = 0.0;
    for (long s = 0; s < S; ++s) {
        if (!exists_site[s]) continue;
        for (int64_t row = site_ptr[s]; row < site_ptr[s + 1]; ++row) {
            const double *w = W + row * q;
            double t = yrow[row] - 0.5;
            for (int a = 0; a < q; ++a) {
                double wo = w[a] * omega_a[row];
                for (int c = 0; c < q; ++c) A[a * q + c] += wo * w[c];
                r[a] += w[a] * t;
            }
        }
    }
    for (int a = 0; a < q * q; ++a) A[a] += a_prec[a];
    for (int a = 0; a < q; ++a) r[a] += a_prec_by_mu[a];
}

/* gibbs/logit.py:241-245 for one site */
double orc_z_prob(int p, int q, const double *xrow, const double *beta, double eta_i, long nrows,
                  const double *Wrows, const double *alpha)
{
    double lin = 0.0;
    for (int a = 0; a < p; ++a) lin += xrow[a] * beta[a];
    double num1 = orc_expit(lin + eta_i);
    double prod = 1.0;
    for (long v = 0; v < nrows; ++v) {
        double wa = 0.0;
        for (int a = 0; a < q; ++a) wa += Wrows[v * q + a] * (-alpha[a]);
        double e = orc_expit(wa);
        prod = (v == 0) ? e : prod * e;
    }
    double num = num1 * prod;
    return num / ((1.0 - num1) + num);
}

/* Edge form of the prior term.  The reference draws E @ (sqrt(tau) eps) with E the dense
 * eigenfactor of Q (gibbs/logit.py:66-67,77), i.e. a N(0, tau Q) vector.  With Q = D - A a weighted
 * graph Laplacian, Q = B'B for the |edges| x n incidence matrix B whose row for edge (lo,hi), lo<hi,
 * is sqrt(w) (e_lo - e_hi); so u = B' eps, eps ~ N(0, I_edges), is N(0, Q) too -- same law, O(nnz)
 * work, no O(n^2) factor.  eps for edge (lo,hi) is the Box-Muller normal of Philox block
 * (c0=lo, c1=hi, iter, ETA_EDGE). */
void orc_edge_prior_term(long n, const int64_t *indptr, const int64_t *indices, const double *qdata,
                         uint64_t key, uint32_t iter, double *u)
{
    for (long i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
            long j = (long)indices[k];
            if (j == i) continue;
            double w = -qdata[k];
            if (!(w > 0.0)) continue;
            uint32_t lo = (uint32_t)(i < j ? i : j), hi = (uint32_t)(i < j ? j : i);
            double e = orc_block_normal(key, lo, hi, iter, ORC_STREAM_ETA_EDGE);
            double t = sqrt(w) * e;
            acc += (i < j) ? t : -t;
        }
        u[i] = acc;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Joint MINRES.  gibbs/logit.py:80-87 builds P = blockdiag(tau Q, tau Q) + diag([omega; omega]) and
 * rhs = [y; 1] and calls scipy.sparse.linalg.minres(P, rhs, x0=previous xz) with the defaults
 * (rtol 1e-5, shift 0, maxiter 5 * 2n, no preconditioner).  The recurrence below restates
 * Paige & Saunders' MINRES in the form of scipy's _isolve/minres.py (the reference's dependency,
 * scipy 1.15.3 installed here; pinned 1.6.1 in poetry.lock has the same recurrence with `tol`).
 * ------------------------------------------------------------------------------------------------ */
static void joint_matvec(long n, const int64_t *indptr, const int64_t *indices, const double *qdata,
                         const double *omega, double tau, const double *v, double *out)
{
    for (int h = 0; h < 2; ++h) {
        const double *vh = v + h * n;
        double *oh = out + h * n;
        for (long i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
                long j = (long)indices[k];
                double a = tau * qdata[k];
                if (j == i) a = a + omega[i];
                acc += a * vh[j];
            }
            oh[i] = acc;
        }
    }
}
static double dotn(long N, const double *a, const double *b)
{
    double s = 0.0;
    for (long i = 0; i < N; ++i) s += a[i] * b[i];
    return s;
}

long orc_minres_joint(long n, const int64_t *indptr, const int64_t *indices, const double *qdata,
                      const double *omega, double tau, const double *rhs, double *xz, double rtol,
                      long maxiter, long *itn_out, int *istop_out)
{
    const long N = 2 * n;
    const double eps = DBL_EPSILON;
    double *buf = (double *)malloc(sizeof(double) * (size_t)N * 8);
    double *b = buf, *r1 = buf + N, *r2 = buf + 2 * N, *y = buf + 3 * N, *v = buf + 4 * N;
    double *w = buf + 5 * N, *w1 = buf + 6 * N, *w2 = buf + 7 * N;
    double *x = xz;
    long itn = 0;
    int istop = 0;

    for (long i = 0; i < n; ++i) { b[i] = rhs[i]; b[n + i] = 1.0; }
    /* r1 = b - A x   (x0=None in the reference is x = 0, for which this is b exactly) */
    joint_matvec(n, indptr, indices, qdata, omega, tau, x, y);
    for (long i = 0; i < N; ++i) r1[i] = b[i] - y[i];
    memcpy(y, r1, sizeof(double) * (size_t)N);
    double beta1 = dotn(N, r1, y);
    if (beta1 == 0.0) goto done_ok;
    if (dotn(N, b, b) == 0.0) { memcpy(x, b, sizeof(double) * (size_t)N); goto done_ok; }
    beta1 = sqrt(beta1);
    {
        double oldb = 0.0, beta = beta1, dbar = 0.0, epsln = 0.0, phibar = beta1;
        double rhs1 = beta1, rhs2 = 0.0, tnorm2 = 0.0, gmax = 0.0, gmin = DBL_MAX, cs = -1.0, sn = 0.0;
        memset(w, 0, sizeof(double) * (size_t)N);
        memset(w2, 0, sizeof(double) * (size_t)N);
        memcpy(r2, r1, sizeof(double) * (size_t)N);
        while (itn < maxiter) {
            itn += 1;
            double s = 1.0 / beta;
            for (long i = 0; i < N; ++i) v[i] = s * y[i];
            joint_matvec(n, indptr, indices, qdata, omega, tau, v, y);
            if (itn >= 2) {
                double f = beta / oldb;
                for (long i = 0; i < N; ++i) y[i] = y[i] - f * r1[i];
            }
            double alfa = dotn(N, v, y);
            {
                double f = alfa / beta;
                for (long i = 0; i < N; ++i) y[i] = y[i] - f * r2[i];
            }
            { double *t = r1; r1 = r2; r2 = t; }
            memcpy(r2, y, sizeof(double) * (size_t)N);
            oldb = beta;
            beta = dotn(N, r2, y);
            if (beta < 0.0) { istop = 7; break; }
            beta = sqrt(beta);
            tnorm2 += alfa * alfa + oldb * oldb + beta * beta;
            if (itn == 1 && beta / beta1 <= 10.0 * eps) istop = -1;

            double oldeps = epsln;
            double delta = cs * dbar + sn * alfa;
            double gbar = sn * dbar - cs * alfa;
            epsln = sn * beta;
            dbar = -cs * beta;
            double root = sqrt(gbar * gbar + dbar * dbar);

            double gamma = sqrt(gbar * gbar + beta * beta);
            if (gamma < eps) gamma = eps;
            cs = gbar / gamma;
            sn = beta / gamma;
            double phi = cs * phibar;
            phibar = sn * phibar;

            double denom = 1.0 / gamma;
            { double *t = w1; w1 = w2; w2 = w; w = t; }
            for (long i = 0; i < N; ++i) w[i] = (v[i] - oldeps * w1[i] - delta * w2[i]) * denom;
            for (long i = 0; i < N; ++i) x[i] = x[i] + phi * w[i];

            if (gamma > gmax) gmax = gamma;
            if (gamma < gmin) gmin = gamma;
            double zz = rhs1 / gamma;
            rhs1 = rhs2 - delta * zz;
            rhs2 = -epsln * zz;

            double Anorm = sqrt(tnorm2);
            double ynorm = sqrt(dotn(N, x, x));
            double epsx = Anorm * ynorm * eps;
            double rnorm = phibar;
            double test1 = (ynorm == 0.0 || Anorm == 0.0) ? INFINITY : rnorm / (Anorm * ynorm);
            double test2 = (Anorm == 0.0) ? INFINITY : root / Anorm;
            double Acond = gmax / gmin;
            if (istop == 0) {
                double t1 = 1.0 + test1, t2 = 1.0 + test2;
                if (t2 <= 1.0) istop = 2;
                if (t1 <= 1.0) istop = 1;
                if (itn >= maxiter) istop = 6;
                if (Acond >= 0.1 / eps) istop = 4;
                if (epsx >= beta1) istop = 3;
                if (test2 <= rtol) istop = 2;
                if (test1 <= rtol) istop = 1;
            }
            if (istop != 0) break;
        }
    }
done_ok:
    free(buf);
    if (itn_out) *itn_out = itn;
    if (istop_out) *istop_out = istop;
    return (istop == 6) ? maxiter : 0;
}

/* ================================================================================================
 * Whole sampler
 * ================================================================================================ */
struct orc_sampler {
    long n, S, R;
    int p, q;
    int64_t *indptr, *indices, *site_id, *site_ptr;
    double *qdata, *X, *W, *yrow;
    double *a_prec, *b_prec, *a_prec_by_mu, *b_prec_by_mu;
    double tau_rate, tau_shape;
    uint64_t key;
    uint32_t iter;
    /* derived index sets (gibbs/base.py:112-137) */
    uint8_t *surveyed_flag; /* n */
    uint8_t *obs_site;      /* S: species seen on some visit */
    /* state */
    double *alpha, *beta, tau, *eta, *z, *k, *omega_b, *omega_a, *xz, *rhs;
    uint8_t *exists_site; /* S */
    long minres_itn;
    int have_guess;
    /* reduced-rank model (LogitRSRGibbs): rdim > 0 */
    int rdim;
    double *K, *Qr, *Er, *theta;
    /* reference-faithful prior draw (logit.py:66-67,77): dense eigenfactor E, n x (n-1) row-major, BORROWED from the
     * caller (800 MB at 100x100: shared by all chains of a baseline run); NULL: edge form */
    const double *dense_E;
};

static void *dupmem(const void *src, size_t bytes)
{
    void *d = malloc(bytes ? bytes : 1);
    if (src) memcpy(d, src, bytes);
    return d;
}

orc_sampler *orc_create(long n, int p, int q, long S, const int64_t *indptr, const int64_t *indices,
                        const double *qdata, const double *X, const int64_t *site_id,
                        const int64_t *site_ptr, const double *W, const double *yrow,
                        const double *a_mu, const double *a_prec, const double *b_mu,
                        const double *b_prec, double tau_rate, double tau_shape, uint64_t key)
{
    orc_sampler *s = (orc_sampler *)calloc(1, sizeof(*s));
    long nnz = (long)indptr[n], R = (long)site_ptr[S];
    s->n = n; s->p = p; s->q = q; s->S = S; s->R = R;
    s->indptr = dupmem(indptr, sizeof(int64_t) * (size_t)(n + 1));
    s->indices = dupmem(indices, sizeof(int64_t) * (size_t)nnz);
    s->qdata = dupmem(qdata, sizeof(double) * (size_t)nnz);
    s->X = dupmem(X, sizeof(double) * (size_t)(n * p));
    s->site_id = dupmem(site_id, sizeof(int64_t) * (size_t)S);
    s->site_ptr = dupmem(site_ptr, sizeof(int64_t) * (size_t)(S + 1));
    s->W = dupmem(W, sizeof(double) * (size_t)(R * q));
    s->yrow = dupmem(yrow, sizeof(double) * (size_t)R);
    s->a_prec = dupmem(a_prec, sizeof(double) * (size_t)(q * q));
    s->b_prec = dupmem(b_prec, sizeof(double) * (size_t)(p * p));
    s->a_prec_by_mu = calloc((size_t)q, sizeof(double));
    s->b_prec_by_mu = calloc((size_t)p, sizeof(double));
    /* base.py:161-162 */
    for (int a = 0; a < q; ++a) for (int c = 0; c < q; ++c) s->a_prec_by_mu[a] += a_prec[a * q + c] * a_mu[c];
    for (int a = 0; a < p; ++a) for (int c = 0; c < p; ++c) s->b_prec_by_mu[a] += b_prec[a * p + c] * b_mu[c];
    s->tau_rate = tau_rate; s->tau_shape = tau_shape; s->key = key; s->iter = 0;
    s->surveyed_flag = calloc((size_t)n, 1);
    s->obs_site = calloc((size_t)S, 1);
    s->exists_site = calloc((size_t)S, 1);
    s->alpha = calloc((size_t)q, sizeof(double));
    s->beta = calloc((size_t)p, sizeof(double));
    s->eta = calloc((size_t)n, sizeof(double));
    s->z = calloc((size_t)n, sizeof(double));
    s->k = calloc((size_t)n, sizeof(double));
    s->omega_b = calloc((size_t)n, sizeof(double));
    s->omega_a = calloc((size_t)(R ? R : 1), sizeof(double));
    s->xz = calloc((size_t)(2 * n), sizeof(double));
    s->rhs = calloc((size_t)n, sizeof(double));
    /* base.py:113-119: z = 1 everywhere, then z[surveyed] = any(y_i); k = z - 1/2 */
    for (long i = 0; i < n; ++i) s->z[i] = 1.0;
    for (long t = 0; t < S; ++t) {
        int any = 0;
        for (int64_t r = site_ptr[t]; r < site_ptr[t + 1]; ++r) any |= (yrow[r] != 0.0);
        s->obs_site[t] = (uint8_t)any;
        s->surveyed_flag[site_id[t]] = 1;
        s->z[site_id[t]] = any ? 1.0 : 0.0;
    }
    for (long i = 0; i < n; ++i) s->k[i] = s->z[i] - 0.5;
    return s;
}

void orc_destroy(orc_sampler *s)
{
    if (!s) return;
    free(s->indptr); free(s->indices); free(s->site_id); free(s->site_ptr); free(s->qdata); free(s->X);
    free(s->W); free(s->yrow); free(s->a_prec); free(s->b_prec); free(s->a_prec_by_mu);
    free(s->b_prec_by_mu); free(s->surveyed_flag); free(s->obs_site); free(s->exists_site);
    free(s->alpha); free(s->beta); free(s->eta); free(s->z); free(s->k); free(s->omega_b);
    free(s->omega_a); free(s->xz); free(s->rhs);
    free(s->K); free(s->Qr); free(s->Er); free(s->theta);
    free(s);
}

/* base.py:188-197 (explicit start; the default start is drawn on the host with numpy exactly as
 * base.py:199-212 does and handed in here) */
void orc_set_start(orc_sampler *s, const double *alpha, const double *beta, double tau, const double *eta)
{
    memcpy(s->alpha, alpha, sizeof(double) * (size_t)s->q);
    memcpy(s->beta, beta, sizeof(double) * (size_t)s->p);
    s->tau = tau;
    memcpy(s->eta, eta, sizeof(double) * (size_t)s->n);
}

static double xdot(const orc_sampler *s, long i, const double *coef)
{
    double acc = 0.0;
    for (int a = 0; a < s->p; ++a) acc += s->X[i * s->p + a] * coef[a];
    return acc;
}

/* gibbs/logit.py:195-204 */
int orc_update_omega_b(orc_sampler *s)
{
    for (long i = 0; i < s->n; ++i) {
        cursor_t c = cursor_open(s->key, (uint32_t)i, s->iter, ORC_STREAM_OMEGA_B);
        s->omega_b[i] = pg1_draw_at(c.key, c.index, c.iter, c.stream, xdot(s, i, s->beta) + s->eta[i]);
    }
    return 0;
}

/* gibbs/logit.py:206-209:  tau = Generator.gamma(shape, 1/rate) = (1/rate) * standard_gamma(shape) */
int orc_update_tau(orc_sampler *s)
{
    double rate;
    if (s->rdim) { /* fixed.Q is K'QK and state.eta is theta (logit.py:453-455, 206-209) */
        double quad = 0.0;
        for (int a = 0; a < s->rdim; ++a) {
            double t = 0.0;
            for (int c = 0; c < s->rdim; ++c) t += s->Qr[a * s->rdim + c] * s->theta[c];
            quad += s->theta[a] * t;
        }
        rate = 0.5 * quad + s->tau_rate;
    } else {
        rate = orc_tau_rate(s->n, s->indptr, s->indices, s->qdata, s->eta, s->tau_rate);
    }
    double g = orc_std_gamma_draw(s->key, s->iter, ORC_STREAM_TAU, s->tau_shape);
    s->tau = (1.0 / rate) * g;
    return 0;
}

/* gibbs/logit.py:211-217 and 73-99 (prior term in edge form, see orc_edge_prior_term) */
/* ---- reduced-rank model ---------------------------------------------------------------------- */
int orc_rsr_theta(long n, int r, const double *K, const double *Qr, const double *Er, const double *b,
                  const double *omega, double tau, const double *eps1, const double *eps2, double *theta)
{
    double *prec = (double *)calloc((size_t)r * r, sizeof(double)), *rhs = (double *)calloc((size_t)r, sizeof(double));
    double st = sqrt(tau);
    /* prec = K' diag(omega) K + tau Qr (logit.py:334: factor1 factor1' + tau Q);  rhs = K'(b + sqrt(omega) eps1) */
    for (long i = 0; i < n; ++i) {
        const double *ki = K + i * r;
        double v = b[i] + sqrt(omega[i]) * eps1[i];
        for (int a = 0; a < r; ++a) {
            rhs[a] += ki[a] * v;
            double ko = ki[a] * omega[i];
            for (int c = a; c < r; ++c) prec[a * r + c] += ko * ki[c];
        }
    }
    for (int a = 0; a < r; ++a) {
        double t = 0.0;
        for (int j = 0; j < r; ++j) t += Er[a * r + j] * eps2[j];
        rhs[a] += st * t;
        for (int c = a; c < r; ++c) prec[a * r + c] += tau * Qr[a * r + c];
    }
    /* upper Cholesky prec = U'U in place, then U'y = rhs, U theta = y (the reference calls np.linalg.solve) */
    int rc = 0;
    for (int j = 0; j < r && !rc; ++j) {
        double s = prec[j * r + j];
        for (int k = 0; k < j; ++k) s -= prec[k * r + j] * prec[k * r + j];
        if (!(s > 0.0)) { rc = ORC_ERR_CHOLESKY; break; }
        double ujj = sqrt(s);
        prec[j * r + j] = ujj;
        for (int i = j + 1; i < r; ++i) {
            double t = prec[j * r + i];
            for (int k = 0; k < j; ++k) t -= prec[k * r + j] * prec[k * r + i];
            prec[j * r + i] = t / ujj;
        }
    }
    if (!rc) {
        for (int i = 0; i < r; ++i) {
            double t = rhs[i];
            for (int k = 0; k < i; ++k) t -= prec[k * r + i] * theta[k];
            theta[i] = t / prec[i * r + i];
        }
        for (int i = r - 1; i >= 0; --i) {
            double t = theta[i];
            for (int k = i + 1; k < r; ++k) t -= prec[i * r + k] * theta[k];
            theta[i] = t / prec[i * r + i];
        }
    }
    free(prec); free(rhs);
    return rc;
}

static void rsr_spatial(orc_sampler *s)
{
    for (long i = 0; i < s->n; ++i) {
        double t = 0.0;
        for (int a = 0; a < s->rdim; ++a) t += s->K[i * s->rdim + a] * s->theta[a];
        s->eta[i] = t;
    }
}

int orc_set_rsr(orc_sampler *s, int r, const double *K, const double *Qr, const double *Er)
{
    if (r < 1) return -1;
    s->rdim = r;
    s->K = dupmem(K, sizeof(double) * (size_t)(s->n * r));
    s->Qr = dupmem(Qr, sizeof(double) * (size_t)r * r);
    s->Er = dupmem(Er, sizeof(double) * (size_t)r * r);
    s->theta = calloc((size_t)r, sizeof(double));
    return 0;
}

/* logit.py:465-485: b = K'(k - omega X beta), theta from the reduced system, spatial = K theta */
static int update_eta_rsr(orc_sampler *s)
{
    long n = s->n;
    int r = s->rdim;
    double *b = (double *)malloc(sizeof(double) * (size_t)n), *e1 = (double *)malloc(sizeof(double) * (size_t)n);
    double *e2 = (double *)malloc(sizeof(double) * (size_t)r);
    for (long i = 0; i < n; ++i) {
        b[i] = s->k[i] - s->omega_b[i] * xdot(s, i, s->beta);
        e1[i] = orc_block_normal(s->key, (uint32_t)i, 0, s->iter, ORC_STREAM_ETA_SITE);
    }
    for (int j = 0; j < r; ++j) e2[j] = orc_block_normal(s->key, (uint32_t)j, 0, s->iter, ORC_STREAM_RSR);
    int rc = orc_rsr_theta(n, r, s->K, s->Qr, s->Er, b, s->omega_b, s->tau, e1, e2, s->theta);
    free(b); free(e1); free(e2);
    if (rc) return rc;
    rsr_spatial(s);
    return 0;
}

/* u = E v, E row-major n x m: the dense matvec the reference pays every iteration (logit.py:77; numpy hands it to BLAS
 * dgemv).  Eight independent partial sums so that the compiler can keep the FMA pipes full without reassociating. */
__attribute__((optimize("O3"))) static void dense_matvec(long n, long m, const double *E, const double *v, double *u)
{
    for (long i = 0; i < n; ++i) {
        const double *row = E + (size_t)i * (size_t)m;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
        long j = 0;
        for (; j + 8 <= m; j += 8) {
            a0 += row[j] * v[j];         a1 += row[j + 1] * v[j + 1]; a2 += row[j + 2] * v[j + 2]; a3 += row[j + 3] * v[j + 3];
            a4 += row[j + 4] * v[j + 4]; a5 += row[j + 5] * v[j + 5]; a6 += row[j + 6] * v[j + 6]; a7 += row[j + 7] * v[j + 7];
        }
        for (; j < m; ++j) a0 += row[j] * v[j];
        u[i] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    }
}

/* Switch the prior term of the eta conditional to the reference's own form (logit.py:64-67, 77): u = E (sqrt(tau) eps_2)
 * with E the n x (n-1) eigenfactor of Q the caller computed (E E' = Q; numpy eigh on the host) and eps_2 the n-1 standard
 * normals of Philox stream ORC_STREAM_ETA_DENSE.  E is borrowed, not copied.  NULL switches back to the edge form. */
void orc_set_dense_eigen(orc_sampler *s, const double *E) { s->dense_E = E; }
/* a new Philox key for the chain's variate streams (the engine's occ_set_keys) */
void orc_set_key(orc_sampler *s, uint64_t key) { s->key = key; }

int orc_update_eta(orc_sampler *s)
{
    if (s->rdim) return update_eta_rsr(s);
    long n = s->n;
    double *u = (double *)malloc(sizeof(double) * (size_t)n);
    double st = sqrt(s->tau);
    if (s->dense_E) {  /* reference-faithful: y = b + sqrt(omega) eps_1 + E (sqrt(tau) eps_2) */
        double *v = (double *)malloc(sizeof(double) * (size_t)(n - 1));
        for (long j = 0; j < n - 1; ++j) v[j] = st * orc_block_normal(s->key, (uint32_t)j, 0, s->iter, ORC_STREAM_ETA_DENSE);
        dense_matvec(n, n - 1, s->dense_E, v, u);
        free(v);
        st = 1.0;
    } else {
        orc_edge_prior_term(n, s->indptr, s->indices, s->qdata, s->key, s->iter, u);
    }
    for (long i = 0; i < n; ++i) {
        double b = s->k[i] - s->omega_b[i] * xdot(s, i, s->beta);
        double e = orc_block_normal(s->key, (uint32_t)i, 0, s->iter, ORC_STREAM_ETA_SITE);
        s->rhs[i] = (b + sqrt(s->omega_b[i]) * e) + st * u[i];
    }
    free(u);
    int istop;
    long info = orc_minres_joint(n, s->indptr, s->indices, s->qdata, s->omega_b, s->tau, s->rhs, s->xz,
                                 1e-5, 5 * 2 * n, &s->minres_itn, &istop);
    s->have_guess = 1;
    if (info) return ORC_ERR_MINRES;
    orc_ensure_sums_to_zero(n, s->xz, s->xz + n, s->eta);
    return 0;
}

/* gibbs/logit.py:226-232 */
int orc_update_beta(orc_sampler *s)
{
    int p = s->p;
    double A[64 * 64], r[64], eps[64], out[64];
    if (p > 64) return ORC_ERR_CHOLESKY;
    orc_beta_system(s->n, p, s->X, s->omega_b, s->k, s->eta, s->b_prec, s->b_prec_by_mu, A, r);
    for (int j = 0; j < p; ++j) eps[j] = orc_block_normal(s->key, (uint32_t)j, 0, s->iter, ORC_STREAM_BETA);
    if (orc_precision_mvnorm(p, r, A, eps, out)) return ORC_ERR_CHOLESKY;
    memcpy(s->beta, out, sizeof(double) * (size_t)p);
    return 0;
}

/* gibbs/logit.py:180-193: rows of every site with a detection plus rows of not-observed sites whose
 * current z is 1; PG(1, w'alpha) per such row.  Row r of the flat W is sub-stream index r. */
int orc_update_omega_a(orc_sampler *s)
{
    for (long t = 0; t < s->S; ++t) {
        int ex = s->obs_site[t] || (s->z[s->site_id[t]] != 0.0);
        s->exists_site[t] = (uint8_t)ex;
        if (!ex) continue;
        for (int64_t r = s->site_ptr[t]; r < s->site_ptr[t + 1]; ++r) {
            double wa = 0.0;
            for (int a = 0; a < s->q; ++a) wa += s->W[r * s->q + a] * s->alpha[a];
            cursor_t c = cursor_open(s->key, (uint32_t)r, s->iter, ORC_STREAM_OMEGA_A);
            s->omega_a[r] = pg1_draw_at(c.key, c.index, c.iter, c.stream, wa);
        }
    }
    return 0;
}

/* gibbs/logit.py:219-224 */
int orc_update_alpha(orc_sampler *s)
{
    int q = s->q;
    double A[64 * 64], r[64], eps[64], out[64];
    if (q > 64) return ORC_ERR_CHOLESKY;
    orc_alpha_system(s->S, q, s->site_ptr, s->exists_site, s->W, s->yrow, s->omega_a, s->a_prec,
                     s->a_prec_by_mu, A, r);
    for (int j = 0; j < q; ++j) eps[j] = orc_block_normal(s->key, (uint32_t)j, 0, s->iter, ORC_STREAM_ALPHA);
    if (orc_precision_mvnorm(q, r, A, eps, out)) return ORC_ERR_CHOLESKY;
    memcpy(s->alpha, out, sizeof(double) * (size_t)q);
    return 0;
}

/* gibbs/logit.py:234-252 */
int orc_update_z(orc_sampler *s)
{
    for (long t = 0; t < s->S; ++t) {
        if (s->obs_site[t]) continue;
        long i = (long)s->site_id[t];
        double pr = orc_z_prob(s->p, s->q, s->X + i * s->p, s->beta, s->eta[i],
                               (long)(s->site_ptr[t + 1] - s->site_ptr[t]), s->W + s->site_ptr[t] * s->q,
                               s->alpha);
        double u = orc_block_uniform(s->key, (uint32_t)i, 0, s->iter, ORC_STREAM_Z);
        s->z[i] = (u < pr) ? 1.0 : 0.0;
    }
    for (long i = 0; i < s->n; ++i) {
        if (s->surveyed_flag[i]) continue;
        double pr = orc_expit(xdot(s, i, s->beta) + s->eta[i]);
        double u = orc_block_uniform(s->key, (uint32_t)i, 0, s->iter, ORC_STREAM_Z);
        s->z[i] = (u < pr) ? 1.0 : 0.0;
    }
    for (long i = 0; i < s->n; ++i) s->k[i] = s->z[i] - 0.5;
    return 0;
}

/* gibbs/logit.py:254-266 */
int orc_step(orc_sampler *s)
{
    int e;
    if ((e = orc_update_omega_b(s))) return e;
    if ((e = orc_update_tau(s))) return e;
    if ((e = orc_update_eta(s))) return e;
    if ((e = orc_update_beta(s))) return e;
    if ((e = orc_update_omega_a(s))) return e;
    if ((e = orc_update_alpha(s))) return e;
    if ((e = orc_update_z(s))) return e;
    s->iter += 1;
    return 0;
}

/* gibbs/base.py:236-239 */
int orc_run(orc_sampler *s, long n_iter, long burnin, double *out_alpha, double *out_beta, double *out_tau)
{
    for (long i = 0; i < n_iter; ++i) {
        int e = orc_step(s);
        if (e) return e;
        if (i >= burnin) {
            long row = i - burnin;
            memcpy(out_alpha + row * s->q, s->alpha, sizeof(double) * (size_t)s->q);
            memcpy(out_beta + row * s->p, s->beta, sizeof(double) * (size_t)s->p);
            out_tau[row] = s->tau;
        }
    }
    return 0;
}

static long copy_out(const double *src, long len, double *out, long cap)
{
    if (out && cap >= len) memcpy(out, src, sizeof(double) * (size_t)len);
    return len;
}

long orc_get(orc_sampler *s, const char *name, double *out, long cap)
{
    if (!strcmp(name, "alpha")) return copy_out(s->alpha, s->q, out, cap);
    if (!strcmp(name, "beta")) return copy_out(s->beta, s->p, out, cap);
    if (!strcmp(name, "tau")) return copy_out(&s->tau, 1, out, cap);
    if (!strcmp(name, "eta")) return copy_out(s->eta, s->n, out, cap);
    if (!strcmp(name, "theta")) return s->rdim ? copy_out(s->theta, s->rdim, out, cap) : -1;
    if (!strcmp(name, "z")) return copy_out(s->z, s->n, out, cap);
    if (!strcmp(name, "k")) return copy_out(s->k, s->n, out, cap);
    if (!strcmp(name, "omega_b")) return copy_out(s->omega_b, s->n, out, cap);
    if (!strcmp(name, "omega_a")) return copy_out(s->omega_a, s->R, out, cap);
    if (!strcmp(name, "xz")) return copy_out(s->xz, 2 * s->n, out, cap);
    if (!strcmp(name, "rhs")) return copy_out(s->rhs, s->n, out, cap);
    if (!strcmp(name, "minres_itn")) { double v = (double)s->minres_itn; return copy_out(&v, 1, out, cap); }
    if (!strcmp(name, "iter")) { double v = (double)s->iter; return copy_out(&v, 1, out, cap); }
    if (!strcmp(name, "exists")) {
        if (out && cap >= s->S) for (long t = 0; t < s->S; ++t) out[t] = s->exists_site[t];
        return s->S;
    }
    return -1;
}

int orc_set(orc_sampler *s, const char *name, const double *in, long len)
{
    double *dst = NULL;
    long want = 0;
    if (!strcmp(name, "alpha")) { dst = s->alpha; want = s->q; }
    else if (!strcmp(name, "beta")) { dst = s->beta; want = s->p; }
    else if (!strcmp(name, "tau")) { dst = &s->tau; want = 1; }
    else if (!strcmp(name, "eta")) { dst = s->eta; want = s->n; }
    else if (!strcmp(name, "omega_b")) { dst = s->omega_b; want = s->n; }
    else if (!strcmp(name, "omega_a")) { dst = s->omega_a; want = s->R; }
    else if (!strcmp(name, "xz")) { dst = s->xz; want = 2 * s->n; }
    else if (!strcmp(name, "z")) {
        if (len != s->n) return -1;
        for (long i = 0; i < s->n; ++i) { s->z[i] = in[i]; s->k[i] = in[i] - 0.5; }
        return 0;
    } else if (!strcmp(name, "iter")) {
        if (len != 1) return -1;
        s->iter = (uint32_t)in[0];
        return 0;
    } else if (!strcmp(name, "theta")) { /* RSR: eta follows as K theta (logit.py:459, 485) */
        if (!s->rdim || len != s->rdim) return -1;
        memcpy(s->theta, in, sizeof(double) * (size_t)len);
        rsr_spatial(s);
        return 0;
    } else return -1;
    if (len != want) return -1;
    memcpy(dst, in, sizeof(double) * (size_t)want);
    return 0;
}
