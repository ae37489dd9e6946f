/*
 * occ_oracle.h -- CPU restatement of the reference's LogitICARGibbs inner loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: it is imported, linked
 * or executed only by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, and
 * only as the checker / the timed CPU baseline.  The product path is the HIP library declared in
 * include/occ_gibbs.h and has no CPU fallback.
 *
 * Pinning: the piece-wise functions below (tau rate, joint MINRES, sum-to-zero
 * projection, beta/alpha systems + precision_mvnorm, z probabilities) are checked against fixtures
 * produced by running the reference's own Python/Cython code (tests/golden/make_golden.py) with the
 * variates the reference consumed injected.  The Polya-Gamma sampler is NOT pinned by the reference:
 * `polyagamma` 1.2.0 (pyproject.toml:39) is a third-party dependency absent from /root/reference and
 * from this image, so PG(1,z) is the published Devroye/Polson-Scott-Windle sampler checked against
 * the distribution's closed-form moments and Laplace transform -- "PG parity unpinned".
 *
 * All citations are to files under /root/reference/occuspytial/ unless stated otherwise.
 */
#ifndef OCC_ORACLE_H
#define OCC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- counter-based RNG shared (as a SPEC, not as code) with the HIP kernels ------------------- */
enum {
    ORC_STREAM_OMEGA_B = 1,
    ORC_STREAM_TAU = 2,
    ORC_STREAM_ETA_SITE = 3,
    ORC_STREAM_ETA_EDGE = 4,
    ORC_STREAM_BETA = 5,
    ORC_STREAM_OMEGA_A = 6,
    ORC_STREAM_ALPHA = 7,
    ORC_STREAM_Z = 8,
    ORC_STREAM_RSR = 9, /* standard normals of the reduced-rank (RSR) prior term: c0 = basis column */
    ORC_STREAM_ETA_DENSE = 10 /* the n - 1 standard normals of the dense-eigenfactor prior draw: c0 = column */
};

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* uniform in (0,1) from one 64-bit word; normal from one Philox block (Box-Muller, cosine branch) */
double orc_u01(uint64_t w);
double orc_block_normal(uint64_t key, uint32_t c0, uint32_t c1, uint32_t iter, uint32_t stream);
double orc_block_uniform(uint64_t key, uint32_t c0, uint32_t c1, uint32_t iter, uint32_t stream);
/* sequential draws from the (index, iter, stream) sub-stream */
void orc_pg1_array(uint64_t key, uint32_t iter, uint32_t stream, long n, const double *z, double *out);
double orc_std_gamma_draw(uint64_t key, uint32_t iter, uint32_t stream, double shape);
double orc_std_gamma_draw_at(uint64_t key, uint32_t index, uint32_t iter, uint32_t stream, double shape);

/* ---- reference pieces with injected variates --------------------------------------------------- */
/* logit.py:208  rate = 0.5 * eta' Q eta + tau_rate */
double orc_tau_rate(long n, const int64_t *indptr, const int64_t *indices, const double *qdata,
                    const double *eta, double tau_rate);
/* logit.py:80-92 + scipy minres.py: joint 2n system [tau Q + diag(omega)] [x z] = [y 1].
 * xz (2n) holds x0 on entry (zeros = reference's x0=None) and the solution on exit.
 * returns scipy's `info` (0 ok, maxiter when the iteration limit was hit). */
long orc_minres_joint(long n, const int64_t *indptr, const int64_t *indices, const double *qdata,
                      const double *omega, double tau, const double *rhs, double *xz, double rtol,
                      long maxiter, long *itn_out, int *istop_out);
/* distributions.pyx:24-39 */
void orc_ensure_sums_to_zero(long n, const double *x, const double *z, double *out);
/* distributions.pyx:42-110 (eps = the standard normals the reference draws; prec is overwritten
 * by its upper Cholesky factor, row-major).  returns 0, or j+1 when pivot j is not positive. */
int orc_precision_mvnorm(int d, const double *b, double *prec, const double *eps, double *out);
/* logit.py:229-231 */
void orc_beta_system(long n, int p, const double *X, const double *omega, const double *k,
                     const double *spat, const double *b_prec, const double *b_prec_by_mu, double *A,
                     double *r);
/* logit.py:187-190,220-223 over the rows of the sites flagged in exists_site (flat row order) */
void orc_alpha_system(long S, int q, const int64_t *site_ptr, const uint8_t *exists_site,
                      const double *W, const double *yrow, const double *omega_a,
                      const double *a_prec, const double *a_prec_by_mu, double *A, double *r);
/* logit.py:241-245: occupancy probability of one not-observed surveyed site */
double orc_z_prob(int p, int q, const double *xrow, const double *beta, double eta_i, long nrows,
                  const double *Wrows, const double *alpha);
double orc_expit(double x);
/* edge form of the ICAR prior term: u = B' eps with Q = B'B (replaces logit.py:66-67,77) */
void orc_edge_prior_term(long n, const int64_t *indptr, const int64_t *indices, const double *qdata,
                         uint64_t key, uint32_t iter, double *u);

/* ---- whole sampler (one chain), Philox variates -------------------------------------------------- */
typedef struct orc_sampler orc_sampler;

orc_sampler *orc_create(long n, int p, int q, long S, const int64_t *indptr, const int64_t *indices,
                        const double *qdata, const double *X, const int64_t *site_id,
                        const int64_t *site_ptr, const double *W, const double *yrow,
                        const double *a_mu, const double *a_prec, const double *b_mu,
                        const double *b_prec, double tau_rate, double tau_shape, uint64_t key);
void orc_destroy(orc_sampler *s);
void orc_set_start(orc_sampler *s, const double *alpha, const double *beta, double tau, const double *eta);
/* individual conditionals in the order of logit.py:254-266; each returns 0 or an error code */
int orc_update_omega_b(orc_sampler *s);
int orc_update_tau(orc_sampler *s);
int orc_update_eta(orc_sampler *s);
int orc_update_beta(orc_sampler *s);
int orc_update_omega_a(orc_sampler *s);
int orc_update_alpha(orc_sampler *s);
int orc_update_z(orc_sampler *s);
int orc_step(orc_sampler *s); /* all seven, then iter += 1 */
/* run n_iter steps, recording alpha|beta|tau of iterations >= burnin (base.py:236-239) */
int orc_run(orc_sampler *s, long n_iter, long burnin, double *out_alpha, double *out_beta, double *out_tau);
/* state access: name in {alpha,beta,tau,eta,z,k,omega_b,omega_a,xz,rhs,exists,minres_itn,iter} */
long orc_get(orc_sampler *s, const char *name, double *out, long cap);
int orc_set(orc_sampler *s, const char *name, const double *in, long len);

#define ORC_ERR_MINRES 1   /* RuntimeError('MINRES solver did not converge!')  logit.py:91-92 */
#define ORC_ERR_CHOLESKY 2 /* RuntimeError('Cholesky factorization/solver failed!') distributions.pyx:21 */

#ifdef __cplusplus
}
#endif
/* ---- reduced-rank spatial effects (LogitRSRGibbs, gibbs/logit.py:269-485) ----------------------------
 * theta ~ N(prec^-1 r, prec^-1), prec = K' diag(omega) K + tau Qr, r = K' b + K'(sqrt(omega) eps1) + sqrt(tau) Er eps2
 * (logit.py:325-337; Er = eigenfactor of Qr, Er Er' = Qr).  K is n x r row-major.  Returns 0 or ORC_ERR_CHOLESKY. */
int orc_rsr_theta(long n, int r, const double *K, const double *Qr, const double *Er, const double *b,
                  const double *omega, double tau, const double *eps1, const double *eps2, double *theta);
/* switch a sampler to the RSR model: eta becomes K theta, tau's rate 1/2 theta' Qr theta (logit.py:206-209 with
 * fixed.Q = K'QK, logit.py:453-455); state names "theta" (r) and "eta" (= spatial, n). */
int orc_set_rsr(orc_sampler *s, int r, const double *K, const double *Qr, const double *Er);
/* reference-faithful prior draw of the eta conditional (logit.py:64-67, 77): dense eigenfactor E (n x (n-1), row-major,
 * E E' = Q, borrowed); the build's own form is the edge factorisation (orc_edge_prior_term).  Same law either way. */
void orc_set_dense_eigen(orc_sampler *s, const double *E);
void orc_set_key(orc_sampler *s, uint64_t key);

#endif
