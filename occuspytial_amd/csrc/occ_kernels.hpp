// occ_kernels.hpp -- the Gibbs iteration as gfx950 kernels (chains batched on blockIdx.y).
//
// One Gibbs iteration of LogitICARGibbs.step() (occuspytial/gibbs/logit.py:254-266) is a fixed
// sequence of small kernels; a kernel boundary is the only grid-wide synchronisation used (cheaper on
// MI355X than any in-kernel all-to-all).  Global sums are "reduce at the consumer":
// every block writes one partial per quantity, and the NEXT kernel's blocks each re-reduce all
// partials in the same fixed order, so scalars are bit-identical across blocks and runs (no float
// atomics anywhere).
//
//   k_omega_b      omega_b ~ PG(1, x'beta + eta) per site; eta'Q eta partials; eta-rhs pieces
//                  (logit.py:195-204, 208, 213, 75-78)
//   k_eta_init     tau ~ Gamma (logit.py:206-209); rhs y; r1 = [y;1] - Lambda x0  (logit.py:78-87)
//   k_minres_a/b   one Lanczos/MINRES iteration of the joint 2n system, two kernels per iteration
//                  (scipy _isolve/minres.py as called at logit.py:87)
//   k_beta_partial eta = x - (sum x / sum z) z (distributions.pyx:24-39); X' Omega X, X'(k - omega eta)
//   k_omega_a      beta draw (block 0; distributions.pyx:42-110); omega_a ~ PG(1, w'alpha) for rows of
//                  existing sites; W' Omega W, W'(y - 1/2)   (logit.py:180-193, 219-223)
//   k_z            alpha draw; z update (logit.py:234-252); record (alpha, beta, tau) (base.py:238-239)
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "occ_rng.hpp"

namespace occ {

constexpr int MAXC = 8;                             // OCC_MAX_COVARIATES
constexpr int NACC_MAX = MAXC * (MAXC + 1) / 2 + MAXC;  // 44
constexpr int NSLOT = 4;
constexpr int MAX_WAVES = 4;  // threads per block <= 256

__host__ __device__ constexpr int nacc(int d) { return d * (d + 1) / 2 + d; }

// MINRES scalar state of one chain; slot s is written by step s and read by step s+1.
#ifndef OCC_SLOT_ALIGN
#define OCC_SLOT_ALIGN 8
#endif
struct alignas(OCC_SLOT_ALIGN) Slot {
    double beta1, beta, oldb, alfa, dbar, epsln, phibar, rhs1, rhs2, tnorm2, gmax, gmin, cs, sn, root;
    int itn;    // Lanczos steps completed
    int istop;  // scipy's istop code
    int done;   // x is final
    int pad;
};

// Slots are read and written FIELD BY FIELD (never `Slot s = *p; ... *q = s;`): a whole-struct copy
// makes hipcc keep the struct in memory (LDS-promoted alloca / scratch) and load it through a mix of
// scalar and vector paths, which cost k_minres_b 13-25 us per launch on MI355X.
#define OCC_SLOT_FIELDS(X) \
    X(beta1) X(beta) X(oldb) X(alfa) X(dbar) X(epsln) X(phibar) X(rhs1) X(rhs2) X(tnorm2) X(gmax) X(gmin) \
    X(cs) X(sn) X(root) X(itn) X(istop) X(done)
__device__ __forceinline__ Slot slot_load(const Slot *p)
{
    Slot s;
#define X(f) s.f = p->f;
    OCC_SLOT_FIELDS(X)
#undef X
    s.pad = 0;
    return s;
}
__device__ __forceinline__ void slot_store(Slot *p, const Slot &s)
{
#define X(f) p->f = s.f;
    OCC_SLOT_FIELDS(X)
#undef X
}

struct Ctl {
    uint32_t it;    // Gibbs iteration number (Philox counter word 2)
    uint32_t koff;  // Krylov steps already spent on the current eta solve by earlier graph replays:
                    // 0 normally; > 0 when a replay ran out of captured steps and the NEXT replay
                    // continues the same solve (no host involvement, same arithmetic)
};

struct ChainScalars {
    double alpha[MAXC], beta[MAXC];
    double tau;
    uint64_t key;
    Ctl next, cur, mid;  // written by k_z / k_omega_b / k_beta_partial respectively (race-free hand-over)
    uint32_t it_stop, it_base, burnin, keep;
    int32_t err;               // OCC_E_* raised on device
    int32_t minres_itn_last;
    unsigned long long krylov_total, krylov_sq_total, solves, carries;
};

// The problem/state descriptor lives in device memory and kernels receive a POINTER to it (plus the
// two hot per-chain tables): a by-value 360-byte kernel argument costs every wave several serialized
// kernarg-segment fetches at kernel entry -- measured 12-25 us per launch on MI355X for kernels whose
// scalar loads the compiler could not batch -- while a 40-byte argument block is one fetch.
struct Ctx {
    int n, S, R, p, q, C;
    int nb_n, nb_r, nb_max;
    long long maxiter;
    // fixed inputs (shared by all chains)
    const int *sell_ptr, *sell_col;
    const double *sell_val, *qdiag;
    const double *Xt, *Wt;
    const uint8_t *yrow;
    const int *row_site;   // site number | obs << 31
    const int *site_sidx;  // surveyed index of a site, -1 if not surveyed
    const int *site_ptr;
    const uint8_t *obs_site;
    const double *hyp;  // a_prec[q*q], a_prec_by_mu[q], b_prec[p*p], b_prec_by_mu[p]
    double tau_rate, tau_shape;
    // per-chain state
    double *eta, *omega_b, *pre, *uprior, *rhs, *omega_a;
    uint8_t *z;
    double2 *Rv[3], *Wv[3], *Xv;
    double *part;       // [C][2][NACC_MAX * nb_max]
    double *part_proj;  // [C][2 * nb_n]
    Slot *slots;        // [C][NSLOT]
    ChainScalars *sc;   // [C]
    double *rec;        // [C][keep][q + p + 1]
};

// ---- reductions ----------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // lane 0 holds the sum
}

// Each block writes one partial per quantity: out[q * nb + blk].  lds: MAX_WAVES * NQ doubles.
template <int NQ>
__device__ __forceinline__ void block_partials(const double (&v)[NQ], double *lds, double *out, int nb, int blk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        const double r = wave_sum(v[qi]);
        if (lane == 0) lds[wave * NQ + qi] = r;
    }
    __syncthreads();
    if ((int)threadIdx.x < NQ) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += lds[w * NQ + threadIdx.x];
        out[threadIdx.x * nb + blk] = s;
    }
}

// Every block reduces all nb partials of nq quantities in the same order -> identical scalars.
__device__ __forceinline__ void reduce_partials(const double *part, int nq, int nb, double *lds_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int qi = wave; qi < nq; qi += nw) {
        double s = 0.0;
        for (int b = lane; b < nb; b += 64) s += part[qi * nb + b];
        s = wave_sum(s);
        if (lane == 0) lds_out[qi] = s;
    }
    __syncthreads();
}

__device__ __forceinline__ double *part_buf(const Ctx &c, int chain, int parity)
{
    return c.part + ((size_t)chain * 2 + parity) * ((size_t)NACC_MAX * c.nb_max);
}

__device__ __forceinline__ double expit(double x)
{
    if (x < 0.0) { const double e = exp(x); return e / (1.0 + e); }
    return 1.0 / (1.0 + exp(-x));
}

__device__ __forceinline__ double xdot(const double *Xt, int n, int i, const double *coef, int p)
{
    double acc = 0.0;
    for (int a = 0; a < p; ++a) acc += Xt[(size_t)a * n + i] * coef[a];
    return acc;
}

// distributions.pyx:95-105 on device, executed by ONE thread on small LDS work arrays (runtime
// dimension d <= MAXC, so no per-dimension template and no register arrays): upper Cholesky U of the
// d x d precision (packed upper accumulators + prior), out = prec^-1 b + U^-1 eps.  U is d x d,
// work is 2d doubles.  Returns false when a pivot is not positive.
__device__ inline bool precision_mvnorm_dev(int d, const double *acc /* nacc(d): upper then rhs */,
                                            const double *prec0, const double *pbm, uint64_t key, uint32_t it,
                                            uint32_t stream, double *U, double *work, double *out)
{
    double *r = work, *o = work + d;
    int t = 0;
    for (int a = 0; a < d; ++a)
        for (int b = a; b < d; ++b) U[a * d + b] = acc[t++] + prec0[a * d + b];
    for (int a = 0; a < d; ++a) r[a] = acc[t++] + pbm[a];
    bool ok = true;
    for (int j = 0; j < d; ++j) {
        double s = U[j * d + j];
        for (int k = 0; k < j; ++k) s -= U[k * d + j] * U[k * d + j];
        if (!(s > 0.0)) ok = false;
        const double ujj = sqrt(s);
        U[j * d + j] = ujj;
        for (int i = j + 1; i < d; ++i) {
            double v = U[j * d + i];
            for (int k = 0; k < j; ++k) v -= U[k * d + j] * U[k * d + i];
            U[j * d + i] = v / ujj;
        }
    }
    for (int i = d - 1; i >= 0; --i) {  // o = U' eps + r ; eps_k drawn once each
        o[i] = 0.0;
    }
    for (int k = 0; k < d; ++k) {
        const double e = block_normal(key, (uint32_t)k, 0, it, stream);
        for (int i = k; i < d; ++i) o[i] += U[k * d + i] * e;
    }
    for (int i = 0; i < d; ++i) o[i] += r[i];
    for (int i = 0; i < d; ++i) {
        double v = o[i];
        for (int k = 0; k < i; ++k) v -= U[k * d + i] * o[k];
        o[i] = v / U[i * d + i];
    }
    for (int i = d - 1; i >= 0; --i) {
        double v = o[i];
        for (int k = i + 1; k < d; ++k) v -= U[i * d + k] * o[k];
        o[i] = v / U[i * d + i];
    }
    for (int i = 0; i < d; ++i) out[i] = o[i];
    return ok;
}

// =================================================================================================
__global__ void __launch_bounds__(256) k_omega_b(const Ctx *__restrict__ cp, ChainScalars *__restrict__ scs, Slot *__restrict__ slots, int chain_base)
{
    __shared__ double s_w[MAX_WAVES];
    const Ctx &c = *cp;
    const int chain = chain_base + blockIdx.y;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.next;
    if (blockIdx.x == 0 && threadIdx.x == 0) sc.cur = ctl;
    if (ctl.koff || ctl.it >= sc.it_stop) return;  // mid-solve chains skip straight to the Krylov steps
    const int n = c.n, i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t key = sc.key;
    const uint32_t it = ctl.it;
    double quad[1] = {0.0};
    if (i < n) {
        const size_t ci = (size_t)chain * n + i;
        const double *eta = c.eta + (size_t)chain * n;
        const double xb = xdot(c.Xt, n, i, sc.beta, c.p);
        const double eta_i = eta[i];
        Cursor cur(key, (uint32_t)i, it, STREAM_OMEGA_B);
        const double om = pg1_draw(cur, xb + eta_i);
        c.omega_b[ci] = om;
        const int slice = i >> 6, lane = i & 63;
        const int base = c.sell_ptr[slice], width = (c.sell_ptr[slice + 1] - base) >> 6;
        double qe = c.qdiag[i] * eta_i, u = 0.0;
        for (int k = 0; k < width; ++k) {
            const int j = c.sell_col[base + k * 64 + lane];
            const double v = c.sell_val[base + k * 64 + lane];
            qe += v * eta[j];
            const double w = -v;
            if (w > 0.0) {
                const uint32_t lo = (uint32_t)min(i, j), hi = (uint32_t)max(i, j);
                const double t = sqrt(w) * block_normal(key, lo, hi, it, STREAM_ETA_EDGE);
                u += (i < j) ? t : -t;
            }
        }
        quad[0] = eta_i * qe;
        const double kz = (double)c.z[ci] - 0.5;
        const double b = kz - om * xb;
        const double e = block_normal(key, (uint32_t)i, 0, it, STREAM_ETA_SITE);
        c.pre[ci] = b + sqrt(om) * e;
        c.uprior[ci] = u;
    }
    block_partials<1>(quad, s_w, part_buf(c, chain, 0), c.nb_n, blockIdx.x);
}

__global__ void __launch_bounds__(256) k_eta_init(const Ctx *__restrict__ cp, ChainScalars *__restrict__ scs, Slot *__restrict__ slots, int chain_base)
{
    __shared__ double s_w[MAX_WAVES], s_red[1], s_tau;
    const Ctx &c = *cp;
    const int chain = chain_base + blockIdx.y;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.cur;
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    const int n = c.n, i = blockIdx.x * blockDim.x + threadIdx.x;
    reduce_partials(part_buf(c, chain, 0), 1, c.nb_n, s_red);
    if (threadIdx.x == 0) {
        const double rate = 0.5 * s_red[0] + c.tau_rate;
        Cursor g(sc.key, 0u, ctl.it, STREAM_TAU);
        const double tau = (1.0 / rate) * std_gamma(g, c.tau_shape);
        s_tau = tau;
        if (blockIdx.x == 0) {
            sc.tau = tau;
            Slot s = {};
            slot_store(&slots[(size_t)chain * NSLOT], s);
        }
    }
    __syncthreads();
    const double tau = s_tau, st = sqrt(tau);
    double bsq[1] = {0.0};
    if (i < n) {
        const size_t ci = (size_t)chain * n + i;
        const double2 *X0 = c.Xv + (size_t)chain * n;
        const double y = c.pre[ci] + st * c.uprior[ci];
        c.rhs[ci] = y;
        const double om = c.omega_b[ci];
        const double2 x0 = X0[i];
        const double d = tau * c.qdiag[i] + om;
        double ax = d * x0.x, az = d * x0.y;
        const int slice = i >> 6, lane = i & 63;
        const int base = c.sell_ptr[slice], width = (c.sell_ptr[slice + 1] - base) >> 6;
        for (int k = 0; k < width; ++k) {
            const int j = c.sell_col[base + k * 64 + lane];
            const double a = tau * c.sell_val[base + k * 64 + lane];
            const double2 xj = X0[j];
            ax += a * xj.x;
            az += a * xj.y;
        }
        double2 r;
        r.x = y - ax;
        r.y = 1.0 - az;
        c.Rv[0][ci] = r;
        bsq[0] = r.x * r.x + r.y * r.y;
    }
    block_partials<1>(bsq, s_w, part_buf(c, chain, 1), c.nb_n, blockIdx.x);
}

// Sum-to-zero projection partials, taken by whichever kernel detects the end of the solve.
__device__ __forceinline__ void projection_partials(const Ctx &c, int chain, int i, double *lds)
{
    double v[2] = {0.0, 0.0};
    if (i < c.n) {
        const double2 x = c.Xv[(size_t)chain * c.n + i];
        v[0] = x.x;
        v[1] = x.y;
    }
    block_partials<2>(v, lds, c.part_proj + (size_t)chain * 2 * c.nb_n, c.nb_n, blockIdx.x);
}

// Step 2k-1: finish iteration k-1 (rotation, w and x updates) once beta_k is known, then the
// Lanczos product of iteration k:  y' = A v_k - (beta_k/beta_{k-1}) r2_{k-2},  alfa_k = v_k . y'.
__global__ void __launch_bounds__(256) k_minres_a(const Ctx *__restrict__ cp, ChainScalars *__restrict__ scs, Slot *__restrict__ slots, int chain_base, int k_launch)
{
    __shared__ double s_w[MAX_WAVES * 2], s_red[2];
    const Ctx &c = *cp;
    const int chain = chain_base + blockIdx.y;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.cur;
    if (ctl.it >= sc.it_stop) return;
    const int k = k_launch + (int)ctl.koff;  // Krylov step of THIS solve (continues across replays)
    Slot s = slot_load(&slots[(size_t)chain * NSLOT + ((2 * k - 2) & (NSLOT - 1))]);
    Slot *out = &slots[(size_t)chain * NSLOT + ((2 * k - 1) & (NSLOT - 1))];
    const bool writer = (blockIdx.x == 0 && threadIdx.x == 0);
    if (s.done) {
        if (writer) slot_store(out, s);
        return;
    }
    const int n = c.n, i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t ci = (size_t)chain * n + i;
    reduce_partials(part_buf(c, chain, 1), 1, c.nb_n, s_red);
    const double bsq = s_red[0];
    __syncthreads();
    const double beta_k = sqrt(bsq);
    const double eps = DBL_EPSILON;
    double part[2] = {0.0, 0.0};
    if (k == 1) {
        if (bsq == 0.0) {  // x0 already solves the system (minres.py: beta1 == 0)
            s.done = 1;
            s.istop = 0;
            s.itn = 0;
            if (writer) slot_store(out, s);
            projection_partials(c, chain, i, s_w);
            return;
        }
        s.beta1 = beta_k; s.oldb = 0.0; s.beta = beta_k; s.dbar = 0.0; s.epsln = 0.0;
        s.phibar = beta_k; s.rhs1 = beta_k; s.rhs2 = 0.0; s.tnorm2 = 0.0; s.gmax = 0.0;
        s.gmin = DBL_MAX; s.cs = -1.0; s.sn = 0.0; s.root = 0.0;
    } else {
        const int j = k - 1;  // iteration being completed
        const double beta_j = s.beta;
        s.oldb = beta_j;
        s.beta = beta_k;
        s.tnorm2 += s.alfa * s.alfa + beta_j * beta_j + beta_k * beta_k;
        if (j == 1 && beta_k / s.beta1 <= 10.0 * eps) s.istop = -1;
        const double oldeps = s.epsln;
        const double delta = s.cs * s.dbar + s.sn * s.alfa;
        const double gbar = s.sn * s.dbar - s.cs * s.alfa;
        s.epsln = s.sn * beta_k;
        s.dbar = -s.cs * beta_k;
        s.root = sqrt(gbar * gbar + s.dbar * s.dbar);
        double gamma = sqrt(gbar * gbar + beta_k * beta_k);
        gamma = fmax(gamma, eps);
        s.cs = gbar / gamma;
        s.sn = beta_k / gamma;
        const double phi = s.cs * s.phibar;
        s.phibar = s.sn * s.phibar;
        const double denom = 1.0 / gamma;
        s.gmax = fmax(s.gmax, gamma);
        s.gmin = fmin(s.gmin, gamma);
        const double zz = s.rhs1 / gamma;
        s.rhs1 = s.rhs2 - delta * zz;
        s.rhs2 = -s.epsln * zz;
        if (i < n) {
            const double sj = 1.0 / beta_j;
            const double2 rjm1 = c.Rv[(j - 1) % 3][ci];  // r2_{j-1}: v_j = s_j * r2_{j-1}
            double2 w1 = make_double2(0.0, 0.0), w2 = make_double2(0.0, 0.0);
            if (j - 2 >= 1) w1 = c.Wv[(j - 2) % 3][ci];
            if (j - 1 >= 1) w2 = c.Wv[(j - 1) % 3][ci];
            double2 w, x = c.Xv[ci];
            w.x = (sj * rjm1.x - oldeps * w1.x - delta * w2.x) * denom;
            w.y = (sj * rjm1.y - oldeps * w1.y - delta * w2.y) * denom;
            x.x = x.x + phi * w.x;
            x.y = x.y + phi * w.y;
            c.Wv[j % 3][ci] = w;
            c.Xv[ci] = x;
            part[1] = x.x * x.x + x.y * x.y;
        }
    }
    if (i < n) {
        const double tau = sc.tau;
        const double sk = 1.0 / beta_k;
        const double2 *r2 = c.Rv[(k - 1) % 3] + (size_t)chain * n;
        const double2 ri = r2[i];
        const double vx = sk * ri.x, vy = sk * ri.y;
        const double d = tau * c.qdiag[i] + c.omega_b[ci];
        double yx = d * vx, yy = d * vy;
        const int slice = i >> 6, lane = i & 63;
        const int base = c.sell_ptr[slice], width = (c.sell_ptr[slice + 1] - base) >> 6;
        for (int kk = 0; kk < width; ++kk) {
            const int j = c.sell_col[base + kk * 64 + lane];
            const double a = tau * c.sell_val[base + kk * 64 + lane];
            const double2 rj = r2[j];
            yx += a * (sk * rj.x);
            yy += a * (sk * rj.y);
        }
        if (k >= 2) {
            const double f = beta_k / s.oldb;
            const double2 r1 = c.Rv[(k - 2) % 3][ci];
            yx = yx - f * r1.x;
            yy = yy - f * r1.y;
        }
        c.Rv[k % 3][ci] = make_double2(yx, yy);
        part[0] = vx * yx + vy * yy;
    }
    if (writer) slot_store(out, s);
    block_partials<2>(part, s_w, part_buf(c, chain, 0), c.nb_n, blockIdx.x);
}

// Step 2k: stopping test of iteration k-1 (needs ||x_{k-1}||), then
//   r2_k = y' - (alfa_k/beta_k) r2_{k-1},  partial ||r2_k||^2.
__global__ void __launch_bounds__(256) k_minres_b(const Ctx *__restrict__ cp, ChainScalars *__restrict__ scs, Slot *__restrict__ slots, int chain_base, int k_launch)
{
    __shared__ double s_w[MAX_WAVES * 2], s_red[2];
    const Ctx &c = *cp;
    const int chain = chain_base + blockIdx.y;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.cur;
    if (ctl.it >= sc.it_stop) return;
    const int k = k_launch + (int)ctl.koff;
    Slot s = slot_load(&slots[(size_t)chain * NSLOT + ((2 * k - 1) & (NSLOT - 1))]);
    Slot *out = &slots[(size_t)chain * NSLOT + ((2 * k) & (NSLOT - 1))];
    const bool writer = (blockIdx.x == 0 && threadIdx.x == 0);
    if (s.done) {
        if (writer) slot_store(out, s);
        return;
    }
    const int n = c.n, i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t ci = (size_t)chain * n + i;
    reduce_partials(part_buf(c, chain, 0), 2, c.nb_n, s_red);
    const double alfa = s_red[0], xn2 = s_red[1];
    __syncthreads();
    if (k >= 2) {
        const int j = k - 1;
        const double eps = DBL_EPSILON, rtol = 1e-5;
        const double Anorm = sqrt(s.tnorm2);
        const double ynorm = sqrt(xn2);
        const double epsx = Anorm * ynorm * eps;
        const double rnorm = s.phibar;
        const double test1 = (ynorm == 0.0 || Anorm == 0.0) ? INFINITY : rnorm / (Anorm * ynorm);
        const double test2 = (Anorm == 0.0) ? INFINITY : s.root / Anorm;
        const double Acond = s.gmax / s.gmin;
        int istop = s.istop;
        if (istop == 0) {
            const double t1 = 1.0 + test1, t2 = 1.0 + test2;
            if (t2 <= 1.0) istop = 2;
            if (t1 <= 1.0) istop = 1;
            if ((long long)j >= c.maxiter) istop = 6;
            if (Acond >= 0.1 / eps) istop = 4;
            if (epsx >= s.beta1) istop = 3;
            if (test2 <= rtol) istop = 2;
            if (test1 <= rtol) istop = 1;
        }
        if (istop != 0) {
            s.istop = istop;
            s.itn = j;
            s.done = 1;
            if (writer) slot_store(out, s);
            projection_partials(c, chain, i, s_w);
            return;
        }
    }
    double bsq[1] = {0.0};
    if (i < n) {
        const double f = alfa / s.beta;
        const double2 yp = c.Rv[k % 3][ci], r2 = c.Rv[(k - 1) % 3][ci];
        double2 y;
        y.x = yp.x - f * r2.x;
        y.y = yp.y - f * r2.y;
        c.Rv[k % 3][ci] = y;
        bsq[0] = y.x * y.x + y.y * y.y;
    }
    s.alfa = alfa;
    s.itn = k;
    if (writer) slot_store(out, s);
    block_partials<1>(bsq, s_w, part_buf(c, chain, 1), c.nb_n, blockIdx.x);
}

template <int P>
__global__ void __launch_bounds__(256) k_beta_partial(const Ctx *__restrict__ cp, ChainScalars *__restrict__ scs, Slot *__restrict__ slots, int chain_base, int k_last_launch)
{
    __shared__ double s_w[MAX_WAVES * nacc(P)], s_red[2];
    const Ctx &c = *cp;
    const int chain = chain_base + blockIdx.y;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.cur;
    // the last Krylov kernel of this launch sequence was step k_last of the solve; its slot is final
    const int k_last = k_last_launch + (int)ctl.koff;
    const Slot *fin = &slots[(size_t)chain * NSLOT + ((2 * k_last) & (NSLOT - 1))];
    struct { int done, itn, istop; } s = {fin->done, fin->itn, fin->istop};
    const bool skip = ctl.it >= sc.it_stop;
    const bool stall_new = !skip && !s.done;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        Ctl m = ctl;
        m.koff = stall_new ? (uint32_t)k_last : 0u;  // carry the solve into the next replay
        sc.mid = m;
        if (stall_new) sc.carries += 1ull;
        if (!skip && s.done) {
            sc.minres_itn_last = s.itn;
            sc.krylov_total += (unsigned long long)s.itn;
            sc.krylov_sq_total += (unsigned long long)s.itn * (unsigned long long)s.itn;
            sc.solves += 1ull;
            if (s.istop == 6) sc.err = -3;  // OCC_E_MINRES (logit.py:91-92)
        }
    }
    if (skip || stall_new) return;
    const int n = c.n, i = blockIdx.x * blockDim.x + threadIdx.x;
    reduce_partials(c.part_proj + (size_t)chain * 2 * c.nb_n, 2, c.nb_n, s_red);
    const double a = -s_red[0] / s_red[1];
    __syncthreads();
    double acc[nacc(P)];
#pragma unroll
    for (int t = 0; t < nacc(P); ++t) acc[t] = 0.0;
    if (i < n) {
        const size_t ci = (size_t)chain * n + i;
        const double2 xz = c.Xv[ci];
        const double eta = xz.x + a * xz.y;
        c.eta[ci] = eta;
        const double om = c.omega_b[ci];
        const double tt = ((double)c.z[ci] - 0.5) - om * eta;
        double x[P];
#pragma unroll
        for (int aa = 0; aa < P; ++aa) x[aa] = c.Xt[(size_t)aa * n + i];
        int t = 0;
#pragma unroll
        for (int aa = 0; aa < P; ++aa) {
            const double xo = x[aa] * om;
#pragma unroll
            for (int bb = aa; bb < P; ++bb) acc[t++] = xo * x[bb];
        }
#pragma unroll
        for (int aa = 0; aa < P; ++aa) acc[t++] = x[aa] * tt;
    }
    block_partials<nacc(P)>(acc, s_w, part_buf(c, chain, 0), c.nb_n, blockIdx.x);
}

template <int Q>
__global__ void __launch_bounds__(256) k_omega_a(const Ctx *__restrict__ cp, ChainScalars *__restrict__ scs, Slot *__restrict__ slots, int chain_base)
{
    __shared__ double s_w[MAX_WAVES * nacc(Q)], s_red[NACC_MAX], s_U[MAXC * MAXC], s_work[2 * MAXC];
    const Ctx &c = *cp;
    const int chain = chain_base + blockIdx.y;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.mid;
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    const uint64_t key = sc.key;
    const uint32_t it = ctl.it;
    if (blockIdx.x == 0) {  // beta draw: only k_z needs it, one block suffices (logit.py:232)
        reduce_partials(part_buf(c, chain, 0), nacc(c.p), c.nb_n, s_red);
        if (threadIdx.x == 0) {
            const double *b_prec = c.hyp + Q * Q + Q, *b_pbm = b_prec + c.p * c.p;
            const bool ok = precision_mvnorm_dev(c.p, s_red, b_prec, b_pbm, key, it, STREAM_BETA, s_U, s_work, sc.beta);
            if (!ok) sc.err = -4;  // OCC_E_CHOLESKY
        }
    }
    const int R = c.R, r = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[nacc(Q)];
#pragma unroll
    for (int t = 0; t < nacc(Q); ++t) acc[t] = 0.0;
    if (r < R) {
        const int info = c.row_site[r];
        const int site = info & 0x7fffffff;
        const bool exists = (info < 0) || (c.z[(size_t)chain * c.n + site] != 0);
        if (exists) {
            double w[Q], wa = 0.0;
#pragma unroll
            for (int a = 0; a < Q; ++a) {
                w[a] = c.Wt[(size_t)a * R + r];
                wa += w[a] * sc.alpha[a];
            }
            Cursor cur(key, (uint32_t)r, it, STREAM_OMEGA_A);
            const double om = pg1_draw(cur, wa);
            c.omega_a[(size_t)chain * R + r] = om;
            const double tt = (double)c.yrow[r] - 0.5;
            int t = 0;
#pragma unroll
            for (int a = 0; a < Q; ++a) {
                const double wo = w[a] * om;
#pragma unroll
                for (int b = a; b < Q; ++b) acc[t++] = wo * w[b];
            }
#pragma unroll
            for (int a = 0; a < Q; ++a) acc[t++] = w[a] * tt;
        }
    }
    block_partials<nacc(Q)>(acc, s_w, part_buf(c, chain, 1), c.nb_r, blockIdx.x);
}

__global__ void __launch_bounds__(256) k_z(const Ctx *__restrict__ cp, ChainScalars *__restrict__ scs, Slot *__restrict__ slots, int chain_base)
{
    __shared__ double s_red[NACC_MAX], s_alpha[MAXC], s_U[MAXC * MAXC], s_work[2 * MAXC];
    const Ctx &c = *cp;
    const int Q = c.q;
    const int chain = chain_base + blockIdx.y;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.mid;
    const bool skip = ctl.koff || ctl.it >= sc.it_stop;
    const bool writer = (blockIdx.x == 0 && threadIdx.x == 0);
    if (writer) {
        Ctl nx = ctl;  // a mid-solve chain keeps its iteration number and its koff
        if (!skip) nx.it = ctl.it + 1;
        sc.next = nx;
    }
    if (skip) return;
    const uint64_t key = sc.key;
    const uint32_t it = ctl.it;
    reduce_partials(part_buf(c, chain, 1), nacc(Q), c.nb_r, s_red);
    if (threadIdx.x == 0) {
        const double *a_prec = c.hyp, *a_pbm = c.hyp + Q * Q;
        const bool ok = precision_mvnorm_dev(Q, s_red, a_prec, a_pbm, key, it, STREAM_ALPHA, s_U, s_work, s_alpha);
        const double *alpha = s_alpha;
        if (blockIdx.x == 0) {
            if (!ok) sc.err = -4;
            for (int a = 0; a < Q; ++a) sc.alpha[a] = alpha[a];
            const uint32_t rel = it - sc.it_base;
            if (c.rec != nullptr && rel >= sc.burnin && rel - sc.burnin < sc.keep) {
                const int P = c.p;
                double *row = c.rec + ((size_t)chain * sc.keep + (rel - sc.burnin)) * (size_t)(Q + P + 1);
                for (int a = 0; a < Q; ++a) row[a] = alpha[a];
                for (int a = 0; a < P; ++a) row[Q + a] = sc.beta[a];
                row[Q + P] = sc.tau;
            }
        }
    }
    __syncthreads();
    const int n = c.n, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int sidx = c.site_sidx[i];
    const bool not_surveyed = sidx < 0;
    if (!not_surveyed && c.obs_site[sidx]) return;  // detection seen: z stays 1 (base.py:116-118)
    const size_t ci = (size_t)chain * n + i;
    const double num1 = expit(xdot(c.Xt, n, i, sc.beta, c.p) + c.eta[ci]);
    double pr = num1;
    if (!not_surveyed) {
        double prod = 1.0;
        const int r0 = c.site_ptr[sidx], r1 = c.site_ptr[sidx + 1];
        for (int r = r0; r < r1; ++r) {
            double wa = 0.0;
            for (int a = 0; a < Q; ++a) wa += c.Wt[(size_t)a * c.R + r] * (-s_alpha[a]);
            const double e = expit(wa);
            prod = (r == r0) ? e : prod * e;
        }
        const double num = num1 * prod;
        pr = num / ((1.0 - num1) + num);
    }
    const double u = block_uniform(key, (uint32_t)i, 0, it, STREAM_Z);
    c.z[ci] = (u < pr) ? 1 : 0;
}

}  // namespace occ
