// occ_rsr.hpp -- the theta conditional of the reduced-rank model (LogitRSRGibbs, reference gibbs/logit.py:269-485) (gfx950).
//
// Spatial effects eta = K theta with K the n x m Moran-operator basis (m <= RSR_MAX_DIM columns, chosen on the host)
// and the reduced precision Qr = K'QK.  Every other conditional of the iteration is the ICAR sampler's and reads
// eta (the reference's `spatial`); only tau and the spatial conditional change (logit.py:206-209 with fixed.Q = Qr,
// and 465-485):
//   rate   = 1/2 theta' Qr theta + tau_rate,  tau ~ Gamma
//   prec   = K' diag(omega_b) K + tau Qr                                      (m x m, dense, symmetric)
//   rhs    = K'(k - omega_b X beta + sqrt(omega_b) eps1) + sqrt(tau) E eps2     (E E' = Qr; eps1 per site, eps2 per column)
//   theta  = prec^-1 rhs  (upper Cholesky in LDS, two triangular solves),  eta = K theta
// Four kernels, all chains batched on blockIdx.y, every sum in a fixed order (no atomics):
//   k_rsr_rhs      K'u, u_i = k_i - omega_i x_i'beta + sqrt(omega_i) eps1_i: 256 sites staged in LDS per trip, one
//                  thread per column, K read coalesced along the columns
//   k_rsr_gram     K' diag(omega) K by 16 x 16 output tiles (upper triangle of tiles), 64 sites staged in LDS per trip
//   k_rsr_solve    one workgroup per chain: tau, prec and rhs assembled in LDS, Cholesky, solves, theta
//   k_rsr_spatial  eta = K theta from the transposed copy of K (coalesced along the sites)
// plus k_beta_partial_rsr: the partial sums of beta's system without the ICAR solve's projection step.
#pragma once
#include "occ_kernels.hpp"

namespace occ {

constexpr int RSR_MAX_DIM = 128;  // m x m doubles of LDS for the Cholesky factor: 128 KB of the CU's 160 KB
constexpr uint32_t STREAM_RSR = 9;

struct RsrArgs {
    int n, m, p, C;
    const double *K;    // [n][m]
    const double *Kt;   // [m][n]
    const double *Qr;   // [m][m]
    const double *E;    // [m][m]
    const double *Xt;   // [p][n]
    const uint8_t *z;   // [C][n]
    const double *omega_b[2], *enorm[2];
    double *theta;      // [C][m]
    double *gram;       // [C][m][m] (upper triangle of 16 x 16 tiles written)
    double *rhs;        // [C][m]
    double *eta;        // [C][n]
    double tau_rate, tau_shape;
    ChainScalars *scs;
};

__global__ void __launch_bounds__(256) k_rsr_rhs(const RsrArgs a, int e)
{
    __shared__ double s_u[256];
    const int chain = blockIdx.y, col = blockIdx.x * 256 + threadIdx.x;
    const ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    const uint32_t it = ctl.it;
    const size_t co = (size_t)chain * a.n;
    double acc = 0.0;
    for (int i0 = 0; i0 < a.n; i0 += 256) {
        const int i = i0 + (int)threadIdx.x;
        double u = 0.0;
        if (i < a.n) {
            const double om = a.omega_b[it & 1][co + i];
            const double xb = xdot(a.Xt, a.n, i, sc.beta, a.p);
            u = fma(sqrt(om), a.enorm[it & 1][co + i], fma(-om, xb, (double)a.z[co + i] - 0.5));
        }
        __syncthreads();
        s_u[threadIdx.x] = u;
        __syncthreads();
        if (col < a.m) {
            const int cnt = min(256, a.n - i0);
            for (int ii = 0; ii < cnt; ++ii) acc = fma(a.K[(size_t)(i0 + ii) * a.m + col], s_u[ii], acc);
        }
    }
    if (col < a.m) a.rhs[(size_t)chain * a.m + col] = acc;
}

__global__ void __launch_bounds__(256) k_rsr_gram(const RsrArgs a, int e)
{
    __shared__ double s_a[64][17], s_c[64][17], s_om[64];
    const int chain = blockIdx.y;
    const ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    const int T = (a.m + 15) / 16;
    // upper triangle of tiles, enumerated row by row
    int ta = 0, rem = (int)blockIdx.x;
    while (rem >= T - ta) { rem -= T - ta; ++ta; }
    const int tc = ta + rem;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // thread (ty, tx) owns G[a0 + ty][c0 + tx]
    const int a0 = ta * 16, c0 = tc * 16;
    const size_t co = (size_t)chain * a.n;
    const double *om = a.omega_b[ctl.it & 1] + co;
    double acc = 0.0;
    for (int i0 = 0; i0 < a.n; i0 += 64) {
        __syncthreads();
        for (int t = threadIdx.x; t < 64 * 16; t += 256) {
            const int ii = t >> 4, cc = t & 15, i = i0 + ii;
            s_a[ii][cc] = (i < a.n && a0 + cc < a.m) ? a.K[(size_t)i * a.m + a0 + cc] : 0.0;
            s_c[ii][cc] = (i < a.n && c0 + cc < a.m) ? a.K[(size_t)i * a.m + c0 + cc] : 0.0;
        }
        if (threadIdx.x < 64) s_om[threadIdx.x] = (i0 + (int)threadIdx.x < a.n) ? om[i0 + threadIdx.x] : 0.0;
        __syncthreads();
#pragma unroll 8
        for (int ii = 0; ii < 64; ++ii) acc = fma(s_a[ii][ty] * s_om[ii], s_c[ii][tx], acc);
    }
    if (a0 + ty < a.m && c0 + tx < a.m) a.gram[((size_t)chain * a.m + a0 + ty) * a.m + c0 + tx] = acc;
}

// One workgroup per chain.  Dynamic LDS: U[m][m] (upper Cholesky factor in place), then four m-vectors.
__global__ void __launch_bounds__(256) k_rsr_solve(const RsrArgs a, int e)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double s_scalar[2];
    __shared__ int s_bad;
    const int chain = blockIdx.y, m = a.m, tid = threadIdx.x;
    ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    const uint32_t it = ctl.it;
    double *U = smem, *th = smem + (size_t)m * m, *rh = th + m, *yv = rh + m, *tmp = yv + m;
    double *theta = a.theta + (size_t)chain * m;
    // ---- tau: rate = 1/2 theta' Qr theta + tau_rate (theta of the previous iteration)
    for (int t = tid; t < m; t += 256) th[t] = theta[t];
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (int r = tid; r < m; r += 256) {
        double t = 0.0;
        for (int c = 0; c < m; ++c) t = fma(a.Qr[(size_t)r * m + c], th[c], t);
        tmp[r] = t;
    }
    __syncthreads();
    if (tid == 0) {
        double quad = 0.0;
        for (int r = 0; r < m; ++r) quad = fma(th[r], tmp[r], quad);
        const double rate = 0.5 * quad + a.tau_rate;
        Cursor g(sc.key, 0u, it, STREAM_TAU);
        const double tau = (1.0 / rate) * std_gamma(g, a.tau_shape);
        sc.tau = tau;
        s_scalar[0] = tau;
        s_scalar[1] = sqrt(tau);
    }
    __syncthreads();
    const double tau = s_scalar[0], st = s_scalar[1];
    // ---- prec = K'OK + tau Qr (upper triangle), rhs = K'u + sqrt(tau) E eps2
    const double *G = a.gram + (size_t)chain * m * m;
    for (int t = tid; t < m * m; t += 256) {
        const int r = t / m, c = t % m;
        U[t] = (c >= r) ? fma(tau, a.Qr[t], G[t]) : 0.0;
    }
    for (int j = tid; j < m; j += 256) tmp[j] = block_normal(sc.key, (uint32_t)j, 0, it, STREAM_RSR);
    __syncthreads();
    for (int r = tid; r < m; r += 256) {
        double t = 0.0;
        for (int j = 0; j < m; ++j) t = fma(a.E[(size_t)r * m + j], tmp[j], t);
        rh[r] = fma(st, t, a.rhs[(size_t)chain * m + r]);
    }
    __syncthreads();
    // ---- upper Cholesky, the oracle's (left-looking) order: column j of U' from the rows above
    for (int j = 0; j < m; ++j) {
        if (tid == 0) {
            double s = U[(size_t)j * m + j];
            for (int k = 0; k < j; ++k) s = fma(-U[(size_t)k * m + j], U[(size_t)k * m + j], s);
            if (!(s > 0.0)) s_bad = 1;
            s_scalar[0] = sqrt(s);
        }
        __syncthreads();
        const double ujj = s_scalar[0];
        for (int i = j + 1 + tid; i < m; i += 256) {
            double t = U[(size_t)j * m + i];
            for (int k = 0; k < j; ++k) t = fma(-U[(size_t)k * m + j], U[(size_t)k * m + i], t);
            U[(size_t)j * m + i] = t / ujj;
        }
        if (tid == 0) U[(size_t)j * m + j] = ujj;
        __syncthreads();
    }
    if (s_bad) {
        if (tid == 0) sc.err = -4;  // OCC_E_CHOLESKY
        return;
    }
    // ---- U'y = rhs (forward), U theta = y (backward): one lane, m^2 operations (m <= 128)
    if (tid == 0) {
        for (int i = 0; i < m; ++i) {
            double t = rh[i];
            for (int k = 0; k < i; ++k) t = fma(-U[(size_t)k * m + i], yv[k], t);
            yv[i] = t / U[(size_t)i * m + i];
        }
        for (int i = m - 1; i >= 0; --i) {
            double t = yv[i];
            for (int k = i + 1; k < m; ++k) t = fma(-U[(size_t)i * m + k], th[k], t);
            th[i] = t / U[(size_t)i * m + i];
        }
    }
    __syncthreads();
    for (int t = tid; t < m; t += 256) theta[t] = th[t];
}

__global__ void __launch_bounds__(256) k_rsr_spatial(const RsrArgs a, int e)
{
    __shared__ double s_th[RSR_MAX_DIM];
    const int chain = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    for (int t = threadIdx.x; t < a.m; t += 256) s_th[t] = a.theta[(size_t)chain * a.m + t];
    __syncthreads();
    if (i >= a.n) return;
    double acc = 0.0;
    for (int c = 0; c < a.m; ++c) acc = fma(a.Kt[(size_t)c * a.n + i], s_th[c], acc);
    a.eta[(size_t)chain * a.n + i] = acc;
}

// Partial sums of beta's system from eta = K theta (k_beta_partial without the ICAR solve's projection).
template <int P>
__global__ void __launch_bounds__(256) k_beta_partial_rsr(OCC_KARGS)
{
    const Ctx &c = *cp;
    const Tile tile = tile_of_block(chain_base);
    const int chain = tile.chain, blk = tile.blk;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.ctl[e];
    const bool skip = ctl.koff || ctl.it >= sc.it_stop;
    if (blk == 0 && threadIdx.x == 0) {
        Ctl m = ctl;
        m.koff = 0u;
        sc.mid[e] = m;
        if (!skip) {
            sc.minres_itn_last = 0;
            sc.solves += 1ull;
        }
    }
    if (skip) return;
    const int n = c.n, i = blk * blockDim.x + threadIdx.x;
    double acc[nacc(P)];
#pragma unroll
    for (int t = 0; t < nacc(P); ++t) acc[t] = 0.0;
    if (i < n) {
        const size_t ci = (size_t)chain * n + i;
        const double eta = c.eta[ci];
        const double om = c.omega_b[ctl.it & 1][ci];
        const double tt = beta_rhs_term(om, eta, (double)c.z[ci]);
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
    block_partials<nacc(P)>(acc, c.part_beta + (size_t)chain * nacc(P) * c.nb_n, c.nb_n, blk);
}

}  // namespace occ
