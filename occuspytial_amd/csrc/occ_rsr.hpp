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
//   k_rsr_rhs      K'u, u_i = k_i - omega_i x_i'beta + sqrt(omega_i) eps1_i: one workgroup per 256 sites stages u in
//                  LDS, one thread per column reads K coalesced along the columns; partial sums per workgroup
//   k_rsr_gram     K' diag(omega) K on the matrix cores: one wave per 16 x 16 output tile (upper triangle of tiles),
//                  v_mfma_f64_16x16x4_f64 over four sites at a time, operands straight from global memory
//   k_rsr_solve    one workgroup per chain: tau, prec and rhs assembled in LDS, right-looking Cholesky, the two
//                  triangular solves by one wave, theta
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
    const double *Et;   // [m][m], the eigenfactor E of Qr (E E' = Qr) TRANSPOSED: Et[j][r] = E[r][j]
    const double *Xt;   // [p][n]
    const uint8_t *z;   // [C][n]
    const double *omega_b[2], *enorm[2];
    double *theta;      // [C][m]
    double *gram;       // [C][m][m] (upper triangle of 16 x 16 tiles written)
    double *rhs;        // [C][nchunk][m] partial sums of K'u per 256-site workgroup
    int nchunk;
    double *eta;        // [C][n]
    double tau_rate, tau_shape;
    ChainScalars *scs;
};

__global__ void __launch_bounds__(256) k_rsr_rhs(const RsrArgs a, int e)
{
    __shared__ double s_u[256];
    const int chain = blockIdx.y, chunk = blockIdx.x, i0 = chunk * 256;
    const ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    const uint32_t it = ctl.it;
    const size_t co = (size_t)chain * a.n;
    const int i = i0 + (int)threadIdx.x;
    double u = 0.0;
    if (i < a.n) {
        const double om = a.omega_b[it & 1][co + i];
        const double xb = xdot(a.Xt, a.n, i, sc.beta, a.p);
        u = fma(sqrt(om), a.enorm[it & 1][co + i], fma(-om, xb, (double)a.z[co + i] - 0.5));
    }
    s_u[threadIdx.x] = u;
    __syncthreads();
    const int cnt = min(256, a.n - i0);
    for (int col = threadIdx.x; col < a.m; col += 256) {
        double acc = 0.0;
        for (int ii = 0; ii < cnt; ++ii) acc = fma(a.K[(size_t)(i0 + ii) * a.m + col], s_u[ii], acc);
        a.rhs[((size_t)chain * a.nchunk + chunk) * a.m + col] = acc;
    }
}

typedef double v4d __attribute__((ext_vector_type(4)));

// G = K' diag(omega) K, one wave per 16 x 16 tile of the upper triangle.  v_mfma_f64_16x16x4_f64 multiplies a 16 x 4
// by a 4 x 16 block: here A = (K[:, a0:a0+16] scaled by omega)' and B = K[:, c0:c0+16] over four consecutive sites.
// Operand layout (wave64, checked against numpy on the device): lane l carries A[l % 16][l / 16] and B[l / 16][l % 16];
// it receives D[4 v + l / 16][l % 16], v = 0..3.  Both operands are 16 consecutive doubles of a row of K per site: 128-byte coalesced loads, no LDS.
__global__ void __launch_bounds__(256) k_rsr_gram(const RsrArgs a, int e)
{
    __shared__ double s_part[3][64][4];  // the partial tiles of waves 1..3
    const int chain = blockIdx.y;
    const ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    const int T = (a.m + 15) / 16;
    int ta = 0, rem = (int)blockIdx.x;  // upper triangle of tiles, enumerated row by row
    while (rem >= T - ta) { rem -= T - ta; ++ta; }
    const int tc = ta + rem;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lc = lane & 15, lk = lane >> 4;
    const int ca = ta * 16 + lc, cc = tc * 16 + lc;
    const bool va = ca < a.m, vc = cc < a.m;
    const double *om = a.omega_b[ctl.it & 1] + (size_t)chain * a.n;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    // the four waves of the workgroup take every fourth block of 32 sites; eight MFMAs per trip, their loads in
    // flight together
    for (int i0 = wave * 32; i0 < a.n; i0 += 128) {
        double av[8], bv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int i = i0 + 4 * t + lk;
            const bool vi = i < a.n;
            const double w = vi ? om[i] : 0.0;
            av[t] = (vi && va) ? a.K[(size_t)i * a.m + ca] * w : 0.0;
            bv[t] = (vi && vc) ? a.K[(size_t)i * a.m + cc] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[t], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) s_part[wave - 1][lane][v] = acc[v];
    }
    __syncthreads();
    if (wave != 0) return;
    double *G = a.gram + (size_t)chain * a.m * a.m;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const double t = ((acc[v] + s_part[0][lane][v]) + s_part[1][lane][v]) + s_part[2][lane][v];  // fixed order
        const int r = ta * 16 + 4 * v + lk;
        if (r < a.m && vc) G[(size_t)r * a.m + cc] = t;
    }
}

// One workgroup per chain.  Dynamic LDS: U[m][m] (upper Cholesky factor in place), then four m-vectors.
__global__ void __launch_bounds__(256) k_rsr_solve(const RsrArgs a, int e)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double s_scalar[2];
    __shared__ int s_bad;
    const int chain = blockIdx.y, m = a.m, tid = threadIdx.x, nt = blockDim.x;
    ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    const uint32_t it = ctl.it;
    double *U = smem, *th = smem + (size_t)m * m, *rh = th + m, *yv = rh + m, *tmp = yv + m;
    double *theta = a.theta + (size_t)chain * m;
    // ---- tau: rate = 1/2 theta' Qr theta + tau_rate (theta of the previous iteration)
    for (int t = tid; t < m; t += nt) th[t] = theta[t];
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (int r = tid; r < m; r += nt) {
        double t = 0.0;
        for (int c = 0; c < m; ++c) t = fma(a.Qr[(size_t)c * m + r], th[c], t);  // Qr is symmetric: coalesced along r
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
    for (int t = tid; t < m * m; t += nt) {
        const int r = t / m, c = t % m;
        U[t] = (c >= r) ? fma(tau, a.Qr[t], G[t]) : 0.0;
    }
    for (int j = tid; j < m; j += nt) tmp[j] = block_normal(sc.key, (uint32_t)j, 0, it, STREAM_RSR);
    __syncthreads();
    for (int r = tid; r < m; r += nt) {
        double t = 0.0;
        for (int j = 0; j < m; ++j) t = fma(a.Et[(size_t)j * m + r], tmp[j], t);  // E[r][j], read from the transposed copy
        double ku = 0.0;  // K'u: the workgroups' partial sums in chunk order
        for (int ch = 0; ch < a.nchunk; ++ch) ku += a.rhs[((size_t)chain * a.nchunk + ch) * m + r];
        rh[r] = fma(st, t, ku);
    }
    __syncthreads();
    // ---- upper Cholesky in place, right-looking: scale row j, rank-one update of the trailing block; two
    // workgroup barriers per column.  An entry receives its updates for j = 0, 1, ... in turn: the order of the
    // oracle's dot products.  Threads form a 16 x 16 grid over the trailing block (no index divisions).
    double *row = yv;  // the scaled row j
    const int ty = tid >> 4, tx = tid & 15;
    for (int j = 0; j < m; ++j) {
        const double piv = U[(size_t)j * m + j];
        if (tid == 0 && !(piv > 0.0)) s_bad = 1;
        const double ujj = sqrt(piv);
        for (int i = j + 1 + tid; i < m; i += nt) {
            const double v = U[(size_t)j * m + i] / ujj;
            row[i] = v;
            U[(size_t)j * m + i] = v;
        }
        __syncthreads();
        if (tid == 0) U[(size_t)j * m + j] = ujj;
        for (int k = j + 1 + ty; k < m; k += 16) {
            const double rk = row[k];
            for (int i = k + tx; i < m; i += 16) U[(size_t)k * m + i] = fma(-rk, row[i], U[(size_t)k * m + i]);
        }
        __syncthreads();
    }
    if (s_bad) {
        if (tid == 0) sc.err = -4;  // OCC_E_CHOLESKY
        return;
    }
    // ---- U'y = rhs (forward), U theta = y (backward) by ONE wave, column-oriented (lanes own entries lane and
    // lane + 64; m <= 128): after y_i is known every later entry subtracts its term -- no workgroup barrier
    if (tid < 64) {
        double r0 = (tid < m) ? rh[tid] : 0.0, r1 = (tid + 64 < m) ? rh[tid + 64] : 0.0;
        for (int i = 0; i < m; ++i) {
            const double num = (i < 64) ? __shfl(r0, i) : __shfl(r1, i - 64);
            const double yi = num / U[(size_t)i * m + i];
            if (tid == (i & 63)) { if (i < 64) r0 = yi; else r1 = yi; }
            if (tid > i && tid < m) r0 = fma(-U[(size_t)i * m + tid], yi, r0);
            if (tid + 64 > i && tid + 64 < m) r1 = fma(-U[(size_t)i * m + tid + 64], yi, r1);
        }
        for (int i = m - 1; i >= 0; --i) {
            const double num = (i < 64) ? __shfl(r0, i) : __shfl(r1, i - 64);
            const double ti = num / U[(size_t)i * m + i];
            if (tid == (i & 63)) { if (i < 64) r0 = ti; else r1 = ti; }
            if (tid < i) r0 = fma(-U[(size_t)tid * m + i], ti, r0);
            if (tid + 64 < i) r1 = fma(-U[(size_t)(tid + 64) * m + i], ti, r1);
        }
        if (tid < m) th[tid] = r0;
        if (tid + 64 < m) th[tid + 64] = r1;
    }
    __syncthreads();
    for (int t = tid; t < m; t += nt) theta[t] = th[t];
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
