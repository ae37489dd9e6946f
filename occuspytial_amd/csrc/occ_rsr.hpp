// occ_rsr.hpp -- the theta conditional of the reduced-rank model (LogitRSRGibbs, reference gibbs/logit.py:269-485) (gfx950).
//
// Spatial effects eta = K theta with K the n x m Moran-operator basis (m <= RSR_MAX_DIM columns, chosen on the host)
// and the reduced precision Qr = K'QK.  Every other conditional of the iteration is the ICAR sampler's and reads
// eta (the reference's `spatial`); only tau and the spatial conditional change (logit.py:206-209 with fixed.Q = Qr,
// and 465-485):
//   rate   = 1/2 theta' Qr theta + tau_rate,  tau ~ Gamma
//   prec   = K' diag(omega_b) K + tau Qr                                      (m x m, dense, symmetric)
//   rhs    = K'(k - omega_b X beta + sqrt(omega_b) eps1) + sqrt(tau) E eps2     (E E' = Qr; eps1 per site, eps2 per column)
//   theta  = prec^-1 rhs  (upper Cholesky with the matrix in registers, two triangular solves),  eta = K theta
// Three kernels, all chains batched on blockIdx.y, every sum in a fixed order (no atomics):
//   k_rsr_gram     K' [diag(omega_b) K | u] on the matrix cores, u_i = k_i - omega_i x_i'beta + sqrt(omega_i) eps1_i: the
//                  Gram matrix and K'u in one pass over K; one workgroup of 16 waves per 16 x 16 output tile (upper
//                  triangle of tiles), v_mfma_f64_16x16x4_f64 over four sites at a time, operands from global memory
//   k_rsr_solve    one workgroup per chain: tau, prec (in registers) and rhs assembled, right-looking Cholesky with
//                  the forward substitution fused, back substitution by one wave, theta
//   k_rsr_eta_beta eta = K theta from the transposed copy of K (coalesced along the sites) and, in the same thread,
//                  the partial sums of beta's system (without the ICAR solve's projection step)
#pragma once
#include "occ_kernels.hpp"

namespace occ {

constexpr int RSR_MAX_DIM = 128;  // m x m doubles of LDS for the Cholesky factor: 128 KB of the CU's 160 KB
constexpr int RSR_BIG_MAX = 2048; // beyond RSR_MAX_DIM: the factor lives in global memory, factorised panel by panel (k_rsrb_*)
constexpr int RSR_PANEL = 32;     // ... columns per panel
constexpr uint32_t STREAM_RSR = 9;

struct RsrArgs {
    int n, m, p, C, ldk;
    const double *K;    // [n][ldk], ldk = 16 ceil(m / 16): rows zero-padded to whole 128-byte lines
    const double *Kt;   // [m][n]
    const double *Qr;   // [m][m]
    const double *Et;   // [m][m], the eigenfactor E of Qr (E E' = Qr) TRANSPOSED: Et[j][r] = E[r][j]
    const double *Xt;   // [p][n]
    const uint8_t *z;   // [C][n]
    const double *omega_b[2], *enorm[2];
    double *theta;      // [C][m]
    double *gram;       // [C][m][m] (upper triangle of 16 x 16 tiles written)
    double *rhs;        // [C][nchunk][m] K'u (nchunk = 1: k_rsr_gram writes the finished sums)
    int nchunk;
    double *eta;        // [C][n]
    // m > RSR_MAX_DIM only (k_rsrb_*): E row-major, the noise of the prior term, {tau, sqrt(tau)}, the finished right-hand side
    const double *E;    // [m][m]
    double *big_eps;    // [C][m]
    double *big_scal;   // [C][2] (unused since round 3)
    double *big_quad;   // [C][ceil(m / 64)] theta' Qr theta by slices of 64 rows (k_rsrb_tau -> k_rsrb_assemble)
    double *big_rhs;    // [C][m]
    double *big_dfac;   // [C][ceil(m / RSR_PANEL)][RSR_PANEL][RSR_PANEL] the factored diagonal blocks (k_rsrb_panel -> k_rsrb_solve)
    double tau_rate, tau_shape;
    ChainScalars *scs;
    unsigned *sync;     // hand-over counters of the two streams (Ctx::sync), or null
};

typedef double v4d __attribute__((ext_vector_type(4)));

#ifdef OCC_SOLVE_STAMPS
#define GRAM_STAMP(pt) if (chain == 0 && threadIdx.x == 0 && (blockIdx.x == 0 || (int)blockIdx.x == ntile)) g_solve_stamps[16 + (blockIdx.x == 0 ? 0 : 4) + (pt)] = wall_clock64();
#else
#define GRAM_STAMP(pt)
#endif
#ifndef OCC_GRAM_WAVES
#define OCC_GRAM_WAVES 16
#endif
constexpr int GRAM_WAVES = OCC_GRAM_WAVES;  // waves per workgroup of k_rsr_gram
constexpr int GRAM_UCHUNK = 2048;  // sites of u staged in LDS at a time by the K'u workgroups

// Workgroups of k_rsr_gram per chain: the T (T + 1) / 2 tiles of the Gram matrix's upper triangle, T = ceil(m / 16),
// then T workgroups for K'u (one per block of 16 columns of K).
__host__ __device__ inline int rsr_gram_tiles(int m)
{
    const int T = (m + 15) / 16;
    return T * (T + 1) / 2 + T;
}

// G = K' diag(omega) K and K'u, u_i = k_i - omega_i x_i'beta + sqrt(omega_i) eps1_i, one workgroup of 16 waves per
// 16 x 16 tile of G's upper triangle and per 16 entries of K'u.  v_mfma_f64_16x16x4_f64 multiplies a 16 x 4 by a
// 4 x 16 block: here A = K[:, a0:a0+16]' and B = (diag(omega) K)[:, c0:c0+16] over four consecutive sites.  Operand
// layout (wave64, checked against numpy on the device): lane l carries A[l % 16][l / 16] and B[l / 16][l % 16]; it
// receives D[4 v + l / 16][l % 16], v = 0..3.  Both operands are 16 consecutive doubles of a row of K per site:
// 128-byte coalesced loads, no LDS.  A wave takes every 16th block of 32 sites (eight MFMAs), the loads of its next
// block issued before the MFMAs of this one; the 16 partial tiles are added in wave order through LDS.
// The K'u workgroups stage u (all 1024 threads, coalesced) in LDS and run the same loop with B = [u 0 ... 0].
__device__ __forceinline__ void rsr_gram_load(const RsrArgs &a, const double *om, int i0, int lk, int ca, int cc, double (&av)[8], double (&bv)[8])
{
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int i = i0 + 4 * t + lk;
        const bool vi = i < a.n;
        const int ii = vi ? i : 0;  // unconditional loads; the padding columns of K hold zeros
        const double w = vi ? om[ii] : 0.0;
        av[t] = a.K[(size_t)ii * a.ldk + ca];
        bv[t] = a.K[(size_t)ii * a.ldk + cc] * w;
    }
}

// sync_on bit 0: hand-overs by device counters; bit 1: ONLY the K'u workgroups (grid = T per chain): the Gram matrix itself
// comes from k_rsr_gram32 (large bases)
__global__ void __launch_bounds__(64 * GRAM_WAVES) k_rsr_gram(const RsrArgs a, int e, int sync_flags)
{
    const int sync_on = sync_flags & 1, u_only = sync_flags & 2;
    __shared__ double s_part[GRAM_WAVES - 1][64][4];  // the partial tiles of waves 1..15
    __shared__ double s_u[GRAM_UCHUNK];
    __shared__ int s_noise_ok;
    const int chain = blockIdx.y;
    // first kernel of the main stream's sequence (as k_iter in the ICAR model): k_z_ob of the previous sequence is
    // complete, the side stream may start this sequence.  Said before anything can return or wait.
    const bool synced = sync_on && a.sync != nullptr;
    if (synced && !u_only && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) sync_set(a.sync + SYNC_MAIN, a.sync[SYNC_MAIN_SEQ + e]);
    const ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    const int T = (a.m + 15) / 16, ntile = T * (T + 1) / 2;
    const int bx = (int)blockIdx.x + (u_only ? ntile : 0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lc = lane & 15, lk = lane >> 4;
    const size_t co = (size_t)chain * a.n;
    const double *om = a.omega_b[ctl.it & 1] + co;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    int ta, tc;
    const bool utile = bx >= ntile;
    GRAM_STAMP(0)
#ifdef OCC_SOLVE_STAMPS
    if (chain == 0 && blockIdx.x == 0 && lane == 0) g_solve_stamps[40 + wave] = wall_clock64();
#endif
    if (!utile) {
        ta = 0;
        int rem = bx;  // upper triangle of tiles, enumerated row by row
        while (rem >= T - ta) { rem -= T - ta; ++ta; }
        tc = ta + rem;
        const int ca = ta * 16 + lc, cc = tc * 16 + lc;
        double av[8], bv[8], an[8], bn[8];
        int i0 = wave * 32;
        if (i0 < a.n) rsr_gram_load(a, om, i0, lk, ca, cc, av, bv);
        for (; i0 < a.n; i0 += 32 * GRAM_WAVES) {
            const int i1 = i0 + 32 * GRAM_WAVES;
            if (i1 < a.n) rsr_gram_load(a, om, i1, lk, ca, cc, an, bn);
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[t], acc, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                av[t] = an[t];
                bv[t] = bn[t];
            }
        }
    } else {
        ta = bx - ntile;
        tc = -1;
        const int ca = ta * 16 + lc;
        const double *en = a.enorm[ctl.it & 1] + co;
        const uint8_t *z = a.z + co;
        if (synced) {  // the noise of this iteration comes from the side stream's previous sequence
            if (threadIdx.x == 0) s_noise_ok = sync_wait(a.sync, SYNC_NOISE, a.sync[SYNC_MAIN_SEQ + e]) ? 1 : 0;
            __syncthreads();
            if (!s_noise_ok) {
                if (threadIdx.x == 0) a.scs[chain].err = -2;  // OCC_E_HIP: the side stream never arrived
                return;
            }
        }
        double beta[MAXC];
#pragma unroll
        for (int j = 0; j < MAXC; ++j) beta[j] = (j < a.p) ? sc.beta[min(j, a.p - 1)] : 0.0;
        for (int c0 = 0; c0 < a.n; c0 += GRAM_UCHUNK) {
            const int cnt = min(GRAM_UCHUNK, a.n - c0);
            for (int t = threadIdx.x; t < cnt; t += 64 * GRAM_WAVES) {
                const int i = c0 + t;
                const double w = om[i], ev = en[i], zv = (double)z[i];
                double x[MAXC];
#pragma unroll
                for (int j = 0; j < MAXC; ++j) x[j] = a.Xt[(size_t)min(j, a.p - 1) * a.n + i];  // all loads out before the sum
                double xb = 0.0;
#pragma unroll
                for (int j = 0; j < MAXC; ++j) xb = fma(x[j], beta[j], xb);
                s_u[t] = fma(sqrt(w), ev, fma(-w, xb, zv - 0.5));
            }
            __syncthreads();
            if (c0 == 0) { GRAM_STAMP(2) }
            for (int i0 = wave * 32; i0 < cnt; i0 += 32 * GRAM_WAVES) {
                double av[8], bv[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int il = i0 + 4 * t + lk;
                    const bool vi = il < cnt;
                    const double ka = a.K[(size_t)(c0 + (vi ? il : 0)) * a.ldk + ca];
                    av[t] = vi ? ka : 0.0;
                    bv[t] = (vi && lc == 0) ? s_u[vi ? il : 0] : 0.0;
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[t], acc, 0, 0, 0);
            }
            __syncthreads();
        }
    }
    GRAM_STAMP(1)
#ifdef OCC_SOLVE_STAMPS
    if (chain == 0 && blockIdx.x == 0 && lane == 0) g_solve_stamps[24 + wave] = wall_clock64();
#endif
    if (wave > 0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) s_part[wave - 1][lane][v] = acc[v];
    }
    __syncthreads();
    if (wave != 0) return;
    GRAM_STAMP(2)
    double *G = a.gram + (size_t)chain * a.m * a.m;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        double t = acc[v];
#pragma unroll
        for (int w = 0; w < GRAM_WAVES - 1; ++w) t += s_part[w][lane][v];  // fixed order
        const int r = ta * 16 + 4 * v + lk, cc = tc * 16 + lc;
        if (!utile && r < a.m && cc < a.m) G[(size_t)r * a.m + cc] = t;
        if (utile && r < a.m && lc == 0) a.rhs[(size_t)chain * a.m + r] = t;
    }
    GRAM_STAMP(3)
}

// ---- the Gram matrix for LARGE bases (m > RSR_MAX_DIM): 32 x 32 output blocks ---------------------------------------------
// k_rsr_gram above reads 2 x 16 columns of K per v_mfma_f64_16x16x4_f64: 2 flop per byte from the L2, exactly the ratio of a
// CU's L2 path (64 B/clk) to its f64 matrix rate (128 flop/clk) -- at m = 1 280 it ran at 36 % of the matrix peak with the
// miss path saturated (3 240 workgroups per chain, each pulling 2 x 1.25 MB of K).  Here a workgroup owns a 32 x 32 block
// of G's upper triangle: per four sites a wave loads 2 x 2 x 16 columns and issues FOUR MFMAs (three on a diagonal block,
// whose lower-left tile is the transpose of its upper-right one) -- twice the flop per byte, a quarter of the workgroups.
// Same operands, same instruction order per output element, the 16 partial tiles of a tile added in wave order: the bits of
// k_rsr_gram.  K's rows are padded to a multiple of 32 columns (zeros).  Dynamic LDS: [4 tiles][16 waves][64 lanes][4] doubles.
// NC chains per workgroup (grid.y = ceil(C / NC)): K's columns are loaded ONCE for the NC chains and weighted by each chain's
// omega where they are used -- the kernel streams 2 x 32 columns of K (5 MB at n = 10 000) per workgroup through the Infinity
// Cache, 16.8 GB per iteration at m = 1 280 with four chains: bound by that stream (7.7 TB/s), not by the matrix cores.  The
// products row[cc] * w are the ones the one-chain form takes, in the same order per output element: the same bits.
template <int NC>
__global__ void __launch_bounds__(64 * GRAM_WAVES) k_rsr_gram32(const RsrArgs a, int e, int sync_on)
{
    extern __shared__ __attribute__((aligned(16))) double s_g32[];
    const int chain0 = (int)blockIdx.y * NC;
    const bool synced = sync_on && a.sync != nullptr;
    // first kernel of the main stream's sequence: k_z_ob of the previous sequence is complete, the side stream may start
    if (synced && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) sync_set(a.sync + SYNC_MAIN, a.sync[SYNC_MAIN_SEQ + e]);
    bool on[NC];
    const double *om[NC];
    bool any = false;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int chain = min(chain0 + c, a.C - 1);
        const ChainScalars &sc = a.scs[chain];
        const Ctl ctl = sc.ctl[e];
        on[c] = chain0 + c < a.C && !(ctl.koff || ctl.it >= sc.it_stop);
        om[c] = a.omega_b[ctl.it & 1] + (size_t)chain * a.n;
        any = any || on[c];
    }
    if (!any) return;
    const int T2 = (a.m + 31) / 32;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lc = lane & 15, lk = lane >> 4;
    int ba = 0, rem = (int)blockIdx.x;  // upper triangle of 32 x 32 blocks, row by row
    while (rem >= T2 - ba) { rem -= T2 - ba; ++ba; }
    const int bc = ba + rem;
    const bool diag = ba == bc;
    const int ca = ba * 32 + lc, cc = bc * 32 + lc;
    v4d acc[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[c][q] = v4d{0.0, 0.0, 0.0, 0.0};
    constexpr int NB = NC == 1 ? 4 : 2;  // four-site groups per batch: two batches in flight
    double a0[NB], a1[NB], b0[NB], b1[NB], w[NC][NB], na0[NB], na1[NB], nb0[NB], nb1[NB], nw[NC][NB];
    auto load = [&](int i0, double (&x0)[NB], double (&x1)[NB], double (&y0)[NB], double (&y1)[NB], double (&ww)[NC][NB]) {
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int i = i0 + 4 * t + lk;
            const bool vi = i < a.n;
            const int ii = vi ? i : 0;
            const double *row = a.K + (size_t)ii * a.ldk;
            x0[t] = row[ca];
            x1[t] = row[ca + 16];
            y0[t] = row[cc];
            y1[t] = row[cc + 16];
#pragma unroll
            for (int c = 0; c < NC; ++c) ww[c][t] = vi ? om[c][ii] : 0.0;
        }
    };
    // A wave's sites are the same in both forms -- 16 consecutive ones out of every 16 x GRAM_WAVES, in ascending order (which
    // is what decides the bits of a partial tile) -- taken in batches of 4 NB: batch b starts at site_of(b).
    constexpr int SUBS = 4 / NB;
    auto site_of = [&](int b) { return wave * 16 + (b / SUBS) * 16 * GRAM_WAVES + (b % SUBS) * 4 * NB; };
    int b = 0;
    if (site_of(0) < a.n) load(site_of(0), a0, a1, b0, b1, w);
    for (; site_of(b) < a.n; ++b) {
        if (site_of(b + 1) < a.n) load(site_of(b + 1), na0, na1, nb0, nb1, nw);
#pragma unroll
        for (int t = 0; t < NB; ++t) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double y0 = b0[t] * w[c][t], y1 = b1[t] * w[c][t];
                acc[c][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[t], y0, acc[c][0], 0, 0, 0);
                acc[c][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[t], y1, acc[c][1], 0, 0, 0);
                if (!diag) acc[c][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t], y0, acc[c][2], 0, 0, 0);
                acc[c][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t], y1, acc[c][3], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            a0[t] = na0[t]; a1[t] = na1[t]; b0[t] = nb0[t]; b1[t] = nb1[t];
#pragma unroll
            for (int c = 0; c < NC; ++c) w[c][t] = nw[c][t];
        }
    }
    // the 16 partial tiles of each of the four tiles, added in wave order (wave q < 4 adds up tile q), chain after chain
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c > 0) __syncthreads();
#pragma unroll
        for (int tile = 0; tile < 4; ++tile)
#pragma unroll
            for (int v = 0; v < 4; ++v) s_g32[(((size_t)tile * GRAM_WAVES + wave) * 64 + lane) * 4 + v] = acc[c][tile][v];
        __syncthreads();
        if (on[c] && wave < 4 && !(diag && wave == 2)) {
            const int ra = ba * 32 + (wave >> 1) * 16, rc = bc * 32 + (wave & 1) * 16;  // tile `wave`: rows ra.., columns rc..
            double *G = a.gram + (size_t)(chain0 + c) * a.m * a.m;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                double t = s_g32[(((size_t)wave * GRAM_WAVES + 0) * 64 + lane) * 4 + v];
#pragma unroll
                for (int ww = 1; ww < GRAM_WAVES; ++ww) t += s_g32[(((size_t)wave * GRAM_WAVES + ww) * 64 + lane) * 4 + v];  // fixed order
                const int r = ra + 4 * v + lk, cidx = rc + lc;
                if (r < a.m && cidx < a.m) G[(size_t)r * a.m + cidx] = t;
            }
        }
    }
}
__host__ __device__ inline int rsr_gram32_blocks(int m)
{
    const int T2 = (m + 31) / 32;
    return T2 * (T2 + 1) / 2;
}
constexpr size_t RSR_GRAM32_LDS = (size_t)4 * GRAM_WAVES * 64 * 4 * sizeof(double);

// Row stride of the Cholesky factor in LDS: odd, so that a column walk (one row per lane) touches every bank once.
__host__ __device__ inline int rsr_ld(int m) { return 16 * ((m + 15) / 16) + 1; }
__host__ __device__ inline size_t rsr_solve_lds_doubles(int m) { return (size_t)m * rsr_ld(m) + 4 * (size_t)m; }

// Broadcast of one lane's double; the lane index is wave-uniform.
__device__ inline double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rsqrt_pivot(double x) { return rsqrt_nr(x); }  // occ_kernels.hpp

#ifdef OCC_SOLVE_STAMPS
#define RSR_STAMP(pt) if (chain == 0 && tid == 0) g_solve_stamps[(pt)] = wall_clock64();
#else
#define RSR_STAMP(pt)
#endif

// ---- the column step of k_rsr_solve's factorisation (see there).  Column j = 16 JA + jj of the factor is in LDS;
// the step applies its rank-one update and publishes row j + 1, which lies in block row JN (JA, or JA + 1 when
// jj = 15).  Order: block row JN first; then the pivot's reciprocal root is started and the other block rows are
// finished while that chain runs (one basic block, the scheduler interleaves them); the owners of row j + 1 scale
// and store it.  One workgroup barrier.
template <int NB, int JN>
__device__ __forceinline__ void rsr_publish_row(double (&u)[NB][NB], double (&rr)[NB], double rinv, int jn, double *Un, double *yv,
                                                double *dinv, int ty, int tx)
{
    const int jjn = jn & 15;
    if (ty == jjn) {
#pragma unroll
        for (int ib = JN; ib < NB; ++ib) {
            double v = u[JN][ib] * rinv;
            if (ib == JN) v = (tx > jjn) ? v : 0.0;  // zeros at and left of the diagonal
            u[JN][ib] = v;
            Un[tx + 16 * ib] = v;
        }
        rr[JN] = rr[JN] * rinv;
        if (tx == jjn) {
            yv[jn] = rr[JN];
            dinv[jn] = rinv;
        }
    }
}

template <int NB, int JA, int JN>
__device__ __forceinline__ void rsr_column(double (&u)[NB][NB], double (&rr)[NB], int jj, int ld, double *U, double *yv, double *dinv,
                                           int &bad, int tid)
{
    const int j = 16 * JA + jj, jn = j + 1, jjn = jn & 15, ty = tid >> 4, tx = tid & 15;
    const double *Uj = U + (size_t)j * ld;
    const double yj = yv[j];
    double ri[NB], rk[NB];
#pragma unroll
    for (int ib = JN; ib < NB; ++ib) {
        ri[ib] = Uj[tx + 16 * ib];
        rk[ib] = Uj[ty + 16 * ib];
    }
#pragma unroll
    for (int ib = JN; ib < NB; ++ib) u[JN][ib] = fma(-rk[JN], ri[ib], u[JN][ib]);
    rr[JN] = fma(-rk[JN], yj, rr[JN]);
    // EVERY wave runs the pivot's chain on the lane its own copy of the owner would be (ten instructions, meaningless
    // outside the owning wave): no branch, so the chain sits in one block with the updates below and hides in them
    const double piv = readlane_f64(u[JN][JN], ((jjn & 3) << 4) | jjn);
    double rinv = rsqrt_pivot(piv);
    asm volatile("" : "+v"(rinv));  // keeps the chain here (it would sink into the store branch)
#pragma unroll
    for (int ia = JN + 1; ia < NB; ++ia) {
#pragma unroll
        for (int ib = ia; ib < NB; ++ib) u[ia][ib] = fma(-rk[ia], ri[ib], u[ia][ib]);
        rr[ia] = fma(-rk[ia], yj, rr[ia]);
    }
    if (ty == jjn) bad |= !(piv > 0.0);
    rsr_publish_row<NB, JN>(u, rr, rinv, jn, U + (size_t)jn * ld, yv, dinv, ty, tx);
    __syncthreads();
}

// Block column JA (16 columns) and, recursively, the ones after it.  Row m - 1 is published by column m - 2.
template <int NB, int JA>
__device__ __forceinline__ void rsr_block(double (&u)[NB][NB], double (&rr)[NB], int m, int ld, double *U, double *yv, double *dinv,
                                          int &bad, int tid)
{
    if constexpr (JA < NB) {
#pragma unroll 1
        for (int jj = 0; jj < 15; ++jj) {
            if (16 * JA + jj + 1 >= m) return;
            rsr_column<NB, JA, JA>(u, rr, jj, ld, U, yv, dinv, bad, tid);
        }
        if constexpr (JA + 1 < NB) {
            if (16 * JA + 16 >= m) return;
            rsr_column<NB, JA, JA + 1>(u, rr, 15, ld, U, yv, dinv, bad, tid);
            rsr_block<NB, JA + 1>(u, rr, m, ld, U, yv, dinv, bad, tid);
        }
    }
}

// One workgroup per chain.  Dynamic LDS: U[m][ld] (the finished rows of the upper Cholesky factor, ld = rsr_ld(m)),
// then four m-vectors.  NB = the number of 16-wide blocks covering m: 16 (NB - 1) < m <= 16 NB.
//
// The matrix lives in REGISTERS: the 256 threads form a 16 x 16 grid (ty, tx) and thread (ty, tx) owns the entries
// (ty + 16 a, tx + 16 b), a <= b < NB (block-cyclic, so the shrinking trailing block stays spread over all threads;
// blocks below the diagonal are never touched).  Column j: every thread reads the values of row j (LDS) that meet
// its rows and columns and applies the rank-one update to its registers; the 16 threads of grid row (j + 1) % 16
// (a quarter of one wave; the pivot comes by v_readlane) scale row j + 1 by 1/sqrt(pivot) as soon as THEIR block
// row is updated and publish it to LDS with zeros at and left of the diagonal; ONE workgroup barrier.  A zero in
// row j at k <= j leaves the finished rows alone, so there is not one mask or branch in the update.  The right-hand side travels as an extra column (rr):
// the forward substitution U'y = rhs is finished when the factor is.  An entry receives its updates for j = 0, 1,
// ... in turn: the order of the oracle's dot products.  The diagonal of U is never stored, its reciprocal is.
template <int NB>
__global__ void __launch_bounds__(256) k_rsr_solve(const RsrArgs a, int e)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double s_scalar[2];
    __shared__ int s_bad;
    const int chain = blockIdx.y, m = a.m, tid = threadIdx.x, nt = blockDim.x, ld = rsr_ld(m);
    ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    const uint32_t it = ctl.it;
    double *U = smem, *th = smem + (size_t)m * ld, *rh = th + m, *yv = rh + m, *tmp = yv + m;
    double *theta = a.theta + (size_t)chain * m;
    const int ty = tid >> 4, tx = tid & 15;
    RSR_STAMP(0)
    // the Gram matrix and Qr of this thread's entries are on their way while tau is drawn
    const double *G = a.gram + (size_t)chain * m * m;
    double u[NB][NB], qr[NB][NB];
#pragma unroll
    for (int ia = 0; ia < NB; ++ia)
#pragma unroll
        for (int ib = ia; ib < NB; ++ib) {
            const int k = ty + 16 * ia, i = tx + 16 * ib;
            const bool in = (k < m) && (i < m) && (i >= k);
            const size_t at = in ? (size_t)k * m + i : 0;  // unconditional loads (a branch per entry would serialise them)
            const double gv = G[at], qv = a.Qr[at];
            u[ia][ib] = in ? gv : 0.0;
            qr[ia][ib] = in ? qv : 0.0;
        }
    // ---- while those loads fly: theta and the noise vector to LDS, then E eps2 (two threads per row: the halves of
    // the sum) and the chunk sums of K'u -- none of it needs tau
    for (int t = tid; t < m; t += nt) {
        th[t] = theta[t];
        yv[t] = block_normal(sc.key, (uint32_t)t, 0, it, STREAM_RSR);
    }
    if (tid == 0) s_bad = 0;
    __syncthreads();
    const int r = tid & 127, half = tid >> 7, hm = (m + 1) >> 1;
    double es = 0.0, ku = 0.0;
    if (r < m) {  // loads in batches (a load per loop trip would be one L2 round trip per term)
        const int j1 = half ? m : hm;
        for (int j0 = half ? hm : 0; j0 < j1; j0 += 16) {
            double ev[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) ev[s] = a.Et[(size_t)min(j0 + s, j1 - 1) * m + r];  // E[r][j], from the transposed copy
#pragma unroll
            for (int s = 0; s < 16; ++s) es = fma(ev[s], (j0 + s < j1) ? yv[j0 + s] : 0.0, es);
        }
        if (half) tmp[r] = es;
        else {
            const double *pr = a.rhs + (size_t)chain * a.nchunk * m + r;
            for (int c0 = 0; c0 < a.nchunk; c0 += 8) {  // chunk order
                double pv[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) pv[s] = pr[(size_t)min(c0 + s, a.nchunk - 1) * m];
#pragma unroll
                for (int s = 0; s < 8; ++s) ku += (c0 + s < a.nchunk) ? pv[s] : 0.0;
            }
        }
    }
    // ---- tau: rate = 1/2 theta' Qr theta + tau_rate (theta of the previous iteration).  theta' Qr theta = the
    // diagonal terms + twice the upper ones: every thread its entries, then a fixed-order reduction (the lanes of a
    // wave by xor shuffles, the four waves by thread 0)
    {
        double tk[NB], ti[NB], part = 0.0;
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) {
            tk[ib] = th[min(ty + 16 * ib, m - 1)];
            ti[ib] = th[min(tx + 16 * ib, m - 1)];
        }
#pragma unroll
        for (int ia = 0; ia < NB; ++ia)
#pragma unroll
            for (int ib = ia; ib < NB; ++ib) {
                const double w = (ty + 16 * ia == tx + 16 * ib) ? 1.0 : 2.0;  // entries outside the triangle hold qr = 0
                part = fma(w * qr[ia][ib], tk[ia] * ti[ib], part);
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if ((tid & 63) == 0) rh[tid >> 6] = part;
    }
    __syncthreads();
    if (tid == 0) {
        const double quad = ((rh[0] + rh[1]) + rh[2]) + rh[3];
        const double rate = 0.5 * quad + a.tau_rate;
        // the standard gamma variate of this iteration was drawn ahead by k_noise (like the ICAR path: occ_kernels.hpp)
        const double tau = (1.0 / rate) * load_agent(&sc.tau_gamma[it & 1]);
        sc.tau = tau;
        s_scalar[0] = tau;
        s_scalar[1] = sqrt(tau);
    }
    __syncthreads();
    const double tau = s_scalar[0], st = s_scalar[1];
    RSR_STAMP(1)
    // ---- prec = K'OK + tau Qr (upper triangle, registers), rhs = K'u + sqrt(tau) E eps2
#pragma unroll
    for (int ia = 0; ia < NB; ++ia)
#pragma unroll
        for (int ib = ia; ib < NB; ++ib) u[ia][ib] = fma(tau, qr[ia][ib], u[ia][ib]);
    if (r < m && !half) rh[r] = fma(st, es + tmp[r], ku);
    __syncthreads();
    double rr[NB];  // the right-hand side of this thread's rows (every thread of a grid row carries a copy)
#pragma unroll
    for (int ia = 0; ia < NB; ++ia) rr[ia] = (ty + 16 * ia < m) ? rh[ty + 16 * ia] : 0.0;
    double *dinv = tmp;  // the reciprocals of U's diagonal
    __syncthreads();
    RSR_STAMP(2)
    // ---- Cholesky + forward substitution: row 0, then one step per column (rsr_column)
    int bad = 0;
    if (tid < 64) {
        const double piv = readlane_f64(u[0][0], 0);
        bad = !(piv > 0.0);
        rsr_publish_row<NB, 0>(u, rr, rsqrt_pivot(piv), 0, U, yv, dinv, ty, tx);
    }
    __syncthreads();
    rsr_block<NB, 0>(u, rr, m, ld, U, yv, dinv, bad, tid);
    if (bad) s_bad = 1;
    __syncthreads();
    RSR_STAMP(3)
    if (s_bad) {
        if (tid == 0) sc.err = -4;  // OCC_E_CHOLESKY
        return;
    }
    // ---- U theta = y (backward) by ONE wave, column-oriented (lanes own entries lane and lane + 64; m <= 128):
    // after theta_i is known every earlier entry subtracts its term -- no workgroup barrier.  Lane t walks ITS rows
    // of U; the loads of four steps are issued before the four dependent steps.
    if (tid < 64) {
        double r0 = (tid < m) ? yv[tid] : 0.0, r1 = (tid + 64 < m) ? yv[tid + 64] : 0.0;
        const double *c0 = U + (size_t)min(tid, m - 1) * ld, *c1 = U + (size_t)min(tid + 64, m - 1) * ld;
        // theta_i of the upper half (lanes' second entries; every first entry takes a term), then of the lower half;
        // the odd steps first, so that the groups of four carry no tail tests
        auto upper = [&](int i, double a0, double a1, double dv) {
            const double ti = readlane_f64(r1, i - 64) * dv;
            r0 = fma(-a0, ti, r0);
            r1 = (tid + 64 < i) ? fma(-a1, ti, r1) : (tid + 64 == i ? ti : r1);
        };
        auto lower = [&](int i, double a0, double dv) {
            const double ti = readlane_f64(r0, i) * dv;
            r0 = (tid < i) ? fma(-a0, ti, r0) : (tid == i ? ti : r0);
        };
        int i = m - 1;
        for (; i >= 64 && ((i - 63) & 3); --i) upper(i, c0[i], c1[i], dinv[i]);
        for (; i >= 64; i -= 4) {
            double a0[4], a1[4], dv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a0[s] = c0[i - s];
                a1[s] = c1[i - s];
                dv[s] = dinv[i - s];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) upper(i - s, a0[s], a1[s], dv[s]);
        }
        for (; i >= 0 && ((i + 1) & 3); --i) lower(i, c0[i], dinv[i]);
        for (; i >= 0; i -= 4) {
            double a0[4], dv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a0[s] = c0[i - s];
                dv[s] = dinv[i - s];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) lower(i - s, a0[s], dv[s]);
        }
        if (tid < m) theta[tid] = r0;
        if (tid + 64 < m) theta[tid + 64] = r1;
    }
    RSR_STAMP(4)
}

// eta = K theta from the transposed copy of K (coalesced along the sites; the column loop's loads go out in batches
// of 16), then, in the same thread, the site's terms of beta's system and their block partial sums (k_beta_partial
// without the ICAR solve's projection step).
template <int P>
__global__ void __launch_bounds__(256) k_rsr_eta_beta(const RsrArgs a, OCC_KARGS)
{
    __shared__ double s_th[RSR_BIG_MAX];
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
    for (int t = threadIdx.x; t < a.m; t += blockDim.x) s_th[t] = a.theta[(size_t)chain * a.m + t];
    __syncthreads();
    const int n = c.n, i = blk * blockDim.x + threadIdx.x, il = min(i, n - 1);
    double eta = 0.0;
    for (int c0 = 0; c0 < a.m; c0 += 16) {
        double kv[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) kv[s] = a.Kt[(size_t)min(c0 + s, a.m - 1) * n + il];
#pragma unroll
        for (int s = 0; s < 16; ++s) eta = fma(kv[s], (c0 + s < a.m) ? s_th[c0 + s] : 0.0, eta);
    }
    double acc[nacc(P)];
#pragma unroll
    for (int t = 0; t < nacc(P); ++t) acc[t] = 0.0;
    if (i < n) {
        const size_t ci = (size_t)chain * n + i;
        c.eta[ci] = eta;
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


// ==== m > RSR_MAX_DIM: the same conditional with the m x m system in GLOBAL memory ===================================
// The reference keeps every Moran eigenvector above its threshold (logit.py:415-446: about 13 % of the sites of a
// lattice at the default r = 0.5 -- 1 280 columns at 100x100); the LDS-resident solve above stops at 128.  Here:
//   k_rsrb_tau       eps2, theta' Qr theta, tau                                       one workgroup per chain
//   k_rsrb_assemble  prec = G + tau Qr (upper triangle, in place in `gram`), rhs = K'u + sqrt(tau) E eps2   one per row
//   k_rsrb_panel     panel by panel (RSR_PANEL rows): the diagonal block's upper Cholesky factor in LDS (every workgroup
//                    for itself), then U_kk' X = P_k,rest for the block row right of it, one column per thread
//   k_rsrb_update    P_ij -= sum_t U_ti U_tj over the panel's rows t, 16 x 16 tiles of the trailing upper triangle
//   k_rsrb_solve     U'y = rhs, U theta = y, blocked by panels                          one workgroup per chain
// k_rsr_gram and k_rsr_eta_beta are the general kernels above.  Every sum has a fixed order (no atomics); the order
// is not the small path's (the two agree to rounding, like the oracle).  Plain kernels: at m = 1 280 the conditional is
// ~10^9 flops of Cholesky per chain and iteration beside a 2 10^10-flop Gram matrix -- milliseconds where the
// reference's host code takes seconds -- and not the path BASELINE's metric is quoted on.
// theta' Qr theta in slices of 64 rows (a workgroup of 16 waves, four rows per wave, columns over the lanes), one partial
// per workgroup; the noise of the prior term for those rows.  (Round 2: one workgroup per chain read all of Qr, 13 MB at
// m = 1 280: 0.56 ms.)
constexpr int RSRB_QROWS = 64;
__global__ void __launch_bounds__(1024) k_rsrb_tau(const RsrArgs a, int e)
{
    __shared__ double s_th[RSR_BIG_MAX];
    __shared__ double s_part[16];
    const int chain = blockIdx.y, m = a.m, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r0 = (int)blockIdx.x * RSRB_QROWS;
    const ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    const uint32_t it = ctl.it;
    for (int t = tid; t < m; t += 1024) s_th[t] = a.theta[(size_t)chain * m + t];
    if (tid < RSRB_QROWS && r0 + tid < m) a.big_eps[(size_t)chain * m + r0 + tid] = block_normal(sc.key, (uint32_t)(r0 + tid), 0, it, STREAM_RSR);
    __syncthreads();
    double part = 0.0;  // this wave's rows r0 + wave, + 16, + 32, + 48: theta_r (Qr theta)_r
    for (int r = r0 + wave; r < min(m, r0 + RSRB_QROWS); r += 16) {
        double acc = 0.0;
        for (int c = lane; c < m; c += 64) acc = fma(a.Qr[(size_t)r * m + c], s_th[c], acc);
        part = fma(s_th[r], wave_sum(acc), part);
    }
    if (lane == 0) s_part[wave] = part;
    __syncthreads();
    if (tid == 0) {
        double quad = 0.0;
        for (int w = 0; w < 16; ++w) quad += s_part[w];
        a.big_quad[(size_t)chain * gridDim.x + blockIdx.x] = quad;
    }
}

// prec = G + tau Qr (upper triangle, in place), rhs = K'u + sqrt(tau) E eps2: one workgroup per row; tau from the slices'
// partial sums of theta' Qr theta, added in slice order by every workgroup (the same bits everywhere)
__global__ void __launch_bounds__(256) k_rsrb_assemble(const RsrArgs a, int e)
{
    __shared__ double s_part[4];
    const int chain = blockIdx.y, r = blockIdx.x, m = a.m, tid = threadIdx.x;
    ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    const int nq = (m + RSRB_QROWS - 1) / RSRB_QROWS;
    double quad = 0.0;
    for (int b = 0; b < nq; ++b) quad += a.big_quad[(size_t)chain * nq + b];
    const double rate = 0.5 * quad + a.tau_rate;
    const double tau = (1.0 / rate) * load_agent(&sc.tau_gamma[ctl.it & 1]);  // the variate was drawn ahead by k_noise
    const double st = sqrt(tau);
    if (r == 0 && tid == 0) sc.tau = tau;
    double *P = a.gram + (size_t)chain * m * m;
    for (int c = r + tid; c < m; c += 256) P[(size_t)r * m + c] = fma(tau, a.Qr[(size_t)r * m + c], P[(size_t)r * m + c]);
    double es = 0.0;
    const double *eps = a.big_eps + (size_t)chain * m;
    for (int j = tid; j < m; j += 256) es = fma(a.E[(size_t)r * m + j], eps[j], es);
    es = wave_sum(es);
    if ((tid & 63) == 0) s_part[tid >> 6] = es;
    __syncthreads();
    if (tid == 0) {
        const double ee = ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3];
        a.big_rhs[(size_t)chain * m + r] = fma(st, ee, a.rhs[(size_t)chain * a.nchunk * m + r]);
    }
}

// Upper Cholesky factor of one RSR_PANEL x RSR_PANEL block by ONE WAVE, the block in registers: lane c < kb owns column c
// (u[r] = entry (r, c)).  Row i: the pivot comes by v_readlane from lane i, the row is divided by its root, and every later
// row r takes its update with U_ir read from lane r -- no LDS, no barrier (round 2: the block in LDS, three workgroup
// barriers per row, every workgroup of the launch for itself: 30 of the panel kernel's 38 us).  The operations are those of
// the row-by-row loop it replaces.  Round 4: the pivot's RECIPROCAL root (v_rsq_f64 + two Newton steps, as in the small path)
// instead of a square root and a division per row -- a dependent chain of ~6 instead of ~25 instructions on the one wave
// every workgroup of the launch waits for, 32 times per panel -- and the diagonal of the stored factor holds 1 / U_ii: nothing
// downstream needs U_ii itself, everything divides by it.  Returns false on a pivot <= 0.
__device__ __forceinline__ bool rsrb_diag_factor(double (&u)[RSR_PANEL], int kb, int lane)
{
    bool ok = true;
#pragma unroll
    for (int i = 0; i < RSR_PANEL; ++i) {
        const double piv = readlane_f64(u[i], i);
        const bool good = piv > 0.0;
        if (i < kb && !good) ok = false;
        const double rinv = rsqrt_pivot(good ? piv : 1.0);
        u[i] = (lane == i) ? rinv : ((lane > i) ? u[i] * rinv : u[i]);
#pragma unroll
        for (int r = i + 1; r < RSR_PANEL; ++r) {
            const double uir = readlane_f64(u[i], r);  // U_ir (lane r's entry of row i)
            if (lane >= r) u[r] = fma(-uir, u[i], u[r]);
        }
    }
    return ok;
}

// Panel step k0: the diagonal block's factor (wave 0 of every workgroup, for itself: 5 us, cheaper than a launch of its
// own), then U_kk' X = P_k,rest for this workgroup's 256 columns of the block row, one column per thread.  The last
// workgroup of the grid takes the right-hand side along -- U_kk' y_k = rhs_k: the forward substitution is finished with
// the factor, as in the small path; workgroup 0 stores the factored block (big_dfac: see below).
__global__ void __launch_bounds__(256) k_rsrb_panel(const RsrArgs a, int e, int k0)
{
    __shared__ double D[RSR_PANEL][RSR_PANEL + 1];
    __shared__ int s_bad;
    const int chain = blockIdx.y, m = a.m, tid = threadIdx.x;
    ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop || sc.err != 0) return;
    const int kb = min(RSR_PANEL, m - k0);
    double *P = a.gram + (size_t)chain * m * m;
    if (tid < 64) {
        double u[RSR_PANEL];
#pragma unroll
        for (int r = 0; r < RSR_PANEL; ++r) {
            const bool in = tid < kb && r < kb && tid >= r;
            const double v = P[(size_t)(k0 + (in ? r : 0)) * m + k0 + (in ? tid : 0)];
            u[r] = in ? v : ((r == tid) ? 1.0 : 0.0);  // (identity outside the block: nothing divides by zero)
        }
        const bool ok = rsrb_diag_factor(u, kb, tid);
        if (tid == 0) s_bad = ok ? 0 : 1;
        if (tid < RSR_PANEL) {
#pragma unroll
            for (int r = 0; r < RSR_PANEL; ++r) D[r][tid] = (r <= tid && tid < kb && r < kb) ? u[r] : 0.0;
        }
    }
    __syncthreads();
    if (s_bad) {
        if (blockIdx.x == 0 && tid == 0) sc.err = -4;  // OCC_E_CHOLESKY
        return;
    }
    // The factored diagonal block goes to a buffer of its own, NOT back into P: every workgroup of this launch loads the
    // unfactored block from P above, and nothing orders those loads before a write-back by workgroup 0.  P's
    // diagonal block is never read again; k_rsrb_solve takes the factor from big_dfac.
    if (blockIdx.x == 0) {
        double *F = a.big_dfac + ((size_t)chain * ((m + RSR_PANEL - 1) / RSR_PANEL) + k0 / RSR_PANEL) * (RSR_PANEL * RSR_PANEL);
        for (int t = tid; t < RSR_PANEL * RSR_PANEL; t += 256) F[t] = D[t / RSR_PANEL][t % RSR_PANEL];
    }
    // the block row right of the diagonal block: U_kk' x = p, one column per thread; the last thread of workgroup 0: the
    // right-hand side's entries of this panel
    // (the LAST workgroup of the grid is the right-hand side's: its thread 0)
    const bool rhs_wg = blockIdx.x == gridDim.x - 1;
    const int j = rhs_wg ? m : k0 + kb + (int)blockIdx.x * 256 + tid;
    const bool is_rhs = rhs_wg && tid == 0;
    double *rhs = a.big_rhs + (size_t)chain * m;
    if (j < m || is_rhs) {
        double x[RSR_PANEL];
#pragma unroll
        for (int t = 0; t < RSR_PANEL; ++t) x[t] = (t < kb) ? (is_rhs ? rhs[k0 + t] : P[(size_t)(k0 + t) * m + j]) : 0.0;
#pragma unroll
        for (int t = 0; t < RSR_PANEL; ++t) {
            if (t < kb) {
                double v = x[t];
#pragma unroll
                for (int q = 0; q < RSR_PANEL; ++q)
                    if (q < t) v = fma(-D[q][t], x[q], v);
                x[t] = v * D[t][t];  // (the diagonal holds 1 / U_tt)
            }
        }
#pragma unroll
        for (int t = 0; t < RSR_PANEL; ++t)
            if (t < kb) {
                if (is_rhs) rhs[k0 + t] = x[t];
                else P[(size_t)(k0 + t) * m + j] = x[t];
            }
    }
}

// Trailing update of panel step k0 on the matrix cores: P_ij -= sum_t U_ti U_tj over the panel's rows t, one WAVE per
// 16 x 16 tile of the trailing upper triangle (four tiles per workgroup): v_mfma_f64_16x16x4_f64 with A = -U[:, i-block]'
// and B = U[:, j-block], four panel rows per instruction, the tile as the accumulator (lane l: rows 4 v + l / 16, column
// l % 16; operands: 4 x 128-byte row segments per load).  (Round 2: one thread per entry, 64 scalar loads and 32 FMAs:
// 10-134 us per launch, 2.1 ms per iteration at m = 1 280.)  The last workgroup row of the grid updates the right-hand
// side: rhs_j -= sum_t U_tj y_t.
__global__ void __launch_bounds__(256) k_rsrb_update(const RsrArgs a, int e, int k0)
{
    const int chain = blockIdx.z, m = a.m, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lc = lane & 15, lk = lane >> 4;
    const ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop || sc.err != 0) return;
    const int kb = min(RSR_PANEL, m - k0), base = k0 + kb;
    double *P = a.gram + (size_t)chain * m * m;
    if (blockIdx.y == gridDim.y - 1) {  // the right-hand side: one thread per trailing entry
        double *rhs = a.big_rhs + (size_t)chain * m;
        const int j = base + (int)blockIdx.x * 256 + (int)threadIdx.x;
        if (j < m) {
            double v = rhs[j];
            for (int t = 0; t < kb; ++t) v = fma(-P[(size_t)(k0 + t) * m + j], rhs[k0 + t], v);
            rhs[j] = v;
        }
        return;
    }
    // tile (ti, tj) of the trailing block, ti <= tj: workgroup (bx, by) holds tiles (2 by + wave / 2, 2 bx + wave % 2)
    const int ti = 2 * (int)blockIdx.y + (wave >> 1), tj = 2 * (int)blockIdx.x + (wave & 1);
    const int i0 = base + 16 * ti, j0 = base + 16 * tj;
    if (tj < ti || i0 >= m || j0 >= m) return;
    v4d acc;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int r = i0 + 4 * v + lk, c = j0 + lc;
        acc[v] = (r < m && c < m) ? P[(size_t)r * m + c] : 0.0;
    }
    double av[RSR_PANEL / 4], bv[RSR_PANEL / 4];
#pragma unroll
    for (int s4 = 0; s4 < RSR_PANEL / 4; ++s4) {
        const int t = 4 * s4 + lk;
        const bool in = t < kb;
        const double *row = P + (size_t)(k0 + (in ? t : 0)) * m;
        const double ua = (in && i0 + lc < m) ? row[min(i0 + lc, m - 1)] : 0.0, ub = (in && j0 + lc < m) ? row[min(j0 + lc, m - 1)] : 0.0;
        av[s4] = -ua;
        bv[s4] = ub;
    }
#pragma unroll
    for (int s4 = 0; s4 < RSR_PANEL / 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int r = i0 + 4 * v + lk, c = j0 + lc;
        if (r < m && c < m && c >= r) P[(size_t)r * m + c] = acc[v];
    }
}

// ---- round 4: the trailing update of panel step k0 AND the panel work of step k0 + kb in ONE launch ---------------------
// k_rsrb_panel + k_rsrb_update were 2 x 40 launches per iteration at m = 1 280 (23 + 12 us and two launch gaps per step:
// 1.7 ms of a 3.5 ms iteration), although a step's panel work needs nothing but the previous step's update of ITS OWN block
// row.  Here the workgroups of the first tile row pair (blockIdx.y == 0) -- whose tiles are the next panel's 32 rows -- go on
// after their update: each forms the NEXT diagonal block's update and factor for itself (three more 16 x 16 tiles and one
// wave's work: cheaper than a hand-over between workgroups, and nothing is assumed about the order workgroups are dispatched
// in -- the panel kernel does the same), then solves U_kk' X = P for its own 32 columns from the tiles it holds; the
// right-hand side's workgroup does the same for the panel's 32 entries.  The rest of the grid is k_rsrb_update unchanged.
// The same operations on the same operands as the two kernels: the same bits.
__device__ __forceinline__ v4d rsrb_tile_update(const double *P, int m, int k0, int kb, int i0, int j0, int lc, int lk)
{
    v4d acc;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int r = i0 + 4 * v + lk, c = j0 + lc;
        acc[v] = (r < m && c < m) ? P[(size_t)r * m + c] : 0.0;
    }
    double av[RSR_PANEL / 4], bv[RSR_PANEL / 4];
#pragma unroll
    for (int s4 = 0; s4 < RSR_PANEL / 4; ++s4) {
        const int t = 4 * s4 + lk;
        const bool in = t < kb;
        const double *row = P + (size_t)(k0 + (in ? t : 0)) * m;
        const double ua = (in && i0 + lc < m) ? row[min(i0 + lc, m - 1)] : 0.0, ub = (in && j0 + lc < m) ? row[min(j0 + lc, m - 1)] : 0.0;
        av[s4] = -ua;
        bv[s4] = ub;
    }
#pragma unroll
    for (int s4 = 0; s4 < RSR_PANEL / 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc, 0, 0, 0);
    return acc;
}
// The next panel's diagonal block (rows / columns base .. base + kb2), updated by panel step k0 and factored, into D (upper
// factor, 1 / U_tt on the diagonal); every thread of the workgroup calls it.  `mine`: this workgroup's own tiles ARE the
// block (acc_own of waves 0, 1, 3); else waves 0, 1, 2 form tiles (0,0), (0,1), (1,1) from P.  Returns false on a pivot <= 0.
__device__ __forceinline__ bool rsrb_next_diag(const double *P, int m, int k0, int kb, int base, int kb2, bool mine, v4d acc_own,
                                               double (&Dblk)[RSR_PANEL][RSR_PANEL + 1], double (&D)[RSR_PANEL][RSR_PANEL + 1], int *s_bad)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lc = lane & 15, lk = lane >> 4;
    int ti = -1, tj = -1;
    if (mine) { if (wave != 2) { ti = wave >> 1; tj = wave & 1; } }  // waves 0, 1, 3 hold (0,0), (0,1), (1,1)
    else if (wave < 3) { ti = wave == 2 ? 1 : 0; tj = wave == 0 ? 0 : 1; }
    if (ti >= 0) {
        const v4d acc = mine ? acc_own : rsrb_tile_update(P, m, k0, kb, base + 16 * ti, base + 16 * tj, lc, lk);
#pragma unroll
        for (int v = 0; v < 4; ++v) Dblk[16 * ti + 4 * v + lk][16 * tj + lc] = acc[v];
    }
    __syncthreads();
    if (tid < 64) {
        double u[RSR_PANEL];
#pragma unroll
        for (int r = 0; r < RSR_PANEL; ++r) {
            const bool in = tid < kb2 && r < kb2 && tid >= r;
            u[r] = in ? Dblk[r][tid & (RSR_PANEL - 1)] : ((r == tid) ? 1.0 : 0.0);
        }
        const bool ok = rsrb_diag_factor(u, kb2, tid);
        if (tid == 0) *s_bad = ok ? 0 : 1;
        if (tid < RSR_PANEL) {
#pragma unroll
            for (int r = 0; r < RSR_PANEL; ++r) D[r][tid] = (r <= tid && tid < kb2 && r < kb2) ? u[r] : 0.0;
        }
    }
    __syncthreads();
    return *s_bad == 0;
}
__global__ void __launch_bounds__(256) k_rsrb_step(const RsrArgs a, int e, int k0)
{
    __shared__ double Dblk[RSR_PANEL][RSR_PANEL + 1], D[RSR_PANEL][RSR_PANEL + 1];
    __shared__ int s_bad;
    const int chain = blockIdx.z, m = a.m, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lc = lane & 15, lk = lane >> 4;
    ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop || sc.err != 0) return;
    const int kb = min(RSR_PANEL, m - k0), base = k0 + kb, kb2 = min(RSR_PANEL, m - base);  // (launched only while base < m)
    double *P = a.gram + (size_t)chain * m * m;
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    if (blockIdx.y == gridDim.y - 1) {  // ---- the right-hand side: k_rsrb_update's row, then the next panel's entries
        double *rhs = a.big_rhs + (size_t)chain * m;
        const int j = base + (int)blockIdx.x * 256 + tid;
        double v = 0.0;
        if (j < m) {
            v = rhs[j];
            for (int t = 0; t < kb; ++t) v = fma(-P[(size_t)(k0 + t) * m + j], rhs[k0 + t], v);
            if (blockIdx.x != 0 || tid >= kb2) rhs[j] = v;
        }
        if (blockIdx.x != 0) return;
        const bool ok = rsrb_next_diag(P, m, k0, kb, base, kb2, false, zero4, Dblk, D, &s_bad);
        if (!ok) return;  // (the tile row's first workgroup reports it)
        if (tid < kb2) Dblk[0][tid] = v;  // (Dblk is free again: the panel's updated entries, one per thread)
        __syncthreads();
        if (tid == 0) {  // U_kk' y_k = rhs_k (k_rsrb_panel's last thread)
            double x[RSR_PANEL];
#pragma unroll
            for (int t = 0; t < RSR_PANEL; ++t) x[t] = (t < kb2) ? Dblk[0][t] : 0.0;
#pragma unroll
            for (int t = 0; t < RSR_PANEL; ++t) {
                if (t < kb2) {
                    double w = x[t];
#pragma unroll
                    for (int q = 0; q < RSR_PANEL; ++q)
                        if (q < t) w = fma(-D[q][t], x[q], w);
                    x[t] = w * D[t][t];
                }
            }
#pragma unroll
            for (int t = 0; t < RSR_PANEL; ++t)
                if (t < kb2) rhs[base + t] = x[t];
        }
        return;
    }
    // ---- tiles of the trailing block (k_rsrb_update): workgroup (bx, by) holds tiles (2 by + wave / 2, 2 bx + wave % 2)
    const int ti = 2 * (int)blockIdx.y + (wave >> 1), tj = 2 * (int)blockIdx.x + (wave & 1);
    const int i0 = base + 16 * ti, j0 = base + 16 * tj;
    const bool have = !(tj < ti || i0 >= m || j0 >= m);
    v4d acc = zero4;
    if (have) acc = rsrb_tile_update(P, m, k0, kb, i0, j0, lc, lk);
    if (blockIdx.y != 0) {
        if (have) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = i0 + 4 * v + lk, c = j0 + lc;
                if (r < m && c < m && c >= r) P[(size_t)r * m + c] = acc[v];
            }
        }
        return;
    }
    // ---- the first tile row pair: the next panel (rows base .. base + kb2)
    const bool first = blockIdx.x == 0;
    const bool ok = rsrb_next_diag(P, m, k0, kb, base, kb2, first, acc, Dblk, D, &s_bad);
    if (!ok) {
        if (first && tid == 0) sc.err = -4;  // OCC_E_CHOLESKY
        return;
    }
    if (first) {  // the factored diagonal block, for k_rsrb_solve (P's diagonal block is never read again)
        double *F = a.big_dfac + ((size_t)chain * ((m + RSR_PANEL - 1) / RSR_PANEL) + base / RSR_PANEL) * (RSR_PANEL * RSR_PANEL);
        for (int t = tid; t < RSR_PANEL * RSR_PANEL; t += 256) F[t] = D[t / RSR_PANEL][t % RSR_PANEL];
        return;
    }
    // this workgroup's 32 columns of the block row: the updated tiles through LDS (Dblk is free), one column per thread
    __syncthreads();
    if (have) {
#pragma unroll
        for (int v = 0; v < 4; ++v) Dblk[16 * (wave >> 1) + 4 * v + lk][16 * (wave & 1) + lc] = acc[v];
    }
    __syncthreads();
    const int j = base + 32 * (int)blockIdx.x + tid;
    if (tid < RSR_PANEL && j < m) {
        double x[RSR_PANEL];
#pragma unroll
        for (int t = 0; t < RSR_PANEL; ++t) x[t] = (t < kb2) ? Dblk[t][tid] : 0.0;
#pragma unroll
        for (int t = 0; t < RSR_PANEL; ++t) {
            if (t < kb2) {
                double w = x[t];
#pragma unroll
                for (int q = 0; q < RSR_PANEL; ++q)
                    if (q < t) w = fma(-D[q][t], x[q], w);
                x[t] = w * D[t][t];  // (the diagonal holds 1 / U_tt)
            }
        }
#pragma unroll
        for (int t = 0; t < RSR_PANEL; ++t)
            if (t < kb2) P[(size_t)(base + t) * m + j] = x[t];
    }
}

// U theta = y (backward), blocked by panels, one workgroup per chain; y = the right-hand side as the panel steps left it
// (the forward substitution travels with the factorisation).
__global__ void __launch_bounds__(1024) k_rsrb_solve(const RsrArgs a, int e)
{
    __shared__ double y[RSR_BIG_MAX];
    const int chain = blockIdx.y, m = a.m, tid = threadIdx.x;
    const ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop || sc.err != 0) return;
    const double *U = a.gram + (size_t)chain * m * m;
    const double *F = a.big_dfac + (size_t)chain * ((m + RSR_PANEL - 1) / RSR_PANEL) * (RSR_PANEL * RSR_PANEL);  // the diagonal blocks' factors
    for (int t = tid; t < m; t += 1024) y[t] = a.big_rhs[(size_t)chain * m + t];
    __syncthreads();
    const int last = ((m - 1) / RSR_PANEL) * RSR_PANEL;
    for (int k0 = last; k0 >= 0; k0 -= RSR_PANEL) {
        const int kb = min(RSR_PANEL, m - k0);
        if (tid < 64) {
            double v = (tid < kb) ? y[k0 + tid] : 0.0;
            const double *Fk = F + (size_t)(k0 / RSR_PANEL) * (RSR_PANEL * RSR_PANEL);
            // (round 4: this lane's row of the block loaded whole, ahead of the 32 dependent steps -- one load per step sat on
            // the chain before)
            double fr[RSR_PANEL];
#pragma unroll
            for (int s2 = 0; s2 < RSR_PANEL; ++s2) fr[s2] = (tid < kb && s2 < kb && tid <= s2) ? Fk[tid * RSR_PANEL + s2] : 0.0;
            double dg = 1.0;
#pragma unroll
            for (int s2 = 0; s2 < RSR_PANEL; ++s2) dg = (tid == s2) ? fr[s2] : dg;
#pragma unroll
            for (int s2 = RSR_PANEL - 1; s2 >= 0; --s2) {
                if (s2 < kb) {
                    const double ts = readlane_f64(v, s2) * readlane_f64(dg, s2);  // (the stored diagonal is 1 / U_ss)
                    if (tid == s2) v = ts;
                    else if (tid < s2) v = fma(-fr[s2], ts, v);  // column s2 of the block
                }
            }
            if (tid < kb) y[k0 + tid] = v;
        }
        __syncthreads();
        // rows above the panel: y_i -= sum_s U_i,k0+s y_k0+s.  A wave takes two rows per load -- lanes 0-31 the 32 entries of
        // one row (256 contiguous bytes), lanes 32-63 those of the next -- and adds the products up over the 32 lanes.  (Until
        // round 4 a thread walked along its own row: every load instruction of a wave touched 64 different cache lines, and
        // the kernel, 15 us per panel, was bound by exactly that.)
        {
            const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, c = lane & 31;
            const double yc = (c < kb) ? y[k0 + c] : 0.0;
            constexpr int UB = 8;  // row pairs in flight per wave (the loads of a batch go out together)
            for (int i0 = 2 * wave; i0 < k0; i0 += 32 * UB) {
                double t[UB];
#pragma unroll
                for (int b = 0; b < UB; ++b) {
                    const int i = i0 + 32 * b + half;
                    t[b] = (i < k0 && c < kb) ? U[(size_t)i * m + k0 + c] : 0.0;
                }
#pragma unroll
                for (int b = 0; b < UB; ++b) {
                    const int i = i0 + 32 * b + half;
                    double v = t[b] * yc;
                    v += dpp_shifted<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
                    v += dpp_shifted<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
                    v += dpp_shifted<0x141, 0xf>(v);  // row_half_mirror
                    v += dpp_shifted<0x140, 0xf>(v);  // row_mirror: every lane of a row of 16 holds the row's sum
                    v += __shfl_xor(v, 16);           // the two rows of 16 of this half
                    if (c == 0 && i < k0) y[i] -= v;
                }
            }
        }
        __syncthreads();
    }
    for (int t = tid; t < m; t += 1024) a.theta[(size_t)chain * m + t] = y[t];
}

}  // namespace occ
