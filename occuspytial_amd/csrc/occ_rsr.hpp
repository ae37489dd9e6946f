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
constexpr int RSR_BIG_MAX = 4096; // (what the reference's own dense n x n set-up reaches: 13 % of 31 000 sites) beyond RSR_MAX_DIM: the factor lives in global memory, factorised panel by panel (k_rsrb_*)
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
    double *big_u;      // [C][n] u = z - 1/2 - omega X beta + sqrt(omega) eps (k_rsrb_u -> k_rsrb_ktu)
    double *big_rhs;    // [C][m]
    double *big_dfac;   // [C][ceil(m / RSR_PANEL)][RSR_PANEL][RSR_PANEL] the factored diagonal blocks (k_rsrb_step -> k_rsrb_solve)
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
// k_rsr_gram above forms one 16 x 16 tile per workgroup (2 x 16 columns of K per MFMA); at m = 1 280 that is 3 240
// workgroups per chain, each pulling 2 x 1.25 MB of K, and it ran at 36 % of the matrix peak.  Here a workgroup owns a
// 32 x 32 block of G's upper triangle: per four sites a wave loads 2 x 2 x 16 columns and issues FOUR MFMAs (three on a
// diagonal block, whose lower-left tile is the transpose of its upper-right one) -- twice the flop per byte, a quarter of the
// workgroups.  Same operands, same instruction order per output element, a tile's WV partial tiles added in wave order.  K's
// rows are padded to a multiple of 32 columns (zeros).  Dynamic LDS: [4 tiles][WV waves][64 lanes][4] doubles.
// NC chains per workgroup (grid.y = ceil(C / NC)): K's columns are loaded ONCE for the NC chains and weighted by each chain's
// omega where they are used.  The products row[cc] * w are the ones the one-chain form takes, in the same order per output
// element: the same bits whether a chain runs alone or beside others (both forms with WV = GRAM32_WAVES).
// What bounds it is neither K's stream nor the matrix pipes' peak but the vector instructions beside the MFMAs (see the loop).
template <int NC, int WV>
__global__ void __launch_bounds__(64 * WV, 4) k_rsr_gram32(const RsrArgs a, int e, int sync_on)
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
    // A wave's sites are the same in both forms -- 16 consecutive ones out of every 16 x WV, in ascending order (which
    // is what decides the bits of a partial tile) -- taken in batches of 4 NB: batch b starts at site_of(b).
    constexpr int SUBS = 4 / NB;
    auto site_of = [&](int b) { return wave * 16 + (b / SUBS) * 16 * WV + (b % SUBS) * 4 * NB; };
    auto mm = [&](const double (&x0)[NB], const double (&x1)[NB], const double (&y0)[NB], const double (&y1)[NB], const double (&ww)[NC][NB]) {
#pragma unroll
        for (int t = 0; t < NB; ++t) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (NC > 1 && !on[c]) continue;  // (an odd chain count's last workgroup row: one chain's products only)
                const double z0 = y0[t] * ww[c][t], z1 = y1[t] * ww[c][t];
                acc[c][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[t], z0, acc[c][0], 0, 0, 0);
                acc[c][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[t], z1, acc[c][1], 0, 0, 0);
                if (!diag) acc[c][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[t], z0, acc[c][2], 0, 0, 0);  // (a diagonal block's lower-left tile is the transpose of its upper-right one)
                acc[c][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[t], z1, acc[c][3], 0, 0, 0);
            }
        }
    };
    // Round 4: on this device an f64 MFMA and the vector ALU's other instructions do not overlap (tools/mfma_f64_peak.hip:
    // 78 TFLOP/s with nothing else in the loop, 63 with 16 v_add_u32 per four MFMAs) -- and this loop had ~80 vector
    // instructions per 16 MFMAs: 64-bit address arithmetic, bounds selects, the copies from the next batch's registers into
    // the current one's (the matrix pipes were busy 69 % of the time).  Now: the full batches in pairs, the two register sets
    // taking turns (no copy); every load as uniform base + 32-bit byte offset, one v_add_u32 per slice and batch; no bounds
    // test (full batches).  Batches in the same ascending order: the same bits.  The rest goes through the checked loop below.
    const char *Ka = reinterpret_cast<const char *>(a.K + (size_t)ba * 32), *Kb = reinterpret_cast<const char *>(a.K + (size_t)bc * 32);
    const unsigned row_bytes = (unsigned)a.ldk * 8u;
    int nfull = 0;
    while (site_of(nfull) + 4 * NB <= a.n) ++nfull;
    const int npairs = ((unsigned long long)a.n * a.ldk * 8ull < (1ull << 32)) ? nfull / 2 : 0;  // (32-bit byte offsets; else everything through the checked loop)
    int b = 2 * npairs;
    if (npairs > 0) {
        unsigned kx[NB], ky[NB], wx[NB], wy[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const unsigned ix = (unsigned)(site_of(0) + 4 * t + lk), iy = (unsigned)(site_of(1) + 4 * t + lk);
            kx[t] = ix * row_bytes + (unsigned)lc * 8u;
            ky[t] = iy * row_bytes + (unsigned)lc * 8u;
            wx[t] = ix * 8u;
            wy[t] = iy * 8u;
        }
        const unsigned kstep = (unsigned)(site_of(2) - site_of(0)) * row_bytes, wstep = (unsigned)(site_of(2) - site_of(0)) * 8u;
        auto loadf = [&](unsigned (&ko)[NB], unsigned (&wo)[NB], double (&x0)[NB], double (&x1)[NB], double (&y0)[NB], double (&y1)[NB], double (&ww)[NC][NB]) {
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                x0[t] = *reinterpret_cast<const double *>(Ka + ko[t]);
                x1[t] = *reinterpret_cast<const double *>(Ka + ko[t] + 128);
                y0[t] = *reinterpret_cast<const double *>(Kb + ko[t]);
                y1[t] = *reinterpret_cast<const double *>(Kb + ko[t] + 128);
#pragma unroll
                for (int c = 0; c < NC; ++c) ww[c][t] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(om[c]) + wo[t]);
                ko[t] += kstep;
                wo[t] += wstep;
            }
        };
        loadf(kx, wx, a0, a1, b0, b1, w);
        for (int k = 0; k + 1 < npairs; ++k) {
            loadf(ky, wy, na0, na1, nb0, nb1, nw);
            mm(a0, a1, b0, b1, w);
            loadf(kx, wx, a0, a1, b0, b1, w);
            mm(na0, na1, nb0, nb1, nw);
        }
        loadf(ky, wy, na0, na1, nb0, nb1, nw);
        mm(a0, a1, b0, b1, w);
        mm(na0, na1, nb0, nb1, nw);
    }
    if (site_of(b) < a.n) load(site_of(b), a0, a1, b0, b1, w);
    for (; site_of(b) < a.n; ++b) {  // what is left: at most one full batch and a partial one
        if (site_of(b + 1) < a.n) load(site_of(b + 1), na0, na1, nb0, nb1, nw);
        mm(a0, a1, b0, b1, w);
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
            for (int v = 0; v < 4; ++v) s_g32[(((size_t)tile * WV + wave) * 64 + lane) * 4 + v] = acc[c][tile][v];
        __syncthreads();
        if (on[c] && wave < 4 && !(diag && wave == 2)) {
            const int ra = ba * 32 + (wave >> 1) * 16, rc = bc * 32 + (wave & 1) * 16;  // tile `wave`: rows ra.., columns rc..
            double *G = a.gram + (size_t)(chain0 + c) * a.m * a.m;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                double t = s_g32[(((size_t)wave * WV + 0) * 64 + lane) * 4 + v];
#pragma unroll
                for (int ww = 1; ww < WV; ++ww) t += s_g32[(((size_t)wave * WV + ww) * 64 + lane) * 4 + v];  // fixed order
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
constexpr int GRAM32_WAVES = 8;  // waves per workgroup of k_rsr_gram32: two workgroups per CU, one's epilogue beside the other's loop (16: 1.25 against 1.20 ms)
__host__ __device__ constexpr size_t rsr_gram32_lds(int wv) { return (size_t)4 * wv * 64 * 4 * sizeof(double); }

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
//   k_rsrb_step      the blocked upper Cholesky factorisation, panels of RSR_PANEL rows, one launch per panel (see there)
//   k_rsrb_solve     U'y = rhs, U theta = y, blocked by panels                          one workgroup per chain
// k_rsr_gram and k_rsr_eta_beta are the general kernels above.  Every sum has a fixed order (no atomics); the order
// is not the small path's (the two agree to rounding, like the oracle).  Plain kernels: at m = 1 280 the conditional is
// ~10^9 flops of Cholesky per chain and iteration beside a 2 10^10-flop Gram matrix -- milliseconds where the
// reference's host code takes seconds -- and not the path BASELINE's metric is quoted on.
// K'u for large bases (round 4).  k_rsr_gram's K'u workgroups each form u for themselves and push K through MFMAs whose B
// operand has one non-zero column, chain by chain: 80 workgroups per chain, 0.1 ms at m = 1 280.  Here: u once per chain
// (k_rsrb_u), then one workgroup per basis column takes that column -- a contiguous row of the transposed copy of K -- once
// for FOUR chains (k_rsrb_ktu).  Sums in a fixed order: a thread's sites ascending, the 64 lanes (wave_sum), the four waves.
__global__ void __launch_bounds__(256) k_rsrb_u(const RsrArgs a, int e, int sync_on)
{
    __shared__ int s_noise_ok;
    const int chain = blockIdx.y, i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop) return;
    if (sync_on && a.sync != nullptr) {  // the noise of this iteration comes from the side stream's previous sequence
        if (threadIdx.x == 0) s_noise_ok = sync_wait(a.sync, SYNC_NOISE, a.sync[SYNC_MAIN_SEQ + e]) ? 1 : 0;
        __syncthreads();
        if (!s_noise_ok) {
            if (threadIdx.x == 0) a.scs[chain].err = -2;  // OCC_E_HIP: the side stream never arrived
            return;
        }
    }
    if (i >= a.n) return;
    const size_t ci = (size_t)chain * a.n + i;
    const double w = a.omega_b[ctl.it & 1][ci], ev = a.enorm[ctl.it & 1][ci], zv = (double)a.z[ci];
    double xb = 0.0;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
        const double x = a.Xt[(size_t)min(j, a.p - 1) * a.n + i], b = (j < a.p) ? sc.beta[min(j, a.p - 1)] : 0.0;
        xb = fma(x, b, xb);
    }
    a.big_u[ci] = fma(sqrt(w), ev, fma(-w, xb, zv - 0.5));
}
__global__ void __launch_bounds__(256) k_rsrb_ktu(const RsrArgs a, int e)
{
    __shared__ double s_p[4][4];
    const int col = blockIdx.x, chain0 = (int)blockIdx.y * 4, tid = threadIdx.x, n = a.n;
    bool on[4];
    const double *u[4];
    bool any = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int chain = min(chain0 + c, a.C - 1);
        const ChainScalars &sc = a.scs[chain];
        const Ctl ctl = sc.ctl[e];
        on[c] = chain0 + c < a.C && !(ctl.koff || ctl.it >= sc.it_stop) && sc.err == 0;
        u[c] = a.big_u + (size_t)chain * n;
        any = any || on[c];
    }
    if (!any) return;
    const double *kr = a.Kt + (size_t)col * n;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    constexpr int UB = 4;
    for (int i0 = tid; i0 < n; i0 += 256 * UB) {
        double kv[UB], uv[4][UB];
#pragma unroll
        for (int b = 0; b < UB; ++b) {
            const int i = min(i0 + 256 * b, n - 1);
            kv[b] = (i0 + 256 * b < n) ? kr[i] : 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c) uv[c][b] = u[c][i];
        }
#pragma unroll
        for (int b = 0; b < UB; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = fma(kv[b], uv[c][b], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double sm = wave_sum(acc[c]);
        if ((tid & 63) == 0) s_p[tid >> 6][c] = sm;
    }
    __syncthreads();
    const unsigned onm = (on[0] ? 1u : 0u) | (on[1] ? 2u : 0u) | (on[2] ? 4u : 0u) | (on[3] ? 8u : 0u);
    if (tid < 4 && ((onm >> tid) & 1u)) a.rhs[(size_t)(chain0 + tid) * a.nchunk * a.m + col] = ((s_p[0][tid] + s_p[1][tid]) + s_p[2][tid]) + s_p[3][tid];
}

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
// (u[r] = entry (r, c)).  Row i: the pivot comes by v_readlane from lane i, the row is multiplied by its reciprocal root
// (v_rsq_f64 + two Newton steps, as in the small path; the stored diagonal holds 1 / U_ii -- nothing downstream needs U_ii
// itself), and a later row r takes its update with U_ir read from lane r -- no barrier.  (Round 2: the block in LDS, three
// workgroup barriers per row: 30 of the panel kernel's 38 us.)
// Round 4, second pass (tools/rsrb_stamps.py: 7.7 us of a 19 us step for the 496 row pairs, five instructions each): the
// block as 2 x 2 blocks of 16.  Rows 0-15 update only rows up to 15 (120 pairs; lanes 16-31 ride along, which IS
// U11' U12 = A12); then A22 -= U12' U12 on the matrix cores -- four v_mfma_f64_16x16x4_f64, operands and tile passed through
// `scr` (512 doubles of LDS; the wave's own, no barrier) -- and rows 16-31 among themselves (120 pairs).  The update is no
// longer masked to lanes >= r: entries below the diagonal take finite garbage that nothing reads (a pair is two v_readlane
// and one FMA).  Row i + 1 is updated first and its pivot's chain (readlane, v_rsq_f64, Newton: ~130 cycles) started before
// the other rows' updates, which hide it.  Returns false on a pivot <= 0.
#ifdef OCC_SOLVE_STAMPS
#define RSRB_SUB(pt) if (stamp) g_solve_stamps[64 + 24 + (pt)] = wall_clock64();
#else
#define RSRB_SUB(pt)
#endif
template <int I0>
__device__ __forceinline__ bool rsrb_factor16(double (&u)[RSR_PANEL], int kb, int lane)
{
    bool ok = true;
    double piv = readlane_f64(u[I0], I0);
    bool good = piv > 0.0;
    double rinv = rsqrt_pivot(good ? piv : 1.0);
#pragma unroll
    for (int i = I0; i < I0 + 16; ++i) {
        if (i < kb && !good) ok = false;
        u[i] = (lane == i) ? rinv : u[i] * rinv;  // (lanes < i: below the diagonal, never read)
        if (i + 1 < I0 + 16) {
            u[i + 1] = fma(-readlane_f64(u[i], i + 1), u[i], u[i + 1]);
            piv = readlane_f64(u[i + 1], i + 1);
            good = piv > 0.0;
            rinv = rsqrt_pivot(good ? piv : 1.0);
            asm volatile("" : "+v"(rinv));  // (keeps the chain's start here, ahead of the updates below)
        }
#pragma unroll
        for (int r = i + 2; r < I0 + 16; ++r) u[r] = fma(-readlane_f64(u[i], r), u[i], u[r]);  // U_ir from lane r
    }
    return ok;
}
__device__ __forceinline__ bool rsrb_diag_factor(double (&u)[RSR_PANEL], int kb, int lane, double *scr, bool stamp = false)
{
    RSRB_SUB(0)
    static_assert(RSR_PANEL == 32, "two blocks of 16");
    const bool ok0 = rsrb_factor16<0>(u, kb, lane);
    RSRB_SUB(1)
    double *T = scr, *S = scr + 256;  // T[k][j] = U12 (row k < 16, column 16 + j), S[r][j] = A22
    const int j = lane & 15, lk = lane >> 4;
    if (lk == 1) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            T[16 * k + j] = u[k];
            S[16 * k + j] = u[16 + k];
        }
    }
    v4d acc;
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] = S[16 * (4 * v + lk) + j];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const double b = T[16 * (4 * s4 + lk) + j];  // A = -U12' and B = U12 take the same entry in this lane
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-b, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) S[16 * (4 * v + lk) + j] = acc[v];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const double sv = S[16 * r + j];
        u[16 + r] = (lk == 1) ? sv : u[16 + r];
    }
    RSRB_SUB(2)
    const bool ok1 = rsrb_factor16<16>(u, kb, lane);
    RSRB_SUB(3)
    return ok0 && ok1;
}
// The inverses of the factor's two diagonal blocks of 16 come with it: lanes 32 + c start as the unit vector e_c in rows
// 0-15 and lanes 48 + c as e_c in rows 16-31, and the row operations of the factorisation -- which every lane takes --
// turn a column b into U^-T b (the forward substitution that also carries the right-hand side): afterwards lane 32 + c
// holds row c of V11 = U11^-1 in u[0..15] and lane 48 + c row c of V22 = U22^-1 in u[16..31].  With them the panel's
// triangular solves are products on the matrix cores (rsrb_apply) instead of 496 dependent FMAs per column (5 us of a step,
// tools/rsrb_stamps.py), and U12 is still where rsrb_diag_factor staged it (scr[16 k + j] = U_k,16+j).
constexpr int RSRB_VS = 17;  // row stride of V11 / V22 in LDS (the writing lanes own rows)
__device__ __forceinline__ void rsrb_store_inverses(const double (&u)[RSR_PANEL], double *Vs, int lane)
{
    if (lane >= 32) {
        const int c = lane & 15;
        if (lane < 48) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Vs[RSRB_VS * c + r] = u[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) Vs[16 * RSRB_VS + RSRB_VS * c + r] = u[16 + r];
        }
    }
}
// X = U^-T P for 16 columns (block cb) of a 32-row panel on the matrix cores, one wave: X1 = V11' P1, then
// P2 - U12' X1, then X2 = V22' (...).  An MFMA's result tile (lane (lk, lc): rows 4 v + lk, column lc) IS the next
// product's B operand, slice by slice -- nothing moves between the three.  Pb: the panel's entries in LDS.
__device__ __forceinline__ void rsrb_apply(const double *U12, const double *Vs, const double (&Pb)[RSR_PANEL][RSR_PANEL + 1], int cb, int lc, int lk,
                                           v4d &x1, v4d &x2)
{
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    v4d p1, p2;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        p1[v] = Pb[4 * v + lk][16 * cb + lc];
        p2[v] = Pb[16 + 4 * v + lk][16 * cb + lc];
    }
    x1 = zero4;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vs[RSRB_VS * (4 * s4 + lk) + lc], p1[s4], x1, 0, 0, 0);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-U12[16 * (4 * s4 + lk) + lc], x1[s4], p2, 0, 0, 0);
    x2 = zero4;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vs[16 * RSRB_VS + RSRB_VS * (4 * s4 + lk) + lc], p2[s4], x2, 0, 0, 0);
}

// ---- the blocked factorisation: ONE kernel per panel step ------------------------------------------------------------
// Launch `k0` does the trailing update of panel step k0 (P_ij -= sum_t U_ti U_tj over the panel's 32 rows t, one WAVE per
// 16 x 16 tile of the trailing upper triangle: v_mfma_f64_16x16x4_f64 with A = -U[:, i-block]' and B = U[:, j-block], the
// tile as the accumulator; the right-hand side's row: rhs_j -= sum_t U_tj y_t) AND the panel work of step k0 + 32: the
// workgroups of the first tile row pair -- whose tiles are the next panel's 32 rows -- go on after their update.  Each forms
// the NEXT diagonal block's update and factor for itself (three 16 x 16 tiles and one wave's work: cheaper than a hand-over
// between workgroups, and nothing is assumed about the order workgroups are dispatched in), then X = U_kk^-T P for its own 32
// columns from the tiles it holds (rsrb_apply); the right-hand side's workgroup does the same for the panel's 32 entries (the
// forward substitution travels with the factorisation, as in the small path).  The launch with k0 < 0 is the head: no
// update, the first panel's work.  (Until round 4: a panel and an update kernel per step, 2 x 40 launches per iteration at
// m = 1 280, 23 + 12 us and two launch gaps per step: 1.7 ms of a 3.5 ms iteration; the triangular solves one column per
// thread.)
// Order inside a workgroup of the first tile row (tools/rsrb_stamps.py): waves 0-2 form the diagonal block's three tiles
// FIRST (wave 3 its own tile meanwhile); wave 3 factors and inverts while waves 0-2 form their own tiles -- the factor, the
// one serial piece, waits for nothing but the three tiles it needs.  Grid: x = tile column pair, y = (row, chain) with row 0
// the right-hand side's and row 1 the first tile row pair, so the workgroups with the serial work are dispatched first, for
// every chain.
__device__ __forceinline__ v4d rsrb_tile_update(const double *P, int m, int k0, int kb, int i0, int j0, int lc, int lk)
{
    // (every load at a clamped, valid address and the value selected afterwards: a conditional load is a branch each)
    v4d acc;
    const int cj = min(j0 + lc, m - 1), ci = min(i0 + lc, m - 1);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int r = i0 + 4 * v + lk;
        const double p = P[(size_t)min(r, m - 1) * m + cj];
        acc[v] = (r < m && j0 + lc < m) ? p : 0.0;
    }
    double av[RSR_PANEL / 4], bv[RSR_PANEL / 4];
#pragma unroll
    for (int s4 = 0; s4 < RSR_PANEL / 4; ++s4) {
        const int t = 4 * s4 + lk;
        const bool in = t < kb;
        const double *row = P + (size_t)(k0 + (in ? t : 0)) * m;
        const double ua = row[ci], ub = row[cj];
        av[s4] = (in && i0 + lc < m) ? -ua : 0.0;
        bv[s4] = (in && j0 + lc < m) ? ub : 0.0;
    }
#pragma unroll
    for (int s4 = 0; s4 < RSR_PANEL / 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc, 0, 0, 0);
    return acc;
}
#ifdef OCC_SOLVE_STAMPS  // tools/rsrb_stamps.py: the second workgroup of the first tile row, chain 0, panel step 20
#define RSRB_STAMP(pt) if (chain == 0 && row == 1 && blockIdx.x == 1 && (threadIdx.x & 63) == 0 && k0 == 20 * RSR_PANEL) g_solve_stamps[64 + 4 * (pt) + (threadIdx.x >> 6)] = wall_clock64();
#else
#define RSRB_STAMP(pt)
#endif
__global__ void __launch_bounds__(256) k_rsrb_step(const RsrArgs a, int e, int k0, int n_chains)
{
    __shared__ double Scr[512], Dblk[RSR_PANEL][RSR_PANEL + 1], Xblk[RSR_PANEL][RSR_PANEL + 1];  // Scr: rsrb_diag_factor's, then U12
    __shared__ int s_bad;
    const int row = (int)blockIdx.y / n_chains, chain = (int)blockIdx.y % n_chains, m = a.m;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lc = lane & 15, lk = lane >> 4;
    ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop || sc.err != 0) return;
    const bool head = k0 < 0;
    const int kr = head ? 0 : k0, kb = head ? 0 : min(RSR_PANEL, m - k0), base = kr + kb, kb2 = min(RSR_PANEL, m - base);  // (base < m)
    double *P = a.gram + (size_t)chain * m * m;
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    const bool rhs_row = row == 0, first = blockIdx.x == 0;
    // tile (ti, tj) of the trailing block, ti <= tj: workgroup (bx, row) holds tiles (2 (row - 1) + wave / 2, 2 bx + wave % 2)
    const int ti = 2 * (row - 1) + (wave >> 1), tj = 2 * (int)blockIdx.x + (wave & 1);
    const int i0 = base + 16 * ti, j0 = base + 16 * tj;
    const bool have = !rhs_row && !(tj < ti || i0 >= m || j0 >= m);
    if (row > 1) {  // ---- the trailing update, nothing else
        if (have) {
            const v4d acc = rsrb_tile_update(P, m, kr, kb, i0, j0, lc, lk);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = i0 + 4 * v + lk, c = j0 + lc;
                if (r < m && c < m && c >= r) P[(size_t)r * m + c] = acc[v];
            }
        }
        return;
    }
    double *rhs = a.big_rhs + (size_t)chain * m;
    double rv = 0.0;
    if (rhs_row) {  // ---- the right-hand side's update; workgroup 0 goes on to the next panel's entries
        // (the panel's 32 entries of this column all loaded before the first is used: as a loop over a run-time kb the
        // compiler took them one round trip at a time, and this workgroup is on every launch's critical path)
        const int j = base + (int)blockIdx.x * 256 + tid, jc = min(j, m - 1);
        double pv[RSR_PANEL];
#pragma unroll
        for (int t = 0; t < RSR_PANEL; ++t) pv[t] = P[(size_t)(kr + (t < kb ? t : 0)) * m + jc];
        rv = rhs[jc];
#pragma unroll
        for (int t = 0; t < RSR_PANEL; ++t) rv = fma(-pv[t], (t < kb) ? rhs[kr + t] : 0.0, rv);
        if (j < m && (!first || tid >= kb2)) rhs[j] = rv;
        if (!first) return;
    }
    RSRB_STAMP(0)
    // ---- the next panel (rows base .. base + kb2): its diagonal block, updated by step k0, into Dblk.  The first workgroup
    // of the tile row holds the block's tiles (waves 0, 1, 3: (0,0), (0,1), (1,1)); elsewhere waves 0-2 form them.
    v4d acc = zero4;
    if (first && !rhs_row) {
        if (have) acc = rsrb_tile_update(P, m, kr, kb, i0, j0, lc, lk);
        if (wave != 2) {
#pragma unroll
            for (int v = 0; v < 4; ++v) Dblk[16 * (wave >> 1) + 4 * v + lk][16 * (wave & 1) + lc] = acc[v];
        }
    } else if (wave < 3) {
        const int di = wave == 2 ? 1 : 0, dj = wave == 0 ? 0 : 1;
        const v4d d = rsrb_tile_update(P, m, kr, kb, base + 16 * di, base + 16 * dj, lc, lk);
#pragma unroll
        for (int v = 0; v < 4; ++v) Dblk[16 * di + 4 * v + lk][16 * dj + lc] = d[v];
    } else if (have) {
        acc = rsrb_tile_update(P, m, kr, kb, i0, j0, lc, lk);
    }
    __syncthreads();
    RSRB_STAMP(1)
    if (wave == 3) {  // the factor and its diagonal blocks' inverses, by one wave
        double u[RSR_PANEL];
#pragma unroll
        for (int r = 0; r < RSR_PANEL; ++r) {
            const double d = Dblk[r][lane & (RSR_PANEL - 1)];
            const bool in = lane < kb2 && r < kb2 && lane >= r;
            // outside a short block: the identity; lanes 32-63: the unit vectors that become V11 and V22
            const int unit = lane < 32 ? lane : lane - 32;  // (lanes 48-63: rows 16-31)
            u[r] = in ? d : ((r == unit && (lane < 32 || (r < 16) == (lane < 48))) ? 1.0 : 0.0);
        }
#ifdef OCC_SOLVE_STAMPS
        const bool stamp = chain == 0 && row == 1 && blockIdx.x == 1 && lane == 0 && k0 == 20 * RSR_PANEL;
        const bool ok = rsrb_diag_factor(u, kb2, lane, Scr, stamp);
#else
        const bool ok = rsrb_diag_factor(u, kb2, lane, Scr);
#endif
        if (lane == 0) s_bad = ok ? 0 : 1;
        RSRB_SUB(4)
        if (first && !rhs_row) {
            // the factored diagonal block, for k_rsrb_solve (P's diagonal block is never read again): F[q][t] = U_qt,
            // row-major, 1 / U_tt on the diagonal -- from the registers, a row of the block per store instruction
            if (ok && lane < RSR_PANEL) {
                double *F = a.big_dfac + ((size_t)chain * ((m + RSR_PANEL - 1) / RSR_PANEL) + base / RSR_PANEL) * (RSR_PANEL * RSR_PANEL);
#pragma unroll
                for (int q = 0; q < RSR_PANEL; ++q) F[q * RSR_PANEL + lane] = (q <= lane) ? u[q] : 0.0;
            }
        } else {
            rsrb_store_inverses(u, &Dblk[0][0], lane);  // (Dblk is this wave's alone since the barrier)
        }
    } else if (!first && have) {  // meanwhile: this workgroup's own tiles
        acc = rsrb_tile_update(P, m, kr, kb, i0, j0, lc, lk);
    }
    RSRB_STAMP(2)
    if (rhs_row) {
        if (tid < RSR_PANEL) Xblk[tid][0] = (tid < kb2) ? rv : 0.0;  // the panel's updated entries: column 0 of a block like the others'
    } else if (!first) {
#pragma unroll
        for (int v = 0; v < 4; ++v) Xblk[16 * (wave >> 1) + 4 * v + lk][16 * (wave & 1) + lc] = acc[v];  // (zero where there is no tile)
    }
    __syncthreads();
    RSRB_STAMP(3)
    if (s_bad) {
        if (first && !rhs_row && tid == 0) sc.err = -4;  // OCC_E_CHOLESKY
        return;
    }
    if (first && !rhs_row) return;  // (wave 3 has stored the factored block)
    // X = U_kk^-T P: this workgroup's 32 columns of the block row, 16 per wave (waves 0, 1); the right-hand side's
    // workgroup: the panel's entries, column 0 of wave 0's block
    if (wave < (rhs_row ? 1 : 2)) {
        v4d x1, x2;
        rsrb_apply(Scr, &Dblk[0][0], Xblk, wave, lc, lk, x1, x2);
        RSRB_STAMP(4)
        if (rhs_row) {
            if (lc == 0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (4 * v + lk < kb2) rhs[base + 4 * v + lk] = x1[v];
                    if (16 + 4 * v + lk < kb2) rhs[base + 16 + 4 * v + lk] = x2[v];
                }
            }
        } else {
            const int j = base + 32 * (int)blockIdx.x + 16 * wave + lc;
            if (j < m) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (4 * v + lk < kb2) P[(size_t)(base + 4 * v + lk) * m + j] = x1[v];
                    if (16 + 4 * v + lk < kb2) P[(size_t)(base + 16 + 4 * v + lk) * m + j] = x2[v];
                }
            }
        }
    }
    RSRB_STAMP(5)
}

// U theta = y (backward), blocked by panels, one workgroup per chain; y = the right-hand side as the panel steps left it
// (the forward substitution travels with the factorisation).
// Round 4: LEFT-looking -- y_k = U_kk^-1 (y_k - sum_{panels j > k} U_kj y_j) with the sums taken along U's ROWS, which is how
// the factor lies in memory (a wave reads 512 contiguous bytes per load), and pipelined: while wave 0 solves panel k's
// triangle (32 dependent steps, the block from LDS), waves 1-15 form panel k - 1's sums over every entry already solved
// (`far`), fetch its diagonal block and its 32 x 32 coupling to panel k (`near`, finished right after panel k).  A panel
// costs the longer of the two; until then every panel's solve was followed by an update of ALL rows above it through
// 256-byte column slabs, one dependent pair per panel (9 us a panel, 0.36 ms at m = 1 280).  Every sum has a fixed order.
__global__ void __launch_bounds__(1024) k_rsrb_solve(const RsrArgs a, int e)
{
    __shared__ double y[RSR_BIG_MAX];
    __shared__ double Fs[2][RSR_PANEL][RSR_PANEL + 1];  // the diagonal blocks' factors (F[q][t] = U_qt, 1 / U_tt on the diagonal)
    __shared__ double s_far[2][RSR_PANEL], s_near[RSR_PANEL];
    const int chain = blockIdx.y, m = a.m, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop || sc.err != 0) return;
    const double *U = a.gram + (size_t)chain * m * m;
    const double *F = a.big_dfac + (size_t)chain * ((m + RSR_PANEL - 1) / RSR_PANEL) * (RSR_PANEL * RSR_PANEL);
    const int npan = (m + RSR_PANEL - 1) / RSR_PANEL;
    for (int t = tid; t < m; t += 1024) y[t] = a.big_rhs[(size_t)chain * m + t];
    {  // the last panel: nothing beyond it
        const double *Fk = F + (size_t)(npan - 1) * (RSR_PANEL * RSR_PANEL);
        Fs[(npan - 1) & 1][tid >> 5][tid & 31] = Fk[tid];
        if (tid < RSR_PANEL) { s_far[(npan - 1) & 1][tid] = 0.0; s_near[tid] = 0.0; }
    }
    // `near`: wave w takes rows 2 w and 2 w + 1 of a panel, lanes 0-31 / 32-63 the 32 columns of the panel after it
    const int nrow = 2 * wave + (lane >> 5), ncol = lane & 31;
    double un = 0.0;  // U[row nrow of the panel about to be solved][column ncol of the panel solved before it]
    __syncthreads();
    for (int k = npan - 1; k >= 0; --k) {
        const int k0 = k * RSR_PANEL, kb = min(RSR_PANEL, m - k0);
        if (k < npan - 1) {  // panel k + 1 has just been solved: its part of panel k's sums
            const int c = k0 + RSR_PANEL + ncol;
            double v = (c < m) ? un * y[c] : 0.0;
            v += dpp_shifted<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
            v += dpp_shifted<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
            v += dpp_shifted<0x141, 0xf>(v);  // row_half_mirror
            v += dpp_shifted<0x140, 0xf>(v);  // row_mirror: every lane of a row of 16 holds the row's sum
            v += __shfl_xor(v, 16);           // the two rows of 16 of this half
            if (ncol == 0) s_near[nrow] = v;
            __syncthreads();
        }
        // the coupling of panel k - 1 to panel k, fetched now, used after this panel's solve
        if (k > 0) {
            const int r = k0 - RSR_PANEL + nrow, c = k0 + ncol;
            un = U[(size_t)r * m + min(c, m - 1)];
        }
        if (wave == 0) {  // ---- panel k's triangle: U_kk y_k = y_k - far - near, from the bottom
            const int r = lane & (RSR_PANEL - 1);
            double v = (lane < kb) ? (y[k0 + lane] - s_far[k & 1][r]) - s_near[r] : 0.0;
            double fr[RSR_PANEL];
#pragma unroll
            for (int s2 = 0; s2 < RSR_PANEL; ++s2) {
                const double f = Fs[k & 1][r][s2];
                fr[s2] = (lane < kb && s2 < kb && lane <= s2) ? f : 0.0;
            }
            double dg = 1.0;
#pragma unroll
            for (int s2 = 0; s2 < RSR_PANEL; ++s2) dg = (lane == s2) ? fr[s2] : dg;
#pragma unroll
            for (int s2 = RSR_PANEL - 1; s2 >= 0; --s2) {
                if (s2 < kb) {
                    const double ts = readlane_f64(v, s2) * readlane_f64(dg, s2);  // (the stored diagonal is 1 / U_ss)
                    v = (lane == s2) ? ts : fma(-fr[s2], ts, v);  // column s2 of the block (fr is zero at and below the diagonal)
                }
            }
            if (lane < kb) y[k0 + lane] = v;
        } else if (k > 0) {  // ---- meanwhile, for panel k - 1: its diagonal block, and its sums over everything solved before panel k
            const int t = tid - 64;  // 0 .. 959
            const double *Fk = F + (size_t)(k - 1) * (RSR_PANEL * RSR_PANEL);
            for (int q = t; q < RSR_PANEL * RSR_PANEL; q += 960) Fs[(k - 1) & 1][q >> 5][q & 31] = Fk[q];
            // rows wave - 1, wave + 14, wave + 29 of panel k - 1, columns k0 + 32 .. m - 1 over the lanes
            const int r0 = wave - 1, nr = (r0 + 30 < RSR_PANEL) ? 3 : 2, j0 = k0 + RSR_PANEL;
            const double *row = U + (size_t)(k0 - RSR_PANEL + r0) * m;
            double acc[3] = {0.0, 0.0, 0.0};
            constexpr int FB = 8;  // column chunks (of 64) in flight per row
            for (int jb = j0; jb < m; jb += 64 * FB) {
                double t0[FB], t1[FB], t2[FB];
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    const int j = min(jb + 64 * b + lane, m - 1);
                    t0[b] = row[j];
                    t1[b] = row[(size_t)15 * m + j];
                    t2[b] = (nr == 3) ? row[(size_t)30 * m + j] : 0.0;
                }
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    const int j = jb + 64 * b + lane;
                    const double yj = (j < m) ? y[j] : 0.0;
                    acc[0] = fma(t0[b], yj, acc[0]);
                    acc[1] = fma(t1[b], yj, acc[1]);
                    acc[2] = fma(t2[b], yj, acc[2]);
                }
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const double sm = wave_sum(acc[q]);
                if (lane == 0 && q < nr) s_far[(k - 1) & 1][r0 + 15 * q] = sm;
            }
        }
        __syncthreads();
    }
    for (int t = tid; t < m; t += 1024) a.theta[(size_t)chain * m + t] = y[t];
}

}  // namespace occ
