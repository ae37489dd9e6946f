// occ_solve.hpp -- the eta solve as ONE persistent launch (gfx950).
//
// k_minres (occ_kernels.hpp) spends one kernel launch per MINRES iteration: at the headline size
// (100x100 sites, 4 chains) a launch moves 7.8 MB -- one microsecond of HBM time -- and costs 5.7-7.3 us,
// all of it launch boundary and cold dependent loads; the number of launches per solve has to be
// guessed (captured graph), so a few of them are empty and the pipeline tail adds three more.
// k_solve keeps the whole solve of a chain on the chip instead:
//   * one 256-thread workgroup per 256 sites, every workgroup of a chain resident at once (the host only
//     takes this path when all of them fit, one per CU);
//   * the vectors of the recurrence (g, p_{k-2}, p_{k-3}, w_{k-3}, w_{k-4}, x) live in registers, and so do
//     p_{k-2}, p_{k-3} AT THE NEIGHBOURS of a site (each lane re-forms them from the same scalars, as
//     k_minres does), so the only vector a step exchanges is g_k = A p_{k-1}: 16 B per site written, 8
//     gathers of 16 B read;
//   * one barrier per step AMONG THE WORKGROUPS OF ONE CHAIN (40 at the headline size): payload stored
//     write-through (sc1), every storing wave drained, one lane adds to the chain's arrival counter and
//     polls it, then every load of the exchanged bytes is an sc1 load (per-CU L1 bypassed; the XCDs' L2s
//     are not coherent with each other, so visibility never depends on where a workgroup runs);
//   * the stopping test runs on the device, so every chain runs exactly the steps its solve needs.
// The arithmetic is k_minres's, through the same functions (minres_scalars, kry_form_p, kry_form_w), with
// partial sums per 64-site slice reduced in the same order: a solve returns the same bits on either path.
//
// Reference: scipy.sparse.linalg.minres as called by _EtaICARPosterior.rvs (occuspytial/gibbs/logit.py:82-92).
#pragma once
#include "occ_kernels.hpp"

namespace occ {

typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int SOLVE_WG = 256;               // threads per workgroup of k_solve
constexpr unsigned SOLVE_SPIN_LIMIT = 1u << 21;  // polls (about a microsecond each) before a barrier gives up
constexpr int BAR_STRIDE = 32;              // unsigned words between the arrival counters of two chains (128 B)

// Developer builds (-DOCC_SOLVE_STAMPS, `make stamps`) record s_memtime at a few points of every step of
// chain 0 / workgroup 0; the product build compiles the hooks away.
#ifdef OCC_SOLVE_STAMPS
constexpr int STAMP_STEPS = 48, STAMP_POINTS = 12;
__device__ unsigned long long g_solve_stamps[STAMP_STEPS * STAMP_POINTS];
#define SOLVE_STAMP(pt)                                                                                  \
    if (chain == 0 && wg == 0 && threadIdx.x == 0 && k < STAMP_STEPS) g_solve_stamps[k * STAMP_POINTS + (pt)] = __builtin_readcyclecounter();
#else
#define SOLVE_STAMP(pt)
#endif

struct SolveArgs {
    KryArgs a;
    unsigned *bar;   // [C][BAR_STRIDE] arrival counter of each chain, zeroed by k_eta_init
    double *part;    // [C][2][nb_n][4] partial sums of the running solve, by step parity
    int nbg;         // workgroups per chain
};

__device__ __forceinline__ v4u pack_d2(double2 v)
{
    v4u r;
    r.x = (unsigned)__double2loint(v.x); r.y = (unsigned)__double2hiint(v.x);
    r.z = (unsigned)__double2loint(v.y); r.w = (unsigned)__double2hiint(v.y);
    return r;
}
__device__ __forceinline__ double2 unpack_d2(v4u r)
{
    return make_double2(__hiloint2double((int)r.y, (int)r.x), __hiloint2double((int)r.w, (int)r.z));
}
// 16-byte write-through store / L1-bypassing load (aux 16 = sc1)
__device__ __forceinline__ void store_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, double2 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(pack_d2(v), r, byte_off, 0, 16);
}
__device__ __forceinline__ double2 load_sc1(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return unpack_d2(__builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}

__global__ void __launch_bounds__(SOLVE_WG) k_solve(const SolveArgs sa, int e)
{
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const KryArgs &a = sa.a;
    const int chain = blockIdx.y, wg = blockIdx.x;
    ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop) return;  // uniform over the chain's workgroups
    const int n = a.n, i = wg * SOLVE_WG + (int)threadIdx.x;
    const bool act = i < n;
    const int lane = threadIdx.x & 63, slice = i >> 6;
    const bool slice_act = slice < a.nb_n;  // a slice with at least one site owns a partial sum
    const size_t co = (size_t)chain * n;
    const double tau = sc.tau;
    const double2 zero2 = make_double2(0.0, 0.0);
    const bool writer = (wg == 0 && threadIdx.x == 0);
    unsigned *cnt = sa.bar + (size_t)chain * BAR_STRIDE;

    // ---- fixed data of this site: matrix row (<= NPRE off-diagonals, checked by the host), omega_b
    int off[NPRE];      // byte offset of neighbour kk in a [n] double2 array
    double av[NPRE];    // tau * Q_ij
    double d = 0.0;
    int width = 0;      // off-diagonals of this 64-row slice (uniform over the wave)
    double2 p0 = zero2, x = zero2;
    double2 nm1[NPRE], nm2[NPRE], ng[NPRE];
#pragma unroll
    for (int kk = 0; kk < NPRE; ++kk) { off[kk] = 0; av[kk] = 0.0; nm1[kk] = zero2; nm2[kk] = zero2; ng[kk] = zero2; }
    if (act) {
        int base;
        if (a.ell_w > 0) { width = a.ell_w; base = slice * a.ell_w * 64; }
        else { base = a.sell_ptr[slice]; width = (a.sell_ptr[slice + 1] - base) >> 6; }
        const double2 *P0 = a.Pv[0] + co;  // p_0 = b - A x0, stored by k_eta_init (an earlier launch: plain loads)
#pragma unroll
        for (int kk = 0; kk < NPRE; ++kk)
            if (kk < width) {
                const int j = a.sell_col[base + kk * 64 + lane];
                off[kk] = j * 16;
                av[kk] = tau * a.sell_val[base + kk * 64 + lane];
                ng[kk] = P0[j];  // p_0 at the neighbour: plays p_{k-1} at step 1
            }
        d = tau * a.qdiag[i] + a.omega_b[ctl.it & 1][co + i];
        p0 = P0[i];
        x = a.Xv[co + i];
    }
    const __amdgpu_buffer_rsrc_t gbuf[2] = {
        __builtin_amdgcn_make_buffer_rsrc((void *)(a.Gv[0] + co), 0, n * 16, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc((void *)(a.Gv[1] + co), 0, n * 16, 0x00020000)};
    double *part_base = sa.part + (size_t)chain * 2 * a.nb_n * 4;
    const __amdgpu_buffer_rsrc_t pbuf[2] = {
        __builtin_amdgcn_make_buffer_rsrc((void *)part_base, 0, a.nb_n * 32, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc((void *)(part_base + (size_t)a.nb_n * 4), 0, a.nb_n * 32, 0x00020000)};

    Slot s = {};
    double2 g = zero2, pm1 = zero2, pm2 = zero2, wm1 = zero2, wm2 = zero2;
    double S0 = 0.0, S1 = 0.0, S2 = 0.0, xn2 = 0.0;
    for (int k = 1;; ++k) {
        SOLVE_STAMP(0)
        const KryStep st = minres_scalars(s, k, S0, S1, S2, xn2, a.maxiter);
        if (st.stop) break;
        SOLVE_STAMP(1)
        double part[4] = {0.0, 0.0, 0.0, 0.0};
        if (st.rotate && act) {  // w_{k-2}, x_{k-2}
            const double2 w = kry_form_w(st, pm2, wm2, wm1);
            x.x = fma(st.phi, w.x, x.x);
            x.y = fma(st.phi, w.y, x.y);
            wm2 = wm1;
            wm1 = w;
            part[3] = dot2(x, x);
        }
        double2 gn = zero2;
        if (act) {
            const double2 p = (k == 1) ? p0 : kry_form_p(st, g, pm2, pm1);  // p_{k-1}
            double gx = d * p.x, gy = d * p.y;
#pragma unroll
            for (int kk = 0; kk < NPRE; ++kk) {
                if (kk >= width) break;
                const double2 pj = (k == 1) ? ng[kk] : kry_form_p(st, ng[kk], nm2[kk], nm1[kk]);
                gx = fma(av[kk], pj.x, gx);
                gy = fma(av[kk], pj.y, gy);
                nm2[kk] = nm1[kk];
                nm1[kk] = pj;
            }
            gn = make_double2(gx, gy);
            part[0] = dot2(p, p);
            part[1] = fma(p.y, gy, p.x * gx);
            if (k >= 2) part[2] = dot2(p, pm1);
            pm2 = pm1;
            pm1 = p;
            g = gn;
            store_sc1(gbuf[k & 1], i * 16, gn);
        }
        if (slice_act) {  // per-slice sums: the granularity (and order) of k_minres at 64 threads per block
            const double t0 = wave_sum(part[0]), t1 = wave_sum(part[1]), t2 = wave_sum(part[2]), t3 = wave_sum(part[3]);
            if (lane == 0) store_sc1(pbuf[k & 1], slice * 32, make_double2(t0, t1));
            if (lane == 1) store_sc1(pbuf[k & 1], slice * 32 + 16, make_double2(t2, t3));
        }
        // ---- barrier among the workgroups of this chain: arrival number k * nbg of a monotonic counter
        SOLVE_STAMP(2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its write-through stores have left
        SOLVE_STAMP(3)
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            SOLVE_STAMP(4)
            const unsigned target = (unsigned)k * (unsigned)sa.nbg;
            int fail = 0;
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SOLVE_SPIN_LIMIT) { fail = 1; break; }  // a workgroup of the chain is not running
            }
            s_fail = fail;
            SOLVE_STAMP(5)
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler only: no load moves above the poll
        SOLVE_STAMP(6)
        if (s_fail) {
            if (writer) {
                sc.err = -2;  // OCC_E_HIP: reported by the host as an over-subscribed persistent launch
                s.done = 1; s.istop = 6; s.itn = k;
            }
            break;
        }
        // ---- everything below reads what other workgroups published in this step: sc1 loads only
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int b = lane; b < a.nb_n; b += 64) {
            const double2 lo = load_sc1(pbuf[k & 1], b * 32), hi = load_sc1(pbuf[k & 1], b * 32 + 16);
            acc[0] += lo.x; acc[1] += lo.y; acc[2] += hi.x; acc[3] += hi.y;
        }
        if (act) {
#pragma unroll
            for (int kk = 0; kk < NPRE; ++kk)
                if (kk < width) ng[kk] = load_sc1(gbuf[k & 1], off[kk]);
        }
        SOLVE_STAMP(7)
        S0 = wave_sum(acc[0]); S1 = wave_sum(acc[1]); S2 = wave_sum(acc[2]); xn2 = wave_sum(acc[3]);
        SOLVE_STAMP(8)
    }
    // ---- the solve is over (uniformly over the chain): hand x and its projection sums to k_beta_partial
    if (writer) slot_store(&a.slots[(size_t)chain * NSLOT], s);
    double v0 = 0.0, v1 = 0.0;
    if (act) {
        a.Xv[co + i] = x;
        v0 = x.x;
        v1 = x.y;
    }
    if (slice_act) {
        const double t0 = wave_sum(v0), t1 = wave_sum(v1);
        if (lane == 0) {
            double *pp = a.part_proj + (size_t)chain * 2 * a.nb_n;
            pp[slice] = t0;
            pp[a.nb_n + slice] = t1;
        }
    }
}

}  // namespace occ
