// occ_iter.hpp -- k_iter: the critical path of one Gibbs iteration of every chain in ONE persistent launch (gfx950).
//
// nbg workgroups of 256 (or, one XCD per chain on larger lattices, 512: 448 sites beside a scalar wave, or 512 sites)
// threads per chain, one site per thread, ALL of them resident at once (the host takes this path only when they fit at
// most two per CU).  Per chain, in order:
//   A  tau ~ Gamma (logit.py:206-209), right-hand side of the eta system (logit.py:75-78, 213),
//      p_0 = b - A x0 with the warm start x0 (logit.py:71, 82-88)                         [k_eta_init]
//   B  joint MINRES for [x z] (logit.py:82-92), vectors in registers, one barrier per step AMONG THE WORKGROUPS
//      OF THE CHAIN; stopping test on the device                                           [k_minres x (K + 3)]
//   C  eta = x - (sum x / sum z) z (distributions.pyx:24-39), partial sums of beta's system
//      (logit.py:226-232)                                                                  [k_beta_partial]
// (in brackets: the stand-alone kernels of occ_kernels.hpp that do the same work with one launch per step; they
// remain the path for problems k_iter does not fit and the independent implementation it is tested against).
//
// Why one launch: at the headline size (100x100 sites, 4 chains) a k_minres launch moves 7.8 MB -- one
// microsecond of HBM time -- and costs 5.7-7.3 us of launch boundary and cold dependent loads, and the launches
// per solve have to be guessed when the graph is captured.  Here the solve runs exactly the steps it needs and
// chains do not wait for each other inside the launch.
//
// Exchange between the workgroups of a chain, two forms (template parameter XL).  Only g = A p (16 B per site) and
// four partial sums per 64-site slice are exchanged per step: p_{k-2}, p_{k-3} at the NEIGHBOURS of a site stay in
// the registers of the lane that re-formed them.
//   XL = 0, any placement: payload stored write-through (sc1), every storing wave drained, one lane adds to the
//     chain's arrival counter (agent scope) and polls it, then every load of exchanged bytes is an sc1 load (per-CU
//     L1 bypassed).  Visibility never depends on where a workgroup runs (the XCDs' L2s are not coherent with each
//     other), and every hop is a round trip to the memory side: 2.3 us per store -> barrier -> load (tools/xcc_probe4).
//   XL = 1, ONE XCD PER CHAIN: a workgroup works for the chain of the XCD it runs on (HW_REG_XCC_ID) and claims its
//     place among that chain's workgroups from a counter, so all workgroups of a chain share one XCD and its L2
//     whatever the dispatcher does with the grid.  Payload and arrival flags are PLAIN stores (they stay in that L2),
//     every load of exchanged bytes and every poll is an sc1 load (L1 bypassed, L2-served): 0.7 us per store ->
//     barrier -> load, no atomics in the loop.  In the solve a slice's RECORD of a step is its arrival flag (see "XL
//     step exchange" below); the barriers of phases A and C use one flag word per workgroup, which carries the XCC_ID
//     of its writer -- every poll checks it against the reader's and a launch whose placement differs fails loudly
//     (OCC_E_HIP) instead of reading another XCD's stale lines.  The host verifies residency with a probe launch
//     before it takes this form (create_impl).  With 512 threads per workgroup the first wave owns no sites and runs
//     the scalar recurrence for the seven site waves (template parameter W512 = 1; see k_iter).
// The arrival counter / flag values are monotonic over the whole run (ChainScalars::bar_base).
//
// The arithmetic is that of the stand-alone kernels, through the same functions (minres_pre/post, kry_form_*,
// eta_rhs_site, ...), with partial sums per 64-site slice reduced in the same order: k_iter, the
// launch-per-MINRES-step path and the eager stepping path return the same bits.
#pragma once
#include "occ_kernels.hpp"

namespace occ {

typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int ITER_WG = 256;                    // threads per workgroup of k_iter, any placement
constexpr int ITER_WG_XL = 512;                 // ... one XCD per chain: two waves per SIMD IN one workgroup (see k_iter)
constexpr int ITER_SITES_SW = 448;              // ... of which the first wave owns no sites (the scalar wave): seven site waves
constexpr unsigned ITER_SPIN_LIMIT = 1u << 21;  // polls (about a microsecond each) before a barrier gives up
constexpr unsigned ITER_PROBE_SPIN_LIMIT = 1u << 14;  // ... in the residency probe at creation (flags bit 1 of k_iter)
constexpr int BAR_STRIDE = 64;                  // unsigned words per chain in IterArgs::bar: the counter, or one flag per workgroup
constexpr int XL_MAX_WG = 64;                   // workgroups per chain of an XCD-local launch (one flag per lane of the polling wave)
constexpr int XL_SLOTS = 8;                     // chains of an XCD-local launch = XCDs the grid's x dimension walks over

// Developer builds (-DOCC_SOLVE_STAMPS, `make stamps`) record s_memtime at a few points of every MINRES step of
// chain 0 / workgroup 0; the product build compiles the hooks away.
#ifdef OCC_SOLVE_STAMPS
constexpr int STAMP_STEPS = 48, STAMP_POINTS = 16;
__device__ unsigned long long g_solve_stamps[STAMP_STEPS * STAMP_POINTS];
#define SOLVE_STAMP(pt)                                                                                  \
    if (chain == 0 && wg == 0 && threadIdx.x == 0 && k < STAMP_STEPS) g_solve_stamps[k * STAMP_POINTS + (pt)] = __builtin_readcyclecounter();
#define PHASE_STAMP(row, pt)                                                                             \
    if (chain == 0 && wg == 0 && threadIdx.x == 0) g_solve_stamps[(row) * STAMP_POINTS + (pt)] = __builtin_readcyclecounter();
#define SITE_STAMP(pt)                                                                                   \
    if (chain == 0 && wg == 0 && threadIdx.x == 64 && k < STAMP_STEPS) g_solve_stamps[k * STAMP_POINTS + (pt)] = __builtin_readcyclecounter();
#else
#define SOLVE_STAMP(pt)
#define SITE_STAMP(pt)
#define PHASE_STAMP(row, pt)
#endif

struct IterArgs {
    KryArgs a;            // the MINRES descriptor of k_minres (matrix, omega_b, G buffers, x, scalars)
    // phase A / C inputs and outputs by value (no dependent load through cp on the critical path)
    const double *Xt;
    const uint8_t *z;
    const double *enorm[2], *uprior[2];
    double *rhs, *eta;
    const double *part_quad;
    double *part_beta;
    double tau_rate, tau_shape;
    unsigned *bar;        // [C][BAR_STRIDE]
    unsigned *claim;      // [C][16] one XCD per chain: the next free workgroup slot of the chain
    unsigned long long *clock;  // Ctx::iter_clock
    unsigned *sync;       // Ctx::sync (stream hand-overs by device counters) or null
    double *part;         // [C][3][nb_n][4] per-slice records of the running solve (two by step parity; XL: three in rotation)
    // k_tiles (occ_tiles.hpp): exchange buffers of p (three in rotation, [C][tiles_npad]), tagged records [C][nb_n][4],
    // tiles per workgroup, workgroups per chain, workgroups per XCD band
    double2 *tex[3];
    double *trec;
    double *tband;               // [C][3][8][4] band records of a step (three buffers in rotation)
    unsigned long long *tflag;   // [C][2][G] "p of step k is out" per workgroup: plain-stored / write-through copy
    int tiles_T, tiles_G, tiles_B, tiles_npad;
    int nbg;              // workgroups per chain
    int chain_base;       // one XCD per chain: first chain of this launch (more than eight chains run as several launches of eight)
    int C, p, q;
};

constexpr int ITER_PART_DOUBLES = 3 * 4;  // per slice and chain (IterArgs::part)

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
// 16-byte write-through store / L1-bypassing load (aux 16 = sc1); XL: a plain store, the line stays in the XCD's L2
template <int XL>
__device__ __forceinline__ void store_x(__amdgpu_buffer_rsrc_t r, int byte_off, double2 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(pack_d2(v), r, byte_off, 0, XL ? 0 : 16);
}
__device__ __forceinline__ double2 load_sc1(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return unpack_d2(__builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}

// XL step exchange: a slice's record of a step IS its arrival flag.  Every site wave (one 64-site slice) adds up its four
// sums of the step and, after its `s_waitcnt vmcnt(0)` (its g is in the XCD's L2), stores ONE record {S0, S1 | S2, S3} of
// two 16-byte halves.  A half that still holds the CANARY (a NaN no sum can produce) has not arrived.  Records rotate
// through three buffers: in step k a wave puts canaries into its record of step k + 1 before that wait, so whoever sees
// a slice's record of step k also sees its g and its canaries of step k + 1 -- no ordering assumption between different
// cache lines.  The buffer of step k + 1 was last read in step k - 2: a workgroup is in step k only after every slice
// has stored its record of step k - 1, which each did after its workgroup had read all of step k - 2.  The poller (one
// wave per workgroup) reads the records of all slices of the chain, lane l those of slices l, l + 64, ..., until none
// shows the canary, adding them up as it goes (poll_slice_records): ONE round trip where "poll the flags, then load the
// per-slice sums" took two, and no reduction inside the workgroup on the way (round 2 had one record per workgroup:
// wave sums through LDS, a workgroup barrier and a tree that tied the summation order to 512-site workgroups).
// 16-byte stores and loads are not torn (MI355X_MICROARCH.md, observed on gfx950); each half is checked on its own.
__device__ __forceinline__ double2 rec_canary()
{
    const double c = __longlong_as_double(0x7ff8dead0ccbeef0LL);
    return make_double2(c, c);
}
__device__ __forceinline__ bool rec_pending(double2 half) { return __double_as_longlong(half.x) == 0x7ff8dead0ccbeef0LL; }

// Barrier among the workgroups of one chain, in two halves so that work which needs nothing from the other
// workgroups can run between them.  ARRIVE: every storing wave drains its write-through stores, then one lane
// adds to the chain's counter.  WAIT: that lane polls until the counter has reached `target` arrivals
// (compared modulo 2^32); `fail_flag` (LDS) is set when the poll gave up (a workgroup of the chain is not
// running).
//
// XL form: ARRIVE stores the workgroup's flag = (barrier number << 4) | XCC_ID (plain store); WAIT: the first wave
// loads all nbg (<= 64) flags of the chain, one per lane (sc1), until every flag has reached the barrier number,
// then checks that every writer sits on the reader's XCD.
// A barrier that gives up says so in the chain's error word at once (agent scope): the other workgroups of the chain
// look at it every 1024 polls, workgroups that start later see it at kernel entry, and every later kernel of the
// enqueued batch skips the chain -- a failed launch drains in milliseconds and the host re-runs the call on the
// launch-per-step path (occ_gibbs.hip, run_with_fallback).
__device__ __forceinline__ int chain_err(const ChainScalars &sc)
{
    return __hip_atomic_load(&sc.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_fail(ChainScalars &sc)
{
    __hip_atomic_store(&sc.err, -2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // OCC_E_HIP
}
#define BAR_STAMP(pt)
#define OCC_CHAIN_ARRIVE()                                                                                 \
    do {                                                                                                   \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
        BAR_STAMP(3)                                                                                       \
        __syncthreads();                                                                                   \
        if (threadIdx.x == 0) {                                                                            \
            if (XL) __builtin_amdgcn_raw_buffer_store_b32(((bar_base + nbar) << 4) | my_xcc, fbuf, wg * 4, 0, 0); \
            else __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);              \
        }                                                                                                  \
        BAR_STAMP(4)                                                                                       \
    } while (0)
#define OCC_CHAIN_WAIT(fail_flag)                                                                          \
    do {                                                                                                   \
        if (XL) {                                                                                          \
            if (threadIdx.x < 64) {                                                                        \
                const unsigned want_ = (bar_base + nbar) << 4;                                             \
                int fail_ = 0;                                                                             \
                unsigned spins_ = 0, flag_;                                                                \
                for (;;) {                                                                                 \
                    flag_ = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(fbuf, (int)threadIdx.x * 4, 0, 16); /* lanes >= nbg: out of range, 0 */ \
                    if ((int)threadIdx.x >= ia.nbg) flag_ = want_ | my_xcc;                                \
                    if (__all((int)((flag_ & ~15u) - want_) >= 0)) break;                                  \
                    __builtin_amdgcn_s_sleep(1);                                                           \
                    if (++spins_ > spin_limit) { fail_ = 1; break; }                                       \
                    if ((spins_ & 1023u) == 0u && chain_err(sc) != 0) { fail_ = 1; break; } /* another workgroup gave up */ \
                }                                                                                          \
                if (!fail_ && __any((flag_ & 15u) != my_xcc)) fail_ = 1; /* a workgroup of the chain on another XCD */ \
                if (threadIdx.x == 0) {                                                                    \
                    fail_flag = fail_;                                                                     \
                    if (fail_) chain_fail(sc);                                                             \
                }                                                                                          \
            }                                                                                              \
        } else if (threadIdx.x == 0) {                                                                     \
            const unsigned target_ = bar_base + nbar * (unsigned)ia.nbg;                                   \
            int fail_ = 0;                                                                                 \
            unsigned spins_ = 0;                                                                           \
            while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target_) < 0) { \
                __builtin_amdgcn_s_sleep(1);                                                               \
                if (++spins_ > spin_limit) { fail_ = 1; break; }                                           \
                if ((spins_ & 1023u) == 0u && chain_err(sc) != 0) { fail_ = 1; break; }                    \
            }                                                                                              \
            fail_flag = fail_;                                                                             \
            if (fail_) chain_fail(sc);                                                                     \
        }                                                                                                  \
        BAR_STAMP(5)                                                                                       \
        __syncthreads();                                                                                   \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); /* compiler only: no load moves above the poll */ \
    } while (0)
#define OCC_CHAIN_BARRIER(fail_flag)                                                                       \
    do {                                                                                                   \
        OCC_CHAIN_ARRIVE();                                                                                \
        OCC_CHAIN_WAIT(fail_flag);                                                                         \
    } while (0)

// Partial sums of beta's system for one 64-site slice (k_beta_partial's, per wave).
template <int P>
__device__ __forceinline__ void beta_partials_slice(const IterArgs &ia, int chain, int i, bool act, int slice, bool slice_act,
                                                    double om, double eta, double zval)
{
    const int n = ia.a.n, nb = ia.a.nb_n, lane = threadIdx.x & 63;
    double acc[nacc(P)];
#pragma unroll
    for (int t = 0; t < nacc(P); ++t) acc[t] = 0.0;
    if (act) {
        const double tt = beta_rhs_term(om, eta, zval);
        double x[P];
#pragma unroll
        for (int aa = 0; aa < P; ++aa) x[aa] = ia.Xt[(size_t)aa * n + i];
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
    if (slice_act) {
        double *out = ia.part_beta + (size_t)chain * nacc(P) * nb;
#pragma unroll
        for (int t = 0; t < nacc(P); ++t) {
            const double r = wave_sum(acc[t]);
            if (lane == 0) out[t * nb + slice] = r;
        }
    }
}

#define OCC_SWITCH_DIM(d, CALL)                                                                            \
    switch (d) {                                                                                           \
        case 1: { constexpr int D = 1; CALL; } break;                                                      \
        case 2: { constexpr int D = 2; CALL; } break;                                                      \
        case 3: { constexpr int D = 3; CALL; } break;                                                      \
        case 4: { constexpr int D = 4; CALL; } break;                                                      \
        case 5: { constexpr int D = 5; CALL; } break;                                                      \
        case 6: { constexpr int D = 6; CALL; } break;                                                      \
        case 7: { constexpr int D = 7; CALL; } break;                                                      \
        default: { constexpr int D = 8; CALL; } break;                                                     \
    }

// The poll of the one-XCD forms' step exchange: lane l reads the records of slices l, l + 64, ... of the chain (two
// 16-byte halves each, L1 bypassed) until none shows the canary, and adds them up in the canonical order
// (sum_slices_canonical: the same rounds, the same additions, one wave sum at the end).  Returns false when it gave up.
struct RecordSet {
    double2 lo[4], hi[4];
};
template <bool PIPELINED>  // (64 more registers: for a wave that has them -- the scalar wave)
__device__ __forceinline__ bool poll_slice_records(__amdgpu_buffer_rsrc_t buf, int nslices, int lane, unsigned spin_limit, const ChainScalars &sc,
                                                   double (&tot)[4])
{
    unsigned spins = 0;
    if (PIPELINED && nslices <= 256) {
        // at most four records per lane: TWO rounds of loads in flight, a new one issued while the previous one returns --
        // the last record is noticed about half an L2 round trip earlier than with "load, wait, look, load again"
        auto issue = [&](RecordSet &s) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // (records past the last slice fall outside the descriptor: zeros)
                s.lo[r] = load_sc1(buf, (64 * r + lane) * 32);
                s.hi[r] = load_sc1(buf, (64 * r + lane) * 32 + 16);
            }
        };
        auto complete = [&](const RecordSet &s) {
            bool pend = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) pend = pend || rec_pending(s.lo[r]) || rec_pending(s.hi[r]);
            return !__any(pend);
        };
        auto add_up = [&](const RecordSet &s) {  // the canonical order (sum_slices_canonical)
#pragma unroll
            for (int q = 0; q < 4; ++q) tot[q] = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) { tot[0] += s.lo[r].x; tot[1] += s.lo[r].y; tot[2] += s.hi[r].x; tot[3] += s.hi[r].y; }
        };
        RecordSet A, B;
        issue(A);
        for (;;) {
            issue(B);
            if (complete(A)) { add_up(A); break; }
            issue(A);
            if (complete(B)) { add_up(B); break; }
            if (++spins > spin_limit) return false;
            if ((spins & 1023u) == 0u && chain_err(sc) != 0) return false;  // another workgroup gave up
        }
        wave_sum4(tot);
        return true;
    }
    for (;;) {
        bool pend = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) tot[q] = 0.0;
        for (int base = 0; base < nslices; base += 256) {
            double2 lo[4], hi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // (records past the last slice fall outside the descriptor: zeros)
                lo[r] = load_sc1(buf, (base + 64 * r + lane) * 32);
                hi[r] = load_sc1(buf, (base + 64 * r + lane) * 32 + 16);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pend = pend || rec_pending(lo[r]) || rec_pending(hi[r]);
                tot[0] += lo[r].x; tot[1] += lo[r].y; tot[2] += hi[r].x; tot[3] += hi[r].y;
            }
        }
        if (!__any(pend)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > spin_limit) return false;
        if ((spins & 1023u) == 0u && chain_err(sc) != 0) return false;  // another workgroup gave up
    }
    wave_sum4(tot);
    return true;
}

// NW = width of the register-resident neighbour window: 8 (<= 256 VGPRs, two workgroups per CU) or 16 (one per CU);
// XL = 1: one XCD per chain (see the head of this file)
//
// W512 (with XL, where a chain's waves outnumber the SIMDs its XCD gives the main stream) runs 512 threads per
// workgroup, two waves per SIMD, as ONE SCALAR WAVE AND SEVEN SITE WAVES (448 sites).  A MINRES step is a chain of
// dependent events -- sums of step k - 1 -> scalar recurrence (~100 dependent f64 instructions) -> coefficients ->
// vectors -> sums -- and with the recurrence on a wave that also owns sites (round 2: wave 0 of eight site waves) every
// wave of the chain waits for it: 45 % of a step.  Here the scalar wave owns no sites.  It polls the records and forms
// what the site waves need next and nothing else -- the three coefficients of p_{k-1} (minres_post_b) and the five of the
// rotation of iteration k - 2 (minres_rotation) -- one hand-over in LDS; while the site waves run the whole step with them
// (h = A g_{k-1} from the gathers, p, g, w, x, the four sums, the record) it evaluates the stopping test of iteration k - 3
// (minres_post_a: the verdict travels with the step's last barrier), updates the slot (minres_post_c), prepares the next
// step (minres_pre) and is polling again before the first record of the step arrives.  Per step: two workgroup barriers
// (hand-over, poll done), no reduction through LDS (every site wave publishes its own record).
//
// flags: bit 0 = hand over to / from the side stream through the device counters; bit 1 = RESIDENCY PROBE: the launch
// does nothing but one barrier among the workgroups of every chain, with a short time limit -- the same kernel, grid,
// registers and LDS as the real launch, so it passes exactly when all workgroups of a chain are resident together
// (and, XL, sit on one XCD: the flags carry their writers' XCC_ID).  The host runs it at creation (create_impl).
template <int NW, int XL, int W512>
__global__ void __launch_bounds__(W512 ? ITER_WG_XL : ITER_WG, (NW == 8 && !W512) ? 2 : 1) k_iter(const IterArgs ia, int e, int flags)
{
    const bool probe = (flags & 2) != 0;
    const int sync_on = probe ? 0 : (flags & 1);
    const unsigned spin_limit = probe ? ITER_PROBE_SPIN_LIMIT : ITER_SPIN_LIMIT;
    constexpr int WGT = W512 ? ITER_WG_XL : ITER_WG;
    constexpr bool SW = W512 == 1;                       // scalar wave + seven site waves (W512 = 2: eight site waves, the first one leads)
    constexpr int SITES_WG = SW ? ITER_SITES_SW : WGT;   // sites per workgroup
    // Wave 0 does the uniform work for all in the 8-wide-window forms.  Redundant scalar work in every wave costs each of them
    // the MINRES state in registers (72 spilled registers in the 256-thread one-XCD form: 60x60 x 8 chains 76.9 k -> 94.4 k
    // chain-it/s with wave 0 alone) and, any placement, four readers of all per-slice sums per workgroup where one will do.
    constexpr bool SHARE = NW == 8;  // (the 16-wide window has one workgroup per CU and registers to spare: every wave for itself, as in round 1)
    __shared__ int s_flag, s_noise_ok;
    __shared__ double s_bcast[12];  // SHARE: tau, then the coefficients of the coming step, from wave 0 to the workgroup
    __shared__ double s_rot[8];     // SW: the rotation's coefficients (second hand-over of a step)
    __shared__ Slot s_slot;         // SHARE without a scalar wave: the MINRES scalar state (wave 0)
    const KryArgs &a = ia.a;
    // grid = (nbg, C), the chain is blockIdx.y (a scalar register: the buffer descriptors below must be provably
    // wave-uniform, or every buffer access becomes a serialising waterfall loop).
    // XL: the chain is THE XCD THE WORKGROUP RUNS ON (HW_REG_XCC_ID, also a scalar register) and its place among the chain's
    // workgroups is claimed from a counter -- nothing is assumed about how the dispatcher deals a grid to the XCDs (round 2
    // took chain = blockIdx.x of an (8, nbg) grid: true while every XCD can take its share of the grid at once, false
    // as soon as the stream's CU mask gives the XCDs different numbers of CUs -- the form with a scalar wave wants 24 CUs
    // on the XCDs that host a chain and leaves the others to the side stream).  The grid is 8 x (nbg + 1) workgroups:
    // those on an XCD without a chain and those that find the chain complete return at once.
    __shared__ int s_claim;
    int chain, wg = 0;
    const unsigned my_xcc = XL ? (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) : 0u;  // HW_REG_XCC_ID
    const bool synced = sync_on && ia.sync != nullptr;
    // k_z_ob of the previous sequence is complete (stream order): the side stream may start this sequence.  Said by the
    // first workgroup of the grid before anything can return or wait.
    int ticket = 0;
    if (XL) {
        if ((int)my_xcc >= min(ia.C - ia.chain_base, XL_SLOTS)) return;
        chain = ia.chain_base + (int)my_xcc;
    } else {
        chain = (int)blockIdx.y;
        wg = (int)blockIdx.x;
    }
    // (a scalar register by force, HERE: left to itself the compiler may sink the chain number's 64-bit extension into the
    // one-lane branch below and merge it back as a VECTOR register -- every buffer descriptor derived from it is then
    // lane-dependent in its eyes and each of the kernel's buffer accesses becomes a waterfall loop: k_iter 40 -> 54 us when a
    // change elsewhere in the kernel tipped that decision, as in k_tiles before it)
    size_t chain64 = (size_t)chain;
    asm volatile("" : "+s"(chain64));
    if (XL) {
        if (threadIdx.x == 0) ticket = (int)__hip_atomic_fetch_add(ia.claim + chain64 * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_setprio(3);  // critical path: issue ahead of any co-resident Polya-Gamma waves
    // what needs neither the slot nor the noise is on its way while the claim's atomic returns: the chain's control words,
    // and one lane's wait for the noise of this iteration (from the side stream's previous sequence; normally it has been
    // there for a whole iteration -- the workgroup looks at the result before phase A's loads of the noise)
    ChainScalars &sc = a.scs[chain64];
    const Ctl ctl = sc.ctl[e];
    const uint32_t it_stop = sc.it_stop;
    const int err0 = sc.err;
    // ... and, on the wave that will draw tau, the first 256 partial sums of eta'Q eta (k_z_ob's, cold): their latency runs
    // beside the claim's instead of in front of phase A's barrier
    double quad_pre[4] = {0.0, 0.0, 0.0, 0.0};
    if (!SHARE || threadIdx.x < 64) {  // (= `lead` below)
        const double *pq0 = ia.part_quad + chain64 * a.nb_n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int bb = (int)(threadIdx.x & 63) + 64 * r;
            const double t = pq0[min(bb, a.nb_n - 1)];
            quad_pre[r] = (bb < a.nb_n) ? t : 0.0;
        }
    }
    if (XL) {
        if (threadIdx.x == 0) s_claim = ticket;
        __syncthreads();
        wg = __builtin_amdgcn_readfirstlane(s_claim);
        if (wg >= ia.nbg) return;
    }
    if (synced && threadIdx.x == 0) {
        const unsigned j = ia.sync[SYNC_MAIN_SEQ + e];      // this sequence's number (the previous sequence left it)
        s_noise_ok = sync_wait(ia.sync, SYNC_NOISE, j) ? 1 : 0;
    }
    const bool writer = (wg == 0 && threadIdx.x == 0);
    // k_z_ob of the previous sequence is complete (stream order): the side stream may start this sequence.  Said before
    // anything can return or wait -- but by the workgroup that claimed the first slot of chain 0, not at kernel entry: the
    // side stream answers with grids of thousands of workgroups, and released a microsecond earlier (by the first
    // workgroup of the grid, before its claim) they got between the dispatcher and k_iter's own workgroups: k_iter 44.7 ->
    // 57 us.  (Later -- after the first barrier among the chain's workgroups -- changes nothing.)
    if (synced && writer && chain == 0) sync_set(ia.sync + SYNC_MAIN, ia.sync[SYNC_MAIN_SEQ + e]);
    // (a chain whose error word is set idles like a finished one: see chain_fail)
    if (!probe && (ctl.koff || ctl.it >= it_stop || err0 != 0)) {  // uniform over the chain's workgroups
        if (writer) sc.mid[e] = ctl;
        return;
    }
    const unsigned long long clk0 = writer ? (unsigned long long)wall_clock64() : 0ull;
    const uint32_t it = ctl.it;
    // the scalar wave (SW): a wave-uniform fact the compiler can see (a scalar register), so that the two roles of the
    // solve below are two loops, each with its own registers, not one loop under an execution mask
    const bool scalar_wave = SW && __builtin_amdgcn_readfirstlane((int)threadIdx.x) < 64;
    const int n = a.n, i = scalar_wave ? n : wg * SITES_WG + (int)threadIdx.x - (SW ? 64 : 0);
    const bool lead = !SHARE || threadIdx.x < 64;  // the wave that does the uniform work
    const bool act = i < n;
    const int lane = threadIdx.x & 63, slice = scalar_wave ? a.nb_n : (i >> 6);
    const bool slice_act = slice < a.nb_n;  // a slice with at least one site owns a partial sum
    const size_t co = chain64 * n;
    const double2 zero2 = make_double2(0.0, 0.0);
    unsigned *cnt = ia.bar + chain64 * BAR_STRIDE;
    const __amdgpu_buffer_rsrc_t fbuf = __builtin_amdgcn_make_buffer_rsrc((void *)cnt, 0, ia.nbg * 4, 0x00020000);  // XL: the chain's flags
    (void)fbuf; (void)my_xcc;
    const unsigned bar_base = sc.bar_base;
    unsigned nbar = 0;  // barriers passed in this launch
    const __amdgpu_buffer_rsrc_t gbuf[2] = {
        __builtin_amdgcn_make_buffer_rsrc((void *)(a.Gv[0] + co), 0, n * 16, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc((void *)(a.Gv[1] + co), 0, n * 16, 0x00020000)};
    // per-slice records of the running solve {S0, S1 | S2, S3}: two buffers by step parity (any placement), three in
    // rotation when the records double as arrival flags (XL, see "step exchange" above)
    double *part_base = ia.part + chain64 * ITER_PART_DOUBLES * a.nb_n;
    const __amdgpu_buffer_rsrc_t pbuf[3] = {
        __builtin_amdgcn_make_buffer_rsrc((void *)part_base, 0, a.nb_n * 32, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc((void *)(part_base + (size_t)a.nb_n * 4), 0, a.nb_n * 32, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc((void *)(part_base + (size_t)a.nb_n * 8), 0, a.nb_n * 32, 0x00020000)};
    if (probe) {  // residency / placement probe: one barrier, nothing else
        ++nbar;
        OCC_CHAIN_BARRIER(s_flag);
        if (writer) sc.bar_base = bar_base + nbar * (XL ? 1u : (unsigned)ia.nbg);
        return;
    }

    // ---- phase A: tau, right-hand side, p_0 = b - A x0 (all inputs come from earlier launches: plain loads)
    PHASE_STAMP(0, 0)
    // Every memory operation of the solve role is issued unconditionally (no per-slot branch: a branch per
    // gather makes the compiler wait for each load before it issues the next).  Unused neighbour slots point at
    // the site itself with coefficient 0; lanes past the last site (and the scalar wave) read row n-1 and their buffer
    // accesses fall outside the descriptors' range (loads return 0, stores are dropped).
    // What does not depend on tau or on the noise is loaded FIRST: the site waves have it in flight while the lead wave
    // draws tau.
    int off[NW];      // byte offset of neighbour kk in a [n] double2 array
    double av[NW];    // Q_ij, then tau * Q_ij
    unsigned hasmask = 0u;
    double2 ng[NW];
    const int ic = act ? i : n - 1;             // clamped row for plain loads
    const int myoff = act ? i * 16 : n * 16;    // byte offset of this site in the exchange buffers
    int width, base;
    {
        const int sl = ic >> 6;
        if (a.ell_w > 0) { width = a.ell_w; base = sl * a.ell_w * 64; }
        else { base = a.sell_ptr[sl]; width = (a.sell_ptr[sl + 1] - base) >> 6; }
    }
    const size_t ci = co + ic;
    const double2 *X0 = a.Xv + co;
    double om = 0.0, zval = 0.0, xb = 0.0, qd = 0.0;
    double2 x = zero2;
    double2 xn[NW];
#pragma unroll
    for (int kk = 0; kk < NW; ++kk) { off[kk] = myoff; av[kk] = 0.0; xn[kk] = zero2; }
    // (the scalar wave owns no site: nothing stands between it and tau.)  The site waves issue their first level of loads,
    // pass the noise hand-over's barrier at once -- the scalar wave is waiting there to start on tau -- and only then
    // take the neighbour gathers, whose addresses are loaded values
    int jraw[NW];
    double vraw[NW];
#pragma unroll
    for (int kk = 0; kk < NW; ++kk) { jraw[kk] = 0; vraw[kk] = 0.0; }
    // DIA (KryArgs::dia_*: the off-diagonals lie on at most eight diagonals with one value each -- any unweighted lattice;
    // the 8-wide window only): a neighbour's index is the site's plus a constant, so the gathers of x_0 go out WITH the first
    // level of loads (row clamped; the mask byte decides the coefficient afterwards) instead of behind the column indices --
    // one cold memory latency less in front of p_0.  Same neighbours in the same order as the stored slots (ascending
    // columns), absent ones with coefficient 0: the same bits (k_minres and k_tiles take the same form).
    const bool dia = NW == 8 && a.dia_n > 0;
    unsigned dmask_raw = 0u;
    auto neighbours = [&]() {
#pragma unroll
        for (int kk = 0; kk < NW; ++kk) {
            const bool has = dia ? (act && kk < a.dia_n && ((dmask_raw >> kk) & 1u)) : (act && kk < width);
            const int j = has ? jraw[kk] : ic;
            off[kk] = has ? j * 16 : myoff;
            av[kk] = dia ? (has ? a.dia_val[kk & 7] : 0.0) : vraw[kk];
            hasmask |= has ? (1u << kk) : 0u;
            if (!dia) xn[kk] = X0[j];
        }
    };
    if (!scalar_wave) {
        om = a.omega_b[it & 1][ci];
        zval = (double)ia.z[ci];
        xb = xdot(ia.Xt, n, ic, sc.beta, ia.p);
        x = X0[ic];
        qd = a.qdiag[ic];
        if (dia) {
            dmask_raw = a.dia_mask[ic];
#pragma unroll
            for (int kk = 0; kk < NW; ++kk) {
                jraw[kk] = min(max(ic + (kk < a.dia_n ? a.dia_off[kk & 7] : 0), 0), n - 1);
                xn[kk] = X0[jraw[kk]];
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < NW; ++kk) {
                const int slot = base + ((kk < width) ? kk * 64 : 0) + (ic & 63);  // always inside the (padded) slot arrays
                jraw[kk] = a.sell_col[slot];
                vraw[kk] = a.sell_val[slot];
            }
        }
        if (!SW) neighbours();  // (without a scalar wave nobody waits at the barrier below: no reason to hold the indices across it)
    }
    if (synced) {  // thread 0's wait for the side stream's noise kernel (started at kernel entry) is over: s_noise_ok is set
        __syncthreads();
        if (!s_noise_ok && writer) sc.err = -2;
    }
    double en = 0.0, up = 0.0;
    if (scalar_wave) {
    } else {
        if (synced) {
            en = load_agent(&ia.enorm[it & 1][ci]);
            up = load_agent(&ia.uprior[it & 1][ci]);
        } else {
            en = ia.enorm[it & 1][ci];
            up = ia.uprior[it & 1][ci];
        }
        if (SW) neighbours();
    }
    double tau = 0.0;
    if (lead) {
        double q = 0.0;
        const double *pq = ia.part_quad + chain64 * a.nb_n;
        // tau's standard gamma variate was drawn one iteration ahead by k_noise (side stream), like the rest of the noise:
        // 4 000 cycles of dependent f64 arithmetic that used to sit here, in front of every load of the phase.  (Loaded first:
        // its latency runs beside that of the partial sums, not behind their reduction.)
        const double gvar = synced ? load_agent(&sc.tau_gamma[it & 1]) : sc.tau_gamma[it & 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) q += quad_pre[r];  // (the first round: loaded at kernel entry; the same order of additions)
        for (int b0 = lane + 256; b0 < a.nb_n; b0 += 256) {
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int bb = b0 + 64 * r;
                const double t = pq[min(bb, a.nb_n - 1)];
                v[r] = (bb < a.nb_n) ? t : 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) q += v[r];
        }
        q = wave_sum(q);
        const double rate = 0.5 * q + ia.tau_rate;
        tau = (1.0 / rate) * gvar;
        if (writer) sc.tau = tau;
        if (SHARE && threadIdx.x == 0) s_bcast[0] = tau;
    }
    if (SHARE) {
        __syncthreads();
        tau = s_bcast[0];
    }
    PHASE_STAMP(0, 1)
    const double y = eta_rhs_site(om, xb, zval, en, up, sqrt(tau));
    const double d = tau * qd + om;
#pragma unroll
    for (int kk = 0; kk < NW; ++kk) av[kk] = ((hasmask >> kk) & 1u) ? tau * av[kk] : 0.0;
    double ax = d * x.x, az = d * x.y;
#pragma unroll
    for (int kk = 0; kk < NW; ++kk) {
        ax = fma(av[kk], xn[kk].x, ax);
        az = fma(av[kk], xn[kk].y, az);
    }
    const double2 p0 = make_double2(y - ax, 1.0 - az);
    if (act) ia.rhs[ci] = y;
    store_x<XL>(gbuf[0], myoff, p0);
    // XL: every site wave puts the canary into its slice's record of step 1 (the records step 1 polls)
    if (XL && !scalar_wave && lane < 2) store_x<1>(pbuf[1], slice * 32 + lane * 16, rec_canary());
    PHASE_STAMP(0, 2)
    ++nbar;
    OCC_CHAIN_BARRIER(s_flag);
    PHASE_STAMP(0, 3)
    bool failed = s_flag != 0;
    if (!SW) {
#pragma unroll
        for (int kk = 0; kk < NW; ++kk) ng[kk] = load_sc1(gbuf[0], off[kk]);  // p_0 at the neighbours: plays p_{k-1} at step 1
    }

    // ---- phase B: MINRES.  The coefficients of step k come from the sums of step k - 1 (minres_post).
    Slot s_reg = {};
    int k = 1;
    double2 g = p0, gm2 = zero2, pm1 = zero2, pm2 = zero2, wm1 = zero2, wm2 = zero2;
    if constexpr (SW) {
        // ======== scalar wave + seven site waves.  Workgroup barriers of step k, in order:
        //   B1  the coefficients of p_{k-1} (s_bcast) and of the rotation of iteration k - 2 (s_rot) are there
        //       (scalar wave: after minres_post_b and minres_rotation, formed while the site waves' gathers of g_{k-1} travel)
        //   B0  the verdict on step k (s_flag) -- 0: every record has arrived, 1: the poll gave up, 2: the solve ended before
        //       the step (scipy's test of iteration k - 3, evaluated beside the vectors: what the step did is dropped)
        // Both roles leave the loop at the same place, after B0.
        // At step 1 the coefficients are ca = 1, cb = cc = 0: kry_form_p then returns its first argument -- p_0, and
        // h = A p_0 from the p_0 gathered at the neighbours -- exactly: the histories are zeros.
        if (scalar_wave) {
            Slot &s = s_reg;
            KryMid mid;
            KryPre pre = minres_pre(s);
            KryStep st = minres_post_b(s, pre, 1, 0.0, 0.0, 0.0, mid);
            double xn2 = 0.0;
            if (threadIdx.x == 0) { s_bcast[0] = 1.0; s_bcast[1] = 0.0; s_bcast[2] = 0.0; s_rot[5] = 0.0; }
            __syncthreads();  // B1 of step 1
            for (; !failed; ++k) {
                SOLVE_STAMP(0)
                // the site waves are on step k; here: (a) the stopping test of iteration k - 3 -- they hear of it at the step's
                // last barrier (what they did in the step is then dropped).  (Uniform by value; said so to the compiler, or k --
                // and with it every buffer descriptor chosen by k -- counts as divergent and each buffer access becomes a
                // waterfall loop.)
                const bool stop = __builtin_amdgcn_readfirstlane((int)(st.stop || minres_post_a(s, pre, k, xn2, a.maxiter))) != 0;
                SOLVE_STAMP(1)
                bool ok = true;
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                if (!stop) {
                    minres_post_c(s, pre, k, st, mid);  // (c) in full: the slot after step k
                    SOLVE_STAMP(12)
                    pre = minres_pre(s);  // the slot-only half of step k + 1, while the site waves finish step k
                    SOLVE_STAMP(13)
                    ok = __builtin_amdgcn_readfirstlane((int)poll_slice_records<true>(pbuf[k % 3], a.nb_n, lane, spin_limit, sc, acc)) != 0;
                }
                SOLVE_STAMP(3)
                if (threadIdx.x == 0) {
                    s_flag = ok ? (stop ? 2 : 0) : 1;
                    if (!ok) chain_fail(sc);
                }
                __syncthreads();  // B0
                SOLVE_STAMP(4)
                if (!ok) { failed = true; break; }
                if (stop) break;
                // (b): the coefficients of p_k, and the part of (c) of step k + 1 the site waves need after them -- the rotation's
                // coefficients -- while their gathers of g_k are on the way
                st = minres_post_b(s, pre, k + 1, acc[0], acc[1], acc[2], mid);
                xn2 = acc[3];
                if (!st.stop) minres_rotation(s, pre, k + 1, st, mid);
                SOLVE_STAMP(5)
                if (threadIdx.x == 0) {
                    s_bcast[0] = st.ca; s_bcast[1] = st.cb; s_bcast[2] = st.cc;
                    s_rot[0] = st.sj; s_rot[1] = st.oldeps; s_rot[2] = st.delta; s_rot[3] = st.denom; s_rot[4] = st.phi;
                    s_rot[5] = st.rotate ? 1.0 : 0.0;
                }
                __syncthreads();  // B1 of step k + 1
            }
            if (writer) {
                if (failed) { s.done = 1; s.istop = 6; s.itn = k; }
                slot_store(&a.slots[chain64 * NSLOT], s);
                sc.minres_itn_last = s.itn;
                sc.krylov_total += (unsigned long long)s.itn;
                sc.krylov_sq_total += (unsigned long long)s.itn * (unsigned long long)s.itn;
                sc.solves += 1ull;
                if (s.istop == 6 && !failed) sc.err = -3;  // OCC_E_MINRES (logit.py:91-92)
            }
        } else {
            KryStep st = {};
            // One step.  The histories are passed in the roles they play in THIS step -- (h1, h2) = p_{k-2}, p_{k-3} at the
            // neighbours, (q1, q2) the same at the site, (u1, u2) = w_{k-3}, w_{k-4} -- and the step leaves the new vector in
            // the older one's registers; the loop below calls it with the roles swapped every other step.  (With one set of
            // names and "older = newer; newer = new" at the end of the body the compiler moved 76 registers per step, a
            // third of what a site wave issued.)  Returns false when the solve is over (stop or failure).
            auto site_step = [&](double2 &g1, double2 &g2, double2 &q1, double2 &q2, double2 &u1, double2 &u2) -> bool {
                // g_{k-1} at the neighbours (step 1: p_0), complete since the poll of step k - 1 (the chain barrier of phase A):
                // the only place that loads it, so the loads land in the registers the step reads them from
                double2 ng[NW];
#pragma unroll
                for (int kk = 0; kk < NW; ++kk) ng[kk] = load_sc1(gbuf[(k - 1) & 1], off[kk]);
                SITE_STAMP(11)
                __syncthreads();  // B1
                SITE_STAMP(6)
                st.ca = s_bcast[0]; st.cb = s_bcast[1]; st.cc = s_bcast[2];
                st.sj = s_rot[0]; st.oldeps = s_rot[1]; st.delta = s_rot[2]; st.denom = s_rot[3]; st.phi = s_rot[4];
                const bool rotate = __builtin_amdgcn_readfirstlane((int)(s_rot[5] != 0.0)) != 0;
                double part[4] = {0.0, 0.0, 0.0, 0.0};
                const double2 h = kry_apply<NW>(d, av, g1, ng);  // A g_{k-1}
                const double2 p = kry_form_p(st, g1, q2, q1);    // p_{k-1}
                const double2 gn = kry_form_p(st, h, g2, g1);    // g_k = A p_{k-1}, by the same three-term recurrence
                g2 = gn;
                store_x<1>(gbuf[k & 1], myoff, gn);
                if (lane < 2) store_x<1>(pbuf[(k + 1) % 3], slice * 32 + lane * 16, rec_canary());  // this slice's record of step k + 1
                part[0] = dot2(p, p);
                part[1] = fma(p.y, gn.y, p.x * gn.x);
                if (k >= 2) part[2] = dot2(p, q1);
                SITE_STAMP(7)
                double2 xk = x;  // x_{k-2}: takes x's place at the end of the step unless the solve turns out to have ended before it
                if (rotate) {  // w_{k-2}, x_{k-2}  (before the first rotation both w's are zeros: the roles may swap)
                    const double2 w = kry_form_w(st, q2, u2, u1);
                    xk.x = fma(st.phi, w.x, x.x);
                    xk.y = fma(st.phi, w.y, x.y);
                    u2 = w;
                    part[3] = dot2(xk, xk);
                }
                q2 = p;
                if (!act) { part[0] = 0.0; part[1] = 0.0; part[2] = 0.0; part[3] = 0.0; }
                wave_sum4(part);  // (a slice past the last site sums zeros)
                SITE_STAMP(9)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's g and its canaries of step k + 1 are in the XCD's L2
                if (lane < 2) store_x<1>(pbuf[k % 3], slice * 32 + lane * 16, lane == 0 ? make_double2(part[0], part[1]) : make_double2(part[2], part[3]));
                SITE_STAMP(10)
                __syncthreads();  // B0: every record of the step has arrived (0), or the poll gave up (1), or the solve ended before this step (2)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler only: no load moves above the poll
                const int how = __builtin_amdgcn_readfirstlane(s_flag);  // (uniform: see the scalar wave)
                if (how == 1) failed = true;
                if (how != 0) return false;
                x = xk;
                ++k;
                return true;
            };
            if (failed) __syncthreads();  // (phase A gave up: the scalar wave's B1 of step 1 still stands)
            while (!failed) {
                if (!site_step(g, gm2, pm1, pm2, wm1, wm2)) break;
                if (!site_step(gm2, g, pm2, pm1, wm2, wm1)) break;
            }
        }
    } else {
    // W512: the scalar state lives in LDS (only wave 0 touches it; in registers it would cost every wave 30 VGPRs)
    Slot &s = SHARE ? s_slot : s_reg;
    if (SHARE && threadIdx.x == 0) {
#define X(f) s_slot.f = 0;
        OCC_SLOT_FIELDS(X)
#undef X
    }
    double S0 = 0.0, S1 = 0.0, S2 = 0.0, xn2 = 0.0;
    KryPre pre = {};
    KryStep st = {};
#define NEXT_STAMP(pt)
#define OCC_NEXT_STEP(kn)                                                                                  \
    do {                                                                                                   \
        if (lead) {                                                                                        \
            if (SHARE) { /* the LDS-resident state through registers: every load out before the first use */ \
                Slot t_ = slot_load(&s);                                                                   \
                st = minres_post(t_, pre, (kn), S0, S1, S2, xn2, a.maxiter);                               \
                slot_store(&s, t_);                                                                        \
            } else {                                                                                       \
                st = minres_post(s, pre, (kn), S0, S1, S2, xn2, a.maxiter);                                \
            }                                                                                              \
            NEXT_STAMP(10)                                                                                 \
            if (SHARE && threadIdx.x == 0) {                                                               \
                s_bcast[0] = st.ca; s_bcast[1] = st.cb; s_bcast[2] = st.cc; s_bcast[3] = st.sj;            \
                s_bcast[4] = st.oldeps; s_bcast[5] = st.delta; s_bcast[6] = st.denom; s_bcast[7] = st.phi; \
                s_bcast[8] = st.rotate ? 1.0 : 0.0; s_bcast[9] = st.stop ? 1.0 : 0.0;                      \
            }                                                                                              \
        }                                                                                                  \
        if (SHARE) {                                                                                       \
            __syncthreads();                                                                               \
            st.ca = s_bcast[0]; st.cb = s_bcast[1]; st.cc = s_bcast[2]; st.sj = s_bcast[3];                \
            st.oldeps = s_bcast[4]; st.delta = s_bcast[5]; st.denom = s_bcast[6]; st.phi = s_bcast[7];     \
            st.rotate = s_bcast[8] != 0.0; st.stop = s_bcast[9] != 0.0;                                    \
        }                                                                                                  \
    } while (0)
    if (lead) pre = minres_pre(s);
    OCC_NEXT_STEP(1);
    st.ca = 1.0;  // step 1: kry_form_p returns its first argument (p_0 at the site and at its neighbours) exactly
#undef BAR_STAMP
#define BAR_STAMP(pt) SOLVE_STAMP(pt)
    for (; !failed; ++k) {
        SOLVE_STAMP(0)
        if (st.stop) break;
        SOLVE_STAMP(1)
        double part[4] = {0.0, 0.0, 0.0, 0.0};
        if (st.rotate) {  // w_{k-2}, x_{k-2}
            const double2 w = kry_form_w(st, pm2, wm2, wm1);
            x.x = fma(st.phi, w.x, x.x);
            x.y = fma(st.phi, w.y, x.y);
            wm2 = wm1;
            wm1 = w;
            part[3] = dot2(x, x);
        }
        {
            const double2 p = kry_form_p(st, g, pm2, pm1);     // p_{k-1}
            const double2 h = kry_apply<NW>(d, av, g, ng);     // A g_{k-1}
            const double2 gn = kry_form_p(st, h, gm2, g);      // g_k = A p_{k-1}, by the same three-term recurrence
            gm2 = g;
            g = gn;
            part[0] = dot2(p, p);
            part[1] = fma(p.y, gn.y, p.x * gn.x);
            if (k >= 2) part[2] = dot2(p, pm1);
            pm2 = pm1;
            pm1 = p;
            store_x<XL>(gbuf[k & 1], myoff, g);
        }
        if (!act) { part[0] = 0.0; part[1] = 0.0; part[2] = 0.0; part[3] = 0.0; }
        if (XL) {
            // ---- records as arrival flags (see "XL step exchange" above): every wave publishes its slice's record
            if (lane < 2) store_x<1>(pbuf[(k + 1) % 3], slice * 32 + lane * 16, rec_canary());
            wave_sum4(part);  // (a slice past the last site sums zeros)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's g and its canaries of step k + 1 are in the XCD's L2
            if (lane < 2) store_x<1>(pbuf[k % 3], slice * 32 + lane * 16, lane == 0 ? make_double2(part[0], part[1]) : make_double2(part[2], part[3]));
            SOLVE_STAMP(2)
            if (lead) pre = minres_pre(s);  // the slot-only half of step k + 1, while the other slices arrive
            SOLVE_STAMP(9)
            if (threadIdx.x < 64) {  // the first wave polls the records of the chain's slices
                double acc[4];
                const bool ok = poll_slice_records<false>(pbuf[k % 3], a.nb_n, lane, spin_limit, sc, acc);
                SOLVE_STAMP(6)
                if (ok) {
#pragma unroll
                    for (int kk = 0; kk < NW; ++kk) ng[kk] = load_sc1(gbuf[k & 1], off[kk]);
                }
                SOLVE_STAMP(7)
                S0 = acc[0]; S1 = acc[1]; S2 = acc[2]; xn2 = acc[3];
                if (threadIdx.x == 0) {
                    s_flag = ok ? 0 : 1;
                    if (!ok) chain_fail(sc);
                    if (!SHARE) { s_bcast[0] = S0; s_bcast[1] = S1; s_bcast[2] = S2; s_bcast[3] = xn2; }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler only: no load moves above the poll
            if (s_flag) { failed = true; break; }
            if (threadIdx.x >= 64) {
#pragma unroll
                for (int kk = 0; kk < NW; ++kk) ng[kk] = load_sc1(gbuf[k & 1], off[kk]);
                if (!SHARE) { S0 = s_bcast[0]; S1 = s_bcast[1]; S2 = s_bcast[2]; xn2 = s_bcast[3]; }
            }
            SOLVE_STAMP(8)
        } else {
        if (slice_act) {  // per-slice sums: the granularity (and order) of k_minres at 64 threads per block
            wave_sum4(part);
            if (lane == 0) store_x<XL>(pbuf[k & 1], slice * 32, make_double2(part[0], part[1]));
            if (lane == 1) store_x<XL>(pbuf[k & 1], slice * 32 + 16, make_double2(part[2], part[3]));
        }
        SOLVE_STAMP(2)
        ++nbar;
        OCC_CHAIN_ARRIVE();
        if (lead) pre = minres_pre(s);  // the slot-only half of step k + 1, while the other workgroups arrive
        SOLVE_STAMP(9)
        OCC_CHAIN_WAIT(s_flag);
        SOLVE_STAMP(6)
        if (s_flag) { failed = true; break; }
        // ---- everything below reads what other workgroups published in this step: sc1 loads only
#pragma unroll
        for (int kk = 0; kk < NW; ++kk) ng[kk] = load_sc1(gbuf[k & 1], off[kk]);
        if (lead) {  // the canonical order from per-slice sums
            double acc[4];
            sum_slices_canonical(a.nb_n, lane, acc, [&](int slice, double (&v)[4]) {
                const double2 lo = load_sc1(pbuf[k & 1], slice * 32), hi = load_sc1(pbuf[k & 1], slice * 32 + 16);  // past the end: zeros
                v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
            });
            SOLVE_STAMP(7)
            S0 = acc[0]; S1 = acc[1]; S2 = acc[2]; xn2 = acc[3];
        }
        SOLVE_STAMP(8)
        }
#undef NEXT_STAMP
#define NEXT_STAMP(pt) SOLVE_STAMP(pt)
        OCC_NEXT_STEP(k + 1);
        SOLVE_STAMP(11)
#undef NEXT_STAMP
#define NEXT_STAMP(pt)
    }
#undef OCC_NEXT_STEP
#undef BAR_STAMP
#define BAR_STAMP(pt)
    if (writer) {
        if (failed) { s.done = 1; s.istop = 6; s.itn = k; }
        slot_store(&a.slots[chain64 * NSLOT], s);
        sc.minres_itn_last = s.itn;
        sc.krylov_total += (unsigned long long)s.itn;
        sc.krylov_sq_total += (unsigned long long)s.itn * (unsigned long long)s.itn;
        sc.solves += 1ull;
        if (s.istop == 6 && !failed) sc.err = -3;  // OCC_E_MINRES (logit.py:91-92)
    }
    }

    // ---- phase C: sum-to-zero projection, eta, partial sums of beta's system.  The solve stopped at the top
    // of step k, uniformly over the chain: buffers of parity k are free (everybody has passed barrier k-1).
    // (scalar-wave form: the step that learnt `stop` has already stored its records in buffer k % 3 -- the sums go to the
    // buffer of step k - 1, which every workgroup has finished reading)
    double proj_a = 0.0;
    const int PROJ_BUF = XL ? (SW ? (k + 2) % 3 : k % 3) : (k & 1);
    PHASE_STAMP(STAMP_STEPS - 1, 0)
    if (!failed) {
        if (slice_act) {
            const double t0 = wave_sum(act ? x.x : 0.0), t1 = wave_sum(act ? x.y : 0.0);
            if (lane == 0) store_x<XL>(pbuf[PROJ_BUF], slice * 32, make_double2(t0, t1));
        }
        ++nbar;
        OCC_CHAIN_BARRIER(s_flag);
        failed = s_flag != 0;
    }
    if (!failed) {
        if (lead) {
            double sx = 0.0, sz = 0.0;
            for (int b0 = lane; b0 < a.nb_n; b0 += 256) {
                double2 v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = load_sc1(pbuf[PROJ_BUF], (b0 + 64 * r) * 32);
#pragma unroll
                for (int r = 0; r < 4; ++r) { sx += v[r].x; sz += v[r].y; }
            }
            sx = wave_sum(sx);
            sz = wave_sum(sz);
            proj_a = -sx / sz;
            if (SHARE && threadIdx.x == 0) s_bcast[10] = proj_a;
        }
        if (SHARE) {
            __syncthreads();
            proj_a = s_bcast[10];
        }
        PHASE_STAMP(STAMP_STEPS - 1, 1)
    } else if (writer) {
        chain_fail(sc);  // OCC_E_HIP: the host falls back to one launch per MINRES step
    }
    double eta = 0.0;
    if (act && !failed) {  // a failed solve leaves the warm start and eta as they were: the host re-runs the iteration
        eta = eta_project(x, proj_a);
        a.Xv[co + i] = x;
        ia.eta[co + i] = eta;
    }
    OCC_SWITCH_DIM(ia.p, beta_partials_slice<D>(ia, chain, i, act, slice, slice_act, om, eta, zval));
    if (writer) {
        Ctl m = ctl;
        m.koff = 0u;
        sc.mid[e] = m;
        sc.bar_base = bar_base + nbar * (XL ? 1u : (unsigned)ia.nbg);
        atomicMin(ia.clock, clk0);
        atomicMax(ia.clock + 1, (unsigned long long)wall_clock64());
    }
    PHASE_STAMP(STAMP_STEPS - 1, 2)
}

}  // namespace occ
