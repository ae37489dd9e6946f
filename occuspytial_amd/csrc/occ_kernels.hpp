// occ_kernels.hpp -- the kernels of the Gibbs iteration other than the fused k_iter (gfx950; chains batched on blockIdx.y).
//
// One Gibbs iteration of LogitICARGibbs.step() (occuspytial/gibbs/logit.py:254-266) is a small DAG of kernels.
// Global sums are "reduce at the consumer": a producer writes one partial per block (or 64-site slice) and
// quantity, and every consumer re-reduces all partials in the same fixed order, so scalars are bit-identical
// everywhere and in every run (no float atomics anywhere).
//
// The reference's order is omega_b, tau, eta, beta, omega_a, alpha, z.  omega_a/alpha of iteration t
// read only alpha and z of iteration t-1, and the next iteration's omega_b reads only beta and eta of
// iteration t, so (with counter-based variates, results do not depend on execution order):
//
//   main stream  [tau, rhs, eta solve, projection, beta sums] -> k_z_ob
//   side stream  k_omega_a -> k_noise (alpha of t, noise of t+1)  (needed by k_z_ob / the next iteration)
//
// The bracket is ONE launch of k_iter (occ_iter.hpp) when all its workgroups fit on the device, else
// k_eta_init -> k_minres x (cap+3) -> k_beta_partial, where a kernel boundary is the only grid-wide
// synchronisation and a solve that needs more launches than were captured carries over (Ctl::koff).
//
//   k_omega_b      omega_b ~ PG(1, x'beta + eta) per site; eta'Q eta partials (logit.py:195-204, 208);
//                  stand-alone only for the first iteration after new start values -- afterwards it is
//                  the second role of k_z_ob
//   k_noise        the alpha draw of this iteration (logit.py:224; first block of every chain), then the variates of the
//                  eta right-hand side that depend on nothing but the iteration number, one iteration ahead: site
//                  normals and the edge form of the ICAR prior term (logit.py:75-77), tau's gamma variate
//   k_eta_init     tau ~ Gamma (logit.py:206-209); rhs y (logit.py:213, 78); r1 = [y;1] - Lambda x0
//   k_minres       one Lanczos/MINRES iteration of the joint 2n system per launch
//                  (scipy _isolve/minres.py as called at logit.py:87)
//   k_beta_partial eta = x - (sum x / sum z) z (distributions.pyx:24-39); X' Omega X, X'(k - omega eta)
//   k_omega_a      omega_a ~ PG(1, w'alpha) for rows of existing sites; W' Omega W, W'(y - 1/2)
//                  (logit.py:180-193, 219-223)
//   k_alpha_draw   alpha draw (logit.py:224) as a kernel of its own: occ_cond_alpha (injected variates) and occ_profile
//   k_z_ob         beta draw (logit.py:232, every wave); role 0: z update (logit.py:234-252), record
//                  (alpha, beta, tau) (base.py:238-239), advance the iteration; role 1 (other half of the
//                  grid): omega_b of the NEXT iteration
//   k_gate         head of a side-stream sequence when the streams hand over through device counters
// (the reduced-rank model's kernels: occ_rsr.hpp)
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "occ_rng.hpp"

namespace occ {

constexpr int MAXC = 8;                             // covariates of the register-resident fast path (templates on P, Q)
constexpr int NACC_MAX = MAXC * (MAXC + 1) / 2 + MAXC;  // 44
constexpr int MAXG = 32;                            // OCC_MAX_COVARIATES: the generic path (P = 0 instantiations, run-time p and q)
constexpr int NACC_G = MAXG * (MAXG + 1) / 2 + MAXG;    // 560
constexpr int NSLOT = 4;
constexpr int MAX_WAVES = 4;  // threads per block <= 256

__host__ __device__ constexpr int nacc(int d) { return d * (d + 1) / 2 + d; }

// MINRES scalar state of one chain; slot s is written by step s and read by step s+1.
#ifndef OCC_SLOT_ALIGN
#define OCC_SLOT_ALIGN 8
#endif
struct alignas(OCC_SLOT_ALIGN) Slot {
    double beta1, beta, oldb, alfa, dbar, epsln, phibar, rhs1, rhs2, tnorm2, gmax, gmin, cs, sn, root;
    double ibeta;  // 1 / beta: the division of the step that formed beta, reused by the next one (its `sj`)
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
    X(cs) X(sn) X(root) X(ibeta) X(itn) X(istop) X(done)
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
    uint32_t koff;  // Krylov launches already spent on the current eta solve by earlier graph replays:
                    // 0 normally; > 0 when a replay ran out of captured launches and the NEXT replay
                    // continues the same solve (no host involvement, same arithmetic)
};

// Control words are handed over between kernels, never updated in place: the kernels of launch
// sequence ("slot") number s read ctl[s & 1]; k_z_ob, the last kernel of the slot, writes ctl[(s+1) & 1];
// k_beta_partial publishes the carry decision of the slot in mid[s & 1].  No kernel reads a word that
// another block of the same kernel writes.
struct ChainScalars {
    double alpha[MAXG], beta[MAXG];
    double tau;
    double tau_gamma[2];       // the standard gamma variate of tau's draw of iteration t in [t & 1] (logit.py:209): it depends on
                               // nothing but (key, t), so k_noise draws it one iteration ahead, off the critical path
    uint64_t key;
    Ctl ctl[2], mid[2];
    uint32_t it_stop, it_base, burnin, keep;
    uint32_t bar_base;         // arrivals counted so far by the chain's barrier counter (occ_iter.hpp), never reset
    int32_t err;               // OCC_E_* raised on device
    int32_t minres_itn_last;
    unsigned long long krylov_total, krylov_sq_total, solves, carries;
};

// Variates handed in by the caller instead of the chain's Philox streams: the per-conditional entry points of the C ABI
// (occ_cond_*, include/occ_gibbs.h) run the kernels below in their INJ instantiation, which take the standard gamma variate
// of tau, the standard normals of the beta / alpha draws and the uniforms of the z update from here (and omega_a / omega_b
// from the state buffers instead of drawing them).  Production launches never read it.
struct Inject {
    double gamma;             // standard gamma variate of shape tau_shape                          logit.py:209
    double beta_eps[MAXG];    // standard normals of precision_mvnorm                               distributions.pyx:95-96
    double alpha_eps[MAXG];
    const double *z_u;        // [n] uniform of site i (sites with a detection ignore theirs)        logit.py:247-251
    int tau_from_gamma;       // k_eta_init: 1 = tau = gamma / rate, 0 = keep the chain's tau
    int do_beta, do_z;        // k_z_ob: draw beta / update z
    int pad_;
};

// The problem/state descriptor lives in device memory and kernels receive a POINTER to it (plus the
// two hot per-chain tables): a by-value 360-byte kernel argument costs every wave several serialized
// kernarg-segment fetches at kernel entry -- measured 12-25 us per launch on MI355X for kernels whose
// scalar loads the compiler could not batch -- while a 40-byte argument block is one fetch.
struct Ctx {
    int n, S, R, p, q, C;
    int nb_n, nb_r;  // blocks over sites / visit rows = number of partial sums per quantity
    long long maxiter;
    // fixed inputs (shared by all chains)
    int ell_w;  // > 0: every 64-row slice has this width (uniform ELL): slice base is arithmetic, no sell_ptr load
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
    // per-chain state ([2] = double-buffered by the parity of the ITERATION number they belong to)
    double *eta, *rhs, *omega_a;
    double *omega_b[2];           // omega_b of iteration t in omega_b[t & 1]
    double *enorm[2], *uprior[2]; // site normals and edge prior term of iteration t in [t & 1]
    uint8_t *z;
    double2 *Gv[2], *Pv[3], *Wv[2], *Xv;  // g_m = A p_{m-1}, p_m = r2_m, w_m, x: (x-part, z-part) interleaved
    // per-wave partial sums (one region per producer, so that independent kernels may overlap)
    double *part_quad;   // [C][nb_n]            eta'Q eta            k_omega_b -> k_eta_init
    double *part_kry;    // [C][2][4 nb_n]       MINRES sums          k_minres  -> k_minres
    double *part_proj;   // [C][2 nb_n]          sum x, sum z         k_minres  -> k_beta_partial
    double *part_beta;   // [C][nacc(p) nb_n]    X'OX, X'(k - o eta)  k_beta_partial -> k_z_ob
    double *part_alpha;  // [C][nacc(q) nb_r]    W'OW, W'(y - 1/2)    k_omega_a -> k_alpha_draw
    Slot *slots;        // [C][NSLOT]
    ChainScalars *sc;   // [C]
    double *rec;        // [C][keep][q + p + 1]
    unsigned *bar;      // [C][32] barrier / ticket counters of k_iter (occ_iter.hpp); null: not used
    unsigned *claim;    // [C][16] one XCD per chain: the next workgroup slot of the chain (k_iter claims, k_z_ob resets); null: not used
    // k_iter's own clock (constant-rate wall clock): {earliest start, latest end} of the running launch over
    // its chains, and {sum of launch durations, launches} since the last occ_run began.  k_iter's chain
    // writers fill the first pair, k_z_ob (the next kernel) folds it into the second.
    unsigned long long *iter_clock;
    // Hand-overs between the main and the side stream by device-side sequence counters instead of events
    // (only when the two streams own disjoint sets of CUs, so that a workgroup spinning on one of them can
    // never keep the producer it waits for off the device); null: not used.  See "Stream hand-overs" below.
    unsigned *sync;
    const Inject *inj;  // injected variates of the occ_cond_* entry points (INJ kernels only); null otherwise
    // Reference-form prior draw (occ_problem::prior_factor): u = F eps2, F row-major n x dense_m; eps2 of iteration t in
    // dense_eps[t & 1] ([C][dense_m]), written by k_noise, multiplied by k_prior_dense.  Null: the edge form.
    const double *dense_F;
    double *dense_eps[2];
    int dense_m;
    // The streams' CU masks may give the XCDs different numbers of CUs (k_iter's scalar-wave form with fewer than eight
    // chains: occ_gibbs.hip).  A grid is dealt to the XCDs in equal shares whatever they can take, so the kernels that fill
    // the device (k_z_ob on the main stream; k_omega_a, k_noise on the side stream) then hand out their tiles in
    // proportion to the CUs their stream owns there: tile_first[k][x] = first tile of XCD x for kernel k (0: k_z_ob, main
    // stream; 1: k_omega_a, 2: k_noise, side stream), tile_most[k] = the largest share; share_on = 0: the plain map.
    int share_on;
    int tile_first[3][9], tile_most[3];
    int surplus_last;  // bit k: kernel k's surplus workgroups come last (see tile_of_block_shared)
};

// ---- reductions ----------------------------------------------------------------------------------
// Wave-level sums use DPP row operations (register-to-register, a few cycles each) instead of
// ds_bpermute shuffles, and leave the total in every lane through v_readlane: no LDS, no barrier.
// Partial sums are kept per WAVE (not per block), so producing and consuming them needs no
// __syncthreads at all; every wave of the consumer re-reduces all partials in the same fixed order,
// which makes the derived scalars bit-identical in every wave, block and run (no float atomics).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shifted(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_shifted<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_shifted<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_shifted<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_shifted<0x140, 0xf>(v);  // row_mirror: every lane of a 16-lane row holds the row sum
    v += dpp_shifted<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_shifted<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3: lane 63 holds the wave sum
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);  // uniform
}

// Four wave sums at once, each in wave_sum's order (bit-identical to four calls): written level by level so that the four
// independent dependency chains sit next to each other -- a DPP move has to wait for the add that feeds it, and four
// calls in a row expose that wait 24 times.  (Kept as the statement of the order; wave_sum4 below returns the same bits
// with half the instructions.)
__device__ __forceinline__ void wave_sum4_plain(double (&v)[4])
{
#define OCC_LEVEL(CTRL, MASK)                                                      \
    {                                                                               \
        double t_[4];                                                               \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) t_[q_] = dpp_shifted<CTRL, MASK>(v[q_]); \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) v[q_] += t_[q_];           \
    }
    OCC_LEVEL(0xB1, 0xf)
    OCC_LEVEL(0x4E, 0xf)
    OCC_LEVEL(0x141, 0xf)
    OCC_LEVEL(0x140, 0xf)
    OCC_LEVEL(0x142, 0xa)
    OCC_LEVEL(0x143, 0xc)
#undef OCC_LEVEL
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long b = __double_as_longlong(v[q]);
        const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
        v[q] = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);  // uniform
    }
}

// The same four sums, the same additions -- (l0+l1), +(l2+l3) within a quad, quads by pairs within a row of 16, rows
// (R0+R1), (R2+R3), then their sum -- with the four quantities TRANSPOSED over the lanes of a quad on the way: after the
// first level a lane carries two of them (even lanes 0 and 1, odd lanes 2 and 3), after the second one, and the remaining
// four levels move one value instead of four: 36 instructions where the plain form takes 80.  Those 80 were 40 % of what a
// site wave of k_iter issues per MINRES step (two such waves share a SIMD) and sit once more on the polling wave's
// path.  The last two levels exchange whole rows with v_permlane16_swap / v_permlane32_swap (gfx950).  Floating-point
// addition is commutative: which lane holds a partial sum does not change its bits (tests/test_gpu_rng.py compares the
// two forms on the device).
typedef unsigned occ_v2u __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ void wave_sum4(double (&v)[4])
{
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool p1 = lane & 1u, p2 = lane & 2u;
    // level 1 (lane ^ 1): even lanes keep quantities 0, 1 and take the neighbour's, odd lanes 2, 3
    double a = p1 ? v[2] : v[0], b = p1 ? v[3] : v[1];
    const double sa = p1 ? v[0] : v[2], sb = p1 ? v[1] : v[3];
    a += dpp_move<0xB1>(sa);  // quad_perm [1,0,3,2]
    b += dpp_move<0xB1>(sb);
    // level 2 (lane ^ 2): lanes 0, 1 of a quad keep a (quantities 0, 2), lanes 2, 3 keep b (quantities 1, 3)
    double c = p2 ? b : a;
    const double sc = p2 ? a : b;
    c += dpp_move<0x4E>(sc);  // quad_perm [2,3,0,1]: lane 4 j + t holds the sum of quad j of quantity {0, 2, 1, 3}[t]
    c += dpp_move<0x114>(c);  // row_shr:4: quads 1 and 3 of a row hold Q0 + Q1, Q2 + Q3
    c += dpp_move<0x118>(c);  // row_shr:8: quad 3 holds the row sum
    {  // rows 1 and 3 take rows 0 and 2, then row 3 takes row 1
        const long long bits = __double_as_longlong(c);
        const occ_v2u lo = __builtin_amdgcn_permlane16_swap((unsigned)bits, (unsigned)bits, false, false);
        const occ_v2u hi = __builtin_amdgcn_permlane16_swap((unsigned)(bits >> 32), (unsigned)(bits >> 32), false, false);
        c += __longlong_as_double(((long long)hi.x << 32) | lo.x);
    }
    {
        const long long bits = __double_as_longlong(c);
        const occ_v2u lo = __builtin_amdgcn_permlane32_swap((unsigned)bits, (unsigned)bits, false, false);
        const occ_v2u hi = __builtin_amdgcn_permlane32_swap((unsigned)(bits >> 32), (unsigned)(bits >> 32), false, false);
        c += __longlong_as_double(((long long)hi.x << 32) | lo.x);
    }
    const long long bits = __double_as_longlong(c);
    constexpr int src[4] = {60, 62, 61, 63};  // lanes 60 .. 63 hold quantities 0, 2, 1, 3
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int lo = __builtin_amdgcn_readlane((int)bits, src[q]), hi = __builtin_amdgcn_readlane((int)(bits >> 32), src[q]);
        v[q] = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);  // uniform
    }
}

// The canonical order of the four MINRES sums over the 64-site slices of a chain, from PER-SLICE sums with coalesced loads:
// lane l adds its slices l, l + 64, l + 128, ... in turn (four rounds of loads in flight), then ONE wave sum.  Every path
// that has to return the same bits takes its totals here or in the same order (k_minres at 64 threads per block, k_iter
// any placement; the one-XCD forms of k_iter while they poll the slices' records: poll_slice_records in occ_iter.hpp).  The
// order does not depend on how slices are grouped into workgroups, so a form of the fused kernel may change its workgroup
// size (448 sites beside a scalar wave, 256, 512) without changing a bit of the result.  (Round 2 had a tree over groups
// of eight slices, which a 512-site workgroup could pre-reduce into one record; it tied the order to that workgroup size.)
// `load(slice, v)` fills the four sums of a slice (zeros past the last one).
template <class F>
__device__ __forceinline__ void sum_slices_canonical(int nslices, int lane, double (&tot)[4], F load)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) tot[q] = 0.0;
    for (int base = 0; base < nslices; base += 256) {  // (uniform trip count: rounds past the end add exact zeros)
        double v[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) load(base + 64 * r + lane, v[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tot[q] += v[r][q];
        }
    }
    wave_sum4(tot);
}

// XCD-aware workgroup -> (chain, tile) map.  MI355X deals workgroups round-robin over its 8 XCDs (each
// with a private L2), so with the plain map consecutive tiles of one chain land on 8 different L2s and
// every neighbour gather of the lattice re-fetches its lines into several of them (measured: fabric
// reads 2.6x the algorithmic bytes of k_minres).  Here workgroups that share an XCD (equal linear id
// mod 8) take CONSECUTIVE tiles, so a tile's neighbours sit in the same L2.  The map is a bijection
// for every grid size; it only changes where a tile runs, never what is computed.
struct Tile {
    int chain, blk;
};
__device__ __forceinline__ Tile tile_of_block(int chain_base)
{
    const unsigned gx = gridDim.x, nwg = gx * gridDim.y;
    const unsigned lin = blockIdx.y * gx + blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = lin & 7u;
    const unsigned first = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    const unsigned id = first + (lin >> 3);
    Tile t;
    t.chain = chain_base + (int)(id / gx);
    t.blk = (int)(id % gx);
    return t;
}

// ... with the tiles handed out in proportion to the CUs the stream owns on each XCD (Ctx::tile_first of kernel `kid`): a 1-D
// grid of 8 x (largest share) workgroups, workgroup `lin` being the (lin / 8)-th of XCD lin % 8 -- that is how a grid is
// dealt, strictly round-robin whatever the XCDs can take (tools/xcc_probe6); only speed depends on it, the map is a
// bijection between workgroups and tiles either way.  An XCD with a smaller share returns its surplus workgroups
// (chain = -1) FIRST: the dealing stalls on a workgroup whose XCD is full, so a surplus workgroup dealt late to a full XCD
// held up the workgroups behind it that other XCDs still had room for (k_z_ob: a second round, 15 -> 22 us).
__device__ __forceinline__ Tile tile_of_block_shared(const Ctx &c, int kid, int per_chain, int chain_base)
{
    if (!c.share_on) return tile_of_block(chain_base);
    const int lin = (int)blockIdx.x, xcd = lin & 7, idx = lin >> 3;
    const int first = c.tile_first[kid][xcd], surplus = c.tile_most[kid] - (c.tile_first[kid][xcd + 1] - first);
    Tile t;
    t.chain = -1;
    t.blk = 0;
    const bool last = (c.surplus_last >> kid) & 1;
    if (last ? idx < c.tile_most[kid] - surplus : idx >= surplus) {
        const int id = first + idx - (last ? 0 : surplus);
        t.chain = chain_base + id / per_chain;
        t.blk = id % per_chain;
    }
    return t;
}

// Partial sums are kept per BLOCK.  With one wave per block (small problems, where latency rules) this
// needs no LDS and no barrier at all; with several waves the block combines its waves through LDS (one
// barrier), so that the number of partials -- which every consumer block re-reads in full -- grows with
// n / threads-per-block, not with the number of waves.
template <int NQ>
__device__ __forceinline__ void block_partials(const double (&v)[NQ], double *out, int nb, int blk)
{
    __shared__ double s_part[MAX_WAVES * NQ];
    const int nw = blockDim.x >> 6, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double r[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) r[qi] = wave_sum(v[qi]);
    if (nw == 1) {
        if (lane == 0) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) out[qi * nb + blk] = r[qi];
        }
        return;
    }
    if (lane == 0) {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) s_part[wave * NQ + qi] = r[qi];
    }
    __syncthreads();
    if ((int)threadIdx.x < NQ) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += s_part[w * NQ + threadIdx.x];
        out[threadIdx.x * nb + blk] = t;
    }
}

// Every block reduces all nb partials of NQ quantities in the same fixed order (registers only when the
// block is one wave; otherwise wave 0 reduces and broadcasts through LDS).
template <int NQ>
__device__ __forceinline__ void reduce_partials(const double *part, int nb, double (&out)[NQ])
{
    __shared__ double s_tot[NQ];
    const int lane = threadIdx.x & 63;
    const bool one_wave = blockDim.x == 64;
    if (one_wave || threadIdx.x < 64) {
        double acc[NQ];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) acc[qi] = 0.0;
        // several rounds of loads in flight at a time (a plain "load, add" loop waits for every round trip in
        // turn); the sums are accumulated in the same order as ever, rounds past the end add an exact 0
        constexpr int R = NQ <= 8 ? 4 : (NQ <= 20 ? 2 : 1);  // registers: R * NQ doubles
        for (int b0 = lane; b0 < nb; b0 += 64 * R) {
            double v[R][NQ];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int b = b0 + 64 * r;
                const int bc = min(b, nb - 1);
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    const double t = part[qi * nb + bc];
                    v[r][qi] = (b < nb) ? t : 0.0;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) acc[qi] += v[r][qi];
            }
        }
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) out[qi] = wave_sum(acc[qi]);
        if (!one_wave && lane == 0) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) s_tot[qi] = out[qi];
        }
    }
    if (!one_wave) {
        __syncthreads();
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) out[qi] = s_tot[qi];
    }
}

// Runtime quantity count (the p x p / q x q systems): waves share the quantities, results go to LDS.
__device__ __forceinline__ void reduce_partials_lds(const double *part, int nq, int nw, double *lds_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    for (int qi = wave; qi < nq; qi += nwb) {
        double s = 0.0;
        const double *pq = part + (size_t)qi * nw;
        for (int b0 = lane; b0 < nw; b0 += 512) {  // eight rounds of loads in flight, added in the order of a plain loop
            double v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int b = b0 + 64 * r;
                const double t = pq[min(b, nw - 1)];
                v[r] = (b < nw) ? t : 0.0;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) s += v[r];
        }
        s = wave_sum(s);
        if (lane == 0) lds_out[qi] = s;
    }
    __syncthreads();
}

// Generic path (more than MAXC covariates: the number of quantities is only known at run time): the terms of the
// p x p / q x q systems are EMITTED one at a time -- wave sum at once, one LDS word per wave and quantity -- and
// combined over the block's waves by finish() (one barrier), in the order of block_partials.
struct PartialEmitter {
    double *out;
    int nb, blk, nq;
    __device__ __forceinline__ void emit(int t, double v)
    {
        extern __shared__ double s_gpart[];  // [waves][nq] (dynamic LDS of the P = 0 kernels)
        const double r = wave_sum(v);
        if ((threadIdx.x & 63) == 0) {
            if (blockDim.x == 64) out[(size_t)t * nb + blk] = r;
            else s_gpart[(threadIdx.x >> 6) * nq + t] = r;
        }
    }
    __device__ __forceinline__ void finish()
    {
        extern __shared__ double s_gpart[];
        if (blockDim.x == 64) return;
        __syncthreads();
        const int nw = blockDim.x >> 6;
        for (int t = threadIdx.x; t < nq; t += blockDim.x) {
            double acc = 0.0;
            for (int w = 0; w < nw; ++w) acc += s_gpart[w * nq + t];
            out[(size_t)t * nb + blk] = acc;
        }
    }
};
__host__ __device__ constexpr size_t generic_lds_bytes(int nq, int threads) { return threads == 64 ? 0 : sizeof(double) * (size_t)(threads / 64) * nq; }

// Off-diagonal slots of the 64-row slice that holds site i: first slot and number of slots per row.
__device__ __forceinline__ void slice_of(const Ctx &c, int i, int &base, int &width)
{
    const int slice = i >> 6;
    if (c.ell_w > 0) {
        width = c.ell_w;
        base = slice * c.ell_w * 64;
    } else {
        base = c.sell_ptr[slice];
        width = (c.sell_ptr[slice + 1] - base) >> 6;
    }
}



__device__ __forceinline__ double expit(double x)
{
    if (x < 0.0) { const double e = exp(x); return e / (1.0 + e); }
    return 1.0 / (1.0 + exp(-x));
}

// ---- site-level arithmetic shared by the stand-alone kernels and the fused iteration kernel (occ_iter.hpp):
// written with explicit contractions so that both evaluate the same operations.
__device__ __forceinline__ double xdot(const double *Xt, int n, int i, const double *coef, int p)
{
    double acc = 0.0;
    for (int a = 0; a < p; ++a) acc = fma(Xt[(size_t)a * n + i], coef[a], acc);
    return acc;
}
// right-hand side of the eta system at one site: k - omega x'beta + sqrt(omega) eps_1 + sqrt(tau) u   (logit.py:76-78, 213)
__device__ __forceinline__ double eta_rhs_site(double om, double xb, double z, double en, double up, double sqrt_tau)
{
    const double b = fma(-om, xb, z - 0.5);
    return fma(sqrt_tau, up, fma(sqrt(om), en, b));
}
// eta = x - (sum x / sum z) z at one site (distributions.pyx:24-39) and the right-hand side term of beta's system
__device__ __forceinline__ double eta_project(double2 xz, double a) { return fma(a, xz.y, xz.x); }
__device__ __forceinline__ double beta_rhs_term(double om, double eta, double z) { return fma(-om, eta, z - 0.5); }

// distributions.pyx:95-105 on device, executed by ONE thread on small LDS work arrays (runtime
// dimension d <= MAXC, so no per-dimension template and no register arrays): upper Cholesky U of the
// d x d precision (packed upper accumulators + prior), out = prec^-1 b + U^-1 eps.  U is d x d,
// work is 2d doubles.  Returns false when a pivot is not positive.
__device__ inline bool precision_mvnorm_dev(int d, const double *acc /* nacc(d): upper then rhs */,
                                            const double *prec0, const double *pbm, uint64_t key, uint32_t it,
                                            uint32_t stream, double *U, double *work, double *out, const double *eps_inj = nullptr)
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
        const double e = eps_inj ? eps_inj[k] : block_normal(key, (uint32_t)k, 0, it, stream);
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

// Same draw with the dimension known at compile time, entirely in registers; executed redundantly by
// every lane (uniform inputs, uniform control flow), so no broadcast is needed afterwards.
template <int D>
__device__ __forceinline__ bool precision_mvnorm_reg(const double (&acc)[nacc(D)], const double *prec0, const double *pbm,
                                                     uint64_t key, uint32_t it, uint32_t stream, double (&out)[D],
                                                     const double *eps_inj = nullptr)
{
    double U[D][D], r[D], o[D];
    int t = 0;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = a; b < D; ++b) U[a][b] = acc[t++] + prec0[a * D + b];
#pragma unroll
    for (int a = 0; a < D; ++a) r[a] = acc[t++] + pbm[a];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double sj = U[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) sj -= U[k][j] * U[k][j];
        if (!(sj > 0.0)) ok = false;
        const double ujj = sqrt(sj);
        U[j][j] = ujj;
#pragma unroll
        for (int i = j + 1; i < D; ++i) {
            double v = U[j][i];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= U[k][j] * U[k][i];
            U[j][i] = v / ujj;
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) o[i] = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double e = eps_inj ? eps_inj[k] : block_normal(key, (uint32_t)k, 0, it, stream);
#pragma unroll
        for (int i = k; i < D; ++i) o[i] += U[k][i] * e;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) o[i] += r[i];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double v = o[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v -= U[k][i] * o[k];
        o[i] = v / U[i][i];
    }
#pragma unroll
    for (int i = D - 1; i >= 0; --i) {
        double v = o[i];
#pragma unroll
        for (int k = i + 1; k < D; ++k) v -= U[i][k] * o[k];
        o[i] = v / U[i][i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) out[i] = o[i];
    return ok;
}


// ---- Stream hand-overs without events ---------------------------------------------------------------
// Launch sequence number j (one Gibbs iteration of every chain) is  k_iter(j), k_z_ob(j)  on the main stream
// and  k_gate(j), k_omega_a(j), k_alpha_draw(j), k_noise(j)  on the side stream.  A kernel never announces its
// own completion: the FIRST thread of the NEXT kernel of the same stream does (stream order and the
// kernel-boundary release make that exact and free -- no tickets, no fences, no write-through stores):
//   sync[SYNC_MAIN]   = j  set by k_iter(j):  k_z_ob(j-1) is complete     k_gate(j) waits for >= j   (z, control words)
//   sync[SYNC_NOISE]  = j  set by k_gate(j):  k_noise(j-1) is complete    k_iter(j) waits for >= j   (noise of j)
//   sync[SYNC_ALPHA] += 1 per chain by k_noise(j) once alpha of j is stored       k_z_ob(j) waits for >= (j+1) C (alpha of j)
// Every kernel sets before it waits, so the two streams cannot wait for each other.  Each stream counts its
// own sequences in words only it touches (SYNC_MAIN_SEQ + parity of the sequence: the last kernel of a main-stream
// sequence writes the word of the NEXT sequence's parity, so no kernel reads a word that is written while it runs;
// SYNC_SIDE_SEQ: written by k_noise, read by k_gate).
// Consumers that wait inside a running kernel (k_iter, k_z_ob) read the data with L1-bypassing agent-scope
// loads; kernels launched after k_gate rely on the kernel boundary.  Waits are bounded.
enum : int { SYNC_MAIN = 0, SYNC_ALPHA = 16, SYNC_NOISE = 32, SYNC_MAIN_SEQ = 48, SYNC_SIDE_SEQ = 52, SYNC_ABORT = 56, SYNC_DEBUG = 57, SYNC_WORDS = 64 };
constexpr unsigned SYNC_SPIN_LIMIT = 1u << 19;  // about a second

__device__ __forceinline__ unsigned sync_read(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sync_set(unsigned *p, unsigned v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane: wait until sync[word] has reached `target` (modulo 2^32); false when the wait gave up.  A wait that gives up
// says so in sync[SYNC_ABORT], and every wait looks at that word: when the streams do not run beside each other (seen
// once in ~10 runs of one test on ROCm 7.2: a fresh engine whose two streams were served one after the other) the first
// wait to time out ends them all -- the enqueued batch drains at once instead of one time-out per hand-over, three per
// iteration -- and the host re-runs the call without hand-overs (occ_gibbs.hip, fallback_to_plain_path).
__device__ __forceinline__ bool sync_wait(unsigned *sync, int word, unsigned target)
{
    unsigned spins = 0;
    while ((int)(sync_read(sync + word) - target) < 0) {
        __builtin_amdgcn_s_sleep(4);
        ++spins;
        if ((spins & 63u) == 0u && sync_read(sync + SYNC_ABORT) != 0u) return false;
        if (spins > SYNC_SPIN_LIMIT) {
            sync_set(sync + SYNC_ABORT, 1u);
            return false;
        }
    }
    return true;
}
__device__ __forceinline__ double load_agent(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Do the two streams run beside each other?  The hand-overs above presume it.  k_stream_probe_wait (side stream, launched
// first) looks for the word k_stream_probe_set (main stream, launched second) writes: streams that share a hardware queue
// run the two in launch order and the word is never seen; streams whose queues the scheduler time-slices (more live
// queues than the device's 24 hardware slots) take a scheduling quantum -- milliseconds -- to get both kernels onto the
// device (occ_gibbs.hip, stream_probe).
__global__ void __launch_bounds__(64) k_stream_probe_wait(unsigned *w)
{
    if (threadIdx.x != 0) return;
    unsigned seen = 0u, spins = 0u;
    for (; spins < (1u << 13); ++spins) {  // ~ 20 ms
        if (sync_read(w) != 0u) { seen = 1u; break; }
        __builtin_amdgcn_s_sleep(16);
    }
    w[16] = seen;
    w[17] = spins;
}
__global__ void __launch_bounds__(64) k_stream_probe_set(unsigned *w)
{
    if (threadIdx.x == 0) sync_set(w, 1u);
}

// Checksum of a device array (after-broadcast check of a group's fixed arrays: occ_gibbs.hip, array_checksum): the 64-bit
// sum of every 8-byte word (a short tail zero-padded) mixed with its position -- integer adds commute, so the atomics
// return the same value whatever the order; two arrays that differ in any bit, or hold the same words in another order,
// differ in it (up to a 2^-64 accident).
__global__ void __launch_bounds__(256) k_checksum(const unsigned char *__restrict__ data, unsigned long long bytes, unsigned long long *__restrict__ out)
{
    const unsigned long long nw = (bytes + 7ull) / 8ull;
    unsigned long long acc = 0ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long w = 0ull;
        if (8ull * i + 8ull <= bytes) w = *reinterpret_cast<const unsigned long long *>(data + 8ull * i);  // (hipMalloc: 256-byte aligned)
        else
            for (unsigned long long b = 8ull * i; b < bytes; ++b) w |= (unsigned long long)data[b] << (8ull * (b - 8ull * i));
        unsigned long long x = w + 0x9E3779B97F4A7C15ull * (i + 1ull);  // splitmix64 finaliser of (word, position)
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        acc += x ^ (x >> 31);
    }
    atomicAdd(out, acc);
}

// The state a call is re-run from after a device-side wait gave up (occ_gibbs.hip, snapshot_take): eta, z, the warm start
// (and theta of the reduced-rank model) of every chain, in ONE launch -- which also resets the fused kernel's clock words
// and opens the call's window (occ_gibbs.hip, open_window): no copy engine between the call's entry and its first kernel.
__global__ void __launch_bounds__(256) k_snapshot(const double *__restrict__ eta, double *__restrict__ s_eta, const uint8_t *__restrict__ z, uint8_t *__restrict__ s_z,
                                                  const double2 *__restrict__ x, double2 *__restrict__ s_x, unsigned long long count,
                                                  const double *__restrict__ theta, double *__restrict__ s_theta, unsigned long long count_theta,
                                                  unsigned long long *__restrict__ clock,  // k_iter's clock words of the call that follows (or null)
                                                  ChainScalars *__restrict__ scs, int n_chains, int parity, uint32_t n_iter, uint32_t burnin, uint32_t keep)
{
    if (clock != nullptr && blockIdx.x == 0 && threadIdx.x < 4) clock[threadIdx.x] = threadIdx.x == 0 ? ~0ull : 0ull;
    // ... and opens the call's window of iterations in the chains' scalars (what set_window would upload: occ_gibbs.hip)
    if (scs != nullptr && blockIdx.x == 0) {
        for (int ch = (int)threadIdx.x; ch < n_chains; ch += (int)blockDim.x) {
            ChainScalars &sc = scs[ch];
            Ctl &ctl = sc.ctl[parity];
            sc.it_base = ctl.it;
            sc.it_stop = ctl.it + n_iter;
            sc.burnin = burnin;
            sc.keep = keep;
            ctl.koff = 0;
        }
    }
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * blockDim.x) {
        s_eta[i] = eta[i];
        s_z[i] = z[i];
        s_x[i] = x[i];
    }
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count_theta; i += (unsigned long long)gridDim.x * blockDim.x) s_theta[i] = theta[i];
}

// First kernel of a side-stream sequence: one lane announces that the previous k_noise is complete, then waits
// until the main stream has finished the previous sequence.
__global__ void __launch_bounds__(64) k_gate(const Ctx *__restrict__ cp, ChainScalars *__restrict__ scs)
{
    const Ctx &c = *cp;
    if (c.sync == nullptr || threadIdx.x != 0) return;
    const unsigned j = c.sync[SYNC_SIDE_SEQ];
    if (c.sync[SYNC_DEBUG] != 0u) return;  // test knob (OCC_DEBUG_BREAK_HANDOVER): a side stream that never announces its noise
    sync_set(c.sync + SYNC_NOISE, j);
    if (!sync_wait(c.sync, SYNC_MAIN, j)) scs[0].err = -2;
}

// =================================================================================================
#define OCC_KARGS const Ctx *__restrict__ cp, ChainScalars *__restrict__ scs, Slot *__restrict__ slots, int chain_base, int e

// eta_i (Q eta)_i: site i's term of eta'Q eta (logit.py:208)
__device__ __forceinline__ double quad_site(const Ctx &c, const double *eta, int i, double eta_i)
{
    const int lane = i & 63;
    int base, width;
    slice_of(c, i, base, width);
    double qe = c.qdiag[i] * eta_i;
    for (int k = 0; k < width; ++k) qe = fma(c.sell_val[base + k * 64 + lane], eta[c.sell_col[base + k * 64 + lane]], qe);
    return eta_i * qe;
}

// omega_b ~ PG(1, x_i'beta + eta_i) of iteration `it` into omega_b[it & 1], and the partials of eta'Q eta
// (logit.py:195-204, 208).  `blk` is the block index within the role's own grid.
template <int P>
__device__ __forceinline__ void omega_b_body(const Ctx &c, const ChainScalars &sc, const double (&beta)[P], int chain, uint32_t it, int blk,
                                             bool per_wave = false)
{
    const int n = c.n, i = blk * blockDim.x + threadIdx.x;
    double quad[1] = {0.0};
    if (i < n) {
        const size_t ci = (size_t)chain * n + i;
        const double *eta = c.eta + (size_t)chain * n;
        double xb = 0.0;
#pragma unroll
        for (int a = 0; a < P; ++a) xb += c.Xt[(size_t)a * n + i] * beta[a];
        const double eta_i = eta[i];
        c.omega_b[it & 1][ci] = pg1_draw(sc.key, (uint32_t)i, it, STREAM_OMEGA_B, xb + eta_i);
        quad[0] = quad_site(c, eta, i, eta_i);
    }
    if (per_wave) {  // several waves per block, partial sums still per 64-site slice (c.nb_n counts slices)
        const double t = wave_sum(quad[0]);
        const int slice = blk * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
        if ((threadIdx.x & 63) == 0 && slice < c.nb_n) c.part_quad[(size_t)chain * c.nb_n + slice] = t;
        return;
    }
    block_partials<1>(quad, c.part_quad + (size_t)chain * c.nb_n, c.nb_n, blk);
}

// ... with beta taken from the chain's scalars and a run-time number of covariates (generic path)
__device__ __forceinline__ void omega_b_body_g(const Ctx &c, const ChainScalars &sc, int chain, uint32_t it, int blk, bool per_wave = false)
{
    const int n = c.n, i = blk * blockDim.x + threadIdx.x;
    double quad[1] = {0.0};
    if (i < n) {
        const size_t ci = (size_t)chain * n + i;
        const double *eta = c.eta + (size_t)chain * n;
        double xb = 0.0;
        for (int a = 0; a < c.p; ++a) xb += c.Xt[(size_t)a * n + i] * sc.beta[a];
        const double eta_i = eta[i];
        c.omega_b[it & 1][ci] = pg1_draw(sc.key, (uint32_t)i, it, STREAM_OMEGA_B, xb + eta_i);
        quad[0] = quad_site(c, eta, i, eta_i);
    }
    if (per_wave) {
        const double t = wave_sum(quad[0]);
        const int slice = blk * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
        if ((threadIdx.x & 63) == 0 && slice < c.nb_n) c.part_quad[(size_t)chain * c.nb_n + slice] = t;
        return;
    }
    block_partials<1>(quad, c.part_quad + (size_t)chain * c.nb_n, c.nb_n, blk);
}

// Stand-alone omega_b of the CURRENT iteration: only needed when the start values or the state were
// just set by the host (afterwards k_z_ob's second role has already produced it).
template <int P>
__global__ void __launch_bounds__(256) k_omega_b(OCC_KARGS)
{
    const Ctx &c = *cp;
    const Tile tile = tile_of_block(chain_base);
    const int chain = tile.chain, blk = tile.blk;
    const ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop || sc.err != 0) return;
    if constexpr (P == 0) {
        omega_b_body_g(c, sc, chain, ctl.it, blk);
    } else {
        double beta[P];
#pragma unroll
        for (int a = 0; a < P; ++a) beta[a] = sc.beta[a];
        omega_b_body<P>(c, sc, beta, chain, ctl.it, blk);
    }
}

// The partial sums of eta'Q eta alone, from the chain's current eta (occ_cond_tau: no Polya-Gamma draw).
__global__ void __launch_bounds__(256) k_quad(OCC_KARGS)
{
    const Ctx &c = *cp;
    const Tile tile = tile_of_block(chain_base);
    const int chain = tile.chain, blk = tile.blk;
    const int n = c.n, i = blk * blockDim.x + threadIdx.x;
    const double *eta = c.eta + (size_t)chain * n;
    double quad[1] = {0.0};
    if (i < n) quad[0] = quad_site(c, eta, i, eta[i]);
    block_partials<1>(quad, c.part_quad + (size_t)chain * c.nb_n, c.nb_n, blk);
}

// Variates of the eta right-hand side that depend only on (key, iteration): the site normals eps_1
// (logit.py:75-76) and u = B'eps, the edge form of the prior term E (sqrt(tau) eps_2) (logit.py:66-67,
// 77; Q = B'B with B the weighted incidence matrix, so u ~ N(0, Q) like E eps_2).  Written for iteration
// ctl.it + ahead into buffer [(it + ahead) & 1]; runs on the side stream, off the critical path.
__device__ __forceinline__ void noise_site(const Ctx &c, uint64_t key, int chain, int i, uint32_t it)
{
    const int n = c.n, lane = i & 63;
    int base, width;
    slice_of(c, i, base, width);
    if (c.dense_F != nullptr) width = 0;  // reference-form prior draw: k_prior_dense writes uprior
    double u = 0.0;
    for (int k = 0; k < width; ++k) {
        const int j = c.sell_col[base + k * 64 + lane];
        const double w = -c.sell_val[base + k * 64 + lane];
        if (w > 0.0) {
            const uint32_t lo = (uint32_t)min(i, j), hi = (uint32_t)max(i, j);
            const double t = sqrt(w) * block_normal(key, lo, hi, it, STREAM_ETA_EDGE);
            u += (i < j) ? t : -t;
        }
    }
    const size_t ci = (size_t)chain * n + i;
    if (c.dense_F == nullptr) c.uprior[it & 1][ci] = u;
    else if (i < c.dense_m) c.dense_eps[it & 1][(size_t)chain * c.dense_m + i] = block_normal(key, (uint32_t)i, 0, it, STREAM_ETA_DENSE);
    c.enorm[it & 1][ci] = block_normal(key, (uint32_t)i, 0, it, STREAM_ETA_SITE);
}

// Reference-form prior term (logit.py:77): uprior = F eps2 for every chain in ONE pass over F (n x m, row-major: the
// bytes that bound this kernel are read once for all chains).  One wave per row, lanes strided over the columns
// (coalesced 512-byte requests), up to NCH chains accumulated side by side, fixed-order wave sums.  Launched right
// behind k_noise (same control words, same `ahead`).
template <int NCH>
__global__ void __launch_bounds__(256) k_prior_dense(const Ctx *__restrict__ cp, ChainScalars *__restrict__ scs, int chain0, int e, int ahead)
{
    const Ctx &c = *cp;
    const int row = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= c.n) return;
    const int m = c.dense_m;
    uint32_t it[NCH];
    bool on[NCH];
    bool any = false;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        const int ch = chain0 + q;
        on[q] = false;
        it[q] = 0;
        if (ch < c.C) {
            const ChainScalars &sc = scs[ch];
            const Ctl ctl = sc.ctl[e];
            on[q] = !(ctl.koff || ctl.it >= sc.it_stop || sc.err != 0);
            it[q] = ctl.it + (uint32_t)ahead;
        }
        any = any || on[q];
    }
    if (!any) return;
    const double *F = c.dense_F + (size_t)row * m;
    double acc[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) acc[q] = 0.0;
    for (int j = lane; j < m; j += 64) {
        const double f = F[j];
#pragma unroll
        for (int q = 0; q < NCH; ++q)
            if (on[q]) acc[q] = fma(f, c.dense_eps[it[q] & 1][(size_t)(chain0 + q) * m + j], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        const double t = wave_sum(acc[q]);
        if (on[q] && lane == 0) c.uprior[it[q] & 1][(size_t)(chain0 + q) * c.n + row] = t;
    }
}

// k_noise also draws alpha ~ N(A^-1 r, A^-1) of THIS iteration (logit.py:224) from the partial sums of k_omega_a, the previous
// kernel of the sequence: the first block of every chain, before its own tile of noise (ahead = 1 only: the stand-alone
// launch after new start values draws no alpha).  As a kernel of its own between k_omega_a and k_noise (k_alpha_draw,
// still used by occ_cond_alpha) the draw was 7 us of four busy workgroups plus a launch boundary on the side stream --
// which bounds the iteration once the solves get short (K ~ 6 late in a run: 56.9 -> 50 us per iteration; the first
// iterations of a run likewise).  With device-side hand-overs the drawing block publishes alpha itself (agent-scope
// stores, then one count per chain in sync[SYNC_ALPHA]; k_z_ob waits for C counts per sequence).
__global__ void __launch_bounds__(256) k_noise(OCC_KARGS, int ahead, int sync_on)
{
    __shared__ double s_red[NACC_G], s_U[MAXG * MAXG], s_work[2 * MAXG], s_alpha[MAXG];
    const Ctx &c = *cp;
    const Tile tile = tile_of_block_shared(c, 2, (c.n + (int)blockDim.x - 1) / (int)blockDim.x, chain_base);
    const int chain = tile.chain, blk = tile.blk;
    const bool synced = sync_on && c.sync;
    if (synced && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        const unsigned j = c.sync[SYNC_SIDE_SEQ];
        c.sync[SYNC_SIDE_SEQ] = j + 1u;         // read next by k_gate, the next kernel of the stream
    }
    if (chain < 0) return;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.ctl[e];
    const bool idle = ctl.koff || ctl.it >= sc.it_stop || sc.err != 0;  // (uniform over the chain's blocks)
    if (ahead == 1 && blk == 0) {
        if (!idle) {
            const int Q = c.q;
            reduce_partials_lds(c.part_alpha + (size_t)chain * nacc(Q) * c.nb_r, nacc(Q), c.nb_r, s_red);
            if (threadIdx.x == 0) {
                const double *a_prec = c.hyp, *a_pbm = c.hyp + Q * Q;
                const bool ok = precision_mvnorm_dev(Q, s_red, a_prec, a_pbm, sc.key, ctl.it, STREAM_ALPHA, s_U, s_work, s_alpha, nullptr);
                if (!ok) sc.err = -4;  // OCC_E_CHOLESKY
                for (int a = 0; a < Q; ++a) __hip_atomic_store(&sc.alpha[a], s_alpha[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (synced && threadIdx.x == 0) {  // (an idle chain counts too: k_z_ob waits for every chain of the sequence)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(c.sync + SYNC_ALPHA, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (idle) return;
    const uint32_t it_for = ctl.it + (uint32_t)ahead;
    if (blk == 0 && threadIdx.x == 0) {  // tau's standard gamma variate of that iteration (the rate comes later)
        Cursor g(sc.key, 0u, it_for, STREAM_TAU);
        sc.tau_gamma[it_for & 1] = std_gamma(g, c.tau_shape);
    }
    const int i = blk * blockDim.x + threadIdx.x;
    if (i >= c.n) return;
    noise_site(c, sc.key, chain, i, it_for);
}

template <int INJ>
__global__ void __launch_bounds__(256) k_eta_init(OCC_KARGS)
{
    __builtin_amdgcn_s_setprio(3);  // critical path (see k_minres)
    const Ctx &c = *cp;
    const Tile tile = tile_of_block(chain_base);
    const int chain = tile.chain, blk = tile.blk;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (ctl.koff || ctl.it >= sc.it_stop || sc.err != 0) return;
    const int n = c.n, i = blk * blockDim.x + threadIdx.x;
    const uint32_t it = ctl.it;
    double quad[1];
    reduce_partials<1>(c.part_quad + (size_t)chain * c.nb_n, c.nb_n, quad);
    // every lane draws the same tau from the same sub-stream (uniform control flow, no broadcast)
    const double rate = 0.5 * quad[0] + c.tau_rate;
    double tau;
    if (INJ) {  // the caller's gamma variate (rng.gamma(shape, 1 / rate) = standard_gamma(shape) * (1 / rate)), or the chain's tau as it is
        tau = c.inj->tau_from_gamma ? (1.0 / rate) * c.inj->gamma : sc.tau;
    } else {
        tau = (1.0 / rate) * sc.tau_gamma[it & 1];  // the variate k_noise drew for this iteration
    }
    if (blk == 0 && threadIdx.x == 0) {
        sc.tau = tau;
        Slot s = {};
        slot_store(&slots[(size_t)chain * NSLOT], s);
    }
    const double st = sqrt(tau);
    if (i < n) {
        const size_t ci = (size_t)chain * n + i;
        const double2 *X0 = c.Xv + (size_t)chain * n;
        const double om = c.omega_b[it & 1][ci];
        const double xb = xdot(c.Xt, n, i, sc.beta, c.p);
        const double y = eta_rhs_site(om, xb, (double)c.z[ci], c.enorm[it & 1][ci], c.uprior[it & 1][ci], st);
        c.rhs[ci] = y;
        const double2 x0 = X0[i];
        const double d = tau * c.qdiag[i] + om;
        double ax = d * x0.x, az = d * x0.y;
        const int lane = i & 63;
        int base, width;
        slice_of(c, i, base, width);
        for (int k = 0; k < width; ++k) {
            const int j = c.sell_col[base + k * 64 + lane];
            const double a = tau * c.sell_val[base + k * 64 + lane];
            const double2 xj = X0[j];
            ax = fma(a, xj.x, ax);
            az = fma(a, xj.y, az);
        }
        double2 r;
        r.x = y - ax;
        r.y = 1.0 - az;
        c.Pv[0][ci] = r;
    }
}

// Sum-to-zero projection partials, taken by the kernel that detects the end of the solve.
template <class A>
__device__ __forceinline__ void projection_partials(const A &c, int chain, int i, int blk)
{
    double v[2] = {0.0, 0.0};
    if (i < c.n) {
        const double2 x = c.Xv[(size_t)chain * c.n + i];
        v[0] = x.x;
        v[1] = x.y;
    }
    block_partials<2>(v, c.part_proj + (size_t)chain * 2 * c.nb_n, c.nb_n, blk);
}

// Diagnostic builds (tools/kbench.hip) define OCC_STAMP to record s_memtime at a few points of
// k_minres; in the product build it expands to nothing.
#ifndef OCC_STAMP
#define OCC_STAMP(n)
#endif


// ---- MINRES: the arithmetic shared by the launch-per-iteration kernel (k_minres) and the persistent
// iteration kernel (k_iter, occ_iter.hpp).  Both kernels run the SAME scalar recurrence and the same
// explicitly contracted vector expressions, so a solve gives the same bits whichever of them ran it.
struct KryStep {
    double ca, cb, cc;                     // p_{k-1} = ca g_{k-1} - cb p_{k-3} - cc p_{k-2}
    double sj, oldeps, delta, denom, phi;  // rotation of iteration k-2: w = (sj p_{k-3} - oldeps w_{k-4} - delta w_{k-3}) denom
    bool rotate;                           // k >= 3: w_{k-2}, x_{k-2} are formed in this step
    bool stop;                             // the solve ended BEFORE this step (slot final: take the projection sums)
};

// Step k of a solve (k = 1, 2, ...), given the slot after step k-1 and the sums of step k-1:
// S0 = ||p_{k-2}||^2, S1 = p_{k-2}.g_{k-1}, S2 = p_{k-2}.p_{k-3}, xn2 = ||x_{k-3}||^2.
//   (a) k >= 4: stopping test of iteration k-3, exactly scipy's (minres.py, "Estimate various norms")
//   (b) k >= 2: beta_{k-1}, alfa_{k-1}
//   (c) k >= 3: rotation of iteration k-2 (needs beta_{k-1})
// The step is split in two so that a kernel can take the part that only needs the slot (minres_pre: two of
// the square roots and half of the divisions) off the critical path -- k_iter runs it while it waits for the
// other workgroups' sums.  Contractions are explicit: the split does not change a bit of the result.
struct KryPre {
    double Anorm, rn2, b1sq;               // (a): Anorm^2, rnorm^2, beta1^2 and the tests that do not need ||x||
    bool root_tiny, root_small, ill_cond, no_norm;
    double t_ab, delta, gbar, gbar2, sj;   // (c): t_ab = alfa_j^2 + beta_j^2
};
// 1 / sqrt(x) for x > 0 away from the ends of the exponent range: v_rsq_f64 (2^-24 relative, measured on gfx950) and two
// Newton steps (1.2 ulp against a long-double reference): 9 dependent f64 instructions where sqrt followed by a
// division is 28.
__device__ __forceinline__ double rsqrt_nr(double x)
{
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double h = 0.5 * y;
        const double err = fma(-(x * y), h, 0.5);  // (1 - x y^2) / 2
        y = fma(y, err, y);
    }
    return y;
}

// The two halves of minres_pre: what the stopping test (a) needs, what the rotation (c) needs.
__device__ __forceinline__ void minres_pre_a(const Slot &s, KryPre &q)
{
    // scipy: test1 = rnorm / (Anorm ynorm), test2 = root / Anorm, Acond = gmax / gmin, epsx = Anorm ynorm eps, then
    // `1 + test <= 1`, `test <= rtol`, `Acond >= 0.1 / eps`, `epsx >= beta1`.  The same decisions on squares and
    // products -- no square root and no division (they were 70 of the ~200 issue-bound f64 instructions of a step):
    // for non-negative x, c, d:  x / d <= c  <=>  x^2 <= c^2 d^2 (d > 0),  and  1 + x <= 1  <=>  x <= 2^-53.
    // Here: everything the slot alone decides (the slot keeps Anorm^2 and root^2); minres_post_a adds ||x||.
    const double eps = DBL_EPSILON, rtol = 1e-5;
    const double tn2 = s.tnorm2, root2 = s.root, tiny2 = 0x1p-106, rtol2 = rtol * rtol;
    q.Anorm = tn2;
    q.no_norm = tn2 == 0.0;  // scipy's tests are inf there
    q.root_tiny = !q.no_norm && root2 <= tiny2 * tn2;
    q.root_small = !q.no_norm && root2 <= rtol2 * tn2;
    q.ill_cond = s.gmax >= (0.1 / eps) * s.gmin;
    q.rn2 = s.phibar * s.phibar;  // rnorm^2 (phibar >= 0)
    q.b1sq = s.beta1 * s.beta1;
}
__device__ __forceinline__ void minres_pre_c(const Slot &s, KryPre &q)
{
    q.t_ab = fma(s.beta, s.beta, s.alfa * s.alfa);
    q.delta = fma(s.sn, s.alfa, s.cs * s.dbar);
    q.gbar = fma(-s.cs, s.alfa, s.sn * s.dbar);
    q.gbar2 = q.gbar * q.gbar;
    q.sj = s.ibeta;  // = 1 / s.beta, bit for bit: the quotient the step that formed s.beta took (its `ca`)
}
__device__ __forceinline__ KryPre minres_pre(const Slot &s)
{
    KryPre q;  // entries that the coming step does not use may be inf / nan (early steps): they are never read
    minres_pre_a(s, q);
    minres_pre_c(s, q);
    return q;
}
// (a): scipy's stopping test of iteration k - 3.  True: the solve ended before step k (istop, itn, done set).
// Reads phibar, beta1, istop of the slot and nothing (b) / (c) write before it in the same step.
__device__ __forceinline__ bool minres_post_a(Slot &s, const KryPre &q, int k, double xn2, long long maxiter)
{
    if (k < 4) return false;
    const double eps = DBL_EPSILON, rtol = 1e-5;
    const int j = k - 3;
    const double ay2 = q.Anorm * xn2;                     // (Anorm ynorm)^2
    const bool inf1 = (xn2 == 0.0 || q.no_norm);
    const double tiny2 = 0x1p-106, rtol2 = rtol * rtol;
    int istop = s.istop;
    if (istop == 0) {  // scipy's order: the last test that holds names the reason
        if (q.root_tiny) istop = 2;
        if (!inf1 && q.rn2 <= tiny2 * ay2) istop = 1;
        if ((long long)j >= maxiter) istop = 6;
        if (q.ill_cond) istop = 4;
        if (ay2 * (eps * eps) >= q.b1sq) istop = 3;
        if (q.root_small) istop = 2;
        if (!inf1 && q.rn2 <= rtol2 * ay2) istop = 1;
    }
    if (istop != 0) {
        s.istop = istop; s.itn = j; s.done = 1;
        return true;
    }
    return false;
}
// (a) + (b): the stopping test, beta_{k-1}, alfa_{k-1} -- `stop` and the coefficients of p_{k-1}, all that the vectors'
// first half of step k (p_{k-1} at the site and its neighbours, g_k = A p_{k-1}) needs.  (c), the rotation of iteration
// k - 2 and the slot's update, follows in minres_post_c: k_iter's scalar wave runs it while the site waves are busy with
// that first half (occ_iter.hpp).  Called one after the other (minres_post) they are the unsplit step: the same
// operations on the same operands.
struct KryMid {
    double beta_km1, beta_km2, alfa_km1;
};
// (b) alone: it reads the slot (beta_{k-2}) and writes nothing to it unless the solve ends here (k == 2, beta1 == 0), so it
// may run before (a) -- k_iter's scalar wave hands the coefficients over first and evaluates the stopping test while the
// site waves already use them (a step too many when the test says stop: its results are dropped).
__device__ __forceinline__ KryStep minres_post_b(Slot &s, const KryPre &q, int k, double S0, double S1, double S2, KryMid &mid)
{
    KryStep st;
    st.ca = st.cb = st.cc = 0.0;
    st.sj = st.oldeps = st.delta = st.denom = st.phi = 0.0;
    st.rotate = false;
    st.stop = false;
    mid.beta_km1 = mid.beta_km2 = mid.alfa_km1 = 0.0;
    if (k >= 2) {
        if (k == 2 && S0 == 0.0) {  // beta1 == 0: x0 already solves the system (minres.py)
            s.done = 1; s.istop = 0; s.itn = 0;
            st.stop = true;
            return st;
        }
        // beta_{k-1} = sqrt(S0) and its reciprocal from ONE reciprocal square root (the step is issue-bound on one SIMD:
        // sqrt + two divisions were 40 dependent instructions, this is 12)
        const double ibeta = rsqrt_nr(S0);
        mid.beta_km1 = S0 * ibeta;         // beta_{k-1}
        mid.beta_km2 = s.beta;             // beta_{k-2} (k >= 3)
        double alfa_km1 = (S1 * ibeta) * ibeta;  // (p.g)/beta^2
        if (k >= 3) alfa_km1 = alfa_km1 - S2 * q.sj;
        mid.alfa_km1 = alfa_km1;
        st.ca = ibeta;
        st.cc = alfa_km1 * st.ca;
        if (k >= 3) st.cb = mid.beta_km1 * q.sj;
    }
    return st;
}
__device__ __forceinline__ KryStep minres_post_ab(Slot &s, const KryPre &q, int k, double S0, double S1, double S2, double xn2, long long maxiter,
                                                  KryMid &mid)
{
    if (minres_post_a(s, q, k, xn2, maxiter)) {
        KryStep st;
        st.ca = st.cb = st.cc = 0.0;
        st.sj = st.oldeps = st.delta = st.denom = st.phi = 0.0;
        st.rotate = false;
        st.stop = true;
        mid.beta_km1 = mid.beta_km2 = mid.alfa_km1 = 0.0;
        return st;
    }
    return minres_post_b(s, q, k, S0, S1, S2, mid);
}
// The part of (c) the vectors wait for -- the rotation's coefficients -- ahead of the rest: the same operations on the same
// operands as minres_post_c below (which forms them again, bit for bit), nothing written to the slot.
__device__ __forceinline__ void minres_rotation(const Slot &s, const KryPre &q, int k, KryStep &st, const KryMid &mid)
{
    if (k < 3) return;
    const double eps = DBL_EPSILON;
    const double beta_n = mid.beta_km1;
    st.oldeps = s.epsln;
    st.delta = q.delta;
    const double g2 = fma(beta_n, beta_n, q.gbar2);
    const bool g_ok = g2 >= eps * eps;
    const double ig = rsqrt_nr(g_ok ? g2 : 1.0);
    st.denom = g_ok ? ig : 1.0 / eps;
    st.phi = (q.gbar * st.denom) * s.phibar;
    st.sj = q.sj;
    st.rotate = true;
}
// (c): the rotation of iteration k - 2 (k >= 3) and the slot after step k
__device__ __forceinline__ void minres_post_c(Slot &s, const KryPre &q, int k, KryStep &st, const KryMid &mid)
{
    const double eps = DBL_EPSILON;
    if (k >= 2) {
        const double beta_km1 = mid.beta_km1;
        if (k == 2) {
            s.beta1 = beta_km1; s.oldb = 0.0; s.dbar = 0.0; s.epsln = 0.0; s.phibar = beta_km1;
            s.rhs1 = beta_km1; s.rhs2 = 0.0; s.tnorm2 = 0.0; s.gmax = 0.0; s.gmin = DBL_MAX;
            s.cs = -1.0; s.sn = 0.0; s.root = 0.0; s.istop = 0;
        } else {  // rotation of iteration j = k-2 with alfa_j (slot), beta_j (slot), beta_{j+1} (new)
            const int j = k - 2;
            const double beta_j = mid.beta_km2, beta_n = beta_km1;
            s.oldb = beta_j;
            s.tnorm2 += fma(beta_n, beta_n, q.t_ab);
            if (j == 1 && beta_n / s.beta1 <= 10.0 * eps) s.istop = -1;
            st.oldeps = s.epsln;
            st.delta = q.delta;
            s.epsln = s.sn * beta_n;
            s.dbar = -s.cs * beta_n;
            s.root = fma(s.dbar, s.dbar, q.gbar2);  // root^2: only the stopping test reads it, as a square
            const double g2 = fma(beta_n, beta_n, q.gbar2);
            const bool g_ok = g2 >= eps * eps;            // scipy: gamma = max(gamma, eps)
            const double ig = rsqrt_nr(g_ok ? g2 : 1.0);
            const double gamma = g_ok ? g2 * ig : eps;
            st.denom = g_ok ? ig : 1.0 / eps;  // 1 / gamma; products with it wherever scipy divides by gamma
            s.cs = q.gbar * st.denom;
            s.sn = beta_n * st.denom;
            st.phi = s.cs * s.phibar;
            s.phibar = s.sn * s.phibar;
            s.gmax = fmax(s.gmax, gamma);
            s.gmin = fmin(s.gmin, gamma);
            const double zz = s.rhs1 * st.denom;
            s.rhs1 = fma(-st.delta, zz, s.rhs2);
            s.rhs2 = -s.epsln * zz;
            st.sj = q.sj;
            st.rotate = true;
        }
        s.beta = beta_km1;
        s.ibeta = st.ca;
        s.alfa = mid.alfa_km1;
    }
    s.itn = k;
}
__device__ __forceinline__ KryStep minres_post(Slot &s, const KryPre &q, int k, double S0, double S1, double S2, double xn2, long long maxiter)
{
    KryMid mid;
    KryStep st = minres_post_ab(s, q, k, S0, S1, S2, xn2, maxiter, mid);
    if (!st.stop) minres_post_c(s, q, k, st, mid);
    return st;
}
__device__ __forceinline__ KryStep minres_scalars(Slot &s, int k, double S0, double S1, double S2, double xn2, long long maxiter)
{
    const KryPre q = minres_pre(s);
    return minres_post(s, q, k, S0, S1, S2, xn2, maxiter);
}

// The same step in SCIPY'S OWN ARITHMETIC -- a square root and a division wherever minres.py has one (beta = sqrt(.),
// alfa / beta, beta / oldb, gbar / gamma, rhs1 / gamma, rnorm / (Anorm ynorm), root / Anorm, gmax / gmin), nothing
// carried as a square or a reciprocal -- for the debug instantiation k_minres<1> only (OCC_DEBUG_EXACT_DIV=1 in
// occ_cond_eta): the production step above replaces those by one reciprocal square root per divisor and products, which
// moves the iterate by ~1e-8; this form stays within 1e-9 of the reference's recorded solves, so that the production
// tolerance cannot hide an indexing or ordering error in the vector part (tests/test_gpu_golden.py).  A solve runs
// entirely in one form (the slot's `root` is a root here, a square there; `ibeta` is not used here).
__device__ __forceinline__ KryStep minres_scalars_exact(Slot &s, int k, double S0, double S1, double S2, double xn2, long long maxiter)
{
    const double eps = DBL_EPSILON, rtol = 1e-5;
    KryStep st;
    st.ca = st.cb = st.cc = 0.0;
    st.sj = st.oldeps = st.delta = st.denom = st.phi = 0.0;
    st.rotate = false;
    st.stop = false;
    if (k >= 4) {  // (a) stopping test of iteration k - 3 (minres.py, "Estimate various norms")
        const int j = k - 3;
        const double Anorm = sqrt(s.tnorm2), ynorm = sqrt(xn2);
        const double epsx = Anorm * ynorm * eps, rnorm = s.phibar;
        const double test1 = (ynorm == 0.0 || Anorm == 0.0) ? INFINITY : rnorm / (Anorm * ynorm);
        const double test2 = (Anorm == 0.0) ? INFINITY : s.root / Anorm;
        const double Acond = s.gmax / s.gmin;
        int istop = s.istop;
        if (istop == 0) {
            const double t1 = 1.0 + test1, t2 = 1.0 + test2;
            if (t2 <= 1.0) istop = 2;
            if (t1 <= 1.0) istop = 1;
            if ((long long)j >= maxiter) istop = 6;
            if (Acond >= 0.1 / eps) istop = 4;
            if (epsx >= s.beta1) istop = 3;
            if (test2 <= rtol) istop = 2;
            if (test1 <= rtol) istop = 1;
        }
        if (istop != 0) {
            s.istop = istop; s.itn = j; s.done = 1;
            st.stop = true;
            return st;
        }
    }
    if (k >= 2) {  // (b) beta_{k-1}, alfa_{k-1}
        if (k == 2 && S0 == 0.0) {
            s.done = 1; s.istop = 0; s.itn = 0;
            st.stop = true;
            return st;
        }
        const double beta_km1 = sqrt(S0), beta_km2 = s.beta;
        double alfa_km1 = S1 / S0;
        if (k >= 3) alfa_km1 = alfa_km1 - S2 / beta_km2;
        if (k == 2) {
            s.beta1 = beta_km1; s.oldb = 0.0; s.dbar = 0.0; s.epsln = 0.0; s.phibar = beta_km1;
            s.rhs1 = beta_km1; s.rhs2 = 0.0; s.tnorm2 = 0.0; s.gmax = 0.0; s.gmin = DBL_MAX;
            s.cs = -1.0; s.sn = 0.0; s.root = 0.0; s.istop = 0;
        } else {  // (c) rotation of iteration k - 2
            const double beta_n = beta_km1;
            const double delta = fma(s.sn, s.alfa, s.cs * s.dbar), gbar = fma(-s.cs, s.alfa, s.sn * s.dbar);
            s.oldb = beta_km2;
            s.tnorm2 += fma(beta_n, beta_n, fma(beta_km2, beta_km2, s.alfa * s.alfa));
            if (k == 3 && beta_n / s.beta1 <= 10.0 * eps) s.istop = -1;
            st.oldeps = s.epsln;
            st.delta = delta;
            s.epsln = s.sn * beta_n;
            s.dbar = -s.cs * beta_n;
            s.root = sqrt(fma(s.dbar, s.dbar, gbar * gbar));
            const double gamma = fmax(sqrt(fma(beta_n, beta_n, gbar * gbar)), eps);
            s.cs = gbar / gamma;
            s.sn = beta_n / gamma;
            st.phi = s.cs * s.phibar;
            s.phibar = s.sn * s.phibar;
            st.denom = 1.0 / gamma;
            s.gmax = fmax(s.gmax, gamma);
            s.gmin = fmin(s.gmin, gamma);
            const double zz = s.rhs1 / gamma;
            s.rhs1 = fma(-delta, zz, s.rhs2);
            s.rhs2 = -s.epsln * zz;
            st.sj = 1.0 / beta_km2;
            st.rotate = true;
            st.cb = beta_km1 / beta_km2;
        }
        st.ca = 1.0 / beta_km1;
        st.cc = alfa_km1 / beta_km1;
        s.beta = beta_km1;
        s.ibeta = st.ca;
        s.alfa = alfa_km1;
    }
    s.itn = k;
    return st;
}

// p_{k-1} = ca g_{k-1} - cb p_{k-3} - cc p_{k-2}, contracted the same way wherever it is formed
__device__ __forceinline__ double2 kry_form_p(const KryStep &st, double2 g, double2 p3, double2 p2)
{
    double2 p;
    p.x = fma(-st.cc, p2.x, fma(-st.cb, p3.x, st.ca * g.x));
    p.y = fma(-st.cc, p2.y, fma(-st.cb, p3.y, st.ca * g.y));
    return p;
}
// w_j = (v_j - oldeps w_{j-2} - delta w_{j-1}) / gamma with v_j = p_{j-1} / beta_j;  x_j = x_{j-1} + phi w_j
__device__ __forceinline__ double2 kry_form_w(const KryStep &st, double2 p3, double2 w1, double2 w2)
{
    double2 w;
    w.x = fma(-st.delta, w2.x, fma(-st.oldeps, w1.x, st.sj * p3.x)) * st.denom;
    w.y = fma(-st.delta, w2.y, fma(-st.oldeps, w1.y, st.sj * p3.y)) * st.denom;
    return w;
}
__device__ __forceinline__ double dot2(double2 a, double2 b) { return fma(a.y, b.y, a.x * b.x); }
// h = A v at one site: d v_i + sum over the NW neighbour slots, in slot order (unused slots carry the coefficient 0)
template <int NW>
__device__ __forceinline__ double2 kry_apply(double d, const double (&av)[NW], double2 v, const double2 (&nv)[NW])
{
    double hx = d * v.x, hy = d * v.y;
#pragma unroll
    for (int kk = 0; kk < NW; ++kk) {
        hx = fma(av[kk], nv[kk].x, hx);
        hy = fma(av[kk], nv[kk].y, hy);
    }
    return make_double2(hx, hy);
}

constexpr int NPRE = 8;  // neighbour slots fetched before the scalars are known (queen lattice: all)

// One MINRES iteration per launch, pipelined so that every inner product is a DIRECT sum (no
// algebraic shortcut): with p_m = r2_m (scipy's Lanczos residual after iteration m, p_0 = b - A x0) and
// g_m = A p_{m-1} (unscaled), scipy's iteration m reads
//     beta_m = ||p_{m-1}||,  v_m = p_{m-1}/beta_m,
//     alfa_m = v_m . (A v_m - (beta_m/beta_{m-1}) p_{m-2}) = (p_{m-1}.g_m)/beta_m^2 - (p_{m-1}.p_{m-2})/beta_{m-1}
//     p_m    = g_m/beta_m - (beta_m/beta_{m-1}) p_{m-2} - (alfa_m/beta_m) p_{m-1}.
// Launch k (k = 1, 2, ...) of a solve:
//   prologue  slot of launch k-1 + its sums {||p_{k-2}||^2, p_{k-2}.g_{k-1}, p_{k-2}.p_{k-3}, ||x_{k-3}||^2}
//     (a) k >= 4: stopping test of iteration k-3, exactly scipy's            -> done: projection partials
//     (b) k >= 2: beta_{k-1}, alfa_{k-1}
//     (c) k >= 3: rotation of iteration k-2 (needs beta_{k-1})  -> w_{k-2}, x_{k-2}
//   vectors   p_{k-1} = g_{k-1}/beta_{k-1} - (beta_{k-1}/beta_{k-2}) p_{k-3} - (alfa_{k-1}/beta_{k-1}) p_{k-2}
//             at the site (stored) and at its neighbours (recomputed from three gathered vectors: this
//             replaces the grid-wide synchronisation between "form p" and "apply A to p");
//             g_k = A p_{k-1}
//   sums      ||p_{k-1}||^2, p_{k-1}.g_k, p_{k-1}.p_{k-2}, ||x_{k-2}||^2
// Iteration j is therefore tested by launch j+3.  All vector loads are issued before the
// partial-sum reduction so that their latency overlaps it.
// Everything k_minres needs to form its addresses, BY VALUE in the kernel argument block: with the
// pointers in device memory every launch paid one more dependent (cache-cold) load level.
struct KryArgs {
    int group_T, group_B;       // group_B > 0: k_tiles' order of the four sums of a step (occ_tiles.hpp): groups of group_T consecutive
                                // blocks added in block order, bands of group_B consecutive groups (one group per lane, a wave
                                // sum), the eight bands added in band order; group_B = 0: block by block, lanes strided
    int n, nb_n, ell_w, dia_n;  // dia_n > 0: the off-diagonals lie on dia_n <= NPRE diagonals with one value each (any
                                // unweighted lattice): column = row + dia_off[k] where bit k of dia_mask[row] is set --
                                // one byte per row instead of 12 bytes per stored slot
    int dia_off[8];
    double dia_val[8];
    const uint8_t *dia_mask;
    long long maxiter;
    const int *sell_ptr, *sell_col;
    const double *sell_val, *qdiag;
    const double *omega_b[2];
    double2 *Gv[2], *Pv[3], *Wv[2], *Xv;
    double *part_kry, *part_proj;
    ChainScalars *scs;
    Slot *slots;
};

// Buffers, slot and partial-sum parity are indexed by the LAUNCH number k_launch (a kernel argument),
// not by the step number of the solve, so that no address depends on a value loaded from memory: the
// dependent chain of a launch is  kernel arguments -> {control word, slot, partial sums, neighbour
// indices} -> {own vectors, neighbour gathers}.  A solve carried into the next launch sequence is
// re-aligned to launch number 1 by k_beta_partial (realign_carried_solve).
template <int EXACT>  // 1: the scalar step in scipy's own arithmetic (minres_scalars_exact; debug, occ_cond_eta only)
__global__ void __launch_bounds__(256) k_minres(const KryArgs a, int chain_base, int e, int k_launch)
{
    OCC_STAMP(0)
    __builtin_amdgcn_s_setprio(3);  // critical path: issue ahead of co-resident Polya-Gamma waves of the side stream
    const Tile tile = tile_of_block(chain_base);
    const int chain = tile.chain, blk = tile.blk;
    const ChainScalars &sc = a.scs[chain];
    const int n = a.n, i = blk * blockDim.x + threadIdx.x;
    const bool act = i < n;
    const size_t co = (size_t)chain * n;
    const double2 zero2 = make_double2(0.0, 0.0);
    const int kl = k_launch;
    // ---- level 1: everything whose address is known from the kernel arguments --------------------
    const Ctl ctl = sc.ctl[e];
    const unsigned it_stop = sc.it_stop;
    const int chain_error = sc.err;  // a chain whose error word is set idles (occ_iter.hpp, chain_fail)
    const double tau = sc.tau;
    Slot s = slot_load(&a.slots[(size_t)chain * NSLOT + ((kl - 1) & (NSLOT - 1))]);
    Slot *out = &a.slots[(size_t)chain * NSLOT + (kl & (NSLOT - 1))];
    // partial sums of the previous launch: issued now, reduced after the vector loads are in flight
    double S[4] = {0.0, 0.0, 0.0, 0.0};
    if (blockDim.x == 64 || threadIdx.x < 64) {
        const double *part = a.part_kry + ((size_t)chain * 2 + (kl & 1)) * ((size_t)4 * a.nb_n);
        const int ln = threadIdx.x & 63;
        if (blockDim.x == 64) {
            // 64-site slices (the sizes the fused kernel also runs): the canonical order (sum_slices_canonical); the totals are
            // uniform already, S0..xn2 below take them as they are
            sum_slices_canonical(a.nb_n, ln, S, [&](int slice, double (&v)[4]) {
                const int bc = min(slice, a.nb_n - 1);
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) {
                    const double t = part[qi * a.nb_n + bc];
                    v[qi] = (slice < a.nb_n) ? t : 0.0;
                }
            });
        } else if (a.group_B > 0) {
            // k_tiles' order (see KryArgs::group_B); the totals are uniform when this is done: S0..xn2 below take them as they are
            const int gT = a.group_T, gB = a.group_B, ngrp = (a.nb_n + gT - 1) / gT;
            for (int x = 0; x < 8; ++x) {
                const int g = x * gB + ln;
                double v[4] = {0.0, 0.0, 0.0, 0.0};
                const bool in = ln < gB && g < ngrp;
                for (int t = 0; t < gT; ++t) {
                    const int b = g * gT + t;
                    const int bc = min(max(b, 0), a.nb_n - 1);
#pragma unroll
                    for (int qi = 0; qi < 4; ++qi) {
                        const double tv = part[qi * a.nb_n + bc];
                        v[qi] += (in && b < a.nb_n) ? tv : 0.0;
                    }
                }
                wave_sum4(v);
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) S[qi] += v[qi];
            }
        } else
        // four rounds of loads in flight (a plain "load, add" loop waits for every round trip in turn: 15 of them at
        // 500x500); the sums are accumulated in the same order, rounds past the end add an exact 0
        for (int b0 = ln; b0 < a.nb_n; b0 += 256) {
            double v[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = b0 + 64 * r;
                const int bc = min(b, a.nb_n - 1);
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) {
                    const double t = part[qi * a.nb_n + bc];
                    v[r][qi] = (b < a.nb_n) ? t : 0.0;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) S[qi] += v[r][qi];
            }
        }
    }
    int width = 0, base = 0;
    const int lane = i & 63;
    int col[NPRE];
    double val[NPRE];
    double qd = 0.0, om0 = 0.0, om1 = 0.0;
    if (act) {
        const int slice = i >> 6;
        if (a.dia_n > 0) {
            width = a.dia_n;
            const unsigned m = a.dia_mask[i];
#pragma unroll
            for (int kk = 0; kk < NPRE; ++kk) {
                const bool on = kk < width && ((m >> kk) & 1u);
                col[kk] = on ? i + a.dia_off[kk] : i;
                val[kk] = on ? a.dia_val[kk] : 0.0;
            }
        } else {
            if (a.ell_w > 0) { width = a.ell_w; base = slice * a.ell_w * 64; }
            else { base = a.sell_ptr[slice]; width = (a.sell_ptr[slice + 1] - base) >> 6; }
#pragma unroll
            for (int kk = 0; kk < NPRE; ++kk) {
                col[kk] = i; val[kk] = 0.0;
                if (kk < width) {
                    col[kk] = a.sell_col[base + kk * 64 + lane];
                    val[kk] = a.sell_val[base + kk * 64 + lane];
                }
            }
        }
        qd = a.qdiag[i];
        om0 = a.omega_b[0][co + i];
        om1 = a.omega_b[1][co + i];
    }
    // ---- level 2: own vectors and neighbour gathers (addresses from k_launch and col[]) ----------
    // p_m lives in Pv[m' % 3], g_m in Gv[m' & 1], w_m in Wv[m' & 1] with m' counted in launches
    const double2 *G1 = a.Gv[(kl - 1) & 1] + co;                  // g_{k-1}
    const double2 *P2 = a.Pv[(kl + 1) % 3] + co;                  // p_{k-2}
    const double2 *P3 = a.Pv[kl % 3] + co;                        // p_{k-3}
    double2 *Pw = a.Pv[(kl + 2) % 3] + co, *Gw = a.Gv[kl & 1] + co;   // p_{k-1}, g_k
    double2 *Ww = a.Wv[kl & 1] + co;                              // holds w_{k-4}, receives w_{k-2}
    const double2 *Wr = a.Wv[(kl - 1) & 1] + co;                  // w_{k-3}
    double2 g1_i = zero2, g2_i = zero2, p2_i = zero2, p3_i = zero2, w1 = zero2, w2 = zero2, x = zero2;
    double2 ng[NPRE];
    if (act) {
        g1_i = G1[i]; g2_i = Gw[i]; p2_i = P2[i]; p3_i = P3[i]; x = a.Xv[co + i]; w1 = Ww[i]; w2 = Wr[i];
#pragma unroll
        for (int kk = 0; kk < NPRE; ++kk) {
            ng[kk] = zero2;
            if (kk < width) ng[kk] = G1[col[kk]];
        }
    }
    if (ctl.it >= it_stop || chain_error != 0) return;
    const bool writer = (blk == 0 && threadIdx.x == 0);
    if (s.done) {
        if (writer) slot_store(out, s);
        return;
    }
    const int k = kl + (int)ctl.koff;  // step within THIS solve (continues across launch sequences)
    const double om = (ctl.it & 1) ? om1 : om0;
    // speculative loads of vectors that do not exist yet at the first steps are discarded
    // (k = 1: p_0, stored by k_eta_init in Pv[0] = this launch's p_{k-1} slot, plays p_{k-1} and -- the operand of the
    // matrix product -- g_{k-1})
    if (k < 5) w1 = zero2;
    if (k < 4) w2 = zero2;
    if (k < 3) { p3_i = zero2; g2_i = zero2; }
    if (k == 1 && act) {
        p2_i = Pw[i];
        g1_i = p2_i;
#pragma unroll
        for (int kk = 0; kk < NPRE; ++kk)
            if (kk < width) ng[kk] = Pw[col[kk]];
    }
    OCC_STAMP(1)
    double S0 = 0.0, S1 = 0.0, S2 = 0.0, xn2 = 0.0;
    {
        __shared__ double s_S[4];
        if (blockDim.x == 64) {  // (already the totals, in the canonical order)
            S0 = S[0]; S1 = S[1]; S2 = S[2]; xn2 = S[3];
        } else if (threadIdx.x < 64) {
            if (a.group_B > 0) { S0 = S[0]; S1 = S[1]; S2 = S[2]; xn2 = S[3]; }  // (k_tiles' order: already the totals)
            else { S0 = wave_sum(S[0]); S1 = wave_sum(S[1]); S2 = wave_sum(S[2]); xn2 = wave_sum(S[3]); }
            if (threadIdx.x == 0) { s_S[0] = S0; s_S[1] = S1; s_S[2] = S2; s_S[3] = xn2; }
        }
        if (blockDim.x != 64) {  // wave 0 reduced; the other waves take the totals from LDS
            __syncthreads();
            S0 = s_S[0]; S1 = s_S[1]; S2 = s_S[2]; xn2 = s_S[3];
        }
    }
    OCC_STAMP(2)
    double part[4] = {0.0, 0.0, 0.0, 0.0};
    const KryStep st = EXACT ? minres_scalars_exact(s, k, S0, S1, S2, xn2, a.maxiter) : minres_scalars(s, k, S0, S1, S2, xn2, a.maxiter);
    if (st.stop) {
        if (writer) slot_store(out, s);
        projection_partials(a, chain, i, blk);
        return;
    }
    if (st.rotate && act) {
        const double2 w = kry_form_w(st, p3_i, w1, w2);
        x.x = fma(st.phi, w.x, x.x);
        x.y = fma(st.phi, w.y, x.y);
        Ww[i] = w;
        a.Xv[co + i] = x;
        part[3] = dot2(x, x);
    }
    OCC_STAMP(3)
    if (act) {
        // h = A g_{k-1} (step 1: A p_0), the diagonal first, then the slots in order
        const double d = tau * qd + om;
        double hx = d * g1_i.x, hy = d * g1_i.y;
#pragma unroll
        for (int kk = 0; kk < NPRE; ++kk)
            if (kk < width) {
                const double av = tau * val[kk];
                hx = fma(av, ng[kk].x, hx);
                hy = fma(av, ng[kk].y, hy);
            }
        for (int kk = NPRE; kk < width; ++kk) {  // rows longer than the prefetch window
            const int jn = a.sell_col[base + kk * 64 + lane];
            const double av = tau * a.sell_val[base + kk * 64 + lane];
            const double2 gj = (k == 1) ? Pw[jn] : G1[jn];
            hx = fma(av, gj.x, hx);
            hy = fma(av, gj.y, hy);
        }
        double2 p, gn;  // p_{k-1} and g_k = A p_{k-1} at this site
        if (k == 1) {
            p = p2_i;  // p_0, already in its place
            gn = make_double2(hx, hy);
        } else {
            p = kry_form_p(st, g1_i, p3_i, p2_i);
            gn = kry_form_p(st, make_double2(hx, hy), g2_i, g1_i);  // the same three-term recurrence, applied to A's images
            Pw[i] = p;
        }
        Gw[i] = gn;
        part[0] = dot2(p, p);
        part[1] = fma(p.y, gn.y, p.x * gn.x);
        if (k >= 2) part[2] = dot2(p, p2_i);
    }
    OCC_STAMP(4)
    if (writer) slot_store(out, s);
    block_partials<4>(part, a.part_kry + ((size_t)chain * 2 + ((kl + 1) & 1)) * ((size_t)4 * a.nb_n), a.nb_n, blk);
    OCC_STAMP(5)
}

// Site i's terms of X' Omega X (upper triangle, row by row) and X'(k - omega eta) (logit.py:229-231)
template <int P>
__device__ __forceinline__ void beta_site_terms(const Ctx &c, int chain, int i, uint32_t it, double eta, double (&acc)[nacc(P)])
{
    const int n = c.n;
    const size_t ci = (size_t)chain * n + i;
    const double om = c.omega_b[it & 1][ci];
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

// ... generic path: the same terms in the same order (upper triangle row by row, then the right-hand side), emitted
// one at a time
__device__ __forceinline__ void beta_site_terms_g(const Ctx &c, int chain, int i, bool act, uint32_t it, double eta, int blk)
{
    const int n = c.n, p = c.p;
    PartialEmitter em{c.part_beta + (size_t)chain * nacc(p) * c.nb_n, c.nb_n, blk, nacc(p)};
    const int ic = act ? i : 0;
    const size_t ci = (size_t)chain * n + ic;
    const double om = c.omega_b[it & 1][ci];
    const double tt = beta_rhs_term(om, eta, (double)c.z[ci]);
    int t = 0;
    for (int aa = 0; aa < p; ++aa) {
        const double xo = c.Xt[(size_t)aa * n + ic] * om;
        for (int bb = aa; bb < p; ++bb) em.emit(t++, act ? xo * c.Xt[(size_t)bb * n + ic] : 0.0);
    }
    for (int aa = 0; aa < p; ++aa) em.emit(t++, act ? c.Xt[(size_t)aa * n + ic] * tt : 0.0);
    em.finish();
}

template <int P>
__global__ void __launch_bounds__(256) k_beta_partial(OCC_KARGS, int k_last_launch)
{
    __builtin_amdgcn_s_setprio(3);  // critical path (see k_minres)
    const Ctx &c = *cp;
    const Tile tile = tile_of_block(chain_base);
    const int chain = tile.chain, blk = tile.blk;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.ctl[e];
    // the last Krylov kernel of this launch sequence wrote slot k_last_launch & 3
    const Slot *fin = &slots[(size_t)chain * NSLOT + (k_last_launch & (NSLOT - 1))];
    struct { int done, itn, istop; } s = {fin->done, fin->itn, fin->istop};
    const bool skip = ctl.it >= sc.it_stop || sc.err != 0;
    const bool carry = !skip && !s.done;
    if (blk == 0 && threadIdx.x == 0) {
        Ctl m = ctl;
        m.koff = carry ? (uint32_t)(k_last_launch + (int)ctl.koff) : 0u;  // carry the solve into the next replay
        sc.mid[e] = m;
        if (carry) {  // the next sequence's first Krylov launch reads slot 0
            sc.carries += 1ull;
            if ((k_last_launch & (NSLOT - 1)) != 0) slot_store(&slots[(size_t)chain * NSLOT], slot_load(fin));
        }
        if (!skip && s.done) {
            sc.minres_itn_last = s.itn;
            sc.krylov_total += (unsigned long long)s.itn;
            sc.krylov_sq_total += (unsigned long long)s.itn * (unsigned long long)s.itn;
            sc.solves += 1ull;
            if (s.istop == 6) sc.err = -3;  // OCC_E_MINRES (logit.py:91-92)
        }
    }
    const int n = c.n, i = blk * blockDim.x + threadIdx.x;
    if (carry) {
        // Re-align the unfinished solve to launch number 1 of the next sequence: k_minres indexes its
        // buffers by launch number, so after L launches the live vectors g_k, g_{k-1} (Gv[L&1], Gv[(L-1)&1]), p_{k-1}, p_{k-2}
        // (Pv[(L+2)%3], Pv[(L+1)%3]), w_{k-2}, w_{k-3} (Wv[L&1], Wv[(L-1)&1]) and the partial sums
        // (parity (L+1)&1) move to where launch 1 looks for them: Gv[0], Gv[1], Pv[2], Pv[1], Wv[0], Wv[1], parity 1.
        const int L = k_last_launch;
        const size_t co = (size_t)chain * n;
        if (i < n && (L % 6) != 0) {
            const double2 g = c.Gv[L & 1][co + i], gb = c.Gv[(L - 1) & 1][co + i];
            const double2 pa = c.Pv[(L + 2) % 3][co + i], pb = c.Pv[(L + 1) % 3][co + i];
            const double2 wa = c.Wv[L & 1][co + i], wb = c.Wv[(L - 1) & 1][co + i];
            c.Gv[0][co + i] = g;
            c.Gv[1][co + i] = gb;
            c.Pv[2][co + i] = pa;
            c.Pv[1][co + i] = pb;
            c.Wv[0][co + i] = wa;
            c.Wv[1][co + i] = wb;
        }
        if (blk == 0 && ((L + 1) & 1) != 1) {
            const double *src = c.part_kry + ((size_t)chain * 2 + 0) * ((size_t)4 * c.nb_n);
            double *dst = c.part_kry + ((size_t)chain * 2 + 1) * ((size_t)4 * c.nb_n);
            for (int t = threadIdx.x; t < 4 * c.nb_n; t += blockDim.x) dst[t] = src[t];
        }
        return;
    }
    if (skip) return;
    double sums[2];
    reduce_partials<2>(c.part_proj + (size_t)chain * 2 * c.nb_n, c.nb_n, sums);
    const double a = -sums[0] / sums[1];
    if constexpr (P == 0) {
        double eta = 0.0;
        if (i < n) {
            const size_t ci = (size_t)chain * n + i;
            eta = eta_project(c.Xv[ci], a);
            c.eta[ci] = eta;
        }
        beta_site_terms_g(c, chain, i, i < n, ctl.it, eta, blk);
    } else {
        double acc[nacc(P)];
#pragma unroll
        for (int t = 0; t < nacc(P); ++t) acc[t] = 0.0;
        if (i < n) {
            const size_t ci = (size_t)chain * n + i;
            const double2 xz = c.Xv[ci];
            const double eta = eta_project(xz, a);
            c.eta[ci] = eta;
            beta_site_terms<P>(c, chain, i, ctl.it, eta, acc);
        }
        block_partials<nacc(P)>(acc, c.part_beta + (size_t)chain * nacc(P) * c.nb_n, c.nb_n, blk);
    }
}

// The partial sums of beta's system from the chain's current eta, omega_b and z (occ_cond_beta: no projection).
template <int P>
__global__ void __launch_bounds__(256) k_beta_sums(OCC_KARGS)
{
    const Ctx &c = *cp;
    const Tile tile = tile_of_block(chain_base);
    const int chain = tile.chain, blk = tile.blk;
    const Ctl ctl = scs[chain].ctl[e];
    const int n = c.n, i = blk * blockDim.x + threadIdx.x;
    if constexpr (P == 0) {
        beta_site_terms_g(c, chain, i, i < n, ctl.it, i < n ? c.eta[(size_t)chain * n + i] : 0.0, blk);
    } else {
        double acc[nacc(P)];
#pragma unroll
        for (int t = 0; t < nacc(P); ++t) acc[t] = 0.0;
        if (i < n) beta_site_terms<P>(c, chain, i, ctl.it, c.eta[(size_t)chain * n + i], acc);
        block_partials<nacc(P)>(acc, c.part_beta + (size_t)chain * nacc(P) * c.nb_n, c.nb_n, blk);
    }
}

// One visit row of the omega_a update: omega_a ~ PG(1, w'alpha) when the site exists (z = 1 or a detection
// was seen, base.py:116-119), and the row's terms of alpha's system W'Omega W, W'(y - 1/2) (logit.py:216-224).
template <int Q, int INJ = 0>
__device__ __forceinline__ void omega_a_row(const Ctx &c, const ChainScalars &sc, int chain, uint32_t it, int r, double (&acc)[nacc(Q)])
{
    const int R = c.R;
#pragma unroll
    for (int t = 0; t < nacc(Q); ++t) acc[t] = 0.0;
    if (r < R) {
        const int info = c.row_site[r];
        const int site = info & 0x7fffffff;
        // (z: written by the MAIN stream's k_z_ob, possibly while this kernel was already waiting at its gate -- an
        // agent-scope load, like every byte that is handed over inside a running kernel; see k_omega_a)
        const bool exists = (info < 0) || (__hip_atomic_load(&c.z[(size_t)chain * c.n + site], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
        if (exists) {
            double w[Q], wa = 0.0;
#pragma unroll
            for (int a = 0; a < Q; ++a) {
                w[a] = c.Wt[(size_t)a * R + r];
                wa = fma(w[a], sc.alpha[a], wa);
            }
            double om;
            if (INJ) {  // omega_a as the caller left it in the state buffer
                om = c.omega_a[(size_t)chain * R + r];
            } else {
                om = pg1_draw(sc.key, (uint32_t)r, it, STREAM_OMEGA_A, wa);
                c.omega_a[(size_t)chain * R + r] = om;
            }
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
}

// ... generic path (run-time q): the same row terms, emitted one at a time
template <int INJ>
__device__ __forceinline__ void omega_a_row_g(const Ctx &c, const ChainScalars &sc, int chain, uint32_t it, int r, int blk)
{
    const int R = c.R, q = c.q;
    PartialEmitter em{c.part_alpha + (size_t)chain * nacc(q) * c.nb_r, c.nb_r, blk, nacc(q)};
    bool exists = false;
    double om = 0.0, tt = 0.0;
    const int rc = (r < R) ? r : 0;
    if (r < R) {
        const int info = c.row_site[r];
        const int site = info & 0x7fffffff;
        exists = (info < 0) || (__hip_atomic_load(&c.z[(size_t)chain * c.n + site], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
        if (exists) {
            double wa = 0.0;
            for (int a = 0; a < q; ++a) wa = fma(c.Wt[(size_t)a * R + r], sc.alpha[a], wa);
            if (INJ) {
                om = c.omega_a[(size_t)chain * R + r];
            } else {
                om = pg1_draw(sc.key, (uint32_t)r, it, STREAM_OMEGA_A, wa);
                c.omega_a[(size_t)chain * R + r] = om;
            }
            tt = (double)c.yrow[r] - 0.5;
        }
    }
    int t = 0;
    for (int a = 0; a < q; ++a) {
        const double wo = c.Wt[(size_t)a * R + rc] * om;
        for (int b = a; b < q; ++b) em.emit(t++, exists ? wo * c.Wt[(size_t)b * R + rc] : 0.0);
    }
    for (int a = 0; a < q; ++a) em.emit(t++, exists ? c.Wt[(size_t)a * R + rc] * tt : 0.0);
    em.finish();
}

// `gate` (head of a side-stream sequence that hands over through the device counters): the kernel does k_gate's work first --
// its first workgroup announces that the previous k_noise is complete (stream order), and every workgroup that can be among
// the first on the device waits for the main stream to have finished the previous sequence before it reads z.  A kernel of
// its own for that (k_gate) cost the side stream a launch boundary per iteration, and the side stream is what bounds a
// long run (K ~ 5: main stream 44 us, side 50).  Workgroups past GATE_FIRST are dispatched only after an earlier one has
// finished, i.e. passed the wait: they need not look.
constexpr unsigned GATE_FIRST = 4096;
template <int Q, int INJ = 0>
__global__ void __launch_bounds__(256, 3) k_omega_a(OCC_KARGS, int gate)
{
    const Ctx &c = *cp;
    if (gate && c.sync != nullptr) {
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (lin < GATE_FIRST) {
            if (threadIdx.x == 0 && c.sync[SYNC_DEBUG] == 0u) {  // (test knob OCC_DEBUG_BREAK_HANDOVER: a side stream that never announces its noise)
                const unsigned j = c.sync[SYNC_SIDE_SEQ];
                if (lin == 0) sync_set(c.sync + SYNC_NOISE, j);
                if (!sync_wait(c.sync, SYNC_MAIN, j)) scs[0].err = -2;
            }
            __syncthreads();
        }
    }
    // What follows reads bytes the MAIN stream's k_z_ob wrote on other XCDs -- the chain's control words, z -- inside a
    // kernel that may have started before they were written: they are read with AGENT-SCOPE loads (L1 bypassed, coherent
    // across the XCDs), the rule of every in-kernel hand-over here (k_iter's noise, k_z_ob's alpha).  As a kernel of its
    // own, k_gate had the kernel boundary for this.  (Not an acquire fence per workgroup: its cache invalidation on every
    // one of thousands of workgroups throws k_iter's exchange out of the XCDs' L2s -- k_iter 40 -> 61 us, measured.)
    const Tile tile = tile_of_block_shared(c, 1, c.nb_r, chain_base);
    const int chain = tile.chain, blk = tile.blk;
    if (chain < 0) return;
    const ChainScalars &sc = scs[chain];
    Ctl ctl;
    ctl.it = __hip_atomic_load(&sc.ctl[e].it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ctl.koff = __hip_atomic_load(&sc.ctl[e].koff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t it_stop = __hip_atomic_load(&sc.it_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ctl.koff || ctl.it >= it_stop || __hip_atomic_load(&sc.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    if constexpr (Q == 0) {
        omega_a_row_g<INJ>(c, sc, chain, ctl.it, blk * blockDim.x + threadIdx.x, blk);
    } else {
        double acc[nacc(Q)];
        omega_a_row<Q, INJ>(c, sc, chain, ctl.it, blk * blockDim.x + threadIdx.x, acc);
        block_partials<nacc(Q)>(acc, c.part_alpha + (size_t)chain * nacc(Q) * c.nb_r, c.nb_r, blk);
    }
}

// alpha ~ N(A^-1 r, A^-1) from the partial sums of k_omega_a (logit.py:224): one block per chain, one wave per
// quantity of the system (512 threads: with one wave the reduction of 5 x 4 883 partial sums took 135 us at 500x500).
template <int INJ>
__global__ void __launch_bounds__(512) k_alpha_draw(OCC_KARGS, int sync_on)
{
    __shared__ double s_red[NACC_G], s_U[MAXG * MAXG], s_work[2 * MAXG];
    const Ctx &c = *cp;
    const int chain = chain_base + blockIdx.x;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.ctl[e];
    if (!(ctl.koff || ctl.it >= sc.it_stop || sc.err != 0)) {
        const int Q = c.q;
        reduce_partials_lds(c.part_alpha + (size_t)chain * nacc(Q) * c.nb_r, nacc(Q), c.nb_r, s_red);
        if (threadIdx.x == 0) {
            const double *a_prec = c.hyp, *a_pbm = c.hyp + Q * Q;
            const bool ok = precision_mvnorm_dev(Q, s_red, a_prec, a_pbm, sc.key, ctl.it, STREAM_ALPHA, s_U, s_work, sc.alpha,
                                                 INJ ? c.inj->alpha_eps : nullptr);
            if (!ok) sc.err = -4;
        }
    }
    (void)sync_on;  // its completion is announced by k_noise, the next kernel of the stream
}

// Last kernel of a launch sequence.  Every wave first draws beta ~ N(A^-1 r, A^-1) from the partial sums
// of k_beta_partial (logit.py:232, distributions.pyx:42-110; redundantly, in registers -- cheaper than a
// kernel of its own).  Blocks [0, nb_n): z update of the current iteration (logit.py:234-252), record
// (alpha, beta, tau) (base.py:238-239), hand the control word to the next sequence.  Blocks [nb_n, 2 nb_n):
// omega_b of the NEXT iteration, which needs only beta and eta of this one -- one launch, two independent
// roles, so the two run concurrently without a second stream.
// The z update of one site (logit.py:234-252) and the record of one iteration (base.py:238-239): shared by k_z_ob
// and by the last phase of k_iter (occ_iter.hpp), contractions explicit so that both evaluate the same operations.
template <int P, int INJ = 0>
__device__ __forceinline__ void z_update_site(const Ctx &c, uint64_t key, int chain, int i, uint32_t it, const double (&beta)[P],
                                              const double (&alpha)[MAXC], double eta_i)
{
    const int sidx = c.site_sidx[i];
    const bool not_surveyed = sidx < 0;
    if (!not_surveyed && c.obs_site[sidx]) return;  // detection seen: z stays 1 (base.py:116-118)
    const int n = c.n, Q = c.q;
    double xb = 0.0;
#pragma unroll
    for (int a = 0; a < P; ++a) xb = fma(c.Xt[(size_t)a * n + i], beta[a], xb);
    const double num1 = expit(xb + eta_i);
    double pr = num1;
    if (!not_surveyed) {
        double prod = 1.0;
        const int r0 = c.site_ptr[sidx], r1 = c.site_ptr[sidx + 1];
        for (int r = r0; r < r1; ++r) {
            double wa = 0.0;
#pragma unroll
            for (int a = 0; a < MAXC; ++a)
                if (a < Q) wa = fma(c.Wt[(size_t)a * c.R + r], -alpha[a], wa);
            const double ex = expit(wa);
            prod = (r == r0) ? ex : prod * ex;
        }
        const double num = num1 * prod;
        pr = num / ((1.0 - num1) + num);
    }
    const double u = INJ ? c.inj->z_u[i] : block_uniform(key, (uint32_t)i, 0, it, STREAM_Z);
    c.z[(size_t)chain * n + i] = (u < pr) ? 1 : 0;
}
template <int P>
__device__ __forceinline__ void record_draws(const Ctx &c, const ChainScalars &sc, int chain, uint32_t it, const double (&alpha)[MAXC],
                                             const double (&beta)[P], double tau)
{
    const uint32_t rel = it - sc.it_base;
    if (c.rec == nullptr || rel < sc.burnin || rel - sc.burnin >= sc.keep) return;
    const int Q = c.q;
    double *row = c.rec + ((size_t)chain * sc.keep + (rel - sc.burnin)) * (size_t)(Q + P + 1);
#pragma unroll
    for (int a = 0; a < MAXC; ++a)
        if (a < Q) row[a] = alpha[a];
#pragma unroll
    for (int a = 0; a < P; ++a) row[Q + a] = beta[a];
    row[Q + P] = tau;
}

template <int P>
__device__ __forceinline__ void z_ob_body(const Ctx &c, ChainScalars *__restrict__ scs, int chain_base, int e, bool synced, unsigned seq,
                                          bool per_wave, int debug_skip = 0,  // debug_skip (timing experiments): 1 = no z update, 2 = no omega_b draw
                                          bool beta_ready = false)            // beta was drawn by k_beta_draw, the previous kernel of the stream
{
    __shared__ int s_wait_ok, s_beta_ok;
    __shared__ double s_beta[P];
    const Tile tile = tile_of_block_shared(c, 0, 2 * (per_wave ? (c.n + (int)blockDim.x - 1) / (int)blockDim.x : c.nb_n), chain_base);
    const int chain = tile.chain, blk = tile.blk;
    if (chain < 0) return;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.mid[e];
    const bool skip = ctl.koff || ctl.it >= sc.it_stop || sc.err != 0;
    // per_wave: 256-thread blocks over 64-site slices (c.nb_n of them): beta is formed by wave 0 for the block
    const int nb = per_wave ? (c.n + (int)blockDim.x - 1) / (int)blockDim.x : c.nb_n;
    const uint32_t it = ctl.it;
    const bool writer = (blk == 0 && threadIdx.x == 0);
    if (writer) {
        Ctl nx = ctl;  // a mid-solve chain keeps its iteration number and its koff
        if (!skip) nx.it = it + 1;
        sc.ctl[e ^ 1] = nx;
        if (chain == chain_base && c.iter_clock != nullptr && c.iter_clock[1] != 0ull) {
            c.iter_clock[2] += c.iter_clock[1] - c.iter_clock[0];
            c.iter_clock[3] += 1ull;
            c.iter_clock[0] = ~0ull;
            c.iter_clock[1] = 0ull;
        }
        // the next sequence has the other parity: its kernels read the word this sequence's kernels do not
        if (chain == chain_base && synced) c.sync[SYNC_MAIN_SEQ + (e ^ 1)] = seq + 1u;
        if (c.claim != nullptr) {  // k_iter / k_tiles of this sequence is complete: the next one claims its places from 0
#pragma unroll
            for (int x = 0; x < 8; ++x) c.claim[(size_t)chain * 16 + x] = 0u;  // (k_tiles: one counter per XCD band)
        }
    }
    if (skip) return;
    double beta[P];
    if (beta_ready) {
#pragma unroll
        for (int a = 0; a < P; ++a) beta[a] = sc.beta[a];
    } else {
        double sums[nacc(P)];
        reduce_partials<nacc(P)>(c.part_beta + (size_t)chain * nacc(P) * c.nb_n, c.nb_n, sums);  // several waves: wave 0 + LDS
        const double *b_prec = c.hyp + c.q * c.q + c.q, *b_pbm = b_prec + P * P;
        bool ok = true;
        if (!per_wave || threadIdx.x < 64) {
            ok = precision_mvnorm_reg<P>(sums, b_prec, b_pbm, sc.key, it, STREAM_BETA, beta);
            if (per_wave && threadIdx.x == 0) {
                s_beta_ok = ok ? 1 : 0;
#pragma unroll
                for (int a = 0; a < P; ++a) s_beta[a] = beta[a];
            }
        }
        if (per_wave) {  // the draw (a Cholesky factor and Box-Muller normals) once per block, not once per wave
            __syncthreads();
            ok = s_beta_ok != 0;
#pragma unroll
            for (int a = 0; a < P; ++a) beta[a] = s_beta[a];
        }
        if (writer) {
            if (!ok) sc.err = -4;  // OCC_E_CHOLESKY
#pragma unroll
            for (int a = 0; a < P; ++a) sc.beta[a] = beta[a];
        }
    }
    // the two roles take turns over the grid (odd blocks: role 1): the Polya-Gamma waves (arithmetic-bound) and the z
    // waves (latency-bound on the visit rows) then share SIMDs from the start -- with the z blocks dealt first the
    // omega_b blocks of a large problem only got onto the device once those had drained (500x500: 22 + 30 us in a row)
    (void)nb;
    if (blk & 1) {  // role 1
        if (debug_skip & 2) return;
        omega_b_body<P>(c, sc, beta, chain, it + 1u, blk >> 1, per_wave);
        return;
    }
    // role 0 needs alpha of THIS iteration, drawn on the side stream
    const int Q = c.q;
    double alpha[MAXC];
    if (synced) {
        if (threadIdx.x == 0) s_wait_ok = sync_wait(c.sync, SYNC_ALPHA, (seq + 1u) * (unsigned)c.C) ? 1 : 0;  // one count per chain and sequence (k_noise)
        __syncthreads();
        if (!s_wait_ok && writer) sc.err = -2;
#pragma unroll
        for (int a = 0; a < MAXC; ++a) alpha[a] = (a < Q) ? load_agent(&sc.alpha[a]) : 0.0;
    } else {
#pragma unroll
        for (int a = 0; a < MAXC; ++a) alpha[a] = (a < Q) ? sc.alpha[a] : 0.0;
    }
    if (writer) record_draws<P>(c, sc, chain, it, alpha, beta, sc.tau);
    const int n = c.n, i = (blk >> 1) * blockDim.x + threadIdx.x;
    if (i >= n || (debug_skip & 1)) return;
    z_update_site<P>(c, sc.key, chain, i, it, beta, alpha, c.eta[(size_t)chain * n + i]);
}

// The z update of one site with run-time numbers of covariates (generic path); alpha and beta come from the chain's
// scalars.  Same operations as z_update_site.
template <int INJ>
__device__ __forceinline__ void z_update_site_g(const Ctx &c, const ChainScalars &sc, int chain, int i, uint32_t it, double eta_i)
{
    const int sidx = c.site_sidx[i];
    const bool not_surveyed = sidx < 0;
    if (!not_surveyed && c.obs_site[sidx]) return;
    const int n = c.n, P = c.p, Q = c.q;
    double xb = 0.0;
    for (int a = 0; a < P; ++a) xb = fma(c.Xt[(size_t)a * n + i], sc.beta[a], xb);
    const double num1 = expit(xb + eta_i);
    double pr = num1;
    if (!not_surveyed) {
        double prod = 1.0;
        const int r0 = c.site_ptr[sidx], r1 = c.site_ptr[sidx + 1];
        for (int r = r0; r < r1; ++r) {
            double wa = 0.0;
            for (int a = 0; a < Q; ++a) wa = fma(c.Wt[(size_t)a * c.R + r], -sc.alpha[a], wa);
            const double ex = expit(wa);
            prod = (r == r0) ? ex : prod * ex;
        }
        const double num = num1 * prod;
        pr = num / ((1.0 - num1) + num);
    }
    const double u = INJ ? c.inj->z_u[i] : block_uniform(sc.key, (uint32_t)i, 0, it, STREAM_Z);
    c.z[(size_t)chain * n + i] = (u < pr) ? 1 : 0;
}

// k_z_ob of the generic path: beta has been drawn by k_beta_draw<0> (the previous kernel of the stream); this path never
// runs with the device-counter hand-overs (it is not taken by the fused kernel nor by the reduced-rank model).
__device__ __forceinline__ void z_ob_body_g(const Ctx &c, ChainScalars *__restrict__ scs, int chain_base, int e, bool per_wave, int debug_skip)
{
    const Tile tile = tile_of_block(chain_base);
    const int chain = tile.chain, blk = tile.blk;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = sc.mid[e];
    const bool skip = ctl.koff || ctl.it >= sc.it_stop || sc.err != 0;
    const uint32_t it = ctl.it;
    const bool writer = (blk == 0 && threadIdx.x == 0);
    if (writer) {
        Ctl nx = ctl;
        if (!skip) nx.it = it + 1;
        sc.ctl[e ^ 1] = nx;
    }
    if (skip) return;
    if (blk & 1) {  // role 1: omega_b of the next iteration
        if (debug_skip & 2) return;
        omega_b_body_g(c, sc, chain, it + 1u, blk >> 1, per_wave);
        return;
    }
    if (writer) {  // record (base.py:238-239)
        const uint32_t rel = it - sc.it_base;
        if (c.rec != nullptr && rel >= sc.burnin && rel - sc.burnin < sc.keep) {
            double *row = c.rec + ((size_t)chain * sc.keep + (rel - sc.burnin)) * (size_t)(c.q + c.p + 1);
            for (int a = 0; a < c.q; ++a) row[a] = sc.alpha[a];
            for (int a = 0; a < c.p; ++a) row[c.q + a] = sc.beta[a];
            row[c.q + c.p] = sc.tau;
        }
    }
    const int n = c.n, i = (blk >> 1) * blockDim.x + threadIdx.x;
    if (i >= n || (debug_skip & 1)) return;
    z_update_site_g<0>(c, sc, chain, i, it, c.eta[(size_t)chain * n + i]);
}

// beta ~ N(A^-1 r, A^-1) of every chain by ONE wave per chain, for problems with so many partial sums (blocks) that
// re-reducing them in every block of k_z_ob is what that kernel spends its time on (500x500: 977 x 5 sums read by each
// of 1 954 blocks, 23 of its 76 us); k_z_ob then takes beta from the chain's scalars (flags bit 2).  The same reduction
// order and the same precision_mvnorm_reg as k_z_ob's own draw: the same bits.
// P = 0 (generic path, launched with 256 threads): the waves share the quantities of the reduction, one thread factors in LDS
// (precision_mvnorm_dev, the run-time-dimension form k_alpha_draw uses).  INJ: the normals of occ_cond_beta.
template <int P, int INJ = 0>
__global__ void __launch_bounds__(256) k_beta_draw(OCC_KARGS)
{
    const Ctx &c = *cp;
    const int chain = chain_base + blockIdx.x;
    ChainScalars &sc = scs[chain];
    const Ctl ctl = INJ ? sc.ctl[e] : sc.mid[e];
    if (ctl.koff || ctl.it >= sc.it_stop || sc.err != 0) return;
    if constexpr (P == 0) {
        __shared__ double s_red[NACC_G], s_U[MAXG * MAXG], s_work[2 * MAXG];
        const int p = c.p;
        reduce_partials_lds(c.part_beta + (size_t)chain * nacc(p) * c.nb_n, nacc(p), c.nb_n, s_red);
        if (threadIdx.x == 0) {
            const double *b_prec = c.hyp + c.q * c.q + c.q, *b_pbm = b_prec + p * p;
            const bool ok = precision_mvnorm_dev(p, s_red, b_prec, b_pbm, sc.key, ctl.it, STREAM_BETA, s_U, s_work, sc.beta,
                                                 INJ ? c.inj->beta_eps : nullptr);
            if (!ok) sc.err = -4;  // OCC_E_CHOLESKY
        }
        return;
    } else {
    double sums[nacc(P)], beta[P];
    reduce_partials<nacc(P)>(c.part_beta + (size_t)chain * nacc(P) * c.nb_n, c.nb_n, sums);
    const double *b_prec = c.hyp + c.q * c.q + c.q, *b_pbm = b_prec + P * P;
    const bool ok = precision_mvnorm_reg<P>(sums, b_prec, b_pbm, sc.key, ctl.it, STREAM_BETA, beta, INJ ? c.inj->beta_eps : nullptr);
    if (threadIdx.x == 0) {
        if (!ok) sc.err = -4;  // OCC_E_CHOLESKY
#pragma unroll
        for (int a = 0; a < P; ++a) sc.beta[a] = beta[a];
    }
    }
}

// (three workgroups per CU: at two -- no spilled registers -- k_z_ob takes 19 us instead of 14 at the headline and 76 instead of 61
// at 500x500, measured in round 4 with the new sampler as in round 2 with the old one)
template <int P>
__global__ void __launch_bounds__(256, 3) k_z_ob(OCC_KARGS, int flags)  // bit 0: stream hand-overs on, bit 1: per_wave, bit 2: beta ready
{
    __builtin_amdgcn_s_setprio(3);  // critical path (see k_minres)
    const Ctx &c = *cp;
    if constexpr (P == 0) {
        z_ob_body_g(c, scs, chain_base, e, (flags & 2) != 0, (flags >> 3) & 3);
    } else {
        const bool synced = (flags & 1) && c.sync != nullptr;
        // this sequence's number: k_iter, the previous kernel of the stream, left it in SYNC_MAIN
        const unsigned seq = synced ? c.sync[SYNC_MAIN] : 0u;
        z_ob_body<P>(c, scs, chain_base, e, synced, seq, (flags & 2) != 0, (flags >> 3) & 3, (flags & 4) != 0);
    }
}


// beta draw and / or z update of one chain with INJECTED variates (occ_cond_beta, occ_cond_z): the arithmetic is k_z_ob's,
// through the same device functions (reduce_partials in the same order, precision_mvnorm_reg, z_update_site); nothing is
// recorded and the iteration number does not advance.
template <int P>
__global__ void __launch_bounds__(256) k_cond_beta_z(OCC_KARGS)
{
    const Ctx &c = *cp;
    const Tile tile = tile_of_block(chain_base);
    const int chain = tile.chain, blk = tile.blk;
    ChainScalars &sc = scs[chain];
    const Inject &inj = *c.inj;
    const Ctl ctl = sc.ctl[e];
    if constexpr (P == 0) {  // generic path: the beta draw is k_beta_draw<0, 1>'s; here only the z update
        const int n = c.n, i = blk * blockDim.x + threadIdx.x;
        if (inj.do_z && i < n) z_update_site_g<1>(c, sc, chain, i, ctl.it, c.eta[(size_t)chain * n + i]);
        return;
    } else {
    double beta[P];
    if (inj.do_beta) {
        double sums[nacc(P)];
        reduce_partials<nacc(P)>(c.part_beta + (size_t)chain * nacc(P) * c.nb_n, c.nb_n, sums);
        const double *b_prec = c.hyp + c.q * c.q + c.q, *b_pbm = b_prec + P * P;
        const bool ok = precision_mvnorm_reg<P>(sums, b_prec, b_pbm, sc.key, ctl.it, STREAM_BETA, beta, inj.beta_eps);
        if (blk == 0 && threadIdx.x == 0) {
            if (!ok) sc.err = -4;  // OCC_E_CHOLESKY
#pragma unroll
            for (int a = 0; a < P; ++a) sc.beta[a] = beta[a];
        }
    } else {
#pragma unroll
        for (int a = 0; a < P; ++a) beta[a] = sc.beta[a];
    }
    if (!inj.do_z) return;
    double alpha[MAXC];
#pragma unroll
    for (int a = 0; a < MAXC; ++a) alpha[a] = (a < c.q) ? sc.alpha[a] : 0.0;
    const int n = c.n, i = blk * blockDim.x + threadIdx.x;
    if (i < n) z_update_site<P, 1>(c, sc.key, chain, i, ctl.it, beta, alpha, c.eta[(size_t)chain * n + i]);
    }
}

// Variates of the generators above, element i from the sub-stream (key, i, iteration, stream) exactly as the kernels of the
// iteration draw them (occ_draw in the C ABI: known-answer and distributional tests on DEVICE draws).
enum : int { DRAW_PG1 = 0, DRAW_STD_GAMMA = 1, DRAW_NORMAL = 2, DRAW_UNIFORM = 3, DRAW_WAVE_SUM_CHECK = 4 };
__global__ void __launch_bounds__(256) k_draw(int kind, uint64_t key, uint32_t it, uint32_t stream, long long n, const double *__restrict__ param,
                                              double *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (kind == DRAW_WAVE_SUM_CHECK) {  // (whole waves: n is a multiple of 64) the three forms of the wave sum agree bit for bit
        double x[4], y[4], z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            x[q] = (i < n) ? param[i] * (1.0 + 0.37 * q) + (q == 3 ? param[i] * param[i] : 0.0) : 0.0;
            y[q] = x[q];
            z[q] = wave_sum(x[q]);
        }
        wave_sum4(x);
        wave_sum4_plain(y);
        bool same = true;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            same = same && __double_as_longlong(x[q]) == __double_as_longlong(y[q]) && __double_as_longlong(x[q]) == __double_as_longlong(z[q]);
        if (i < n) out[i] = same ? x[(int)(i & 3)] : __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    if (i >= n) return;
    Cursor cur(key, (uint32_t)i, it, stream);
    double v;
    switch (kind) {
        case DRAW_PG1: v = pg1_draw(key, (uint32_t)i, it, stream, param[i]); break;
        case DRAW_STD_GAMMA: v = std_gamma(cur, param[i]); break;
        case DRAW_NORMAL: v = block_normal(key, (uint32_t)i, 0, it, stream); break;
        default: v = block_uniform(key, (uint32_t)i, 0, it, stream); break;
    }
    out[i] = v;
}

}  // namespace occ
