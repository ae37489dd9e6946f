// occ_tiles.hpp -- k_tiles: the critical path of one Gibbs iteration in ONE persistent launch for problems whose sites
// outnumber the lanes k_iter can keep resident (BASELINE config 4: 500x500, 250 000 sites).  (gfx950)
//
// k_iter (occ_iter.hpp) keeps a site's recurrence vectors AND the histories of p at its eight neighbours in registers:
// 255 VGPRs, two workgroups per CU, at most 131 072 site lanes on the device.  Beyond that round 2 ran one launch per
// MINRES step: 9-13 launch boundaries per iteration, every step streaming the seven vectors through HBM (68 MB at
// 500x500), the number of launches guessed when the graph is captured.  k_tiles is the same phases A / B / C
// (tau, right-hand side, p_0 | joint MINRES | projection, eta, beta sums -- logit.py:206-217, 75-92, distributions.pyx:24-39,
// logit.py:226-231, scipy minres.py) with
//   * TILES of 256 sites, T consecutive tiles per 256-thread workgroup, the tile's own histories (p_{k-2}, p_{k-3}, w_{k-3},
//     w_{k-4}: 64 bytes per site) in LDS for the whole solve -- 16 KB per tile -- and g_{k-1}, g_{k-2}, x in registers;
//   * g EXCHANGED: a step stores g_k = A p_{k-1} (16 B per site) and the next one gathers it at the eight neighbours for
//     h = A g_k, from which p_k and g_{k+1} follow by the three-term recurrence (occ_kernels.hpp, k_minres) -- no neighbour
//     histories, and no exchange of its own: a workgroup stores g_k, drains, and only then publishes its sums of step k, so
//     whoever holds the chain's sums of step k (which every workgroup waits for anyway) may gather g_k.  (Round 3 first
//     exchanged p_{k-1} within the step, with a flag per workgroup that its neighbours waited for: a second hand-over per
//     step.)  Only p_0 still goes out behind such flags (phase B's head);
//   * ONE XCD PER BAND of consecutive workgroups: a workgroup works for the band of the XCD it runs on (HW_REG_XCC_ID) and
//     claims its place in it, as k_iter's one-XCD forms do for a chain.  Neighbours inside a band are read through that
//     XCD's L2 (plain stores, L1-bypassing loads); only the 128-byte lines of the exchange buffer that hold a site with
//     a neighbour in ANOTHER band are stored write-through (sc1) -- two lattice rows per band edge.  Every line is written
//     whole by one store instruction of one wave in one of the two forms (MI355X_MICROARCH.md, Valid forms);
//   * the four sums of a step in TWO LEVELS: every workgroup (its T tiles added in tile order) stores one 32-byte record
//     that is its own arrival flag (canary halves, three buffers in rotation -- occ_iter.hpp "XL step exchange") with
//     plain stores; the first workgroup of a band adds up its band's records (one per lane, through the band's L2) and
//     publishes the band's sums write-through; every workgroup polls the EIGHT band records and adds them in band order.
//     (One level -- every workgroup polling all 489 group records across the XCDs -- cost 9 us per step.)
// Same arithmetic, through the same functions (minres_pre / post, kry_form_*, eta_rhs_site, ...), and the same summation
// order as the launch-per-step kernels at 256 threads per block with KryArgs::group_T = T, group_B = B (block partials
// combined in wave order, groups of T consecutive blocks added in block order, a band's groups one per lane and a wave sum,
// the bands added in band order): k_tiles, k_minres and the eager stepping path return the same bits.
#pragma once
#include "occ_iter.hpp"

namespace occ {

constexpr int TILE = 256;
constexpr int TILE_VECS = 4;  // in LDS: p (two by parity), w (two by parity); g and x live in registers
enum : int { TV_P = 0, TV_W = 2 };
__host__ __device__ constexpr size_t tiles_lds_bytes(int T) { return (size_t)T * TILE_VECS * TILE * sizeof(double2); }
// Workgroups per CU by tiles per workgroup: registers (512 per lane and SIMD: 128 / 168 / 256 per wave at 4 / 3 / 2 workgroups
// of four waves) and LDS (16 KB per tile of the CU's 160 KB).  Tiles per CU: 4, 6, 9, 8.
__host__ __device__ constexpr int tiles_wg_per_cu(int T) { return T == 1 ? 4 : (T == 4 ? 2 : 3); }

// The invariant between launches: every group record and every band record of record buffer 1 holds the canary (step 1
// of the next solve polls them).  k_tiles restores it at its end; this kernel establishes it at creation and after
// anything that may have left the buffers in another state (residency probes, a failed launch).
__global__ void __launch_bounds__(256) k_tiles_reset(const IterArgs ia)
{
    const int chain = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2LL * ia.tiles_G) reinterpret_cast<double2 *>(ia.part + ((size_t)chain * 3 + 1) * ia.a.nb_n * 4)[i] = rec_canary();
    if (i < 2LL * XL_SLOTS) reinterpret_cast<double2 *>(ia.tband + ((size_t)chain * 3 + 1) * XL_SLOTS * 4)[i] = rec_canary();
}

// Broadcast of one lane's double; the lane index is wave-uniform.
__device__ __forceinline__ double readlane_f64_t(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Records {half0, half1} of 16 bytes each with a TAG in the second double of both halves (the projection's sums, the
// residency probe): no reset between uses -- a half whose tag is not the awaited one has not arrived.
__device__ __forceinline__ bool poll_tagged(__amdgpu_buffer_rsrc_t buf, int nrec, int lane, double tag, unsigned spin_limit, const ChainScalars &sc,
                                            double &s0, double &s1)
{
    unsigned spins = 0;
    for (;;) {
        bool pend = false;
        s0 = 0.0;
        s1 = 0.0;
        for (int base = 0; base < nrec; base += 256) {  // the order of reduce_partials: lane l its records l, l + 64, ..., four rounds in flight
            double2 lo[4], hi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lo[r] = load_sc1(buf, (base + 64 * r + lane) * 32);
                hi[r] = load_sc1(buf, (base + 64 * r + lane) * 32 + 16);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool in = base + 64 * r + lane < nrec;
                pend = pend || (in && (lo[r].y != tag || hi[r].y != tag));
                s0 += in ? lo[r].x : 0.0;
                s1 += in ? hi[r].x : 0.0;
            }
        }
        if (!__any(pend)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > spin_limit) return false;
        if ((spins & 1023u) == 0u && chain_err(sc) != 0) return false;
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    return true;
}

// A band's sums of a step: lane l reads the record of the band's l-th group (plain-stored by a workgroup of this XCD:
// L2-served) until none shows the canary; one wave sum.  Returns false when it gave up.
__device__ __forceinline__ bool poll_band_groups(__amdgpu_buffer_rsrc_t buf, int soff, int first, int count, int lane, unsigned spin_limit, const ChainScalars &sc,
                                                 double (&tot)[4])
{
    unsigned spins = 0;
    for (;;) {
        const bool in = lane < count;
        const int o = (first + (in ? lane : 0)) * 32;
        const double2 lo = unpack_d2(__builtin_amdgcn_raw_buffer_load_b128(buf, o, soff, 16)), hi = unpack_d2(__builtin_amdgcn_raw_buffer_load_b128(buf, o + 16, soff, 16));
        tot[0] = in ? lo.x : 0.0; tot[1] = in ? lo.y : 0.0; tot[2] = in ? hi.x : 0.0; tot[3] = in ? hi.y : 0.0;
        if (!__any(in && (rec_pending(lo) || rec_pending(hi)))) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > spin_limit) return false;
        if ((spins & 1023u) == 0u && chain_err(sc) != 0) return false;
    }
    wave_sum4(tot);
    return true;
}
// The chain's sums of a step: the eight band records (write-through stores, L1-bypassing loads), added in band order.
// Lane x < nbands loads band x's record (eight registers where "every lane all eight" took 64); the totals are formed from
// the lanes' values by v_readlane, band 0 first.  (`nbands`: bands that hold a workgroup.)
__device__ __forceinline__ bool poll_bands(__amdgpu_buffer_rsrc_t buf, int soff, int nbands, int lane, unsigned spin_limit, const ChainScalars &sc, double (&tot)[4])
{
    unsigned spins = 0;
    double2 lo, hi;
    for (;;) {
        const bool in = lane < nbands;
        const int o = (in ? lane : 0) * 32;
        lo = unpack_d2(__builtin_amdgcn_raw_buffer_load_b128(buf, o, soff, 16));
        hi = unpack_d2(__builtin_amdgcn_raw_buffer_load_b128(buf, o + 16, soff, 16));
        if (!__any(in && (rec_pending(lo) || rec_pending(hi)))) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > spin_limit) return false;
        if ((spins & 1023u) == 0u && chain_err(sc) != 0) return false;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) tot[q] = 0.0;
    for (int x = 0; x < nbands; ++x) {
        tot[0] += readlane_f64_t(lo.x, x); tot[1] += readlane_f64_t(lo.y, x);
        tot[2] += readlane_f64_t(hi.x, x); tot[3] += readlane_f64_t(hi.y, x);
    }
    return true;
}

// Site i's terms of X' Omega X (upper triangle, row by row) and X'(k - omega eta) (beta_site_terms, occ_kernels.hpp), their
// wave sums into `out` (one wave's row of the tile's block partials).
template <int D>
__device__ __forceinline__ void tile_beta_terms(const double *Xt, int n, int i, bool act, double om, double eta, double zval, double *out, int lane)
{
    double acc[nacc(D)], xx[D];
    const double tt = beta_rhs_term(om, eta, zval);
#pragma unroll
    for (int aa = 0; aa < D; ++aa) xx[aa] = act ? Xt[(size_t)aa * n + i] : 0.0;
    int u = 0;
#pragma unroll
    for (int aa = 0; aa < D; ++aa) {
        const double xo = xx[aa] * om;
#pragma unroll
        for (int bb = aa; bb < D; ++bb) acc[u++] = act ? xo * xx[bb] : 0.0;
    }
#pragma unroll
    for (int aa = 0; aa < D; ++aa) acc[u++] = act ? xx[aa] * tt : 0.0;
#pragma unroll
    for (int u2 = 0; u2 < nacc(D); ++u2) {
        const double r = wave_sum(acc[u2]);
        if (lane == 0) out[u2] = r;
    }
}

// NW: neighbour slots per site held in registers (rows of at most NW off-diagonals); T: tiles per workgroup.
// flags: bit 0 = hand over to / from the side stream through the device counters; bit 1 = residency probe (one exchange
// among the workgroups of every chain with a short time limit, nothing else -- same grid, registers and LDS as the real
// launch).
// DIA = 1: the off-diagonals lie on at most eight diagonals with one value each (KryArgs::dia_*: any unweighted lattice) -- a
// site's neighbours and coefficients are (mask bit, constant offset, constant value): one register per tile where the
// general (SELL) form keeps 24, which is what lets two tiles per workgroup fit four workgroups of 128 registers on a CU.
// GB: tiles of the workgroup whose gathers of g are in flight together in a step (1: tile by tile, as in round 3 -- with the
// step loop spilling registers more in flight was slower; 2, 4: the round trips of T / GB batches in a row instead of T).
template <int NW, int T, int DIA, int GB = 1>
__global__ void __launch_bounds__(TILE, tiles_wg_per_cu(T)) k_tiles(const IterArgs ia, int e, int flags)
{
    extern __shared__ __attribute__((aligned(16))) double2 s_state[];  // [T][TILE_VECS][TILE]; phase C: the block partials of beta's system
    __shared__ int s_flag, s_noise_ok, s_claim, s_wlo[4], s_whi[4];
    __shared__ double s_bcast[12];
    __shared__ double s_part[T][4][4];  // per tile and wave: the block partials of a step's sums (wave order)
    __shared__ Slot s_slot;
    const bool probe = (flags & 2) != 0;
    const int sync_on = probe ? 0 : (flags & 1);
    const unsigned spin_limit = probe ? ITER_PROBE_SPIN_LIMIT : ITER_SPIN_LIMIT;
    const KryArgs &a = ia.a;
    // (a scalar register by force: with plain `blockIdx.y` the compiler sank the 64-bit extension of the chain number into the
    // one-lane branch of the claim below, merged it back as a VECTOR register, and every buffer descriptor derived from it
    // became lane-dependent in its eyes: a waterfall loop around each of the kernel's 65 buffer accesses)
    const int chain = (int)blockIdx.y, tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    size_t chain64 = (size_t)chain;
    asm volatile("" : "+s"(chain64));  // pinned to scalar registers HERE, in front of the one-lane branch below
    const unsigned my_xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;  // HW_REG_XCC_ID
    const bool synced = sync_on && ia.sync != nullptr;
    // ---- which group?  The band of the XCD this workgroup runs on, the next free place in it
    const int G = ia.tiles_G, B = ia.tiles_B;
    const int band_first = (int)my_xcc * B, band_size = min(B, G - band_first);
    if (band_size <= 0) return;
    int ticket = 0;
    if (tid == 0) ticket = (int)__hip_atomic_fetch_add(ia.claim + chain64 * 16 + my_xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_setprio(3);
    ChainScalars &sc = a.scs[chain];
    const Ctl ctl = sc.ctl[e];
    const uint32_t it_stop = sc.it_stop;
    const int err0 = sc.err;
    if (tid == 0) { s_claim = ticket; s_flag = 0; }
    __syncthreads();
    const int place = __builtin_amdgcn_readfirstlane(s_claim);
    if (place >= band_size) return;
    const int wg = band_first + place;
    if (synced && tid == 0) {
        const unsigned j = ia.sync[SYNC_MAIN_SEQ + e];
        s_noise_ok = sync_wait(ia.sync, SYNC_NOISE, j) ? 1 : 0;
    }
    const bool writer = (wg == 0 && tid == 0);
    if (synced && writer && chain == 0) sync_set(ia.sync + SYNC_MAIN, ia.sync[SYNC_MAIN_SEQ + e]);
    if (!probe && (ctl.koff || ctl.it >= it_stop || err0 != 0)) {
        if (writer) sc.mid[e] = ctl;
        return;
    }
    const unsigned long long clk0 = writer ? (unsigned long long)wall_clock64() : 0ull;
    const uint32_t it = ctl.it;
    const int n = a.n, nt = a.nb_n;  // sites; tiles (= blocks of 256 of the launch-per-step kernels)
    const size_t co = chain64 * n;
    const bool lead = tid < 64;
    const double2 zero2 = make_double2(0.0, 0.0);
    // exchange buffers (p of a step; three in rotation), group records (three in rotation), tagged records
    // ONE descriptor per family of rotating buffers, the buffer of a step chosen by the instruction's SCALAR offset (three
    // descriptors each cost 36 more scalar registers, spilled and re-read every step).  The descriptors span all three
    // buffers, so nothing is clipped at a buffer's end: lanes without a site store nothing and gather their own row n - 1.
    const int e_stride = ia.C * ia.tiles_npad * 16, p_stride = nt * 32, b_stride = XL_SLOTS * 32;  // bytes between the rotating buffers
    const __amdgpu_buffer_rsrc_t ebuf = __builtin_amdgcn_make_buffer_rsrc((void *)(ia.tex[0] + chain64 * ia.tiles_npad), 0, 2 * e_stride + n * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t pbuf = __builtin_amdgcn_make_buffer_rsrc((void *)(ia.part + chain64 * ITER_PART_DOUBLES * nt), 0, 3 * p_stride, 0x00020000);
    const __amdgpu_buffer_rsrc_t tbuf = __builtin_amdgcn_make_buffer_rsrc((void *)(ia.trec + chain64 * nt * 4), 0, nt * 32, 0x00020000);
    const __amdgpu_buffer_rsrc_t bbuf = __builtin_amdgcn_make_buffer_rsrc((void *)(ia.tband + chain64 * 3 * XL_SLOTS * 4), 0, 3 * b_stride, 0x00020000);
    // "my p_0 is out": two words per workgroup -- [0][wg] plain (readers of this band), [1][wg] write-through (others)
    const __amdgpu_buffer_rsrc_t fbuf = __builtin_amdgcn_make_buffer_rsrc((void *)(ia.tflag + chain64 * 2 * G), 0, 2 * G * 8, 0x00020000);
    const bool band_leader = place == 0;
    const unsigned bar_base = sc.bar_base;
    auto st_vec = [&](int t, int v) -> double2 & { return s_state[((size_t)t * TILE_VECS + v) * TILE + tid]; };

    // The tag of this launch's tagged records: the chain's launch counter (ChainScalars::bar_base: every launch that gets
    // this far adds one at its end, the host adds 16 whenever it re-establishes the invariant) -- never reused, so a record
    // of an earlier launch (or of a launch that failed half-way) is never taken for this one's
    const double tag = 1.0 + (double)bar_base;
    if (probe) {  // every tile publishes a tagged record, every workgroup waits for all of them
        if (tid < 2 * T && wg * T + (tid >> 1) < nt)
            __builtin_amdgcn_raw_buffer_store_b128(pack_d2(make_double2(0.0, -tag)), tbuf, (wg * T + (tid >> 1)) * 32 + (tid & 1) * 16, 0, 16);
        if (lead) {
            double s0, s1;
            const bool ok = poll_tagged(tbuf, nt, lane, -tag, spin_limit, sc, s0, s1);
            if (tid == 0 && !ok) chain_fail(sc);
        }
        __syncthreads();
        if (writer) sc.bar_base = bar_base + 1u;
        return;
    }

    // ---- phase A: tau, right-hand side, p_0 = b - A x0 (inputs come from earlier launches: plain loads) -----------
    PHASE_STAMP(0, 0)
    constexpr int TS = DIA ? 1 : T;  // (DIA: the two arrays below are not used)
    int off[TS][NW];   // byte offset of neighbour kk in a [n] double2 array (the site itself where there is none)
    double av[TS][NW]; // Q_ij, then tau * Q_ij
    unsigned dmask = 0u;  // DIA: bits 8 t .. 8 t + 7 = which of the diagonals site t of this lane has
    double tau_r = 0.0;   // DIA: tau (a coefficient is tau * dia_val[kk], formed where it is used: the same product)
    double dg[T];      // tau * Q_ii + omega_b
    // neighbour kk of tile t's site: offset in the exchange buffers and coefficient
    // (DIA, in the step loop: a neighbour a site does not have is gathered from an offset OUTSIDE the exchange buffers'
    // descriptor -- the load returns zeros without touching memory -- so its coefficient can stay the diagonal's uniform
    // tau * value, a scalar operand: fma(c, 0, h) = h as fma(0, g_i, h) was.  Round 3 selected offset AND coefficient per lane:
    // 96 v_cndmask per step, or, hoisted out of the loop by the compiler, 96 registers.)
    constexpr int OOB = 0x7ffffff0;
    auto nb_off = [&](int t, int kk, int i, int myoff) -> int {
        if constexpr (DIA) return ((dmask >> (8 * t + kk)) & 1u) ? (i + a.dia_off[kk]) * 16 : OOB;
        else return off[t][kk];
    };
    auto nb_av = [&](int t, int kk) -> double {
        if constexpr (DIA) return tau_r * a.dia_val[kk];
        else return av[t][kk];
    };
    unsigned sc1mask = 0u;  // bit t: this lane's 128-byte line of tile t holds a site with a neighbour in another band
    int wlo = wg, whi = wg;  // the workgroups that hold this workgroup's neighbours (a superset: the range between the extremes)
    bool act[T];
    double2 gr[T], xr[T];  // g_{k-1} and x of this lane's sites (registers: with them in LDS a CU held six tiles, now eight)
    {
        // (omega_b and z of the sites are needed here and again in phase C, not in between: phase C loads them again --
        // 16 registers per lane at T = 4 that the step loop spilled around until round 4)
        double om[T], zv[T];
        double xb[T], qd[T], en[T], up[T], xav[T][NW];
        double2 x0[T], xn[T][NW];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int i = (wg * T + t) * TILE + tid;
            act[t] = i < n;
            const int ic = act[t] ? i : n - 1;
            const size_t ci = co + ic;
            int base = 0, width = 0;
            if constexpr (!DIA) {
                const int sl = ic >> 6;
                if (a.ell_w > 0) { width = a.ell_w; base = sl * a.ell_w * 64; }
                else { base = a.sell_ptr[sl]; width = (a.sell_ptr[sl + 1] - base) >> 6; }
            }
            om[t] = a.omega_b[it & 1][ci];
            zv[t] = (double)ia.z[ci];
            xb[t] = xdot(ia.Xt, n, ic, sc.beta, ia.p);
            x0[t] = a.Xv[ci];
            qd[t] = a.qdiag[ic];
            bool remote = false;
            const unsigned dm = (DIA && act[t]) ? (unsigned)a.dia_mask[ic] : 0u;
            if constexpr (DIA) dmask |= dm << (8 * t);
#pragma unroll
            for (int kk = 0; kk < NW; ++kk) {
                bool has;
                int j;
                if constexpr (DIA) {
                    has = kk < a.dia_n && ((dm >> kk) & 1u);
                    j = has ? ic + a.dia_off[kk] : ic;
                    xav[t][kk] = has ? a.dia_val[kk] : 0.0;
                } else {
                    const int slot = base + ((kk < width) ? kk * 64 : 0) + (ic & 63);
                    has = act[t] && kk < width;
                    j = has ? a.sell_col[slot] : ic;
                    xav[t][kk] = has ? a.sell_val[slot] : 0.0;
                    off[t][kk] = (has ? j : ic) * 16;
                }
                xn[t][kk] = a.Xv[co + j];
                const int jw = j / (T * TILE);  // the workgroup that holds the neighbour
                remote = remote || (has && jw / B != (int)my_xcc);
                wlo = min(wlo, has ? jw : wg);
                whi = max(whi, has ? jw : wg);
            }
            // (the 8 lanes of a 128-byte line decide together: a line is stored whole in ONE of the two forms)
            const unsigned long long m = __ballot(remote);
            if ((m >> (lane & 56)) & 0xffull) sc1mask |= 1u << t;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            wlo = min(wlo, __shfl_xor(wlo, o));
            whi = max(whi, __shfl_xor(whi, o));
        }
        if (lane == 0) { s_wlo[wave] = wlo; s_whi[wave] = whi; }
        __syncthreads();  // (also: thread 0's wait for the side stream's noise is over)
        wlo = min(min(s_wlo[0], s_wlo[1]), min(s_wlo[2], s_wlo[3]));
        whi = max(max(s_whi[0], s_whi[1]), max(s_whi[2], s_whi[3]));
        if (synced && !s_noise_ok && writer) sc.err = -2;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int i = (wg * T + t) * TILE + tid;
            const size_t ci = co + (act[t] ? i : n - 1);
            if (synced) {
                en[t] = load_agent(&ia.enorm[it & 1][ci]);
                up[t] = load_agent(&ia.uprior[it & 1][ci]);
            } else {
                en[t] = ia.enorm[it & 1][ci];
                up[t] = ia.uprior[it & 1][ci];
            }
        }
        PHASE_STAMP(0, 1)
        double tau = 0.0;
        if (lead) {  // the order of reduce_partials<1> (k_eta_init at 256 threads per block)
            double q = 0.0;
            const double *pq = ia.part_quad + chain64 * nt;
            for (int b0 = lane; b0 < nt; b0 += 256) {
                double v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int bb = b0 + 64 * r;
                    const double tv = pq[min(bb, nt - 1)];
                    v[r] = (bb < nt) ? tv : 0.0;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) q += v[r];
            }
            q = wave_sum(q);
            const double rate = 0.5 * q + ia.tau_rate;
            const double gvar = synced ? load_agent(&sc.tau_gamma[it & 1]) : sc.tau_gamma[it & 1];
            tau = (1.0 / rate) * gvar;
            if (writer) sc.tau = tau;
            if (tid == 0) {
                s_bcast[0] = tau;
#define X(f) s_slot.f = 0;
                OCC_SLOT_FIELDS(X)
#undef X
            }
        }
        __syncthreads();
        tau = s_bcast[0];
        tau_r = tau;
        PHASE_STAMP(0, 2)
        const double sqt = sqrt(tau);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int i = (wg * T + t) * TILE + tid;
            const double y = eta_rhs_site(om[t], xb[t], zv[t], en[t], up[t], sqt);
            dg[t] = tau * qd[t] + om[t];
            double ax = dg[t] * x0[t].x, az = dg[t] * x0[t].y;
#pragma unroll
            for (int kk = 0; kk < NW; ++kk) {
                const double c = tau * xav[t][kk];
                if constexpr (!DIA) av[t][kk] = c;
                ax = fma(c, xn[t][kk].x, ax);
                az = fma(c, xn[t][kk].y, az);
            }
            if (act[t]) ia.rhs[co + i] = y;
            gr[t] = make_double2(y - ax, 1.0 - az);  // p_0: plays g at step 1 (ca = 1, cb = cc = 0)
            st_vec(t, TV_P) = zero2; st_vec(t, TV_P + 1) = zero2;
            st_vec(t, TV_W) = zero2; st_vec(t, TV_W + 1) = zero2;
            xr[t] = x0[t];
        }
    }

    PHASE_STAMP(0, 3)
    // ---- phase B: MINRES.  Step k: g_{k-1} gathered at the neighbours, h = A g_{k-1}, p_{k-1} and g_k = A p_{k-1} by the same
    // three-term recurrence (occ_kernels.hpp, k_minres), g_k published, the rotation of iteration k - 2, the four sums; the
    // coefficients of step k + 1 come from the sums of step k.  Whoever has the sums of step k has every workgroup's g_k --
    // each stored it, drained, before its record.  p_0 (which plays g_0) has no sums to travel with: it goes out behind a
    // FLAG per workgroup, (launch counter, 0), that the workgroups holding its neighbours wait for -- on a lattice the one
    // before and the one after (a plain word for readers in the band, a write-through copy for the others; 3 us where a
    // record of zeros through the two-level reduction -- a "step 0" -- took 7.4).
    Slot &s = s_slot;
    KryPre pre = {};
    KryStep st = {};
    bool failed = false;
    if (lead) {
        pre = minres_pre(s);
        Slot t_ = slot_load(&s);
        st = minres_post(t_, pre, 1, 0.0, 0.0, 0.0, 0.0, a.maxiter);
        slot_store(&s, t_);
    }
    st.ca = 1.0; st.cb = 0.0; st.cc = 0.0; st.rotate = false; st.stop = false;
    double2 g2r[T];  // g_{k-2} of this lane's sites
    {
        const unsigned long long fv = (unsigned long long)bar_base << 32;
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int t = 0; t < T; ++t) {
            g2r[t] = zero2;
            const int i = (wg * T + t) * TILE + tid;
            if (act[t]) {  // (two instructions with complementary lane sets, decided per 128-byte line: see the head of the file)
                if ((sc1mask >> t) & 1u) __builtin_amdgcn_raw_buffer_store_b128(pack_d2(gr[t]), ebuf, i * 16, 0, 16);
                else __builtin_amdgcn_raw_buffer_store_b128(pack_d2(gr[t]), ebuf, i * 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's p_0 is out (L2, or memory for the write-through lines)
        __syncthreads();
        if (tid < 2) {
            v2u w2;
            w2.x = (unsigned)fv; w2.y = (unsigned)(fv >> 32);
            if (tid == 0) __builtin_amdgcn_raw_buffer_store_b64(w2, fbuf, wg * 8, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b64(w2, fbuf, (G + wg) * 8, 0, 16);
        }
        if (lead) {  // one wave looks at the neighbours' flags (lane l: workgroup wlo + l)
            unsigned spins = 0;
            bool ok = true;
            for (int w0 = wlo; w0 <= whi && ok; w0 += 64) {
                for (;;) {
                    const int w = w0 + lane;
                    const bool in = w <= whi && w != wg;
                    const int wc = in ? w : wg;
                    const bool local = wc / B == (int)my_xcc;
                    const v2u f0 = __builtin_amdgcn_raw_buffer_load_b64(fbuf, (local ? wc : G + wc) * 8, 0, 16);
                    const unsigned long long f = ((unsigned long long)f0.y << 32) | f0.x;
                    if (!__any(in && f != fv)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > spin_limit || ((spins & 1023u) == 0u && chain_err(sc) != 0)) { ok = false; break; }
                }
            }
            if (!ok && lane == 0) { s_flag = 1; chain_fail(sc); }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler only: no gather moves above the flags
        failed = __builtin_amdgcn_readfirstlane(s_flag) != 0;
    }
    int k = 1;
    for (;; ++k) {
        SOLVE_STAMP(0)
        if (st.stop || failed) break;
        // (the step number is wave-uniform, but the loop leaves on values read from LDS and the compiler then takes everything
        // derived from k for lane-dependent: a buffer descriptor picked by k % 3 became a waterfall loop around EVERY gather,
        // each with its own s_waitcnt -- 9.7 us per step.  Scalar registers by force.)
        k = __builtin_amdgcn_readfirstlane(k);
        const int kb = __builtin_amdgcn_readfirstlane(k % 3), kn = __builtin_amdgcn_readfirstlane((k + 1) % 3), kp = __builtin_amdgcn_readfirstlane((k + 2) % 3);
        const int pa = TV_P + (k & 1), pb = TV_P + ((k + 1) & 1), wa = TV_W + (k & 1), wb = TV_W + ((k + 1) & 1);
        const int e_so = kb * e_stride, e_sp = kp * e_stride, p_so = kb * p_stride, p_sn = kn * p_stride, b_so = kb * b_stride, b_sn = kn * b_stride;  // scalar offsets
        constexpr int GBT = GB > T ? T : GB;
#pragma unroll
        for (int t0 = 0; t0 < T; t0 += GBT) {
            // g_{k-1} at the neighbours (complete since the sums of step k - 1 arrived), GBT tiles' worth in flight
            double2 gj[GBT][NW];
#pragma unroll
            for (int tt = 0; tt < GBT; ++tt) {
                const int t = t0 + tt;
                if (t < T) {
                    const int i = (wg * T + t) * TILE + tid;
                    const int myoff = (act[t] ? i : n - 1) * 16;
#pragma unroll
                    for (int kk = 0; kk < NW; ++kk) gj[tt][kk] = unpack_d2(__builtin_amdgcn_raw_buffer_load_b128(ebuf, nb_off(t, kk, i, myoff), e_sp, 16));
                }
            }
#pragma unroll
            for (int tt = 0; tt < GBT; ++tt) {
                const int t = t0 + tt;
                if (t >= T) continue;
                const int i = (wg * T + t) * TILE + tid;
                double2 gn;
                double part_t[4] = {0.0, 0.0, 0.0, 0.0};
                {
                    const double2 g = gr[t], p2 = st_vec(t, pa), p3 = st_vec(t, pb);  // g_{k-1}, p_{k-2}, p_{k-3}
                    // while the gathers travel: w_{k-2}, x_{k-2} (the rotation of iteration k - 2)
                    if (st.rotate) {
                        const double2 w = kry_form_w(st, p3, st_vec(t, wa), st_vec(t, wb));  // (p_{k-3}, w_{k-4}, w_{k-3})
                        xr[t].x = fma(st.phi, w.x, xr[t].x);
                        xr[t].y = fma(st.phi, w.y, xr[t].y);
                        st_vec(t, wa) = w;
                        part_t[3] = dot2(xr[t], xr[t]);
                    }
                    const double2 p = kry_form_p(st, g, p3, p2);  // p_{k-1}
                    double hx = dg[t] * g.x, hy = dg[t] * g.y;    // h = A g_{k-1}: the diagonal, then the slots in order
#pragma unroll
                    for (int kk = 0; kk < NW; ++kk) {
                        const double c = nb_av(t, kk);
                        hx = fma(c, gj[tt][kk].x, hx);
                        hy = fma(c, gj[tt][kk].y, hy);
                    }
                    gn = kry_form_p(st, make_double2(hx, hy), g2r[t], g);  // g_k
                    st_vec(t, pb) = p;
                    g2r[t] = g;
                    gr[t] = gn;
                    part_t[0] = dot2(p, p);
                    part_t[1] = fma(p.y, gn.y, p.x * gn.x);
                    part_t[2] = (k >= 2) ? dot2(p, p2) : 0.0;
                    if (!act[t]) { part_t[0] = 0.0; part_t[1] = 0.0; part_t[2] = 0.0; part_t[3] = 0.0; }
                }
                // g_k published (two instructions with complementary lane sets, decided per 128-byte line: see the head of the file)
                if (act[t]) {
                    if ((sc1mask >> t) & 1u) __builtin_amdgcn_raw_buffer_store_b128(pack_d2(gn), ebuf, i * 16, e_so, 16);
                    else __builtin_amdgcn_raw_buffer_store_b128(pack_d2(gn), ebuf, i * 16, e_so, 0);
                }
                // the tile's block partials (block_partials<4>: wave sums, the waves added in wave order further down), out of the
                // registers at once: held for all T tiles they were 32 registers of the 44 the loop spilled
                wave_sum4(part_t);  // (the bits of four wave_sum calls)
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) s_part[t][wave][q] = part_t[q];
                }
            }
        }
        SOLVE_STAMP(1)
        // this group's record of step k + 1 (and, band leader, the band's) shows the canary before the one of step k is out
        if (tid < 2) __builtin_amdgcn_raw_buffer_store_b128(pack_d2(rec_canary()), pbuf, wg * 32 + tid * 16, p_sn, 0);
        if (band_leader && tid >= 2 && tid < 4) __builtin_amdgcn_raw_buffer_store_b128(pack_d2(rec_canary()), bbuf, (int)my_xcc * 32 + (tid - 2) * 16, b_sn, 16);
        // -- the sums: the tiles' block partials (above) added in wave order, the group's tiles in tile order, one record
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's g_k (L2, or memory for the write-through lines) and the canaries of step k + 1 are out
        __syncthreads();
        SOLVE_STAMP(3)
        if (tid < 4) {
            double grp = 0.0;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                double tv = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) tv += s_part[t][w][tid];
                if ((wg * T + t) < nt) grp += tv;
            }
            s_bcast[4 + tid] = grp;
        }
        __syncthreads();
        if (tid < 2) __builtin_amdgcn_raw_buffer_store_b128(pack_d2(make_double2(s_bcast[4 + 2 * tid], s_bcast[5 + 2 * tid])), pbuf, wg * 32 + tid * 16, p_so, 0);
        SOLVE_STAMP(4)
        if (lead) {
            bool ok = true;
            if (band_leader) {  // the band's sums: its groups' records, one per lane; published write-through
                double bs[4];
                ok = poll_band_groups(pbuf, p_so, band_first, band_size, lane, spin_limit, sc, bs);
                if (lane < 2) __builtin_amdgcn_raw_buffer_store_b128(pack_d2(lane == 0 ? make_double2(bs[0], bs[1]) : make_double2(bs[2], bs[3])), bbuf, (int)my_xcc * 32 + lane * 16, b_so, 16);
            }
            SOLVE_STAMP(5)
            pre = minres_pre(s);  // the slot-only half of step k + 1, while the bands arrive
            double acc[4];
            if (ok) ok = poll_bands(bbuf, b_so, (G + B - 1) / B, lane, spin_limit, sc, acc);
            SOLVE_STAMP(6)
            Slot t_ = slot_load(&s);
            if (ok) st = minres_post(t_, pre, k + 1, acc[0], acc[1], acc[2], acc[3], a.maxiter);
            slot_store(&s, t_);
            if (tid == 0) {
                if (!ok) { s_flag = 1; chain_fail(sc); }
                s_bcast[0] = st.ca; s_bcast[1] = st.cb; s_bcast[2] = st.cc; s_bcast[3] = st.sj;
                s_bcast[8] = st.oldeps; s_bcast[9] = st.delta; s_bcast[10] = st.denom; s_bcast[11] = st.phi;
                s_claim = (st.rotate ? 1 : 0) | (st.stop ? 2 : 0);
            }
            SOLVE_STAMP(7)
        }
        __syncthreads();
        SOLVE_STAMP(8)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler only: no load moves above the poll
        if (__builtin_amdgcn_readfirstlane(s_flag)) { failed = true; ++k; break; }
        st.ca = s_bcast[0]; st.cb = s_bcast[1]; st.cc = s_bcast[2]; st.sj = s_bcast[3];
        st.oldeps = s_bcast[8]; st.delta = s_bcast[9]; st.denom = s_bcast[10]; st.phi = s_bcast[11];
        {
            const int ctl_ = __builtin_amdgcn_readfirstlane(s_claim);
            st.rotate = (ctl_ & 1) != 0;
            st.stop = (ctl_ & 2) != 0;
        }
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

    // ---- phase C: sum-to-zero projection, eta, partial sums of beta's system --------------------------------------
    PHASE_STAMP(STAMP_STEPS - 1, 0)
    if (failed) {
        if (writer) chain_fail(sc);
        return;  // (the host re-runs the call on the launch-per-step path and re-establishes the canaries before coming back)
    }
    {
        // projection partials per tile (block_partials<2>), one TAGGED record per tile: {sum x, tag | sum z, tag}
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const double2 x = xr[t];
            const double r0 = wave_sum(act[t] ? x.x : 0.0), r1 = wave_sum(act[t] ? x.y : 0.0);
            if (lane == 0) { s_part[t][wave][0] = r0; s_part[t][wave][1] = r1; }
        }
        __syncthreads();
        if (tid < 2 * T) {
            const int t = tid >> 1, q = tid & 1;
            double tv = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) tv += s_part[t][w][q];
            if (wg * T + t < nt) __builtin_amdgcn_raw_buffer_store_b128(pack_d2(make_double2(tv, tag)), tbuf, (wg * T + t) * 32 + q * 16, 0, 16);
        }
        double proj_a = 0.0;
        if (lead) {
            double sx, sz;
            const bool ok = poll_tagged(tbuf, nt, lane, tag, spin_limit, sc, sx, sz);
            proj_a = -sx / sz;
            if (tid == 0) {
                s_bcast[0] = proj_a;
                s_flag = ok ? 0 : 1;
                if (!ok) chain_fail(sc);
            }
        }
        __syncthreads();
        if (s_flag) return;
        proj_a = s_bcast[0];
        PHASE_STAMP(STAMP_STEPS - 1, 1)
        // every workgroup of the chain has stored its projection record, i.e. has left the solve: nobody polls the
        // step buffers any more -- the invariant between launches (k_tiles_reset) is restored here
        if (tid < 2) __builtin_amdgcn_raw_buffer_store_b128(pack_d2(rec_canary()), pbuf, wg * 32 + tid * 16, p_stride, 0);
        if (band_leader && tid >= 2 && tid < 4) __builtin_amdgcn_raw_buffer_store_b128(pack_d2(rec_canary()), bbuf, (int)my_xcc * 32 + (tid - 2) * 16, b_stride, 16);
        const int P = ia.p;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int i = (wg * T + t) * TILE + tid;
            const double2 x = xr[t];
            double eta = 0.0;
            const size_t ci = co + (act[t] ? i : n - 1);
            const double om_t = a.omega_b[it & 1][ci], zv_t = (double)ia.z[ci];  // (as phase A read them)
            if (act[t]) {
                eta = eta_project(x, proj_a);
                a.Xv[co + i] = x;
                ia.eta[co + i] = eta;
            }
            // the site's terms of beta's system (beta_site_terms), block partials per tile (block_partials<nacc(P)>)
            int nq = 0;
            double *bpart = reinterpret_cast<double *>(s_state) + (size_t)t * 4 * NACC_MAX;  // [4 waves][NACC_MAX]: the solve's vectors are dead
            OCC_SWITCH_DIM(P, { nq = nacc(D); tile_beta_terms<D>(ia.Xt, n, i, act[t], om_t, eta, zv_t, bpart + wave * NACC_MAX, lane); });
            __syncthreads();
            if (tid < nq && wg * T + t < nt) {
                double tv = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) tv += bpart[w * NACC_MAX + tid];
                ia.part_beta[(chain64 * nq + tid) * nt + (wg * T + t)] = tv;
            }
        }
    }
    PHASE_STAMP(STAMP_STEPS - 1, 2)
    if (writer) {
        Ctl m = ctl;
        m.koff = 0u;
        sc.mid[e] = m;
        sc.bar_base = bar_base + 1u;
        atomicMin(ia.clock, clk0);
        atomicMax(ia.clock + 1, (unsigned long long)wall_clock64());
    }
}

}  // namespace occ
