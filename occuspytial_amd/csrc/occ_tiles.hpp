// occ_tiles.hpp -- k_tiles: the critical path of one Gibbs iteration in ONE persistent launch for problems whose sites
// outnumber the lanes k_iter can keep resident (BASELINE config 4: 500x500, 250 000 sites).  (gfx950)
//
// k_iter (occ_iter.hpp) keeps a site's recurrence vectors AND the histories of p at its eight neighbours in registers:
// 255 VGPRs, two workgroups per CU, at most 131 072 site lanes on the device.  Beyond that round 2 ran one launch per
// MINRES step: 9-13 launch boundaries per iteration, every step streaming the seven vectors through HBM (68 MB at
// 500x500), the number of launches guessed when the graph is captured.  k_tiles is the same phases A / B / C
// (tau, right-hand side, p_0 | joint MINRES | projection, eta, beta sums -- logit.py:206-217, 75-92, distributions.pyx:24-39,
// logit.py:226-231, scipy minres.py) with
//   * TILES of 256 sites, T consecutive tiles per 256-thread workgroup, the tile's own vectors (g, p_{k-2}, p_{k-3},
//     w_{k-3}, w_{k-4}, x: 96 bytes per site) in LDS for the whole solve -- 24 KB per tile, six tiles per CU;
//   * p EXCHANGED instead of re-formed at the neighbours: a step stores p_{k-1} (16 B per site) and gathers it at the
//     eight neighbours -- no neighbour histories, 120 registers, three workgroups per CU.  An entry of the exchange
//     buffer that still holds the CANARY has not been written yet: the gather polls it, so there is no barrier between
//     "store p" and "apply A to p" -- the only chain-wide synchronisation of a step is the reduction of its four sums;
//   * ONE XCD PER BAND of consecutive workgroups: a workgroup works for the band of the XCD it runs on (HW_REG_XCC_ID) and
//     claims its place in it, as k_iter's one-XCD forms do for a chain.  Neighbours inside a band are read through that
//     XCD's L2 (plain stores, L1-bypassing loads); only the 128-byte lines of the exchange buffer that hold a site with
//     a neighbour in ANOTHER band are stored write-through (sc1) -- two lattice rows per band edge.  Every line is
//     written whole by one store instruction of one wave in one of the two forms (MI355X_MICROARCH.md, Valid forms);
//   * the four sums of a step per GROUP (= workgroup: its T tiles added in tile order) as one 32-byte record that is its
//     own arrival flag (canary halves, three buffers in rotation -- occ_iter.hpp "XL step exchange"), stored
//     write-through; every workgroup's first wave polls the records of all groups of the chain, all loads of a poll in
//     flight at once, and runs the scalar recurrence for its workgroup.
// Same arithmetic, through the same functions (minres_pre / post, kry_form_*, eta_rhs_site, ...), and the same summation
// order as the launch-per-step kernels at 256 threads per block with KryArgs::group_T = T (block partials combined in
// wave order, groups of T consecutive blocks added in block order, lanes strided over the groups, one wave sum):
// k_tiles, k_minres and the eager stepping path return the same bits.
#pragma once
#include "occ_iter.hpp"

namespace occ {

constexpr int TILE = 256;
constexpr int TILE_VECS = 6;  // g, p (two by parity), w (two by parity), x
enum : int { TV_G = 0, TV_P = 1, TV_W = 3, TV_X = 5 };
__host__ __device__ constexpr size_t tiles_lds_bytes(int T) { return (size_t)T * TILE_VECS * TILE * sizeof(double2); }

// The invariant between launches: every entry of exchange buffer 1 and every group record of record buffer 1 holds
// the canary (step 1 of the next solve polls them).  k_tiles restores it at its end; this kernel establishes it at
// creation and after anything that may have left the buffers in another state (residency probes, a failed launch).
__global__ void __launch_bounds__(256) k_tiles_reset(const IterArgs ia)
{
    const int chain = blockIdx.y, n = ia.a.n;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ia.tex[1][(size_t)chain * ia.tiles_npad + i] = rec_canary();
    if (i < 2LL * ia.tiles_G) reinterpret_cast<double2 *>(ia.part + ((size_t)chain * 3 + 1) * ia.a.nb_n * 4)[i] = rec_canary();
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

// The poll of a step: lane l reads the records of groups l, l + 64, ... (at most GR per lane, ALL loads in flight at once)
// until none shows the canary, and adds them up in the canonical order.
template <int GR>
__device__ __forceinline__ bool poll_group_records(__amdgpu_buffer_rsrc_t buf, int ngroups, int lane, unsigned spin_limit, const ChainScalars &sc,
                                                   double (&tot)[4])
{
    unsigned spins = 0;
    for (;;) {
        double2 lo[GR], hi[GR];
#pragma unroll
        for (int r = 0; r < GR; ++r) {  // (records past the last group fall outside the descriptor: zeros)
            lo[r] = load_sc1(buf, (64 * r + lane) * 32);
            hi[r] = load_sc1(buf, (64 * r + lane) * 32 + 16);
        }
        bool pend = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) tot[q] = 0.0;
#pragma unroll
        for (int r = 0; r < GR; ++r) {
            pend = pend || rec_pending(lo[r]) || rec_pending(hi[r]);
            tot[0] += lo[r].x; tot[1] += lo[r].y; tot[2] += hi[r].x; tot[3] += hi[r].y;
        }
        if (!__any(pend)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > spin_limit) return false;
        if ((spins & 1023u) == 0u && chain_err(sc) != 0) return false;
    }
    wave_sum4(tot);
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
template <int NW, int T>
__global__ void __launch_bounds__(TILE, T == 1 ? 4 : 3) k_tiles(const IterArgs ia, int e, int flags)
{
    extern __shared__ __attribute__((aligned(16))) double2 s_state[];  // [T][TILE_VECS][TILE]
    __shared__ int s_flag, s_noise_ok, s_claim;
    __shared__ double s_bcast[12];
    __shared__ double s_part[T][4][NACC_MAX];  // per tile and wave: the block partials of a reduction (wave order)
    __shared__ Slot s_slot;
    const bool probe = (flags & 2) != 0;
    const int sync_on = probe ? 0 : (flags & 1);
    const unsigned spin_limit = probe ? ITER_PROBE_SPIN_LIMIT : ITER_SPIN_LIMIT;
    const KryArgs &a = ia.a;
    const int chain = (int)blockIdx.y, tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned my_xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;  // HW_REG_XCC_ID
    const bool synced = sync_on && ia.sync != nullptr;
    // ---- which group?  The band of the XCD this workgroup runs on, the next free place in it
    const int G = ia.tiles_G, B = ia.tiles_B;
    const int band_first = (int)my_xcc * B, band_size = min(B, G - band_first);
    if (band_size <= 0) return;
    int ticket = 0;
    if (tid == 0) ticket = (int)__hip_atomic_fetch_add(ia.claim + (size_t)chain * 16 + my_xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    const size_t co = (size_t)chain * n;
    const bool lead = tid < 64;
    const double2 zero2 = make_double2(0.0, 0.0);
    // exchange buffers (p of a step; three in rotation), group records (three in rotation), tagged records
    const size_t cox = (size_t)chain * ia.tiles_npad;  // (a chain's exchange buffer starts on a 128-byte line)
    const __amdgpu_buffer_rsrc_t ebuf[3] = {
        __builtin_amdgcn_make_buffer_rsrc((void *)(ia.tex[0] + cox), 0, n * 16, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc((void *)(ia.tex[1] + cox), 0, n * 16, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc((void *)(ia.tex[2] + cox), 0, n * 16, 0x00020000)};
    double *part_base = ia.part + (size_t)chain * 3 * nt * 4;
    const __amdgpu_buffer_rsrc_t pbuf[3] = {
        __builtin_amdgcn_make_buffer_rsrc((void *)part_base, 0, G * 32, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc((void *)(part_base + (size_t)nt * 4), 0, G * 32, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc((void *)(part_base + (size_t)nt * 8), 0, G * 32, 0x00020000)};
    const __amdgpu_buffer_rsrc_t tbuf = __builtin_amdgcn_make_buffer_rsrc((void *)(ia.trec + (size_t)chain * nt * 4), 0, nt * 32, 0x00020000);
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
    int off[T][NW];    // byte offset of neighbour kk in a [n] double2 array (the site itself where there is none)
    double av[T][NW];  // Q_ij, then tau * Q_ij
    double dg[T];      // tau * Q_ii + omega_b
    unsigned sc1mask = 0u;  // bit t: this lane's 128-byte line of tile t holds a site with a neighbour in another band
    bool act[T];
    double om[T], zv[T];
    {
        double xb[T], qd[T], en[T], up[T];
        double2 x0[T], xn[T][NW];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int i = (wg * T + t) * TILE + tid;
            act[t] = i < n;
            const int ic = act[t] ? i : n - 1;
            const size_t ci = co + ic;
            int base, width;
            {
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
#pragma unroll
            for (int kk = 0; kk < NW; ++kk) {
                const int slot = base + ((kk < width) ? kk * 64 : 0) + (ic & 63);
                const bool has = act[t] && kk < width;
                const int j = has ? a.sell_col[slot] : ic;
                av[t][kk] = has ? a.sell_val[slot] : 0.0;
                off[t][kk] = has ? j * 16 : (act[t] ? i * 16 : n * 16);
                xn[t][kk] = a.Xv[co + j];
                remote = remote || (has && (j / (T * TILE)) / B != (int)my_xcc);
            }
            // (the 8 lanes of a 128-byte line decide together: a line is stored whole in ONE of the two forms)
            const unsigned long long m = __ballot(remote);
            if ((m >> (lane & 56)) & 0xffull) sc1mask |= 1u << t;
        }
        if (synced) {
            __syncthreads();
            if (!s_noise_ok && writer) sc.err = -2;
        }
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
        double tau = 0.0;
        if (lead) {  // the order of reduce_partials<1> (k_eta_init at 256 threads per block)
            double q = 0.0;
            const double *pq = ia.part_quad + (size_t)chain * nt;
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
        const double sqt = sqrt(tau);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int i = (wg * T + t) * TILE + tid;
            const double y = eta_rhs_site(om[t], xb[t], zv[t], en[t], up[t], sqt);
            dg[t] = tau * qd[t] + om[t];
            double ax = dg[t] * x0[t].x, az = dg[t] * x0[t].y;
#pragma unroll
            for (int kk = 0; kk < NW; ++kk) {
                av[t][kk] = tau * av[t][kk];
                ax = fma(av[t][kk], xn[t][kk].x, ax);
                az = fma(av[t][kk], xn[t][kk].y, az);
            }
            if (act[t]) ia.rhs[co + i] = y;
            st_vec(t, TV_G) = make_double2(y - ax, 1.0 - az);  // p_0: plays g at step 1 (ca = 1, cb = cc = 0)
            st_vec(t, TV_P) = zero2; st_vec(t, TV_P + 1) = zero2;
            st_vec(t, TV_W) = zero2; st_vec(t, TV_W + 1) = zero2;
            st_vec(t, TV_X) = x0[t];
        }
    }

    // ---- phase B: MINRES.  Step k: p_{k-1} formed and published, the rotation of iteration k - 2, g_k = A p_{k-1} from the
    // neighbours' p_{k-1}, the four sums; the coefficients of step k + 1 come from the sums of step k.
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
    int k = 1;
    for (;; ++k) {
        if (st.stop) break;
        const int kb = k % 3, kn = (k + 1) % 3, pa = TV_P + (k & 1), pb = TV_P + ((k + 1) & 1), wa = TV_W + (k & 1), wb = TV_W + ((k + 1) & 1);
        double part[T][4];
        // -- first half: p_{k-1} at the site (published), w_{k-2}, x_{k-2}
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int i = (wg * T + t) * TILE + tid;
            const int myoff = act[t] ? i * 16 : n * 16;
            const double2 g = st_vec(t, TV_G), p2 = st_vec(t, pa), p3 = st_vec(t, pb);  // g_{k-1}, p_{k-2}, p_{k-3}
            const double2 p = kry_form_p(st, g, p3, p2);
            const int aux = ((sc1mask >> t) & 1u) ? 16 : 0;
            // (two instructions with complementary lane sets, decided per 128-byte line: see the head of the file)
            if (aux) {
                __builtin_amdgcn_raw_buffer_store_b128(pack_d2(p), ebuf[kb], myoff, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(pack_d2(rec_canary()), ebuf[kn], myoff, 0, 16);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(pack_d2(p), ebuf[kb], myoff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(pack_d2(rec_canary()), ebuf[kn], myoff, 0, 0);
            }
            part[t][0] = dot2(p, p);
            part[t][2] = (k >= 2) ? dot2(p, p2) : 0.0;
            part[t][3] = 0.0;
            if (st.rotate) {
                const double2 w = kry_form_w(st, p3, st_vec(t, wa), st_vec(t, wb));  // (p_{k-3}, w_{k-4}, w_{k-3})
                double2 x = st_vec(t, TV_X);
                x.x = fma(st.phi, w.x, x.x);
                x.y = fma(st.phi, w.y, x.y);
                st_vec(t, wa) = w;
                st_vec(t, TV_X) = x;
                part[t][3] = dot2(x, x);
            }
            st_vec(t, pb) = p;  // p_{k-1} takes p_{k-3}'s place
        }
        // this group's record of step k + 1 shows the canary before its record of step k is out
        if (tid < 2) __builtin_amdgcn_raw_buffer_store_b128(pack_d2(rec_canary()), pbuf[kn], wg * 32 + tid * 16, 0, 16);
        // -- second half: g_k = A p_{k-1}; the neighbours' p_{k-1} are polled (the canary: not written yet)
        bool gave_up = false;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            double2 pj[NW];
            unsigned spins = 0;
            for (;;) {
                bool pend = false;
#pragma unroll
                for (int kk = 0; kk < NW; ++kk) {
                    pj[kk] = load_sc1(ebuf[kb], off[t][kk]);
                    pend = pend || (av[t][kk] != 0.0 && rec_pending(pj[kk]));
                }
                if (!__any(pend)) break;
                if (++spins > spin_limit || ((spins & 1023u) == 0u && chain_err(sc) != 0)) { gave_up = true; break; }
            }
            const double2 p = st_vec(t, pb);
            double gx = dg[t] * p.x, gy = dg[t] * p.y;
#pragma unroll
            for (int kk = 0; kk < NW; ++kk) {
                // (a slot without a neighbour has coefficient 0 and may hold the canary, a NaN: select, do not multiply)
                const double2 q = (av[t][kk] != 0.0) ? pj[kk] : zero2;
                gx = fma(av[t][kk], q.x, gx);
                gy = fma(av[t][kk], q.y, gy);
            }
            st_vec(t, TV_G) = make_double2(gx, gy);
            part[t][1] = fma(p.y, gy, p.x * gx);
            if (!act[t]) { part[t][0] = 0.0; part[t][1] = 0.0; part[t][2] = 0.0; part[t][3] = 0.0; }
        }
        // -- the sums: block partials per tile (block_partials<4>: wave sums, waves added in wave order), the group's tiles in
        // tile order, one record
#pragma unroll
        for (int t = 0; t < T; ++t) {
            wave_sum4(part[t]);  // (the bits of four wave_sum calls)
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) s_part[t][wave][q] = part[t][q];
            }
        }
        if (gave_up && lane == 0) s_flag = 1;  // (any wave; read after the barrier below)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's p, its canaries of step k + 1: out before the record
        __syncthreads();
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
        if (tid < 2) __builtin_amdgcn_raw_buffer_store_b128(pack_d2(make_double2(s_bcast[4 + 2 * tid], s_bcast[5 + 2 * tid])), pbuf[kb], wg * 32 + tid * 16, 0, 16);
        if (lead) {
            pre = minres_pre(s);  // the slot-only half of step k + 1, while the other groups arrive
            double acc[4];
            bool ok = poll_group_records<8>(pbuf[kb], G, lane, spin_limit, sc, acc);
            if (s_flag) ok = false;
            Slot t_ = slot_load(&s);
            if (ok) st = minres_post(t_, pre, k + 1, acc[0], acc[1], acc[2], acc[3], a.maxiter);
            slot_store(&s, t_);
            if (tid == 0) {
                if (!ok) { s_flag = 1; chain_fail(sc); }
                s_bcast[0] = st.ca; s_bcast[1] = st.cb; s_bcast[2] = st.cc; s_bcast[3] = st.sj;
                s_bcast[8] = st.oldeps; s_bcast[9] = st.delta; s_bcast[10] = st.denom; s_bcast[11] = st.phi;
                s_claim = (st.rotate ? 1 : 0) | (st.stop ? 2 : 0);
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler only: no load moves above the poll
        if (s_flag) { failed = true; ++k; break; }
        st.ca = s_bcast[0]; st.cb = s_bcast[1]; st.cc = s_bcast[2]; st.sj = s_bcast[3];
        st.oldeps = s_bcast[8]; st.delta = s_bcast[9]; st.denom = s_bcast[10]; st.phi = s_bcast[11];
        st.rotate = (s_claim & 1) != 0; st.stop = (s_claim & 2) != 0;
    }
    if (writer) {
        if (failed) { s.done = 1; s.istop = 6; s.itn = k; }
        slot_store(&a.slots[(size_t)chain * NSLOT], s);
        sc.minres_itn_last = s.itn;
        sc.krylov_total += (unsigned long long)s.itn;
        sc.krylov_sq_total += (unsigned long long)s.itn * (unsigned long long)s.itn;
        sc.solves += 1ull;
        if (s.istop == 6 && !failed) sc.err = -3;  // OCC_E_MINRES (logit.py:91-92)
    }

    // ---- phase C: sum-to-zero projection, eta, partial sums of beta's system --------------------------------------
    if (failed) {
        if (writer) chain_fail(sc);
        return;  // (the host re-runs the call on the launch-per-step path and re-establishes the canaries before coming back)
    }
    {
        // projection partials per tile (block_partials<2>), one TAGGED record per tile: {sum x, tag | sum z, tag}
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const double2 x = st_vec(t, TV_X);
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
        // every workgroup of the chain has stored its projection record, i.e. has left the solve: nobody polls the
        // step buffers any more -- the invariant between launches (k_tiles_reset) is restored here
        if (tid < 2) __builtin_amdgcn_raw_buffer_store_b128(pack_d2(rec_canary()), pbuf[1], wg * 32 + tid * 16, 0, 16);
        const int P = ia.p;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int i = (wg * T + t) * TILE + tid;
            const int myoff = act[t] ? i * 16 : n * 16;
            if ((sc1mask >> t) & 1u) __builtin_amdgcn_raw_buffer_store_b128(pack_d2(rec_canary()), ebuf[1], myoff, 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b128(pack_d2(rec_canary()), ebuf[1], myoff, 0, 0);
            const double2 x = st_vec(t, TV_X);
            double eta = 0.0;
            if (act[t]) {
                eta = eta_project(x, proj_a);
                a.Xv[co + i] = x;
                ia.eta[co + i] = eta;
            }
            // the site's terms of beta's system (beta_site_terms), block partials per tile (block_partials<nacc(P)>)
            int nq = 0;
            OCC_SWITCH_DIM(P, { nq = nacc(D); tile_beta_terms<D>(ia.Xt, n, i, act[t], om[t], eta, zv[t], s_part[t][wave], lane); });
            __syncthreads();
            if (tid < nq && wg * T + t < nt) {
                double tv = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) tv += s_part[t][w][tid];
                ia.part_beta[((size_t)chain * nq + tid) * nt + (wg * T + t)] = tv;
            }
        }
    }
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
