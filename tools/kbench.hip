// kbench.hip -- developer micro-benchmark: times single kernels of occ_kernels.hpp on a synthetic
// lattice state, interleaved rounds in one process (not part of the product or the test-suite).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I occuspytial_amd/csrc tools/kbench.hip -o tools/kbench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

__device__ unsigned long long g_stamp[24 * 8 * 8192];
#define KSTAMP_K k_launch
#define OCC_STAMP(n) { unsigned long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if ((threadIdx.x & 63) == 0) g_stamp[((size_t)(KSTAMP_K % 24) * 8192 + (blockIdx.y * gridDim.x * (blockDim.x >> 6) + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6))) * 8 + n] = t_; }
#include "occ_kernels.hpp"

using namespace occ;

namespace occ {
template <int V>
__device__ inline double pg1_var(Cursor &c, double z)
{
    const double Z = 0.5 * fabs(z);
    const double fz = 0.125 * kPi * kPi + 0.5 * Z * Z;
    const double ptail = (V == 1) ? 0.5 / (1.0 + Z) : pg_mass_texpon(Z);
    for (;;) {
        double X;
        if (c.unif() < ptail) X = kPgT + c.expo() / fz;
        else X = pg_rtigauss(c, Z);
        if (V == 2) return 0.25 * X;
        double S = pg_a(0, X);
        const double Y = c.unif() * S;
        int n = 0;
        for (;;) {
            ++n;
            if (n & 1) { S -= pg_a(n, X); if (Y <= S) return 0.25 * X; }
            else { S += pg_a(n, X); if (Y > S) break; }
        }
    }
}
template <int V>
__global__ void __launch_bounds__(256) k_pgvar(const Ctx *cp, ChainScalars *scs, double *out)
{
    const Ctx &c = *cp;
    const int chain = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n) return;
    const ChainScalars &sc = scs[chain];
    const double xb = xdot(c.Xt, c.n, i, sc.beta, c.p) + c.eta[(size_t)chain * c.n + i];
    Cursor cur(sc.key, (uint32_t)i, 7u, STREAM_OMEGA_B);
    double r;
    if (V == 3) { r = cur.unif() + cur.expo() + cur.unif(); }            // 2 philox blocks + 1 log
    else if (V == 4) { r = pg_mass_texpon(0.5 * fabs(xb)); }               // mixture weight only
    else if (V == 5) { r = cur.norm(); }                                   // one Box-Muller normal
    else r = pg1_var<V>(cur, xb);
    out[(size_t)chain * c.n + i] = r;
}
}
__global__ void k_empty(const Ctx *cp, ChainScalars *scs, Slot *slots, int chain_base, int e, int k) {}
__global__ void k_ctl_only(const Ctx *cp, ChainScalars *scs, Slot *slots, int chain_base, int e, int k, double *sink)
{
    const Ctl ctl = scs[chain_base + blockIdx.y].ctl[e];
    if (ctl.it == 0xffffffffu) sink[0] = 1.0;
}

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e = (x);                                                         \
        if (e != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                  \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

template <class T>
T *dalloc(size_t n, int fill = 0)
{
    T *p;
    CK(hipMalloc(&p, n * sizeof(T)));
    CK(hipMemset(p, fill, n * sizeof(T)));
    return p;
}
template <class T>
T *dup(const std::vector<T> &h)
{
    T *p;
    CK(hipMalloc(&p, h.size() * sizeof(T)));
    CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

int main(int argc, char **argv)
{
    const int side = argc > 1 ? atoi(argv[1]) : 100, C = argc > 2 ? atoi(argv[2]) : 4;
    const int tpb = argc > 3 ? atoi(argv[3]) : 64, V = 5, reps = 200;
    const int n = side * side, R = n * V, p = 2, q = 2;
    Ctx c{};
    c.n = n; c.S = n; c.R = R; c.p = p; c.q = q; c.C = C;
    c.nb_n = (n + tpb - 1) / tpb; c.nb_r = (R + tpb - 1) / tpb; 
    c.maxiter = 10LL * n; c.ell_w = 8; c.tau_rate = 0.005; c.tau_shape = 0.5 * n;
    // queen lattice SELL-64
    const int nslice = (n + 63) / 64;
    std::vector<int> sell_ptr(nslice + 1, 0), sell_col((size_t)nslice * 64 * 8);
    std::vector<double> sell_val((size_t)nslice * 64 * 8, 0.0), qdiag(n, 0.0);
    for (int s = 0; s <= nslice; ++s) sell_ptr[s] = s * 64 * 8;
    for (int i = 0; i < nslice * 64; ++i) {
        int kk = 0;
        const int sl = i / 64, lane = i % 64, r = i / side, cc = i % side;
        if (i < n)
            for (int dr = -1; dr <= 1; ++dr)
                for (int dc = -1; dc <= 1; ++dc) {
                    if (!dr && !dc) continue;
                    const int rr = r + dr, c2 = cc + dc;
                    if (rr < 0 || rr >= side || c2 < 0 || c2 >= side) continue;
                    sell_col[(size_t)sl * 512 + kk * 64 + lane] = rr * side + c2;
                    sell_val[(size_t)sl * 512 + kk * 64 + lane] = -1.0;
                    ++kk;
                }
        if (i < n) qdiag[i] = kk;
        for (; kk < 8; ++kk) sell_col[(size_t)sl * 512 + kk * 64 + lane] = std::min(i, n - 1);
    }
    std::vector<double> Xt((size_t)n * p), Wt((size_t)R * q), hyp(q * q + q + p * p + p, 0.0);
    for (int i = 0; i < n; ++i) { Xt[i] = 1.0; Xt[n + i] = ((i * 37) % 100) / 25.0 - 2.0; }
    for (int r = 0; r < R; ++r) { Wt[r] = 1.0; Wt[R + r] = ((r * 53) % 100) / 25.0 - 2.0; }
    hyp[0] = hyp[3] = 0.1; hyp[6] = hyp[9] = 0.1;
    std::vector<uint8_t> yrow(R, 0), obs(n, 0);
    std::vector<int> row_site(R), sidx(n), sptr(n + 1);
    for (int i = 0; i < n; ++i) {
        sidx[i] = i; sptr[i] = i * V;
        obs[i] = (i % 5) < 2;
        for (int v = 0; v < V; ++v) { row_site[i * V + v] = i | (obs[i] ? 0x80000000 : 0); yrow[i * V + v] = obs[i] && v == 0; }
    }
    sptr[n] = R;
    c.sell_ptr = dup(sell_ptr); c.sell_col = dup(sell_col); c.sell_val = dup(sell_val); c.qdiag = dup(qdiag);
    c.Xt = dup(Xt); c.Wt = dup(Wt); c.yrow = dup(yrow); c.row_site = dup(row_site);
    c.site_sidx = dup(sidx); c.site_ptr = dup(sptr); c.obs_site = dup(obs); c.hyp = dup(hyp);
    const size_t Cn = (size_t)C * n;
    c.eta = dalloc<double>(Cn); c.rhs = dalloc<double>(Cn); c.omega_a = dalloc<double>((size_t)C * R);
    for (int b = 0; b < 2; ++b) { c.omega_b[b] = dalloc<double>(Cn); c.enorm[b] = dalloc<double>(Cn); c.uprior[b] = dalloc<double>(Cn); }
    c.z = dalloc<uint8_t>(Cn, 1);
    for (int b = 0; b < 3; ++b) c.Pv[b] = dalloc<double2>(Cn);
    for (int b = 0; b < 2; ++b) { c.Gv[b] = dalloc<double2>(Cn); c.Wv[b] = dalloc<double2>(Cn); }
    c.Xv = dalloc<double2>(Cn);
    c.part_quad = dalloc<double>((size_t)C * c.nb_n); c.part_kry = dalloc<double>((size_t)C * 8 * c.nb_n);
    c.part_proj = dalloc<double>((size_t)C * 2 * c.nb_n); c.part_beta = dalloc<double>((size_t)C * nacc(p) * c.nb_n);
    c.part_alpha = dalloc<double>((size_t)C * nacc(q) * c.nb_r);
    c.slots = dalloc<Slot>((size_t)C * NSLOT);
    c.sc = dalloc<ChainScalars>(C);
    c.rec = nullptr;
    std::vector<ChainScalars> sc(C);
    memset(sc.data(), 0, sizeof(ChainScalars) * C);
    for (int ch = 0; ch < C; ++ch) {
        sc[ch].key = 1234567 + ch; sc[ch].tau = 1.0; sc[ch].it_stop = 1u << 30;
        sc[ch].alpha[0] = 0.3; sc[ch].alpha[1] = -0.5; sc[ch].beta[0] = 0.2; sc[ch].beta[1] = 0.7;
    }
    CK(hipMemcpy(c.sc, sc.data(), sizeof(ChainScalars) * C, hipMemcpyHostToDevice));

    Ctx *cp; CK(hipMalloc(&cp, sizeof(Ctx))); CK(hipMemcpy(cp, &c, sizeof(Ctx), hipMemcpyHostToDevice));
    KryArgs ka{};
    ka.n = c.n; ka.nb_n = c.nb_n; ka.ell_w = c.ell_w; ka.maxiter = c.maxiter; ka.sell_ptr = c.sell_ptr; ka.sell_col = c.sell_col;
    ka.sell_val = c.sell_val; ka.qdiag = c.qdiag; ka.omega_b[0] = c.omega_b[0]; ka.omega_b[1] = c.omega_b[1];
    for (int b = 0; b < 2; ++b) { ka.Gv[b] = c.Gv[b]; ka.Wv[b] = c.Wv[b]; }
    for (int b = 0; b < 3; ++b) ka.Pv[b] = c.Pv[b];
    ka.Xv = c.Xv; ka.part_kry = c.part_kry; ka.part_proj = c.part_proj; ka.scs = c.sc; ka.slots = c.slots;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 gs(c.nb_n, C), gr(c.nb_r, C), blk(tpb);
    // a sane mid-solve state
    hipLaunchKernelGGL(k_omega_b<2>, gs, blk, 0, st, cp, c.sc, c.slots, 0, 0);
    hipLaunchKernelGGL(k_noise, gs, blk, 0, st, cp, c.sc, c.slots, 0, 0, 0);
    hipLaunchKernelGGL(k_eta_init, gs, blk, 0, st, cp, c.sc, c.slots, 0, 0);
    for (int k = 1; k <= 4; ++k) hipLaunchKernelGGL(k_minres<0>, gs, blk, 0, st, ka, 0, 0, k);
    CK(hipStreamSynchronize(st));

    const bool eager = getenv("KB_EAGER") != nullptr;
    auto time_graph = [&](const char *name, std::function<void()> one) {
        if (eager) {  // plain launches (for counter collection, which cannot follow graph launches)
            for (int r = 0; r < 20; ++r) one();
            CK(hipStreamSynchronize(st));
            printf("%-28s eager x20\n", name);
            return;
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int r = 0; r < reps; ++r) one();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st));
        float best = 1e30f;
        for (int round = 0; round < 3; ++round) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        printf("%-28s %8.3f us/launch\n", name, 1000.0 * best / reps);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    };
    {   // realistic cold sequence: a whole solve captured in a graph, replayed; stamps per launch number
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_eta_init, gs, blk, 0, st, cp, c.sc, c.slots, 0, 0);
        for (int k = 1; k <= 12; ++k) hipLaunchKernelGGL(k_minres<0>, gs, blk, 0, st, ka, 0, 0, k);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 50; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("solve graph (eta_init + 12 minres): %.2f us per replay (stamped build)\n", 1000.0 * ms / 50);
        std::vector<unsigned long long> h((size_t)24 * 8 * 8192);
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamp), h.size() * 8));
        const int nw = c.nb_n * C;
        for (int k : {2, 6, 10}) {
            double m[6] = {0};
            for (int w = 0; w < nw; ++w) for (int q2 = 1; q2 < 6; ++q2) m[q2] += (double)(h[((size_t)k * 8192 + w) * 8 + q2] - h[((size_t)k * 8192 + w) * 8]);
            printf("  launch %2d: loads landed %.0f | sums reduced %.0f | scalars %.0f | vector math %.0f | partials written %.0f cycles\n", k, m[1] / nw, m[2] / nw, m[3] / nw, m[4] / nw, m[5] / nw);
        }
    }
    for (int round = 0; round < 2; ++round) {
        time_graph("minres k=5", [&] { hipLaunchKernelGGL(k_minres<0>, gs, blk, 0, st, ka, 0, 0, 5); });
        time_graph("empty kernel", [&] { hipLaunchKernelGGL(k_empty, gs, blk, 0, st, cp, c.sc, c.slots, 0, 0, 5); });
        time_graph("empty kernel 1 block", [&] { hipLaunchKernelGGL(k_empty, dim3(1), blk, 0, st, cp, c.sc, c.slots, 0, 0, 5); });
        time_graph("ctl-only kernel", [&] { hipLaunchKernelGGL(k_ctl_only, gs, blk, 0, st, cp, c.sc, c.slots, 0, 0, 5, (double *)c.rhs); });
        time_graph("omega_b", [&] { hipLaunchKernelGGL(k_omega_b<2>, gs, blk, 0, st, cp, c.sc, c.slots, 0, 0); });
        time_graph("PG full (V0)", [&] { hipLaunchKernelGGL(k_pgvar<0>, gs, blk, 0, st, cp, c.sc, c.rhs); });
        time_graph("PG cheap mixture weight (V1)", [&] { hipLaunchKernelGGL(k_pgvar<1>, gs, blk, 0, st, cp, c.sc, c.rhs); });
        time_graph("PG no series test (V2)", [&] { hipLaunchKernelGGL(k_pgvar<2>, gs, blk, 0, st, cp, c.sc, c.rhs); });
        time_graph("2 philox + 1 log (V3)", [&] { hipLaunchKernelGGL(k_pgvar<3>, gs, blk, 0, st, cp, c.sc, c.rhs); });
        time_graph("mass_texpon only (V4)", [&] { hipLaunchKernelGGL(k_pgvar<4>, gs, blk, 0, st, cp, c.sc, c.rhs); });
        time_graph("one BM normal (V5)", [&] { hipLaunchKernelGGL(k_pgvar<5>, gs, blk, 0, st, cp, c.sc, c.rhs); });
        time_graph("eta_init", [&] { hipLaunchKernelGGL(k_eta_init, gs, blk, 0, st, cp, c.sc, c.slots, 0, 0); });
        time_graph("noise", [&] { hipLaunchKernelGGL(k_noise, gs, blk, 0, st, cp, c.sc, c.slots, 0, 0, 1); });
        time_graph("omega_a", [&] { hipLaunchKernelGGL(k_omega_a<2>, gr, blk, 0, st, cp, c.sc, c.slots, 0, 0); });
#ifdef KB_EXTRA
        KB_EXTRA
#endif
    }
    return 0;
}
