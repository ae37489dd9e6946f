// Developer probe (round 3; DESIGN 7, HISTORY.md): when do two streams of one process stop running BESIDE each other?
// A kernel on stream B waits (bounded) for a word that a kernel launched AFTER it on stream A sets.  Streams that
// are served one after the other (one hardware queue, or queues the scheduler time-slices) never see the word.
//   part 1: K live pairs of CU-masked streams (the engine's kind), K = 1 .. kmax: probe the newest and the oldest pair
//   part 2: all but the first pair destroyed: probe again
//   part 3: the same with priority streams (high / low) and with plain non-blocking streams of equal priority
// Prints one line per step.  Every wait is bounded (~20 ms).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_wait(unsigned *w)
{
    if (threadIdx.x != 0) return;
    unsigned seen = 0u, spins = 0u;
    for (; spins < (1u << 13); ++spins) {
        if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { seen = 1u; break; }
        __builtin_amdgcn_s_sleep(16);
    }
    w[16] = seen;
    w[17] = spins;
}
__global__ void k_set(unsigned *w)
{
    if (threadIdx.x == 0) __hip_atomic_store(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Pair { hipStream_t a = nullptr, b = nullptr; };

static unsigned *g_w = nullptr;

// -> seen (1 / 0), spins, host microseconds
static void probe(const Pair &p, unsigned *seen, unsigned *spins, double *us)
{
    hipMemset(g_w, 0, 128);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, p.b, g_w);
    hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, p.a, g_w);
    hipStreamSynchronize(p.a);
    hipStreamSynchronize(p.b);
    *us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    unsigned h[32];
    hipMemcpy(h, g_w, sizeof(h), hipMemcpyDeviceToHost);
    *seen = h[16];
    *spins = h[17];
}

static Pair make_masked(int nmain)
{
    std::vector<uint32_t> A(8, 0u), B(8, 0u);
    for (int i = 0; i < 256; ++i) (i / 8 < nmain / 8 ? A : B)[i / 32] |= 1u << (i % 32);
    Pair p;
    if (hipExtStreamCreateWithCUMask(&p.a, 8, A.data()) != hipSuccess) p.a = nullptr;
    if (hipExtStreamCreateWithCUMask(&p.b, 8, B.data()) != hipSuccess) p.b = nullptr;
    return p;
}
static Pair make_prio(bool same)
{
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    Pair p;
    hipStreamCreateWithPriority(&p.a, hipStreamNonBlocking, same ? lo : hi);
    hipStreamCreateWithPriority(&p.b, hipStreamNonBlocking, lo);
    return p;
}

static void sweep(const char *name, int kmax, int kind)
{
    std::vector<Pair> ps;
    int first_bad_new = -1, first_bad_old = -1;
    for (int k = 1; k <= kmax; ++k) {
        Pair p = kind == 0 ? make_masked(160) : make_prio(kind == 2);
        if (!p.a || !p.b) { std::printf("%s: stream creation failed at pair %d: %s\n", name, k, hipGetErrorString(hipGetLastError())); break; }
        ps.push_back(p);
        unsigned sn, cn, so, co;
        double un, uo;
        probe(ps.back(), &sn, &cn, &un);
        probe(ps.front(), &so, &co, &uo);
        std::printf("%s: %2d live pairs: newest pair seen=%u spins=%u %.0f us | oldest pair seen=%u spins=%u %.0f us\n", name, k, sn, cn, un, so, co, uo);
        if (!sn && first_bad_new < 0) first_bad_new = k;
        if (!so && first_bad_old < 0) first_bad_old = k;
    }
    std::printf("%s: first failure of the newest pair at %d live pairs, of the oldest at %d (-1: never)\n", name, first_bad_new, first_bad_old);
    // cross probes with everything alive: wait on pair i's side, set on pair j's main
    if (ps.size() >= 6) {
        for (size_t i : {size_t(0), ps.size() / 2, ps.size() - 1})
            for (size_t j : {size_t(1), ps.size() - 2}) {
                Pair x;
                x.a = ps[j].a;
                x.b = ps[i].b;
                unsigned s, c;
                double u;
                probe(x, &s, &c, &u);
                std::printf("%s: cross probe wait on pair %zu, set on pair %zu: seen=%u spins=%u\n", name, i, j, s, c);
            }
    }
    while (ps.size() > 1) {
        hipStreamDestroy(ps.back().a);
        hipStreamDestroy(ps.back().b);
        ps.pop_back();
    }
    hipDeviceSynchronize();
    if (!ps.empty()) {
        unsigned s, c;
        double u;
        probe(ps[0], &s, &c, &u);
        std::printf("%s: all but the first pair destroyed: seen=%u spins=%u %.0f us\n", name, s, c, u);
        Pair p = kind == 0 ? make_masked(160) : make_prio(kind == 2);
        probe(p, &s, &c, &u);
        std::printf("%s: a fresh pair after the destruction: seen=%u spins=%u %.0f us\n", name, s, c, u);
        hipStreamDestroy(p.a); hipStreamDestroy(p.b);
        hipStreamDestroy(ps[0].a); hipStreamDestroy(ps[0].b);
    }
}

int main(int argc, char **argv)
{
    const int kmax = argc > 1 ? std::atoi(argv[1]) : 40;
    hipSetDevice(0);
    hipMalloc((void **)&g_w, 128);
    const char *mq = std::getenv("GPU_MAX_HW_QUEUES");
    std::printf("GPU_MAX_HW_QUEUES=%s\n", mq ? mq : "(unset)");
    sweep("cu-masked", kmax, 0);
    sweep("priority hi/lo", kmax, 1);
    sweep("equal priority", kmax, 2);
    return 0;
}
