// Developer probe: how is a grid dealt to the XCDs when the stream's CU mask gives them different numbers of CUs?
// Mask: 24 CUs on XCDs 0-3, 16 on XCDs 4-7 (the main stream of the scalar-wave form at 4 chains) and its complement.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out, int spin)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
    if (threadIdx.x == 0) out[blockIdx.x] = xcc;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}
static void run(const char *name, const std::vector<uint32_t> &m, int nwg, int tpb, int spin, unsigned *d)
{
    hipStream_t st;
    (void)hipExtStreamCreateWithCUMask(&st, (uint32_t)m.size(), m.data());
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(tpb), 0, st, d, spin);
    (void)hipStreamSynchronize(st);
    std::vector<unsigned> h(nwg);
    (void)hipMemcpy(h.data(), d, sizeof(unsigned) * nwg, hipMemcpyDeviceToHost);
    (void)hipStreamDestroy(st);
    int per[8] = {0}, match = 0;
    for (int i = 0; i < nwg; ++i) { per[h[i] & 7]++; match += ((int)h[i] == (i & 7)); }
    printf("%-16s %5d wg x %3d thr spin %4d: per xcc", name, nwg, tpb, spin);
    for (int x = 0; x < 8; ++x) printf(" %4d", per[x]);
    printf("   xcc == id %% 8 for %d\n", match);
}
int main()
{
    unsigned *d; (void)hipMalloc(&d, sizeof(unsigned) * 8192);
    std::vector<uint32_t> M(8, 0u), S(8, 0u), U(8, 0u);
    for (int i = 0; i < 256; ++i) {
        const int x = i % 8, j = i / 8;
        ((j < (x < 4 ? 24 : 16)) ? M : S)[i / 32] |= 1u << (i % 32);
        if (i < 160) U[i / 32] |= 1u << (i % 32);
    }
    for (int spin : {0, 20, 200}) {
        run("symmetric 160", U, 384, 256, spin, d);
        run("main 24/16", M, 384, 256, spin, d);
        run("main 24/16", M, 840, 64, spin, d);
        run("main 24/16", M, 4176, 64, spin, d);
        run("side 8/16", S, 384, 256, spin, d);
        run("side 8/16", S, 840, 64, spin, d);
        run("side 8/16", S, 4176, 64, spin, d);
    }
    return 0;
}
