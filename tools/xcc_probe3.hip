// Developer probe: CU sets enabled by the two stream masks the engine uses (bits [0,nmain) and [nmain,256)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
__global__ void probe(unsigned *out)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (hw & 0xff00);
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(64);
}
static std::set<unsigned> run(const std::vector<uint32_t> &m, unsigned *d, int nwg, int per_xcc[8])
{
    hipStream_t st;
    hipExtStreamCreateWithCUMask(&st, (uint32_t)m.size(), m.data());
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(nwg);
    hipMemcpy(h.data(), d, sizeof(unsigned) * nwg, hipMemcpyDeviceToHost);
    hipStreamDestroy(st);
    std::set<unsigned> s(h.begin(), h.end());
    for (int x = 0; x < 8; ++x) per_xcc[x] = 0;
    for (unsigned v : s) per_xcc[(v >> 16) & 7]++;
    return s;
}
int main(int argc, char **argv)
{
    const int nwg = 4096;
    unsigned *d; hipMalloc(&d, sizeof(unsigned) * nwg);
    for (int a = 1; a < argc; ++a) {
        const int nmain = atoi(argv[a]);
        std::vector<uint32_t> A(8, 0u), B(8, 0u);
        for (int i = 0; i < 256; ++i) (i < nmain ? A : B)[i / 32] |= 1u << (i % 32);
        int pa[8], pb[8];
        std::set<unsigned> sa = run(A, d, nwg, pa), sb = run(B, d, nwg, pb);
        int common = 0;
        for (unsigned v : sa) common += sb.count(v);
        printf("split %3d: main mask %zu CUs (per xcc:", nmain, sa.size());
        for (int x = 0; x < 8; ++x) printf(" %d", pa[x]);
        printf("), side mask %zu CUs (per xcc:", sb.size());
        for (int x = 0; x < 8; ++x) printf(" %d", pb[x]);
        printf("), common %d\n", common);
    }
    return 0;
}
