// Developer probe: which CU (XCC, shader engine, CU id) does bit i of a stream's CU mask enable?
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
    for (int i = 0; i < 20; ++i) __builtin_amdgcn_s_sleep(64);
}
int main()
{
    const int nwg = 64;
    unsigned *d; hipMalloc(&d, sizeof(unsigned) * nwg);
    for (int bit = 0; bit < 256; ++bit) {
        std::vector<uint32_t> m(8, 0u);
        m[bit / 32] |= 1u << (bit % 32);
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, 8, m.data()) != hipSuccess) { printf("bit %d: stream failed\n", bit); continue; }
        hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 0, st, d);
        hipStreamSynchronize(st);
        std::vector<unsigned> h(nwg);
        hipMemcpy(h.data(), d, sizeof(unsigned) * nwg, hipMemcpyDeviceToHost);
        hipStreamDestroy(st);
        std::set<unsigned> s(h.begin(), h.end());
        printf("bit %3d:", bit);
        for (unsigned v : s) printf(" xcc %u se %u sh %u cu %u", (v >> 16) & 15, (v >> 13) & 7, (v >> 12) & 1, (v >> 8) & 15);
        printf("\n");
    }
    return 0;
}
