// Developer probe: which (XCC, SE, CU) does CU-mask bit b enable?   ./tools/xcc_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
int main()
{
    unsigned *d; hipMalloc(&d, sizeof(unsigned) * 2 * 16);
    unsigned h[32];
    const int bits[] = {0, 1, 2, 3, 4, 7, 8, 9, 15, 16, 31, 32, 33, 63, 64, 127, 128, 255};
    for (int b : bits) {
        std::vector<uint32_t> m(8, 0u);
        m[b / 32] |= 1u << (b % 32);
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, 8, m.data()) != hipSuccess) { printf("bit %d: mask failed\n", b); continue; }
        hipLaunchKernelGGL(probe, dim3(16), dim3(64), 0, st, d);
        hipStreamSynchronize(st);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("bit %3d ->", b);
        for (int i = 0; i < 16; ++i) { bool dup = false; for (int j = 0; j < i; ++j) dup |= (h[2*i] == h[2*j] && (h[2*i+1] & 0xff00) == (h[2*j+1] & 0xff00)); if (!dup) printf(" (xcc %u se %u sh %u cu %u)", h[2*i], (h[2*i+1] >> 13) & 7, (h[2*i+1] >> 12) & 1, (h[2*i+1] >> 8) & 15); }
        printf("\n");
        hipStreamDestroy(st);
    }
    return 0;
}
