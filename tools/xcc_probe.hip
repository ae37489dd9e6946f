// Developer probe: which XCD / CU does workgroup i of a 1-D grid land on, with and without a CU mask on the stream?
//   hipcc --offload-arch=gfx950 -O2 -o tools/xcc_probe tools/xcc_probe.hip && ./tools/xcc_probe 128 4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void probe(unsigned *out, int spin)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;   // HW_REG_XCC_ID
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);           // HW_REG_HW_ID: cu [11:8], sh [12], se [15:13]
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);              // keep the workgroups co-resident
}
int main(int argc, char **argv)
{
    const int nwg = argc > 1 ? atoi(argv[1]) : 128, nxcd = argc > 2 ? atoi(argv[2]) : 4, threads = argc > 3 ? atoi(argv[3]) : 320;
    unsigned *d; hipMalloc(&d, sizeof(unsigned) * 2 * nwg);
    std::vector<unsigned> h(2 * nwg);
    for (int masked = 0; masked < 2; ++masked) {
        hipStream_t st;
        if (masked) {
            std::vector<uint32_t> m(8, 0u);
            const bool whole = argc > 4;  // 5th argument: enable whole XCCs 0..nxcd-1 assuming bit i -> XCC i % 8
            for (int i = 0; i < 256; ++i)
                if (whole ? (i % 8) < nxcd : i < nxcd * 32) m[i / 32] |= 1u << (i % 32);
            if (hipExtStreamCreateWithCUMask(&st, 8, m.data()) != hipSuccess) { printf("mask failed\n"); return 1; }
        } else hipStreamCreate(&st);
        hipLaunchKernelGGL(probe, dim3(nwg), dim3(threads), 0, st, d, 2000);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * nwg, hipMemcpyDeviceToHost);
        printf("%s: wg -> xcc:", masked ? "masked" : "unmasked");
        for (int i = 0; i < nwg && i < 48; ++i) printf(" %u", h[2 * i]);
        printf("\n");
        int cnt[16] = {0};
        for (int i = 0; i < nwg; ++i) cnt[h[2 * i] & 15]++;
        printf("  workgroups per xcc:");
        for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
        // distinct (xcc, se, sh, cu) tuples
        int distinct = 0;
        for (int i = 0; i < nwg; ++i) { bool dup = false; for (int j = 0; j < i; ++j) if (h[2*i] == h[2*j] && (h[2*i+1] & 0xff00) == (h[2*j+1] & 0xff00)) dup = true; distinct += !dup; }
        printf("   distinct CUs used: %d of %d workgroups\n", distinct, nwg);
        hipStreamDestroy(st);
    }
    return 0;
}
