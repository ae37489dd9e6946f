// Developer probe: what v_mfma_f64_16x16x4_f64 sustains on this device with nothing else in the loop (no loads, no LDS),
// the ceiling any f64 matrix kernel here is measured against (k_rsr_gram*).   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak mfma_f64_peak.hip
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(1024) k_peak(double *out, int iters, double x)
{
    v4d acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
    double a = x + threadIdx.x, b = x - threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    double s = 0.0;
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if (s == 12345.678) out[0] = s;
}
// the same with k_rsr_gram64's other instructions of a slice: MULS f64 multiplies feeding every group of four MFMAs and,
// LDS != 0, their five operands read from LDS first
template <int MULS, int LDS>
__global__ void __launch_bounds__(1024) k_mix(double *out, int iters, double x)
{
    __shared__ double sh[1024 * 2];
    v4d acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
    sh[threadIdx.x] = x + threadIdx.x;
    sh[1024 + threadIdx.x] = x - threadIdx.x;
    __syncthreads();
    double a0 = x + threadIdx.x, a1 = x * 2, b0 = x - threadIdx.x, b1 = x * 3, w = 1.0 + 1e-9 * x;
    const int lane = threadIdx.x & 63, base = (threadIdx.x >> 6) * 64;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (LDS) {
                const int o = (i + g) & 7;
                a0 = sh[base + ((lane + o) & 63)]; a1 = sh[1024 + base + ((lane + o) & 63)];
                b0 = sh[base + ((lane + 2 * o) & 63)]; b1 = sh[1024 + base + ((lane + 2 * o) & 63)];
                w = sh[base + (o & 63)];
            }
            double y0 = b0, y1 = b1;
            if (MULS) { y0 = b0 * w; y1 = b1 * w; }
            acc[4 * g + 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, y0, acc[4 * g + 0], 0, 0, 0);
            acc[4 * g + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, y0, acc[4 * g + 1], 0, 0, 0);
            acc[4 * g + 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, y1, acc[4 * g + 2], 0, 0, 0);
            acc[4 * g + 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, y1, acc[4 * g + 3], 0, 0, 0);
            if (MULS && !LDS) { w = w * 1.0000001; b0 = b0 + 1e-9; }
        }
    }
    double s = 0.0;
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if (s == 12345.678) out[0] = s;
}
template <int MULS, int LDS>
static void run_mix(int wgs, int threads, int iters, double *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix<MULS, LDS>), dim3(wgs), dim3(threads), 0, 0, d, 10, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<MULS, LDS>), dim3(wgs), dim3(threads), 0, 0, d, iters, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)wgs * (threads / 64) * iters * 8 * 2048.0;
    std::printf("%4d workgroups x %4d threads, groups of 4 MFMAs, %s%s: %8.3f ms, %6.1f TFLOP/s\n", wgs, threads, MULS ? "2 f64 multiplies per group" : "no multiplies",
                LDS ? ", 5 operands from LDS per group" : "", ms, flop / ms * 1e-9);
}
// ... and with NI 32-bit integer vector instructions per group of four MFMAs (address arithmetic, register copies)
template <int NI>
__global__ void __launch_bounds__(1024) k_int(double *out, int iters, double x)
{
    v4d acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
    double a0 = x + threadIdx.x, a1 = x * 2, b0 = x - threadIdx.x, b1 = x * 3;
    unsigned u[4] = {threadIdx.x, threadIdx.x * 3u, threadIdx.x * 5u, threadIdx.x * 7u};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            acc[4 * g + 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[4 * g + 0], 0, 0, 0);
            acc[4 * g + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[4 * g + 1], 0, 0, 0);
            acc[4 * g + 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[4 * g + 2], 0, 0, 0);
            acc[4 * g + 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[4 * g + 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NI; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k & 3]) : "v"(u[(k + 1) & 3]));
        }
    }
    double s = (double)(u[0] ^ u[1] ^ u[2] ^ u[3]);
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if (s == 12345.678) out[0] = s;
}
template <int NI>
static void run_int(int wgs, int threads, int iters, double *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_int<NI>), dim3(wgs), dim3(threads), 0, 0, d, 10, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_int<NI>), dim3(wgs), dim3(threads), 0, 0, d, iters, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)wgs * (threads / 64) * iters * 8 * 2048.0;
    std::printf("%4d workgroups x %4d threads, groups of 4 MFMAs, %2d v_add_u32 per group: %8.3f ms, %6.1f TFLOP/s\n", wgs, threads, NI, ms, flop / ms * 1e-9);
}
// ... and with NL global loads (8 bytes per lane, L1-resident lines) per group of four MFMAs, their values feeding the next group
template <int NL, int WIDE>
__global__ void __launch_bounds__(1024) k_ld(double *out, const double *src, int iters, double x)
{
    v4d acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
    double a0 = x + threadIdx.x, a1 = x * 2, b0 = x - threadIdx.x, b1 = x * 3;
    const char *base = reinterpret_cast<const char *>(src);
    unsigned off = (threadIdx.x & 63) * (WIDE ? 16u : 8u);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            double v[NL > 0 ? NL : 1];
            if (WIDE) {
#pragma unroll
                for (int k = 0; k < NL; k += 2) {
                    typedef double v2 __attribute__((ext_vector_type(2)));
                    const v2 t = *reinterpret_cast<const v2 *>(base + off + 1024 * k);
                    v[k] = t[0]; v[k + 1] = t[1];
                }
            } else {
#pragma unroll
                for (int k = 0; k < NL; ++k) v[k] = *reinterpret_cast<const double *>(base + off + 512 * k);
            }
            acc[4 * g + 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[4 * g + 0], 0, 0, 0);
            acc[4 * g + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[4 * g + 1], 0, 0, 0);
            acc[4 * g + 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[4 * g + 2], 0, 0, 0);
            acc[4 * g + 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[4 * g + 3], 0, 0, 0);
            if (NL > 0) { a0 = v[0]; b0 = v[NL - 1]; }
            if (NL > 2) { a1 = v[1]; b1 = v[NL - 2]; }
        }
    }
    double s = 0.0;
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if (s == 12345.678) out[0] = s;
}
template <int NL, int WIDE>
static void run_ld(int wgs, int threads, int iters, double *d, const double *src)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_ld<NL, WIDE>), dim3(wgs), dim3(threads), 0, 0, d, src, 10, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_ld<NL, WIDE>), dim3(wgs), dim3(threads), 0, 0, d, src, iters, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)wgs * (threads / 64) * iters * 8 * 2048.0;
    std::printf("%4d workgroups x %4d threads, groups of 4 MFMAs, %d doubles per lane loaded per group by %s: %8.3f ms, %6.1f TFLOP/s\n", wgs, threads, NL,
                WIDE ? "global_load_dwordx4" : "global_load_dwordx2", ms, flop / ms * 1e-9);
}
template <int NACC>
static void run(int wgs, int threads, int iters, double *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_peak<NACC>, dim3(wgs), dim3(threads), 0, 0, d, 10, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_peak<NACC>, dim3(wgs), dim3(threads), 0, 0, d, iters, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)wgs * (threads / 64) * iters * NACC * 2048.0;
    std::printf("%4d workgroups x %4d threads, %d independent accumulators per wave: %8.3f ms, %6.1f TFLOP/s\n", wgs, threads, NACC, ms, flop / ms * 1e-9);
}
int main()
{
    double *d;
    hipMalloc(&d, 64);
    double *src;
    hipMalloc(&src, 1 << 16);
    hipMemset(src, 0, 1 << 16);
    for (int rep = 0; rep < 1; ++rep) {
        run<8>(256, 1024, 20000, d);   // 4 waves per SIMD
        run<8>(512, 512, 20000, d);    // 4 waves per SIMD, two workgroups per CU
        run<8>(256, 256, 80000, d);    // 1 wave per SIMD
        run<4>(256, 1024, 40000, d);
        run<1>(256, 1024, 160000, d);  // one accumulator: every MFMA waits for the one before
        run<8>(2560, 1024, 2000, d);   // ten rounds of workgroups
        run_ld<4, 0>(256, 1024, 20000, d, src);
        run_ld<6, 0>(256, 1024, 20000, d, src);
        run_ld<4, 1>(256, 1024, 20000, d, src);
        run_ld<8, 0>(256, 1024, 20000, d, src);
        run_ld<8, 1>(256, 1024, 20000, d, src);
        run_int<4>(256, 1024, 20000, d);
        run_int<8>(256, 1024, 20000, d);
        run_int<16>(256, 1024, 20000, d);
        run_int<32>(256, 1024, 20000, d);
        run_int<16>(256, 512, 40000, d);
        run_mix<0, 0>(256, 1024, 20000, d);
        run_mix<1, 0>(256, 1024, 20000, d);
        run_mix<1, 1>(256, 1024, 20000, d);
        run_mix<1, 1>(256, 512, 40000, d);
        run_mix<1, 1>(256, 256, 80000, d);
    }
    hipFree(d);
    return 0;
}
