// kb_boundary.hip -- developer probe: are the LAST stores of a kernel visible to the FIRST loads of the next kernel of the same
// stream, whatever XCD wrote and whatever XCD reads?  (Plain launches and a captured graph; hipcc -O3 --offload-arch=gfx950.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_write(double *out, int n, double tag, int spin)
{
    // stagger the blocks' ends; the store is the block's last act
    unsigned long long t0 = wall_clock64();
    const unsigned long long wait = (unsigned long long)((blockIdx.x * 2654435761u) % (unsigned)spin);
    while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(1);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = tag + 1e-9 * i;
}
__global__ void k_read(const double *in, int n, double tag, unsigned long long *bad, int late)
{
    if (late) { unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < 200ull) __builtin_amdgcn_s_sleep(1); }  // ~2 us at 100 MHz
    // the first thing the kernel does: read what the previous kernel wrote last (every block reads a slice of everything)
    unsigned long long miss = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (in[i] != tag + 1e-9 * i) ++miss;
    if (miss) atomicAdd(bad, miss);
}
int main(int argc, char **argv)
{
    const int n = 1 << 16, reps = argc > 1 ? atoi(argv[1]) : 5000, spin = 400;  // spin: up to ~4 us of stagger (100 MHz wall clock)
    double *buf; unsigned long long *bad;
    CK(hipMalloc(&buf, n * sizeof(double))); CK(hipMalloc(&bad, 2 * sizeof(unsigned long long)));
    CK(hipMemset(bad, 0, 16)); CK(hipMemset(buf, 0, n * sizeof(double)));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int late = 0; late < 2; ++late) {
        for (int r = 1; r <= reps; ++r) {
            hipLaunchKernelGGL(k_write, dim3(n / 256), dim3(256), 0, st, buf, n, (double)r, spin);
            hipLaunchKernelGGL(k_read, dim3(64), dim3(256), 0, st, buf, n, (double)r, bad + late, late);
        }
        CK(hipStreamSynchronize(st));
    }
    unsigned long long h[2]; CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    printf("plain launches: %d pairs, stale values seen by an immediate read %llu, by a read 2 us into the kernel %llu\n", reps, h[0], h[1]);
    // the same inside a captured graph (two pairs per graph, replayed)
    CK(hipMemset(bad, 0, 16));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_write, dim3(n / 256), dim3(256), 0, st, buf, n, 1.0, spin);
    hipLaunchKernelGGL(k_read, dim3(64), dim3(256), 0, st, buf, n, 1.0, bad, 0);
    hipLaunchKernelGGL(k_write, dim3(n / 256), dim3(256), 0, st, buf, n, 2.0, spin);
    hipLaunchKernelGGL(k_read, dim3(64), dim3(256), 0, st, buf, n, 2.0, bad, 0);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < reps / 2; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    printf("graph replays: %d pairs, stale values seen by an immediate read %llu\n", reps, h[0]);
    return 0;
}
