// Developer probe: what does a barrier + neighbour exchange among the workgroups of one chain cost when the chain's
// workgroups all sit on ONE XCD and talk through that XCD's L2 (sc0: L1 bypass only), against today's placement
// (chain = blockIdx.y, spread over the eight XCDs, agent-scope sc1 traffic through the memory side)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/xcc_probe4 tools/xcc_probe4.hip && tools/xcc_probe4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int NBG = 40, WG = 256, STEPS = 2000;
constexpr unsigned SPIN_LIMIT = 1u << 20;

struct Out {
    unsigned xcc[8 * NBG];
    unsigned long long cycles[8];
    unsigned bad[8];
    unsigned timeout[8];
};

// LOCAL = 1: chain = linear id % 8 (expected XCD), sc0 / workgroup scope.  LOCAL = 0: chain = blockIdx.y, sc1 / agent scope.
template <int LOCAL, int ATOM_LOCAL, int POLL_LOCAL, int DATA_AUX, int INV = 0>
__global__ void __launch_bounds__(WG) exchange(v4u *data, unsigned *cnt, Out *out, int nchain)
{
    __shared__ int s_fail;
    const int id = blockIdx.y * gridDim.x + blockIdx.x;
    int chain, wg;
    if (LOCAL) { chain = id & 7; wg = id >> 3; } else { chain = blockIdx.y; wg = blockIdx.x; }
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
    if (threadIdx.x == 0) out->xcc[chain * NBG + wg] = xcc;
    if (chain >= nchain) return;
    constexpr int AUX = DATA_AUX;
    v4u *mine = data + ((size_t)chain * NBG + wg) * WG, *next = data + ((size_t)chain * NBG + (wg + 1) % NBG) * WG;
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)mine, 0, WG * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void *)next, 0, WG * 16, 0x00020000);
    unsigned *c = cnt + chain * 32;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)c, 0, 4, 0x00020000);
    unsigned bad = 0;
    if (threadIdx.x == 0) s_fail = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int k = 1; k <= STEPS; ++k) {
        v4u v = {(unsigned)k, (unsigned)wg, threadIdx.x, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(v, rm, threadIdx.x * 16, 0, AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (ATOM_LOCAL) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)k * NBG;
            unsigned spins = 0;
            for (;;) {
                // workgroup scope would be served by this CU's L1: the poll must bypass it (sc0), nothing more
                unsigned inv_now = 0;
                if (INV) asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(inv_now) : "v"(c) : "memory");
                const unsigned now = INV ? inv_now
                                   : POLL_LOCAL ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rc, 0, 0, 1)
                                           : __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(now - target) >= 0) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { s_fail = 1; break; }
            }
        }
        __syncthreads();
        if (s_fail) break;
        if (INV) asm volatile("buffer_inv sc1" ::: "memory");
        const v4u r = __builtin_amdgcn_raw_buffer_load_b128(rn, threadIdx.x * 16, 0, AUX);
        bad += (r.x != (unsigned)k && r.x != (unsigned)k + 1u);  // the neighbour may already have stored its next step
        // the neighbour must not overwrite before everyone has read: a second barrier, as the solver's next step has
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (bad) atomicAdd(&out->bad[chain], bad);
    if (threadIdx.x == 0 && s_fail) atomicAdd(&out->timeout[chain], 1u);
    if (threadIdx.x == 0 && wg == 0) out->cycles[chain] = t1 - t0;
}

// One XCD per chain; data: plain (or sc1) stores, sc1 loads; barrier: one flag word per workgroup, plain (or sc1) store,
// the first wave polls all flags with sc1 loads.
template <int STORE_AUX>
__global__ void __launch_bounds__(WG) exchange_flags(v4u *data, unsigned *flags, Out *out, int nchain, int local)
{
    __shared__ int s_fail;
    const int id = blockIdx.y * gridDim.x + blockIdx.x;
    int chain, wg;
    if (local) { chain = id & 7; wg = id >> 3; } else { chain = blockIdx.y; wg = blockIdx.x; }
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
    if (threadIdx.x == 0) out->xcc[chain * NBG + wg] = xcc;
    if (chain >= nchain) return;
    v4u *mine = data + ((size_t)chain * NBG + wg) * WG, *next = data + ((size_t)chain * NBG + (wg + 1) % NBG) * WG;
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)mine, 0, WG * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void *)next, 0, WG * 16, 0x00020000);
    unsigned *f = flags + chain * 64;
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)f, 0, 64 * 4, 0x00020000);
    unsigned bad = 0;
    if (threadIdx.x == 0) s_fail = 0;
    for (int k = 1; k <= STEPS; ++k) {
        v4u v = {(unsigned)k, (unsigned)wg, threadIdx.x, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(v, rm, threadIdx.x * 16, 0, STORE_AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x < 64) {
            if (threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)k, rf, wg * 4, 0, STORE_AUX);
            unsigned spins = 0;
            for (;;) {
                const unsigned now = threadIdx.x < NBG ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rf, threadIdx.x * 4, 0, 16) : (unsigned)k;
                if (__all((int)(now - (unsigned)k) >= 0)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { s_fail = 1; break; }
            }
        }
        __syncthreads();
        if (s_fail) break;
        const v4u r = __builtin_amdgcn_raw_buffer_load_b128(rn, threadIdx.x * 16, 0, 16);
        bad += (r.x != (unsigned)k && r.x != (unsigned)k + 1u);
        __syncthreads();
    }
    if (bad) atomicAdd(&out->bad[chain], bad);
    if (threadIdx.x == 0 && s_fail) atomicAdd(&out->timeout[chain], 1u);
}

int main()
{
    v4u *data; unsigned *cnt; Out *out;
    hipMalloc(&data, sizeof(v4u) * 8 * NBG * WG);
    hipMalloc(&cnt, sizeof(unsigned) * 8 * 32);
    hipMalloc(&out, sizeof(Out));
    hipStream_t st; hipStreamCreate(&st);
    for (int local = 0; local < 13; ++local)
        for (int nchain : {4}) {
            hipMemset(cnt, 0, sizeof(unsigned) * 8 * 32);
            hipMemset(out, 0, sizeof(Out));
            hipMemset(data, 0, sizeof(v4u) * 8 * NBG * WG);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, st);
            if (local == 1) hipLaunchKernelGGL((exchange<1, 1, 1, 1>), dim3(NBG, 8), dim3(WG), 0, st, data, cnt, out, nchain);
            else if (local == 2) hipLaunchKernelGGL((exchange<1, 0, 1, 1>), dim3(NBG, 8), dim3(WG), 0, st, data, cnt, out, nchain);
            else if (local == 3) hipLaunchKernelGGL((exchange<1, 1, 0, 1>), dim3(NBG, 8), dim3(WG), 0, st, data, cnt, out, nchain);
            else if (local == 4) hipLaunchKernelGGL((exchange<1, 0, 0, 1>), dim3(NBG, 8), dim3(WG), 0, st, data, cnt, out, nchain);
            else if (local == 9) hipLaunchKernelGGL((exchange_flags<0>), dim3(NBG, 8), dim3(WG), 0, st, data, cnt, out, nchain, 1);
            else if (local == 10) hipLaunchKernelGGL((exchange_flags<16>), dim3(NBG, 8), dim3(WG), 0, st, data, cnt, out, nchain, 1);
            else if (local == 11) hipLaunchKernelGGL((exchange_flags<16>), dim3(NBG, nchain), dim3(WG), 0, st, data, cnt, out, nchain, 0);
            else if (local == 12) hipLaunchKernelGGL((exchange_flags<0>), dim3(NBG, nchain), dim3(WG), 0, st, data, cnt, out, nchain, 0);
            else if (local == 7) hipLaunchKernelGGL((exchange<1, 1, 0, 0, 1>), dim3(NBG, 8), dim3(WG), 0, st, data, cnt, out, nchain);
            else if (local == 8) hipLaunchKernelGGL((exchange<1, 1, 0, 1, 1>), dim3(NBG, 8), dim3(WG), 0, st, data, cnt, out, nchain);
            else if (local == 5) hipLaunchKernelGGL((exchange<1, 0, 0, 16>), dim3(NBG, 8), dim3(WG), 0, st, data, cnt, out, nchain);
            else if (local == 6) hipLaunchKernelGGL((exchange<1, 1, 0, 16>), dim3(NBG, 8), dim3(WG), 0, st, data, cnt, out, nchain);
            else hipLaunchKernelGGL((exchange<0, 0, 0, 16>), dim3(NBG, nchain), dim3(WG), 0, st, data, cnt, out, nchain);
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            Out h; hipMemcpy(&h, out, sizeof(Out), hipMemcpyDeviceToHost);
            const char *names[13] = {"chain = blockIdx.y, all sc1/agent", "one XCD per chain, data sc0, atomic wg-scope, poll sc0", "one XCD per chain, data sc0, atomic agent, poll sc0",
                                    "one XCD per chain, data sc0, atomic wg-scope, poll agent", "one XCD per chain, data sc0, atomic agent, poll agent",
                                    "one XCD per chain, data sc1, atomic agent, poll agent", "one XCD per chain, data sc1, atomic wg-scope, poll agent",
                                    "one XCD per chain, plain data + buffer_inv sc0, wg-scope atomics for arrive and poll", "one XCD per chain, sc0 data + buffer_inv sc0, wg-scope atomics for arrive and poll",
                                    "one XCD per chain, PLAIN stores, sc1 loads, flag per workgroup", "one XCD per chain, sc1 stores, sc1 loads, flag per workgroup",
                                    "chain = blockIdx.y, sc1 stores, sc1 loads, flag per workgroup", "chain = blockIdx.y, PLAIN stores, sc1 loads, flag per workgroup (expected stale)"};
            unsigned hc[8 * 32]; hipMemcpy(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost);
            printf("%s; %d chains of %d workgroups: %.3f us per exchange step (kernel %.2f ms); counter of chain 0 = %u\n", names[local], nchain, NBG,
                   1e3 * ms / STEPS, ms, hc[0]);
            for (int ch = 0; ch < 1; ++ch) {
                int hist[8] = {0};
                for (int w = 0; w < NBG; ++w) hist[h.xcc[ch * NBG + w] & 7]++;
                printf("  chain %d: workgroups per XCD", ch);
                for (int x = 0; x < 8; ++x) printf(" %d", hist[x]);
                printf("; stale reads %u, timeouts %u\n", h.bad[ch], h.timeout[ch]);
            }
        }
    return 0;
}
