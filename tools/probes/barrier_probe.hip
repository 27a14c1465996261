// Probe: cost of in-kernel grid barriers on MI355X (256 blocks, 1 per CU), flat counter vs XCD-hierarchical.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/barrier_probe tools/probes/barrier_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned int u32;

__device__ __forceinline__ u32 ld_relaxed(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 xcc_id() { u32 v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

// flat: one monotonic counter
__device__ __forceinline__ bool bar_flat(u32* ctr, u32 target, u32* timeout) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u32 spins = 0;
        while (ld_relaxed(ctr) < target) { __builtin_amdgcn_s_sleep(1); if (++spins > 4000000u) { *timeout = 1; ok = false; break; } }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return ok;
}

// hierarchical: per-XCD arrival counter -> leader arrives at top -> leader publishes per-XCD generation
struct hbar { u32 xcd_cnt[8 * 32]; u32 xcd_gen[8 * 32]; u32 top[32]; };  // one word per 128-B line
__device__ __forceinline__ bool bar_xcd(hbar* b, u32 xcc, u32 per_xcd, u32 nxcd, u32 epoch, u32* timeout) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u32 old = __hip_atomic_fetch_add(&b->xcd_cnt[xcc * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u32 spins = 0;
        if (old == epoch * per_xcd - 1) {  // last of this XCD in this epoch: leader
            __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_relaxed(&b->top[0]) < epoch * nxcd) { __builtin_amdgcn_s_sleep(1); if (++spins > 4000000u) { *timeout = 2; ok = false; break; } }
            __hip_atomic_store(&b->xcd_gen[xcc * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_relaxed(&b->xcd_gen[xcc * 32]) < epoch) { __builtin_amdgcn_s_sleep(1); if (++spins > 4000000u) { *timeout = 3; ok = false; break; } }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return ok;
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// fence-free hierarchical barrier: the payload is published write-through (sc1) and read with sc1 loads, so the
// barrier itself is only atomics + polls (no buffer_wbl2 / buffer_inv)
template <int SLEEP>
__device__ __forceinline__ bool bar_xcd_nofence(hbar* b, u32 xcc, u32 per_xcd, u32 nxcd, u32 epoch, u32* timeout) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's sc1 stores have left
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const u32 old = __hip_atomic_fetch_add(&b->xcd_cnt[xcc * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u32 spins = 0;
        if (old == epoch * per_xcd - 1) {
            __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_relaxed(&b->top[0]) < epoch * nxcd) { if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); if (++spins > 4000000u) { *timeout = 2; ok = false; break; } }
            __hip_atomic_store(&b->xcd_gen[xcc * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_relaxed(&b->xcd_gen[xcc * 32]) < epoch) { if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); if (++spins > 4000000u) { *timeout = 3; ok = false; break; } }
        }
    }
    __syncthreads();
    return ok;
}
template <int SLEEP>
__global__ void __launch_bounds__(256) k_xcd_nf(hbar* b, const u32* census, int iters, u32* timeout, float* data, u32* mism) {
    const u32 xcc = xcc_id();
    const u32 per = census[xcc];
    u32 nx = 0;
    for (int i = 0; i < 8; ++i) nx += census[i] > 0;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(data, 0, 256 * 256 * 4, 0x00020000);
    for (int i = 0; i < iters; ++i) {
        // publish 4 B per thread with an sc1 dword store
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)(i + 1)), rsrc, (blockIdx.x * 256 + threadIdx.x) * 4, 0, 16);
        if (!bar_xcd_nofence<SLEEP>(b, xcc, per, nx, (u32)(2 * i + 1), timeout)) return;
        const int other = (blockIdx.x * 37 + 11 + i) % gridDim.x;
        const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (other * 256 + threadIdx.x) * 4, 0, 16));
        if (v < (float)(i + 1)) atomicAdd(mism, 1u);
        if (!bar_xcd_nofence<SLEEP>(b, xcc, per, nx, (u32)(2 * i + 2), timeout)) return;
    }
}

__global__ void __launch_bounds__(256) k_flat(u32* ctr, int iters, u32* timeout, float* data) {
    for (int i = 0; i < iters; ++i) {
        data[blockIdx.x * 256 + threadIdx.x] += 1.0f;  // something to publish
        if (!bar_flat(ctr, (u32)(i + 1) * gridDim.x, timeout)) return;
    }
}
__global__ void __launch_bounds__(256) k_census(u32* census, u32* xcc_of_block) {
    if (threadIdx.x == 0) { const u32 x = xcc_id(); xcc_of_block[blockIdx.x] = x; atomicAdd(&census[x], 1u); }
}
__global__ void __launch_bounds__(256) k_xcd(hbar* b, const u32* census, int iters, u32* timeout, float* data, u32* mism) {
    const u32 xcc = xcc_id();
    const u32 per = census[xcc];
    u32 nx = 0;
    for (int i = 0; i < 8; ++i) nx += census[i] > 0;
    for (int i = 0; i < iters; ++i) {
        data[blockIdx.x * 256 + threadIdx.x] = (float)(i + 1);
        if (!bar_xcd(b, xcc, per, nx, (u32)(2 * i + 1), timeout)) return;
        // every block checks another block's slot (visibility test)
        const int other = (blockIdx.x * 37 + 11 + i) % gridDim.x;
        const float v = __builtin_nontemporal_load(&data[other * 256 + threadIdx.x]);
        if (v < (float)(i + 1)) atomicAdd(mism, 1u);
        if (!bar_xcd(b, xcc, per, nx, (u32)(2 * i + 2), timeout)) return;
    }
}

int main() {
    const int nb = 256, iters = 200;
    u32 *ctr, *timeout, *census, *xcc_of, *mism; float* data; hbar* hb;
    CHECK(hipMalloc(&ctr, 256)); CHECK(hipMalloc(&timeout, 4)); CHECK(hipMalloc(&census, 64)); CHECK(hipMalloc(&xcc_of, nb * 4));
    CHECK(hipMalloc(&mism, 4)); CHECK(hipMalloc(&data, nb * 256 * 4)); CHECK(hipMalloc(&hb, sizeof(hbar)));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(ctr, 0, 256)); CHECK(hipMemset(timeout, 0, 4)); CHECK(hipMemset(data, 0, nb * 256 * 4));
        CHECK(hipEventRecord(a)); k_flat<<<nb, 256>>>(ctr, iters, timeout, data); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); u32 to; CHECK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
        printf("flat counter barrier: %.2f us per barrier (timeout flag %u)\n", ms * 1000 / iters, to);
    }
    CHECK(hipMemset(census, 0, 64)); k_census<<<nb, 256>>>(census, xcc_of); CHECK(hipDeviceSynchronize());
    u32 hc[16]; CHECK(hipMemcpy(hc, census, 64, hipMemcpyDeviceToHost));
    printf("census per XCC:"); for (int i = 0; i < 8; ++i) printf(" %u", hc[i]); printf("\n");
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(hb, 0, sizeof(hbar))); CHECK(hipMemset(timeout, 0, 4)); CHECK(hipMemset(mism, 0, 4));
        CHECK(hipEventRecord(a)); k_xcd<<<nb, 256>>>(hb, census, iters, timeout, data, mism); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); u32 to, mm; CHECK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&mm, mism, 4, hipMemcpyDeviceToHost));
        printf("xcd-hierarchical barrier: %.2f us per barrier (2 per iter; timeout flag %u, stale reads %u)\n", ms * 1000 / (2 * iters), to, mm);
    }
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(hb, 0, sizeof(hbar))); CHECK(hipMemset(timeout, 0, 4)); CHECK(hipMemset(mism, 0, 4)); CHECK(hipMemset(data, 0, nb * 256 * 4));
        CHECK(hipEventRecord(a)); k_xcd_nf<1><<<nb, 256>>>(hb, census, iters, timeout, data, mism); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); u32 to, mm; CHECK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&mm, mism, 4, hipMemcpyDeviceToHost));
        printf("xcd barrier, no fences, sc1 payload, sleep1: %.2f us per barrier (timeout %u, stale reads %u)\n", ms * 1000 / (2 * iters), to, mm);
    }
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(hb, 0, sizeof(hbar))); CHECK(hipMemset(timeout, 0, 4)); CHECK(hipMemset(mism, 0, 4)); CHECK(hipMemset(data, 0, nb * 256 * 4));
        CHECK(hipEventRecord(a)); k_xcd_nf<0><<<nb, 256>>>(hb, census, iters, timeout, data, mism); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); u32 to, mm; CHECK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&mm, mism, 4, hipMemcpyDeviceToHost));
        printf("xcd barrier, no fences, sc1 payload, no sleep: %.2f us per barrier (timeout %u, stale reads %u)\n", ms * 1000 / (2 * iters), to, mm);
    }
    return 0;
}
