// Probe: what bounds the operand intake of the GEMM kernels?  Every GEMM of the library stages its operands by LDS-DMA at 7-10 TB/s
// chip-wide whatever its K loop looks like (tools/probes/ring_ablate.sh: gate/up of a decode step takes 57.6 of 63.1 us with the
// MFMAs and the fragment reads compiled out).  This probe runs ONLY the intake -- 256 workgroups of 512 threads, a four-stage ring
// of 32-KB stages, counted vmcnt + one barrier per stage, no arithmetic -- over access patterns that separate the candidates:
//   where the bytes come from   shared: every workgroup reads the same 2 MB (L2 hits after the first pass)
//                               xcd:    the workgroups of an XCD (blockIdx % 8) share a 2-MB region (L2 hits, 16 MB in all)
//                               mall:   every workgroup its own 2 MB of a 512-MB buffer that fits the Infinity Cache
//                               hbm:    every workgroup its own 2 MB of an 8-GB buffer, a new one per launch
//   the shape of a 1-KB piece   rows128: 8 rows x 128 B, rows 4096 B apart (a K-step of 64 of a row-major operand, K = 2048)
//                               rows64:  16 rows x 64 B, rows 4096 B apart (K-steps of 32)
//                               rows128p: 8 rows x 128 B, rows 4352 B apart (17 x 256: no power-of-two stride)
//                               flat:    1024 contiguous bytes (a fragment-major / K-blocked operand)
//   the path                    dma: global_load_lds_dwordx4;  reg: global_load_dwordx4 into registers (no LDS)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/ingest_probe tools/probes/ingest_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int STAGE = 32 * 1024, STAGES = 4, WAVES = 8, PPW = STAGE / 1024 / WAVES;  // 4 pieces per wave per stage

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// shape: 0 rows128, 1 rows64, 2 rows128p, 3 flat.  The region of a workgroup is `region` bytes; step s, piece p cover 1 KB of it.
__device__ __forceinline__ const char* piece_addr(const char* base, int shape, int step, int piece, int lane) {
    // piece index within the stage: 0 .. 31; a stage = 32 KB = (256 rows x 128 B) or (512 rows x 64 B) of a [rows][K] operand
    if (shape == 3) return base + ((size_t)step * 32 + piece) * 1024 + lane * 16;
    if (shape == 1) {  // 16 rows x 64 B: row = piece * 16 + lane / 4, K offset = step * 64 B
        const int row = piece * 16 + (lane >> 2);
        return base + (size_t)row * 4096 + (size_t)(step % 64) * 64 + (lane & 3) * 16;
    }
    const size_t stride = shape == 2 ? 4352 : 4096;
    const int row = piece * 8 + (lane >> 3);      // 8 rows x 128 B: K offset = step * 128 B (32 steps cover a 4-KB row)
    return base + (size_t)row * stride + (size_t)(step % 32) * 128 + (lane & 7) * 16;
}

template <bool DMA>
__global__ void __launch_bounds__(512) k_ingest(const char* __restrict__ buf, size_t region_stride, int share_mod, int shape, int steps,
                                                float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int reg = share_mod > 0 ? (int)(blockIdx.x % share_mod) : (int)blockIdx.x;
    const char* base = buf + (size_t)reg * region_stride;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    float acc = 0.f;
    auto issue = [&](int s) {
#pragma unroll
        for (int g = 0; g < PPW; ++g) {
            const int piece = wid + g * WAVES;
            const char* src = piece_addr(base, shape, s, piece, lane);
            if constexpr (DMA) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(src), "s"(lds0 + (unsigned)(s % STAGES) * STAGE + (unsigned)piece * 1024u)
                             : "memory");
            } else {
                const float4 v = *reinterpret_cast<const float4*>(src);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    };
    if constexpr (DMA) {
        for (int s = 0; s < STAGES - 1 && s < steps; ++s) issue(s);
        for (int s = 0; s < steps; ++s) {
            const int ahead = min(STAGES - 2, steps - 1 - s);
            if (ahead >= 2) wait_vm<2 * PPW>();
            else if (ahead == 1) wait_vm<PPW>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            if (s + STAGES - 1 < steps) issue(s + STAGES - 1);
        }
    } else {
        for (int s = 0; s < steps; ++s) issue(s);
    }
    if (acc == 12345.678f) sink[blockIdx.x * 512 + tid] = acc + smem[tid];
}

// ---- the GEMM-like MIX (gate/up of a decode step on 320 x 192 tiles, K-steps of 64): per stage 320 rows x 128 B of A, which EVERY
// workgroup of a row tile reads (L2 hits: the activations are 2.4 MB in all), and 192 rows x 128 B of W, which two workgroups read
// (HBM, a new weight matrix per launch).  64 KB per stage, `nst` stages in the ring (nst x 64 KB of LDS), 32 stages = K 2048.
//   split = 0: every wave issues pieces of both operands (the ring kernels): one in-order queue per wave for hits and misses
//   split = 1: waves 0-4 issue the A pieces, waves 5-7 the W pieces: a wave's wait covers one kind only
//   pf > 0:    a ninth wave touches one dword of every W line `pf` stages ahead (a prefetch into L2; it keeps no data)
template <int NST>
__global__ void __launch_bounds__(576) k_mix(const char* __restrict__ abuf, const char* __restrict__ wbuf, int split, int pf, int steps,
                                             float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SB = 64 * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* abase = abuf + (size_t)(blockIdx.x & 1) * (320 * 4096);              // two row tiles of A
    const char* wbase = wbuf + (size_t)(blockIdx.x >> 1) * (192 * 4096);             // the column tile's W rows (shared by the pair)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    if (wid == 8) {  // the prefetch wave: 192 W rows x 128 B per stage = 192 lines = three wave-instructions of one dword per lane,
        // requested `pf` stages ahead of the loaders (it joins every barrier of the workgroup; its loads are never waited for)
        unsigned r0 = 0, r1 = 0, r2 = 0;
        for (int s = -pf; s < steps; ++s) {
            const int sp = s + pf;
            if (pf > 0 && sp < steps) {
                const char* p0 = wbase + (size_t)lane * 4096 + (size_t)sp * 128;
                const char* p1 = p0 + (size_t)64 * 4096;
                const char* p2 = p0 + (size_t)128 * 4096;
                asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %4, off\n\tglobal_load_dword %2, %5, off"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2)
                             : "v"(p0), "v"(p1), "v"(p2)
                             : "memory");
            }
            if (s >= 0) asm volatile("s_barrier" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (r0 + r1 + r2 == 0x12345678u) sink[blockIdx.x] = 1.f;
        return;
    }
    // pieces of a stage: 40 of A (320 rows / 8) + 24 of W (192 rows / 8) = 64
    auto issue_piece = [&](int s, int pc) {
        const bool is_a = pc < 40;
        const int row = (is_a ? pc : pc - 40) * 8 + (lane >> 3);
        const char* src = (is_a ? abase : wbase) + (size_t)row * 4096 + (size_t)s * 128 + (lane & 7) * 16;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(lds0 + (unsigned)(s % NST) * SB + (unsigned)pc * 1024u)
                     : "memory");
    };
    // pieces per wave per stage: split 0 -> 8 each (64 / 8); split 1 -> waves 0-4: 8 A pieces each, waves 5-7: 8 W pieces each
    auto issue = [&](int s) {
#pragma unroll
        for (int g = 0; g < 8; ++g) issue_piece(s, split ? wid * 8 + g : wid + g * 8);
    };
    for (int s = 0; s < NST - 1 && s < steps; ++s) issue(s);
    for (int s = 0; s < steps; ++s) {
        const int ahead = min(NST - 2, steps - 1 - s);
        switch (ahead) {
            case 6: wait_vm<48>(); break;
            case 5: wait_vm<40>(); break;
            case 4: wait_vm<32>(); break;
            case 3: wait_vm<24>(); break;
            case 2: wait_vm<16>(); break;
            case 1: wait_vm<8>(); break;
            default: wait_vm<0>(); break;
        }
        // (the barrier of the eight loading waves only: the prefetch wave runs free)
        asm volatile("s_barrier" ::: "memory");
        if (s + NST - 1 < steps) issue(s + NST - 1);
    }
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    const size_t big = (size_t)8 << 30;
    char* buf;
    float* sink;
    CHECK(hipMalloc(&buf, big + (8 << 20)));
    CHECK(hipMalloc(&sink, 1 << 20));
    CHECK(hipMemset(buf, 1, big + (8 << 20)));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ingest<true>), hipFuncAttributeMaxDynamicSharedMemorySize, STAGE * STAGES));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const char* shapes[] = {"rows128", "rows64", "rows128p", "flat"};
    struct Src { const char* name; size_t stride; int mod; size_t span; } srcs[] = {
        {"shared", 0, 1, 0}, {"xcd", (size_t)2 << 20, 8, 0}, {"mall", (size_t)2 << 20, 0, (size_t)512 << 20}, {"hbm", (size_t)2 << 20, 0, big}};
    const int steps = 64;  // 64 stages x 32 KB = 2 MB per workgroup (rows shapes: 256 rows x 4 KB walked twice / 512 rows x 4 KB once)
    for (int dma = 1; dma >= 0; --dma)
        for (const Src& sc : srcs)
            for (int shape = 0; shape < 4; ++shape) {
                const size_t region = (size_t)2 << 20;
                const size_t per_launch = sc.mod == 0 ? (size_t)wgs * sc.stride : 0;
                const int rot = per_launch ? (int)(sc.span / per_launch) : 1;
                auto launch = [&](int it) {
                    const char* base = buf + (per_launch ? (size_t)(it % (rot > 0 ? rot : 1)) * per_launch : 0);
                    if (dma) hipLaunchKernelGGL(k_ingest<true>, dim3(wgs), dim3(512), STAGE * STAGES, 0, base, sc.stride, sc.mod, shape, steps, sink);
                    else hipLaunchKernelGGL(k_ingest<false>, dim3(wgs), dim3(512), 0, 0, base, sc.stride, sc.mod, shape, steps, sink);
                };
                for (int i = 0; i < 3; ++i) launch(i);
                CHECK(hipDeviceSynchronize());
                const int iters = 20;
                CHECK(hipEventRecord(a, 0));
                for (int i = 0; i < iters; ++i) launch(3 + i);
                CHECK(hipEventRecord(b, 0));
                CHECK(hipEventSynchronize(b));
                float ms;
                CHECK(hipEventElapsedTime(&ms, a, b));
                const double us = ms * 1000.0 / iters, bytes = (double)wgs * region;
                printf("%s %-7s %-9s: %7.1f us per launch, %6.2f TB/s chip-wide, %5.1f GB/s per workgroup\n", dma ? "dma" : "reg", sc.name, shapes[shape], us,
                       bytes / us / 1e6, bytes / us / 1e3 / wgs);
                fflush(stdout);
            }
    // ---- the mix
    {
        const int wg = 230, stepsm = 32;
        const size_t wbytes = (size_t)115 * 192 * 4096;  // 90 MB of "weights" per launch, rotated through the 8-GB buffer
        char* abuf = buf + big - ((size_t)4 << 20);
        auto run = [&](auto kern, int lds, int split, int pf, const char* tag) {
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            auto launch = [&](int it) { hipLaunchKernelGGL(kern, dim3(wg), dim3(576), lds, 0, abuf, buf + (size_t)(it % 40) * wbytes, split, pf, stepsm, sink); };
            for (int i = 0; i < 3; ++i) launch(i);
            CHECK(hipDeviceSynchronize());
            const int iters = 20;
            CHECK(hipEventRecord(a, 0));
            for (int i = 0; i < iters; ++i) launch(3 + i);
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            const double us = ms * 1000.0 / iters, bytes = (double)wg * stepsm * 65536;
            printf("mix %-28s: %6.1f us per launch, %5.2f TB/s staged (%.0f MB: 2 x 90 MB of W + 115 x 2.6 MB of A)\n", tag, us, bytes / us / 1e6, bytes / 1e6);
            fflush(stdout);
        };
        run(k_mix<2>, 2 * 65536, 0, 0, "2 stages, mixed waves");
        run(k_mix<2>, 2 * 65536, 1, 0, "2 stages, split waves");
        run(k_mix<2>, 2 * 65536, 0, 4, "2 stages, mixed, prefetch 4");
        run(k_mix<2>, 2 * 65536, 1, 4, "2 stages, split, prefetch 4");
        run(k_mix<2>, 2 * 65536, 0, 8, "2 stages, mixed, prefetch 8");
        run(k_mix<2>, 2 * 65536, 0, 16, "2 stages, mixed, prefetch 16");
    }
    return 0;
}
