// Probe: where the time of the eight-phase 256 x 256 GEMM K loop goes (k_gemm_p8 of zoomearth_amd/csrc/ze_gemm.hip, its
// loop restated here with ABLATIONS selected by a template parameter; results of the ablated forms are wrong by design --
// only the clock is read).  P = 0: the loop as shipped; 1: one MFMA of each quadrant's sixteen; 2: no fragment reads inside the
// loop; 3: no LDS-DMA inside the loop; 4: no barrier behind a quadrant's MFMAs; 5: no s_setprio; 6: both wave rows in lockstep;
// 7: MFMAs only (no reads, no DMA); 8: reads + DMA + barriers, no MFMA at all; 9: the DMA pieces issued from inside the MFMA clusters
// (a candidate schedule, not an ablation).
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/p8_probe tools/probes/p8_probe.hip ; run: /tmp/p8_probe [M N K]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define GEMM_BK 64

template <int N>
__device__ __forceinline__ void ring_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int P>
__global__ void __launch_bounds__(512) k_p8(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
                                            float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 256, BN = 256, HALF = 128 * 128, BUF = 4 * HALF;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int nbx = (N + BN - 1) / BN, nby = (M + BM - 1) / BM;
    const int nwg = nbx * nby;
    const bool col_major = N > M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int nk = K / GEMM_BK;
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);
    constexpr bool INMFMA = P == 9;
    constexpr bool DMA = P != 3 && P != 7 && P != 9, READS = P != 2 && P != 7, BAR2 = P != 4, PRIO = P != 5, SKEW = P != 6;

    int bm0 = 0, bn0 = 0, bid = 0;
    auto place = [&](int tile) {
        const int q = nwg / 8, r = nwg % 8, xcd = tile % 8, idx = tile / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        bm0 = (col_major ? bid % nby : bid / nbx) * BM;
        bn0 = (col_major ? bid / nby : bid % nbx) * BN;
    };
    unsigned offA[2][2], offB[2][2];
    auto sources = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int hr = (wid * 2 + q) * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((hr >> 1) & 7);
                const int ra = min(bm0 + (hr >> 6) * 128 + h * 64 + (hr & 63), M - 1);
                const int rb = min(bn0 + (hr >> 5) * 64 + h * 32 + (hr & 31), N - 1);
                offA[h][q] = (unsigned)(((size_t)ra * lda + c * 8) * sizeof(bf16_t));
                offB[h][q] = (unsigned)(((size_t)rb * ldw + c * 8) * sizeof(bf16_t));
            }
    };
    auto stage_piece = [&](int slot, int t, int q) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_lds + (t & 1) * BUF + slot * HALF + wid * 2048);
        const bf16_t* base = (slot < 2 ? A : W) + (size_t)t * GEMM_BK;
        const unsigned off = slot == 0 ? offA[0][q] : slot == 1 ? offA[1][q] : slot == 2 ? offB[0][q] : offB[1][q];
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(off), "s"(dst + q * 1024), "s"(base)
                     : "memory");
    };
    auto stage = [&](int slot, int t) {
        stage_piece(slot, t, 0);
        stage_piece(slot, t, 1);
    };
    auto prologue = [&]() {
        stage(0, 0);
        stage(2, 0);
        stage(3, 0);
        stage(1, 0);
        if (nk > 1) {
            stage(0, 1);
            stage(2, 1);
            stage(3, 1);
        }
    };
    const int sw = (fr >> 1) & 7;
    const unsigned aoff0 = (wr * 64 + fr) * 128 + (((0 + fq) ^ sw) << 4), aoff1 = (wr * 64 + fr) * 128 + (((4 + fq) ^ sw) << 4);
    const unsigned boff0 = (wc * 32 + fr) * 128 + (((0 + fq) ^ sw) << 4), boff1 = (wc * 32 + fr) * 128 + (((4 + fq) ^ sw) << 4);
    bf16x8 fa[2][4], fb0[2][2], fb1[2][2];
    f32x4 acc[8][4];
    auto read_a = [&](int b, int mh) {
        const uint8_t* p = smem + b * BUF + mh * HALF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[0][i] = *reinterpret_cast<const bf16x8*>(p + aoff0 + i * 2048);
            fa[1][i] = *reinterpret_cast<const bf16x8*>(p + aoff1 + i * 2048);
        }
    };
    auto read_b = [&](bf16x8 (&f)[2][2], int b, int nh) {
        const uint8_t* p = smem + b * BUF + (2 + nh) * HALF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f[0][j] = *reinterpret_cast<const bf16x8*>(p + boff0 + j * 2048);
            f[1][j] = *reinterpret_cast<const bf16x8*>(p + boff1 + j * 2048);
        }
    };
    // (P = 9: the phase's two DMA pieces go out from INSIDE the cluster of MFMAs -- behind the 4th and the 10th -- instead of in
    //  front of the phase's first barrier; slot < 0: nothing to stage)
    auto quadrant = [&](int mh, int nh, const bf16x8 (&f)[2][2], int slot = -1, int tt = 0) {
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (P == 8) continue;
                    if (P == 1 && (kk | i | j)) continue;
                    acc[mh * 4 + i][nh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][i], f[kk][j], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
                    if (P == 9 && slot >= 0) {
                        const int n = kk * 8 + i * 2 + j;
                        if (n == 3) stage_piece(slot, tt, 0);
                        if (n == 9) stage_piece(slot, tt, 1);
                    }
                }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    // keeps fragment reads alive where no MFMA consumes them
    auto sink = [&](const bf16x8& v) { asm volatile("" ::"v"(v)); };

    int tile = blockIdx.x;
    if (tile >= nwg) return;
    place(tile);
    sources();
    prologue();
    for (;;) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tile == (int)blockIdx.x && nk > 1) ring_wait<6>();
        else ring_wait<0>();
        __builtin_amdgcn_s_barrier();
        if (SKEW && wr == 1) __builtin_amdgcn_s_barrier();
        if (!READS) {
            read_a(0, 0);
            read_b(fb0, 0, 0);
            read_b(fb1, 0, 1);
        }

        for (int t = 0; t < nk; ++t) {
            const int b = t & 1;
            // ---- phase 1
            if (READS) {
                read_a(b, 0);
                __builtin_amdgcn_sched_barrier(0);
                read_b(fb0, b, 0);
            }
            if (DMA && t + 1 < nk) stage(1, t + 1);
            if (READS) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            quadrant(0, 0, fb0, (INMFMA && t + 1 < nk) ? 1 : -1, t + 1);
            if (P == 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sink(fa[0][i]), sink(fa[1][i]);
                sink(fb0[0][0]), sink(fb0[0][1]), sink(fb0[1][0]), sink(fb0[1][1]);
            }
            if (BAR2) __builtin_amdgcn_s_barrier();
            // ---- phase 2
            if (READS) read_b(fb1, b, 1);
            if (DMA && t + 2 < nk) stage(0, t + 2);
            __builtin_amdgcn_s_barrier();
            quadrant(0, 1, fb1, (INMFMA && t + 2 < nk) ? 0 : -1, t + 2);
            if (P == 8) sink(fb1[0][0]), sink(fb1[0][1]), sink(fb1[1][0]), sink(fb1[1][1]);
            if (BAR2) __builtin_amdgcn_s_barrier();
            // ---- phase 3
            if (READS) read_a(b, 1);
            if (DMA && t + 2 < nk) stage(2, t + 2);
            __builtin_amdgcn_s_barrier();
            quadrant(1, 1, fb1, (INMFMA && t + 2 < nk) ? 2 : -1, t + 2);
            if (P == 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sink(fa[0][i]), sink(fa[1][i]);
            }
            if (BAR2) __builtin_amdgcn_s_barrier();
            // ---- phase 4
            if (DMA) {
                if (t + 2 < nk) {
                    stage(3, t + 2);
                    ring_wait<6>();
                } else {
                    ring_wait<0>();
                }
            }
            if (INMFMA) {  // outstanding at this point: A-half 1 (t+1), A-half 0 (t+2), W-half 0 (t+2); W-half 1 (t+1) has to be back
                if (t + 2 < nk) ring_wait<6>();
                else ring_wait<0>();
            }
            __builtin_amdgcn_s_barrier();
            quadrant(1, 0, fb0, (INMFMA && t + 2 < nk) ? 3 : -1, t + 2);
            if (BAR2) __builtin_amdgcn_s_barrier();
        }
        if (SKEW && wr == 0) __builtin_amdgcn_s_barrier();
        const int done_bm0 = bm0, done_bn0 = bn0;
        tile += gridDim.x;
        const bool more = tile < nwg;
        if (more) {
            place(tile);
            sources();
            prologue();
        }
        // minimal epilogue: one value per lane (keeps the accumulators alive, no store traffic to speak of)
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        const int row = done_bm0 + wr * 128 + fq * 4, col = done_bn0 + wc * 64 + fr;
        if (row < M && col < N) C[(size_t)row * N + col] = sacc;
        if (!more) break;
    }
}

template <int P>
static float run(const bf16_t* A, const bf16_t* W, float* C, int M, int N, int K, int iters) {
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
    const int grid = tiles < 256 ? tiles : 256;
    const size_t lds = 128 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_p8<P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_p8<P>), dim3(grid), dim3(512), lds, 0, A, K, W, K, C, M, N, K);
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_p8<P>), dim3(grid), dim3(512), lds, 0, A, K, W, K, C, M, N, K);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 12832, N = argc > 2 ? atoi(argv[2]) : 22016, K = argc > 3 ? atoi(argv[3]) : 2048;
    bf16_t *A, *W;
    float* C;
    CHECK(hipMalloc(&A, (size_t)M * K * 2));
    CHECK(hipMalloc(&W, (size_t)N * K * 2));
    CHECK(hipMalloc(&C, (size_t)M * N * 4));
    std::vector<bf16_t> h((size_t)(M > N ? M : N) * K);
    unsigned s = 12345u;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        const float f = ((s >> 8) & 0xffff) / 65536.0f - 0.5f;   // uniform in [-0.5, 0.5): ordinary activations, not a power virus
        unsigned u;
        memcpy(&u, &f, 4);
        v = (bf16_t)(u >> 16);
    }
    CHECK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    const double fl = 2.0 * M * N * K;
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
    const double ktiles_per_wg = (double)tiles / (tiles < 256 ? tiles : 256) * (K / 64);
    const char* names[] = {"as shipped", "1 MFMA of 16 per quadrant", "no fragment reads in the loop", "no LDS-DMA in the loop",
                           "no barrier behind the MFMAs", "no s_setprio", "wave rows in lockstep", "MFMAs only", "no MFMA (reads + DMA + barriers)",
                           "DMA pieces from inside the MFMA clusters"};
#define RUN(P)                                                                                                                 \
    do {                                                                                                                       \
        const float us = run<P>(A, W, C, M, N, K, 10);                                                                         \
        printf("P=%d %-34s %9.1f us  %7.1f TFLOP/s-equivalent  %6.3f us per K-tile per workgroup\n", P, names[P], us,          \
               fl / us / 1e6, us / ktiles_per_wg);                                                                             \
    } while (0)
    printf("M=%d N=%d K=%d: %d tiles, %.1f K-tiles per workgroup\n", M, N, K, tiles, ktiles_per_wg);
    RUN(0); RUN(1); RUN(2); RUN(3); RUN(4); RUN(5); RUN(6); RUN(7); RUN(8); RUN(9); RUN(0); RUN(9); RUN(0); RUN(9);
    return 0;
}
