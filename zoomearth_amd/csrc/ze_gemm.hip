// bf16 MFMA GEMM for the prefill / ViT path:
//   C[M,N] = A[M,K] * W[N,K]^T  (+bias) (+epilogue), fp32 accumulate, ONE rounding to bf16.
// Both operands are K-contiguous ("TN"), which is the natural MFMA feed: a 16x16x32 fragment is 16 B per lane.
//
// Tile: BM x BN x 64, 256 threads = 4 waves (2 x 2), each wave owns (BM/2) x (BN/2) as 16x16 MFMA tiles.
// LDS image per operand: [k-chunk (8 x 16 B)][row][16 B] with the row XOR-swizzled by the chunk index
//   addr(row, chunk) = ((chunk * ROWS) + (row ^ (chunk & 7))) * 16
// -> the fragment read (16 rows of one chunk, ds_read_b128) touches 16 distinct 16-B slots (conflict-free) and
//    the staging write (8 lanes = 8 chunks of one row, ds_write_b128) covers all 32 banks (conflict-free).
// Staging is register based and split (issue global loads for tile t+1 before the MFMAs of tile t, write LDS after),
// two LDS buffers, one barrier per K-tile.
#include "ze_kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define GEMM_BK 64

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int BM, int BN, int EPI>
__global__ void __launch_bounds__(256) k_gemm_tn(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W,
                                                 int ldw, const bf16_t* __restrict__ bias,
                                                 const bf16_t* __restrict__ R, int ldr, bf16_t* __restrict__ C,
                                                 int ldc, const int* __restrict__ c_rows, int M, int N, int K,
                                                 int ksplit, float* __restrict__ slab, unsigned* __restrict__ tickets) {
    constexpr int TM = BM / 32, TN = BN / 32;        // MFMA tiles per wave
    constexpr int A_LOADS = BM * 8 / 256;            // uint4 per thread per K-tile
    constexpr int B_LOADS = BN * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint4* const sA0 = reinterpret_cast<uint4*>(smem);                // 2 x [8 chunks][BM rows]
    uint4* const sB0 = reinterpret_cast<uint4*>(smem) + 2 * BM * 8;   // 2 x [8 chunks][BN rows]

    // XCD-aware block remap: consecutive remapped ids (sharing the A row panel) land on one XCD's L2
    const int nbx = (N + BN - 1) / BN, nby = (M + BM - 1) / BM;
    const int nwg = nbx * nby;
    const int ks = blockIdx.x / nwg;  // split-K slice of this block (all slices of a tile share its output)
    int bid = blockIdx.x % nwg;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm0 = (bid / nbx) * BM, bn0 = (bid % nbx) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid >> 1) * (BM / 2), wn0 = (wid & 1) * (BN / 2);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[A_LOADS], rb[B_LOADS];
    const int ld_chunk = tid & 7;   // 16-B chunk along K
    const int ld_row = tid >> 3;    // 0..31

    auto load_tiles = [&](int k0) {
        const int kc = k0 + ld_chunk * 8;
        const bool kin = kc < K;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int row = bm0 + ld_row + i * 32;
            ra[i] = (kin && row < M) ? *reinterpret_cast<const uint4*>(A + (size_t)row * lda + kc) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int row = bn0 + ld_row + i * 32;
            rb[i] = (kin && row < N) ? *reinterpret_cast<const uint4*>(W + (size_t)row * ldw + kc) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int row = ld_row + i * 32;
            sA0[buf * BM * 8 + ld_chunk * BM + (row ^ ld_chunk)] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int row = ld_row + i * 32;
            sB0[buf * BN * 8 + ld_chunk * BN + (row ^ ld_chunk)] = rb[i];
        }
    };

    const int nk_all = (K + GEMM_BK - 1) / GEMM_BK;
    const int nk_per = (nk_all + ksplit - 1) / ksplit;
    const int kt0 = ks * nk_per;
    const int nk = max(0, min(nk_all - kt0, nk_per));
    load_tiles(kt0 * GEMM_BK);
    store_tiles(0);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles((kt0 + kt + 1) * GEMM_BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ks * 4 + fq;
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm0 + i * 16 + fr;
                const uint4 v = sA0[buf * BM * 8 + chunk * BM + (row ^ chunk)];
                fa[i] = *reinterpret_cast<const bf16x8*>(&v);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + j * 16 + fr;
                const uint4 v = sB0[buf * BN * 8 + chunk * BN + (row ^ chunk)];
                fb[j] = *reinterpret_cast<const bf16x8*>(&v);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---------------- split-K: deterministic reduction by the last-arriving slice of the tile.
    // Every slice parks its fp32 accumulators in a slab (one float4 per thread per MFMA tile), publishes them with an
    // agent-scope release + ticket; the block that draws the last ticket acquires, adds the slabs IN SLICE ORDER
    // (bit-reproducible, independent of arrival order) and runs the epilogue.  (cdna_hip_programming.md, Projection
    // GEMM item 2: one release + one acquire per tile episode, never __threadfence per call.)
    if (ksplit > 1) {
        // slabs go out WRITE-THROUGH (sc1 buffer stores): no L2 write-back fence is needed before the ticket
        // (publish-large: 3.0 us vs 8.2 us for plain stores + release fence at 64 KB per workgroup)
        {
            const size_t slab_bytes = (size_t)ksplit * nwg * 256 * (TM * TN) * 16;
            auto rsrc = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)min(slab_bytes, (size_t)0x7fffffff), 0x00020000);
            const unsigned base = (unsigned)((((size_t)(ks * nwg + bid) * 256 + tid) * (TM * TN)) * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    u32x4 v;
                    v.x = __float_as_uint(acc[i][j][0]);
                    v.y = __float_as_uint(acc[i][j][1]);
                    v.z = __float_as_uint(acc[i][j][2]);
                    v.w = __float_as_uint(acc[i][j][3]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, base + (unsigned)(i * TN + j) * 16, 0, 16 /* sc1 */);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the ticket
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem);  // the staging buffers are dead now
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(&tickets[bid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = (old == (unsigned)ksplit - 1u) ? 1u : 0u;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&tickets[bid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
            }
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0u) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < ksplit; ++q) {
            const float4* src = reinterpret_cast<const float4*>(slab) + ((size_t)(q * nwg + bid) * 256 + tid) * (TM * TN);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float4 v = src[i * TN + j];
                    acc[i][j][0] += v.x;
                    acc[i][j][1] += v.y;
                    acc[i][j][2] += v.z;
                    acc[i][j][3] += v.w;
                }
        }
    }

    // ---------------- epilogue: acc[i][j][r] -> row = wm0 + i*16 + fq*4 + r, col = wn0 + j*16 + fr
    if (EPI == ZE_EPI_SWIGLU) {
        // W rows are interleaved in blocks of 16: [gate 0..15 | up 0..15 | gate 16..31 | ...]; even j = gate, odd = up
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; j += 2) {
                const int ncol = bn0 + wn0 + j * 16 + fr;         // gate row in the packed weight
                const int ocol = (bn0 + wn0 + j * 16) / 2 + fr;   // output column
                if (ncol >= N) continue;
                const float bg = bias ? bf16_to_f32(bias[ncol]) : 0.f;
                const float bu = bias ? bf16_to_f32(bias[ncol + 16]) : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = bm0 + wm0 + i * 16 + fq * 4 + r;
                    if (row >= M) continue;
                    const float g = bf16_round(acc[i][j][r] + bg);
                    const float u = bf16_round(acc[i][j + 1][r] + bu);
                    C[(size_t)row * ldc + ocol] = f32_to_bf16(bf16_round(silu_f(g)) * u);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = bn0 + wn0 + j * 16 + fr;
            if (col >= N) continue;
            const float b = bias ? bf16_to_f32(bias[col]) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bm0 + wm0 + i * 16 + fq * 4 + r;
                if (row >= M) continue;
                float v = bf16_round(acc[i][j][r] + b);
                if (EPI == ZE_EPI_GELU) v = gelu_erf(v);
                if (EPI == ZE_EPI_RESIDUAL) v = bf16_to_f32(R[(size_t)row * ldr + col]) + v;
                const int orow = c_rows ? c_rows[row] : row;
                if (EPI == ZE_EPI_F32)  // fp32 copy of the bf16-rounded value (HF: logits.float())
                    reinterpret_cast<float*>(C)[(size_t)orow * ldc + col] = v;
                else
                    C[(size_t)orow * ldc + col] = f32_to_bf16(v);
            }
        }
}

// split-K workspace (fp32 slabs + per-tile tickets), owned by the engine and passed once
static float* g_slab = nullptr;
static unsigned* g_tickets = nullptr;
static size_t g_slab_floats = 0;
static int g_ticket_cap = 0;
void ze_gemm_set_workspace(float* slab, size_t slab_floats, unsigned* tickets, int ticket_cap) {
    g_slab = slab;
    g_slab_floats = slab_floats;
    g_tickets = tickets;
    g_ticket_cap = ticket_cap;
}

template <int BM, int BN>
static void launch_cfg(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias,
                       const bf16_t* R, int ldr, bf16_t* C, int ldc, const int* c_rows, int M, int N, int K,
                       hipStream_t s, bool stream_mode) {
    const int nwg = ze_cdiv(M, BM) * ze_cdiv(N, BN);
    // Split K only in weight-streaming mode (batched decode, few rows): the slice count is a function of (N, K)
    // alone -- computed for ONE row tile -- so a chain's result never depends on how many chains share the step.
    // Prefill / ViT never split: their accumulation order must not depend on M (prefix-KV reuse is bit-exact).
    int ksplit = 1;
    if (stream_mode) {
        const int nk = ze_cdiv(K, GEMM_BK);
        const int tiles_n = ze_cdiv(N, BN);
        while (tiles_n * ksplit < 200 && ksplit < 8 && nk / (ksplit * 2) >= 4) ksplit *= 2;
    }
    if (ksplit > 1 && (!g_slab || (size_t)ksplit * nwg * BM * BN > g_slab_floats || nwg > g_ticket_cap)) ksplit = 1;
    const int grid = nwg * ksplit;
    const size_t lds = (size_t)2 * (BM + BN) * 8 * 16;
#define ZE_GEMM_LAUNCH(E)                                                                                          \
    hipLaunchKernelGGL((k_gemm_tn<BM, BN, E>), dim3(grid), dim3(256), lds, s, A, lda, W, ldw, bias, R, ldr, C, ldc, \
                       c_rows, M, N, K, ksplit, g_slab, g_tickets)
    switch (epi) {
        case ZE_EPI_NONE: ZE_GEMM_LAUNCH(ZE_EPI_NONE); break;
        case ZE_EPI_GELU: ZE_GEMM_LAUNCH(ZE_EPI_GELU); break;
        case ZE_EPI_RESIDUAL: ZE_GEMM_LAUNCH(ZE_EPI_RESIDUAL); break;
        case ZE_EPI_SWIGLU: ZE_GEMM_LAUNCH(ZE_EPI_SWIGLU); break;
        case ZE_EPI_F32: ZE_GEMM_LAUNCH(ZE_EPI_F32); break;
    }
#undef ZE_GEMM_LAUNCH
}

void ze_launch_gemm_stream(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias,
                           const bf16_t* R, int ldr, bf16_t* C, int ldc, int M, int N, int K, hipStream_t s) {
    if (M <= 0 || N <= 0) return;
    launch_cfg<64, 64>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, nullptr, M, N, K, s, true);
}

void ze_launch_gemm(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R,
                    int ldr, bf16_t* C, int ldc, const int* c_rows, int M, int N, int K, hipStream_t s) {
    if (M <= 0 || N <= 0) return;
    // pick the largest tile that still gives the 256 CUs about one block each (split-K covers the rest)
    const long b128 = (long)ze_cdiv(M, 128) * ze_cdiv(N, 128);
    const long b64x128 = (long)ze_cdiv(M, 64) * ze_cdiv(N, 128);
    if (b128 >= 200)
        launch_cfg<128, 128>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, false);
    else if (b64x128 >= 160)
        launch_cfg<64, 128>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, false);
    else
        launch_cfg<64, 64>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, false);
}
