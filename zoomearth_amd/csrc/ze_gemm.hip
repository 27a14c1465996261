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
#include <algorithm>
#include <type_traits>

#include "ze_kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define GEMM_BK 64

extern int ze_live_engines;    // ze_gemv.hip: engines alive in this process
extern int ze_gemv_knobs[24];  // [5]: 1 = no skinny kernel in the weight-streaming launcher; [6]: 0 = shipped policy, 1 = register-staged kernel everywhere, 2 = ring wherever it applies

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Tail shared by the GEMM kernels: optional split-K reduction (deterministic, by the last-arriving slice of the tile)
// and the epilogue.  acc[i][j][r] -> row = wm0 + i*16 + fq*4 + r, col = wn0 + j*16 + fr.
template <int BM, int BN, int EPI, int WM = 2, int WN = 2, bool LDS_EPI = false>
__device__ __forceinline__ void gemm_finish(f32x4 (&acc)[BM / (16 * WM)][BN / (16 * WN)], uint8_t* smem, const bf16_t* __restrict__ bias,
                                            const bf16_t* __restrict__ R, int ldr, bf16_t* __restrict__ C, int ldc,
                                            const int* __restrict__ c_rows, int M, int N, int ksplit, int ks, int bid,
                                            int nwg, int bm0, int bn0, float* __restrict__ slab,
                                            unsigned* __restrict__ tickets) {
    constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN), NT = 64 * WM * WN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid / WN) * (BM / WM), wn0 = (wid % WN) * (BN / WN);
    const int fr = lane & 15, fq = lane >> 4;
    // ---------------- split-K: deterministic reduction by the last-arriving slice of the tile.
    // Every slice parks its fp32 accumulators in a slab (one float4 per thread per MFMA tile) with write-through
    // (sc1) stores, drains, and takes a ticket; the block that draws the last ticket adds the slabs IN SLICE ORDER
    // (bit-reproducible, independent of arrival order) reading them with sc1 loads, and runs the epilogue.  No
    // release / acquire fence anywhere (cdna_hip_programming.md Guideline 16, the sc1 hand-off with a counter).
    if (ksplit > 1) {
        // slabs go out WRITE-THROUGH (sc1 buffer stores): no L2 write-back fence is needed before the ticket
        // (publish-large: 3.0 us vs 8.2 us for plain stores + release fence at 64 KB per workgroup)
        {
            const size_t slab_bytes = (size_t)ksplit * nwg * NT * (TM * TN) * 16;
            auto rsrc = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)min(slab_bytes, (size_t)0x7fffffff), 0x00020000);
            // slab of (slice, tile): [MFMA tile t][thread]: a wave-instruction moves 1 KiB of consecutive bytes (with the
            // thread-major order of round 2 -- TM * TN float4 per thread back to back -- every lane of a store or load hit
            // a line of its own: at 8 tiles per thread the reduction of the 256-row down projection took 43 us)
            const unsigned base = (unsigned)(((size_t)(ks * nwg + bid) * (TM * TN) * NT + tid) * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    u32x4 v;
                    v.x = __float_as_uint(acc[i][j][0]);
                    v.y = __float_as_uint(acc[i][j][1]);
                    v.z = __float_as_uint(acc[i][j][2]);
                    v.w = __float_as_uint(acc[i][j][3]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, base + (unsigned)(i * TN + j) * NT * 16, 0, 16 /* sc1 */);
                }
        }
        if (tickets == nullptr) return;  // slab-only launch: k_splitk_reduce adds the slices (ze_launch_gemm_wide, long K)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the ticket
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem);  // the staging buffers are dead now
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(&tickets[bid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = (old == (unsigned)ksplit - 1u) ? 1u : 0u;
            if (last)  // ready for the next launch
                __hip_atomic_store(&tickets[bid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0u) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the slabs were published write-through and are read with sc1 (L1-bypassing) loads: no acquire fence
        // (every load of the handed-off bytes is such a load; the ticket was taken after every wave's drain)
        {
            const size_t slab_bytes = (size_t)ksplit * nwg * NT * (TM * TN) * 16;
            auto rsrc = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)min(slab_bytes, (size_t)0x7fffffff), 0x00020000);
            if constexpr (TM * TN <= 4) {
                // every slice's slab requested up front (at most 8 slices: one memory round trip instead of one per
                // slice -- the reducer is the tail of the whole launch), then added in slice order as before
                u32x4 v[8][TM * TN];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const unsigned base =
                        (unsigned)(((size_t)(min(q, ksplit - 1) * nwg + bid) * (TM * TN) * NT + tid) * 16);
#pragma unroll
                    for (int t = 0; t < TM * TN; ++t)
                        v[q][t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + (unsigned)t * NT * 16, 0, 16);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < ksplit) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) {
                                acc[i][j][0] += __uint_as_float(v[q][i * TN + j].x);
                                acc[i][j][1] += __uint_as_float(v[q][i * TN + j].y);
                                acc[i][j][2] += __uint_as_float(v[q][i * TN + j].z);
                                acc[i][j][3] += __uint_as_float(v[q][i * TN + j].w);
                            }
                    }
            } else if constexpr (TM * TN <= 8) {
                // larger tiles per thread: the slabs of FOUR slices requested together (32 loads in flight per thread: two
                // memory round trips for eight slices instead of eight), still added in slice order
                for (int q0 = 0; q0 < ksplit; q0 += 4) {
                    u32x4 v[4][TM * TN];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned base =
                            (unsigned)(((size_t)(min(q0 + q, ksplit - 1) * nwg + bid) * (TM * TN) * NT + tid) * 16);
#pragma unroll
                        for (int t = 0; t < TM * TN; ++t)
                            v[q][t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + (unsigned)t * NT * 16, 0, 16);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (q0 + q < ksplit) {
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int j = 0; j < TN; ++j) {
                                    acc[i][j][0] += __uint_as_float(v[q][i * TN + j].x);
                                    acc[i][j][1] += __uint_as_float(v[q][i * TN + j].y);
                                    acc[i][j][2] += __uint_as_float(v[q][i * TN + j].z);
                                    acc[i][j][3] += __uint_as_float(v[q][i * TN + j].w);
                                }
                        }
                }
            } else {
                for (int q = 0; q < ksplit; ++q) {
                    const unsigned base = (unsigned)(((size_t)(q * nwg + bid) * (TM * TN) * NT + tid) * 16);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + (unsigned)(i * TN + j) * NT * 16, 0, 16);
                            acc[i][j][0] += __uint_as_float(v.x);
                            acc[i][j][1] += __uint_as_float(v.y);
                            acc[i][j][2] += __uint_as_float(v.z);
                            acc[i][j][3] += __uint_as_float(v.w);
                        }
                }
            }
        }
    }

    // ---------------- wide-store epilogue (LDS_EPI; c_rows unused): the MFMA result layout gives a lane four ROWS of
    // one column, so the plain epilogue below stores 2-byte values 32 B at a time (and loads the residual the same way): at
    // 256 rows it cost the decode projections 6-10 us.  Here the rounded values cross the (idle) staging LDS once and leave
    // as 16-byte row pieces; the arithmetic and its order of roundings are those of the plain epilogue, value for value.
    if constexpr (LDS_EPI) {
        constexpr int OW = (EPI == ZE_EPI_SWIGLU) ? BN / 2 : BN;           // output columns of the tile
        constexpr int EB = (EPI == ZE_EPI_F32) ? 4 : 2;                    // bytes per staged element
        constexpr int ROWB = OW * EB + 16;                                 // row pitch in LDS: 16 B of padding spread the banks
        static_assert((OW * EB) % 16 == 0, "tile width");
        __syncthreads();  // (split-K: the ticket flag lives in this LDS)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; j += (EPI == ZE_EPI_SWIGLU ? 2 : 1)) {
                const int ncol = bn0 + wn0 + j * 16 + fr;
                float b0 = 0.f, b1 = 0.f;
                if (bias && ncol < N) {
                    b0 = bf16_to_f32(bias[ncol]);
                    if (EPI == ZE_EPI_SWIGLU) b1 = bf16_to_f32(bias[ncol + 16]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm0 + i * 16 + fq * 4 + r;
                    if (EPI == ZE_EPI_SWIGLU) {
                        const float g = bf16_round(acc[i][j][r] + b0);
                        const float u = bf16_round(acc[i][j + 1][r] + b1);
                        *reinterpret_cast<bf16_t*>(smem + row * ROWB + ((wn0 + j * 16) / 2 + fr) * 2) = f32_to_bf16(bf16_round(silu_f(g)) * u);
                    } else if (EPI == ZE_EPI_F32) {
                        *reinterpret_cast<float*>(smem + row * ROWB + (wn0 + j * 16 + fr) * 4) = bf16_round(acc[i][j][r] + b0);
                    } else {
                        float v = bf16_round(acc[i][j][r] + b0);
                        if (EPI == ZE_EPI_GELU) v = gelu_erf(v);
                        *reinterpret_cast<bf16_t*>(smem + row * ROWB + (wn0 + j * 16 + fr) * 2) = f32_to_bf16(v);
                    }
                }
            }
        __syncthreads();
        constexpr int PPR = OW * EB / 16;  // 16-byte pieces per row
        const int oc0 = (EPI == ZE_EPI_SWIGLU) ? bn0 / 2 : bn0, on = (EPI == ZE_EPI_SWIGLU) ? N / 2 : N;
        for (int pc = tid; pc < BM * PPR; pc += NT) {
            const int lrow = pc / PPR, c16 = pc % PPR;
            const int row = bm0 + lrow;  // (bm0 = 0 for the tiles that take every row; the 192-row tiles have two row tiles)
            if (row >= M) continue;
            const int col = oc0 + c16 * (16 / EB);  // first output column of the piece
            if (col >= on) continue;
            uint4 v = *reinterpret_cast<const uint4*>(smem + lrow * ROWB + c16 * 16);
            if (EPI == ZE_EPI_RESIDUAL) {  // out = bf16(residual + value), per element as the plain epilogue
                const uint4 rr = *reinterpret_cast<const uint4*>(R + (size_t)row * ldr + col);
                const uint32_t* pv = reinterpret_cast<const uint32_t*>(&v);
                const uint32_t* pr = reinterpret_cast<const uint32_t*>(&rr);
                uint32_t o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float lo = __uint_as_float(pr[q] << 16) + __uint_as_float(pv[q] << 16);
                    const float hi = __uint_as_float(pr[q] & 0xffff0000u) + __uint_as_float(pv[q] & 0xffff0000u);
                    o[q] = (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
                }
                v = make_uint4(o[0], o[1], o[2], o[3]);
            }
            if (EPI == ZE_EPI_F32) *reinterpret_cast<uint4*>(reinterpret_cast<float*>(C) + (size_t)row * ldc + col) = v;
            else *reinterpret_cast<uint4*>(C + (size_t)row * ldc + col) = v;
        }
        return;
    }
    // ---------------- qkv projection of a decode step + M-RoPE + KV append (ze_kernels.h: ZE_EPI_QKV_ROPE): tiles j, j + 1 of a
    // wave hold dims d and d + 64 of one head (permuted weight rows); R carries the device-resident ze_qkv_epi
    if constexpr (EPI == ZE_EPI_QKV_ROPE) {
        static_assert(EPI != ZE_EPI_QKV_ROPE || TN % 2 == 0, "the rotate_half partners are two 16-column tiles of a wave");
        const ze_qkv_epi* __restrict__ qa = reinterpret_cast<const ze_qkv_epi*>(R);
        const ze_seq_dev* __restrict__ st = qa->st;
        const int* __restrict__ seq_ids = qa->seq_ids;
        const bf16_t* __restrict__ cosT = qa->cosT;
        const bf16_t* __restrict__ sinT = qa->sinT;
        const int heads = qa->heads, kvh = qa->kv_heads, max_ctx = qa->max_ctx;
        const size_t sstride = qa->cache_seq_stride;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int seq[4], ctx[4], pos[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = min(bm0 + wm0 + i * 16 + fq * 4 + r, M - 1);
                seq[r] = seq_ids[row];
                ctx[r] = st[seq[r]].ctx;
                pos[r] = ctx[r] + st[seq[r]].rope_delta;
            }
#pragma unroll
            for (int j = 0; j < TN; j += 2) {
                const int pcol = bn0 + wn0 + j * 16 + fr;  // permuted column of the first partner
                if (pcol >= N) continue;
                const int head = pcol >> 7, d = ((pcol & 127) >> 5) * 16 + fr;  // dims d and d + 64 of `head`
                const float b1 = bias ? bf16_to_f32(bias[pcol]) : 0.f, b2 = bias ? bf16_to_f32(bias[pcol + 16]) : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = bm0 + wm0 + i * 16 + fq * 4 + r;
                    if (row >= M) continue;
                    const float x1 = bf16_round(acc[i][j][r] + b1), x2 = bf16_round(acc[i][j + 1][r] + b2);
                    bf16_t o1, o2;
                    if (head < heads + kvh) {
                        const float c = bf16_to_f32(cosT[(size_t)pos[r] * 64 + d]), sn = bf16_to_f32(sinT[(size_t)pos[r] * 64 + d]);
                        o1 = f32_to_bf16(bf16_round(x1 * c) + bf16_round(-x2 * sn));
                        o2 = f32_to_bf16(bf16_round(x2 * c) + bf16_round(x1 * sn));
                    } else {
                        o1 = f32_to_bf16(x1);
                        o2 = f32_to_bf16(x2);
                    }
                    bf16_t* dst;
                    if (head < heads) dst = C + (size_t)row * ldc + head * 128;
                    else if (head < heads + kvh) dst = qa->kcache + seq[r] * sstride + ((size_t)(head - heads) * max_ctx + ctx[r]) * 128;
                    else dst = qa->vcache + seq[r] * sstride + ((size_t)(head - heads - kvh) * max_ctx + ctx[r]) * 128;
                    dst[d] = o1;
                    dst[d + 64] = o2;
                }
            }
        }
        return;
    }
    // ---------------- epilogue: acc[i][j][r] -> row = wm0 + i*16 + fq*4 + r, col = wn0 + j*16 + fr
    if (EPI == ZE_EPI_SWIGLU) {
        static_assert(EPI != ZE_EPI_SWIGLU || TN % 2 == 0, "SwiGLU pairs two 16-column tiles of a wave");
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

    // XCD-aware block remap: consecutive remapped ids land on one XCD's L2
    const int nbx = (N + BN - 1) / BN, nby = (M + BM - 1) / BM;
    const int nwg = nbx * nby;
    const int ks = blockIdx.x / nwg;  // split-K slice of this block (all slices of a tile share its output)
    int bid = blockIdx.x % nwg;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // Workgroups with consecutive remapped ids share an XCD (its L2).  Walk the tile grid so that they share the
    // panel of the LARGER operand: with N > M (weights larger than activations: every GEMM of this path) ids run
    // down a column of tiles, an XCD then streams 1/8 of W once and re-reads the small A from L2 / MALL; the other
    // order makes all eight XCDs pull the whole of W through the fabric (gate/up at M = 802: 723 MB against 117 MB).
    const bool col_major = N > M;
    const int bm0 = (col_major ? bid % nby : bid / nbx) * BM, bn0 = (col_major ? bid / nby : bid % nbx) * BN;

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

    gemm_finish<BM, BN, EPI>(acc, smem, bias, R, ldr, C, ldc, c_rows, M, N, ksplit, ks, bid, nwg, bm0, bn0, slab, tickets);
}

// ------------------------------------------------------------------ LDS-DMA ring kernel (K % 64 == 0)
// The prefill / ViT GEMMs have 300-1300 rows: a launch is one or two rounds of workgroups whose K loops are only
// 20-170 steps long, so the register-staged kernel above (one tile of prefetch, ~0.1 us of MFMA per step against
// 1-2 us of load latency) is latency-bound at 200-500 TFLOP/s.  This kernel keeps STAGES - 1 K-tiles in flight per
// workgroup with `global_load_lds` (no VGPR cost):
//   * stage image per operand: [rows][128 B], the 16-B chunk index XOR-ed with (row >> 1) & 7.  One wave-instruction
//     moves 8 full 128-B rows (lanes l -> row l >> 3, LDS chunk l & 7); LDS-DMA writes lane-linearly, so the swizzle
//     lives on the SOURCE address and, as the same involution, on the fragment read: 16 rows x one chunk by
//     ds_read_b128 then touch 16 distinct 16-B slots of the 256-B bank row (conflict-free).
//   * K loop: counted s_waitcnt vmcnt(N) (this wave's DMAs of the stage about to be read have landed), ONE raw
//     s_barrier (everybody's have, and everybody is done reading the stage about to be refilled), issue the DMAs of
//     stage kt + STAGES - 1, fragment reads + MFMA of stage kt.  No __syncthreads() and no ordinary global load inside
//     the loop (either would drain the DMA queue), all LDS in one array.
// Rows past M / N re-read the last valid row (their outputs are never stored).  Accumulation order per output
// element is that of the kernel above (same MFMA, same K order), so the two produce identical results.
template <int ROWS, int NT>
__device__ __forceinline__ void ring_issue(const bf16_t* __restrict__ G, int ld, int row0, int row_max, int k0,
                                           unsigned lds_img, int wid, int lane) {
    // ROWS / 8 wave-instructions per image, NT / 64 waves share them
#pragma unroll
    for (int g = 0; g < ROWS / 8 / (NT / 64); ++g) {
        const int grp = wid + g * (NT / 64);
        const int row = grp * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const bf16_t* src = G + (size_t)min(row0 + row, row_max) * ld + k0 + c * 8;
        // Issued from inline asm on purpose: for a DMA it knows about (__builtin_amdgcn_global_load_lds) hipcc puts
        // s_waitcnt vmcnt(0) in front of the next ds_read of the array, which drains the ring every K-step.  The
        // ordering DMA -> ds_read is provided by ring_wait + the barrier in the K loop instead.
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(src), "s"(lds_img + grp * 1024)
            : "memory");
    }
}

// one 1-KiB piece (8 rows x 128 B) of a stage image: group `grp` of operand G
__device__ __forceinline__ void ring_issue_one(const bf16_t* __restrict__ G, int ld, int row0, int row_max, int k0,
                                               unsigned lds_img, int grp, int lane) {
    const int row = grp * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    const bf16_t* src = G + (size_t)min(row0 + row, row_max) * ld + k0 + c * 8;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_img + grp * 1024)
        : "memory");
}

// K-steps of 32 (k_gemm_ring<..., BK = 32>): a stage image row is 64 B -- exactly the 16 x 32 operand of one MFMA per 16 rows --
// and a 1-KiB piece is 16 rows x 64 B of the image [A rows | W rows] (lane l -> row l >> 2, chunk l & 3).  The 16-B chunk is
// XOR-ed with (-(row >> 2)) & 3 on the source address and on the fragment read: the four 16-lane groups of a ds_read_b128 then
// touch 16 distinct 16-B slots of the 256-B bank row (rows r, r + 4, r + 8, r + 12 share the slots 4 (r & 3) .. + 3; a group
// holds one of them with each of two chunk indices, and the table 0, 3, 2, 1 separates all four).  `piece` is wave-uniform.
template <int BM>
__device__ __forceinline__ void ring_issue32_one(const bf16_t* __restrict__ A, int lda, int bm0, int mmax,
                                                 const bf16_t* __restrict__ W, int ldw, int bn0, int nmax, int k0,
                                                 unsigned lds_img, int piece, int lane) {
    const bool is_a = piece < BM / 16;
    const bf16_t* G = is_a ? A : W;
    const int ld = is_a ? lda : ldw, row0 = is_a ? bm0 : bn0, rmax = is_a ? mmax : nmax;
    const int row = (is_a ? piece : piece - BM / 16) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((-(row >> 2)) & 3);
    const bf16_t* src = G + (size_t)min(row0 + row, rmax) * ld + k0 + c * 8;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_img + piece * 1024)
        : "memory");
}

template <int N>
__device__ __forceinline__ void ring_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int STAGES, int EPI, int WM = 2, int WN = 2, bool SPREAD = false, bool WIDE_EPI = false, int BK = GEMM_BK, int KPB = 1>
__global__ void __launch_bounds__(64 * WM * WN) k_gemm_ring(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W,
                                                   int ldw, const bf16_t* __restrict__ bias,
                                                   const bf16_t* __restrict__ R, int ldr, bf16_t* __restrict__ C,
                                                   int ldc, const int* __restrict__ c_rows, int M, int N, int K,
                                                   int ksplit, float* __restrict__ slab,
                                                   unsigned* __restrict__ tickets) {
    // BK = 32 (round 4): K-steps of HALF the depth, so a stage of a 320 x 192 tile is 32 KB and four of them fit the LDS -- the
    // decode step's gate/up at 513 .. 640 chains as two row tiles x 115 column tiles in ONE round (ze_launch_gemm_wide).  Per
    // output element the same MFMAs over the same ascending 32-element K chunks: the same bits as BK = 64.
    static_assert(BK == 64 || BK == 32, "K-step");
    constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN), NT = 64 * WM * WN;
    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    constexpr int LPW = (BM + BN) * BK * 2 / 1024 / (WM * WN);  // DMA instructions per wave per stage
    // (BK = 32 only: REM waves carry one piece more -- 320 x 128 is 28 pieces, 384 x 192 is 36, on eight waves -- and count it
    //  in their own waits; wid is wave-uniform, so the branch costs a scalar compare)
    constexpr int REM = (BM + BN) * BK * 2 / 1024 % (WM * WN);
    // (BK = 64, round 5: the tall decode tiles -- 80 / 96 / 112 / 128 rows x 64 columns on BM / 16 x 2 waves, one 16 x 32 output tile
    //  each -- have (BM + 64) / 8 pieces on 10 .. 16 waves: the combined piece list [A rows | W rows] is dealt the same way)
    static_assert((REM == 0 || BK == 32 || !SPREAD) && BM % 16 == 0 && BN % 16 == 0, "a stage's pieces divide evenly among the waves");
    constexpr bool UNEVEN64 = BK == 64 && ((BM / 8) % (WM * WN) != 0 || (BN / 8) % (WM * WN) != 0);
    static_assert(!UNEVEN64 || !SPREAD, "the spread refill walks the A and W pieces separately");
    static_assert(STAGES >= 2 && STAGES <= 8 && (STAGES > 4 ? STAGES - 2 : 2) * (LPW + (REM ? 1 : 0)) < 64, "ring depth (vmcnt is a 6-bit counter)");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int nbx = (N + BN - 1) / BN, nby = (M + BM - 1) / BM;
    const int nwg = nbx * nby;
    const int ks = blockIdx.x / nwg;
    int bid = blockIdx.x % nwg;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // Workgroups with consecutive remapped ids share an XCD (its L2).  Walk the tile grid so that they share the
    // panel of the LARGER operand: with N > M (weights larger than activations: every GEMM of this path) ids run
    // down a column of tiles, an XCD then streams 1/8 of W once and re-reads the small A from L2 / MALL; the other
    // order makes all eight XCDs pull the whole of W through the fabric (gate/up at M = 802: 723 MB against 117 MB).
    const bool col_major = N > M;
    const int bm0 = (col_major ? bid % nby : bid / nbx) * BM, bn0 = (col_major ? bid / nby : bid % nbx) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wid / WN) * (BM / WM), wn0 = (wid % WN) * (BN / WN);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (the K slices of a split are cut in 64-element K-tiles whatever the K-step: the same slices, the same bits)
    const int nk_all = K / BK;
    const int nk_per = (K / GEMM_BK + ksplit - 1) / ksplit * (GEMM_BK / BK);
    const int kt0 = ks * nk_per;
    const int nk = max(0, min(nk_all - kt0, nk_per));

    // LDS byte address of the staging array (wave-uniform: it goes to M0)
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);
    auto issue = [&](int kt) {
        const unsigned img = smem_lds + (kt % STAGES) * STAGE_BYTES;
        const int k0 = (kt0 + kt) * BK;
        if constexpr (BK == 32) {
#pragma unroll
            for (int g = 0; g < LPW; ++g)
                ring_issue32_one<BM>(A, lda, bm0, M - 1, W, ldw, bn0, N - 1, k0, img, wid + g * (NT / 64), lane);
            if (REM > 0 && wid < REM)
                ring_issue32_one<BM>(A, lda, bm0, M - 1, W, ldw, bn0, N - 1, k0, img, wid + LPW * (NT / 64), lane);
        } else if constexpr (UNEVEN64) {
            // piece p = rows 8 p .. 8 p + 7 of the image [A rows | W rows] (BM is a multiple of 8: a piece never straddles)
            auto one = [&](int pc) {
                if (pc < BM / 8) ring_issue_one(A, lda, bm0, M - 1, k0, img, pc, lane);
                else ring_issue_one(W, ldw, bn0, N - 1, k0, img + BM * 128, pc - BM / 8, lane);
            };
#pragma unroll
            for (int g = 0; g < LPW; ++g) one(wid + g * (NT / 64));
            if (REM > 0 && wid < REM) one(wid + LPW * (NT / 64));
        } else {
            ring_issue<BM, NT>(A, lda, bm0, M - 1, k0, img, wid, lane);
            ring_issue<BN, NT>(W, ldw, bn0, N - 1, k0, img + BM * 128, wid, lane);
        }
    };
    const int fr = lane & 15, fq = lane >> 4;
    // KPB > 1 (round 5; BK = 64, no spread refill, nk a multiple of KPB -- the launcher checks): KPB K-steps per barrier.  The narrow
    // projections of a decode step (qkv, o: 16 x 32 outputs per wave) spend a K-step on its fixed chain -- counted wait, barrier,
    // DMA issue, fragment-read latency -- not on its four MFMAs: 32 K-steps took 14 us of a 16-us launch whatever the tile and
    // whatever the ring depth (4 / 6 / 8 stages: 16.2 / 18.9 / 18.2 us).  Here the ring moves in GROUPS of KPB stages: one wait, one
    // barrier and one batch of refill DMAs per group, then the group's 2 x KPB fragment reads and MFMA rows back to back.  Same
    // MFMAs in the same K order per output element: the same bits.
    // (Round 5, tried and dropped: a two-phase loop -- waves 0-3 read step k's fragments and request refills while their SIMD partners,
    // waves 4-7, run the MFMAs of step k - 1, then the roles swap; bit-identical; gate/up on 320 x 192 tiles 62.2 against 63.4 us, on
    // 384 x 192 tiles 71.3 against 77.9 -- the launch is bound by its LDS-DMA intake, see ze_launch_gemm_wide.  git history has it.)
    if constexpr (KPB > 1) {
        static_assert(BK == 64 && !SPREAD && STAGES % KPB == 0 && STAGES / KPB >= 2 && STAGES / KPB <= 4, "groups of K-steps");
        constexpr int G = STAGES / KPB;                          // groups the ring holds
        constexpr int GP = KPB * LPW, GPR = KPB * (LPW + 1);     // pieces of a group per wave (waves with the extra piece: GPR)
        static_assert((G - 2) * GPR < 64, "vmcnt is a 6-bit counter");
        const int ng = nk / KPB;
        auto issue_group = [&](int g) {
#pragma unroll
            for (int u = 0; u < KPB; ++u) issue(g * KPB + u);
        };
#pragma unroll
        for (int g = 0; g < G - 1; ++g)
            if (g < ng) issue_group(g);
        for (int g = 0; g < ng; ++g) {
            const int ahead = min(G - 2, ng - 1 - g);  // groups still allowed in flight behind group g
            if (REM > 0 && wid < REM) {
                if (ahead >= 2) ring_wait<(G > 3 ? 2 : 0) * GPR>();
                else if (ahead == 1) ring_wait<(G > 2 ? 1 : 0) * GPR>();
                else ring_wait<0>();
            } else {
                if (ahead >= 2) ring_wait<(G > 3 ? 2 : 0) * GP>();
                else if (ahead == 1) ring_wait<(G > 2 ? 1 : 0) * GP>();
                else ring_wait<0>();
            }
            __builtin_amdgcn_s_barrier();
            if (g + G - 1 < ng) issue_group(g + G - 1);
#pragma unroll
            for (int u = 0; u < KPB; ++u) {
                const uint8_t* imgA = smem + ((g * KPB + u) % STAGES) * STAGE_BYTES;
                const uint8_t* imgB = imgA + BM * 128;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int chunk = kk * 4 + fq;
                    bf16x8 fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = wm0 + i * 16 + fr;
                        fa[i] = *reinterpret_cast<const bf16x8*>(imgA + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int row = wn0 + j * 16 + fr;
                        fb[j] = *reinterpret_cast<const bf16x8*>(imgB + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        gemm_finish<BM, BN, EPI, WM, WN, WIDE_EPI>(acc, smem, bias, R, ldr, C, ldc, c_rows, M, N, ksplit, ks, bid, nwg, bm0, bn0, slab, tickets);
        return;
    }
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue(s);

    for (int kt = 0; kt < nk; ++kt) {
        // stages still allowed in flight behind stage kt: min(STAGES - 2, nk - 1 - kt)
        const int ahead = min(STAGES - 2, nk - 1 - kt);
        if constexpr (STAGES > 4) {
            if (REM > 0 && wid < REM) {  // (the waves that carry one piece more per stage count it in their own waits)
                switch (ahead) {
                    case 6: ring_wait<6 * (LPW + 1)>(); break;
                    case 5: ring_wait<5 * (LPW + 1)>(); break;
                    case 4: ring_wait<4 * (LPW + 1)>(); break;
                    case 3: ring_wait<3 * (LPW + 1)>(); break;
                    case 2: ring_wait<2 * (LPW + 1)>(); break;
                    case 1: ring_wait<LPW + 1>(); break;
                    default: ring_wait<0>(); break;
                }
            } else {
                switch (ahead) {
                    case 6: ring_wait<6 * LPW>(); break;
                    case 5: ring_wait<5 * LPW>(); break;
                    case 4: ring_wait<4 * LPW>(); break;
                    case 3: ring_wait<3 * LPW>(); break;
                    case 2: ring_wait<2 * LPW>(); break;
                    case 1: ring_wait<LPW>(); break;
                    default: ring_wait<0>(); break;
                }
            }
        } else if (REM > 0 && wid < REM) {
            if (ahead >= 2) ring_wait<2 * (LPW + 1)>();
            else if (ahead == 1) ring_wait<LPW + 1>();
            else ring_wait<0>();
        } else {
            if (ahead >= 2) ring_wait<2 * LPW>();
            else if (ahead == 1) ring_wait<LPW>();
            else ring_wait<0>();
        }
        __builtin_amdgcn_s_barrier();
        // SPREAD: the refill DMAs of this K-step go out between the rows of MFMAs below instead of in one burst
        // behind the barrier (every wave of the workgroup issuing its pieces at once leaves the matrix pipe idle
        // for the length of the burst)
        const bool more = kt + STAGES - 1 < nk;
        if (!SPREAD && more) issue(kt + STAGES - 1);
        if constexpr (BK == 32) {
            const unsigned img_next = smem_lds + ((kt + STAGES - 1) % STAGES) * STAGE_BYTES;
            const int k_next = (kt0 + kt + STAGES - 1) * BK;
            const uint8_t* imgA = smem + (kt % STAGES) * STAGE_BYTES;
            const uint8_t* imgB = imgA + BM * 64;
            const int sw = (-(fr >> 2)) & 3;  // (tile and wave origins are multiples of 16 rows: the swizzle term is the lane's)
            bf16x8 fa[TM], fb[TN];
            // (ablation builds of tools/probes/ring_ablate.sh, never shipped: ZE_RING_ABLATE = 1 no MFMAs, 2 no fragment reads either)
#if !defined(ZE_RING_ABLATE) || ZE_RING_ABLATE < 2
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(imgA + (wm0 + i * 16 + fr) * 64 + ((fq ^ sw) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(imgB + (wn0 + j * 16 + fr) * 64 + ((fq ^ sw) << 4));
#else
            (void)imgA; (void)imgB; (void)sw;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = bf16x8{};
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = bf16x8{};
#endif
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#if !defined(ZE_RING_ABLATE)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
#elif ZE_RING_ABLATE == 1
#pragma unroll
                for (int j = 0; j < TN; ++j) {  // keep the fragment reads alive without the matrix pipe
                    acc[i][j][0] += __uint_as_float((unsigned)fa[i][0] ^ (unsigned)fb[j][0]);
                }
#endif
                if (SPREAD && more) {
#pragma unroll
                    for (int pc = 0; pc < LPW; ++pc)
                        if (pc * TM / LPW == i)
                            ring_issue32_one<BM>(A, lda, bm0, M - 1, W, ldw, bn0, N - 1, k_next, img_next, wid + pc * (NT / 64), lane);
                    if (REM > 0 && i == TM - 1 && wid < REM)
                        ring_issue32_one<BM>(A, lda, bm0, M - 1, W, ldw, bn0, N - 1, k_next, img_next, wid + LPW * (NT / 64), lane);
                }
            }
            continue;
        }
        constexpr int LPA = BM / 8 / (WM * WN);  // pieces of the A image per wave
        const unsigned img_next = smem_lds + ((kt + STAGES - 1) % STAGES) * STAGE_BYTES;
        const int k_next = (kt0 + kt + STAGES - 1) * GEMM_BK;
        const uint8_t* imgA = smem + (kt % STAGES) * STAGE_BYTES;
        const uint8_t* imgB = imgA + BM * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = kk * 4 + fq;
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm0 + i * 16 + fr;
                fa[i] = *reinterpret_cast<const bf16x8*>(imgA + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + j * 16 + fr;
                fb[j] = *reinterpret_cast<const bf16x8*>(imgB + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                if (SPREAD && more) {
                    const int r = kk * TM + i;  // row of MFMAs just issued, of 2 * TM per K-step
#pragma unroll
                    for (int pc = 0; pc < LPW; ++pc)
                        // two stages: the refill has to land by the next barrier, so its pieces go out behind the
                        // FIRST rows of the step (one per row) rather than evenly over all of them (4096^3: 1042 against
                        // 993 TFLOP/s, the batched-prefill shapes within +-3 %)
                        if ((STAGES == 2 ? min(pc, 2 * TM - 1) : pc * (2 * TM) / LPW) == r) {
                            if (pc < LPA)
                                ring_issue_one(A, lda, bm0, M - 1, k_next, img_next, wid + pc * (NT / 64), lane);
                            else
                                ring_issue_one(W, ldw, bn0, N - 1, k_next, img_next + BM * 128,
                                               wid + (pc - LPA) * (NT / 64), lane);
                        }
                }
            }
        }
    }
    __syncthreads();  // the tail reuses the staging LDS
    gemm_finish<BM, BN, EPI, WM, WN, WIDE_EPI>(acc, smem, bias, R, ldr, C, ldc, c_rows, M, N, ksplit, ks, bid, nwg, bm0, bn0, slab, tickets);
}


// ----------------------------------------------------------------------------------------------------------------
// 256 x 256 tiles, eight phases per two K-tiles (the many-round grids of the batched prefill and of the ViT).  Same tile,
// same eight waves (2 x 4, 128 x 64 outputs each) and the same two 64-KB K-tile buffers as k_gemm_ring<256, 256, 2>, but the
// K-tile is staged and consumed in four HALF-TILES of 16 KB -- A rows {wr*128 + mh*64 + 0..63} for both wave rows (mh = 0, 1),
// W rows {wc*64 + nh*32 + 0..31} for the four wave columns (nh = 0, 1) -- and a K-tile's 64 MFMAs per wave run as four
// quadrants (mh, nh) of 16, one per phase:
//   phase 1: read A-half 0 + W-half 0 (12 ds_read_b128), stage A-half 1 of tile t+1;   MFMA quadrant (0, 0)
//   phase 2: read W-half 1 (4),                          stage A-half 0 of tile t+2;   MFMA quadrant (0, 1)
//   phase 3: read A-half 1 (8),                          stage W-half 0 of tile t+2;   MFMA quadrant (1, 1)
//   phase 4: no read,                                    stage W-half 1 of tile t+2;   MFMA quadrant (1, 0)
// each phase = reads + 2 LDS-DMA instructions per wave, s_barrier, the 16 MFMAs at raised priority, s_barrier.  A half-tile
// is restaged as soon as its last fragment read is over (two phases later; A-half 0 one phase later, behind an lgkmcnt that
// retires its reads before the barrier), so three half-tiles (48 KB per workgroup) are in flight at the ONE counted wait of a
// K-tile -- s_waitcnt vmcnt(6) in phase 4, never 0 inside the loop -- against one K-tile issued behind the barrier and
// drained at the next in the two-stage ring.  The two wave rows run one barrier apart (wave w and w + 4 share a SIMD): while
// one row's waves issue MFMAs the other's read fragments and issue DMAs (cdna_hip_programming.md, the 256^2 8-phase
// template).  A staged half-tile is read one phase after the wait + barrier that retire it.  Per output element the MFMAs and
// their K order are k_gemm_ring's: identical bits.
// Wide-store epilogue of k_gemm_p8 (round 4).  The MFMA result layout gives a lane four ROWS of one column, so the plain
// epilogue (gemm_finish) leaves a 256 x 256 tile as 128 two-byte stores per lane, 32 B of a row per wave-instruction -- for the
// ViT's K = 1280 that is a fifth of a tile's time.  The staging LDS holds the NEXT tile's first half-tiles by now, so every wave
// gets 4 KB of its own behind the two K-tile buffers (128 KB + 8 x 4 KB <= 160 KB): its 128 x 64 outputs cross that scratch in
// four chunks of 32 rows -- written as the accumulators hold them, read back as 16-byte row pieces (XOR-swizzled by the row:
// conflict-free both ways) -- and leave as 64 lanes x 16 B = eight whole 128-byte rows per wave-instruction.  No workgroup
// barrier: a wave's LDS operations execute in order.  Values, roundings and their order are the plain epilogue's, element for
// element (bias, bf16 rounding, GELU, residual add in fp32 on the rounded value, SwiGLU on rounded gate / up): same bits.
template <int EPI>
__device__ __forceinline__ void p8_finish_wide(f32x4 (&acc)[8][4], uint8_t* scratch, const bf16_t* __restrict__ bias,
                                               const bf16_t* __restrict__ R, int ldr, bf16_t* __restrict__ C, int ldc,
                                               const int* __restrict__ c_rows, int M, int N, int row_base, int col_base, int lane) {
    constexpr bool SW = EPI == ZE_EPI_SWIGLU, F32 = EPI == ZE_EPI_F32;
    constexpr int EB = F32 ? 4 : 2;           // bytes per output element (F32: the fp32 copy of the bf16-rounded logit)
    constexpr int OW = SW ? 32 : 64;          // output columns of the wave's tile
    constexpr int ROWB = OW * EB;             // bytes per staged row: 64 (SwiGLU), 128, or 256 (fp32)
    constexpr int PPR = ROWB / 16;            // 16-byte pieces per output row
    constexpr int CR = 4096 / ROWB > 32 ? 32 : 4096 / ROWB;  // rows per chunk: 32 (16 for fp32): one or two 16-row MFMA tiles
    constexpr int TI = CR / 16;               // row tiles per chunk
    const int fr = lane & 15, fq = lane >> 4;
    float b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ncol = col_base + j * 16 + fr;
        b[j] = (bias && ncol < N) ? bf16_to_f32(bias[ncol]) : 0.f;
    }
    const int ocol_base = SW ? col_base / 2 : col_base, on = SW ? N / 2 : N;
#pragma unroll
    for (int c = 0; c < 128 / CR; ++c) {
        // ---- CR rows x OW columns into the scratch: element (lrow, col) at lrow * ROWB + ((col * EB) ^ swizzle(lrow) << 4)
#pragma unroll
        for (int ii = 0; ii < TI; ++ii) {
            const int i = TI * c + ii;
#pragma unroll
            for (int j = 0; j < 4; j += (SW ? 2 : 1)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lrow = ii * 16 + fq * 4 + r;
                    const int swz = (lrow & (PPR - 1)) << 4;
                    float v;
                    int col;
                    if (SW) {
                        const float g = bf16_round(acc[i][j][r] + b[j]);
                        const float u = bf16_round(acc[i][j + 1][r] + b[j + 1]);
                        v = bf16_round(silu_f(g)) * u;
                        col = (j / 2) * 16 + fr;
                    } else {
                        v = bf16_round(acc[i][j][r] + b[j]);
                        if (EPI == ZE_EPI_GELU) v = gelu_erf(v);
                        col = j * 16 + fr;
                    }
                    if (F32) *reinterpret_cast<float*>(scratch + lrow * ROWB + ((col * 4) ^ swz)) = v;
                    else *reinterpret_cast<bf16_t*>(scratch + lrow * ROWB + ((col * 2) ^ swz)) = f32_to_bf16(v);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- back as 16-byte row pieces: 64 lanes = (64 / PPR) rows per instruction
        constexpr int RPI = 64 / PPR;  // rows per wave-instruction: 16 (SwiGLU), 8, or 4 (fp32)
#pragma unroll
        for (int it = 0; it < CR / RPI; ++it) {
            const int lrow = it * RPI + lane / PPR, piece = lane % PPR;
            uint4 v = *reinterpret_cast<const uint4*>(scratch + lrow * ROWB + ((piece ^ (lrow & (PPR - 1))) << 4));
            const int row = row_base + c * CR + lrow;
            const int col = ocol_base + piece * (16 / EB);
            if (row < M && col < on) {
                const int orow = c_rows ? c_rows[row] : row;
                if (EPI == ZE_EPI_RESIDUAL) {  // out = bf16(residual + value), per element as the plain epilogue
                    const uint4 rr = *reinterpret_cast<const uint4*>(R + (size_t)row * ldr + col);
                    const uint32_t* pv = reinterpret_cast<const uint32_t*>(&v);
                    const uint32_t* pr = reinterpret_cast<const uint32_t*>(&rr);
                    uint32_t o[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = __uint_as_float(pr[q] << 16) + __uint_as_float(pv[q] << 16);
                        const float hi = __uint_as_float(pr[q] & 0xffff0000u) + __uint_as_float(pv[q] & 0xffff0000u);
                        o[q] = (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
                    }
                    v = make_uint4(o[0], o[1], o[2], o[3]);
                }
                if (F32) *reinterpret_cast<uint4*>(reinterpret_cast<float*>(C) + (size_t)orow * ldc + col) = v;
                else *reinterpret_cast<uint4*>(C + (size_t)orow * ldc + col) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are back before the next chunk overwrites the scratch
    }
}

// Split-K tail of k_gemm_p8 (round 6): gemm_finish's protocol on the eight-phase kernel's 512 threads x 32 MFMA tiles -- every slice
// parks its fp32 accumulators write-through ([MFMA tile][thread] slabs), drains, takes the tile's ticket; the last arriver adds the
// slices IN SLICE ORDER from zero (whatever order they arrived in) and returns true: it runs the epilogue.  `flag` is LDS the next
// tile's prefetch does not touch (behind the two K-tile buffers).
__device__ __forceinline__ bool p8_splitk_reduce(f32x4 (&acc)[8][4], unsigned* flag, int ksplit, int ks, int bid, int nwg,
                                                 float* __restrict__ slab, unsigned* __restrict__ tickets) {
    constexpr int NT = 512, T = 32;
    const int tid = threadIdx.x;
    const size_t slab_bytes = (size_t)ksplit * nwg * NT * T * 16;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)min(slab_bytes, (size_t)0x7fffffff), 0x00020000);
    {
        const unsigned base = (unsigned)(((size_t)(ks * nwg + bid) * T * NT + tid) * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 v;
                v.x = __float_as_uint(acc[i][j][0]);
                v.y = __float_as_uint(acc[i][j][1]);
                v.z = __float_as_uint(acc[i][j][2]);
                v.w = __float_as_uint(acc[i][j][3]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, base + (unsigned)(i * 4 + j) * NT * 16, 0, 16 /* sc1 */);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the ticket
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(&tickets[bid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = (old == (unsigned)ksplit - 1u) ? 1u : 0u;
        if (last) __hip_atomic_store(&tickets[bid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        *flag = last;
    }
    __syncthreads();
    const unsigned last = *flag;
    __syncthreads();  // (every wave has read the flag before the epilogue's scratch, which shares its LDS, is written)
    if (last == 0u) return false;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < ksplit; ++q) {
        const unsigned base = (unsigned)(((size_t)(q * nwg + bid) * T * NT + tid) * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + (unsigned)(i * 4 + j) * NT * 16, 0, 16);
                acc[i][j][0] += __uint_as_float(v.x);
                acc[i][j][1] += __uint_as_float(v.y);
                acc[i][j][2] += __uint_as_float(v.z);
                acc[i][j][3] += __uint_as_float(v.w);
            }
    }
    return true;
}

template <int EPI, bool WIDE = false>
__global__ void __launch_bounds__(512) k_gemm_p8(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
                                                 const bf16_t* __restrict__ bias, const bf16_t* __restrict__ R, int ldr,
                                                 bf16_t* __restrict__ C, int ldc, const int* __restrict__ c_rows, int M, int N,
                                                 int K, int blk, int ksplit, float* __restrict__ slab, unsigned* __restrict__ tickets) {
    constexpr int BM = 256, BN = 256, HALF = 128 * 128, BUF = 4 * HALF;  // buffer = [A-half 0][A-half 1][W-half 0][W-half 1]
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int nbx = (N + BN - 1) / BN, nby = (M + BM - 1) / BM;
    const int nwg = nbx * nby;
    const bool col_major = N > M;  // (see k_gemm_ring)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int fr = lane & 15, fq = lane >> 4;
    // SPLIT K (round 6: the long-K projection of a prefill pass, ze_prefill_ksplit): a unit of work = (tile, K slice); the slices of a
    // tile are cut in 64-element K-tiles exactly as k_gemm_ring / k_gemm_tn cut theirs (ceil(nk / ksplit) per slice) and meet in the
    // fp32 slabs in slice order (p8_splitk_reduce) -- the same sums in the same order whichever kernel serves a row count.
    const int nk_all = K / GEMM_BK;
    const int nk_per = (nk_all + ksplit - 1) / ksplit;
    int ks = 0, kt0 = 0, nk = nk_all;
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);

    // PERSISTENT: workgroup g takes tiles g, g + gridDim.x, ... of the launch order (the XCD-aware remap of k_gemm_ring applied to
    // that order: gridDim.x is a multiple of 8, so a workgroup's tiles keep its XCD label), and the first seven half-tiles of
    // its NEXT tile are requested before the epilogue of the current one -- the ~2 us a tile's prologue waits for its first
    // K-tile pass behind the stores (K = 2048: 32 K-tiles of ~1.9 us per tile; K = 1280, the ViT: 20).
    // BLOCKED WALK (round 5).  The remap gives an XCD a contiguous range of the launch order, and that order ran DOWN A COLUMN of
    // tiles: the 32 workgroups an XCD runs at a time then share ONE W tile and touch 32 different A tiles -- and for every one of its
    // columns the XCD pulls ALL of A through the fabric again (its L2 holds 4 MB; the 16-chain prefill's gate/up: 10.75 columns x 52 MB
    // per XCD = 4.5 GB per launch from the Infinity Cache, which serves ~7 TB/s against ~20 for L2 hits: tools/probes/ingest_probe.hip;
    // the down projection: every XCD reads the whole 288-MB A -- from HBM, it does not fit the Infinity Cache).  With blk = R << 8 | C
    // the order runs through blocks of R row tiles x C column tiles (panels of C columns, row blocks of R inside a panel, down a column
    // inside a block): 32 concurrent tiles = an 8 x 4 block share 8 A tiles and 4 W tiles, and an A tile crosses the fabric once per
    // FOUR columns.  Which workgroup computes which tile never changes a tile's arithmetic: same bits.  blk = 0: the column walk.
    int bm0 = 0, bn0 = 0, bid = 0;
    auto place = [&](int unit) {
        // (the slices of a tile are consecutive units: neighbouring workgroups run them side by side, the last arriver waits for no one)
        const int tile = ksplit > 1 ? unit / ksplit : unit;
        if (ksplit > 1) {
            ks = unit - tile * ksplit;
            kt0 = ks * nk_per;
            nk = max(0, min(nk_all - kt0, nk_per));
        }
        const int q = nwg / 8, r = nwg % 8, xcd = tile % 8, idx = tile / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        if (blk > 0) {
            // (major = the dimension the old walk ran along: rows when col_major)
            const int nmaj = col_major ? nby : nbx, nmin = col_major ? nbx : nby;
            const int RB = blk >> 8, CB = blk & 255;                      // block = RB tiles along major x CB along minor
            const int per_full = nmaj * CB, full = nmin / CB;
            int panel, o, cw;
            if (bid < full * per_full) { panel = bid / per_full; o = bid % per_full; cw = CB; }
            else { panel = full; o = bid - full * per_full; cw = nmin - full * CB; }
            const int per_rb = RB * cw, fb = nmaj / RB;
            int rb, oo, rw;
            if (o < fb * per_rb) { rb = o / per_rb; oo = o % per_rb; rw = RB; }
            else { rb = fb; oo = o - fb * per_rb; rw = nmaj - fb * RB; }
            const int tmaj = rb * RB + oo % rw, tmin = panel * CB + oo / rw;
            bm0 = (col_major ? tmaj : tmin) * BM;
            bn0 = (col_major ? tmin : tmaj) * BN;
            return;
        }
        bm0 = (col_major ? bid % nby : bid / nbx) * BM;
        bn0 = (col_major ? bid / nby : bid % nbx) * BN;
    };
    // DMA sources: this wave moves pieces 2 wid and 2 wid + 1 (8 half-tile rows x 128 B each) of every half-tile; byte offsets
    // from the operand's base (32 bits: the operands of this path are far below 4 GB), the K offset rides in the scalar base
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
    // slot 0..3 = A-half 0, A-half 1, W-half 0, W-half 1
    auto stage = [&](int slot, int t) {
        const unsigned dst = smem_lds + (t & 1) * BUF + slot * HALF + wid * 2048;
        const bf16_t* base = (slot < 2 ? A : W) + (size_t)(kt0 + t) * GEMM_BK;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned off = slot == 0 ? offA[0][q] : slot == 1 ? offA[1][q] : slot == 2 ? offB[0][q] : offB[1][q];
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(off), "s"(dst + q * 1024), "s"(base)
                         : "memory");
        }
    };
    // a tile's prologue, in the loop's order of issue: tile 0 whole, then the three half-tiles of tile 1 that phases 2-4 stage
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
    // fragment addresses: half-tile row = (wr * 64 | wc * 32) + 16 i + fr; the swizzle term (row >> 1) & 7 = fr >> 1 for all i
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
    auto quadrant = [&](int mh, int nh, const bf16x8 (&f)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mh * 4 + i][nh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][i], f[kk][j], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    const int nunits = nwg * ksplit;
    int tile = blockIdx.x;   // (the UNIT this workgroup is on: a tile, or a K slice of one)
    if (tile >= nunits) return;
    place(tile);
    sources();
    prologue();
    for (;;) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (a first tile: the seven half-tiles just issued; a later one: they were issued before the previous tile's epilogue,
        //  whose stores share the counter -- everything has to be back)
        if (tile == (int)blockIdx.x && nk > 1) ring_wait<6>();
        else ring_wait<0>();
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave row runs one barrier behind the first

        for (int t = 0; t < nk; ++t) {
            const int b = t & 1;
            // ---- phase 1
            read_a(b, 0);
            __builtin_amdgcn_sched_barrier(0);
            read_b(fb0, b, 0);
            if (t + 1 < nk) stage(1, t + 1);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");  // the A-half 0 reads are over: phase 2 restages it
            __builtin_amdgcn_s_barrier();
            quadrant(0, 0, fb0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 2
            read_b(fb1, b, 1);
            if (t + 2 < nk) stage(0, t + 2);
            __builtin_amdgcn_s_barrier();
            quadrant(0, 1, fb1);
            __builtin_amdgcn_s_barrier();
            // ---- phase 3
            read_a(b, 1);
            if (t + 2 < nk) stage(2, t + 2);
            __builtin_amdgcn_s_barrier();
            quadrant(1, 1, fb1);
            __builtin_amdgcn_s_barrier();
            // ---- phase 4: everything older than the three half-tiles just issued has landed (= all of tile t + 1)
            if (t + 2 < nk) {
                stage(3, t + 2);
                ring_wait<6>();
            } else {
                ring_wait<0>();
            }
            __builtin_amdgcn_s_barrier();
            quadrant(1, 0, fb0);
            __builtin_amdgcn_s_barrier();
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
        // every wave is past its last fragment read (wave row 0: of phase 3, two barriers back; row 1: likewise behind the
        // barrier it just shared): the LDS is free for the next tile's first half-tiles
        const int done_bid = bid, done_bm0 = bm0, done_bn0 = bn0, done_ks = ks;
        tile += gridDim.x;
        const bool more = tile < nunits;
        if (more) {
            place(tile);
            sources();
            prologue();
        }
        bool finish = true;
        if (ksplit > 1)   // (workgroup-uniform) park this slice; the tile's last arriver adds the slices in slice order and finishes
            finish = p8_splitk_reduce(acc, reinterpret_cast<unsigned*>(smem + 2 * BUF), ksplit, done_ks, done_bid, nwg, slab, tickets);
        if (!finish) {
            if (!more) break;
            continue;
        }
        if constexpr (WIDE)
            p8_finish_wide<EPI>(acc, smem + 2 * BUF + wid * 4096, bias, R, ldr, C, ldc, c_rows, M, N, done_bm0 + wr * 128, done_bn0 + wc * 64, lane);
        else
            gemm_finish<BM, BN, EPI, 2, 4>(acc, smem, bias, R, ldr, C, ldc, c_rows, M, N, 1, 0, done_bid, nwg, done_bm0, done_bn0, nullptr, nullptr);
        if (!more) break;
    }
}

static void launch_p8(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R, int ldr,
                      bf16_t* C, int ldc, const int* c_rows, int M, int N, int K, hipStream_t s, int ksplit = 1,
                      const ze_gemm_ws& ws = ze_gemm_ws()) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cus = std::max(8, cus / 8 * 8);  // (a multiple of 8: a persistent workgroup's tiles keep its XCD label)
    }
    const int tiles = ze_cdiv(M, 256) * ze_cdiv(N, 256);
    // Persistent workgroups (a tile's first half-tiles requested behind the previous tile's epilogue) -- unless several engines
    // share the GPU (lanes: knob 4 = 1, set by clone_lane / bench.py): a persistent launch holds every CU's LDS until its LAST tile
    // is done, and the other lane's decode kernels wait for all of it (the o projection's 200 workgroups averaged 127 us in the
    // two-lane stream against 16 alone); one tile per workgroup lets them in at tile boundaries: stream 80.5 -> 81.6 questions/s
    // same box, twice.  (knob 7 = 9: the same for A/B runs)
    // Round 6: the library knows how many engines are alive (ze_live_engines), so the form follows that by itself -- knob 4 = 0: tile-
    // granular iff more than one engine exists; 1: always tile-granular; 2: always persistent (measurements).  No caller flips a
    // process-wide switch on behalf of engines it does not own any more.
    const bool shared_gpu = ze_gemv_knobs[4] == 1 || (ze_gemv_knobs[4] == 0 && ze_live_engines > 1);
    const int units = tiles * std::max(1, ksplit);   // (split K: a unit = a K slice of a tile)
    const int grid = (ze_gemv_knobs[7] == 9 || shared_gpu) ? units : std::min(units, cus);
    // the wide-store epilogue (p8_finish_wide: 16-byte row pieces through 4 KB of LDS per wave) wherever rows are 16-byte
    // aligned; knob 7 = 10: the plain two-byte epilogue, for A/B runs and the bit-equality test
    const bool sw = epi == ZE_EPI_SWIGLU;
    const bool wide = ze_gemv_knobs[7] != 10 && (ldc % 8) == 0 && ((size_t)C % 16) == 0 && ((sw ? N / 2 : N) % 8) == 0 &&
                      (epi != ZE_EPI_RESIDUAL || ((ldr % 8) == 0 && ((size_t)R % 16) == 0));
    const size_t lds = wide ? (128 + 32) * 1024 : (128 + 1) * 1024;   // (+ 1 KB: the split-K flag lives behind the K-tile buffers)
    // the blocked walk of the tile grid (k_gemm_p8: `place`): 8 x 4 blocks; knob 21 = 1: the column walk of rounds 3-4, any other
    // value v > 1: blocks of (v >> 8) x (v & 255)
    const int blk = ze_gemv_knobs[21] == 1 ? 0 : (ze_gemv_knobs[21] > 1 ? ze_gemv_knobs[21] : ((8 << 8) | 4));
#define ZE_P8_LAUNCH(E, WD)                                                                                                    \
    do {                                                                                                                       \
        static bool attr_set = false;                                                                                          \
        if (!attr_set) {                                                                                                       \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_p8<E, WD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                                   \
        }                                                                                                                      \
        hipLaunchKernelGGL((k_gemm_p8<E, WD>), dim3(grid), dim3(512), lds, s, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, blk, \
                           std::max(1, ksplit), ws.slab, ws.tickets);                                                          \
    } while (0)
    if (wide) {
        switch (epi) {
            case ZE_EPI_NONE: ZE_P8_LAUNCH(ZE_EPI_NONE, true); break;
            case ZE_EPI_GELU: ZE_P8_LAUNCH(ZE_EPI_GELU, true); break;
            case ZE_EPI_RESIDUAL: ZE_P8_LAUNCH(ZE_EPI_RESIDUAL, true); break;
            case ZE_EPI_SWIGLU: ZE_P8_LAUNCH(ZE_EPI_SWIGLU, true); break;
            case ZE_EPI_F32: ZE_P8_LAUNCH(ZE_EPI_F32, true); break;
        }
        return;
    }
    switch (epi) {
        case ZE_EPI_NONE: ZE_P8_LAUNCH(ZE_EPI_NONE, false); break;
        case ZE_EPI_GELU: ZE_P8_LAUNCH(ZE_EPI_GELU, false); break;
        case ZE_EPI_RESIDUAL: ZE_P8_LAUNCH(ZE_EPI_RESIDUAL, false); break;
        case ZE_EPI_SWIGLU: ZE_P8_LAUNCH(ZE_EPI_SWIGLU, false); break;
        case ZE_EPI_F32: ZE_P8_LAUNCH(ZE_EPI_F32, false); break;
    }
#undef ZE_P8_LAUNCH
}

// ----------------------------------------------------------------------------------------------------------------
// Weight-streaming GEMM of the batched decode step in the row-streaming regime (65..256 chains; ze_launch_gemm_wide).
// Rows = chains: a workgroup takes ALL rows (BM = 128 or 256, rows past M re-read the last one) and BN weight columns, so
// the weights cross the chip once; the activations (M x K, L2-resident) are what every workgroup re-reads.  What bounds the
// ring kernel above on this shape is BYTES IN FLIGHT: its stages hold activations and weights alike, three of them fill the
// LDS, and two stages of weights in flight per CU against the ~2.5 us of an HBM miss are 25 GB/s per CU (gate/up at 256
// chains: 43.6 us for 90 MB).  Here the two operands are pipelined separately, each by its own waves and to its own depth:
//   * waves 4..7 stage the ACTIVATIONS with the LDS-DMA ring of k_gemm_ring (SA stages of BM x 64; L2 latency: two stages
//     in flight suffice), counted vmcnt on their own queue;
//   * waves 0..3 stream the WEIGHTS into REGISTERS, DW K-steps ahead (non-temporal 16-B loads of full 128-B row pieces,
//     DW x BN x 128 B in flight per workgroup: 96 KB for gate/up), and write the step after next into one of two small LDS
//     stages with the ring's swizzle (ds_write_b128) -- the compiler's counted vmcnt covers exactly these loads, since
//     these waves issue no DMA;
//   * all eight waves run the MFMAs (8 x 1 layout: 16 or 32 rows x BN columns per wave), ONE raw barrier per K-step.
// Accumulation order per output element = k_gemm_ring's (same MFMA, K in sequence; split-K slices through gemm_finish), so
// the results are bit-identical to the ring / register-staged kernels whatever tile serves a row count.
template <int BM, int BN, int SA, int DW, int NK, int EPI>
__global__ void __launch_bounds__(512) k_gemm_wstream(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W,
                                                      int ldw, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ R,
                                                      int ldr, bf16_t* __restrict__ C, int ldc, int M, int N, int K, int ksplit,
                                                      float* __restrict__ slab, unsigned* __restrict__ tickets) {
    constexpr int WM = 8, WN = 1;
    constexpr int TM = BM / (16 * WM), TN = BN / 16;
    constexpr int A_STAGE = BM * 128, W_STAGE = BN * 128;
    constexpr int LA = BM / 8 / 4;   // activation pieces (8 rows x 128 B) per loader wave per K-step
    constexpr int PW = BN / 8 / 4;   // weight pieces per loader wave per K-step
    static_assert(BM % 128 == 0 && BN % 32 == 0 && SA >= 2 && SA <= 4 && DW >= 2 && DW <= 8 && LA * (SA - 1) < 64 && PW * DW < 64, "tile");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const smemW = smem + SA * A_STAGE;

    const int nwg = (N + BN - 1) / BN;  // one row tile: all M rows
    // K slices of a tile on consecutive block ids: they run on different XCDs, the workgroups of ONE slice (ids equal
    // mod ksplit = 8) share an XCD and with it the slice of the activations they all read
    const int ks = blockIdx.x % ksplit, bid = blockIdx.x / ksplit;
    const int bn0 = bid * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = wid * (BM / WM);
    const int fr = lane & 15, fq = lane >> 4;
    const bool wload = wid < 4;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_all = K / GEMM_BK;
    const int nk_per = (nk_all + ksplit - 1) / ksplit;
    const int kt0 = ks * nk_per;
    const int nk = max(0, min(nk_all - kt0, nk_per));

    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);
    // activation loader: piece g of the stage image = rows 8g..8g+7
    auto issue_a = [&](int kt) {
        const unsigned img = smem_lds + (kt % SA) * A_STAGE;
        const int k0 = (kt0 + kt) * GEMM_BK;
#pragma unroll
        for (int g = 0; g < LA; ++g) ring_issue_one(A, lda, 0, M - 1, k0, img, (wid - 4) + g * 4, lane);
    };
    // weight loader: this lane's 16 bytes of piece p (row 8p + lane / 8, chunk lane % 8) of K-step kt: buffer loads with a
    // loop-invariant per-lane offset and the K offset in a scalar register, issued and awaited from inline asm with a
    // COUNTED s_waitcnt (the load and the LDS write that waits for it are one asm statement each: hipcc's own counting
    // degrades to a two-step prefetch around any branch or loop -- it put vmcnt(0) / vmcnt(5) where vmcnt(21) is exact).
    // A weight-loader wave issues nothing else while the loop runs, so its queue holds PW loads per step in flight.
    const int wrow = lane >> 3, wchunk = lane & 7;
    u32x4 wrsrc;
    {
        const unsigned long long wp = (unsigned long long)(size_t)W;
        wrsrc.x = __builtin_amdgcn_readfirstlane((unsigned)wp);
        wrsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(wp >> 32) & 0xffffu);  // stride 0
        wrsrc.z = 0x7fffffffu;
        wrsrc.w = 0x00020000u;
    }
    unsigned woff[PW], wlds[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int row = (wid + p * 4) * 8 + wrow;
        woff[p] = (unsigned)(((size_t)min(bn0 + row, N - 1) * ldw + wchunk * 8) * sizeof(bf16_t));
        wlds[p] = smem_lds + SA * A_STAGE + row * 128 + ((wchunk ^ ((row >> 1) & 7)) << 4);
    }
    u32x4 wreg[DW][PW];
    auto load_w = [&](u32x4 (&dst)[PW], int kt) {
        const int koff = __builtin_amdgcn_readfirstlane((kt0 + min(kt, nk - 1)) * GEMM_BK * (int)sizeof(bf16_t));
#pragma unroll
        for (int p = 0; p < PW; ++p)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(dst[p]) : "v"(woff[p]), "s"(wrsrc), "s"(koff) : "memory");
    };
    // waits until at most `left` of this wave's loads are outstanding (the step's PW loads are the oldest in the queue),
    // then writes the step into weight stage (kt & 1)
#define ZE_WS_STORE(SRC, KT, LEFT)                                                                                          \
    do {                                                                                                                    \
        _Pragma("unroll") for (int p = 0; p < PW; ++p)                                                                      \
            asm volatile("s_waitcnt vmcnt(%2)\n\tds_write_b128 %0, %1"                                                      \
                         :: "v"(wlds[p] + ((KT) & 1) * W_STAGE), "v"((SRC)[p]), "n"(LEFT) : "memory");                       \
    } while (0)

    // MFMAs of K-step kt on activation stage kt % SA and weight stage kt & 1
    auto compute = [&](int kt) {
        const uint8_t* imgA = smem + (kt % SA) * A_STAGE;
        const uint8_t* imgB = smemW + (kt & 1) * W_STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = kk * 4 + fq;
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm0 + i * 16 + fr;
                fa[i] = *reinterpret_cast<const bf16x8*>(imgA + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = j * 16 + fr;
                fb[j] = *reinterpret_cast<const bf16x8*>(imgB + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };

    // The K loop is unrolled IN FULL and runs NK steps on every workgroup (NK = the K-steps of a slice, a template parameter;
    // the last slice of an uneven split owns fewer: its surplus steps stage clamped re-reads and skip the MFMAs), so every
    // wait count is a compile-time constant.  The two kinds of waves run their own copy of the loop (one barrier per step in
    // each: the counts match), which keeps the weight path one straight-line region -- tools/check_wstream_asm.py verifies
    // in the assembly that nothing touches a weight register between its load and the wait that retires it.
    if (wload) {
#pragma unroll
        for (int j = 0; j < DW; ++j)
            if (j < NK) load_w(wreg[j], j);
        ZE_WS_STORE(wreg[0], 0, PW * ((DW < NK ? DW : NK) - 1));
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            constexpr int dw_ = DW;
            const int u = kt % dw_;
            __builtin_amdgcn_s_barrier();
            // wreg[u] held step kt (written to LDS one iteration ago): refill it DW steps ahead, then put step kt + 1 into
            // the weight stage nobody reads any more; steps kt + 2 .. min(kt + DW, NK - 1) stay in flight
            if (kt + DW < NK) load_w(wreg[u], kt + DW);
            if (kt + 1 < NK) {
                const int last = kt + DW < NK - 1 ? kt + DW : NK - 1;  // newest step requested so far
                switch (last - (kt + 1)) {  // (a literal per case: the asm operand has to be an immediate)
                    case 0: ZE_WS_STORE(wreg[(u + 1) % dw_], kt + 1, 0); break;
                    case 1: ZE_WS_STORE(wreg[(u + 1) % dw_], kt + 1, PW * 1); break;
                    case 2: ZE_WS_STORE(wreg[(u + 1) % dw_], kt + 1, PW * 2); break;
                    case 3: ZE_WS_STORE(wreg[(u + 1) % dw_], kt + 1, PW * 3); break;
                    case 4: ZE_WS_STORE(wreg[(u + 1) % dw_], kt + 1, PW * 4); break;
                    case 5: ZE_WS_STORE(wreg[(u + 1) % dw_], kt + 1, PW * 5); break;
                    case 6: ZE_WS_STORE(wreg[(u + 1) % dw_], kt + 1, PW * 6); break;
                    default: ZE_WS_STORE(wreg[(u + 1) % dw_], kt + 1, PW * 7); break;
                }
            }
            if (kt < nk) compute(kt);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (nothing of the loop's queue survives into the epilogue)
    } else {
#pragma unroll
        for (int s = 0; s < SA - 1; ++s)
            if (s < nk) issue_a(s);
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            // this wave's DMAs of activation stage kt have landed
            const int ahead = min(SA - 2, nk - 1 - kt);
            if (ahead >= 2) ring_wait<2 * LA>();
            else if (ahead == 1) ring_wait<LA>();
            else ring_wait<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + SA - 1 < nk) issue_a(kt + SA - 1);
            if (kt < nk) compute(kt);
        }
    }
#undef ZE_WS_STORE
    __syncthreads();  // the tail reuses the staging LDS
    gemm_finish<BM, BN, EPI, WM, WN, true>(acc, smem, bias, R, ldr, C, ldc, nullptr, M, N, ksplit, ks, bid, nwg, 0, bn0, slab, tickets);
}

template <int BM, int BN, int SA, int DW, int NK, int E>
static void launch_wstream_one(const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R, int ldr,
                               bf16_t* C, int ldc, int M, int N, int K, int ksplit, const ze_gemm_ws& ws, hipStream_t s) {
    const int grid = ze_cdiv(N, BN) * ksplit;
    const size_t lds = (size_t)SA * BM * 128 + (size_t)2 * BN * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_wstream<BM, BN, SA, DW, NK, E>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_gemm_wstream<BM, BN, SA, DW, NK, E>), dim3(grid), dim3(512), lds, s, A, lda, W, ldw, bias, R, ldr, C,
                       ldc, M, N, K, ksplit, ws.slab, ws.tickets);
}

// The one-pass instances (gate/up with SwiGLU, the lm_head, plain) at the K-step count of the 3B hidden size (K = 2048);
// false = no instance for this shape (the caller falls back to the ring kernels, which give the same bits).
template <int BM, int BN, int SA, int DW>
static bool launch_wstream(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R,
                           int ldr, bf16_t* C, int ldc, int M, int N, int K, const ze_gemm_ws& ws, hipStream_t s) {
    if (K / GEMM_BK != 32) return false;
#define ZE_WS_ONE(NK, E) \
    launch_wstream_one<BM, BN, SA, DW, NK, E>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, 1, ws, s); \
    return true
    if (epi == ZE_EPI_SWIGLU) { ZE_WS_ONE(32, ZE_EPI_SWIGLU); }
    if (epi == ZE_EPI_F32) { ZE_WS_ONE(32, ZE_EPI_F32); }
    if (epi == ZE_EPI_NONE) { ZE_WS_ONE(32, ZE_EPI_NONE); }
#undef ZE_WS_ONE
    return false;
}

// ----------------------------------------------------------------------------------------------------------------
// The ring on the block-scaled FP8 matrix instruction (BASELINE configs[4]: "fp8 weights on CDNA4 fp8 MFMA", prefill side;
// ze_set_fp8_activations): C = (A8 * 2^ka) (W8 * 2^kw)^T with A8 [M, K] and W8 [N, K] E4M3 bytes, K contiguous, and ONE
// power-of-two scale per row of each -- the activations' from k_rmsnorm(act8 = 3), the weights' from k_quantize_rows.
// v_mfma_scale_f32_16x16x128_f8f6f4 multiplies 16 x 128 by 128 x 16 bytes per instruction (twice the MACs per cycle of
// the bf16 form) and takes an E8M0 scale per lane and operand: a lane holds 32 consecutive K bytes of its row, and the MX
// format's scale per 32-element block is simply the row's scale here -- the exponent field of the fp32 power of two --
// so the instruction computes exactly the products the bf16 kernels form from the dequantised copies (q_a 2^ka x q_w 2^kw
// is exact in fp32), in another summation order.  Staging is k_gemm_ring's, byte for byte: a 128-element K-step is the same
// 128-B LDS row as 64 bf16, swizzle and all; a fragment is two 16-B chunks (32 bytes) per lane.
typedef __attribute__((ext_vector_type(8))) int mx_i32x8;
// the scale bytes are packed four to a register and picked by op_sel, which has to be a literal: dispatch over the 16
// (byte of A, byte of B) pairs -- the selectors are loop counters of fully unrolled loops, so one case survives
template <int OA, int OB>
__device__ __forceinline__ f32x4 mx_mfma(const mx_i32x8& a, const mx_i32x8& b, const f32x4& c, int ea, int eb) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OA, ea, OB, eb);
}
__device__ __forceinline__ f32x4 mx_mfma_sel(const mx_i32x8& a, const mx_i32x8& b, const f32x4& c, int oa, int ea, int ob, int eb) {
    switch (oa * 4 + ob) {
        case 0: return mx_mfma<0, 0>(a, b, c, ea, eb);
        case 1: return mx_mfma<0, 1>(a, b, c, ea, eb);
        case 2: return mx_mfma<0, 2>(a, b, c, ea, eb);
        case 3: return mx_mfma<0, 3>(a, b, c, ea, eb);
        case 4: return mx_mfma<1, 0>(a, b, c, ea, eb);
        case 5: return mx_mfma<1, 1>(a, b, c, ea, eb);
        case 6: return mx_mfma<1, 2>(a, b, c, ea, eb);
        case 7: return mx_mfma<1, 3>(a, b, c, ea, eb);
        case 8: return mx_mfma<2, 0>(a, b, c, ea, eb);
        case 9: return mx_mfma<2, 1>(a, b, c, ea, eb);
        case 10: return mx_mfma<2, 2>(a, b, c, ea, eb);
        case 11: return mx_mfma<2, 3>(a, b, c, ea, eb);
        case 12: return mx_mfma<3, 0>(a, b, c, ea, eb);
        case 13: return mx_mfma<3, 1>(a, b, c, ea, eb);
        case 14: return mx_mfma<3, 2>(a, b, c, ea, eb);
        default: return mx_mfma<3, 3>(a, b, c, ea, eb);
    }
}

template <int BM, int BN, int STAGES, int EPI, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN) k_gemm_ring_mx(const uint8_t* __restrict__ A, int lda, const float* __restrict__ sa,
                                                      const uint8_t* __restrict__ W, int ldw, const float* __restrict__ sw,
                                                      const bf16_t* __restrict__ bias, const bf16_t* __restrict__ R, int ldr,
                                                      bf16_t* __restrict__ C, int ldc, int M, int N, int K) {
    constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN), NT = 64 * WM * WN;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int LPA = BM / 8 / (WM * WN), LPW = (BM + BN) / 8 / (WM * WN);  // DMA instructions per wave per stage
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int nbx = (N + BN - 1) / BN, nby = (M + BM - 1) / BM;
    const int nwg = nbx * nby;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const bool col_major = N > M;  // (see k_gemm_ring)
    const int bm0 = (col_major ? bid % nby : bid / nbx) * BM, bn0 = (col_major ? bid / nby : bid % nbx) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wid / WN) * (BM / WM), wn0 = (wid % WN) * (BN / WN);
    const int fr = lane & 15, fq = lane >> 4;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // E8M0 scales of this lane's operand rows: the exponent field of the fp32 power of two
    // (four to a register: the instruction's op_sel picks the byte)
    static_assert(TM % 4 == 0 || TM < 4, "scale packing");
    int ea[(TM + 3) / 4], eb[(TN + 3) / 4];
#pragma unroll
    for (int i = 0; i < (TM + 3) / 4; ++i) ea[i] = 0;
#pragma unroll
    for (int j = 0; j < (TN + 3) / 4; ++j) eb[j] = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
        ea[i >> 2] |= (int)(((__float_as_uint(sa[min(bm0 + wm0 + i * 16 + fr, M - 1)]) >> 23) & 0xffu) << (8 * (i & 3)));
#pragma unroll
    for (int j = 0; j < TN; ++j)
        eb[j >> 2] |= (int)(((__float_as_uint(sw[min(bn0 + wn0 + j * 16 + fr, N - 1)]) >> 23) & 0xffu) << (8 * (j & 3)));

    const int nk = K / 128;
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);
    // (the byte matrices go through the bf16 issue helpers as [rows][ld / 2] "elements": the same 16-B chunks)
    const bf16_t* A16 = reinterpret_cast<const bf16_t*>(A);
    const bf16_t* W16 = reinterpret_cast<const bf16_t*>(W);
    const int lda16 = lda >> 1, ldw16 = ldw >> 1;
    auto issue = [&](int kt) {
        const unsigned img = smem_lds + (kt % STAGES) * STAGE_BYTES;
        ring_issue<BM, NT>(A16, lda16, bm0, M - 1, kt * 64, img, wid, lane);
        ring_issue<BN, NT>(W16, ldw16, bn0, N - 1, kt * 64, img + BM * 128, wid, lane);
    };
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue(s);

    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(STAGES - 2, nk - 1 - kt);
        if (ahead >= 2) ring_wait<2 * LPW>();
        else if (ahead == 1) ring_wait<LPW>();
        else ring_wait<0>();
        __builtin_amdgcn_s_barrier();
        const bool more = kt + STAGES - 1 < nk;
        const unsigned img_next = smem_lds + ((kt + STAGES - 1) % STAGES) * STAGE_BYTES;
        const int k_next = (kt + STAGES - 1) * 64;
        const uint8_t* imgA = smem + (kt % STAGES) * STAGE_BYTES;
        const uint8_t* imgB = imgA + BM * 128;
        mx_i32x8 fb[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wn0 + j * 16 + fr, sz = (row >> 1) & 7;
            const u32x4 lo = *reinterpret_cast<const u32x4*>(imgB + row * 128 + (((2 * fq) ^ sz) << 4));
            const u32x4 hi = *reinterpret_cast<const u32x4*>(imgB + row * 128 + (((2 * fq + 1) ^ sz) << 4));
            fb[j] = mx_i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm0 + i * 16 + fr, sz = (row >> 1) & 7;
            const u32x4 lo = *reinterpret_cast<const u32x4*>(imgA + row * 128 + (((2 * fq) ^ sz) << 4));
            const u32x4 hi = *reinterpret_cast<const u32x4*>(imgA + row * 128 + (((2 * fq + 1) ^ sz) << 4));
            const mx_i32x8 fa = mx_i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = mx_mfma_sel(fa, fb[j], acc[i][j], i & 3, ea[i >> 2], j & 3, eb[j >> 2]);
            if (more) {  // the refill of this K-step goes out between the rows of MFMAs (k_gemm_ring, SPREAD)
#pragma unroll
                for (int pc = 0; pc < LPW; ++pc)
                    if ((STAGES == 2 ? min(pc, TM - 1) : pc * TM / LPW) == i) {
                        if (pc < LPA) ring_issue_one(A16, lda16, bm0, M - 1, k_next, img_next, wid + pc * (NT / 64), lane);
                        else ring_issue_one(W16, ldw16, bn0, N - 1, k_next, img_next + BM * 128, wid + (pc - LPA) * (NT / 64), lane);
                    }
            }
        }
    }
    __syncthreads();  // the tail reuses the staging LDS
    gemm_finish<BM, BN, EPI, WM, WN>(acc, smem, bias, R, ldr, C, ldc, nullptr, M, N, 1, 0, bid, nwg, bm0, bn0, nullptr, nullptr);
}

template <int BM, int BN, int ST>
static void launch_ring_mx(int epi, const uint8_t* A, int lda, const float* sa, const uint8_t* W, int ldw, const float* sw,
                           const bf16_t* bias, const bf16_t* R, int ldr, bf16_t* C, int ldc, int M, int N, int K, hipStream_t s) {
    const int grid = ze_cdiv(M, BM) * ze_cdiv(N, BN);
    const size_t lds = (size_t)(BM + BN) * 128 * ST;
#define ZE_MX_LAUNCH(E)                                                                                              \
    do {                                                                                                             \
        static bool attr_set = false;                                                                                \
        if (!attr_set) {                                                                                             \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ring_mx<BM, BN, ST, E, 2, 4>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((k_gemm_ring_mx<BM, BN, ST, E, 2, 4>), dim3(grid), dim3(512), lds, s, A, lda, sa, W, ldw, \
                           sw, bias, R, ldr, C, ldc, M, N, K);                                                       \
    } while (0)
    switch (epi) {
        case ZE_EPI_NONE: ZE_MX_LAUNCH(ZE_EPI_NONE); break;
        case ZE_EPI_SWIGLU: ZE_MX_LAUNCH(ZE_EPI_SWIGLU); break;
        default: break;
    }
#undef ZE_MX_LAUNCH
}

// C = (A8 2^ka) (W8 2^kw)^T (+ bias) (+ SwiGLU): epi NONE or SWIGLU; K a multiple of 128, lda / ldw even.  Tiles as in the
// bf16 policy for many-tile grids: 256 x 256 when the grid is many rounds or a nearly full one, else 128 x 256.
bool ze_launch_gemm_mx(int epi, const uint8_t* A, int lda, const float* sa, const uint8_t* W, int ldw, const float* sw,
                       const bf16_t* bias, bf16_t* C, int ldc, int M, int N, int K, hipStream_t s) {
    if (M <= 0 || N <= 0) return true;
    if (K % 128 != 0 || K / 128 < 2 || (lda & 15) || (ldw & 15) || (epi != ZE_EPI_NONE && epi != ZE_EPI_SWIGLU)) return false;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    const int grid4 = ze_cdiv(M, 256) * ze_cdiv(N, 256);
    const bool big = (2 * grid4 >= 3 * cus) || (grid4 <= cus && 10 * grid4 >= 9 * cus);
    if (big) launch_ring_mx<256, 256, 2>(epi, A, lda, sa, W, ldw, sw, bias, nullptr, 0, C, ldc, M, N, K, s);
    else launch_ring_mx<128, 256, 3>(epi, A, lda, sa, W, ldw, sw, bias, nullptr, 0, C, ldc, M, N, K, s);
    return true;
}


// ----------------------------------------------------------------------------------------------------------------
// Skinny GEMM of the batched decode step: M <= 64 rows (one per chain), every weight byte read ONCE, straight from
// HBM into MFMA operand registers (no LDS staging, no barrier in the loop -- the GEMV of the single-chain step with
// the matrix cores doing the 64 dot products).  A workgroup owns BN = 16 * TN weight rows over its K range; its four
// waves split that range (a quarter each, whole 32-deep MFMA slices), every wave multiplies its quarter against all
// live 16-row tiles of the activations, which it reads from L2 in fragment layout (16 B per lane, like the weights).
// The four partial tiles meet in LDS and are added in wave order; split-K slices across workgroups (long K: the
// down projection) then go through the slab / ticket reduction of gemm_finish.  The order of every sum is a function
// of (K, ksplit) alone, so a chain's result does not depend on how many chains share the step.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;  // an FP8 weight fragment of one lane: 8 bytes
template <int TN, int CH, int MT, bool W8 = false, bool A8 = false>
struct skinny_frag {
    typename std::conditional<A8, u32x2_t, bf16x8>::type a[CH][MT];
    typename std::conditional<W8, u32x2_t, bf16x8>::type b[CH][TN];
};

// FP8 (E4M3) weight fragment -> bf16 MFMA operand: 8 bytes per lane, all of ONE weight row, times that row's
// power-of-two scale (v_cvt_scalef32_pk_bf16_fp8: two values per instruction).  q * 2^k is exact in bf16, so the
// product is the one the dequantised bf16 copy gives, bit for bit, at half the weight bytes.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16x8 deq_fp8x8(u32x2_t w, float scale) {
    union {
        bf16x2_t h[4];
        bf16x8 v;
    } u;
    u.h[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.x, scale, false);
    u.h[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.x, scale, true);
    u.h[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.y, scale, false);
    u.h[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.y, scale, true);
    return u.v;
}

// FRAG: both operands are stored MFMA-fragment-major (k_pack_fragments / k_rmsnorm(frag)), so every fragment load
// of a wave is one contiguous 1-KiB read (the row-major form reads 16 rows x 64 B per load, which the vector memory
// path serves at a fraction of that rate: it is why the row-major kernel only pays on the narrow projections).
// MT: 16-row tiles of the activations the kernel handles (1, 2 or 4: M <= 16 * MT), a compile-time count so that no
// load sits behind a branch -- with `if (tile < live tiles)` around them hipcc has to size every s_waitcnt vmcnt(N) for
// the path that issues the FEWEST loads, which at 64 chains made each MFMA group wait for the prefetched chunk too.
// BAL: the grid is one workgroup per CU and a workgroup takes a contiguous range of 32-row pairs of blocks (gate | up),
// floor(b P / G) .. floor((b + 1) P / G) of the P = N / 32 pairs -- at most TN / 2 of them: the weight bytes spread over
// the CUs as evenly as the pair count allows while every workgroup passes over the activations ONCE.
// W8: the fragment-major weights are FP8 (8 B per lane per fragment, k_pack_fragments8) with one power-of-two scale per
// weight row in `wscale`; they are dequantised in registers (deq_fp8x8) right before the MFMA.
// A8 (with W8): the activations are FP8 fragments as well (k_rmsnorm_row act8 = 2) and the product runs on
// v_mfma_f32_16x16x32_fp8_fp8; `ascale` holds one power-of-two scale per activation row, applied with the weight row's to
// the fp32 sums before the epilogue.
template <int TN, int EPI, bool FRAG = false, int MT = 4, bool BAL = false, bool W8 = false, bool A8 = false>
__global__ void __launch_bounds__(256) k_gemm_skinny(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W,
                                                     int ldw, const bf16_t* __restrict__ bias,
                                                     const bf16_t* __restrict__ R, int ldr, bf16_t* __restrict__ C,
                                                     int ldc, int M, int N, int K, int ksplit,
                                                     float* __restrict__ slab, unsigned* __restrict__ tickets,
                                                     const float* __restrict__ wscale = nullptr,
                                                     const float* __restrict__ ascale = nullptr) {
    static_assert(!W8 || FRAG, "fp8 weights come fragment-major");
    static_assert(!A8 || W8, "fp8 activations go with fp8 weights");
    constexpr int BN = 16 * TN, CH = (TN == 1) ? 4 : 2;  // MFMA slices per prefetch chunk
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int nwg = (N + BN - 1) / BN;
    int ks = blockIdx.x / nwg, bid = blockIdx.x % nwg;
    int bn0 = bid * BN;
    if (BAL) {  // this workgroup's range of 32-row pairs; columns past it are not stored (N is cut to the range's end)
        const int P = N >> 5, G = gridDim.x;
        nwg = G;
        ks = 0;
        bid = blockIdx.x;
        bn0 = (int)(((long)bid * P) / G) * 32;
        N = (int)(((long)(bid + 1) * P) / G) * 32;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int mt = MT;  // 16-row tiles of the activations (rows past M re-read the last row; never stored)

    // 32-deep slices: this workgroup's share of K, then this wave's quarter of it
    const int s_wg = (K / 32) / ksplit;
    const int base = s_wg >> 2, rem = s_wg & 3;
    const int s0 = ks * s_wg + wid * base + min(wid, rem), ns = base + (wid < rem ? 1 : 0);

    // per-slice stride of a fragment pointer, in elements: 32 columns of a row (row-major) or one 1-KiB fragment
    constexpr int SSTR = FRAG ? 512 : 32;
    constexpr int WSTR = W8 ? 256 : SSTR;  // an fp8 fragment is 512 B = 256 bf16-sized elements
    constexpr int ASTR = A8 ? 256 : SSTR;
    const bf16_t* wp[TN];
    const bf16_t* ap[4];
    float wsc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) wsc[j] = 1.0f;
    if (FRAG) {
        const int ns_all = K / 32, nb_last = (N >> 4) - 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int blk = min((bn0 >> 4) + j, nb_last);
            wp[j] = W + ((size_t)blk * ns_all * 64 + lane) * (W8 ? 4 : 8);
            if (W8) wsc[j] = wscale[blk * 16 + fr];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) ap[i] = A + ((size_t)min(i, mt - 1) * ns_all * 64 + lane) * (A8 ? 4 : 8);
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) wp[j] = W + (size_t)min(bn0 + j * 16 + fr, N - 1) * ldw + fq * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) ap[i] = A + (size_t)min(i * 16 + fr, M - 1) * lda + fq * 8;
    }

    f32x4 acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    typedef typename std::conditional<W8, u32x2_t, bf16x8>::type wfrag_t;
    typedef typename std::conditional<A8, u32x2_t, bf16x8>::type afrag_t;
    auto load = [&](skinny_frag<TN, CH, MT, W8, A8>& f, int s) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int k = (s + c) * ASTR, kw = (s + c) * WSTR;
#pragma unroll
            for (int j = 0; j < TN; ++j)  // streamed once: non-temporal
                f.b[c][j] = __builtin_nontemporal_load(reinterpret_cast<const wfrag_t*>(wp[j] + kw));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < mt) f.a[c][i] = *reinterpret_cast<const afrag_t*>(ap[i] + k);
        }
    };
    auto mac = [&](const skinny_frag<TN, CH, MT, W8, A8>& f) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if constexpr (A8) {  // both operands as they came from memory
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < mt) {
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(*reinterpret_cast<const long*>(&f.a[c][i]),
                                                                                   *reinterpret_cast<const long*>(&f.b[c][j]),
                                                                                   acc[i][j], 0, 0, 0);
                    }
            } else {
                bf16x8 fb[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (W8) fb[j] = deq_fp8x8(f.b[c][j], wsc[j]);
                    else fb[j] = f.b[c][j];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < mt) {
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[c][i], fb[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
    };
    const int nch = ns / CH;
    {
        skinny_frag<TN, CH, MT, W8, A8> f0, f1;
        if (nch > 0) load(f0, s0);
        int c = 0;
        for (; c + 2 <= nch; c += 2) {
            load(f1, s0 + (c + 1) * CH);
            mac(f0);
            if (c + 2 < nch) load(f0, s0 + (c + 2) * CH);
            mac(f1);
        }
        if (c < nch) mac(f0);
    }
    for (int s = s0 + nch * CH; s < s0 + ns; ++s) {  // fewer than CH slices left
        const int k = s * ASTR, kw = s * WSTR;
        bf16x8 fb[TN];
        wfrag_t raw[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            raw[j] = __builtin_nontemporal_load(reinterpret_cast<const wfrag_t*>(wp[j] + kw));
            if constexpr (W8 && !A8) fb[j] = deq_fp8x8(raw[j], wsc[j]);
            else if constexpr (!W8) fb[j] = raw[j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < mt) {
                const afrag_t fa = *reinterpret_cast<const afrag_t*>(ap[i] + k);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (A8)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(*reinterpret_cast<const long*>(&fa),
                                                                               *reinterpret_cast<const long*>(&raw[j]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[j], acc[i][j], 0, 0, 0);
                }
            }
    }

    // the four K quarters meet in LDS: [wave][row tile][column tile][lane] float4; wave w then owns row tile w
    f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < mt) {
#pragma unroll
            for (int j = 0; j < TN; ++j) red[((wid * 4 + i) * TN + j) * 64 + lane] = acc[i][j];
        }
    __syncthreads();
    f32x4 out[1][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wid < mt) {
            v = red[((0 * 4 + wid) * TN + j) * 64 + lane];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const f32x4 t = red[((q * 4 + wid) * TN + j) * 64 + lane];
                v[0] += t[0];
                v[1] += t[1];
                v[2] += t[2];
                v[3] += t[3];
            }
        }
        if constexpr (A8) {  // fp8 x fp8 sums -> values: activation-row and weight-row powers of two (exact)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= ascale[min(wid * 16 + fq * 4 + r, M - 1)] * wsc[j];
        }
        out[0][j] = v;
    }
    __syncthreads();  // gemm_finish reuses the LDS for its ticket flag
    gemm_finish<64, BN, EPI, 4, 1>(out, smem, bias, R, ldr, C, ldc, nullptr, M, N, ksplit, ks, bid, nwg, 0, bn0, slab, tickets);
}

// The split-K workspace (fp32 slabs + per-tile tickets) belongs to the calling engine and travels with every call
// (ze_gemm_ws); the launch macros below name its two pointers g_slab / g_tickets.
// one ring instantiation, every epilogue
template <int BM, int BN, int ST, int WM, int WN, bool SPR, bool WIDE = false, bool QKV = false, int BK = GEMM_BK, int KPB = 1>
static void launch_ring_variant(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias,
                                const bf16_t* R, int ldr, bf16_t* C, int ldc, const int* c_rows, int M, int N, int K,
                                hipStream_t s, int ksplit = 1, const ze_gemm_ws& ws = ze_gemm_ws()) {
    float* g_slab = ws.slab;
    unsigned* g_tickets = ws.tickets;
    const int grid = ze_cdiv(M, BM) * ze_cdiv(N, BN) * ksplit;
    const size_t lds = (size_t)(BM + BN) * BK * 2 * ST;
#define ZE_RINGV_LAUNCH(E)                                                                                          \
    do {                                                                                                            \
        static bool attr_set = false;                                                                               \
        if (!attr_set) {                                                                                            \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ring<BM, BN, ST, E, WM, WN, SPR, WIDE, BK, KPB>),  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((k_gemm_ring<BM, BN, ST, E, WM, WN, SPR, WIDE, BK, KPB>), dim3(grid), dim3(64 * WM * WN), lds, s, A, \
                           lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, ksplit, g_slab, g_tickets);          \
    } while (0)
    if constexpr (QKV) {   // (the decode step's qkv projection with rope + KV append)
        if (epi == ZE_EPI_QKV_ROPE) ZE_RINGV_LAUNCH(ZE_EPI_QKV_ROPE);
    } else if constexpr (WIDE) {  // the wide-store epilogue (gemm_finish LDS_EPI): 16-byte row pieces through the idle staging LDS
        switch (epi) {
            case ZE_EPI_NONE: ZE_RINGV_LAUNCH(ZE_EPI_NONE); break;
            case ZE_EPI_GELU: ZE_RINGV_LAUNCH(ZE_EPI_GELU); break;
            case ZE_EPI_RESIDUAL: ZE_RINGV_LAUNCH(ZE_EPI_RESIDUAL); break;
            case ZE_EPI_SWIGLU: ZE_RINGV_LAUNCH(ZE_EPI_SWIGLU); break;
            case ZE_EPI_F32: ZE_RINGV_LAUNCH(ZE_EPI_F32); break;
        }
    } else {
        // Every caller without a row scatter and with 16-byte-aligned rows takes the wide-store instantiation of the same tile
        // (round 4: the two-byte stores of the plain epilogue were a fifth of a 256 x 256 tile's time at K = 1280; same values,
        // same roundings: the same bits).  knob 7 = 10: plain epilogue everywhere, for A/B runs.  Not for slab-only split-K
        // launches (no epilogue runs) and not where the staged tile would not fit the ring's LDS.
        {
            const bool sw = epi == ZE_EPI_SWIGLU, f32 = epi == ZE_EPI_F32;
            const size_t stage_b = (size_t)BM * ((sw ? BN / 2 : BN) * (f32 ? 4 : 2) + 16);
            const bool wide_ok = ze_gemv_knobs[7] != 10 && c_rows == nullptr && !(ksplit > 1 && g_tickets == nullptr) && stage_b <= lds &&
                                 (ldc % 8) == 0 && ((size_t)C % 16) == 0 && ((sw ? N / 2 : N) % 8) == 0 && (!sw || BN % 32 == 0) &&
                                 (epi != ZE_EPI_RESIDUAL || ((ldr % 8) == 0 && ((size_t)R % 16) == 0)) && epi != ZE_EPI_QKV_ROPE;
            if (wide_ok) {
                launch_ring_variant<BM, BN, ST, WM, WN, SPR, true, false, BK, KPB>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, ksplit, ws);
                return;
            }
        }
        switch (epi) {
            case ZE_EPI_NONE: ZE_RINGV_LAUNCH(ZE_EPI_NONE); break;
            case ZE_EPI_GELU: ZE_RINGV_LAUNCH(ZE_EPI_GELU); break;
            case ZE_EPI_RESIDUAL: ZE_RINGV_LAUNCH(ZE_EPI_RESIDUAL); break;
            case ZE_EPI_SWIGLU: ZE_RINGV_LAUNCH(ZE_EPI_SWIGLU); break;
            case ZE_EPI_F32: ZE_RINGV_LAUNCH(ZE_EPI_F32); break;
        }
    }
#undef ZE_RINGV_LAUNCH
}

// Round 6, BUILT, MEASURED, NOT SHIPPED (default off; knob 20 = 3 turns it on for A/B runs and its bit-equality test):
// the long-K projection of a prefill pass -- the down projection, K = 11008 (3B) / 18944 (7B) -- in THREE K slices on every kernel of
// the prefill family.  The idea: N = hidden is 8 column tiles of 256, so a pass has few, long tiles (408 of ~290 us at 12.8 K rows: 1.6
// rounds of workgroups; 168 at 5.3 K rows: two thirds of the chip); with three slices the units are a third as long and three times
// as many (1224 = 4.8 rounds; 504 = 1.97), and the other lane's decode kernels, which get CUs at unit boundaries, wait a third as long.
// It keeps every invariance: the count is a function of K ALONE -- never of M, of the tile or of the kernel -- the slices are cut at the
// same K-tiles everywhere (ceil(K / 64 / 3) per slice) and added in slice order from zero, so prefill rows stay independent of their
// pass (the whole GPU suite is green with it on).  What it costs: every unit parks 256 KB of fp32 accumulators write-through and the
// tile's last arriver reads three of them back -- 630 MB of slab traffic for a 12.8 K-row pass whose operands are 330 MB -- behind a
// drain of the next unit's prefetch.  Measured on the replayed passes (rocprofv3, profiles/r06_prefill_by_shape_splitk.csv against
// r06_prefill_by_shape.csv): down 469 -> 725 us at 12,832 rows, 217 -> 284 at 5,280; the question stream, same box, two runs each:
// 86.75 / 86.1 unsplit, 82.2 / 82.6 split.  The grid's idle tail is cheaper than the slabs.  Needs the engine's prefill slabs (ze_gemm_ws).
static int ze_prefill_ksplit(int K, bool have_slab) {
    return (have_slab && ze_gemv_knobs[20] == 3 && K > 4096 && K % GEMM_BK == 0 && K / GEMM_BK >= 3 * 8) ? 3 : 1;
}
static int ze_prefill_split_dropped(int M, int N, int K, int ksplit) {
    static bool told = false;
    if (!told) {
        told = true;
        fprintf(stderr, "zoomearth: prefill split-K workspace too small for M=%d N=%d K=%d (ksplit %d -> 1): this call sums K in one run, "
                        "engine passes in three slices\n", M, N, K, ksplit);
    }
    return 1;
}

template <int BM, int BN>
static void launch_cfg(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias,
                       const bf16_t* R, int ldr, bf16_t* C, int ldc, const int* c_rows, int M, int N, int K,
                       hipStream_t s, bool stream_mode, const ze_gemm_ws& ws = ze_gemm_ws()) {
    float* g_slab = ws.slab;
    unsigned* g_tickets = ws.tickets;
    const size_t g_slab_floats = ws.slab_floats;
    const int g_ticket_cap = ws.ticket_cap;
    const int nwg = ze_cdiv(M, BM) * ze_cdiv(N, BN);
    // Split K only in weight-streaming mode (batched decode, few rows): the slice count is a function of (N, K)
    // alone -- computed for ONE row tile -- so a chain's result never depends on how many chains share the step.
    // Prefill / ViT never split: their accumulation order must not depend on M (prefix-KV reuse is bit-exact).
    int ksplit = 1;
    if (stream_mode) {
        const int nk = ze_cdiv(K, GEMM_BK);
        const int tiles_n = ze_cdiv(N, BN);
        while (tiles_n * ksplit < 200 && ksplit < 8 && nk / (ksplit * 2) >= 4) ksplit *= 2;
    } else {
        ksplit = ze_prefill_ksplit(K, g_slab != nullptr);   // (round 6) a function of K alone: the same on every tile and kernel
    }
    if (ksplit > 1 && (!g_slab || (size_t)ksplit * nwg * BM * BN > g_slab_floats || nwg > g_ticket_cap)) {
        // never silently: one slice instead of `ksplit` changes the order of every sum of this projection, i.e. a chain's bits
        // would follow the row count.  An engine sizes its slabs for its max_seqs (ze_engine_create), so this is a caller
        // without a workspace (unit ops) or a bug.
        static bool told = false;
        if (g_slab && !told) {
            told = true;
            fprintf(stderr, "zoomearth: split-K workspace too small for M=%d N=%d K=%d (ksplit %d -> 1): %s\n", M, N, K, ksplit,
                    stream_mode ? "results no longer batch-invariant" : "this call sums K in one run, engine passes in three slices");
        }
        ksplit = 1;
    }
    const int grid = nwg * ksplit;
    // The ring wins where a launch is at most one round of workgroups (few tiles, long K: o / down / merger / ViT
    // proj, 1.3-1.9x) and loses 15-25 % to the register-staged kernel on grids of many rounds, which hide latency
    // with 2-3 co-resident workgroups per CU instead (qkv, gate/up): measured per shape, tools/bench_gemm.py.
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    // (weight-streaming mode, rows = chains of a batched decode step: the ring wins at every grid, 5.04 vs 5.32 ms per
    //  step at 64 chains, 3.78 vs 4.17 at 8)
    // Many-tile grids (128 x 128 policy): eight waves (2 x 4) on the LDS-DMA ring, one workgroup per CU, the refill
    // DMAs of a K-step spread between its rows of MFMAs (SPREAD: a burst of 6-8 DMAs per wave right behind the barrier
    // leaves the matrix pipe idle while every wave of the workgroup is issuing).
    //   * 128 x 256 tiles, three stages (64 x 64 per wave): 4096^3 926 TFLOP/s (870 unspread, 756 on the register-
    //     staged 128 x 128 kernel), gate/up at M = 802 / 518: 112 / 73.5 us (124 / 81 unspread), ViT qkv 29.2 (31.5),
    //     ViT gate/up on its 1.2-round grid 54.3 (59.1 register-staged).
    //   * 256 x 256 tiles, two 64-KB stages (128 x 64 per wave, 253 VGPRs): half the bytes staged per FLOP; wins where
    //     the grid is many rounds -- the batched prefill (16 chains x 802 rows): gate/up 1487 -> 1250 us (926
    //     TFLOP/s), down 663 -> 561 (1030), qkv 165 -> 141 (955); 4096^3 1018 -- and loses to quantisation on the
    //     single-chain grids (344 tiles = 1.3 rounds at M = 802).
    //   * tried on the 256 x 256 tile and dropped (round 2, tools/bench_gemm_shapes.py, random operands): four 32-deep
    //     stages instead of two 64-deep ones (three K-steps in flight, twice the barriers) 5-10 % slower; the fragment
    //     reads software-pipelined by hand across the barrier (next half-step's twelve ds_read_b128 ahead of the current
    //     32 MFMAs) within 1 %; the loop with NO refill DMAs at all 1.30 PFLOP/s at 4096^3 against 1.17 with them: the
    //     loop is within 10 % of its own no-memory ceiling, the rest of the gap to the 2.5 PFLOP/s peak is clock under
    //     matrix load (zero-filled operands run 15-20 % faster than random ones: MI355X_MICROARCH.md).
    // All of them accumulate an output element in the same K order: results are bit-identical across the choices.
    // knob 7: 3 = register-staged only, 4 = always 256 x 256, 6 = always 128 x 256.
    if (BM == 128 && BN == 128 && (ksplit == 1 || !stream_mode) && K % GEMM_BK == 0 && K / GEMM_BK >= 4 && ze_gemv_knobs[7] != 3 &&
        ze_gemv_knobs[6] == 0) {
        static int cus8 = 0;
        if (!cus8) {
            int dev = 0;
            hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&cus8, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus8 <= 0) cus8 = 256;
        }
        const int grid4 = ze_cdiv(M, 256) * ze_cdiv(N, 256);
        const bool big = (2 * grid4 >= 3 * cus8) || (grid4 <= cus8 && 10 * grid4 >= 9 * cus8);
        // (the 256 x 256 tile runs on the eight-phase kernel; knob 7 = 4: on the two-stage ring, 8: eight-phase at every grid)
        const bool p8_ok = (lda % 8) == 0 && (ldw % 8) == 0 && (size_t)M * lda < ((size_t)1 << 31) && (size_t)N * ldw < ((size_t)1 << 31);
        // its grid against the 128 x 256 ring's, in rounds of workgroups: a 128 x 256 tile takes ~0.6 of a 256 x 256 one
        // (measured on the 3B prefill / ViT shapes at 1-13 K rows, tools/bench_prefill_shapes.py: e.g. 192 tiles of 256 x 256 =
        // 0.75 round beat 384 of 128 x 256 = 2 rounds, 215 against 291 us for down at 6144 rows; 128 tiles lose, 210 against 160)
        const int r8 = ze_cdiv(grid4, cus8), ra = ze_cdiv(2 * grid4, cus8);
        const bool p8_wins = 10 * r8 <= 6 * ra;
        // (a split K -- ze_prefill_ksplit -- rides along: the slab holds ksplit x M_pad x N_pad floats whatever the tile; the 256 x 256
        //  tiles need their own share of the capacity check above, which was made for BM x BN = 128 x 128)
        const bool fits256 = ksplit == 1 || ((size_t)ksplit * grid4 * 256 * 256 <= g_slab_floats && grid4 <= g_ticket_cap);
        const bool fits128x256 = ksplit == 1 || ((size_t)ksplit * ze_cdiv(M, 128) * ze_cdiv(N, 256) * 128 * 256 <= g_slab_floats);
        if (fits256 && p8_ok && (ze_gemv_knobs[7] == 8 || ((ze_gemv_knobs[7] == 0 || ze_gemv_knobs[7] == 9) && p8_wins)))
            launch_p8(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, ksplit, ws);
        else if (fits256 && (ze_gemv_knobs[7] == 4 || (ze_gemv_knobs[7] != 6 && big)))
            launch_ring_variant<256, 256, 2, 2, 4, true>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, ksplit, ws);
        else if (fits128x256)
            launch_ring_variant<128, 256, 3, 2, 4, true>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, ksplit, ws);
        else
            launch_ring_variant<128, 128, 4, 2, 4, true>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, ksplit, ws);
        return;
    }
    const bool ring = ze_gemv_knobs[6] == 2 || (ze_gemv_knobs[6] == 0 && (grid <= cus || stream_mode));
    if (K % GEMM_BK == 0 && K / GEMM_BK / ksplit >= 4 && ring) {
        // LDS-DMA ring: four stages when they fit beside nothing else (one workgroup per CU), else three
        constexpr int STAGES = ((BM + BN) * 128 * 4 <= 128 * 1024) ? 4 : 3;
        const size_t lds_ring = (size_t)(BM + BN) * 128 * STAGES;
        // (SPREAD on these four-wave tiles, one wave per SIMD, is 10-25 % slower: o 24.7 vs 20.9 us, down 91.6 vs 73.5)
        // 64 x 128 tiles of a one-round grid: eight waves (2 x 4, 32 x 32 per wave), two per SIMD, hide each other's DMA
        // issue and LDS latency: o 20.4 -> 16.9 us, down 73.4 -> 62.7 (M = 802), 72.7 -> 58.4 (M = 518), ViT proj
        // 15.6 -> 12.4, ViT down 27.3 -> 22.9, merger 37.9 -> 31.4 (1 x 8 waves and the spread refill are 2-8 % behind)
        if (BM == 64 && BN == 128 && (ksplit == 1 || !stream_mode) && ze_gemv_knobs[7] != 3) {
            launch_ring_variant<64, 128, 4, 2, 4, false>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, ksplit, ws);
            return;
        }
        // the same on 64 x 64 tiles (4 x 2 waves, 16 x 32 per wave: the SwiGLU epilogue pairs two 16-column tiles of a
        // wave), split-K included: merger 26.8 -> 23.3 us; the weight-streaming
        // GEMMs of the batched decode step 4.48 -> 4.33 ms per 64-chain step, 3.27 -> 3.14 at 8 chains
        if (BM == 64 && BN == 64 && ze_gemv_knobs[7] != 3) {
            // (round 3, the split-K stream of the down projection, us at 64 / 256 rows: this tile 17.4 / 35.2; eight or six
            //  stages 17.8 / 44.1 and 17.7 / 42.9; 128 x 64 tiles 21.3 / 36.7; 256 x 64 tiles, three stages 32.8 / 39.1)
            // (qkv / o of a decode step -- 128-240 tiles of 32 K-steps, at most one workgroup per CU -- with EIGHT stages, 112 KB in
            //  flight per CU instead of 48: 13.8 against 12.9 us; these launches are bound by the per-step chain wait -> barrier ->
            //  fragment reads -> 4 MFMAs, not by bytes in flight)
            launch_ring_variant<64, 64, 4, 4, 2, false>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, ksplit, ws);
            return;
        }
#define ZE_RING_LAUNCH(E)                                                                                              \
    do {                                                                                                               \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ring<BM, BN, STAGES, E>),                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ring);                            \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL((k_gemm_ring<BM, BN, STAGES, E>), dim3(grid), dim3(256), lds_ring, s, A, lda, W, ldw, bias, \
                           R, ldr, C, ldc, c_rows, M, N, K, ksplit, g_slab, g_tickets);                                \
    } while (0)
        switch (epi) {
            case ZE_EPI_NONE: ZE_RING_LAUNCH(ZE_EPI_NONE); break;
            case ZE_EPI_GELU: ZE_RING_LAUNCH(ZE_EPI_GELU); break;
            case ZE_EPI_RESIDUAL: ZE_RING_LAUNCH(ZE_EPI_RESIDUAL); break;
            case ZE_EPI_SWIGLU: ZE_RING_LAUNCH(ZE_EPI_SWIGLU); break;
            case ZE_EPI_F32: ZE_RING_LAUNCH(ZE_EPI_F32); break;
        }
#undef ZE_RING_LAUNCH
        return;
    }
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

// Fragment-major operands, no split-K.  What bounds these launches is the memory traffic a CU can keep in flight
// (about 25 GB/s of HBM misses per CU, the figure the decode GEMVs also show), so the weight bytes have to spread
// evenly over all 256 CUs: with 64 rows per workgroup gate/up is 344 workgroups and the CUs that get two of them set
// the time (22.6 us at 8 chains); 32 rows (688 workgroups) run 18.3 us, against 23.7 on the LDS-DMA ring and 15.1 for
// the single-chain GEMV.  Every workgroup passes over ALL activation rows (from L2), which is what 64 chains pay for
// the finer grid: 24.1 us (26.8 at 64 rows per workgroup, 25.1 at 128, 27 on the ring).  Narrow matrices (qkv) take
// 16 rows per workgroup so that their grid still covers the chip: 5.5 / 8.1 us at 8 / 64 chains (9.8 / 13.9 row-major,
// 15 on the ring).  lm_head: 96 us at 8 chains (120 on the ring), 140 at 64 (145).
template <int TN, int MT, bool W8 = false, bool A8 = false>
static void launch_frag_mt(int epi, const bf16_t* Xf, const bf16_t* Wf, const bf16_t* bias, const bf16_t* R, int ldr,
                        bf16_t* C, int ldc, int M, int N, int K, hipStream_t s, int ksplit, const ze_gemm_ws& ws,
                        const float* wscale = nullptr, const float* ascale = nullptr) {
    float* g_slab = ws.slab;
    unsigned* g_tickets = ws.tickets;
    const int grid = ze_cdiv(N, 16 * TN) * ksplit;
    // A grid of at most one workgroup per CU asks for more LDS than half a CU has, so that the dispatcher cannot put
    // two of them on one CU and leave another idle (the launch time is the bytes of the busiest CU).
    static int cus_f = 0;
    if (!cus_f) {
        int dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus_f, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus_f <= 0) cus_f = 256;
    }
    const size_t lds = (grid <= cus_f) ? (size_t)96 * 1024 : (size_t)16 * TN * 1024;
    const size_t lds_max = (size_t)96 * 1024;
#define ZE_FRAG_LAUNCH(E)                                                                                           \
    do {                                                                                                            \
        static bool attr_set = false;                                                                               \
        if (!attr_set) {                                                                                            \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_skinny<TN, E, true, MT, false, W8, A8>),      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);                          \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((k_gemm_skinny<TN, E, true, MT, false, W8, A8>), dim3(grid), dim3(256), lds, s, Xf, 0,   \
                           Wf, 0, bias, R, ldr,                                                                     \
                           C, ldc, M, N, K, ksplit, g_slab, g_tickets, wscale, ascale);                             \
    } while (0)
    if constexpr (TN % 2 == 0) {
        if (epi == ZE_EPI_SWIGLU) {
            ZE_FRAG_LAUNCH(ZE_EPI_SWIGLU);
            return;
        }
    }
    if constexpr (A8) return;  // fp8 activations: gate/up only
    switch (epi) {
        case ZE_EPI_NONE: ZE_FRAG_LAUNCH(ZE_EPI_NONE); break;
        case ZE_EPI_RESIDUAL: ZE_FRAG_LAUNCH(ZE_EPI_RESIDUAL); break;
        case ZE_EPI_F32: ZE_FRAG_LAUNCH(ZE_EPI_F32); break;
        default: break;
    }
#undef ZE_FRAG_LAUNCH
}

template <int TN>
static void launch_frag(int epi, const bf16_t* Xf, const bf16_t* Wf, const bf16_t* bias, const bf16_t* R, int ldr,
                        bf16_t* C, int ldc, int M, int N, int K, hipStream_t s, int ksplit = 1,
                        const ze_gemm_ws& ws = ze_gemm_ws(), const float* wscale = nullptr, const float* ascale = nullptr) {
    if constexpr (TN == 2) {
        if (wscale && ascale && epi == ZE_EPI_SWIGLU && ksplit == 1) {  // fp8 x fp8
            if (M <= 16) launch_frag_mt<TN, 1, true, true>(epi, Xf, Wf, bias, R, ldr, C, ldc, M, N, K, s, 1, ws, wscale, ascale);
            else if (M <= 32) launch_frag_mt<TN, 2, true, true>(epi, Xf, Wf, bias, R, ldr, C, ldc, M, N, K, s, 1, ws, wscale, ascale);
            else launch_frag_mt<TN, 4, true, true>(epi, Xf, Wf, bias, R, ldr, C, ldc, M, N, K, s, 1, ws, wscale, ascale);
            return;
        }
    }
    if (wscale) {  // fp8 fragment stream
        if (M <= 16) launch_frag_mt<TN, 1, true>(epi, Xf, Wf, bias, R, ldr, C, ldc, M, N, K, s, ksplit, ws, wscale);
        else if (M <= 32) launch_frag_mt<TN, 2, true>(epi, Xf, Wf, bias, R, ldr, C, ldc, M, N, K, s, ksplit, ws, wscale);
        else launch_frag_mt<TN, 4, true>(epi, Xf, Wf, bias, R, ldr, C, ldc, M, N, K, s, ksplit, ws, wscale);
        return;
    }
    if (M <= 16) launch_frag_mt<TN, 1>(epi, Xf, Wf, bias, R, ldr, C, ldc, M, N, K, s, ksplit, ws);
    else if (M <= 32) launch_frag_mt<TN, 2>(epi, Xf, Wf, bias, R, ldr, C, ldc, M, N, K, s, ksplit, ws);
    else launch_frag_mt<TN, 4>(epi, Xf, Wf, bias, R, ldr, C, ldc, M, N, K, s, ksplit, ws);
}

// gate/up (SwiGLU, wide): one workgroup per CU (or a multiple), each with a balanced range of at most three 32-row
// pairs and ONE pass over the activations
template <int MT, bool W8 = false, bool A8 = false>
static void launch_frag_balanced(const bf16_t* Xf, const bf16_t* Wf, const bf16_t* bias, bf16_t* C, int ldc, int M, int N,
                                 int K, int grid, hipStream_t s, const float* wscale = nullptr, const float* ascale = nullptr) {
    const size_t lds = (size_t)16 * 6 * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_skinny<6, ZE_EPI_SWIGLU, true, MT, true, W8, A8>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_gemm_skinny<6, ZE_EPI_SWIGLU, true, MT, true, W8, A8>), dim3(grid), dim3(256), lds, s, Xf, 0, Wf, 0,
                       bias, nullptr, 0, C, ldc, M, N, K, 1, nullptr, nullptr, wscale, ascale);
}

void ze_launch_gemm_frag(int epi, const bf16_t* Xf, const bf16_t* Wf, const bf16_t* bias, const bf16_t* R, int ldr,
                         bf16_t* C, int ldc, int M, int N, int K, hipStream_t s, const float* wscale, const float* ascale) {
    if (M <= 0 || N <= 0) return;
    // (more than 32 chains only: below that the activation pass is small and the finer 32-row grid wins, 2.84 against
    //  2.91 ms per step at 8 chains; at 64 chains 3.86 against 3.95.  Both forms add an output's K quarters in the same
    //  order, so the choice does not change a chain's result.)
    if (epi == ZE_EPI_SWIGLU && N > 4096 && N % 32 == 0 && M > 32 && ze_gemv_knobs[5] != 2) {
        static int cus_b = 0;
        if (!cus_b) {
            int dev = 0;
            hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&cus_b, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus_b <= 0) cus_b = 256;
        }
        const int P = N / 32;
        const int grid = cus_b * ze_cdiv(P, 3 * cus_b);  // at most three pairs per workgroup
        if (wscale && ascale) launch_frag_balanced<4, true, true>(Xf, Wf, bias, C, ldc, M, N, K, grid, s, wscale, ascale);
        else if (wscale) launch_frag_balanced<4, true>(Xf, Wf, bias, C, ldc, M, N, K, grid, s, wscale);
        else launch_frag_balanced<4>(Xf, Wf, bias, C, ldc, M, N, K, grid, s);
        return;
    }
    if (N <= 4096 && epi != ZE_EPI_SWIGLU)
        launch_frag<1>(epi, Xf, Wf, bias, R, ldr, C, ldc, M, N, K, s, 1, ze_gemm_ws(), wscale);
    else
        launch_frag<2>(epi, Xf, Wf, bias, R, ldr, C, ldc, M, N, K, s, 1, ze_gemm_ws(), wscale, ascale);
}

// Batched decode: the skinny kernel when the shape allows (M <= 64, whole MFMA slices), else the ring.
template <int TN, int MT>
static void launch_skinny_mt(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R,
                          int ldr, bf16_t* C, int ldc, int M, int N, int K, int ksplit, hipStream_t s) {
    float* g_slab = nullptr;  // never split: no workspace
    unsigned* g_tickets = nullptr;
    const int grid = ze_cdiv(N, 16 * TN) * ksplit;
    const size_t lds = (size_t)16 * TN * 1024;
#define ZE_SKINNY_LAUNCH(E)                                                                                        \
    do {                                                                                                           \
        static bool attr_set = false;                                                                              \
        if (!attr_set) {                                                                                           \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_skinny<TN, E, false, MT>),                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                             \
            attr_set = true;                                                                                       \
        }                                                                                                          \
        hipLaunchKernelGGL((k_gemm_skinny<TN, E, false, MT>), dim3(grid), dim3(256), lds, s, A, lda, W, ldw, bias,  \
                           R, ldr, C,                                                                               \
                           ldc, M, N, K, ksplit, g_slab, g_tickets);                                               \
    } while (0)
    switch (epi) {
        case ZE_EPI_NONE: ZE_SKINNY_LAUNCH(ZE_EPI_NONE); break;
        case ZE_EPI_GELU: ZE_SKINNY_LAUNCH(ZE_EPI_GELU); break;
        case ZE_EPI_RESIDUAL: ZE_SKINNY_LAUNCH(ZE_EPI_RESIDUAL); break;
        case ZE_EPI_F32: ZE_SKINNY_LAUNCH(ZE_EPI_F32); break;
        default: break;  // SWIGLU pairs 16-row blocks: never routed here
    }
#undef ZE_SKINNY_LAUNCH
}

template <int TN>
static void launch_skinny(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R,
                          int ldr, bf16_t* C, int ldc, int M, int N, int K, int ksplit, hipStream_t s) {
    if (M <= 16) launch_skinny_mt<TN, 1>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, ksplit, s);
    else if (M <= 32) launch_skinny_mt<TN, 2>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, ksplit, s);
    else launch_skinny_mt<TN, 4>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, ksplit, s);
}

// ----------------------------------------------------------------------------------------------------------------
// Second launch of the two-launch split-K form (the down projection of a batched decode step with more than 64 chains):
// k_gemm_ring<BM, BN, ...> ran with a null ticket array and left every slice's fp32 accumulators in the slabs, laid out
// [slice][tile][MFMA tile t][thread] exactly as gemm_finish parks them.  One thread here = one thread of the producing
// workgroup for one MFMA tile: it adds the slices IN SLICE ORDER starting from zero and applies the epilogue -- the
// arithmetic of gemm_finish's last-arriver reduction, value for value, so the result is bit-identical to the one-launch
// form on any tile size (a row's bits do not depend on which form served the step).  The whole chip reduces (16.8 MB of
// slabs at 256 rows) instead of one workgroup per tile reading 1 MB.
template <int BM, int BN, int WM, int WN, int EPI>
__global__ void __launch_bounds__(64 * WM * WN) k_splitk_reduce(const float* __restrict__ slab, int ksplit,
                                                                 const bf16_t* __restrict__ bias, const bf16_t* __restrict__ R,
                                                                 int ldr, bf16_t* __restrict__ C, int ldc, int M, int N) {
    constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN), NT = 64 * WM * WN;
    const int nbx = (N + BN - 1) / BN, nby = (M + BM - 1) / BM, nwg = nbx * nby;
    const int bid = blockIdx.x / (TM * TN), t = blockIdx.x % (TM * TN);
    const int i = t / TN, j = t % TN;
    const bool col_major = N > M;  // (k_gemm_ring's walk of the tile grid)
    const int bm0 = (col_major ? bid % nby : bid / nbx) * BM, bn0 = (col_major ? bid / nby : bid % nbx) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid / WN) * (BM / WM), wn0 = (wid % WN) * (BN / WN);
    const int fr = lane & 15, fq = lane >> 4;
    const int col = bn0 + wn0 + j * 16 + fr;
    const int row0 = bm0 + wm0 + i * 16 + fq * 4;
    if (row0 >= M) return;  // (whole waves of a ragged last row tile leave before they load anything)
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const float4* sl = reinterpret_cast<const float4*>(slab);
    float4 v[8];
    for (int q0 = 0; q0 < ksplit; q0 += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            v[q] = sl[((size_t)(min(q0 + q, ksplit - 1) * nwg + bid) * (TM * TN) + t) * NT + tid];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q0 + q < ksplit) {
                acc[0] += v[q].x;
                acc[1] += v[q].y;
                acc[2] += v[q].z;
                acc[3] += v[q].w;
            }
    }
    if (col >= N) return;
    const float b = bias ? bf16_to_f32(bias[col]) : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + r;
        if (row >= M) continue;
        float o = bf16_round(acc[r] + b);
        if (EPI == ZE_EPI_RESIDUAL) o = bf16_to_f32(R[(size_t)row * ldr + col]) + o;
        C[(size_t)row * ldc + col] = f32_to_bf16(o);
    }
}

// (Round 5, tried and dropped: a reducer that owns whole rows -- four rows x all N columns per workgroup, coalesced slab reads, the rows
// through LDS into k_rmsnorm_row's element order -- and also writes the next layer's normalised input.  Bit-identical to the two
// launches, but M / 4 workgroups are too few to keep the slab reads going: 52.6 us against 48.1 + 3.7 for the pair at 576 rows, 30.7
// against 21.1 + 3.5 at 128, and the question stream did not move.  DESIGN.md 7f.)
// the slice count of the weight-streaming split (launch_cfg, stream mode, 64-column tiles): a function of (N, K) alone
static int stream_ksplit(int N, int K) {
    int ksplit = 1;
    const int nk = ze_cdiv(K, GEMM_BK), tiles_n = ze_cdiv(N, 64);
    while (tiles_n * ksplit < 200 && ksplit < 8 && nk / (ksplit * 2) >= 4) ksplit *= 2;
    return ksplit;
}

// long-K projection, more than 64 rows: big tiles on the slices of the one-launch form, then the chip-wide reduction;
// false = does not apply
template <int BM, int BN, int ST, bool SPR, int WM = 2, int WN = 4, int BK = GEMM_BK>
static bool launch_splitk_two(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R,
                              int ldr, bf16_t* C, int ldc, int M, int N, int K, const ze_gemm_ws& ws, hipStream_t s) {
    if (epi != ZE_EPI_RESIDUAL && epi != ZE_EPI_NONE) return false;
    const int ksplit = stream_ksplit(N, K);
    const int nwg = ze_cdiv(M, BM) * ze_cdiv(N, BN);
    if (ksplit < 2 || K % GEMM_BK != 0 || K / GEMM_BK / ksplit < 4 || !ws.slab || (size_t)ksplit * nwg * BM * BN > ws.slab_floats)
        return false;
    ze_gemm_ws slabs_only = ws;
    slabs_only.tickets = nullptr;
    launch_ring_variant<BM, BN, ST, WM, WN, SPR, false, false, BK>(ZE_EPI_NONE, A, lda, W, ldw, nullptr, nullptr, 0, C, ldc, nullptr, M, N, K, s, ksplit, slabs_only);
    const int grid = nwg * (BM / (16 * WM)) * (BN / (16 * WN));
    if (epi == ZE_EPI_RESIDUAL)
        hipLaunchKernelGGL((k_splitk_reduce<BM, BN, WM, WN, ZE_EPI_RESIDUAL>), dim3(grid), dim3(64 * WM * WN), 0, s, ws.slab, ksplit, bias, R, ldr, C, ldc, M, N);
    else
        hipLaunchKernelGGL((k_splitk_reduce<BM, BN, WM, WN, ZE_EPI_NONE>), dim3(grid), dim3(64 * WM * WN), 0, s, ws.slab, ksplit, bias, R, ldr, C, ldc, M, N);
    return true;
}

// dst row p of every 128-row head <- src row head * 128 + dim(p): blocks of 32 = [d0 .. d0 + 15 | 64 + d0 .. 64 + d0 + 15]
__global__ void __launch_bounds__(256) k_permute_qkv(const bf16_t* __restrict__ W, int ldw, const bf16_t* __restrict__ bias, int K,
                                                     bf16_t* __restrict__ Wp, bf16_t* __restrict__ bias_p) {
    const int p = blockIdx.x, w = p & 127, i = w & 31;
    const int dim = i < 16 ? (w >> 5) * 16 + i : 64 + (w >> 5) * 16 + (i - 16);
    const int src = (p & ~127) + dim;
    for (int c = threadIdx.x * 8; c < K; c += 256 * 8)
        *reinterpret_cast<uint4*>(Wp + (size_t)p * K + c) = *reinterpret_cast<const uint4*>(W + (size_t)src * ldw + c);
    if (threadIdx.x == 0 && bias && bias_p) bias_p[p] = bias[src];
}
void ze_launch_permute_qkv(const bf16_t* W, int ldw, const bf16_t* bias, int n_heads_total, int K, bf16_t* Wp, bf16_t* bias_p,
                           hipStream_t s) {
    if (n_heads_total <= 0) return;
    k_permute_qkv<<<n_heads_total * 128, 256, 0, s>>>(W, ldw, bias, K, Wp, bias_p);
}

// Tall tiles for the narrow projections of a decode step (qkv, o: N <= 4096, K = hidden) at 385 .. 768 rows (round 5).  These
// launches are bound by what a CU takes in from L2 (10.5 MB of weights; every workgroup stages its rows of A and of W
// over the whole K): 64 x 64 tiles are the cheapest tile per output (512 KB staged per 4096 outputs) -- but 7 .. 12 row tiles x 40
// column tiles are 280 .. 480 workgroups, so 24 .. 224 CUs take in TWO tiles (1 MB) while the rest wait.  A tile of BM x 64 with
// BM = 80 / 96 / 112 / 128 -- BM / 16 x 2 waves, every wave the same 16 x 32 outputs as on the 64 x 64 tile -- makes the grid ONE
// round of at most 256 workgroups of (BM + 64) x 4 KB each: 576 KB on the busiest CU instead of 1 MB at 399 rows, 640 KB at 576.
// K in sequence per output element on every tile: the same bits (test_linear_wide_decode).  false = 64-row tiles already fit one round
// (or the shape is not a narrow projection): the caller's old rule applies.  knob 13 = 4: off.
template <bool QKV>
static bool launch_tall(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R, int ldr, bf16_t* C,
                        int ldc, int M, int N, int K, hipStream_t s) {
    const int nct = ze_cdiv(N, 64);
    if (ze_gemv_knobs[13] == 4 || K % GEMM_BK != 0 || K / GEMM_BK < 4 || M <= 64 || M > 768) return false;
    if (!QKV && epi != ZE_EPI_NONE && epi != ZE_EPI_RESIDUAL) return false;
    // one workgroup per CU.  Default: six stages in three groups of two K-steps (one barrier per 128 of K; two groups in flight: 16.0 /
    // 11.5 us for qkv / o at 399 rows against 16.4 / 12.1 on the plain four-stage ring, knob 19 = 4; eight stages in four groups of two:
    // level; in two groups of four: 17.8 / 13.6; a plain ring of six or eight stages: 18.9 / 18.2 -- what bounds these launches is what
    // a CU takes in per microsecond, not the depth of its ring)
#define ZE_TALL_G(BM_, ST_, KPB_) launch_ring_variant<(BM_), 64, ST_, (BM_) / 16, 2, false, false, QKV, GEMM_BK, KPB_>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, nullptr, M, N, K, s)
#define ZE_TALL_ST(BM_)                                                        \
    do {                                                                       \
        const int nk_ = K / GEMM_BK;                                           \
        if (ze_gemv_knobs[19] == 4 || nk_ % 2) ZE_TALL_G(BM_, 4, 1);           \
        else ZE_TALL_G(BM_, 6, 2);                                             \
    } while (0)
    if (ze_cdiv(M, 64) * nct <= 256) ZE_TALL_ST(64);
    else if (ze_cdiv(M, 80) * nct <= 256) ZE_TALL_ST(80);
    else if (ze_cdiv(M, 96) * nct <= 256) ZE_TALL_ST(96);
    else if (ze_cdiv(M, 112) * nct <= 256) ZE_TALL_ST(112);
    else if (ze_cdiv(M, 128) * nct <= 256) ZE_TALL_ST(128);
    else return false;
#undef ZE_TALL_ST
#undef ZE_TALL_G
    return true;
}

// The tile follows ze_launch_gemm's choice for these shapes (64 x 64 up to one round of workgroups, 64 x 128 beyond): K in
// sequence on both, so a chain's bits do not depend on the row count -- and equal the unfused pair of launches.
void ze_launch_gemm_qkv_rope(const bf16_t* A, int lda, const bf16_t* Wp, int ldw, const bf16_t* bias_p, const ze_qkv_epi* dev_args,
                             bf16_t* C, int ldc, int M, int N, int K, hipStream_t s) {
    if (M <= 0 || N <= 0) return;
    const long b64 = (long)ze_cdiv(M, 64) * ze_cdiv(N, 64);
    const bf16_t* R = reinterpret_cast<const bf16_t*>(dev_args);
    if (launch_tall<true>(ZE_EPI_QKV_ROPE, A, lda, Wp, ldw, bias_p, R, 0, C, ldc, M, N, K, s)) return;
    // more than 768 rows (round 6: an engine with more than 768 chain slots -- the one-lane A/B of VERDICT r5 #1b): one round of 128 x 128
    // tiles, the tile ze_launch_gemm gives a prefill pass of this size (9 x 20 = 180 workgroups at 1152 rows); K in sequence: same bits
    if (M > 768 && K % GEMM_BK == 0 && K / GEMM_BK >= 4 && (long)ze_cdiv(M, 128) * ze_cdiv(N, 128) <= 256) {
        launch_ring_variant<128, 128, 4, 2, 4, true, false, true>(ZE_EPI_QKV_ROPE, A, lda, Wp, ldw, bias_p, R, 0, C, ldc, nullptr, M, N, K, s);
        return;
    }
    // (two 64-KB workgroups share a CU: up to 512 tiles stay on 64 x 64 -- 16.4 against 20.8 us at 410 rows, 19.9 / 21.0 at 580, 20.9 /
    //  22.7 at 768; knob 13 = 3: 64 x 128 from 257 tiles on, the rule before)
    if (b64 <= (ze_gemv_knobs[13] == 3 ? 256 : 512))
        launch_ring_variant<64, 64, 4, 4, 2, false, false, true>(ZE_EPI_QKV_ROPE, A, lda, Wp, ldw, bias_p, R, 0, C, ldc, nullptr, M, N, K, s);
    else
        launch_ring_variant<64, 128, 4, 2, 4, false, false, true>(ZE_EPI_QKV_ROPE, A, lda, Wp, ldw, bias_p, R, 0, C, ldc, nullptr, M, N, K, s);
}

void ze_launch_gemm_p8(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R, int ldr,
                       bf16_t* C, int ldc, const int* c_rows, int M, int N, int K, hipStream_t s, const ze_gemm_ws& ws) {
    if (M <= 0 || N <= 0) return;
    int ks = ze_prefill_ksplit(K, ws.slab != nullptr);   // (as ze_launch_gemm: the direct entry sums like the policy's)
    const long tiles = (long)ze_cdiv(M, 256) * ze_cdiv(N, 256);
    if (ks > 1 && ((size_t)ks * tiles * 256 * 256 > ws.slab_floats || tiles > ws.ticket_cap)) ks = ze_prefill_split_dropped(M, N, K, ks);
    launch_p8(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, ks, ws);
}

void ze_launch_gemm_stream(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias,
                           const bf16_t* R, int ldr, bf16_t* C, int ldc, int M, int N, int K, const ze_gemm_ws& ws,
                           hipStream_t s) {
    if (M <= 0 || N <= 0) return;
    // Short, narrow weight matrices (qkv, o: K <= 4096, N <= 4096): the skinny kernel, 16 weight rows per workgroup,
    // no split-K (measured, device time per launch at 8 / 64 chains: qkv 9.8 / 13.9 us against 15.0 / 15.6 on the
    // ring, o 8.9 / 12.9 against 14.0 / 14.9).  With 64 rows per workgroup it loses everywhere the ring streams well
    // (gate/up 43.9 against 27.5 us at 64 chains: its 16-row x 64-B fragment loads use the vector memory path at a
    // fraction of what the LDS-DMA's 128-B rows do), so wide or long matrices stay on the ring.  The choice is a
    // function of (N, K, epilogue) alone: batch-composition invariance.
    if (M <= 64 && K % 32 == 0 && K <= 4096 && N <= 4096 && epi != ZE_EPI_SWIGLU && ze_gemv_knobs[5] != 1 &&
        (lda % 8) == 0 && (ldw % 8) == 0) {
        launch_skinny<1>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, 1, s);
        return;
    }
    // (64 x 128 tiles and every forced slice count 1 / 2 / 4 lose to this policy at 64 chains: gate/up 27.5 us here,
    //  27.2-59 there; down 21.3 here, 21-67 there)
    launch_cfg<64, 64>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, nullptr, M, N, K, s, true, ws);
}

#define ze_ring32_from (ze_gemv_knobs[3] > 0 ? ze_gemv_knobs[3] : 512)  // knob 3: first row count (exclusive) of the 320 x 192 tiles
void ze_launch_gemm_wide(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R,
                         int ldr, bf16_t* C, int ldc, int M, int N, int K, const ze_gemm_ws& ws, hipStream_t s) {
    if (M <= 0 || N <= 0) return;
    // long K (the down projection): eight K slices, reduced in slice order, on the split-K ring (k_gemm_wstream with a
    // split K loses to it at every row count: 36.7 against 34.1 us at 256 rows, 23.4 against 17.4 at 64 -- its K loop alone
    // takes 19 us there, but 256-row tiles make the slab reduction eight times the bytes per reducer)
    if (K > 4096) {
        // more than 64 rows: TWO launches -- tiles sized for about one round of workgroups (64 x 128 up to 128 rows, 128 x 128
        // up to 256, 128 x 256 beyond: 64 x 64 outputs per wave = half the LDS fragment reads per MFMA of the 64 x 64 tile's
        // 16 x 32) leave the K slices in the slabs, k_splitk_reduce adds them on the whole chip (the one-launch form has ONE
        // workgroup per tile read all its slabs).  3B down projection, us at 128 / 217 / 256 / 261 / 344 / 384 / 440 / 512 rows:
        // 20.9 / 25.7 / 27.0 / 31.2 / 35.4 / 36.9 / 37.4 / 40.4 against 22.5 / 31.2 / 33.8 / 37.2 / 43.2 / 45.3 / 50.4 / 57.0 in one
        // launch on 64 x 64 tiles (with 128 x 256 tiles at 217 / 256 rows: 32.9 / 34.4 -- 128 workgroups, half the chip).  Same
        // slices, same order of additions: the same bits whatever the form.  knob 15 = 6: one launch at every row count
        if (ze_gemv_knobs[15] != 6) {
            // 257 .. 384 rows: 192 x 128 tiles -- two row tiles instead of three of 128 (no wasted third), 256 workgroups
            // instead of 192 (every CU), 17 % fewer bytes staged per workgroup; knob 15 = 4: the 128 x 256 tiles there too
            if (M > 256 && M <= 384 && ze_gemv_knobs[15] != 4 &&
                launch_splitk_two<192, 128, 3, false, 4, 2>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, ws, s)) return;
            // 513 .. 640 / 641 .. 768 rows: 320 x 128 / 384 x 128 tiles with K-steps of 32 (four stages): two row tiles x 16
            // column tiles x 8 slices = 256 workgroups, one round, where five or six row tiles of 128 x 256 are 320 / 384
            // workgroups -- a second round for a quarter / half of the chip.  knob 15 = 7: off
            // (knob 18, as for gate/up: 0 K-steps of 64 in two stages -- 56.3 -> 53.3 us at 704 rows, level at 576 --, 1 round 4's form)
            if (M > 512 && M <= 640 && ze_gemv_knobs[15] != 7) {
                const int f = ze_gemv_knobs[18];
                if (f == 1 ? launch_splitk_two<320, 128, 4, true, 4, 2, 32>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, ws, s)
                           : launch_splitk_two<320, 128, 2, false, 4, 2>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, ws, s)) return;
            }
            if (M > 640 && M <= 768 && ze_gemv_knobs[15] != 7) {
                const int f = ze_gemv_knobs[18];
                if (f == 1 ? launch_splitk_two<384, 128, 4, true, 4, 2, 32>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, ws, s)
                           : launch_splitk_two<384, 128, 2, false, 4, 2>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, ws, s)) return;
            }
            // (round 5: 128 x 256 without the spread refill 35.7 against 35.1 us at 399 rows, in two stages 43.8)
            if (M > 256 && launch_splitk_two<128, 256, 3, true>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, ws, s)) return;
            if (M > 128 && M <= 256 && launch_splitk_two<128, 128, 4, true>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, ws, s)) return;
            if (M > 64 && M <= 128 && launch_splitk_two<64, 128, 4, false>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, ws, s)) return;
        }
        ze_launch_gemm_stream(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, ws, s);
        return;
    }
    // One pass over K: K in sequence on every tile, so the choice of tile never changes a result and may follow the row
    // count.  Wide matrices (gate/up, lm_head) take k_gemm_wstream where it is faster (3B shape, us per launch at 64 / 128 /
    // 217 / 256 rows: gate/up 26.6 / 28.1 / 37.9 / 38.0 against 24.3 / 29.7 / 39.9 / 41.3 on the ring tiles, lm_head 168 /
    // 181 / 240 / 245 against 192 / 215 / 226 / 228); knob 15 = 9: ring tiles only, 1: k_gemm_wstream wherever it applies.
    // (the wide-store epilogue of k_gemm_wstream moves whole 16-byte row pieces: aligned rows, N a multiple of 32)
    // more than 768 rows (round 6, the one-lane A/B of VERDICT r5 #1b: an engine with up to 1536 chain slots): the decode tiles above are
    // cut for at most 768 rows -- beyond, the row count is a small prefill pass's and its policy serves it best (3B shape at 1152 rows:
    // qkv 21.7 us, o 21.1, gate/up 101.7 on the prefill policy against 38.1 / 22.3 / 129.9 on two blocks of the decode tiles).  K in
    // sequence on every tile: the same bits, so the choice may follow the row count.
    if (M > 768 && epi != ZE_EPI_F32 && ze_gemv_knobs[15] != 9) {
        ze_launch_gemm(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, nullptr, M, N, K, s);
        return;
    }
    const bool ok = K % GEMM_BK == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && (size_t)N * ldw * sizeof(bf16_t) < ((size_t)1 << 31) &&
                    N % 32 == 0 && (ldc % 8) == 0 && ((size_t)A % 16) == 0 && ((size_t)C % 16) == 0;
    const int v = ze_gemv_knobs[15];
    const bool want = v == 1 || (v != 9 && ((epi == ZE_EPI_SWIGLU && M > 96) || (epi == ZE_EPI_F32 && M <= 160)));
    bool done = false;
    if (ok && want && N >= 8192) {
        // more than 256 rows (engines with up to 512 chain slots): blocks of 256 rows, one launch each -- the second pass over
        // the weights is served by the Infinity Cache (90 MB of gate/up against 256 MB), rows stay independent of each other
        const size_t crow = (epi == ZE_EPI_F32) ? 2 * (size_t)ldc : (size_t)ldc;  // (fp32 output rows are twice as long in bf16 units)
        done = true;
        // (gate/up: blocks of up to 512 rows on the 384- and 512-row instances -- a pass costs ~18 us whatever its rows and
        //  ~10 us per 128 rows on top, so 261 chains in one pass instead of a 256- and a 5-row pass; knob 15 = 5: 256 at most)
        // 513 .. 640 rows (where the question stream spends most of its decode steps: two lanes x 768 slots): 320 x 192 tiles on
        // the ring with K-steps of 32 -- four 32-KB stages, two row tiles x 115 column tiles = 230 workgroups in ONE round, 512
        // staged rows per 320 x 192 outputs -- instead of a 512-row and a 128-row weight-streaming pass.  knob 15 = 7: off
        if (epi == ZE_EPI_SWIGLU && M > ze_ring32_from && M <= 640 && v != 7 && v != 4) {
            // (round 5: the two-phase loop -- the waves of a SIMD in opposite phases; knob 18 = 1: the plain loop with the spread refill)
            // Round 5.  The launch is bound by what its workgroups take in through LDS-DMA -- with the MFMAs AND the fragment reads compiled
            // out (tools/probes/ring_ablate.sh) it still takes 57.6 of 63.1 us at 576 rows: 230 workgroups x 2 MB = 460 MB at 8 TB/s, the
            // rate every GEMM of this library stages at -- so neither a two-phase K loop (the waves of a SIMD in opposite phases: -1 us),
            // nor fragments read a step ahead (0), nor deeper rings changed it.  What did: K-steps of 64 in TWO stages (a DMA piece is
            // eight whole 128-byte rows instead of sixteen half lines): 63.7 -> 59.4 us at 576 rows, 77.5 -> 69.1 at 768.  Same MFMAs in
            // the same K order on every form: the same bits.  knob 18: 0 this default, 1 round 4's form (K-steps of 32, four stages,
            // spread refill).
            const int f = ze_gemv_knobs[18];
            if (f == 1) launch_ring_variant<320, 192, 4, 4, 2, true, true, false, 32>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, nullptr, M, N, K, s);
            else launch_ring_variant<320, 192, 2, 4, 2, false, true>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, nullptr, M, N, K, s);
            return;
        }
        // 641 .. 768 rows: 384 x 192 tiles the same way (four 36-KB stages, 96 x 96 outputs per wave)
        if (epi == ZE_EPI_SWIGLU && M > 640 && M <= 768 && v != 7 && v != 4) {
            const int f = ze_gemv_knobs[18];
            if (f == 1) launch_ring_variant<384, 192, 4, 4, 2, true, true, false, 32>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, nullptr, M, N, K, s);
            else launch_ring_variant<384, 192, 2, 4, 2, false, true>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, nullptr, M, N, K, s);
            return;
        }
        // 385 .. 512 rows (round 5): 256 x 192 tiles on K-steps of 64 in two stages WITHOUT the spread refill -- two row tiles x 115
        // column tiles = 230 workgroups staging 1.8 MB each, against 2.4 MB for the 512 x 96 weight-streaming tile (round 4 had tried
        // this tile with the spread refill: 61.9 against 57.4 us at 440 rows).  knob 15 = 8: the weight-streaming tile
        if (epi == ZE_EPI_SWIGLU && M > 384 && M <= 512 && v != 8 && v != 4 && v != 7) {
            launch_ring_variant<256, 192, 2, 4, 2, false, true>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, nullptr, M, N, K, s);
            return;
        }
        const int blk = (epi == ZE_EPI_SWIGLU && K / GEMM_BK == 32 && v != 5) ? 512 : 256;
        for (int r0 = 0; r0 < M && done; r0 += blk) {
            const int mb = std::min(blk, M - r0);
            const bf16_t* Ab = A + (size_t)r0 * lda;
            bf16_t* Cb = C + (size_t)r0 * crow;
            // 257 .. 384 rows: 192 x 192 tiles on the ring (two row tiles x 115 column tiles = 230 workgroups, one round): a
            // workgroup stages (192 + 192) rows per K-step for 192 x 192 outputs -- 96 FLOP per staged byte against 77 for
            // the 384 x 96 weight-streaming tile, which re-reads ALL the activations in every workgroup.  Same K order per
            // output element: the same bits.  knob 15 = 4: the weight-streaming tile there too
            if (mb > 256 && mb <= 384 && epi == ZE_EPI_SWIGLU && v != 4 && r0 == 0 && M <= 384) {
                // (the refill DMAs spread between the rows of MFMAs: 40.6 against 41.6 us at 344 rows)
                // (round 5: without the spread refill 41.4 us, in two stages 46.9, against 40.3 as is at 344 rows)
                launch_ring_variant<192, 192, 3, 4, 2, true, true>(epi, Ab, lda, W, ldw, bias, R, ldr, Cb, ldc, nullptr, mb, N, K, s);
                continue;
            }
            // (tried for 385 .. 512 rows and dropped: 256 x 192 tiles in two stages, 61.9 against 57.4 us at 440 rows)
            // (tried for 385 .. 448 rows and dropped: 224 x 192 tiles on FOUR waves of 112 x 96 outputs -- the only wave grid that
            //  divides 224 rows and keeps the SwiGLU pairs in a wave -- 67.0 against 56.8 us at 440 rows)
            if (mb > 384) launch_wstream_one<512, 96, 2, 6, 32, ZE_EPI_SWIGLU>(Ab, lda, W, ldw, bias, R, ldr, Cb, ldc, mb, N, K, 1, ws, s);
            else if (mb > 256) launch_wstream_one<384, 96, 2, 8, 32, ZE_EPI_SWIGLU>(Ab, lda, W, ldw, bias, R, ldr, Cb, ldc, mb, N, K, 1, ws, s);
            else if (mb > 128) done = launch_wstream<256, 96, 3, 8>(epi, Ab, lda, W, ldw, bias, R, ldr, Cb, ldc, mb, N, K, ws, s);
            else done = launch_wstream<128, 96, 3, 8>(epi, Ab, lda, W, ldw, bias, R, ldr, Cb, ldc, mb, N, K, ws, s);
            if (!done && r0 > 0) done = true, ze_launch_gemm(epi, Ab, lda, W, ldw, bias, R, ldr, Cb, ldc, nullptr, mb, N, K, s);
        }
    }
    if (!done && N <= 4096 && launch_tall<false>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, s)) return;
    if (!done) ze_launch_gemm(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, nullptr, M, N, K, s);
}

void ze_launch_gemm(int epi, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, const bf16_t* R,
                    int ldr, bf16_t* C, int ldc, const int* c_rows, int M, int N, int K, hipStream_t s, const ze_gemm_ws& ws) {
    if (M <= 0 || N <= 0) return;
    // Many tiles: the largest tile (the register-staged kernel hides latency with several workgroups per CU).
    // Otherwise the largest tile whose grid is one round of at least half the CUs, which then runs on the LDS-DMA
    // ring kernel when K % 64 == 0 (launch_cfg); 64 x 64 tiles for what is left.
    const long b128 = (long)ze_cdiv(M, 128) * ze_cdiv(N, 128);
    const long b64x128 = (long)ze_cdiv(M, 64) * ze_cdiv(N, 128);
    const long b64 = (long)ze_cdiv(M, 64) * ze_cdiv(N, 64);
    const bool ringable = K % GEMM_BK == 0;
    // In-between grids (more than one round of 64 x 128 tiles, too few 128 x 128 tiles for the many-tile policy --
    // the LLM qkv projection at M = 802: 260 / 140 tiles): one round of 128 x 128 tiles on the eight-wave ring with the
    // spread refill, 23.6 us against 33.7 register-staged on 64 x 128 (30.4 / 25.7 for the 64 x 128 ring in two rounds /
    // the unspread 128 x 128 ring).
    if (ringable && K / GEMM_BK >= 4 && b128 < 200 && b128 >= 96 && b64x128 > 256 && ze_gemv_knobs[6] == 0 &&
        ze_gemv_knobs[7] != 3) {
        int ks_ = ze_prefill_ksplit(K, ws.slab != nullptr);
        if (ks_ > 1 && ((size_t)ks_ * b128 * 128 * 128 > ws.slab_floats || b128 > ws.ticket_cap)) ks_ = ze_prefill_split_dropped(M, N, K, ks_);
        launch_ring_variant<128, 128, 4, 2, 4, true>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, ks_, ws);
        return;
    }
    if (b128 >= 200)
        launch_cfg<128, 128>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, false, ws);
    else if (b64x128 >= 160 || (ringable && (b64x128 >= 128 || b64 > 256)))
        launch_cfg<64, 128>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, false, ws);
    else
        launch_cfg<64, 64>(epi, A, lda, W, ldw, bias, R, ldr, C, ldc, c_rows, M, N, K, s, false, ws);
}
