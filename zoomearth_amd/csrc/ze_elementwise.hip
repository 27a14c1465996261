// Streaming / elementwise kernels: weight packing + synthetic fill, RMSNorm, row gathers, vision RoPE,
// M-RoPE + KV-cache append, token embedding.  All HBM-bound; bf16 I/O is vectorised 16 B per lane,
// reductions use 64-lane wavefront shuffles.
#include <hip/hip_fp16.h>

#include "ze_kernels.h"
#include "ze_prng.h"

// ------------------------------------------------------------------ weight packing
__device__ __forceinline__ int map_row(int r, int mode, int offset) {
    return mode == 0 ? offset + r : (r >> 4) * 32 + (r & 15) + offset;
}

// src: [rows, cols] of dtype (ZE_F32 / ZE_F16 / ZE_BF16), row-major, starting at logical row row0.
__global__ void __launch_bounds__(256) k_pack_rows(const void* __restrict__ src, int dtype, int row0, int nrows,
                                                   int cols, bf16_t* __restrict__ dst, int ld, int mode, int offset) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)nrows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    bf16_t v;
    if (dtype == ZE_BF16) {
        v = ((const bf16_t*)src)[i];
    } else if (dtype == ZE_F16) {
        v = f32_to_bf16(__half2float(((const __half*)src)[i]));
    } else {
        v = f32_to_bf16(((const float*)src)[i]);
    }
    dst[(size_t)map_row(row0 + r, mode, offset) * ld + c] = v;
}

// Synthetic tensor: value(r, c) = base + normal_ih4(seed, r*cols + c) * scale_const, rounded to bf16.
__global__ void __launch_bounds__(256) k_fill_rows(uint64_t seed, float c_scale, float base, int rows, int cols,
                                                   bf16_t* __restrict__ dst, int ld, int mode, int offset) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    const float v = (c_scale != 0.0f) ? base + ze_normal_ih4(seed, i, c_scale) : base;
    dst[(size_t)map_row(r, mode, offset) * ld + c] = f32_to_bf16(v);
}

void ze_launch_pack_rows(const void* src, int dtype, int row0, int nrows, int cols, bf16_t* dst, int ld, int mode,
                         int offset, hipStream_t s) {
    const size_t n = (size_t)nrows * cols;
    if (n == 0) return;
    k_pack_rows<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(src, dtype, row0, nrows, cols, dst, ld, mode, offset);
}
void ze_launch_fill_rows(uint64_t seed, float c_scale, float base, int rows, int cols, bf16_t* dst, int ld, int mode,
                         int offset, hipStream_t s) {
    const size_t n = (size_t)rows * cols;
    if (n == 0) return;
    k_fill_rows<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(seed, c_scale, base, rows, cols, dst, ld, mode, offset);
}

// ------------------------------------------------------------------ RMSNorm
// y = w * bf16(x * rsqrt(mean(x^2) + eps)); one wave per row, 16-B vector loads.
// (HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:64-79: cast to input dtype BEFORE the weight multiply.)
// frag != 0: y is written MFMA-fragment-major for the batched decode GEMMs (k_gemm_skinny<..., FRAG>):
//   element (row r, col k) -> (((r / 16) * (cols / 32) + k / 32) * 64 + ((k % 32) / 8) * 16 + r % 16) * 8 + k % 8
// i.e. a wave's A fragment of 16 rows x 32 columns is one contiguous 1-KiB block (cols % 32 == 0).
// FP8 activations (ze_set_fp8_activations: BASELINE configs[4], "fp8 weights on fp8 MFMA"): the normalised row -- the
// input of the qkv and gate/up projections -- is quantised to E4M3 with ONE power-of-two scale per row, the rule of the
// weight rows (ze_quant.hip, oracle/fp8.py): k = the smallest integer with max|y| / 2^k <= 448, q = e4m3(y / 2^k).
// q * 2^k is exact in bf16, so a path without an fp8 MFMA kernel (prefill, single-chain GEMV, > 64 chains) computes with
// the SAME values by writing q * 2^k back as bf16 ("fake quantisation"), and the model stays one model.
__device__ __forceinline__ int fp8_row_exponent(float amax) {
    if (!(amax > 0.f)) return 0;
    int e;
    const float m = frexpf(amax / 448.0f, &e);  // amax / 448 = m * 2^e, m in [0.5, 1)
    return (m == 0.5f) ? e - 1 : e;
}
typedef float ew_f32x2 __attribute__((ext_vector_type(2)));
// two bf16 values (one packed word) -> the two E4M3 bytes of value / 2^k (low 16 bits of the result)
__device__ __forceinline__ uint32_t fp8_pack2(uint32_t w, float inv_s) {
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(w) * inv_s, bf16hi(w) * inv_s, 0, false) & 0xffffu;
}
// ... and back to a packed bf16 word holding q * 2^k (exact)
__device__ __forceinline__ uint32_t fp8_fake2(uint32_t w, float inv_s, float s) {
    const ew_f32x2 back = __builtin_amdgcn_cvt_pk_f32_fp8((int)fp8_pack2(w, inv_s), false);
    return pack_bf16x2(back.x * s, back.y * s);
}

// act8 == 3 (prefill on the block-scaled fp8 MFMA, ze_gemm_mx.hip): the row goes out as E4M3 BYTES, row-major in y8 (leading
// dimension cols), with its scale 2^k in yscale[row]; nothing is written to y
__global__ void __launch_bounds__(256) k_rmsnorm(const bf16_t* __restrict__ x, int ldx,
                                                 const bf16_t* __restrict__ w, bf16_t* __restrict__ y, int ldy,
                                                 int rows, int cols, float eps, int frag, int act8,
                                                 uint8_t* __restrict__ y8, float* __restrict__ yscale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const bf16_t* xr = x + (size_t)row * ldx;
    float ss = 0.f;
    const int nv = cols >> 3;  // cols % 8 == 0
    for (int v = lane; v < nv; v += 64) {
        const uint4 q = *reinterpret_cast<const uint4*>(xr + v * 8);
        const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = bf16lo(u[j]), b = bf16hi(u[j]);
            ss += a * a + b * b;
        }
    }
    ss = wave_sum(ss);
    const float inv = rsqrtf(ss / (float)cols + eps);
    bf16_t* yr = y + (size_t)row * ldy;
    auto norm_vec = [&](int v, uint32_t (&o)[4]) {
        const uint4 q = *reinterpret_cast<const uint4*>(xr + v * 8);
        const uint4 g = *reinterpret_cast<const uint4*>(w + v * 8);
        const uint32_t u[4] = {q.x, q.y, q.z, q.w};
        const uint32_t gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = bf16_round(bf16lo(u[j]) * inv) * bf16lo(gw[j]);
            const float b = bf16_round(bf16hi(u[j]) * inv) * bf16hi(gw[j]);
            o[j] = pack_bf16x2(a, b);
        }
    };
    float s8 = 1.f, inv8 = 1.f;
    if (act8) {  // a pass for the row's largest magnitude (of the bf16 outputs), then the quantising pass below
        float amax = 0.f;
        for (int v = lane; v < nv; v += 64) {
            uint32_t o[4];
            norm_vec(v, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(bf16lo(o[j])), fabsf(bf16hi(o[j]))));
        }
        const int k = fp8_row_exponent(wave_max(amax));
        s8 = ldexpf(1.0f, k);
        inv8 = ldexpf(1.0f, -k);
        if (act8 == 3 && lane == 0) yscale[row] = s8;
    }
    for (int v = lane; v < nv; v += 64) {
        uint32_t o[4];
        norm_vec(v, o);
        if (act8 == 3) {
            uint2 b;
            b.x = fp8_pack2(o[0], inv8) | (fp8_pack2(o[1], inv8) << 16);
            b.y = fp8_pack2(o[2], inv8) | (fp8_pack2(o[3], inv8) << 16);
            *reinterpret_cast<uint2*>(y8 + (size_t)row * cols + v * 8) = b;
            continue;
        }
        if (act8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fp8_fake2(o[j], inv8, s8);
        }
        bf16_t* dst = yr + v * 8;
        if (frag)
            dst = y + ((((size_t)(row >> 4) * (cols >> 5) + (v >> 2)) * 64 + (v & 3) * 16 + (row & 15)) << 3);
        *reinterpret_cast<uint4*>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// Few rows (the batched decode step: one row per chain): ONE WORKGROUP per row, the row and the weight in registers
// (NV 16-B vectors per thread), so the launch is a single memory round trip on 64 CUs instead of a two-pass loop on 16
// (7.3 -> about 3.5 us at 64 x 2048: the kernel is pure latency).  Sum of squares: lane partials in vector order, the
// wave sums, then the four waves in order -- a function of the row alone.
// act8: 0 bf16 output; 1 fake-quantised bf16 output (same layout); 2 FP8 fragment-major output in y8 (fragment = 64
// lanes x 8 B: this thread's 8 consecutive columns are exactly one lane's bytes) + the row's scale in yscale[row]
template <int NV>
__global__ void __launch_bounds__(256) k_rmsnorm_row(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ w,
                                                     bf16_t* __restrict__ y, int ldy, int cols, float eps, int frag,
                                                     int act8, uint8_t* __restrict__ y8, float* __restrict__ yscale) {
    const int row = blockIdx.x, t = threadIdx.x;
    const int nv = cols >> 3;
    const bf16_t* xr = x + (size_t)row * ldx;
    uint4 q[NV], g[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {  // branch-free: vectors past the row re-read its last one and are not used
        const int v = min(t + i * 256, nv - 1);
        q[i] = *reinterpret_cast<const uint4*>(xr + v * 8);
        g[i] = *reinterpret_cast<const uint4*>(w + v * 8);
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const uint32_t u[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
        float p = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = bf16lo(u[j]), b = bf16hi(u[j]);
            p += a * a + b * b;
        }
        if (t + i * 256 < nv) ss += p;
    }
    ss = wave_sum(ss);
    __shared__ float part[4];
    if ((t & 63) == 0) part[t >> 6] = ss;
    __syncthreads();
    const float inv = rsqrtf((((part[0] + part[1]) + part[2]) + part[3]) / (float)cols + eps);
    uint32_t o[NV][4];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const uint32_t u[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
        const uint32_t gw[4] = {g[i].x, g[i].y, g[i].z, g[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = bf16_round(bf16lo(u[j]) * inv) * bf16lo(gw[j]);
            const float b = bf16_round(bf16hi(u[j]) * inv) * bf16hi(gw[j]);
            o[i][j] = pack_bf16x2(a, b);
            if (t + i * 256 < nv) amax = fmaxf(amax, fmaxf(fabsf(bf16lo(o[i][j])), fabsf(bf16hi(o[i][j]))));
        }
    }
    float s8 = 1.f, inv8 = 1.f;
    if (act8) {  // workgroup-uniform
        amax = wave_max(amax);
        __shared__ float pmax[4];
        if ((t & 63) == 0) pmax[t >> 6] = amax;
        __syncthreads();
        const int k = fp8_row_exponent(fmaxf(fmaxf(pmax[0], pmax[1]), fmaxf(pmax[2], pmax[3])));
        s8 = ldexpf(1.0f, k);
        inv8 = ldexpf(1.0f, -k);
        if (act8 == 2 && t == 0) yscale[row] = s8;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = t + i * 256;
        if (v >= nv) continue;
        const size_t fslot = ((size_t)(row >> 4) * (cols >> 5) + (v >> 2)) * 64 + (v & 3) * 16 + (row & 15);
        if (act8 == 2) {
            uint2 b;
            b.x = fp8_pack2(o[i][0], inv8) | (fp8_pack2(o[i][1], inv8) << 16);
            b.y = fp8_pack2(o[i][2], inv8) | (fp8_pack2(o[i][3], inv8) << 16);
            *reinterpret_cast<uint2*>(y8 + (fslot << 3)) = b;
            continue;
        }
        if (act8 == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[i][j] = fp8_fake2(o[i][j], inv8, s8);
        }
        bf16_t* dst = y + (size_t)row * ldy + v * 8;
        if (frag) dst = y + (fslot << 3);
        *reinterpret_cast<uint4*>(dst) = make_uint4(o[i][0], o[i][1], o[i][2], o[i][3]);
    }
}

void ze_launch_rmsnorm(const bf16_t* x, int ldx, const bf16_t* w, bf16_t* y, int ldy, int rows, int cols, float eps,
                       hipStream_t s, int frag, int act8, uint8_t* y8, float* yscale) {
    if (rows == 0) return;
    // frag != 0 marks the batched decode step (<= 64 rows): its rows take the one-workgroup-per-row form, so that a
    // chain's normalised row is the same bits whatever the batch; every other caller keeps the wave-per-row kernel
    // frag == 2: a batched decode step beyond the fragment kernels (row-major output, any number of chains): the same
    // one-workgroup-per-row kernel (5.5 -> 3.6 us at 256 rows x 2048)
    if (((frag == 1 && rows <= 64) || frag == 2) && cols <= 8192 && cols % 8 == 0) {
        const int nv = cols >> 3, fm = frag == 1 ? 1 : 0;
        if (nv <= 256) k_rmsnorm_row<1><<<rows, 256, 0, s>>>(x, ldx, w, y, ldy, cols, eps, fm, act8, y8, yscale);
        else if (nv <= 512) k_rmsnorm_row<2><<<rows, 256, 0, s>>>(x, ldx, w, y, ldy, cols, eps, fm, act8, y8, yscale);
        else k_rmsnorm_row<4><<<rows, 256, 0, s>>>(x, ldx, w, y, ldy, cols, eps, fm, act8, y8, yscale);
        return;
    }
    if (frag == 2) frag = 0;
    k_rmsnorm<<<ze_cdiv(rows, 4), 256, 0, s>>>(x, ldx, w, y, ldy, rows, cols, eps, frag, act8 == 3 ? 3 : (act8 ? 1 : 0), y8, yscale);
}

// W [n, k] row-major (leading dimension ldw) -> MFMA-fragment-major copy: fragment (nb, ks) = rows 16 nb .. +15,
// columns 32 ks .. +31, stored as 64 lanes x 16 B, lane = (col % 32) / 8 * 16 + row % 16.  One wave per fragment.
// Rows past n (a last, partial block of 16) repeat row n - 1.
// rope_dim = D > 0 (the qkv projection, n % D == 0): block nb = head h = nb / (D/16), group j = nb % (D/16) holds rows
// h*D + 8j .. +7 and h*D + D/2 + 8j .. +7 -- the rotate_half partners of M-RoPE side by side (ze_gemm_oneshot.hip).
__global__ void __launch_bounds__(256) k_pack_fragments(const bf16_t* __restrict__ W, int ldw, int n, int k,
                                                        bf16_t* __restrict__ Wf, int rope_dim) {
    const size_t frag = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ns = k >> 5;
    if (frag >= (size_t)((n + 15) >> 4) * ns) return;
    const int lane = threadIdx.x & 63, nb = (int)(frag / ns), ks = (int)(frag % ns);
    int row = min(nb * 16 + (lane & 15), n - 1);
    if (rope_dim > 0) {
        const int bph = rope_dim >> 4, h = nb / bph, j = nb % bph, r = lane & 15;
        row = h * rope_dim + (r < 8 ? 8 * j + r : (rope_dim >> 1) + 8 * j + (r - 8));
    }
    const uint4 v = *reinterpret_cast<const uint4*>(W + (size_t)row * ldw + ks * 32 + (lane >> 4) * 8);
    *reinterpret_cast<uint4*>(Wf + (frag * 64 + lane) * 8) = v;
}
__global__ void __launch_bounds__(256) k_pack_fragments8(const uint8_t* __restrict__ W, int ld8, int n, int k,
                                                         uint8_t* __restrict__ Wf, int rope_dim) {
    const size_t frag = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ns = k >> 5;
    if (frag >= (size_t)((n + 15) >> 4) * ns) return;
    const int lane = threadIdx.x & 63, nb = (int)(frag / ns), ks = (int)(frag % ns);
    int row = min(nb * 16 + (lane & 15), n - 1);
    if (rope_dim > 0) {
        const int bph = rope_dim >> 4, h = nb / bph, j = nb % bph, r = lane & 15;
        row = h * rope_dim + (r < 8 ? 8 * j + r : (rope_dim >> 1) + 8 * j + (r - 8));
    }
    const uint2 v = *reinterpret_cast<const uint2*>(W + (size_t)row * ld8 + ks * 32 + (lane >> 4) * 8);
    *reinterpret_cast<uint2*>(Wf + (frag * 64 + lane) * 8) = v;
}
void ze_launch_pack_fragments8(const uint8_t* W8, int ld8, int n, int k, uint8_t* Wf8, hipStream_t s, int rope_dim) {
    const size_t frags = (size_t)((n + 15) >> 4) * (k >> 5);
    if (frags == 0) return;
    k_pack_fragments8<<<(unsigned)((frags + 3) / 4), 256, 0, s>>>(W8, ld8, n, k, Wf8, rope_dim);
}

void ze_launch_pack_fragments(const bf16_t* W, int ldw, int n, int k, bf16_t* Wf, hipStream_t s, int rope_dim) {
    const size_t frags = (size_t)((n + 15) >> 4) * (k >> 5);
    if (frags == 0) return;
    k_pack_fragments<<<(unsigned)((frags + 3) / 4), 256, 0, s>>>(W, ldw, n, k, Wf, rope_dim);
}

// ------------------------------------------------------------------ ViT input: f32 pixel rows -> bf16, gathered, K-padded
// dst[r][0..kp) = bf16(src[perm[r]][0..k)) with zero pad; perm may be null (identity).
__global__ void __launch_bounds__(256) k_gather_cast_rows(const float* __restrict__ src, int k,
                                                          const int* __restrict__ perm, bf16_t* __restrict__ dst,
                                                          int kp, int rows) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * kp) return;
    const int r = (int)(i / kp), c = (int)(i % kp);
    const int sr = perm ? perm[r] : r;
    dst[i] = c < k ? f32_to_bf16(src[(size_t)sr * k + c]) : (bf16_t)0;
}
void ze_launch_gather_cast_rows(const float* src, int k, const int* perm, bf16_t* dst, int kp, int rows,
                                hipStream_t s) {
    const size_t n = (size_t)rows * kp;
    if (n == 0) return;
    k_gather_cast_rows<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(src, k, perm, dst, kp, rows);
}

// ------------------------------------------------------------------ vision RoPE (fp32 math, in place on q and k of the qkv buffer)
// qkv: [N, 3, heads, D] bf16; cs: [N, D/2] float2? -> separate cos/sin f32 tables [N, D/2] where table column j
// serves dims j and j + D/2 (emb = cat(rot, rot)); rotate_half pairs (j, j + D/2).
// (HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:153-171)
__global__ void __launch_bounds__(256) k_vision_rope(bf16_t* __restrict__ qkv, const float* __restrict__ cosT,
                                                     const float* __restrict__ sinT, int n, int heads, int D) {
    const int half = D >> 1;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)n * 2 * heads * half;
    if (i >= total) return;
    const int j = (int)(i % half);
    const int hh = (int)((i / half) % heads);
    const int which = (int)((i / ((size_t)half * heads)) % 2);  // 0 q, 1 k
    const int row = (int)(i / ((size_t)half * heads * 2));
    bf16_t* p = qkv + (((size_t)row * 3 + which) * heads + hh) * D;
    const float c = cosT[(size_t)row * half + j], s = sinT[(size_t)row * half + j];
    const float x1 = bf16_to_f32(p[j]), x2 = bf16_to_f32(p[j + half]);
    // q*cos + rotate_half(q)*sin, rotate_half = cat(-x2, x1); fp32 products and sum as the CPU reference forms them (no
    // contraction into an fma), then one rounding
    p[j] = f32_to_bf16(__fadd_rn(__fmul_rn(x1, c), __fmul_rn(-x2, s)));
    p[j + half] = f32_to_bf16(__fadd_rn(__fmul_rn(x2, c), __fmul_rn(x1, s)));
}
// the same, eight rotate_half pairs per thread with 16-byte accesses (half % 8 == 0: the 80-wide heads of the ViT give five
// vectors per half): the scalar form ran 148 us per ViT block on a 24-image call (2-byte accesses: 2.1 TB/s)
__global__ void __launch_bounds__(256) k_vision_rope_vec(bf16_t* __restrict__ qkv, const float* __restrict__ cosT,
                                                         const float* __restrict__ sinT, int n, int heads, int D) {
    const int half = D >> 1, hv = half >> 3;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)n * 2 * heads * hv;
    if (i >= total) return;
    const int j = (int)(i % hv) * 8;
    const int hh = (int)((i / hv) % heads);
    const int which = (int)((i / ((size_t)hv * heads)) % 2);  // 0 q, 1 k
    const int row = (int)(i / ((size_t)hv * heads * 2));
    bf16_t* p = qkv + (((size_t)row * 3 + which) * heads + hh) * D;
    const uint4 a = *reinterpret_cast<const uint4*>(p + j), b = *reinterpret_cast<const uint4*>(p + j + half);
    const float4 c0 = *reinterpret_cast<const float4*>(cosT + (size_t)row * half + j);
    const float4 c1 = *reinterpret_cast<const float4*>(cosT + (size_t)row * half + j + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(sinT + (size_t)row * half + j);
    const float4 s1 = *reinterpret_cast<const float4*>(sinT + (size_t)row * half + j + 4);
    const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const uint32_t* pa = reinterpret_cast<const uint32_t*>(&a);
    const uint32_t* pb = reinterpret_cast<const uint32_t*>(&b);
    uint32_t r1[4], r2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x1l = __uint_as_float(pa[q] << 16), x1h = __uint_as_float(pa[q] & 0xffff0000u);
        const float x2l = __uint_as_float(pb[q] << 16), x2h = __uint_as_float(pb[q] & 0xffff0000u);
        const uint32_t l1 = f32_to_bf16(__fadd_rn(__fmul_rn(x1l, cs[2 * q]), __fmul_rn(-x2l, sn[2 * q])));
        const uint32_t h1 = f32_to_bf16(__fadd_rn(__fmul_rn(x1h, cs[2 * q + 1]), __fmul_rn(-x2h, sn[2 * q + 1])));
        const uint32_t l2 = f32_to_bf16(__fadd_rn(__fmul_rn(x2l, cs[2 * q]), __fmul_rn(x1l, sn[2 * q])));
        const uint32_t h2 = f32_to_bf16(__fadd_rn(__fmul_rn(x2h, cs[2 * q + 1]), __fmul_rn(x1h, sn[2 * q + 1])));
        r1[q] = l1 | (h1 << 16);
        r2[q] = l2 | (h2 << 16);
    }
    *reinterpret_cast<uint4*>(p + j) = make_uint4(r1[0], r1[1], r1[2], r1[3]);
    *reinterpret_cast<uint4*>(p + j + half) = make_uint4(r2[0], r2[1], r2[2], r2[3]);
}
void ze_launch_vision_rope(bf16_t* qkv, const float* cosT, const float* sinT, int n, int heads, int D,
                           hipStream_t s) {
    const size_t total = (size_t)n * 2 * heads * (D / 2);
    if (total == 0) return;
    if ((D / 2) % 8 == 0 && ((size_t)qkv % 16) == 0) {
        const size_t tv = total / 8;
        k_vision_rope_vec<<<(unsigned)((tv + 255) / 256), 256, 0, s>>>(qkv, cosT, sinT, n, heads, D);
        return;
    }
    k_vision_rope<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(qkv, cosT, sinT, n, heads, D);
}

int ze_mrope_vec_ok = 1;  // (process-wide, conservative: any engine with odd M-RoPE sections switches the 16-byte form off)
#define mrope_vec_ok (ze_mrope_vec_ok != 0)

// (rope8: ze_kernels.h -- the prefill flash kernel applies it to its Q fragments)

// ------------------------------------------------------------------ text M-RoPE + KV append (prefill)
// qkv: [T, (heads + 2*kv_heads) * D] bf16 (q | k | v).  cos/sin tables: bf16 [max_pos, D/2] (already rounded to the
// model dtype like HF's cos.to(x.dtype)).  pos3: int32 [3, T].  axis_of[j] in {0,1,2} for j < D/2 picks t/h/w.
// q is rotated in place; k (rotated) and v are written to the cache rows [past + t] of their kv head.
// bf16 arithmetic with HF's three roundings: bf16(bf16(x*cos) + bf16(rot*sin)).
// (HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:557-599, 654-668)
__global__ void __launch_bounds__(256) k_mrope_kv(bf16_t* __restrict__ qkv, int T, int heads, int kv_heads, int D,
                                                  const bf16_t* __restrict__ cosT, const bf16_t* __restrict__ sinT,
                                                  const int* __restrict__ pos3, const int* __restrict__ axis_of,
                                                  bf16_t* __restrict__ kcache, bf16_t* __restrict__ vcache,
                                                  int max_ctx, int past, const int* __restrict__ row_aux,
                                                  size_t cache_seq_stride) {
    const int half = D >> 1;
    const int nh = heads + 2 * kv_heads;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)T * nh * half;
    if (i >= total) return;
    const int j = (int)(i % half);
    const int hh = (int)((i / half) % nh);
    const int t = (int)(i / ((size_t)half * nh));
    // cache row of this token: past + t of the one chain, or (chain slot, cache position) per row (batched prefill)
    int cpos = past + t;
    if (row_aux) {
        kcache += (size_t)row_aux[2 * t] * cache_seq_stride;
        vcache += (size_t)row_aux[2 * t] * cache_seq_stride;
        cpos = row_aux[2 * t + 1];
    }
    bf16_t* p = qkv + (size_t)t * nh * D + (size_t)hh * D;
    if (hh >= heads + kv_heads) {  // v: plain copy into the cache
        const int kvh = hh - heads - kv_heads;
        bf16_t* d = vcache + ((size_t)kvh * max_ctx + cpos) * D;
        d[j] = p[j];
        d[j + half] = p[j + half];
        return;
    }
    const int pos = pos3[axis_of[j] * T + t];
    const float c = bf16_to_f32(cosT[(size_t)pos * half + j]);
    const float s = bf16_to_f32(sinT[(size_t)pos * half + j]);
    const float x1 = bf16_to_f32(p[j]), x2 = bf16_to_f32(p[j + half]);
    const bf16_t o1 = f32_to_bf16(bf16_round(x1 * c) + bf16_round(-x2 * s));
    const bf16_t o2 = f32_to_bf16(bf16_round(x2 * c) + bf16_round(x1 * s));
    if (hh < heads) {
        p[j] = o1;
        p[j + half] = o2;
    } else {
        const int kvh = hh - heads;
        bf16_t* d = kcache + ((size_t)kvh * max_ctx + cpos) * D;
        d[j] = o1;
        d[j + half] = o2;
    }
}
// The same, eight pairs per thread with 16-byte accesses (D % 16 == 0, the three M-RoPE sections multiples of 8 pairs --
// Qwen2.5-VL: 16 / 24 / 24 -- so the eight pairs of a thread share their position axis; rows 16-byte aligned).
__global__ void __launch_bounds__(256) k_mrope_kv_vec(bf16_t* __restrict__ qkv, int T, int heads, int kv_heads, int D,
                                                      const bf16_t* __restrict__ cosT, const bf16_t* __restrict__ sinT,
                                                      const int* __restrict__ pos3, const int* __restrict__ axis_of,
                                                      bf16_t* __restrict__ kcache, bf16_t* __restrict__ vcache,
                                                      int max_ctx, int past, const int* __restrict__ row_aux,
                                                      size_t cache_seq_stride, int q_skip) {
    const int half = D >> 1, hv = half >> 3;
    const int nh = heads + 2 * kv_heads;
    const int nw = q_skip ? 2 * kv_heads : nh;   // q_skip: the K and V heads only (the flash kernel ropes Q as it loads it)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * nw * hv) return;
    const int j = (int)(i % hv) * 8;
    const int hh = (int)((i / hv) % nw) + (q_skip ? heads : 0);
    const int t = (int)(i / ((size_t)hv * nw));
    int cpos = past + t;
    if (row_aux) {
        kcache += (size_t)row_aux[2 * t] * cache_seq_stride;
        vcache += (size_t)row_aux[2 * t] * cache_seq_stride;
        cpos = row_aux[2 * t + 1];
    }
    bf16_t* p = qkv + (size_t)t * nh * D + (size_t)hh * D;
    const uint4 a = *reinterpret_cast<const uint4*>(p + j), b = *reinterpret_cast<const uint4*>(p + j + half);
    if (hh >= heads + kv_heads) {  // v: plain copy into the cache
        bf16_t* d = vcache + ((size_t)(hh - heads - kv_heads) * max_ctx + cpos) * D;
        *reinterpret_cast<uint4*>(d + j) = a;
        *reinterpret_cast<uint4*>(d + j + half) = b;
        return;
    }
    const int pos = pos3[axis_of[j] * T + t];
    const uint4 c4 = *reinterpret_cast<const uint4*>(cosT + (size_t)pos * half + j);
    const uint4 s4 = *reinterpret_cast<const uint4*>(sinT + (size_t)pos * half + j);
    uint4 o1, o2;
    rope8(a, b, c4, s4, o1, o2);
    bf16_t* d = hh < heads ? p : kcache + ((size_t)(hh - heads) * max_ctx + cpos) * D;
    *reinterpret_cast<uint4*>(d + j) = o1;
    *reinterpret_cast<uint4*>(d + j + half) = o2;
}
void ze_launch_mrope_kv(bf16_t* qkv, int T, int heads, int kv_heads, int D, const bf16_t* cosT, const bf16_t* sinT,
                        const int* pos3, const int* axis_of, bf16_t* kcache, bf16_t* vcache, int max_ctx, int past,
                        const int* row_aux, size_t cache_seq_stride, hipStream_t s, int q_skip) {
    const size_t total = (size_t)T * (heads + 2 * kv_heads) * (D / 2);
    if (total == 0) return;
    if (D % 16 == 0 && mrope_vec_ok) {
        const size_t tv = (size_t)T * (q_skip ? 2 * kv_heads : heads + 2 * kv_heads) * (D / 2) / 8;
        k_mrope_kv_vec<<<(unsigned)((tv + 255) / 256), 256, 0, s>>>(qkv, T, heads, kv_heads, D, cosT, sinT, pos3, axis_of, kcache,
                                                                  vcache, max_ctx, past, row_aux, cache_seq_stride, q_skip);
        return;
    }
    k_mrope_kv<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(qkv, T, heads, kv_heads, D, cosT, sinT, pos3,
                                                               axis_of, kcache, vcache, max_ctx, past, row_aux,
                                                               cache_seq_stride);
}

// ------------------------------------------------------------------ embedding gather + image scatter (K13)
// src[t] >= 0: embed_tokens row; src[t] < 0: image_embeds row (-1 - src[t]).
__global__ void __launch_bounds__(256) k_embed_rows(const int* __restrict__ src, const bf16_t* __restrict__ embed,
                                                    const bf16_t* __restrict__ image_embeds,
                                                    bf16_t* __restrict__ out, int T, int hidden) {
    const int nv = hidden >> 3;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * nv) return;
    const int t = (int)(i / nv), v = (int)(i % nv);
    const int sidx = src[t];
    const bf16_t* row = sidx >= 0 ? embed + (size_t)sidx * hidden : image_embeds + (size_t)(-1 - sidx) * hidden;
    *reinterpret_cast<uint4*>(out + (size_t)t * hidden + v * 8) = *reinterpret_cast<const uint4*>(row + v * 8);
}
void ze_launch_embed_rows(const int* src, const bf16_t* embed, const bf16_t* image_embeds, bf16_t* out, int T,
                          int hidden, hipStream_t s) {
    const size_t n = (size_t)T * (hidden / 8);
    if (n == 0) return;
    k_embed_rows<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(src, embed, image_embeds, out, T, hidden);
}

// rows scatter/gather of bf16 rows: dst[dst_idx[r]] = src[r] (dst_idx null = identity)
__global__ void __launch_bounds__(256) k_scatter_rows(const bf16_t* __restrict__ src, int lds_,
                                                      const int* __restrict__ dst_idx, bf16_t* __restrict__ dst,
                                                      int ldd, int rows, int cols) {
    const int nv = cols >> 3;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * nv) return;
    const int r = (int)(i / nv), v = (int)(i % nv);
    const int d = dst_idx ? dst_idx[r] : r;
    *reinterpret_cast<uint4*>(dst + (size_t)d * ldd + v * 8) =
        *reinterpret_cast<const uint4*>(src + (size_t)r * lds_ + v * 8);
}
void ze_launch_scatter_rows(const bf16_t* src, int lds_, const int* dst_idx, bf16_t* dst, int ldd, int rows,
                            int cols, hipStream_t s) {
    const size_t n = (size_t)rows * (cols / 8);
    if (n == 0) return;
    k_scatter_rows<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(src, lds_, dst_idx, dst, ldd, rows, cols);
}

// ------------------------------------------------------------------ batched decode: one token per chain
__global__ void __launch_bounds__(256) k_embed_tokens_batch(const ze_seq_dev* __restrict__ st,
                                                            const int* __restrict__ seq_ids, int n,
                                                            const bf16_t* __restrict__ embed, bf16_t* __restrict__ out,
                                                            int hidden) {
    const int nv = hidden >> 3;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * nv) return;
    const int b = (int)(i / nv), v = (int)(i % nv);
    const int tok = st[seq_ids[b]].token;
    *reinterpret_cast<uint4*>(out + (size_t)b * hidden + v * 8) =
        *reinterpret_cast<const uint4*>(embed + (size_t)tok * hidden + v * 8);
}
struct ze_int_pack {
    int v[128];
};
__global__ void k_set_ints(int* __restrict__ dst, ze_int_pack p, int n) {
    const int i = threadIdx.x;
    if (i < n) dst[i] = p.v[i];
}
void ze_launch_set_ints(int* dst, const int* host_vals, int n, hipStream_t s) {
    for (int o = 0; o < n; o += 128) {
        ze_int_pack p;
        const int m = n - o < 128 ? n - o : 128;
        for (int i = 0; i < 128; ++i) p.v[i] = i < m ? host_vals[o + i] : 0;
        hipLaunchKernelGGL(k_set_ints, dim3(1), dim3(128), 0, s, dst + o, p, m);
    }
}

__global__ void k_scatter_ints(int* __restrict__ dst, ze_int_pack p, int n) {
    const int i = threadIdx.x;
    if (i < n) dst[p.v[i]] = p.v[64 + i];
}
void ze_launch_scatter_ints(int* dst, const int* host_idx, const int* host_vals, int n, hipStream_t s) {
    for (int o = 0; o < n; o += 64) {
        ze_int_pack p;
        const int m = n - o < 64 ? n - o : 64;
        for (int i = 0; i < 64; ++i) {
            p.v[i] = i < m ? host_idx[o + i] : 0;
            p.v[64 + i] = i < m ? host_vals[o + i] : 0;
        }
        hipLaunchKernelGGL(k_scatter_ints, dim3(1), dim3(64), 0, s, dst, p, m);
    }
}

void ze_launch_embed_tokens_batch(const ze_seq_dev* st, const int* seq_ids, int n, const bf16_t* embed, bf16_t* out,
                                  int hidden, hipStream_t s) {
    const size_t tot = (size_t)n * (hidden / 8);
    if (tot == 0) return;
    k_embed_tokens_batch<<<(unsigned)((tot + 255) / 256), 256, 0, s>>>(st, seq_ids, n, embed, out, hidden);
}

// qkv: [n, (heads + 2 kv_heads) * D]; row b belongs to chain seq_ids[b] at position ctx + rope_delta (all three
// M-RoPE axes equal during decode).  Same bf16 arithmetic as k_mrope_kv.
__global__ void __launch_bounds__(256) k_rope_kv_batch(bf16_t* __restrict__ qkv, int n, int heads, int kv_heads, int D,
                                                       const bf16_t* __restrict__ cosT, const bf16_t* __restrict__ sinT,
                                                       const ze_seq_dev* __restrict__ st,
                                                       const int* __restrict__ seq_ids, bf16_t* __restrict__ kcache,
                                                       bf16_t* __restrict__ vcache, size_t cache_seq_stride,
                                                       int max_ctx) {
    const int half = D >> 1;
    const int nh = heads + 2 * kv_heads;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * nh * half) return;
    const int j = (int)(i % half);
    const int hh = (int)((i / half) % nh);
    const int b = (int)(i / ((size_t)half * nh));
    const int seq = seq_ids[b];
    const int ctx = st[seq].ctx, pos = ctx + st[seq].rope_delta;
    bf16_t* p = qkv + (size_t)b * nh * D + (size_t)hh * D;
    if (hh >= heads + kv_heads) {
        bf16_t* d = vcache + seq * cache_seq_stride + ((size_t)(hh - heads - kv_heads) * max_ctx + ctx) * D;
        d[j] = p[j];
        d[j + half] = p[j + half];
        return;
    }
    const float c = bf16_to_f32(cosT[(size_t)pos * half + j]);
    const float sn = bf16_to_f32(sinT[(size_t)pos * half + j]);
    const float x1 = bf16_to_f32(p[j]), x2 = bf16_to_f32(p[j + half]);
    const bf16_t o1 = f32_to_bf16(bf16_round(x1 * c) + bf16_round(-x2 * sn));
    const bf16_t o2 = f32_to_bf16(bf16_round(x2 * c) + bf16_round(x1 * sn));
    if (hh < heads) {
        p[j] = o1;
        p[j + half] = o2;
    } else {
        bf16_t* d = kcache + seq * cache_seq_stride + ((size_t)(hh - heads) * max_ctx + ctx) * D;
        d[j] = o1;
        d[j + half] = o2;
    }
}
__global__ void __launch_bounds__(256) k_rope_kv_batch_vec(bf16_t* __restrict__ qkv, int n, int heads, int kv_heads, int D,
                                                           const bf16_t* __restrict__ cosT, const bf16_t* __restrict__ sinT,
                                                           const ze_seq_dev* __restrict__ st, const int* __restrict__ seq_ids,
                                                           bf16_t* __restrict__ kcache, bf16_t* __restrict__ vcache,
                                                           size_t cache_seq_stride, int max_ctx) {
    const int half = D >> 1, hv = half >> 3;
    const int nh = heads + 2 * kv_heads;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * nh * hv) return;
    const int j = (int)(i % hv) * 8;
    const int hh = (int)((i / hv) % nh);
    const int b_ = (int)(i / ((size_t)hv * nh));
    const int seq = seq_ids[b_];
    const int ctx = st[seq].ctx, pos = ctx + st[seq].rope_delta;
    bf16_t* p = qkv + (size_t)b_ * nh * D + (size_t)hh * D;
    const uint4 a = *reinterpret_cast<const uint4*>(p + j), b = *reinterpret_cast<const uint4*>(p + j + half);
    if (hh >= heads + kv_heads) {
        bf16_t* d = vcache + seq * cache_seq_stride + ((size_t)(hh - heads - kv_heads) * max_ctx + ctx) * D;
        *reinterpret_cast<uint4*>(d + j) = a;
        *reinterpret_cast<uint4*>(d + j + half) = b;
        return;
    }
    const uint4 c4 = *reinterpret_cast<const uint4*>(cosT + (size_t)pos * half + j);
    const uint4 s4 = *reinterpret_cast<const uint4*>(sinT + (size_t)pos * half + j);
    uint4 o1, o2;
    rope8(a, b, c4, s4, o1, o2);
    bf16_t* d = hh < heads ? p : kcache + seq * cache_seq_stride + ((size_t)(hh - heads) * max_ctx + ctx) * D;
    *reinterpret_cast<uint4*>(d + j) = o1;
    *reinterpret_cast<uint4*>(d + j + half) = o2;
}
void ze_launch_rope_kv_batch(bf16_t* qkv, int n, int heads, int kv_heads, int D, const bf16_t* cosT, const bf16_t* sinT,
                             const ze_seq_dev* st, const int* seq_ids, bf16_t* kcache, bf16_t* vcache,
                             size_t cache_seq_stride, int max_ctx, hipStream_t s) {
    const size_t tot = (size_t)n * (heads + 2 * kv_heads) * (D / 2);
    if (tot == 0) return;
    if (D % 16 == 0) {  // (eight pairs per thread, 16-byte accesses; the arithmetic of the scalar kernel)
        const size_t tv = tot / 8;
        k_rope_kv_batch_vec<<<(unsigned)((tv + 255) / 256), 256, 0, s>>>(qkv, n, heads, kv_heads, D, cosT, sinT, st, seq_ids, kcache,
                                                                       vcache, cache_seq_stride, max_ctx);
        return;
    }
    k_rope_kv_batch<<<(unsigned)((tot + 255) / 256), 256, 0, s>>>(qkv, n, heads, kv_heads, D, cosT, sinT, st, seq_ids,
                                                                  kcache, vcache, cache_seq_stride, max_ctx);
}


// ------------------------------------------------------------------ KV prefix copy (shared prompt prefixes)
// The first n cached tokens of chain `src` -> chain `dst`, every layer and kv head, K and V: blockIdx.y = (layer, kv
// head, K|V) picks one contiguous n x 256-byte run of the cache, blockIdx.x strides over its 16-byte pieces.
__global__ void __launch_bounds__(256) k_kv_copy_prefix(bf16_t* __restrict__ kcache, bf16_t* __restrict__ vcache,
                                                        size_t layer_stride, size_t seq_stride, size_t head_stride,
                                                        int kv_heads, int src, int dst, int n_vec) {
    const int which = blockIdx.y & 1, kvh = (blockIdx.y >> 1) % kv_heads, layer = (blockIdx.y >> 1) / kv_heads;
    bf16_t* base = (which ? vcache : kcache) + (size_t)layer * layer_stride + (size_t)kvh * head_stride;
    const uint4* s = reinterpret_cast<const uint4*>(base + (size_t)src * seq_stride);
    uint4* d = reinterpret_cast<uint4*>(base + (size_t)dst * seq_stride);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_vec; i += gridDim.x * 256) d[i] = s[i];
}
void ze_launch_kv_copy_prefix(bf16_t* kcache, bf16_t* vcache, size_t layer_stride, size_t seq_stride, size_t head_stride,
                              int layers, int kv_heads, int D, int src, int dst, int n_tokens, hipStream_t s) {
    if (n_tokens <= 0) return;
    const int n_vec = n_tokens * D * 2 / 16;
    k_kv_copy_prefix<<<dim3(std::max(1, std::min(8, (n_vec + 255) / 256)), layers * kv_heads * 2), 256, 0, s>>>(
        kcache, vcache, layer_stride, seq_stride, head_stride, kv_heads, src, dst, n_vec);
}

// ------------------------------------------------------------------ the numeric helpers of every epilogue, alone (parity ledger, round 6)
// out[i] = f32_to_bf16(x[i]) | f32_to_bf16(bf16_round(silu_f(x[i])) * y[i]) << 16  -- the SwiGLU epilogue's arithmetic on one pair --
// and out2[i] = pack_bf16x2(x[i], y[i]); tests/test_gpu_ops_kernels.py holds both to float64 over a dense grid.
__global__ void __launch_bounds__(256) k_numeric_helpers(const float* __restrict__ x, const float* __restrict__ y, uint32_t* __restrict__ out,
                                                         uint32_t* __restrict__ out2, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = x[i], b = y[i];
    out[i] = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(bf16_round(silu_f(a)) * b) << 16);
    out2[i] = pack_bf16x2(a, b);
}
void ze_launch_numeric_helpers(const float* x, const float* y, uint32_t* out, uint32_t* out2, int n, hipStream_t s) {
    if (n > 0) k_numeric_helpers<<<(n + 255) / 256, 256, 0, s>>>(x, y, out, out2, n);
}
