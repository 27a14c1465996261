// Image front-end kernels: Pillow-exact bicubic crop+resize on u8 RGB and
// LUT normalise + patchify.  Pure integer / LUT work, HBM-bound: one coalesced read of the tile
// (75 MB for 5000x5000) dominates; coefficient tables are generated on the host in double
// precision (identical IEEE arithmetic to Pillow's precompute_coeffs) and uploaded as int32.
//
// Replaces: PIL Image.crop/resize(BICUBIC) called from /root/reference/src/eval/infer.py:41-85 and the
// HF image processor (HF:models/qwen2_vl/image_processing_pil_qwen2_vl.py:126-187).
#include "ze_kernels.h"

#define PRECISION_BITS 22

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Horizontal pass over the crop box: dst[y][xx][c] for y in [0, box_h), xx in [0, out_w).
// Source pixel (bx0 + xmin + j, by0 + y) is zero outside the image (PIL crop semantics).
// Block = 64 output columns x 4 rows; the needed source span of a row is staged in LDS with
// coalesced byte loads, then each thread accumulates its taps for 3 channels.
#define HZ_COLS 64
#define HZ_ROWS 4
#define HZ_MAXSPAN 2816  // pixels; (64 cols * scale + 2*support) for scale <= ~40

__global__ void __launch_bounds__(256) k_resize_h(const uint8_t* __restrict__ src, int src_h, int src_w, int bx0,
                                                  int by0, int box_h, uint8_t* __restrict__ dst, int out_w,
                                                  const int* __restrict__ xmin, const int* __restrict__ xcnt,
                                                  const int* __restrict__ kk, int ksize) {
    __shared__ uint8_t row_lds[HZ_ROWS][HZ_MAXSPAN * 3];
    const int col0 = blockIdx.x * HZ_COLS;
    const int row0 = blockIdx.y * HZ_ROWS;
    const int ncol = min(HZ_COLS, out_w - col0);
    const int span_lo = xmin[col0];
    const int last = col0 + ncol - 1;
    const int span_hi = xmin[last] + xcnt[last];  // exclusive, in box coordinates
    const int span = span_hi - span_lo;
    // stage: HZ_ROWS rows x span pixels x 3 bytes
    for (int r = 0; r < HZ_ROWS; ++r) {
        const int y = row0 + r;
        if (y >= box_h) break;
        const int sy = by0 + y;
        const bool row_in = (sy >= 0 && sy < src_h);
        const uint8_t* srow = src + (size_t)sy * src_w * 3;
        for (int i = threadIdx.x; i < span * 3; i += 256) {
            const int px = bx0 + span_lo + i / 3;
            uint8_t v = 0;
            if (row_in && px >= 0 && px < src_w) v = srow[(size_t)(bx0 + span_lo) * 3 + i];
            row_lds[r][i] = v;
        }
    }
    __syncthreads();
    const int c = threadIdx.x & (HZ_COLS - 1);
    const int r = threadIdx.x >> 6;
    const int y = row0 + r;
    if (c >= ncol || y >= box_h) return;
    const int xx = col0 + c;
    const int lo = xmin[xx] - span_lo;
    const int n = xcnt[xx];
    const int* k = kk + (size_t)xx * ksize;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    const uint8_t* p = &row_lds[r][lo * 3];
    for (int j = 0; j < n; ++j) {
        const int w = k[j];
        s0 += w * p[3 * j + 0];
        s1 += w * p[3 * j + 1];
        s2 += w * p[3 * j + 2];
    }
    uint8_t* d = dst + ((size_t)y * out_w + xx) * 3;
    d[0] = clip8(s0);
    d[1] = clip8(s1);
    d[2] = clip8(s2);
}

// The same pass for a box that lies INSIDE the image (every view / crop of the zoom chain except crops that leave the
// tile): the 75-MB tile is the one large HBM read of the front-end, so the span is staged with 16-BYTE loads from the
// 16-B-aligned address at or below its first byte (the misalignment is carried as an offset into the LDS row), 8 rows per
// workgroup, and a thread runs its taps for two rows from one read of the coefficient table.  Arithmetic and rounding are
// those of k_resize_h (bit-exact with Pillow).  5000^2 -> 512^2: 224 us (0.33 TB/s) with the byte-staged kernel, 68-72 us
// (1.1 TB/s) here; what is left is the unpacking of interleaved RGB bytes on the vector ALUs (~60 instructions per four
// taps of two rows), not memory: dword taps, 24-bit multiplies, pipelined staging and coefficients in LDS each moved it by
// a few us only.
#define HV_ROWS 8
__global__ void __launch_bounds__(256) k_resize_h_vec(const uint8_t* __restrict__ src, int src_h, int src_w, int bx0,
                                                      int by0, int box_h, uint8_t* __restrict__ dst, int out_w,
                                                      const int* __restrict__ xmin, const int* __restrict__ xcnt,
                                                      const int* __restrict__ kk, int ksize, int row_stride) {
    extern __shared__ __attribute__((aligned(16))) uint8_t hv_lds[];  // HV_ROWS rows of row_stride bytes
    const int col0 = blockIdx.x * HZ_COLS;
    const int row0 = blockIdx.y * HV_ROWS;
    const int ncol = min(HZ_COLS, out_w - col0);
    const int span_lo = xmin[col0];
    const int last = col0 + ncol - 1;
    const int nb = (xmin[last] + xcnt[last] - span_lo) * 3;  // bytes of the span
    const size_t row_bytes = (size_t)src_w * 3;
    const uint8_t* end = src + (size_t)src_h * row_bytes;
    // staging: the 16-byte pieces of all 8 rows as ONE flat index space, four loads in flight per thread before the
    // first LDS store (row after row with a load -> store dependency each was eight exposed round trips per workgroup)
    int* offs_w = reinterpret_cast<int*>(hv_lds + (size_t)HV_ROWS * row_stride);
    // the 64 columns' coefficient rows (contiguous in the table) go to LDS too: the tap loop then has no global load in it
    int* kl = offs_w + HV_ROWS;
    {
        const int* ksrc = kk + (size_t)col0 * ksize;
        for (int i = threadIdx.x; i < ncol * ksize; i += 256) kl[i] = ksrc[i];
    }
    const int nrows = min(HV_ROWS, box_h - row0);
    const int nvec_max = (15 + nb + 15) >> 4;
    const int total = nrows * nvec_max;
    for (int f0 = 0; f0 < total; f0 += 4 * 256) {
        uint4 val[4];
        int dsto[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = f0 + u * 256 + (int)threadIdx.x;
            const int r = min(f / nvec_max, nrows - 1), v = f % nvec_max;
            const uint8_t* a = src + (size_t)(by0 + row0 + r) * row_bytes + (size_t)(bx0 + span_lo) * 3;
            const int off = (int)((uintptr_t)a & 15);
            const uint8_t* p16 = a - off + (size_t)v * 16;
            const bool need = f < total && v * 16 < off + nb;
            const bool safe = p16 >= src && p16 + 16 <= end;
            dsto[u] = need ? (r * row_stride + v * 16) | (safe ? 0 : (1 << 30)) : -1;
            val[u] = (need && safe) ? *reinterpret_cast<const uint4*>(p16) : make_uint4(0, 0, 0, 0);
            if (v == 0 && f < total) offs_w[r] = off;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (dsto[u] < 0) continue;
            if (dsto[u] & (1 << 30)) {  // first / last 16 bytes of the allocation: byte by byte, no over-read
                const int d = dsto[u] & ~(1 << 30), r = d / row_stride, v = (d % row_stride) >> 4;
                const uint8_t* a = src + (size_t)(by0 + row0 + r) * row_bytes + (size_t)(bx0 + span_lo) * 3;
                const int off = (int)((uintptr_t)a & 15);
                for (int b = 0; b < 16; ++b) {
                    const int i = v * 16 + b - off;
                    if (i >= 0 && i < nb) hv_lds[d + b] = a[i];
                }
            } else {
                *reinterpret_cast<uint4*>(hv_lds + dsto[u]) = val[u];
            }
        }
    }
    __syncthreads();
    const int c = threadIdx.x & (HZ_COLS - 1);
    const int rg = threadIdx.x >> 6;  // rows rg and rg + 4
    if (c >= ncol) return;
    const int xx = col0 + c;
    const int lo = (xmin[xx] - span_lo) * 3;
    const int n = xcnt[xx];
    const int* offs = reinterpret_cast<const int*>(hv_lds + (size_t)HV_ROWS * row_stride);
    const int* k = offs + HV_ROWS + c * ksize;
    const int ya = row0 + rg, yb = ya + 4;
    const bool has_a = ya < box_h, has_b = yb < box_h;
    const uint8_t* pa = hv_lds + (size_t)rg * row_stride + (has_a ? offs[rg] : 0) + lo;
    const uint8_t* pb = hv_lds + (size_t)(rg + 4) * row_stride + (has_b ? offs[rg + 4] : 0) + lo;
    if (!has_b) pb = pa;  // shadow row a: no branch in the tap loop
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0, b0 = a0, b1 = a0, b2 = a0;
    // Four taps = 12 bytes = three dwords per row: aligned ds_read_b32 + v_alignbyte instead of twelve byte reads (the
    // byte-read form of this loop was LDS-instruction-bound: 72 us for 5000^2 -> 512^2), and 24-bit multiplies
    // (|coefficient| < 2^22, pixel < 2^8: v_mad_i32_i24 runs at full rate, v_mul_lo_u32 at a quarter).
    const uint32_t* qa = reinterpret_cast<const uint32_t*>(reinterpret_cast<uintptr_t>(pa) & ~(uintptr_t)3);
    const uint32_t* qb = reinterpret_cast<const uint32_t*>(reinterpret_cast<uintptr_t>(pb) & ~(uintptr_t)3);
    const unsigned sha = (unsigned)(reinterpret_cast<uintptr_t>(pa) & 3), shb = (unsigned)(reinterpret_cast<uintptr_t>(pb) & 3);
    uint32_t wa0 = qa[0], wb0 = qb[0];
    int j = 0, di = 1;
    for (; j + 4 <= n; j += 4, di += 3) {
        const uint32_t wa1 = qa[di], wa2 = qa[di + 1], wa3 = qa[di + 2];
        const uint32_t wb1 = qb[di], wb2 = qb[di + 1], wb3 = qb[di + 2];
        const uint32_t da0 = __builtin_amdgcn_alignbyte(wa1, wa0, sha), da1 = __builtin_amdgcn_alignbyte(wa2, wa1, sha),
                       da2 = __builtin_amdgcn_alignbyte(wa3, wa2, sha);
        const uint32_t db0 = __builtin_amdgcn_alignbyte(wb1, wb0, shb), db1 = __builtin_amdgcn_alignbyte(wb2, wb1, shb),
                       db2 = __builtin_amdgcn_alignbyte(wb3, wb2, shb);
        wa0 = wa3;
        wb0 = wb3;
        const int k0 = k[j], k1 = k[j + 1], k2 = k[j + 2], k3 = k[j + 3];
#define HV_B(d, i) (int)(((d) >> (8 * (i))) & 255u)
        a0 += __mul24(k0, HV_B(da0, 0)) + __mul24(k1, HV_B(da0, 3)) + __mul24(k2, HV_B(da1, 2)) + __mul24(k3, HV_B(da2, 1));
        a1 += __mul24(k0, HV_B(da0, 1)) + __mul24(k1, HV_B(da1, 0)) + __mul24(k2, HV_B(da1, 3)) + __mul24(k3, HV_B(da2, 2));
        a2 += __mul24(k0, HV_B(da0, 2)) + __mul24(k1, HV_B(da1, 1)) + __mul24(k2, HV_B(da2, 0)) + __mul24(k3, HV_B(da2, 3));
        b0 += __mul24(k0, HV_B(db0, 0)) + __mul24(k1, HV_B(db0, 3)) + __mul24(k2, HV_B(db1, 2)) + __mul24(k3, HV_B(db2, 1));
        b1 += __mul24(k0, HV_B(db0, 1)) + __mul24(k1, HV_B(db1, 0)) + __mul24(k2, HV_B(db1, 3)) + __mul24(k3, HV_B(db2, 2));
        b2 += __mul24(k0, HV_B(db0, 2)) + __mul24(k1, HV_B(db1, 1)) + __mul24(k2, HV_B(db2, 0)) + __mul24(k3, HV_B(db2, 3));
#undef HV_B
    }
    for (; j < n; ++j) {
        const int w = k[j];
        a0 += __mul24(w, (int)pa[3 * j + 0]);
        a1 += __mul24(w, (int)pa[3 * j + 1]);
        a2 += __mul24(w, (int)pa[3 * j + 2]);
        b0 += __mul24(w, (int)pb[3 * j + 0]);
        b1 += __mul24(w, (int)pb[3 * j + 1]);
        b2 += __mul24(w, (int)pb[3 * j + 2]);
    }
    if (has_a) {
        uint8_t* d = dst + ((size_t)ya * out_w + xx) * 3;
        d[0] = clip8(a0);
        d[1] = clip8(a1);
        d[2] = clip8(a2);
    }
    if (has_b) {
        uint8_t* d = dst + ((size_t)yb * out_w + xx) * 3;
        d[0] = clip8(b0);
        d[1] = clip8(b1);
        d[2] = clip8(b2);
    }
}

// Fallback horizontal pass without LDS staging (span too wide for the staged kernel).
__global__ void __launch_bounds__(256) k_resize_h_direct(const uint8_t* __restrict__ src, int src_h, int src_w,
                                                         int bx0, int by0, int box_h, uint8_t* __restrict__ dst,
                                                         int out_w, const int* __restrict__ xmin,
                                                         const int* __restrict__ xcnt, const int* __restrict__ kk,
                                                         int ksize) {
    const int xx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xx >= out_w || y >= box_h) return;
    const int sy = by0 + y;
    const bool row_in = (sy >= 0 && sy < src_h);
    const int lo = xmin[xx], n = xcnt[xx];
    const int* k = kk + (size_t)xx * ksize;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int j = 0; j < n; ++j) {
        const int px = bx0 + lo + j;
        if (row_in && px >= 0 && px < src_w) {
            const uint8_t* p = src + ((size_t)sy * src_w + px) * 3;
            const int w = k[j];
            s0 += w * p[0];
            s1 += w * p[1];
            s2 += w * p[2];
        }
    }
    uint8_t* d = dst + ((size_t)y * out_w + xx) * 3;
    d[0] = clip8(s0);
    d[1] = clip8(s1);
    d[2] = clip8(s2);
}

// Vertical pass: src u8 [in_h, w, 3] (already cropped/zero-filled by the horizontal pass, or read
// through the crop box when the horizontal pass was skipped) -> dst [out_h, w, 3].
// One thread per output byte column element; adjacent threads read adjacent bytes (coalesced).
__global__ void __launch_bounds__(256) k_resize_v(const uint8_t* __restrict__ src, int src_h, int src_w, int bx0,
                                                  int by0, int row_bytes, uint8_t* __restrict__ dst, int out_h,
                                                  const int* __restrict__ ymin, const int* __restrict__ ycnt,
                                                  const int* __restrict__ kk, int ksize, int boxed) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // byte index inside an output row
    const int yy = blockIdx.y;
    if (i >= row_bytes) return;
    const int lo = ymin[yy], n = ycnt[yy];
    const int* k = kk + (size_t)yy * ksize;
    int s = 1 << (PRECISION_BITS - 1);
    if (!boxed) {
        const uint8_t* p = src + (size_t)lo * row_bytes + i;
        for (int j = 0; j < n; ++j) s += k[j] * p[(size_t)j * row_bytes];
    } else {
        const int px = bx0 + i / 3;
        const bool col_in = (px >= 0 && px < src_w);
        for (int j = 0; j < n; ++j) {
            const int sy = by0 + lo + j;
            if (col_in && sy >= 0 && sy < src_h) s += k[j] * src[((size_t)sy * src_w + px) * 3 + (i % 3)];
        }
    }
    dst[(size_t)yy * row_bytes + i] = clip8(s);
}

// Pure crop with zero fill.
__global__ void __launch_bounds__(256) k_crop(const uint8_t* __restrict__ src, int src_h, int src_w, int bx0,
                                              int by0, uint8_t* __restrict__ dst, int out_h, int out_w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (i >= out_w * 3) return;
    const int px = bx0 + i / 3, sy = by0 + y;
    uint8_t v = 0;
    if (px >= 0 && px < src_w && sy >= 0 && sy < src_h) v = src[((size_t)sy * src_w + px) * 3 + (i % 3)];
    dst[(size_t)y * out_w * 3 + i] = v;
}

// LUT normalise + patchify: out[row][col] with row = ((by*gwm + bx)*m + my)*m + mx (block-major over
// merge x merge), col = ((c*T + t)*P + py)*P + px; value = lut[c][img[y][x][c]].
// One thread per (row, c, py) writes T*P contiguous-ish floats; consecutive threads walk px fastest.
__global__ void __launch_bounds__(256) k_patchify(const uint8_t* __restrict__ img, int h, int w,
                                                  const float* __restrict__ lut, float* __restrict__ out, int P,
                                                  int M, int T, int C) {
    const int cols = C * T * P * P;
    const int gh = h / P, gw = w / P;
    const size_t total = (size_t)gh * gw * cols;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int col = (int)(idx % cols);
    const int row = (int)(idx / cols);
    const int px = col % P;
    const int py = (col / P) % P;
    const int c = col / (P * P * T);
    const int mx = row % M;
    const int my = (row / M) % M;
    const int blk = row / (M * M);
    const int bx = blk % (gw / M);
    const int by = blk / (gw / M);
    const int y = (by * M + my) * P + py;
    const int x = (bx * M + mx) * P + px;
    out[idx] = lut[c * 256 + img[((size_t)y * w + x) * 3 + c]];
}

void ze_launch_resize_h(const uint8_t* src, int src_h, int src_w, int bx0, int by0, int box_h, uint8_t* dst,
                        int out_w, const int* xmin, const int* xcnt, const int* kk, int ksize, int max_span,
                        hipStream_t s, int inside) {
    // box inside the image: 16-byte staging, 8 rows per workgroup (rows of max_span * 3 bytes + 16 of misalignment + 16
    // of vector tail, a multiple of 16; up to 160 KB of LDS)
    const int row_stride = (max_span * 3 + 32 + 15) & ~15;
    const size_t lds = (size_t)HV_ROWS * row_stride + HV_ROWS * sizeof(int) + (size_t)64 * ksize * sizeof(int);
    if (inside && lds <= 64 * 1024) {
        dim3 gv(ze_cdiv(out_w, 64), ze_cdiv(box_h, HV_ROWS));
        k_resize_h_vec<<<gv, 256, lds, s>>>(src, src_h, src_w, bx0, by0, box_h, dst, out_w, xmin, xcnt, kk, ksize, row_stride);
        return;
    }
    dim3 grid(ze_cdiv(out_w, 64), ze_cdiv(box_h, 4));
    if (max_span <= HZ_MAXSPAN)
        k_resize_h<<<grid, 256, 0, s>>>(src, src_h, src_w, bx0, by0, box_h, dst, out_w, xmin, xcnt, kk, ksize);
    else
        k_resize_h_direct<<<grid, 256, 0, s>>>(src, src_h, src_w, bx0, by0, box_h, dst, out_w, xmin, xcnt, kk,
                                               ksize);
}

void ze_launch_resize_v(const uint8_t* src, int src_h, int src_w, int bx0, int by0, int row_bytes, uint8_t* dst,
                        int out_h, const int* ymin, const int* ycnt, const int* kk, int ksize, int boxed,
                        hipStream_t s) {
    dim3 grid(ze_cdiv(row_bytes, 256), out_h);
    k_resize_v<<<grid, 256, 0, s>>>(src, src_h, src_w, bx0, by0, row_bytes, dst, out_h, ymin, ycnt, kk, ksize,
                                    boxed);
}

void ze_launch_crop(const uint8_t* src, int src_h, int src_w, int bx0, int by0, uint8_t* dst, int out_h, int out_w,
                    hipStream_t s) {
    dim3 grid(ze_cdiv(out_w * 3, 256), out_h);
    k_crop<<<grid, 256, 0, s>>>(src, src_h, src_w, bx0, by0, dst, out_h, out_w);
}

void ze_launch_patchify(const uint8_t* img, int h, int w, const float* lut, float* out, int P, int M, int T, int C,
                        hipStream_t s) {
    const size_t total = (size_t)(h / P) * (w / P) * C * T * P * P;
    k_patchify<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(img, h, w, lut, out, P, M, T, C);
}
