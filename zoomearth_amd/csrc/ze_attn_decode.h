// Device bodies of the one-token (decode) attention (ze_attn_decode.hip; until round 3 also the fused per-layer kernel).  The
// FRESH form moves cross-workgroup data with sc1 (write-through / L1-bypassing) accesses so it can be handed over inside a
// running launch (the in-launch merge of the slices).
#pragma once
#include "ze_kernels.h"

#define AD_GMAX 8
#define AD_STRIDE 132  // floats per (split, head) partial: m, l, pad, pad, o[128]
#define AD_TOK 64
#ifndef AD_MINCHUNK
#define AD_MINCHUNK 64  // tokens per slice, at least (a multiple of AD_TOK)
#endif

__device__ __forceinline__ void split_geometry(int ctx, int max_splits, int& chunk, int& nsplit) {
    chunk = (ctx + max_splits - 1) / max_splits;
    chunk = (chunk + AD_MINCHUNK - 1) / AD_MINCHUNK * AD_MINCHUNK;
    nsplit = (ctx + chunk - 1) / chunk;
}

// LDS of one slice block (carved by the caller so fused kernels can share their LDS)
struct ad_split_lds {
    __attribute__((aligned(16))) uint8_t sK[AD_TOK * 256];  // K tile image (ad_off layout), 16 KB
    __attribute__((aligned(16))) uint8_t sV[AD_TOK * 256];  // V tile image, 16 KB
};

typedef __attribute__((ext_vector_type(8))) __bf16 ad_bf16x8;
typedef __attribute__((ext_vector_type(4))) float ad_f32x4;
typedef short ad_v4s __attribute__((ext_vector_type(4)));
typedef short ad_v8s __attribute__((ext_vector_type(8)));

// LDS image of a [64 keys][128 d] bf16 tile with 256-byte rows: byte offset of 16-B chunk ch (0..15) of row `row`
// (the image of the prefill flash kernel, ze_attention.hip; the swizzle ze_kv_swz, ze_kernels.h, is conflict-free for the
// ds_read_b128 row reads of K as an MFMA operand and for the ds_read_b64_tr_b16 transposed reads of V^T).
__device__ __forceinline__ int ad_off(int row, int ch) { return 256 * row + 16 * (ch ^ ze_kv_swz(row)); }
// f32 pair -> packed bf16 (round to nearest even: v_cvt_pk_bf16_f32).  Written as a CONVERSION, not as inline asm (round 5): on
// gfx950 an MFMA that reads a VGPR a VALU instruction wrote needs two wait states in between; hipcc's hazard recogniser inserts
// them for instructions it knows, but it cannot see a VALU write INSIDE an asm statement -- with `asm("v_cvt_pk_bf16_f32 ...")`
// feeding P^T to the P V MFMA one instruction later, the MFMA read the register's OLD value (keys 2, 3 of every group of four
// lost in two instantiations of k_attn_decode_wave_long; DESIGN.md 3, rule "no VALU result leaves an asm statement towards an
// MFMA"; tools/check_mfma_hazards.py proves it on the generated code of the whole library).  Same instruction, same bits.
// AD_PACK_ASM / AD_PACK_PRE / AD_PACK_POST: the old form and its padding, for tools/probes/fence_hunt.sh only.
__device__ __forceinline__ uint32_t ad_pack_bf16(float lo, float hi) {
    uint32_t r;
#ifdef AD_PACK_ASM
#ifndef AD_PACK_PRE
#define AD_PACK_PRE ""
#endif
#ifndef AD_PACK_POST
#define AD_PACK_POST ""
#endif
    asm(AD_PACK_PRE "v_cvt_pk_bf16_f32 %0, %1, %2" AD_PACK_POST : "=v"(r) : "v"(lo), "v"(hi));
#else
    typedef __attribute__((ext_vector_type(2))) float ad_f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 ad_bf16x2_;
    const ad_bf16x2_ b = __builtin_convertvector(ad_f32x2_{lo, hi}, ad_bf16x2_);
    __builtin_memcpy(&r, &b, 4);
#endif
    return r;
}

// FRESH: q / K / V / partials cross workgroups INSIDE the running launch (fused layer kernel): every such access
// is a write-through (sc1) store or an L1-bypassing (sc1) load.  Arithmetic is identical in both forms.
// Addresses are (wave-uniform base, per-lane byte offset): the buffer descriptor of the sc1 forms must live in
// SGPRs -- a per-lane base pointer makes hipcc wrap every access in a 64-trip readfirstlane waterfall loop
// (measured: the fused kernel 4x slower than the four launches it replaces).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ad_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
template <bool FRESH>
__device__ __forceinline__ uint4 ad_load16(const void* base, uint32_t off) {
    if (FRESH) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(ad_rsrc(base), off, 0, 16);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(base) + off);
}
template <bool FRESH>
__device__ __forceinline__ float ad_load4(const void* base, uint32_t off) {
    if (FRESH) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ad_rsrc(base), off, 0, 16));
    return *reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(base) + off);
}
template <bool FRESH>
__device__ __forceinline__ void ad_store2(void* base, uint32_t off, bf16_t v) {
    if (FRESH)
        __builtin_amdgcn_raw_buffer_store_b16(v, ad_rsrc(base), off, 0, 16);
    else
        *reinterpret_cast<bf16_t*>(reinterpret_cast<uint8_t*>(base) + off) = v;
}
template <bool FRESH>
__device__ __forceinline__ void ad_store16(void* base, uint32_t off, float a, float b, float c, float d) {
    if (FRESH) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
        u32x4_t v;
        v.x = __float_as_uint(a);
        v.y = __float_as_uint(b);
        v.z = __float_as_uint(c);
        v.w = __float_as_uint(d);
        __builtin_amdgcn_raw_buffer_store_b128(v, ad_rsrc(base), off, 0, 16);
    } else {
        *reinterpret_cast<float4*>(reinterpret_cast<uint8_t*>(base) + off) = make_float4(a, b, c, d);
    }
}

// FRESH: q and the newest K/V row were written inside this launch (fused layer kernel).  PUB: the partials are
// published write-through (sc1) because they are merged inside this launch (by the fused kernel's merge phase or by
// the last-arriving slice workgroup of the stand-alone kernel).
// NS: consecutive slices one workgroup handles (split = first of them).  NS = 2 requests the K/V of BOTH slices before
// computing either (the batched step: half the workgroups, twice the bytes in flight per workgroup, the load latency
// exposed once per two slices); results per slice -- and therefore the merge -- are the same for every NS.
template <bool FRESH, bool PUB = FRESH, int NS = 1>
__device__ __forceinline__ void attn_split_body(ad_split_lds& L, const bf16_t* __restrict__ q,
                                                const bf16_t* __restrict__ kcache, const bf16_t* __restrict__ vcache,
                                                int ctx, int kvh, int split, int heads, int kv_heads, int max_ctx,
                                                float scale_log2e, float* __restrict__ ws, int max_splits) {
    // One slice of the context for the (up to) 8 q heads of a kv head, on the matrix cores with the SWAPPED products
    // of the prefill kernel: S^T = K Q^T (16 keys x 16 head columns per MFMA tile, columns >= G are zero queries),
    // O^T = V^T P^T.  Per 64-key round: every thread stages 4 K + 4 V 16-B pieces into the two LDS images; EVERY
    // wave then forms the whole S^T of the round (16 MFMAs -- redundant across the four waves, which keeps the
    // softmax statistics of a head lane-local plus two shuffles and needs no cross-wave exchange) and owns two of
    // the eight 16-d tiles of O^T (4 MFMAs, V^T fragments by ds_read_b64_tr_b16, P from the accumulator straight
    // into the B operand, rounded to bf16 as HF's eager attention rounds it).  Replaces 131 k fp32 FMAs per round
    // on the vector ALUs (the slice kernel spent about 2 us of its 11.5 there).
    constexpr int D = 128;
    const int G = heads / kv_heads;
    int chunk, nsplit;
    split_geometry(ctx, max_splits, chunk, nsplit);
    if (split >= nsplit) return;
    const int tid = threadIdx.x, gid = tid >> 4, li = tid & 15, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    // Q^T fragments: lane (head column fr, k-group fq) holds d = ks*32 + fq*8 .. +7 of its head
    ad_bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (fr < G) u = ad_load16<FRESH>(q, (uint32_t)(((kvh * G + fr) * D + (ks * 4 + fq) * 8) * 2));
        qf[ks] = *reinterpret_cast<const ad_bf16x8*>(&u);
    }
    const bf16_t* kb = kcache + (size_t)kvh * max_ctx * D;  // wave-uniform bases, per-lane byte offsets
    const bf16_t* vb = vcache + (size_t)kvh * max_ctx * D;
    ad_f32x4 oacc[2];
    float m_run, l_run;
    auto reset = [&]() {
        oacc[0] = oacc[1] = ad_f32x4{0.f, 0.f, 0.f, 0.f};
        m_run = -INFINITY;
        l_run = 0.f;
    };

    // ---------------- loads of one 64-key round.  Only the newest row was written inside this launch (sc1 stores of the
    // fused QKV phase) and is read with sc1 loads; every older row comes from earlier launches.  The newest row is
    // fetched once per lane up front (workgroup-uniform branch) and selected in, so the eight row loads stay free of
    // per-lane control flow.
    auto load_round = [&](int base, int t1, uint4 (&ku)[4], uint4 (&vu)[4]) {
        uint4 kn = make_uint4(0, 0, 0, 0), vn = make_uint4(0, 0, 0, 0);
        const int tn = ctx - 1;
        if (FRESH && tn >= base && tn < base + AD_TOK) {
            const uint32_t off = (uint32_t)((tn * D + li * 8) * 2);
            kn = ad_load16<true>(kb, off);
            vn = ad_load16<true>(vb, off);
        }
        // Branch-free: rows past the slice re-read its last row (a valid address) and are zeroed when the images are
        // written -- a per-lane "in range ? load : 0" puts every load behind an exec-mask branch (and with TWO such
        // guarded groups in flight, the NS = 2 form, the build of ROCm 7.2's hipcc produced wrong rows for the second).
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = base + gid + 16 * i;
            const uint32_t off = (uint32_t)((min(t, t1 - 1) * D + li * 8) * 2);
            ku[i] = ad_load16<false>(kb, off);
            vu[i] = ad_load16<false>(vb, off);
            if (FRESH && t == tn) {
                ku[i] = kn;
                vu[i] = vn;
            }
        }
    };
    // ---------------- one 64-key round from staged registers: LDS images, S^T, online softmax, O^T
    auto compute_round = [&](int base, int t1, const uint4 (&ku)[4], const uint4 (&vu)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned m = (base + gid + 16 * i < t1) ? 0xffffffffu : 0u;  // rows past the slice are zero
            *reinterpret_cast<uint4*>(L.sK + ad_off(gid + 16 * i, li)) = make_uint4(ku[i].x & m, ku[i].y & m, ku[i].z & m, ku[i].w & m);
            *reinterpret_cast<uint4*>(L.sV + ad_off(gid + 16 * i, li)) = make_uint4(vu[i].x & m, vu[i].y & m, vu[i].z & m, vu[i].w & m);
        }
        __syncthreads();
        // S^T = K Q^T: 4 key tiles x 4 steps of 32 d
        ad_f32x4 sacc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) sacc[n] = ad_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const uint4 ka = *reinterpret_cast<const uint4*>(L.sK + ad_off(n * 16 + fr, ks * 4 + fq));
                sacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const ad_bf16x8*>(&ka), qf[ks],
                                                                 sacc[n], 0, 0, 0);
            }
        // online softmax of head fr over keys base + n*16 + fq*4 + r
        float p[4][4];
        float mx = -INFINITY;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = base + n * 16 + fq * 4 + r < t1;
                const float sv = ok ? sacc[n][r] * scale_log2e : -INFINITY;
                p[n][r] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = exp2f(m_run - m_use);  // m_run = -inf -> 0
        float rs = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(p[n][r] - m_use);  // arguments <= 0
                p[n][r] = e;
                rs += e;
            }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            oacc[jj][0] *= alpha;
            oacc[jj][1] *= alpha;
            oacc[jj][2] *= alpha;
            oacc[jj][3] *= alpha;
        }
        // O^T += V^T P^T for this wave's d-tiles 2*wid, 2*wid + 1: per 32-key step the lane's 8 keys are
        // (ks*32 + fq*4 + 0..3) and (ks*32 + 16 + fq*4 + 0..3), the same permutation on both operands
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint32_t pw[4];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                pw[2 * h2] = ad_pack_bf16(p[2 * ks + h2][0], p[2 * ks + h2][1]);
                pw[2 * h2 + 1] = ad_pack_bf16(p[2 * ks + h2][2], p[2 * ks + h2][3]);
            }
            const uint4 pq = make_uint4(pw[0], pw[1], pw[2], pw[3]);
            const ad_bf16x8 pb = *reinterpret_cast<const ad_bf16x8*>(&pq);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * wid + jj;
                const int tq = fr >> 2, tp = fr & 3;
                const int r0a = ks * 32 + fq * 4, r0b = r0a + 16;
                const int ch = j * 2 + (tp >> 1), half = 8 * (tp & 1);
                const ad_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) ad_v4s*)(L.sV + ad_off(r0a + tq, ch) + half));
                const ad_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) ad_v4s*)(L.sV + ad_off(r0b + tq, ch) + half));
                const ad_v8s va = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                oacc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const ad_bf16x8*>(&va), pb, oacc[jj],
                                                                  0, 0, 0);
            }
        }
        __syncthreads();  // the images are rewritten by the next round (and reused by the caller after the last)
    };
    // partial of head fr for slice sp: (m, l) and O[d = (2*wid + jj)*16 + fq*4 .. +3]
    auto store_partial = [&](int sp) {
        if (fr < G) {
            const uint32_t dst = (uint32_t)((sp * heads + kvh * G + fr) * AD_STRIDE * 4);
            if (wid == 0 && fq == 0) ad_store16<PUB>(ws, dst, m_run, l_run, 0.f, 0.f);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                ad_store16<PUB>(ws, dst + (uint32_t)((4 + (2 * wid + jj) * 16 + fq * 4) * 4), oacc[jj][0], oacc[jj][1],
                                oacc[jj][2], oacc[jj][3]);
        }
    };

    if (NS == 2 && chunk == AD_TOK) {
        // two one-round slices: every load of both is in flight before the first product
        const int sa = split * 2, sb = sa + 1;
        if (sa >= nsplit) return;
        const bool two = sb < nsplit;  // workgroup-uniform
        uint4 ku[4], vu[4], ku2[4], vu2[4];
        load_round(sa * chunk, min(ctx, sa * chunk + chunk), ku, vu);
        if (two) load_round(sb * chunk, min(ctx, sb * chunk + chunk), ku2, vu2);
        reset();
        compute_round(sa * chunk, min(ctx, sa * chunk + chunk), ku, vu);
        store_partial(sa);
        if (two) {
            reset();
            compute_round(sb * chunk, min(ctx, sb * chunk + chunk), ku2, vu2);
            store_partial(sb);
        }
        return;
    }
#pragma unroll 1
    for (int sp = split * NS; sp < min(nsplit, split * NS + NS); ++sp) {
        const int t0 = sp * chunk, t1 = min(ctx, t0 + chunk);
        reset();
        for (int base = t0; base < t1; base += AD_TOK) {
            uint4 ku[4], vu[4];
            load_round(base, t1, ku, vu);
            compute_round(base, t1, ku, vu);
        }
        store_partial(sp);
    }
}

// grid = heads, 128 threads (one per output dim): split weights are computed lane-parallel by the first wave.
// merge of the slices of head h by threads 0..127 of a block (d = thread = output dim); sW: 64 floats, sInv: 1 float
template <bool FRESH>
__device__ __forceinline__ void attn_combine_body(float* sW, float* sInvp, const float* __restrict__ ws, int ctx, int h,
                                                  int d, int heads, int max_splits, bf16_t* __restrict__ out) {
    constexpr int D = 128;
    float& sInv = *sInvp;
    int chunk, nsplit;
    split_geometry(ctx, max_splits, chunk, nsplit);
    if (d < 64) {
        float m = -INFINITY, l = 0.f;
        if (d < nsplit) {
            const uint4 ml = ad_load16<FRESH>(ws, (uint32_t)((d * heads + h) * AD_STRIDE * 4));
            m = __uint_as_float(ml.x);
            l = __uint_as_float(ml.y);
        }
        const float mx = wave_max(m);
        const float w = (m == -INFINITY) ? 0.f : exp2f(m - mx);
        const float tot = wave_sum(w * l);
        sW[d] = w;
        if (d == 0) sInv = tot > 0.f ? 1.0f / tot : 0.f;
    }
    __syncthreads();
    float acc = 0.f;
    const uint32_t p = (uint32_t)((h * AD_STRIDE + 4 + d) * 4);
    const uint32_t step = (uint32_t)(heads * AD_STRIDE * 4);
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = ad_load4<FRESH>(ws, p + (s + u) * step);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(sW[s + u], x[u], acc);
    }
    for (; s < nsplit; ++s) acc = fmaf(sW[s], ad_load4<FRESH>(ws, p + s * step), acc);
    ad_store2<FRESH>(out, (uint32_t)((h * D + d) * 2), f32_to_bf16(acc * sInv));
}

// Merge of the slices of ALL q heads of one kv head by one 256-thread workgroup (the last-arriving slice workgroup):
// wave w computes the slice weights of heads 2w, 2w+1 (lane = slice), then thread (head og, dims od..od+3)
// accumulates in slice order -- per output element the arithmetic of attn_combine_body.  Partials are read with
// sc1 loads (they were published sc1 inside this launch).  sW: [AD_GMAX][64] floats, sInv: [AD_GMAX].
template <int MB = 24>
// out: the chain's output row (row-major), or, with frag_slices = heads * 128 / 32 != 0, the base of a fragment-major
// buffer in which this chain is row `frag_row` (the layout of k_rmsnorm(frag), read by k_gemm_skinny<..., FRAG>).
__device__ __forceinline__ void attn_merge_group(float* sW, float* sInv, const float* __restrict__ ws, int ctx, int kvh,
                                                 int heads, int kv_heads, int max_splits, bf16_t* __restrict__ out,
                                                 int frag_slices = 0, int frag_row = 0, int nsplit_given = 0) {
    constexpr int D = 128;
    const int G = heads / kv_heads;
    int chunk, nsplit;
    split_geometry(ctx, max_splits, chunk, nsplit);
    if (nsplit_given > 0) nsplit = nsplit_given;  // the caller's own partition (ze_attn_batch.hip)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int og = tid >> 5, od = (tid & 31) * 4;
    const int hc = kvh * G + min(og, G - 1);  // idle threads (og >= G) shadow the last head: no branch around loads
    const uint32_t p = (uint32_t)((hc * AD_STRIDE + 4 + od) * 4), step = (uint32_t)(heads * AD_STRIDE * 4);
    // Everything the merge needs is requested at once -- the (m, l) pairs and the first 24 slices of this thread's
    // four output dims (contexts up to 1536 tokens) -- so the last arriver pays one memory round trip, not one per
    // dependent step.
    uint4 ml[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int g = min(wid * 2 + hh, G - 1);
        ml[hh] = ad_load16<true>(ws, (uint32_t)((min(lane, nsplit - 1) * heads + kvh * G + g) * AD_STRIDE * 4));
    }
    uint4 x0[MB];
#pragma unroll
    for (int u = 0; u < MB; ++u) x0[u] = ad_load16<true>(ws, p + min(u, nsplit - 1) * step);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int g = wid * 2 + hh;
        if (g < G) {  // wave-uniform
            const float m = lane < nsplit ? __uint_as_float(ml[hh].x) : -INFINITY;
            const float l = lane < nsplit ? __uint_as_float(ml[hh].y) : 0.f;
            const float mx = wave_max(m);
            const float w = (m == -INFINITY) ? 0.f : exp2f(m - mx);
            const float tot = wave_sum(w * l);
            sW[g * 64 + lane] = w;
            if (lane == 0) sInv[g] = tot > 0.f ? 1.0f / tot : 0.f;
        }
    }
    __syncthreads();
    if (og >= G) return;
    const int h = kvh * G + og;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int u = 0; u < MB; ++u) {
        if (u < nsplit) {
            const float w = sW[og * 64 + u];
            a0 = fmaf(w, __uint_as_float(x0[u].x), a0);
            a1 = fmaf(w, __uint_as_float(x0[u].y), a1);
            a2 = fmaf(w, __uint_as_float(x0[u].z), a2);
            a3 = fmaf(w, __uint_as_float(x0[u].w), a3);
        }
    }
    int s = MB;
    for (; s + 8 <= nsplit; s += 8) {
        uint4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = ad_load16<true>(ws, p + (s + u) * step);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float w = sW[og * 64 + s + u];
            a0 = fmaf(w, __uint_as_float(x[u].x), a0);
            a1 = fmaf(w, __uint_as_float(x[u].y), a1);
            a2 = fmaf(w, __uint_as_float(x[u].z), a2);
            a3 = fmaf(w, __uint_as_float(x[u].w), a3);
        }
    }
    for (; s < nsplit; ++s) {
        const uint4 x = ad_load16<true>(ws, p + s * step);
        const float w = sW[og * 64 + s];
        a0 = fmaf(w, __uint_as_float(x.x), a0);
        a1 = fmaf(w, __uint_as_float(x.y), a1);
        a2 = fmaf(w, __uint_as_float(x.z), a2);
        a3 = fmaf(w, __uint_as_float(x.w), a3);
    }
    const float inv = sInv[og];
    uint2 o;
    o.x = pack_bf16x2(a0 * inv, a1 * inv);
    o.y = pack_bf16x2(a2 * inv, a3 * inv);
    const int col = h * D + od;
    bf16_t* dst = out + col;
    if (frag_slices)
        dst = out + ((((size_t)(frag_row >> 4) * frag_slices + (col >> 5)) * 64 + ((col & 31) >> 3) * 16 + (frag_row & 15)) << 3) +
              (col & 7);
    *reinterpret_cast<uint2*>(dst) = o;
}
