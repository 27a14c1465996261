// Probe: the per-wave decode attention kernels of ze_attn_batch.hip on one synthetic chain -- the shipped 384-key pipelined form (knob 8 = 0)
// against the 192-key kernel (knob 8 = 4), element by element, and both against a float64 host reference; "onehot" / "spike T D ..."
// arguments put structured V rows in (how the mis-scheduled instantiations of round 4 were narrowed down: DESIGN.md 7e).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -fno-slp-vectorize -Iinclude -Izoomearth_amd/csrc \
//        -o tools/probes/bin/attn_wave_probe tools/probes/attn_wave_probe.hip zoomearth_amd/csrc/ze_attn_batch.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "ze_kernels.h"

int ze_gemv_knobs[24] = {0};
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

static bf16_t f2b(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
static float b2f(bf16_t b) {
    unsigned u = (unsigned)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char** argv) {
    const int heads = 16, kvh = 2, D = 128, max_ctx = 1024, G = heads / kvh;
    const int wparts = (max_ctx + 191) / 192;
    std::vector<bf16_t> hq(heads * D), hk((size_t)kvh * max_ctx * D), hv((size_t)kvh * max_ctx * D);
    unsigned s = 7u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& x : hq) x = f2b(rnd());
    for (auto& x : hk) x = f2b(rnd());
    for (auto& x : hv) x = f2b(rnd());
    const bool onehot = argc > 1 && !strcmp(argv[1], "onehot");  // V[t][d] = (t % 128 == d): out[d] = sum of the p of keys t = d (mod 128)
    const bool spike = argc > 3 && !strcmp(argv[1], "spike");  // V = 1 at (key T0, dim D0) only: out[d] = p_T0 at d = D0
    if (spike) {
        for (auto& x : hv) x = 0;
        for (int a = 2; a + 1 < argc; a += 2) {  // (any number of "key dim" pairs)
            const int T0 = atoi(argv[a]), D0 = atoi(argv[a + 1]);
            for (int kh = 0; kh < kvh; ++kh) hv[((size_t)kh * max_ctx + T0) * D + D0] = f2b(1.0f);
        }
    }
    if (onehot)
        for (int kh = 0; kh < kvh; ++kh)
            for (int t = 0; t < max_ctx; ++t)
                for (int d = 0; d < D; ++d) hv[((size_t)kh * max_ctx + t) * D + d] = f2b(t % 128 == d ? 1.0f : 0.0f);
    bf16_t *q, *k, *v, *out;
    float* ws;
    unsigned* tickets;
    ze_seq_dev* st;
    int* ids;
    CHECK(hipMalloc(&q, hq.size() * 2));
    CHECK(hipMalloc(&k, hk.size() * 2));
    CHECK(hipMalloc(&v, hv.size() * 2));
    CHECK(hipMalloc(&out, heads * D * 2));
    CHECK(hipMalloc(&ws, (size_t)wparts * heads * 132 * 4 * 2));
    CHECK(hipMalloc(&tickets, 64));
    CHECK(hipMalloc(&st, sizeof(ze_seq_dev)));
    CHECK(hipMalloc(&ids, 4));
    CHECK(hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(k, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(v, hv.data(), hv.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemset(tickets, 0, 64));
    int zero = 0;
    CHECK(hipMemcpy(ids, &zero, 4, hipMemcpyHostToDevice));
    const float scale = 1.0f / sqrtf((float)D);
    const int ctxs_all[] = {64, 128, 191, 192, 193, 256, 384, 385, 500, 768, 1000};
    const int ctxs_few[] = {128, 191};
    std::vector<int> ctxs((onehot || spike) ? std::vector<int>(ctxs_few, ctxs_few + 2) : std::vector<int>(ctxs_all, ctxs_all + 11));
    for (int ctx : ctxs) {
        ze_seq_dev h = {};
        h.ctx = ctx - 1;  // the kernel attends over ctx + 1 rows
        CHECK(hipMemcpy(st, &h, sizeof(h), hipMemcpyHostToDevice));
        // float64 reference (P rounded to bf16 as the kernels do is NOT modelled: tolerance below)
        std::vector<double> ref(heads * D);
        for (int hd = 0; hd < heads; ++hd) {
            const int kh = hd / G;
            std::vector<double> sc(ctx);
            double mx = -1e300;
            for (int t = 0; t < ctx; ++t) {
                double a = 0;
                for (int d = 0; d < D; ++d) a += (double)b2f(hq[hd * D + d]) * b2f(hk[((size_t)kh * max_ctx + t) * D + d]);
                sc[t] = a * scale;
                mx = fmax(mx, sc[t]);
            }
            double l = 0;
            for (int t = 0; t < ctx; ++t) { sc[t] = exp(sc[t] - mx); l += sc[t]; }
            for (int d = 0; d < D; ++d) {
                double o = 0;
                for (int t = 0; t < ctx; ++t) o += sc[t] * b2f(hv[((size_t)kh * max_ctx + t) * D + d]);
                ref[hd * D + d] = o / l;
            }
        }
        std::vector<bf16_t> base(heads * D), got(heads * D);
        const int live = (ctx + 1 + 191) / 192;
        for (int knob : {0, 4}) {  // 0: k_attn_decode_wave_long (384-key parts), 4: k_attn_decode_wave (192-key parts)
            ze_gemv_knobs[8] = knob;
            CHECK(hipMemset(out, 0xff, heads * D * 2));
            ze_launch_attn_decode_stream(q, heads * D, k, v, (size_t)kvh * max_ctx * D, out, heads * D, st, ids, 1, heads, kvh, max_ctx, scale, ws,
                                         wparts, tickets, 0, 0, live, nullptr);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(got.data(), out, heads * D * 2, hipMemcpyDeviceToHost));
            if (knob == 0) base = got;
            double eref = 0, ebase = 0;
            int worst = 0;
            for (int i = 0; i < heads * D; ++i) {
                eref = fmax(eref, fabs(b2f(got[i]) - ref[i]));
                const double e = fabs(b2f(got[i]) - b2f(base[i]));
                if (e > ebase) { ebase = e; worst = i; }
            }
            // which 16-wide d tiles of which heads deviate from the 192-key kernel by more than 0.01
            char map[17] = {0};
            for (int hd = 0; hd < heads; ++hd) {
                int bad = 0;
                for (int d = 0; d < D; ++d) bad += fabs(b2f(got[hd * D + d]) - b2f(base[hd * D + d])) > 0.01;
                map[hd] = bad ? (bad > 9 ? '#' : '0' + bad) : '.';
            }
            if (spike && ctx == 191) {
                printf("   ctx %d knob %d head 0 non-zero outputs:", ctx, knob);
                for (int d = 0; d < D; ++d)
                    if (fabs(b2f(got[d])) > 1e-6) printf(" d %d: %.5f (ref %.5f)", d, b2f(got[d]), ref[d]);
                printf("\n");
            }
            if (onehot && knob != 0 && ebase > 0.002 && ctx <= 193) {
                printf("   ctx %d knob %d head 0, keys whose probability mass differs from knob 0 (d: got / knob 0):", ctx, knob);
                for (int d = 0; d < D; ++d)
                    if (fabs(b2f(got[d]) - b2f(base[d])) > 0.0005) printf(" %d: %.4f / %.4f", d, b2f(got[d]), b2f(base[d]));
                printf("\n");
            }
            printf("ctx %4d knob %d: max |got - float64| %.4f, max |got - knob 0| %.4f (head %d, d %d)  heads off by > 0.01: %s\n", ctx, knob, eref,
                   ebase, worst / D, worst % D, map);
        }
    }
    return 0;
}
