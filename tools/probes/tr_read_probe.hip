// Probe: lane/element mapping of ds_read_b64_tr_b16 (gfx950), the transposed LDS read used for the PV operand of
// the flash-attention kernel.  Expectation (cdna_hip_programming.md T10): per 16-lane group, lane 4q+p supplies the
// address of row q, columns 4p..4p+3 of a 4 x 16 block; lane i receives column i, rows 0..3 in elements 0..3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const uint16_t* in, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t s[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) s[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15, q = l >> 2, p = l & 3;
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(s + (4 * g + q) * 64 + 4 * p));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (uint16_t)v[e];
}
int main() {
    uint16_t h[64 * 64], o[256];
    for (int i = 0; i < 64 * 64; ++i) h[i] = (uint16_t)i;  // value = row * 64 + col
    uint16_t *di, *dout;
    hipMalloc(&di, sizeof(h));
    hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
            const int g = lane >> 4, i = lane & 15, want = (4 * g + e) * 64 + i;
            if (o[lane * 4 + e] != want) ++bad;
        }
    printf("tr_read_probe: %d mismatches against 'lane i <- column i, element e <- row e'\n", bad);
    for (int lane = 0; lane < 20; ++lane)
        printf("lane %2d: (r%d,c%d) (r%d,c%d) (r%d,c%d) (r%d,c%d)\n", lane, o[lane * 4] / 64, o[lane * 4] % 64, o[lane * 4 + 1] / 64,
               o[lane * 4 + 1] % 64, o[lane * 4 + 2] / 64, o[lane * 4 + 2] % 64, o[lane * 4 + 3] / 64, o[lane * 4 + 3] % 64);
    return bad != 0;
}
