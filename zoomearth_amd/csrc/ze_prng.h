// Device/host mirror of oracle/prng.py (counter-based splitmix64 stream; Irwin-Hall(4 x u16) normals).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define ZE_HD __host__ __device__ __forceinline__
#else
#define ZE_HD inline
#endif

#define ZE_IH4_STD 37837.22722412452

ZE_HD uint64_t ze_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
ZE_HD uint64_t ze_stream64(uint64_t seed, uint64_t i) { return ze_mix64(seed + (i + 1) * 0x9E3779B97F4A7C15ull); }

// float32(int32(s) - 131070) * c  with c = float32(std / IH4_STD) computed by the caller
ZE_HD float ze_normal_ih4(uint64_t seed, uint64_t i, float c) {
    const uint64_t h = ze_stream64(seed, i);
    const int s = (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) + (int)(h >> 48);
    return (float)(s - 131070) * c;
}

inline uint64_t ze_fnv1a64(const char* s) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (; *s; ++s) {
        h ^= (uint8_t)*s;
        h *= 0x100000001B3ull;
    }
    return h;
}
inline uint64_t ze_tensor_seed(uint64_t global_seed, const char* name) {
    return ze_mix64(global_seed ^ ze_fnv1a64(name));
}
