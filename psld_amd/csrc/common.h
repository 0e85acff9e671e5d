// Shared device/host helpers for libpsld_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PSLD_OK 0
#define PSLD_ERR_ARG 1      // bad argument (shape / alignment / null)
#define PSLD_ERR_LAUNCH 2   // hip launch error
#define PSLD_ERR_NUMERIC 3  // NaN in SDE coefficients (reference raises ValueError, psld.py:166-171)

void psld_set_error(const char* fmt, ...);

#define PSLD_CHECK_ARG(cond, ...)            \
    do {                                     \
        if (!(cond)) {                       \
            psld_set_error(__VA_ARGS__);     \
            return PSLD_ERR_ARG;             \
        }                                    \
    } while (0)

#define PSLD_CHECK_LAUNCH(name)                                                  \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            psld_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return PSLD_ERR_LAUNCH;                                              \
        }                                                                        \
    } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// wave64 reductions (CDNA wavefront = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float silu_f(float z) { return z / (1.0f + expf(-z)); }
// d/dz [z*sigmoid(z)] = s*(1 + z*(1-s))
__device__ __forceinline__ float dsilu_f(float z) {
    float s = 1.0f / (1.0f + expf(-z));
    return s * (1.0f + z * (1.0f - s));
}

// Counter-based dropout decision (splitmix64 finaliser over seed and element index): the forward
// and backward kernels re-derive the same mask without storing it.
__device__ __forceinline__ bool psld_dropout_keep(unsigned long long seed, unsigned long long idx, float p) {
    unsigned long long z = seed + (idx + 1ULL) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);  // 24-bit uniform in [0,1)
    return u >= p;
}
