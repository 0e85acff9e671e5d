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

// One-time set-up that is PER DEVICE (hipFuncSetAttribute, occupancy and CU-count queries): a slot per device of the process,
// so that a second GPU driven from the same process is configured too instead of inheriting the first one's "done" flag.
constexpr int PSLD_MAX_DEVICES = 32;
static inline int psld_device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= PSLD_MAX_DEVICES) d = 0;
    return d;
}
struct PsldPerDeviceFlag {
    bool done[PSLD_MAX_DEVICES] = {};
    bool& here() { return done[psld_device_slot()]; }
};

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

// sigmoid through v_exp_f32 / v_rcp_f32 (1 ulp each; e^-z = 2^(-z log2 e)): the IEEE expf + division sequence
// it replaces made the GroupNorm+SiLU kernels VALU-bound instead of HBM-bound.  z -> -inf gives 1/(1+inf) = 0.
__device__ __forceinline__ float sigmoid_f(float z) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}
__device__ __forceinline__ float silu_f(float z) { return z * sigmoid_f(z); }
// d/dz [z*sigmoid(z)] = s*(1 + z*(1-s))
__device__ __forceinline__ float dsilu_f(float z) {
    const float s = sigmoid_f(z);
    return s * (1.0f + z * (1.0f - s));
}

// Counter-based dropout decision: a 32-bit PCG-style permutation of (element index, seed) — the forward and
// backward kernels re-derive the same mask without storing it.  32-bit integer ops only: the 64-bit
// splitmix used at first cost 12 % of gn_apply's bandwidth.
__device__ __forceinline__ bool psld_dropout_keep(unsigned long long seed, unsigned long long idx, float p) {
    unsigned int s = (unsigned int)idx ^ (unsigned int)(idx >> 32) * 0x85EBCA6Bu;
    s = s * 747796405u + ((unsigned int)seed ^ (unsigned int)(seed >> 32)) * 2891336453u + 2891336453u;
    unsigned int w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    w = (w >> 22u) ^ w;
    const float u = (float)(w >> 8) * (1.0f / 16777216.0f);  // 24-bit uniform in [0,1)
    return u >= p;
}
