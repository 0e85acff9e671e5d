// Sustained dense MFMA rate of the chip, registers only (no LDS / global traffic in the loop): the practical ceiling
// a compute kernel can reach under the board's power limit.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip
// -o tools/mfma_peak ; run: tools/mfma_peak [waves_per_simd] [milliseconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 1e-3f + i); b[i] = (__bf16)(seed - i * 0.5f); }
    if constexpr (KIND == 0) {          // v_mfma_f32_16x16x32_bf16, 16 independent accumulators
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if constexpr (KIND == 1) {   // v_mfma_f32_32x32x16_bf16, 8 independent accumulators
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if constexpr (KIND == 3) {   // dconv_kernel's register pattern: 4x4 accumulators, 4x3 A and 4x3 B limb
                                        // fragments holding pseudo-random bf16 data (operand bits toggle between MFMAs)
        bf16x8 fa[4][3], fb[4][3];
        unsigned h = 0x9E3779B9u * (threadIdx.x + 1) + (unsigned)(seed * 1000.f);
        for (int i = 0; i < 4; ++i)
            for (int l = 0; l < 3; ++l)
                for (int e = 0; e < 8; ++e) {
                    h = h * 1664525u + 1013904223u;
                    fa[i][l][e] = (__bf16)((float)(int)(h >> 16) * (1.f / 65536.f) - 0.5f);
                    h = h * 1664525u + 1013904223u;
                    fb[i][l][e] = (__bf16)((float)(int)(h >> 16) * (1.f / 65536.f) - 0.5f);
                }
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mb][PA[t]], fb[nb][PB[t]], acc[mb][nb], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if constexpr (KIND == 4) {   // 32x32x16 with the same pseudo-random operand rotation: 2x2 accumulators of a 64x64 wave tile
        bf16x8 fa[2][3], fb[2][3];
        unsigned h = 0x9E3779B9u * (threadIdx.x + 1) + (unsigned)(seed * 1000.f);
        for (int i = 0; i < 2; ++i)
            for (int l = 0; l < 3; ++l)
                for (int e = 0; e < 8; ++e) {
                    h = h * 1664525u + 1013904223u;
                    fa[i][l][e] = (__bf16)((float)(int)(h >> 16) * (1.f / 65536.f) - 0.5f);
                    h = h * 1664525u + 1013904223u;
                    fb[i][l][e] = (__bf16)((float)(int)(h >> 16) * (1.f / 65536.f) - 0.5f);
                }
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mb][PA[t]], fb[nb][PB[t]], acc[mb][nb], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {                            // v_mfma_f32_32x32x2_f32, 8 independent accumulators
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        const float fa = seed + threadIdx.x, fb = seed - 1.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

template <int KIND>
void run(const char* name, double flop_per_mfma, int mfma_per_iter, int wps, double target_ms, float* out) {
    const int blocks = 256 * wps;            // one 4-wave block per CU and wave slot
    int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * 4 * iters * mfma_per_iter * flop_per_mfma;
        if (rep == 2) printf("%-28s waves/SIMD %d  %8.2f ms  %8.1f TFLOP/s\n", name, wps, ms, fl / ms / 1e9);
        iters = (int)(iters * target_ms / (ms > 0.01f ? ms : 0.01f));
        if (iters < 100) iters = 100;
    }
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const double ms = argc > 2 ? atof(argv[2]) : 50.0;
    float* out;
    (void)hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
    run<0>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, 16, wps, ms, out);
    run<1>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, 8, wps, ms, out);
    run<2>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, 8, wps, ms, out);
    run<3>("16x16x32 bf16, dconv pattern", 2.0 * 16 * 16 * 32, 96, wps, ms, out);
    run<4>("32x32x16 bf16, random data", 2.0 * 32 * 32 * 16, 24, wps, ms, out);
    (void)hipFree(out);
    return 0;
}
