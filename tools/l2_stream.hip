// How fast can the waves of a CU stream L2-RESIDENT fragments, 1 KB per wave-instruction (global_load_dwordx4, the weight-
// fragment stream of the Winograd / direct limb kernels)?  Every wave walks its own contiguous slice of a small buffer
// that all workgroups share (the U slice of one channel tile: 3 MB, L2-resident on every XCD), DEPTH loads in flight,
// optionally with MFMAs between the loads (12 per 3 loads: the Winograd kernel's ratio).
// Build: hipcc -O3 --offload-arch=gfx950 tools/l2_stream.hip -o tools/l2_stream ; run: tools/l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int DEPTH, int MFMA>
__global__ void __launch_bounds__(512) stream_kernel(const u32x4* __restrict__ buf, long long slice_u4, int iters, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave w of every workgroup streams slice w (as wave w of every Winograd workgroup streams channel block w)
    const u32x4* p = buf + (long long)wave * slice_u4 + lane;
    u32x4 ring[DEPTH];
    f32x4 acc[4] = {};
    u32x4 x = {0, 0, 0, 0};
    long long pos = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { ring[d] = p[pos]; pos += 64; if (pos >= slice_u4) pos = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const u32x4 v = ring[d];
            ring[d] = p[pos];
            pos += 64;
            if (pos >= slice_u4) pos = 0;
            if constexpr (MFMA > 0) {
#pragma unroll
                for (int m = 0; m < MFMA; ++m)
                    acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, v), acc[m & 3], 0, 0, 0);
            } else {
                x ^= v;
            }
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) x ^= ring[d];
    out[blockIdx.x * 512 + threadIdx.x] = x[0] ^ x[1] ^ x[2] ^ x[3] ^ __float_as_uint(acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]);
}

template <int DEPTH, int MFMA>
void run(const u32x4* buf, long long slice_u4, unsigned* out, int wgs, int threads) {
    const int iters = 4096 / DEPTH * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<DEPTH, MFMA>), dim3(wgs), dim3(threads), 0, 0, buf, slice_u4, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) {
            const double bytes = (double)wgs * (threads / 64) * (double)iters * DEPTH * 1024.0;
            printf("waves/CU %2d  in flight %2d KB/wave  MFMAs per load %d : %7.1f us  %6.2f TB/s chip  %5.1f GB/s per CU  %5.1f B/clk/CU at 1.9 GHz\n",
                   threads / 64, DEPTH, MFMA, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / wgs, bytes / (ms * 1e-3) / wgs / 1.9e9);
        }
    }
}

int main() {
    const long long slice_u4 = 384 * 1024 / 16;          // 384 KB per wave slice: 8 slices = 3 MB (one 128-channel U slice at 256 input channels)
    u32x4* buf; unsigned* out;
    hipMalloc(&buf, 16 * slice_u4 * 16);
    hipMalloc(&out, 256 * 1024 * 4);
    std::vector<unsigned> h(16 * slice_u4 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
    hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("loads only\n");
    run<3, 0>(buf, slice_u4, out, 256, 512); run<6, 0>(buf, slice_u4, out, 256, 512); run<9, 0>(buf, slice_u4, out, 256, 512);
    run<12, 0>(buf, slice_u4, out, 256, 512); run<24, 0>(buf, slice_u4, out, 256, 512);
    run<6, 0>(buf, slice_u4, out, 256, 256); run<12, 0>(buf, slice_u4, out, 256, 256); run<24, 0>(buf, slice_u4, out, 256, 256);
    printf("with 4 MFMAs per 1 KB load (the Winograd kernel's 12 per 3 KB)\n");
    run<6, 4>(buf, slice_u4, out, 256, 512); run<9, 4>(buf, slice_u4, out, 256, 512); run<12, 4>(buf, slice_u4, out, 256, 512);
    run<6, 4>(buf, slice_u4, out, 256, 256); run<12, 4>(buf, slice_u4, out, 256, 256);
    return 0;
}
