// Test helper (tests/test_fullsize_gpu.py: the co-residency test; tools/rccl_occupancy.py).
// A stand-in for RCCL's channel kernel on a 1-GPU box: `blocks` workgroups of `threads`
// threads that hold `lds_bytes` of LDS each and spin for `usec` microseconds - what a ring all-reduce does to the CUs it sits
// on for the time the bytes are on the wire - and touch `bytes` of memory on the way (the collective's own read + write of the
// bucket).  NOT part of the product library: __graft_entry__.build() compiles it to tests/helpers/libcuhog.so.
#include <hip/hip_runtime.h>

extern "C" __global__ void cu_hog_kernel(float* buf, long long n, long long ticks) {
    extern __shared__ float lds[];
    const long long t0 = wall_clock64();       // constant-rate counter (100 MHz)
    lds[threadIdx.x] = (float)threadIdx.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float acc = 0.f;
    while (wall_clock64() - t0 < ticks) {
        if (n > 0) {                            // a slow walk over the bucket: reads and rewrites it in place
            if (i >= n) i -= n;
            const float v = buf[i];
            buf[i] = v;                         // same value back: the gradient is untouched
            acc += v;
            i += stride;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 123.456f) lds[0] = acc;          // keep the loads
    if (lds[threadIdx.x] < 0.f) buf[0] = lds[0];
}

extern "C" int cu_hog(float* buf, long long n, int blocks, int threads, int lds_bytes, double usec, hipStream_t stream) {
    const long long ticks = (long long)(usec * 100.0);
    hipLaunchKernelGGL(cu_hog_kernel, dim3(blocks), dim3(threads), lds_bytes, stream, buf, n, ticks);
    return (int)hipGetLastError();
}
