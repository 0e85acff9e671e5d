// Per-step parameter maintenance over FLAT fp32 buffers: global grad-norm, clip, Adam and EMA for
// all 97.6 M parameters in one launch each (the reference loops over 749 tensors in Python:
// main/models/wrapper.py:82-89 (clip_grad_norm_, Adam.step), :128-155 (Adam/LambdaLR config),
// main/callbacks.py:57-64 (EMA)).  HBM-bound: Adam+EMA reads p,g,m,v,ema and writes p,m,v,ema
// = 9 x 4 B per parameter.
#include "common.h"
#include "psld_hip.h"

namespace {

#define GRID_STRIDE(i, n) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

constexpr int NORM_BLOCKS = 2048;

__global__ void sumsq_partial_kernel(const float* __restrict__ g, long long n, double* __restrict__ part) {
    __shared__ double red[4];
    double acc = 0.0;
    float local = 0.f;
    int cnt = 0;
    const long long n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? n / 4 : 0;
    GRID_STRIDE(i, n4) {
        const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
        local += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        if (++cnt == 8) { acc += (double)local; local = 0.f; cnt = 0; }
    }
    GRID_STRIDE(j, n - n4 * 4) {
        const float v = g[n4 * 4 + j];
        local += v * v;
    }
    acc += (double)local;
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void norm_final_kernel(const double* __restrict__ part, int nparts, double* __restrict__ out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc += part[i];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sqrt(red[0] + red[1] + red[2] + red[3]);
}

struct AdamArgs {
    // (1-beta) and (1-tau) are formed in double on the host and then rounded to f32, exactly as
    // torch does for python-float scalars (1 - 0.999 -> f32(0.001), not 1.0f - 0.999f).
    float beta2, eps, wd, step_size, inv_sqrt_bc2, max_norm, tau, omb1, omb2, omtau;
};

// torch.optim.Adam (single-tensor form): m.lerp_(g, 1-b1); v = v*b2 + (1-b2) g*g;
// denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m/denom.   Clip: g *= min(max_norm/(norm+1e-6), 1).
__global__ void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, float* __restrict__ ema, long long n,
                                const double* __restrict__ norm, AdamArgs a, float* __restrict__ g_mut,
                                const float* __restrict__ hyper_dev, const unsigned long long* __restrict__ err_word,
                                float* __restrict__ poison) {
    // A device-side error of this step's backward (today: a team GroupNorm backward that gave up waiting, norm_act.hip) must
    // not reach the parameters: the step becomes a no-op for p / m / v / ema and the loss scalar the caller logs turns NaN
    // (psld.py:166-171: fail loudly, never continue on bad numerics).  No host read: works inside a captured step.
    if (err_word && err_word[0] != 0ull) {
        if (poison && blockIdx.x == 0 && threadIdx.x == 0) poison[0] = __builtin_nanf("");
        return;
    }
    if (hyper_dev) {                 // captured training step: the two step-dependent scalars come from device memory
        a.step_size = hyper_dev[0];
        a.inv_sqrt_bc2 = hyper_dev[1];
    }
    float coef = 1.0f;
    if (a.max_norm > 0.f && norm) {
        const float c = a.max_norm / ((float)norm[0] + 1e-6f);
        coef = c < 1.0f ? c : 1.0f;
    }
    GRID_STRIDE(i, n) {
        float gi = g[i] * coef;
        if (g_mut) g_mut[i] = gi;
        float pi = p[i];
        if (a.wd != 0.f) gi += a.wd * pi;
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * a.omb1;
        vi = vi * a.beta2 + a.omb2 * gi * gi;
        const float denom = sqrtf(vi) * a.inv_sqrt_bc2 + a.eps;
        pi = pi - a.step_size * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
        if (ema) ema[i] = ema[i] * a.tau + pi * a.omtau;
    }
}

__global__ void ema_kernel(float* __restrict__ target, const float* __restrict__ src, long long n, float tau,
                           float omtau, const unsigned long long* __restrict__ err_word) {
    if (err_word && err_word[0] != 0ull) return;      // the step that would have moved src was refused (adam_ema_kernel)
    GRID_STRIDE(i, n) target[i] = target[i] * tau + src[i] * omtau;
}

}  // namespace

extern "C" int psld_grad_norm_f32(const float* g, long long n, double* norm_out, void* workspace, hipStream_t stream) {
    PSLD_CHECK_ARG(g && norm_out && workspace && n > 0, "psld_grad_norm_f32: bad args");
    double* part = reinterpret_cast<double*>(workspace);
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > NORM_BLOCKS) blocks = NORM_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, stream, g, n, part);
    PSLD_CHECK_LAUNCH("sumsq_partial_kernel");
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, stream, part, blocks, norm_out);
    PSLD_CHECK_LAUNCH("norm_final_kernel");
    return PSLD_OK;
}

extern "C" int psld_adam_ema_f32(float* p, const float* g, float* m, float* v, float* ema, long long n,
                                 const double* norm, double max_norm, double lr, double beta1, double beta2,
                                 double eps, double weight_decay, int step, double ema_tau, int write_clipped_grad,
                                 float* g_mut, const float* hyper_dev, const unsigned long long* err_word, float* poison,
                                 hipStream_t stream) {
    PSLD_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "psld_adam_ema_f32: bad args");
    PSLD_CHECK_ARG(max_norm <= 0.f || norm, "psld_adam_ema_f32: clipping needs the norm buffer");
    AdamArgs a;
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    a.beta2 = (float)beta2; a.eps = (float)eps; a.wd = (float)weight_decay;
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    a.step_size = (float)(lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.max_norm = (float)max_norm; a.tau = (float)ema_tau; a.omtau = (float)(1.0 - ema_tau);
    long long b = (n + 1023) / 1024;
    if (b > 256 * 16) b = 256 * 16;
    hipLaunchKernelGGL(adam_ema_kernel, dim3((int)b), dim3(256), 0, stream, p, g, m, v, ema, n, norm, a,
                       write_clipped_grad ? g_mut : nullptr, hyper_dev, err_word, poison);
    PSLD_CHECK_LAUNCH("adam_ema_kernel");
    return PSLD_OK;
}

extern "C" void psld_adam_step_scalars(double lr, double beta1, double beta2, int step, float* out2_host) {
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    out2_host[0] = (float)(lr / bc1);
    out2_host[1] = (float)(1.0 / sqrt(bc2));
}

// The same two scalars written to DEVICE memory by a kernel that takes them by value: the captured training step reads
// them from a 2-float buffer, and a pinned host staging buffer rewritten every step would race with its own asynchronous
// copies once the host runs several replays ahead.
__global__ void set2_kernel(float* __restrict__ dst, float a, float b) {
    if (threadIdx.x == 0) { dst[0] = a; dst[1] = b; }
}
extern "C" int psld_adam_step_scalars_dev(double lr, double beta1, double beta2, int step, float* out2_dev, hipStream_t stream) {
    PSLD_CHECK_ARG(out2_dev && step >= 1, "psld_adam_step_scalars_dev: bad args");
    float h[2];
    psld_adam_step_scalars(lr, beta1, beta2, step, h);
    hipLaunchKernelGGL(set2_kernel, dim3(1), dim3(64), 0, stream, out2_dev, h[0], h[1]);
    PSLD_CHECK_LAUNCH("set2_kernel");
    return PSLD_OK;
}

extern "C" int psld_ema_f32(float* target, const float* src, long long n, double tau, const unsigned long long* err_word,
                            hipStream_t stream) {
    PSLD_CHECK_ARG(target && src && n > 0, "psld_ema_f32: bad args");
    long long b = (n + 1023) / 1024;
    if (b > 256 * 16) b = 256 * 16;
    hipLaunchKernelGGL(ema_kernel, dim3((int)b), dim3(256), 0, stream, target, src, n, (float)tau,
                       (float)(1.0 - tau), err_word);
    PSLD_CHECK_LAUNCH("ema_kernel");
    return PSLD_OK;
}
