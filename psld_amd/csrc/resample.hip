// FIR resampling (upfirdn2d) and the unused-but-inventoried fused_bias_act, for gfx950.
//
// Replaces the reference's two JIT-built CUDA ops:
//   op/upfirdn2d_kernel.cu:49-207 (kernels), :209-369 (host op), bound at op/upfirdn2d.cpp:12-22
//   op/fused_bias_act_kernel.cu:18-49, bound at op/fused_bias_act.cpp:11-20
// Semantics follow the CPU path the reference itself uses as ground truth
// (op/upfirdn2d.py:159-200): zero-insert upsample, pad/crop, correlate with the flipped
// kernel, decimate.  HBM-bound: each thread produces one float4 of channels (NHWC) or one
// pixel (NCHW) and reads its <= kh*kw taps through L1/L2 (neighbouring outputs share taps).
#include "common.h"
#include "psld_hip.h"

namespace {

constexpr int MAX_TAPS = 64;

struct Fir {
    float w[MAX_TAPS];  // already flipped: w[ky*kw+kx] = kernel[kh-1-ky][kw-1-kx]
    int kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0;
};

// out[oy][ox] = sum_{ky,kx} w[ky][kx] * U[oy*down_y + ky - pad_y0][ox*down_x + kx - pad_x0],
// U = zero-inserted upsample of the input (U[y*up][x*up] = in[y][x]), zero outside.
__global__ void upfirdn_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, const Fir f, int batch,
                                    int c, int in_h, int in_w, int out_h, int out_w, int accumulate) {
    const int cq = c >> 2;
    const long long total = (long long)batch * out_h * out_w * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        long long t = i / cq;
        const int ox = (int)(t % out_w);
        t /= out_w;
        const int oy = (int)(t % out_h);
        const int n = (int)(t / out_h);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // only taps that land on a real (not zero-inserted) sample: ky = ky0, ky0 + up, ...
        int ky0 = (f.pad_y0 - oy * f.down_y) % f.up_y;
        if (ky0 < 0) ky0 += f.up_y;
        int kx0 = (f.pad_x0 - ox * f.down_x) % f.up_x;
        if (kx0 < 0) kx0 += f.up_x;
        for (int ky = ky0; ky < f.kh; ky += f.up_y) {
            const int py = oy * f.down_y + ky - f.pad_y0;
            if (py < 0) continue;
            const int iy = py / f.up_y;
            if (iy >= in_h) break;
            for (int kx = kx0; kx < f.kw; kx += f.up_x) {
                const int px = ox * f.down_x + kx - f.pad_x0;
                if (px < 0) continue;
                const int ix = px / f.up_x;
                if (ix >= in_w) break;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long long)n * in_h + iy) * in_w + ix) * c + q * 4);
                acc += v * f.w[ky * f.kw + kx];
            }
        }
        float* op = y + (((long long)n * out_h + oy) * out_w + ox) * c + q * 4;
        if (accumulate) acc += *reinterpret_cast<const f32x4*>(op);
        *reinterpret_cast<f32x4*>(op) = acc;
    }
}

// Fast NHWC forms of the two resamplers the network uses with the 4x4 FIR (up_or_down_sampling.py:195-257): x2 up
// (2x2 contributing taps per output) and x2 down (all 16 taps); compile-time tap structure, one block row per output
// row so that no thread divides by a run-time extent.  grid = (ceil(out_w*cq / 256), batch*out_h).
template <int UP, int DOWN>
__global__ void upfirdn4_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, const Fir f, int c, int in_h,
                                     int in_w, int out_h, int out_w, int accumulate) {
    const int cq = c >> 2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= out_w * cq) return;
    const int ox = t / cq, q = t - ox * cq;
    const int n = blockIdx.y / out_h, oy = blockIdx.y - n * out_h;
    const float* xin = x + (long long)n * in_h * in_w * c + q * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int STEP = UP;                       // taps that land on real samples are UP apart
    int ky0 = 0, kx0 = 0;
    if (UP == 2) {
        ky0 = (f.pad_y0 - oy) & 1;
        kx0 = (f.pad_x0 - ox) & 1;
    }
#pragma unroll
    for (int a = 0; a < 4 / STEP; ++a) {
        const int ky = ky0 + a * STEP;
        const int py = oy * DOWN + ky - f.pad_y0;
        const int iy = UP == 2 ? py >> 1 : py;
        if (py < 0 || iy >= in_h) continue;
#pragma unroll
        for (int b = 0; b < 4 / STEP; ++b) {
            const int kx = kx0 + b * STEP;
            const int px = ox * DOWN + kx - f.pad_x0;
            const int ix = UP == 2 ? px >> 1 : px;
            if (px < 0 || ix >= in_w) continue;
            acc += *reinterpret_cast<const f32x4*>(xin + ((long long)iy * in_w + ix) * c) * f.w[ky * 4 + kx];
        }
    }
    float* op = y + (((long long)n * out_h + oy) * out_w + ox) * c + q * 4;
    if (accumulate) acc += *reinterpret_cast<const f32x4*>(op);
    *reinterpret_cast<f32x4*>(op) = acc;
}

// ---- 2 x 2 outputs per thread --------------------------------------------------------------------------------------
// x2 UP with a 4x4 FIR: output row 2i + r reads the two input rows i + m_r, i + m_r + 1 with the two taps ky = ((pad - r) &
// 1) + 2a, where m_r = (r + ((pad - r) & 1) - pad) / 2 - so the row pair (2i, 2i + 1) needs input rows i + m_0 .. and,
// when pad is even, one more (m_1 = m_0 + 1): a 3 x 3 (even pads) or 2 x 2 (odd pads) neighbourhood feeds four outputs:
// 9/4 loads per output instead of 4, and 4 stores per thread keep the write stream busy.  PY / PX = pad parity.
template <int PY, int PX>
__global__ void upfirdn4_up2_quad_kernel(const float* __restrict__ x, float* __restrict__ y, const Fir f, int c, int in_h,
                                         int in_w, int out_h, int out_w, int accumulate) {
    constexpr int NR = PY ? 2 : 3, NC = PX ? 2 : 3;
    const int cq = c >> 2;
    const int pairs_x = (out_w + 1) >> 1, pairs_y = (out_h + 1) >> 1;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pairs_x * cq) return;
    const int j = t / cq, q = t - j * cq;
    const int n = blockIdx.y / pairs_y, i = blockIdx.y - n * pairs_y;
    // m_0 for rows / columns (floor division: the numerators are even)
    const int my = (((f.pad_y0) & 1) - f.pad_y0) >> 1, mx = (((f.pad_x0) & 1) - f.pad_x0) >> 1;
    const float* xin = x + (long long)n * in_h * in_w * c + q * 4;
    f32x4 v[NR][NC];
#pragma unroll
    for (int a = 0; a < NR; ++a) {
        const int iy = i + my + a;
#pragma unroll
        for (int b = 0; b < NC; ++b) {
            const int ix = j + mx + b;
            const bool ok = iy >= 0 && iy < in_h && ix >= 0 && ix < in_w;
            v[a][b] = ok ? *reinterpret_cast<const f32x4*>(xin + ((long long)iy * in_w + ix) * c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oy = 2 * i + r;
        if (oy >= out_h) continue;
        const int ky0 = (f.pad_y0 - r) & 1;
        const int ry = PY ? 0 : r;                       // first source row of output row r inside v
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            const int ox = 2 * j + sx;
            if (ox >= out_w) continue;
            const int kx0 = (f.pad_x0 - sx) & 1;
            const int rx = PX ? 0 : sx;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc += v[ry + a][rx + b] * f.w[(ky0 + 2 * a) * 4 + kx0 + 2 * b];
            float* op = y + (((long long)n * out_h + oy) * out_w + ox) * c + q * 4;
            if (accumulate) acc += *reinterpret_cast<const f32x4*>(op);
            *reinterpret_cast<f32x4*>(op) = acc;
        }
    }
}

__global__ void upfirdn_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, const Fir f, int planes,
                                    int in_h, int in_w, int out_h, int out_w, int accumulate) {
    const long long total = (long long)planes * out_h * out_w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % out_w);
        long long t = i / out_w;
        const int oy = (int)(t % out_h);
        const long long pl = t / out_h;
        const float* xp = x + pl * in_h * in_w;
        float acc = 0.f;
        for (int ky = 0; ky < f.kh; ++ky) {
            const int py = oy * f.down_y + ky - f.pad_y0;
            if (py < 0 || py % f.up_y) continue;
            const int iy = py / f.up_y;
            if (iy >= in_h) continue;
            for (int kx = 0; kx < f.kw; ++kx) {
                const int px = ox * f.down_x + kx - f.pad_x0;
                if (px < 0 || px % f.up_x) continue;
                const int ix = px / f.up_x;
                if (ix >= in_w) continue;
                acc += xp[iy * in_w + ix] * f.w[ky * f.kw + kx];
            }
        }
        if (accumulate) acc += y[i];
        y[i] = acc;
    }
}

// grad = 0: y = act(x + b) * scale; grad = 1 (first derivative: x is the incoming gradient, ref the forward OUTPUT):
// y = (act == lrelu && ref <= 0 ? x * alpha : x) * scale; grad = 2 (second derivative): 0 - the switch of
// op/fused_bias_act_kernel.cu:35-47, cases 10 11 12 / 30 31 32.
__global__ void fused_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b, const float* __restrict__ ref,
                                      float* __restrict__ y, long long n, int size_b, int step_b, int act, int grad,
                                      float alpha, float scale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float v = x[i];
        if (b) v += b[(i / step_b) % size_b];
        const float r = ref ? ref[i] : 0.f;
        if (grad == 2) v = 0.f;
        else if (act == 3) v = ((grad == 1 ? r : v) > 0.f) ? v : v * alpha;
        y[i] = v * scale;
    }
}

}  // namespace

extern "C" int psld_upfirdn2d_f32(const float* x, float* y, int batch, int c, int in_h, int in_w,
                                  const float* kernel_host, int kh, int kw, int up_x, int up_y, int down_x,
                                  int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, int layout,
                                  int accumulate, hipStream_t stream) {
    PSLD_CHECK_ARG(x && y && kernel_host, "psld_upfirdn2d_f32: null pointer");
    PSLD_CHECK_ARG(kh >= 1 && kw >= 1 && kh * kw <= MAX_TAPS, "psld_upfirdn2d_f32: kernel %dx%d too large", kh, kw);
    PSLD_CHECK_ARG(up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1, "psld_upfirdn2d_f32: bad factors");
    const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;  // op/upfirdn2d_kernel.cu:237-240
    const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    PSLD_CHECK_ARG(out_h > 0 && out_w > 0, "psld_upfirdn2d_f32: empty output");
    Fir f;
    f.kh = kh; f.kw = kw; f.up_x = up_x; f.up_y = up_y; f.down_x = down_x; f.down_y = down_y;
    f.pad_x0 = pad_x0; f.pad_y0 = pad_y0;
    for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < kw; ++kx) f.w[ky * kw + kx] = kernel_host[(kh - 1 - ky) * kw + (kw - 1 - kx)];
    if (layout == 1) {
        PSLD_CHECK_ARG(c % 4 == 0, "psld_upfirdn2d_f32: NHWC needs C%%4==0 (C=%d)", c);
        const bool k4 = kh == 4 && kw == 4 && up_x == up_y && down_x == down_y && (long long)batch * out_h < 65536;
        if (k4 && up_x == 2 && down_x == 1) {
            // x2 up, 2 x 2 outputs per thread: 42.9 -> 34.8 us on 128x16x16x256 (a pure fill of the output: 20.6 us).  The
            // same form for x2 down (6 x 6 inputs -> 2 x 2 outputs, 9 loads per output instead of 16) measured SLOWER
            // (56.5 vs 47.4 us), and so did a branch-free one-output form with all sixteen loads issued up front (48.9 us): the
            // generic one-output kernel stays for x2 down.
            const dim3 grid((unsigned)cdiv((long long)((out_w + 1) / 2) * (c / 4), 256), (unsigned)(batch * ((out_h + 1) / 2)));
#define PSLD_FIR_UP(PY, PX)                                                                                              \
    hipLaunchKernelGGL((upfirdn4_up2_quad_kernel<PY, PX>), grid, dim3(256), 0, stream, x, y, f, c, in_h, in_w, out_h, out_w, \
                       accumulate)
            const int py = pad_y0 & 1, px = pad_x0 & 1;
            if (py && px) PSLD_FIR_UP(1, 1);
            else if (py) PSLD_FIR_UP(1, 0);
            else if (px) PSLD_FIR_UP(0, 1);
            else PSLD_FIR_UP(0, 0);
#undef PSLD_FIR_UP
            PSLD_CHECK_LAUNCH("psld_upfirdn2d_f32");
            return PSLD_OK;
        }
        if (k4 && ((up_x == 2 && down_x == 1) || (up_x == 1 && down_x == 2))) {
            const dim3 grid((unsigned)cdiv((long long)out_w * (c / 4), 256), (unsigned)(batch * out_h));
            if (up_x == 2)
                hipLaunchKernelGGL((upfirdn4_nhwc_kernel<2, 1>), grid, dim3(256), 0, stream, x, y, f, c, in_h, in_w, out_h,
                                   out_w, accumulate);
            else
                hipLaunchKernelGGL((upfirdn4_nhwc_kernel<1, 2>), grid, dim3(256), 0, stream, x, y, f, c, in_h, in_w, out_h,
                                   out_w, accumulate);
            PSLD_CHECK_LAUNCH("psld_upfirdn2d_f32");
            return PSLD_OK;
        }
        const long long total = (long long)batch * out_h * out_w * (c / 4);
        const int blocks = (int)min((long long)cdiv(total, 256), 256LL * 32);
        hipLaunchKernelGGL(upfirdn_nhwc_kernel, dim3(blocks), dim3(256), 0, stream, x, y, f, batch, c, in_h, in_w,
                           out_h, out_w, accumulate);
    } else {
        const long long total = (long long)batch * c * out_h * out_w;
        const int blocks = (int)min((long long)cdiv(total, 256), 256LL * 32);
        hipLaunchKernelGGL(upfirdn_nchw_kernel, dim3(blocks), dim3(256), 0, stream, x, y, f, batch * c, in_h, in_w,
                           out_h, out_w, accumulate);
    }
    PSLD_CHECK_LAUNCH("psld_upfirdn2d_f32");
    return PSLD_OK;
}

extern "C" int psld_fused_bias_act_grad_f32(const float* x, const float* b, const float* refer, float* y, long long n,
                                            int size_b, int step_b, int act, int grad, float alpha, float scale,
                                            hipStream_t stream) {
    PSLD_CHECK_ARG(x && y && n >= 0 && (!b || (size_b > 0 && step_b > 0)), "psld_fused_bias_act_f32: bad args");
    PSLD_CHECK_ARG(act == 1 || act == 3, "psld_fused_bias_act_f32: act must be 1 (linear) or 3 (lrelu)");
    PSLD_CHECK_ARG(grad >= 0 && grad <= 2, "psld_fused_bias_act_f32: grad must be 0, 1 or 2");
    if (n == 0) return PSLD_OK;
    const int blocks = (int)min((long long)cdiv(n, 256), 256LL * 32);
    hipLaunchKernelGGL(fused_bias_act_kernel, dim3(blocks), dim3(256), 0, stream, x, b, refer, y, n, size_b, step_b, act,
                       grad, alpha, scale);
    PSLD_CHECK_LAUNCH("psld_fused_bias_act_f32");
    return PSLD_OK;
}

extern "C" int psld_fused_bias_act_f32(const float* x, const float* b, float* y, long long n, int size_b,
                                       int step_b, int act, float alpha, float scale, hipStream_t stream) {
    return psld_fused_bias_act_grad_f32(x, b, nullptr, y, n, size_b, step_b, act, 0, alpha, scale, stream);
}
