// PSLD SDE kernels: perturbation kernel p(z_t | x_0), HSM loss, Euler-Maruyama reverse step.
// Compiled with -ffp-contract=off so that the f32/f64 operation order of the reference's eager
// elementwise chains is reproduced without fused multiply-adds.
//
// Reference: main/models/sde/psld.py:38-44 (b_t, beta_t), :62-84 (_mean), :86-152 (_cov),
// :154-186 (get_coeff), :188-220 (get_inv_coeff), :230-260 (get_score), :262-287 (perturb_data),
// :330-364 (sde / reverse_sde); main/losses.py:94-130; main/samplers/sde.py:16-36.
// All public tensors here are NCHW like the reference ([B, 2C, H, W] state = [x | m]).
#include "common.h"
#include "psld_hip.h"

namespace {

constexpr int COEFF_STRIDE = 12;

#define GRID_STRIDE(i, n) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

inline int grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

// psld.py:86-152 with the numerical_eps of :152 on the diagonal
__device__ void psld_cov(const psld_sde_params_t& p, double xx_0, double mm_0, double b, double& xx, double& xm,
                         double& mm) {
    const double nu = p.nu, ga = p.gamma, mi = p.m_inv, m = 1.0 / p.m_inv;
    const double lam = (nu + ga) / 2;
    const double b2 = b * b;
    const double sc = exp(-lam * b), isc = exp(lam * b);
    xx = (mi / 4 * b2 * xx_0 + mi * mi / 4 * b2 * mm_0 + (nu - ga) / 2 * b * xx_0 + (-mi / 2) * b2 +
          (ga - nu) / 2 * b + (isc - 1) + xx_0) * sc + p.numerical_eps;
    xm = ((ga - nu) / 8 * b2 * xx_0 + mi * (ga - nu) / 8 * b2 * mm_0 + (-1.0 / 2) * b * xx_0 + mi / 2 * b * mm_0 +
          (nu - ga) / 4 * b2) * sc;
    mm = (1.0 / 4 * b2 * xx_0 + mi / 4 * b2 * mm_0 + (ga - nu) / 2 * b * mm_0 + (-1.0 / 2) * b2 +
          m * (nu - ga) / 2 * b + m * (isc - 1) + mm_0) * sc + p.numerical_eps;
}

__global__ void perturb_coeffs_kernel(const double* __restrict__ t, int batch, const psld_sde_params_t p,
                                      double xx_0, double mm_0, double* __restrict__ out, int* __restrict__ nan_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const double tt = t[i];
    const double b = p.beta_0 * tt + 0.5 * (tt * tt) * (p.beta_1 - p.beta_0);  // psld.py:42-44
    const double sf = exp(-((p.nu + p.gamma) / 4) * b);                        // psld.py:65-67
    double xx, xm, mm;
    psld_cov(p, xx_0, mm_0, b, xx, xm, mm);
    double c11, c12, c21, c22;
    if (p.decomp_lower) {  // psld.py:160-164
        c11 = sqrt(xx);
        c12 = 0.0;
        c21 = xm / c11;
        c22 = sqrt(mm - c21 * c21);
    } else {  // psld.py:174-178
        c22 = sqrt(mm);
        c12 = xm / c22;
        c11 = sqrt(xx - c12 * c12);
        c21 = 0.0;
    }
    if (isnan(c11) || isnan(c12) || isnan(c21) || isnan(c22)) atomicExch(nan_flag, 1);
    double* o = out + (long long)i * COEFF_STRIDE;
    o[0] = b; o[1] = sf; o[2] = 0; o[3] = 0;
    o[4] = c11; o[5] = c12; o[6] = c21; o[7] = c22;
    o[8] = xx; o[9] = xm; o[10] = mm; o[11] = 0;
}

// one thread per FOUR consecutive pixels of one (b, c) plane of the x half (hw % 4 == 0: 16-byte accesses); writes
// both halves.  VEC = 1 is the element-wise form for odd plane sizes.
template <int VEC>
__global__ void perturb_kernel(const float* __restrict__ x0, const float* __restrict__ m0,
                               const float* __restrict__ eps, const double* __restrict__ coeffs,
                               const psld_sde_params_t p, int batch, int c, int hw, float* __restrict__ z,
                               double* __restrict__ u, double* __restrict__ mu_out) {
    typedef float fv __attribute__((ext_vector_type(VEC)));
    typedef double dv __attribute__((ext_vector_type(VEC)));
    const long long n = (long long)batch * c * hw / VEC;
    const long long plane = (long long)c * hw / VEC;
    const double A1 = (p.nu - p.gamma) / 4, A2 = (p.gamma - p.nu) * (p.gamma - p.nu) / 8;
    const double C1 = -0.5, C2 = (p.gamma - p.nu) / 4;
    GRID_STRIDE(i, n) {
        const int b = (int)(i / plane);
        const long long r = i - (long long)b * plane;  // offset inside the half (in VEC units)
        const double* k = coeffs + (long long)b * COEFF_STRIDE;
        const double bt = k[0], sf = k[1];
        const fv xf = reinterpret_cast<const fv*>(x0)[i];
        fv mf = 0.f;
        if (m0) mf = reinterpret_cast<const fv*>(m0)[i];
        const long long ox = (long long)b * 2 * plane + r, om = ox + plane;
        const fv exf = reinterpret_cast<const fv*>(eps)[ox], emf = reinterpret_cast<const fv*>(eps)[om];
        fv zx, zm;
        dv ux, um, mx, mm;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const double xv = (double)xf[e], mv = (double)mf[e];
            // psld.py:76-83
            const double mu_x = (A1 * xv * bt + A2 * mv * bt + xv) * sf;
            const double mu_m = (C1 * xv * bt + C2 * mv * bt + mv) * sf;
            const double ex = (double)exf[e], em = (double)emf[e];
            // psld.py:277-283
            const double nx = k[4] * ex + k[5] * em;
            const double nm = k[6] * ex + k[7] * em;
            ux[e] = mu_x + nx; um[e] = mu_m + nm;
            mx[e] = mu_x; mm[e] = mu_m;
            zx[e] = (float)ux[e]; zm[e] = (float)um[e];
        }
        if (z) { reinterpret_cast<fv*>(z)[ox] = zx; reinterpret_cast<fv*>(z)[om] = zm; }
        if (u) { reinterpret_cast<dv*>(u)[ox] = ux; reinterpret_cast<dv*>(u)[om] = um; }
        if (mu_out) { reinterpret_cast<dv*>(mu_out)[ox] = mx; reinterpret_cast<dv*>(mu_out)[om] = mm; }
    }
}

// ---- loss ---------------------------------------------------------------------------------------
__global__ void sqerr_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                     double* __restrict__ part, float* __restrict__ grad, float gscale) {
    __shared__ double red[4];
    double acc = 0.0;
    float local = 0.f;
    int cnt = 0;
    // 16 bytes per lane (n4 float4 items), then the < 4 leftover elements
    const long long n4 = n >> 2;
    GRID_STRIDE(i, n4) {
        const f32x4 va = reinterpret_cast<const f32x4*>(a)[i], vb = reinterpret_cast<const f32x4*>(b)[i];
        const f32x4 d = va - vb;
        local += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        if (grad) reinterpret_cast<f32x4*>(grad)[i] = d * (-gscale);   // d/d(b) of (a-b)^2 * (gscale/2)
        if (++cnt == 4) { acc += (double)local; local = 0.f; cnt = 0; }
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        const float d = a[i] - b[i];
        local += d * d;
        if (grad) grad[i] = -gscale * d;
    }
    acc += (double)local;
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sqerr_final_kernel(const double* __restrict__ part, int nparts, double denom, float* __restrict__ loss) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) acc += part[i];
    acc = wave_sum_d(acc);
    if (threadIdx.x == 0) loss[0] = (float)(acc / denom);
}

// ---- reverse SDE / Euler-Maruyama ---------------------------------------------------------------
struct RevOut {
    double fbx, fbm, gx, gm;
};

__device__ __forceinline__ RevOut reverse_terms(const psld_em_coeffs_t& k, double xv, double mv, float ex, float em,
                                                bool have_ex) {
    // psld.py:336-337 drift, :339-340 diffusion
    const double fx = 0.5 * k.beta * (k.m_inv * mv - k.gamma * xv);
    const double fm = 0.5 * k.beta * (-k.nu * mv - xv);
    const double gx = sqrt(k.beta * k.gamma);
    const double gm = sqrt(k.beta * k.m * k.nu);
    // psld.py:240-259 score in f32 with f32-cast coefficients
    float sx, sm;
    if (k.score_mode == 1) {         // score_m, lower: eps is the momentum part only
        sx = 0.f;
        sm = -k.c22 * em;
    } else if (k.score_mode == 2) {  // score_x, upper
        sx = -k.c11 * ex;
        sm = 0.f;
    } else {
        sx = -k.c11 * ex - k.c12 * em;
        sm = -k.c21 * ex - k.c22 * em;
    }
    (void)have_ex;
    if (k.probability_flow) { sx = 0.5f * sx; sm = 0.5f * sm; }
    RevOut o;
    o.fbx = -fx + (gx * gx) * (double)sx;   // psld.py:360
    o.fbm = -fm + (gm * gm) * (double)sm;
    o.gx = k.probability_flow ? 0.0 : gx;
    o.gm = k.probability_flow ? 0.0 : gm;
    return o;
}

__device__ __forceinline__ void fetch_eps(const float* eps, const psld_em_coeffs_t& k, int b, long long r, int c, int hw,
                                          float& ex, float& em) {
    if (k.score_mode == 0) {
        const long long o = (long long)b * 2 * c * hw + r;
        ex = eps[o];
        em = eps[o + (long long)c * hw];
    } else {
        const float v = eps[(long long)b * c * hw + r];
        ex = v;
        em = v;
    }
}

__global__ void em_step_kernel(double* __restrict__ x, const float* __restrict__ eps, const double* __restrict__ z,
                               const psld_em_coeffs_t k, int batch, int c, int hw, float* __restrict__ xf) {
    const long long n = (long long)batch * c * hw;
    const double sdt = sqrt(k.dt);
    GRID_STRIDE(i, n) {
        const int b = (int)(i / ((long long)c * hw));
        const long long r = i - (long long)b * c * hw;
        const long long ox = (long long)b * 2 * c * hw + r, om = ox + (long long)c * hw;
        float ex, em;
        fetch_eps(eps, k, b, r, c, hw, ex, em);
        const double xv = x[ox], mv = x[om];
        const RevOut o = reverse_terms(k, xv, mv, ex, em, true);
        double nx = xv + o.fbx * k.dt;   // samplers/sde.py:23
        double nm = mv + o.fbm * k.dt;
        if (z) {                          // samplers/sde.py:24-25
            nx = nx + o.gx * sdt * z[ox];
            nm = nm + o.gm * sdt * z[om];
        }
        x[ox] = nx;
        x[om] = nm;
        if (xf) { xf[ox] = (float)nx; xf[om] = (float)nm; }
    }
}

__global__ void reverse_sde_kernel(const double* __restrict__ x, const float* __restrict__ eps,
                                   const psld_em_coeffs_t k, int batch, int c, int hw, double* __restrict__ fbar,
                                   double* __restrict__ gbar) {
    const long long n = (long long)batch * c * hw;
    GRID_STRIDE(i, n) {
        const int b = (int)(i / ((long long)c * hw));
        const long long r = i - (long long)b * c * hw;
        const long long ox = (long long)b * 2 * c * hw + r, om = ox + (long long)c * hw;
        float ex, em;
        fetch_eps(eps, k, b, r, c, hw, ex, em);
        const RevOut o = reverse_terms(k, x[ox], x[om], ex, em, true);
        fbar[ox] = o.fbx; fbar[om] = o.fbm;
        if (gbar) { gbar[ox] = o.gx; gbar[om] = o.gm; }
    }
}

// Per-sample times (psld.py:330-364 take t[B]): every thread derives its sample's scalars from t[b] on the device -
// no host read of t.  beta_t (psld.py:38-40), _cov (:86-152) and get_inv_coeff (:188-220) in f64, the inverse
// coefficients cast to f32 before they touch eps (:253-258) exactly like the one-time path's host scalars.
__global__ void reverse_sde_rows_kernel(const double* __restrict__ x, const float* __restrict__ eps,
                                        const double* __restrict__ t_rev, const psld_sde_params_t p, double xx_0,
                                        double mm_0, int score_mode, int probability_flow, int forward_only,
                                        int c, int hw, double* __restrict__ fbar, double* __restrict__ gbar,
                                        int* __restrict__ nan_flag) {
    const int b = blockIdx.y;
    const double tt = t_rev[b];
    psld_em_coeffs_t k;
    k.beta = p.beta_0 + tt * (p.beta_1 - p.beta_0);
    k.m_inv = p.m_inv; k.gamma = p.gamma; k.nu = p.nu; k.m = 1.0 / p.m_inv;
    k.dt = 0.0; k.score_mode = score_mode; k.probability_flow = probability_flow;
    k.c11 = k.c12 = k.c21 = k.c22 = 0.f;
    if (!forward_only) {
        const double bt = p.beta_0 * tt + 0.5 * (tt * tt) * (p.beta_1 - p.beta_0);
        double xx, xm, mm;
        psld_cov(p, xx_0, mm_0, bt, xx, xm, mm);
        const double det = xx * mm - xm * xm;
        double c11, c12, c21, c22;
        if (p.decomp_lower) {   // psld.py:194-199
            c11 = sqrt(1 / xx); c12 = -xm / (sqrt(xx) * sqrt(det)); c21 = 0.0; c22 = sqrt(xx / det);
        } else {                // psld.py:207-212
            c11 = sqrt(mm / det); c12 = 0.0; c21 = -xm / (sqrt(mm) * sqrt(det)); c22 = sqrt(1 / mm);
        }
        if ((isnan(c11) || isnan(c12) || isnan(c21) || isnan(c22)) && threadIdx.x == 0 && blockIdx.x == 0)
            atomicExch(nan_flag, 1);
        k.c11 = (float)c11; k.c12 = (float)c12; k.c21 = (float)c21; k.c22 = (float)c22;
    }
    const long long n = (long long)c * hw;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
        const long long ox = (long long)b * 2 * c * hw + r, om = ox + (long long)c * hw;
        float ex = 0.f, em = 0.f;
        if (!forward_only) fetch_eps(eps, k, b, r, c, hw, ex, em);
        const RevOut o = reverse_terms(k, x[ox], x[om], ex, em, true);
        if (forward_only) {     // sde(): (f, g) themselves; -(-f + 0) is exact
            fbar[ox] = -o.fbx; fbar[om] = -o.fbm;
        } else {
            fbar[ox] = o.fbx; fbar[om] = o.fbm;
        }
        if (gbar) { gbar[ox] = o.gx; gbar[om] = o.gm; }
    }
}

// ---- symmetric-splitting sampler (samplers/sde.py:227-370) ----------------------------------------
// analytic half step: u' = M u + L z with host-computed 2x2 mean matrix (incl. exp scaling) and
// Cholesky factor of the transition covariance (sde.py:236-311)
__global__ void sscs_analytic_kernel(double* __restrict__ x, const double* __restrict__ z, const psld_sscs_coeffs_t k,
                                     int batch, int c, int hw, float* __restrict__ xf) {
    const long long n = (long long)batch * c * hw;
    GRID_STRIDE(i, n) {
        const int b = (int)(i / ((long long)c * hw));
        const long long r = i - (long long)b * c * hw;
        const long long ox = (long long)b * 2 * c * hw + r, om = ox + (long long)c * hw;
        const double xv = x[ox], mv = x[om];
        const double zx = z[ox], zm = z[om];
        const double nx = (k.a_xx * xv + k.a_xm * mv) + (k.c11 * zx + k.c12 * zm);
        const double nm = (k.a_mx * xv + k.a_mm * mv) + (k.c21 * zx + k.c22 * zm);
        x[ox] = nx;
        x[om] = nm;
        if (xf) { xf[ox] = (float)nx; xf[om] = (float)nm; }
    }
}

// score step (sde.py:313-329): x += dt*gamma*beta*(score_x + x);  m += dt*m*nu*beta*(score_m + m_inv*m)
__global__ void sscs_score_kernel(double* __restrict__ x, const float* __restrict__ eps, const psld_em_coeffs_t k,
                                  int batch, int c, int hw) {
    const long long n = (long long)batch * c * hw;
    GRID_STRIDE(i, n) {
        const int b = (int)(i / ((long long)c * hw));
        const long long r = i - (long long)b * c * hw;
        const long long ox = (long long)b * 2 * c * hw + r, om = ox + (long long)c * hw;
        float ex, em;
        fetch_eps(eps, k, b, r, c, hw, ex, em);
        float sx, sm;
        if (k.score_mode == 1) { sx = 0.f; sm = -k.c22 * em; }
        else if (k.score_mode == 2) { sx = -k.c11 * ex; sm = 0.f; }
        else { sx = -k.c11 * ex - k.c12 * em; sm = -k.c21 * ex - k.c22 * em; }
        const double xv = x[ox], mv = x[om];
        x[ox] = xv + k.dt * k.gamma * k.beta * ((double)sx + xv);
        x[om] = mv + k.dt * k.m * k.nu * k.beta * ((double)sm + k.m_inv * mv);
    }
}

// ---- adaptive RK45 building blocks (black-box probability-flow ODE sampler, samplers/ode.py) ---------
struct RkArgs {
    const double* v[8];
    double c[8];
    int nv;
};
// out = base + sum_j c_j v_j  (base may be null); optional f32 copy for the next network call
__global__ void rk_lincomb_kernel(double* __restrict__ out, const double* __restrict__ base, const RkArgs a,
                                  long long n, float* __restrict__ out32) {
    GRID_STRIDE(i, n) {
        double acc = base ? base[i] : 0.0;
#pragma unroll 8
        for (int j = 0; j < a.nv; ++j) acc += a.c[j] * a.v[j][i];
        out[i] = acc;
        if (out32) out32[i] = (float)acc;
    }
}
// partial sums of ((sum_j c_j v_j) / (atol + rtol * max(|p|,|q|)))^2
__global__ void rk_scaled_sq_kernel(const RkArgs a, const double* __restrict__ p, const double* __restrict__ q,
                                    double atol, double rtol, long long n, double* __restrict__ part) {
    __shared__ double red[4];
    double acc = 0.0;
    GRID_STRIDE(i, n) {
        double e = 0.0;
#pragma unroll 8
        for (int j = 0; j < a.nv; ++j) e += a.c[j] * a.v[j][i];
        const double sc = atol + rtol * fmax(fabs(p[i]), fabs(q[i]));
        const double r = e / sc;
        acc += r * r;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void rk_final_kernel(const double* __restrict__ part, int nparts, double* __restrict__ out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) acc += part[i];
    acc = wave_sum_d(acc);
    if (threadIdx.x == 0) out[0] = acc;
}

// ---- VP-SDE baseline (main/models/sde/vpsde.py) ---------------------------------------------------
// x_t = exp(lmc) * x0 + sqrt(1 - exp(2 lmc)) * eps, lmc = -0.25 t^2 (b1-b0) - 0.5 t b0   (vpsde.py:29-37, 72-83)
__global__ void vp_perturb_kernel(const float* __restrict__ x0, const float* __restrict__ eps,
                                  const double* __restrict__ t, double beta0, double beta1, int batch, long long per,
                                  float* __restrict__ z, double* __restrict__ u) {
    const long long n = (long long)batch * per;
    GRID_STRIDE(i, n) {
        const int b = (int)(i / per);
        const double tt = t[b];
        const double lmc = -0.25 * (tt * tt) * (beta1 - beta0) - 0.5 * tt * beta0;
        const double mean = exp(lmc) * (double)x0[i];
        const double sd = sqrt(1.0 - exp(2.0 * lmc));
        const double v = mean + (double)eps[i] * sd;
        if (z) z[i] = (float)v;
        if (u) u[i] = v;
    }
}
// reverse drift f_bar = 0.5 beta x + beta * score, score = -eps/std (f64), x0.5 for probability flow
// (vpsde.py:26-27, 39-66); mode 0: write f_bar (and g_bar), mode 1: Euler-Maruyama update of x in place
__global__ void vp_reverse_kernel(double* __restrict__ x, const float* __restrict__ eps, const double* __restrict__ z,
                                  double beta, double sd, double dt, int pf, int mode, long long n,
                                  double* __restrict__ fbar, float* __restrict__ xf) {
    const double g = sqrt(beta);
    const double sdt = sqrt(dt);
    GRID_STRIDE(i, n) {
        const double xv = x[i];
        double score = -(double)eps[i] / sd;
        if (pf) score = 0.5 * score;
        const double fb = -(-0.5 * beta * xv) + (g * g) * score;
        if (mode == 0) {
            fbar[i] = fb;
        } else {
            double nx = xv + fb * dt;
            if (z) nx = nx + g * sdt * z[i];
            x[i] = nx;
            if (xf) xf[i] = (float)nx;
        }
    }
}

}  // namespace

extern "C" int psld_vp_perturb_f32(const float* x0, const float* eps, const double* t, double beta0, double beta1,
                                   int batch, long long per_image, float* z_f32, double* u_f64, hipStream_t stream) {
    PSLD_CHECK_ARG(x0 && eps && t && batch > 0 && per_image > 0 && (z_f32 || u_f64), "psld_vp_perturb_f32: bad args");
    hipLaunchKernelGGL(vp_perturb_kernel, dim3(grid_for((long long)batch * per_image)), dim3(256), 0, stream, x0, eps, t,
                       beta0, beta1, batch, per_image, z_f32, u_f64);
    PSLD_CHECK_LAUNCH("psld_vp_perturb_f32");
    return PSLD_OK;
}

extern "C" int psld_vp_reverse_f64(double* x, const float* eps_pred, const double* z, double beta, double std, double dt,
                                   int probability_flow, int mode, long long n, double* f_bar, float* x_f32_out,
                                   hipStream_t stream) {
    PSLD_CHECK_ARG(x && eps_pred && n > 0 && (mode == 1 || f_bar), "psld_vp_reverse_f64: bad args");
    hipLaunchKernelGGL(vp_reverse_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, eps_pred, z, beta, std, dt,
                       probability_flow, mode, n, f_bar, x_f32_out);
    PSLD_CHECK_LAUNCH("psld_vp_reverse_f64");
    return PSLD_OK;
}

// x <- x*(1 - mask) + u*mask on both halves [x | m] of the state (samplers/sde.py:170-174): mask is {0, 1}, so this
// selects; the reference's arithmetic form is kept (built with -ffp-contract=off) to stay bit-identical with it.
__global__ void mask_combine_kernel(double* __restrict__ x, const double* __restrict__ u, const float* __restrict__ mask,
                                    long long per_img, long long total, float* __restrict__ x_f32) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long img = i / (2 * per_img), r = i - img * 2 * per_img;
        const double m = (double)mask[img * per_img + (r < per_img ? r : r - per_img)];
        const double v = x[i] * (1.0 - m) + u[i] * m;
        x[i] = v;
        if (x_f32) x_f32[i] = (float)v;
    }
}

// Classifier guidance (samplers/sde.py:90-94): adding g^2 * grad to the drift moves the state by g^2 * grad * dt;
// g^2 differs between the position and momentum halves (psld.py:201-203), so one coefficient per half.
__global__ void guide_kernel(double* __restrict__ x, const float* __restrict__ grad, double cx, double cm,
                             long long per_img, long long total, float* __restrict__ x_f32) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i % (2 * per_img);
        const double v = x[i] + (r < per_img ? cx : cm) * (double)grad[i];
        x[i] = v;
        if (x_f32) x_f32[i] = (float)v;
    }
}

extern "C" int psld_guide_f64(double* x, const float* grad, double coef_x, double coef_m, int batch, int c, int hw,
                              float* x_f32_out, hipStream_t stream) {
    PSLD_CHECK_ARG(x && grad && batch > 0 && c > 0 && hw > 0, "psld_guide_f64: bad args");
    const long long per_img = (long long)c * hw, total = 2 * per_img * batch;
    hipLaunchKernelGGL(guide_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, grad, coef_x, coef_m, per_img, total,
                       x_f32_out);
    PSLD_CHECK_LAUNCH("psld_guide_f64");
    return PSLD_OK;
}

extern "C" int psld_mask_combine_f64(double* x, const double* u, const float* mask, int batch, int c, int hw,
                                     float* x_f32_out, hipStream_t stream) {
    PSLD_CHECK_ARG(x && u && mask && batch > 0 && c > 0 && hw > 0, "psld_mask_combine_f64: bad args");
    const long long per_img = (long long)c * hw, total = 2 * per_img * batch;
    hipLaunchKernelGGL(mask_combine_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, u, mask, per_img, total,
                       x_f32_out);
    PSLD_CHECK_LAUNCH("psld_mask_combine_f64");
    return PSLD_OK;
}

extern "C" int psld_lincomb_f64(double* out, const double* base, const double* const* v, const double* coef, int nv,
                                long long n, float* out_f32, hipStream_t stream) {
    PSLD_CHECK_ARG(out && v && coef && nv >= 0 && nv <= 8 && n > 0, "psld_lincomb_f64: bad args");
    RkArgs a{};
    a.nv = nv;
    for (int j = 0; j < nv; ++j) { a.v[j] = v[j]; a.c[j] = coef[j]; }
    hipLaunchKernelGGL(rk_lincomb_kernel, dim3(grid_for(n)), dim3(256), 0, stream, out, base, a, n, out_f32);
    PSLD_CHECK_LAUNCH("psld_lincomb_f64");
    return PSLD_OK;
}

extern "C" int psld_scaled_norm_sq_f64(const double* const* v, const double* coef, int nv, const double* p,
                                       const double* q, double atol, double rtol, long long n, double* out,
                                       void* workspace, hipStream_t stream) {
    PSLD_CHECK_ARG(v && coef && nv >= 1 && nv <= 8 && p && q && out && workspace && n > 0,
                   "psld_scaled_norm_sq_f64: bad args");
    RkArgs a{};
    a.nv = nv;
    for (int j = 0; j < nv; ++j) { a.v[j] = v[j]; a.c[j] = coef[j]; }
    const int blocks = grid_for(n);
    double* part = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(rk_scaled_sq_kernel, dim3(blocks), dim3(256), 0, stream, a, p, q, atol, rtol, n, part);
    PSLD_CHECK_LAUNCH("rk_scaled_sq_kernel");
    hipLaunchKernelGGL(rk_final_kernel, dim3(1), dim3(64), 0, stream, part, blocks, out);
    PSLD_CHECK_LAUNCH("rk_final_kernel");
    return PSLD_OK;
}

extern "C" int psld_sscs_analytic_f64(double* x, const double* z, const psld_sscs_coeffs_t* k, int batch, int c,
                                      int hw, float* x_f32_out, hipStream_t stream) {
    PSLD_CHECK_ARG(x && z && k && batch > 0 && c > 0 && hw > 0, "psld_sscs_analytic_f64: bad args");
    const long long n = (long long)batch * c * hw;
    hipLaunchKernelGGL(sscs_analytic_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, z, *k, batch, c, hw,
                       x_f32_out);
    PSLD_CHECK_LAUNCH("psld_sscs_analytic_f64");
    return PSLD_OK;
}

extern "C" int psld_sscs_score_step_f64(double* x, const float* eps_pred, const psld_em_coeffs_t* k, int batch,
                                        int c, int hw, hipStream_t stream) {
    PSLD_CHECK_ARG(x && eps_pred && k && batch > 0 && c > 0 && hw > 0, "psld_sscs_score_step_f64: bad args");
    const long long n = (long long)batch * c * hw;
    hipLaunchKernelGGL(sscs_score_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, eps_pred, *k, batch, c, hw);
    PSLD_CHECK_LAUNCH("psld_sscs_score_step_f64");
    return PSLD_OK;
}

extern "C" long long psld_reduce_workspace_bytes(long long n) {
    (void)n;
    return 4096LL * sizeof(double);
}

extern "C" int psld_perturb_coeffs_f64(const double* t, int batch, const psld_sde_params_t* p, double xx_0,
                                       double mm_0, double* coeffs, int* nan_flag, hipStream_t stream) {
    PSLD_CHECK_ARG(t && p && coeffs && nan_flag && batch > 0, "psld_perturb_coeffs_f64: bad args");
    hipLaunchKernelGGL(perturb_coeffs_kernel, dim3(cdiv(batch, 128)), dim3(128), 0, stream, t, batch, *p, xx_0, mm_0,
                       coeffs, nan_flag);
    PSLD_CHECK_LAUNCH("psld_perturb_coeffs_f64");
    return PSLD_OK;
}

extern "C" int psld_perturb_f32(const float* x0, const float* m0, const float* eps, const double* coeffs,
                                   const psld_sde_params_t* p, int batch, int c, int hw, float* z_f32, double* u_f64,
                                   double* mu_f64, hipStream_t stream) {
    PSLD_CHECK_ARG(x0 && eps && coeffs && p && batch > 0 && c > 0 && hw > 0, "psld_perturb_f32: bad args");
    const long long n = (long long)batch * c * hw;
    auto al = [](const void* q, uintptr_t a) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; };
    if (hw % 4 == 0 && al(x0, 16) && al(m0, 16) && al(eps, 16) && al(z_f32, 16) && al(u_f64, 32) && al(mu_f64, 32))
        hipLaunchKernelGGL(perturb_kernel<4>, dim3(grid_for(n / 4)), dim3(256), 0, stream, x0, m0, eps, coeffs, *p, batch, c,
                           hw, z_f32, u_f64, mu_f64);
    else
        hipLaunchKernelGGL(perturb_kernel<1>, dim3(grid_for(n)), dim3(256), 0, stream, x0, m0, eps, coeffs, *p, batch, c, hw,
                           z_f32, u_f64, mu_f64);
    PSLD_CHECK_LAUNCH("psld_perturb_f32");
    return PSLD_OK;
}

extern "C" int psld_sqerr_loss_f32(const float* eps, const float* eps_pred, long long n, int reduce_mean, float* loss,
                                   float* grad, float grad_scale, void* workspace, hipStream_t stream) {
    PSLD_CHECK_ARG(eps && eps_pred && loss && workspace && n > 0, "psld_sqerr_loss_f32: bad args");
    PSLD_CHECK_ARG(((reinterpret_cast<uintptr_t>(eps) | reinterpret_cast<uintptr_t>(eps_pred) |
                     reinterpret_cast<uintptr_t>(grad)) & 15) == 0, "psld_sqerr_loss_f32: operands must be 16-byte aligned");
    int blocks = grid_for((n + 3) / 4);
    if (blocks > 4096) blocks = 4096;
    double* part = reinterpret_cast<double*>(workspace);
    // d loss / d eps_pred = 2*(pred - eps)/denom * upstream  ==  -(gscale) * (eps - pred)
    const double denom = reduce_mean ? (double)n : 1.0;
    const float gs = (float)(2.0 * (double)grad_scale / denom);
    hipLaunchKernelGGL(sqerr_partial_kernel, dim3(blocks), dim3(256), 0, stream, eps, eps_pred, n, part, grad, gs);
    PSLD_CHECK_LAUNCH("sqerr_partial_kernel");
    hipLaunchKernelGGL(sqerr_final_kernel, dim3(1), dim3(64), 0, stream, part, blocks, denom, loss);
    PSLD_CHECK_LAUNCH("sqerr_final_kernel");
    return PSLD_OK;
}

extern "C" int psld_em_step_f64(double* x, const float* eps_pred, const double* z, const psld_em_coeffs_t* k, int batch,
                                int c, int hw, float* x_f32_out, hipStream_t stream) {
    PSLD_CHECK_ARG(x && eps_pred && k && batch > 0 && c > 0 && hw > 0, "psld_em_step_f64: bad args");
    const long long n = (long long)batch * c * hw;
    hipLaunchKernelGGL(em_step_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, eps_pred, z, *k, batch, c, hw,
                       x_f32_out);
    PSLD_CHECK_LAUNCH("psld_em_step_f64");
    return PSLD_OK;
}

extern "C" int psld_reverse_sde_f64(const double* x, const float* eps_pred, const psld_em_coeffs_t* k, int batch, int c,
                                    int hw, double* f_bar, double* g_bar, hipStream_t stream) {
    PSLD_CHECK_ARG(x && eps_pred && k && f_bar && batch > 0, "psld_reverse_sde_f64: bad args");
    const long long n = (long long)batch * c * hw;
    hipLaunchKernelGGL(reverse_sde_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, eps_pred, *k, batch, c, hw,
                       f_bar, g_bar);
    PSLD_CHECK_LAUNCH("psld_reverse_sde_f64");
    return PSLD_OK;
}

extern "C" int psld_reverse_sde_rows_f64(const double* x, const float* eps_pred, const double* t_rev,
                                         const psld_sde_params_t* p, double xx_0, double mm_0, int score_mode,
                                         int probability_flow, int batch, int c, int hw, double* f_bar, double* g_bar,
                                         int* nan_flag, hipStream_t stream) {
    PSLD_CHECK_ARG(x && t_rev && p && f_bar && nan_flag && batch > 0 && c > 0 && hw > 0 && batch <= 65535,
                   "psld_reverse_sde_rows_f64: bad args");
    const long long n = (long long)c * hw;
    int bx = (int)((n + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(reverse_sde_rows_kernel, dim3(bx, batch), dim3(256), 0, stream, x, eps_pred, t_rev, *p, xx_0, mm_0,
                       score_mode, probability_flow, eps_pred ? 0 : 1, c, hw, f_bar, g_bar, nan_flag);
    PSLD_CHECK_LAUNCH("psld_reverse_sde_rows_f64");
    return PSLD_OK;
}

// ---- ScoreLoss beyond the eps-MSE (main/losses.py:38-39, 55-63) --------------------------------------------
// mode 1: L1 criterion |eps - eps_pred| (f32 like nn.L1Loss); mode 2: 'nll' weighting, per element
// (score(eps_pred) - score(eps))^2 * g(t)^2 with score = -eps / std(t) (vpsde.py:26-27) and g(t)^2 = beta(t)
// (vpsde.py:97-99), evaluated in f64 like the reference (t is f64 there).
__global__ void vp_score_loss_kernel(const float* __restrict__ e, const float* __restrict__ ep,
                                     const double* __restrict__ t, double beta0, double beta1, long long per,
                                     long long n, int mode, double* __restrict__ part, float* __restrict__ grad,
                                     double gscale) {
    __shared__ double red[4];
    double acc = 0.0;
    GRID_STRIDE(i, n) {
        if (mode == 1) {
            const float d = e[i] - ep[i];
            acc += (double)fabsf(d);
            if (grad) grad[i] = (float)(gscale * (d > 0.f ? -1.0 : (d < 0.f ? 1.0 : 0.0)));
        } else {
            const double tt = t[i / per];
            const double lmc = -0.25 * (tt * tt) * (beta1 - beta0) - 0.5 * tt * beta0;   // vpsde.py:74-76
            const double sd = sqrt(1.0 - exp(2.0 * lmc));
            const double g2 = beta0 + tt * (beta1 - beta0);
            const double d = (-(double)ep[i] / sd) - (-(double)e[i] / sd);
            acc += d * d * g2;
            if (grad) grad[i] = (float)(gscale * 2.0 * d * g2 * (-1.0 / sd));
        }
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void scaled_final_kernel(const double* __restrict__ part, int nparts, double denom, double* __restrict__ out64,
                                    float* __restrict__ out32) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) acc += part[i];
    acc = wave_sum_d(acc);
    if (threadIdx.x == 0) {
        if (out64) out64[0] = acc / denom;
        if (out32) out32[0] = (float)(acc / denom);
    }
}

extern "C" int psld_vp_score_loss(const float* eps, const float* eps_pred, const double* t, double beta0, double beta1,
                                  int batch, long long per, int mode, int reduce_mean, void* loss, float* grad,
                                  float grad_scale, void* workspace, hipStream_t stream) {
    PSLD_CHECK_ARG(eps && eps_pred && loss && workspace && batch > 0 && per > 0 && (mode == 1 || (mode == 2 && t)),
                   "psld_vp_score_loss: bad args");
    const long long n = (long long)batch * per;
    int blocks = grid_for(n);
    double* part = reinterpret_cast<double*>(workspace);
    const double denom = reduce_mean ? (double)n : 1.0;
    hipLaunchKernelGGL(vp_score_loss_kernel, dim3(blocks), dim3(256), 0, stream, eps, eps_pred, t, beta0, beta1, per, n,
                       mode, part, grad, (double)grad_scale / denom);
    PSLD_CHECK_LAUNCH("vp_score_loss_kernel");
    hipLaunchKernelGGL(scaled_final_kernel, dim3(1), dim3(64), 0, stream, part, blocks, denom,
                       mode == 2 ? reinterpret_cast<double*>(loss) : nullptr,
                       mode == 1 ? reinterpret_cast<float*>(loss) : nullptr);
    PSLD_CHECK_LAUNCH("scaled_final_kernel");
    return PSLD_OK;
}
