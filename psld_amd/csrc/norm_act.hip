// GroupNorm(+SiLU) forward / backward on NHWC activations — the HBM-bound part of the U-Net.
//
// Reference: nn.GroupNorm(num_groups=min(C//4,32), eps=1e-6) followed by nn.SiLU
// (song_sde/layerspp.py:219,231,243,264; :67,77 (attention, no act); ncsnpp.py:276-280,427).
//
// Thread mapping for every kernel here: a block owns one image `n` and a contiguous chunk of
// its pixels.  Thread -> (channel quad q = tid % CQ, pixel lane pl = tid / CQ), CQ = C/4, so a
// wave reads whole 16-byte-per-lane contiguous runs of the NHWC row (coalesced float4).
// Reductions: per-thread fp32 partials over <= ~64 pixels, then fp64 across threads (LDS) and
// across chunks (finalize kernel) — deterministic, no atomics.
#include "common.h"
#include "psld_hip.h"

namespace {

constexpr int MAXT = 256;
constexpr int MAXG = 32;

struct Map {
    int cq, pl, threads, chunks, chunk_px;
};

// elementwise: true for the apply kernels (no cross-chunk reduction behind them: more, shorter blocks stream better —
// measured 43 vs 48 us on 128x32x32x256), false for the statistics kernels whose finalize pass walks the chunks
inline Map make_map(int batch, int hw, int c, bool elementwise = false) {
    Map m;
    m.cq = c / 4;
    m.pl = MAXT / m.cq;
    if (m.pl < 1) m.pl = 1;
    if (m.pl > hw) m.pl = hw;
    m.threads = m.cq * m.pl;
    // aim for ~1024 (statistics) / ~8192 (elementwise) blocks but <= 64 pixels per thread per chunk
    int chunks = cdiv(elementwise ? 8192 : 1024, batch);
    if (!elementwise && chunks > 32) chunks = 32;   // the finalize kernels walk the chunks serially
    int max_chunks = cdiv(hw, m.pl);          // at least one pixel per thread
    if (chunks > max_chunks) chunks = max_chunks;
    int min_chunks = cdiv(hw, m.pl * 64);
    if (chunks < min_chunks) chunks = min_chunks;
    if (chunks < 1) chunks = 1;
    m.chunk_px = cdiv(hw, chunks);
    m.chunks = cdiv(hw, m.chunk_px);
    return m;
}

// ---- forward statistics -------------------------------------------------------------------
__global__ void gn_partial_kernel(const float* __restrict__ x, int hw, int c, int groups, int cq, int pl,
                                  int chunk_px, int chunks, double* __restrict__ part) {
    __shared__ double ts[MAXT][2];               // per-thread (sum, sum of squares), combined in thread order
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int tid = threadIdx.x;
    const int q = tid % cq, l = tid / cq;
    const int p0 = chunk * chunk_px, p1 = min(hw, p0 + chunk_px);
    const float* base = x + ((long long)n * hw) * c + q * 4;
    float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
#pragma unroll 4
    for (int p = p0 + l; p < p1; p += pl) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)p * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[e] += v[e];
            ss[e] += v[e] * v[e];
        }
    }
    const int cpg = c / groups;
    double* out = part + (((long long)n * chunks + chunk) * groups) * 2;
    if (cpg % 4 == 0) {
        // a channel quad lies in one group: group g = quads [g*cpg/4, (g+1)*cpg/4) x all pixel lanes
        ts[tid][0] = (double)s[0] + (double)s[1] + (double)s[2] + (double)s[3];
        ts[tid][1] = (double)ss[0] + (double)ss[1] + (double)ss[2] + (double)ss[3];
        __syncthreads();
        if (tid < groups * 2) {
            const int g = tid >> 1, w = tid & 1, qpg = cpg / 4;
            double t = 0.0;
            for (int ll = 0; ll < pl; ++ll)
                for (int qq = g * qpg; qq < (g + 1) * qpg; ++qq) t += ts[ll * cq + qq][w];
            out[tid] = t;
        }
    } else {
        double t = 0.0;                          // thread (g, w) accumulates over the four element rounds
        for (int e = 0; e < 4; ++e) {
            ts[tid][0] = (double)s[e];
            ts[tid][1] = (double)ss[e];
            __syncthreads();
            if (tid < groups * 2) {
                const int g = tid >> 1, w = tid & 1;
                for (int ll = 0; ll < pl; ++ll)
                    for (int qq = 0; qq < cq; ++qq)
                        if ((qq * 4 + e) / cpg == g) t += ts[ll * cq + qq][w];
            }
            __syncthreads();
        }
        if (tid < groups * 2) out[tid] = t;
    }
}

// fine: partial sums per group in `part` (1 from gn_partial_kernel; channels-per-group / 8 for the 8-channel sums a
// limb kernel's epilogue wrote), added in index order
__global__ void gn_finalize_kernel(const double* __restrict__ part, int hw, int c, int groups, int chunks, int fine,
                                   float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ mean, float* __restrict__ rstd,
                                   float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ float smean[MAXG], srstd[MAXG];
    __shared__ double ps[8][MAXG][2];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int cpg = c / groups;
    // chunk lane kl adds chunks kl, kl + 8, ... of group g (independent loads in flight: a single thread walking all
    // chunks pays one memory latency per chunk, 9 us per launch); the 8 lanes are then combined in lane order
    {
        const int g = tid & 31, kl = tid >> 5;
        double s = 0, ss = 0;
        if (g < groups)
            for (int k = kl; k < chunks; k += 8) {
                const double* pp = part + (((long long)n * chunks + k) * groups + g) * fine * 2;
                for (int f = 0; f < fine; ++f) {
                    s += pp[2 * f];
                    ss += pp[2 * f + 1];
                }
            }
        ps[kl][g][0] = s;
        ps[kl][g][1] = ss;
    }
    __syncthreads();
    if (tid < groups) {
        double s = 0, ss = 0;
#pragma unroll
        for (int kl = 0; kl < 8; ++kl) {
            s += ps[kl][tid][0];
            ss += ps[kl][tid][1];
        }
        const double cnt = (double)cpg * hw;
        const double mu = s / cnt;
        double var = ss / cnt - mu * mu;
        if (var < 0) var = 0;
        const float m = (float)mu, r = (float)(1.0 / sqrt(var + (double)eps));
        smean[tid] = m;
        srstd[tid] = r;
        mean[n * groups + tid] = m;
        rstd[n * groups + tid] = r;
    }
    __syncthreads();
    for (int ch = tid; ch < c; ch += blockDim.x) {
        const int g = ch / cpg;
        const float sc = srstd[g] * gamma[ch];
        scale[(long long)n * c + ch] = sc;
        shift[(long long)n * c + ch] = beta[ch] - smean[g] * sc;
    }
}

__global__ void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                const float* __restrict__ shift, float* __restrict__ y, int hw, int c, int cq,
                                int pl, int chunk_px, int act, float drop_p, unsigned long long seed,
                                const unsigned long long* __restrict__ seed_dev) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    if (seed_dev) seed += seed_dev[0];        // per-step seed kept in device memory (graph-captured training)
    const int q = threadIdx.x % cq, l = threadIdx.x / cq;
    const int p0 = chunk * chunk_px, p1 = min(hw, p0 + chunk_px);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + (long long)n * c + q * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + (long long)n * c + q * 4);
    const long long off = ((long long)n * hw) * c + q * 4;
    const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
#pragma unroll 4
    for (int p = p0 + l; p < p1; p += pl) {
        const long long idx = off + (long long)p * c;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + idx);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = v[e] * sc[e] + sh[e];
            float a = act ? silu_f(z) : z;
            if (drop_p > 0.f) a = psld_dropout_keep(seed, (unsigned long long)(idx + e), drop_p) ? a * keep_scale : 0.f;
            o[e] = a;
        }
        *reinterpret_cast<f32x4*>(y + idx) = o;
    }
}

// The same pass writing bf16 LIMB PLANES [pixel][c/32][3 limbs][32 channels] instead of fp32 (include/psld_hip.h: the
// 3x3 convolution that consumes the activation stages them by LDS-DMA, no split in the MFMA kernel).  hi = rne_bf16(a),
// mid = rne_bf16(a - hi), lo = a - hi - mid: exactly the decomposition the convolution kernels apply to fp32 input.
__device__ __forceinline__ void gn_split3(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const bf16x2_t ph = {(__bf16)x0, (__bf16)x1};
    hi = __builtin_bit_cast(unsigned, ph);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    const bf16x2_t pm = {(__bf16)r0, (__bf16)r1};
    mid = __builtin_bit_cast(unsigned, pm);
    const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
    const bf16x2_t pl = {(__bf16)s0, (__bf16)s1};
    lo = __builtin_bit_cast(unsigned, pl);
}

__global__ void gn_apply_limb_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                     const float* __restrict__ shift, unsigned char* __restrict__ y, int hw, int c, int cq,
                                     int pl, int chunk_px, int act, float drop_p, unsigned long long seed,
                                const unsigned long long* __restrict__ seed_dev) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
    const int n = blockIdx.y, chunk = blockIdx.x;
    if (seed_dev) seed += seed_dev[0];        // per-step seed kept in device memory (graph-captured training)
    const int q = threadIdx.x % cq, l = threadIdx.x / cq;
    const int p0 = chunk * chunk_px, p1 = min(hw, p0 + chunk_px);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + (long long)n * c + q * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + (long long)n * c + q * 4);
    const long long off = ((long long)n * hw) * c + q * 4;
    const long long yoff = ((long long)n * hw) * c * 6 + (q >> 3) * 192 + (q & 7) * 8;
    const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
#pragma unroll 4
    for (int p = p0 + l; p < p1; p += pl) {
        const long long idx = off + (long long)p * c;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + idx);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = v[e] * sc[e] + sh[e];
            float a = act ? silu_f(z) : z;
            if (drop_p > 0.f) a = psld_dropout_keep(seed, (unsigned long long)(idx + e), drop_p) ? a * keep_scale : 0.f;
            o[e] = a;
        }
        unsigned h0, m0, l0, h1, m1, l1;
        gn_split3(o[0], o[1], h0, m0, l0);
        gn_split3(o[2], o[3], h1, m1, l1);
        unsigned char* d = y + yoff + (long long)p * c * 6;
        *reinterpret_cast<u32x2_t*>(d) = u32x2_t{h0, h1};
        *reinterpret_cast<u32x2_t*>(d + 64) = u32x2_t{m0, m1};
        *reinterpret_cast<u32x2_t*>(d + 128) = u32x2_t{l0, l1};
    }
}

// ---- backward -------------------------------------------------------------------------------
// pass 1: per (n, chunk, channel): s1 = sum dz, s2 = sum dz * xhat      (dz = dy * act'(z))
__global__ void gn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, int hw,
                                      int c, int groups, int cq, int pl, int chunk_px, int chunks, int act,
                                      float drop_p, unsigned long long seed,
                                      const unsigned long long* __restrict__ seed_dev, float* __restrict__ part) {
    extern __shared__ float red[];  // [pl][cq][8]
    const int n = blockIdx.y, chunk = blockIdx.x;
    if (seed_dev) seed += seed_dev[0];        // per-step seed kept in device memory (graph-captured training)
    const int tid = threadIdx.x;
    const int q = tid % cq, l = tid / cq;
    const int p0 = chunk * chunk_px, p1 = min(hw, p0 + chunk_px);
    const int cpg = c / groups;
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ch = q * 4 + e;
        const int g = ch / cpg;
        mu[e] = mean[n * groups + g];
        rs[e] = rstd[n * groups + g];
        ga[e] = gamma[ch];
        be[e] = beta[ch];
    }
    const long long off = ((long long)n * hw) * c + q * 4;
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
#pragma unroll 4
    for (int p = p0 + l; p < p1; p += pl) {
        const long long idx = off + (long long)p * c;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + idx);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(dy + idx);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xv[e] - mu[e]) * rs[e];
            float dz = gv[e];
            if (drop_p > 0.f) dz = psld_dropout_keep(seed, (unsigned long long)(idx + e), drop_p) ? dz * keep_scale : 0.f;
            if (act) dz *= dsilu_f(xh * ga[e] + be[e]);
            s1[e] += dz;
            s2[e] += dz * xh;
        }
    }
    float* my = red + ((long long)l * cq + q) * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        my[e] = s1[e];
        my[4 + e] = s2[e];
    }
    __syncthreads();
    // threads 0..cq*8-1 reduce over pixel lanes
    for (int i = tid; i < cq * 8; i += blockDim.x) {
        double acc = 0;
        for (int ll = 0; ll < pl; ++ll) acc += (double)red[(long long)ll * cq * 8 + i];
        const int qq = i / 8, k = i % 8;
        const int ch = qq * 4 + (k & 3);
        // layout [n][chunk][2][c]
        part[(((long long)n * chunks + chunk) * 2 + (k >> 2)) * c + ch] = (float)acc;
    }
}

// pass 2 (three-pass form only: maps whose (image, 32-channel) slab does not fit registers), one block per image: sum the
// chunk partials -> sums[n][0][c] = sum_p dz, sums[n][1][c] = sum_p dz * xhat (what the one-pass kernels write directly; the
// parameter gradients dbeta / dgamma are their sums over the batch: psld_param_reduce*_f32), group means m1, m2 and the
// coefficient rows
//   coef[n][0][c] = rstd*gamma  (multiplies dz)
//   coef[n][1][c] = rstd*m1_g   (subtracted)
//   coef[n][2][c] = rstd*m2_g   (multiplies xhat, subtracted)
// No atomics anywhere: bitwise repeatable.
__global__ void __launch_bounds__(1024) gn_bwd_coef_kernel(const float* __restrict__ part, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, int hw, int c, int groups,
                                                           int chunks, float* __restrict__ coef, float* __restrict__ sums) {
    __shared__ double sh[2 * MAXT * 4];          // s1*gamma, s2*gamma per channel
    __shared__ double g1[MAXG], g2[MAXG];
    const int tid = threadIdx.x;
    const int n = blockIdx.x;
    const int cpg = c / groups;
    for (int ch = tid; ch < c; ch += blockDim.x) {
        double a = 0, b = 0;
        // eight chunks' loads in flight at a time, added in chunk order (a load-add-load-add loop pays one memory
        // latency per chunk)
        for (int k0 = 0; k0 < chunks; k0 += 8) {
            float va[8], vb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = min(k0 + j, chunks - 1);
                va[j] = part[(((long long)n * chunks + k) * 2 + 0) * c + ch];
                vb[j] = part[(((long long)n * chunks + k) * 2 + 1) * c + ch];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j < chunks) {
                    a += (double)va[j];
                    b += (double)vb[j];
                }
        }
        sums[((long long)n * 2 + 0) * c + ch] = (float)a;
        sums[((long long)n * 2 + 1) * c + ch] = (float)b;
        sh[ch] = a * (double)gamma[ch];
        sh[MAXT * 4 + ch] = b * (double)gamma[ch];
    }
    __syncthreads();
    if (tid < groups) {
        double a = 0, b = 0;
        for (int i = 0; i < cpg; ++i) {
            a += sh[tid * cpg + i];
            b += sh[MAXT * 4 + tid * cpg + i];
        }
        g1[tid] = a;
        g2[tid] = b;
    }
    __syncthreads();
    const double cnt = (double)cpg * hw;
    for (int ch = tid; ch < c; ch += blockDim.x) {
        const int g = ch / cpg;
        const float r = rstd[n * groups + g];
        coef[((long long)n * 3 + 0) * c + ch] = r * gamma[ch];
        coef[((long long)n * 3 + 1) * c + ch] = (float)((double)r * g1[g] / cnt);
        coef[((long long)n * 3 + 2) * c + ch] = (float)((double)r * g2[g] / cnt);
    }
}

// dx = k0 dz - k1 - xhat k2 in ONE spelled-out order (two fused multiply-adds): the three kernels that form it must agree bit
// for bit, and left to itself hipcc contracts the expression differently from kernel to kernel
__device__ __forceinline__ float gn_dx(float k0, float dz, float k1, float xh, float k2) {
    return __builtin_fmaf(-xh, k2, __builtin_fmaf(k0, dz, -k1));
}

// pass 3: dx = coef0*dz - coef1 - xhat*coef2
__global__ void gn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ coef, int hw, int c, int groups, int cq, int pl,
                                    int chunk_px, int act, float drop_p, unsigned long long seed,
                                    const unsigned long long* __restrict__ seed_dev, int accumulate,
                                    const float* __restrict__ add, float add_scale, float* __restrict__ dx) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    if (seed_dev) seed += seed_dev[0];        // per-step seed kept in device memory (graph-captured training)
    const int q = threadIdx.x % cq, l = threadIdx.x / cq;
    const int p0 = chunk * chunk_px, p1 = min(hw, p0 + chunk_px);
    const int cpg = c / groups;
    float mu[4], rs[4], ga[4], be[4], c0[4], c1[4], c2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ch = q * 4 + e;
        const int g = ch / cpg;
        mu[e] = mean[n * groups + g];
        rs[e] = rstd[n * groups + g];
        ga[e] = gamma[ch];
        be[e] = beta[ch];
        c0[e] = coef[((long long)n * 3 + 0) * c + ch];
        c1[e] = coef[((long long)n * 3 + 1) * c + ch];
        c2[e] = coef[((long long)n * 3 + 2) * c + ch];
    }
    const long long off = ((long long)n * hw) * c + q * 4;
    const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
#pragma unroll 4
    for (int p = p0 + l; p < p1; p += pl) {
        const long long idx = off + (long long)p * c;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + idx);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(dy + idx);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xv[e] - mu[e]) * rs[e];
            float dz = gv[e];
            if (drop_p > 0.f) dz = psld_dropout_keep(seed, (unsigned long long)(idx + e), drop_p) ? dz * keep_scale : 0.f;
            if (act) dz *= dsilu_f(xh * ga[e] + be[e]);
            o[e] = gn_dx(c0[e], dz, c1[e], xh, c2[e]);
        }
        float* dp = dx + idx;
        if (add) {                       // gradient of a parallel identity branch: dx += add_scale * add
            const f32x4 av = *reinterpret_cast<const f32x4*>(add + idx);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += add_scale * av[e];
        }
        if (accumulate) {
            const f32x4 old = *reinterpret_cast<const f32x4*>(dp);
            o += old;
        }
        *reinterpret_cast<f32x4*>(dp) = o;
    }
}

#ifdef PSLD_ABLATIONS
// ablation library only (tools/gnb_stamps.py): per-workgroup time stamps and timing-only modes of the fused backward
__device__ unsigned long long* g_gnb_dbg = nullptr;     // [workgroups][8]
__device__ int g_gnb_mode = 0;                          // bit 0 delay odd workgroups by (mode >> 8) x 3.4 us, bit 1 no reduction, bit 2 no stores
#define GNB_STAMP(k)                                                                                                   \
    do {                                                                                                               \
        if (g_gnb_dbg && threadIdx.x == 0)                                                                             \
            g_gnb_dbg[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] =                                    \
                (k) >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();                            \
    } while (0)
#define GNB_MODE g_gnb_mode
#define GNP_STAMP(k)                                                                                                   \
    do {                                                                                                               \
        if (g_gnb_dbg && stamped && threadIdx.x == 0) g_gnb_dbg[(long long)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define GNB_STAMP(k)
#define GNB_MODE 0
#define GNP_STAMP(k)
#endif

// workgroup barrier that orders LDS accesses only: a __syncthreads also waits for every outstanding global load (vmcnt(0)),
// which would drain loads that are meant to fly across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- backward in ONE pass over (dy, x) ----------------------------------------------------------------------------
// A block owns image n and a slab of `gb` whole groups (cw = gb * cpg <= 32 channels); its threads keep the slab's dy and
// x values in REGISTERS (ITEMS float4 of each per thread): pass 1 turns them into dz and xhat in place and reduces the
// per-channel sums (fp32 per thread, fp64 across the pixel lanes through LDS, like the three-kernel path), the group
// terms follow from those, pass 2 writes dx straight from the registers.  dy and x are read once instead of twice (20 ->
// 12 bytes per element) and two of the three launches go away; the per-image channel sums land in part[n][2][c] (the
// caller's `sums`), whose sums over the batch are dbeta / dgamma (psld_param_reduce*_f32, one launch for many layers).
// csum_img: column sums per image of the FINAL dx values this block writes - the bias gradient of the layer whose output
// gradient dx is.  Without a third operand from a closed form of the sums the kernel reduces anyway; with one (add /
// accumulate_dx: this block is the last writer of a residual stream's gradient) by summing the written values: per-thread
// fp32 over its items, fp64 across the pixel lanes, like every other reduction here.
// EARLY (launched only with a third operand): the parallel branch's gradient (or, without one, the previous dx) is asked for
// as soon as pass 1 has consumed the raw values - a third register set - and lands while the workgroup sits in the barriers
// and the reduction, the quarter of a workgroup's life in which its CU's memory pipe used to be idle (gnb_stamps.md); the
// barriers then wait for LDS only.  Same arithmetic, same order: bitwise the plain form.
// Thread -> (channel quad q = tid % cq, pixel lane l = tid / cq), pixels l, l + pl, ...
template <int ITEMS, bool EARLY = false>
__global__ void __launch_bounds__(512) gn_bwd_fused_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int hw, int c, int groups, int gb, int pl, int act, float drop_p,
                                                           unsigned long long seed,
                                                           const unsigned long long* __restrict__ seed_dev, int accumulate,
                                                           const float* __restrict__ add, float add_scale,
                                                           float* __restrict__ dx, float* __restrict__ part,
                                                           float* __restrict__ csum_img, int ld_img) {
    extern __shared__ float red[];               // [pl][cq][8] thread sums, then (as doubles) [cw][2] channel sums + [gb][2],
                                                 // then (column sums of dx) [pl][cq][4] thread sums of xhat, [cw] + [cw] channel sums
    const int n = blockIdx.y, slab = blockIdx.x;
    GNB_STAMP(6);
    GNB_STAMP(0);
    if ((GNB_MODE & 1) && ((blockIdx.x + blockIdx.y) & 1))
        for (int i = 0; i < (GNB_MODE >> 8); ++i) __builtin_amdgcn_s_sleep(127);
    if ((GNB_MODE & 8) && blockIdx.y * gridDim.x + blockIdx.x < 256) {      // first round only: phase (id / 8) % 4 of four
        const int ph = ((blockIdx.y * gridDim.x + blockIdx.x) >> 3) & ((GNB_MODE & 16) ? 1 : 3);
        for (int i = 0; i < ph * (GNB_MODE >> 8); ++i) __builtin_amdgcn_s_sleep(127);
    }
    if (seed_dev) seed += seed_dev[0];
    const int tid = threadIdx.x;
    const int cpg = c / groups;
    const int cw = gb * cpg, cq = cw >> 2;
    const int q = tid % cq, l = tid / cq;
    const int c0 = slab * cw;
    const int ch0 = c0 + q * 4;
    const int g = ch0 / cpg;                     // cpg % 4 == 0: a quad lies in one group
    const float mu = mean[n * groups + g], rs = rstd[n * groups + g];
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + ch0), be = *reinterpret_cast<const f32x4*>(beta + ch0);
    // byte offset of this thread's first element / between its items (tensors < 4 GB: gn_bwd_fused_plan): one uniform base
    // and one 32-bit offset per access instead of a 64-bit address per item and tensor
    const unsigned uo = (unsigned)((((long long)n * hw + l) * c + ch0) * 4), istride = (unsigned)pl * c * 4;
    auto at = [&](const float* base, int i) {
        return reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(base) + (uo + i * istride));
    };
    const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    f32x4 xv[ITEMS], gv[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int p = l + i * pl;
        if (p < hw) {
            xv[i] = *at(x, i);
            gv[i] = *at(dy, i);
        } else {
            xv[i] = gv[i] = f32x4{0.f, 0.f, 0.f, 0.f};      // (pass 2 computes on every item and stores the valid ones)
        }
    }
    GNB_STAMP(1);
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, s3[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int p = l + i * pl;
        if (p < hw) {
            const long long idx = (long long)((uo + i * istride) >> 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[i][e] - mu) * rs;
                float dz = gv[i][e];
                if (drop_p > 0.f) dz = psld_dropout_keep(seed, (unsigned long long)(idx + e), drop_p) ? dz * keep_scale : 0.f;
                if (act) dz *= dsilu_f(xh * ga[e] + be[e]);
                s1[e] += dz;
                s2[e] += dz * xh;
                s3[e] += xh;
                xv[i][e] = xh;
                gv[i][e] = dz;
            }
        }
    }
    GNB_STAMP(2);
    float* my = red + ((long long)l * cq + q) * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        my[e] = s1[e];
        my[4 + e] = s2[e];
    }
    double* chs = reinterpret_cast<double*>(red + (long long)pl * cq * 8);      // [cw][2]: gamma-weighted channel sums
    double* grp = chs + cw * 2;                                                  // [gb][2]
    // column sums of dx per image (the bias / time-embedding gradient of the convolution whose output gradient this dx is)
    // without a pass over dx: sum_p dx = k0 sum_p dz - hw k1 - k2 sum_p xhat per channel
    float* red3 = reinterpret_cast<float*>(grp + gb * 2);                        // [pl][cq][4]
    float* cs1 = red3 + (long long)pl * cq * 4;                                  // [cw] sum_p dz (as stored in part)
    float* cs3 = cs1 + cw;                                                       // [cw] sum_p xhat
    const bool third = add != nullptr || accumulate != 0;
    const bool closed = csum_img != nullptr && !third;     // column sums of dx from the channel sums
    if (closed) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red3[((long long)l * cq + q) * 4 + e] = s3[e];
    }
    [[maybe_unused]] f32x4 tv[EARLY ? ITEMS : 1];
    if constexpr (EARLY) {
        const float* third_src = add ? add : dx;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int p = l + i * pl;
            tv[i] = p < hw ? *at(third_src, i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        lds_barrier();
    } else {
        __syncthreads();
    }
    GNB_STAMP(3);
    if (!(GNB_MODE & 2))
    for (int i = tid; i < cq * 8; i += blockDim.x) {
        double acc = 0;
        for (int ll = 0; ll < pl; ++ll) acc += (double)red[(long long)ll * cq * 8 + i];
        const int qq = i >> 3, k = i & 7;
        const int chl = qq * 4 + (k & 3);
        const float rounded = (float)acc;                                          // what the three-kernel path stores
        part[(((long long)n * 2 + (k >> 2))) * c + c0 + chl] = rounded;
        chs[chl * 2 + (k >> 2)] = (double)rounded * (double)gamma[c0 + chl];
        if (closed && k < 4) cs1[chl] = rounded;
    }
    if (closed && tid >= cq * 8 && tid < cq * 12) {
        const int j = tid - cq * 8;
        double acc = 0;
        for (int ll = 0; ll < pl; ++ll) acc += (double)red3[(long long)ll * cq * 4 + j];
        cs3[j] = (float)acc;
    }
    if constexpr (EARLY) lds_barrier(); else __syncthreads();
    if (tid < gb * 2) {
        const int gg = tid >> 1, w = tid & 1;
        double t = 0;
        for (int i = 0; i < cpg; ++i) t += chs[(gg * cpg + i) * 2 + w];
        grp[gg * 2 + w] = t;
    }
    if constexpr (EARLY) lds_barrier(); else __syncthreads();
    GNB_STAMP(4);
    const int gl = (q * 4) / cpg;
    const double cnt = (double)cpg * hw;
    const float k1 = (float)((double)rs * grp[gl * 2 + 0] / cnt), k2 = (float)((double)rs * grp[gl * 2 + 1] / cnt);
    f32x4 k0;
#pragma unroll
    for (int e = 0; e < 4; ++e) k0[e] = rs * ga[e];
    if (closed && tid < cw) {        // one thread per channel of the slab, with the coefficients the stores below use
        const int gc = tid / cpg;
        const float rc = rstd[n * groups + (c0 + tid) / cpg];
        const float c0k = rc * gamma[c0 + tid];
        const float c1k = (float)((double)rc * grp[gc * 2 + 0] / cnt), c2k = (float)((double)rc * grp[gc * 2 + 1] / cnt);
        csum_img[(long long)n * ld_img + c0 + tid] =
            (float)((double)c0k * (double)cs1[tid] - (double)hw * (double)c1k - (double)c2k * (double)cs3[tid]);
    }
    // dx into gv; the parallel branch's gradient / the previous dx are loaded for ALL items before the first is used (xv is
    // free by then): a load - wait - store chain per item costs one memory latency each
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[i][e] = gn_dx(k0[e], gv[i][e], k1, xv[i][e], k2);
    if (add) {
        if constexpr (!EARLY) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int p = l + i * pl;
                if (p < hw) xv[i] = *at(add, i);
            }
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[i][e] += add_scale * (EARLY ? tv[i][e] : xv[i][e]);
    }
    if (accumulate) {
        if (!EARLY || add) {             // (EARLY without a parallel branch: the previous dx is what came early)
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int p = l + i * pl;
                if (p < hw) xv[i] = *at(dx, i);
            }
        }
        if (EARLY && !add) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) gv[i] += tv[i];
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) gv[i] += xv[i];
        }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int p = l + i * pl;
        if (p < hw && (!(GNB_MODE & 4) || gv[i][0] == 12345.678f))
            *const_cast<f32x4*>(at(dx, i)) = gv[i];
    }
    if (csum_img && third) {         // column sums of the values just written (red3 is unused on this path until here)
        float s4[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int p = l + i * pl;
            if (p < hw) {
#pragma unroll
                for (int e = 0; e < 4; ++e) s4[e] += gv[i][e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) red3[((long long)l * cq + q) * 4 + e] = s4[e];
        __syncthreads();
        if (tid < cw) {
            double acc = 0;
            for (int ll = 0; ll < pl; ++ll) acc += (double)red3[(long long)ll * cw + tid];
            csum_img[(long long)n * ld_img + c0 + tid] = (float)acc;
        }
    }
    GNB_STAMP(5);
    GNB_STAMP(7);
}

// ---- the one-pass backward as resident workgroups with the NEXT slab's x prefetched into LDS ---------------------------
// Time stamps inside gn_bwd_fused_kernel (profiles/r04/gnb_stamps.md): a workgroup that holds a 32-channel slab of a
// 32x32 image (128 KB of x + 128 KB of dy in registers) keeps its CU's memory pipe busy for 15 of the 21 us it lives -
// loads 256 KB, THEN waits at the barrier, reduces, THEN stores 128 KB - and a CU moves ~25 GB/s whatever the other CUs
// do (staggering the workgroups' phases changed nothing), so the idle quarter is lost bandwidth.  Registers hold one slab
// per CU; LDS (160 KB) is free: a workgroup that walks `per` slabs of the same channels asks, as soon as its dy loads have
// landed, for the NEXT slab's x by asynchronous global -> LDS loads (global_load_lds_dwordx4: no VGPR destination,
// lane-linear 1 KB per wave instruction, every thread later reads back exactly the 16 bytes per item it asked for, so no
// barrier guards the image), which then fly during the barriers, the reduction, pass 2 and the stores.  The loads of dy and of
// the image are asm with hand-placed counted vmcnt waits (hipcc does not count asm loads, and a counted wait of its own
// behind them would drain them); the barriers are raw s_barrier with an lgkmcnt-only wait (a __syncthreads waits
// vmcnt(0)).  Arithmetic and summation order are those of gn_bwd_fused_kernel: bitwise equal results.  Only without a
// third operand (the dropout GroupNorm of every block): with the parallel branch's gradient or a previous dx to add, a
// variant that lands THAT operand in the image (asked for behind x and dy, read in pass 2) measured 128 / 118 us against 126 /
// 126 us of the one-slab kernel on 128x32x32x256 - 512 KB per slab keep the CU's pipe busy for 20 of its 29 us anyway -
// and was dropped; this form: 97 against 107-112 us.
__device__ __forceinline__ void glds16(const float* base, unsigned byte_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_off), "s"(base), "s"(lds_dst)
                 : "memory");
}

template <int ITEMS>
__global__ void __launch_bounds__(512) gn_bwd_pipe_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int hw, int c, int groups, int gb, int pl, int act, float drop_p,
                                                          unsigned long long seed,
                                                          const unsigned long long* __restrict__ seed_dev,
                                                          float* __restrict__ dx, float* __restrict__ part, int slabs,
                                                          int units, int per, float* __restrict__ csum_img, int ld_img) {
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];   // [ITEMS][waves][1 KB] x image | thread sums | channel / group sums | (mean, rstd) per slab
    if (seed_dev) seed += seed_dev[0];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    const int cpg = c / groups;
    const int cw = gb * cpg, cq = cw >> 2;
    const int q = tid % cq, l = tid / cq;
    const unsigned img0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)psm;
    const unsigned my_img = img0 + wave * 1024;                         // + i * nw * 1024 (+ lane * 16 by the hardware)
    const unsigned char* my_x = psm + wave * 1024 + lane * 16;
    float* red = reinterpret_cast<float*>(psm + (size_t)ITEMS * nw * 1024);
    double* chs = reinterpret_cast<double*>(red + (long long)pl * cq * 8);      // [cw][2]
    double* grp = chs + cw * 2;                                                  // [gb][2]
    float* red3 = reinterpret_cast<float*>(grp + gb * 2);                        // [pl][cq][4] thread sums of xhat (column sums of dx)
    float* cs1 = red3 + (long long)pl * cq * 4;                                  // [cw] sum_p dz, [cw] sum_p xhat
    float* cs3 = cs1 + cw;
    float* tab = cs3 + cw;                                                       // [per][gb][2]
    const unsigned istride = (unsigned)pl * c * 4;                               // bytes between a thread's items
    const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const double cnt = (double)cpg * hw;
    // gridDim.x is a multiple of the slabs per image: this workgroup's slabs cover the same channels of `per` images.
    // Everything that hipcc loads itself is loaded (and awaited) HERE: inside the loop the loads of dy and of the image are
    // asm whose waits are placed by hand - hipcc does not count them, and a counted wait of its own for an older load
    // would drain the image loads.
    const int slab = blockIdx.x % slabs, n0 = blockIdx.x / slabs, nstep = gridDim.x / slabs;
    const int c0 = slab * cw, ch0 = c0 + q * 4;
    const int gl = (q * 4) / cpg;                                                // group of this quad inside the slab
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + ch0), be = *reinterpret_cast<const f32x4*>(beta + ch0);
    const int ri = tid < cq * 8 ? tid : 0;                                       // reduction column of this thread
    const float rgam = gamma[c0 + (ri >> 3) * 4 + (ri & 3)];
    const float cgam = gamma[c0 + (tid < cw ? tid : 0)];                         // (column sums of dx: one thread per channel)
    if (tid < per * gb) {
        const int k = tid / gb, j = tid - k * gb, n = n0 + k * nstep;
        const bool in = (long long)n * slabs + slab < units;
        tab[tid * 2 + 0] = in ? mean[n * groups + slab * gb + j] : 0.f;
        tab[tid * 2 + 1] = in ? rstd[n * groups + slab * gb + j] : 0.f;
    }
    asm volatile("" ::"v"(ga), "v"(be), "v"(rgam), "v"(cgam));
    __syncthreads();
    const unsigned nbytes = (unsigned)nstep * hw * c * 4;                        // from a slab to this workgroup's next
    unsigned o = (unsigned)((((long long)n0 * hw + l) * c + ch0) * 4);           // this thread's first element (tensors < 4 GB)
    auto x_to_image = [&](unsigned at) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            glds16(x, at + i * istride, __builtin_amdgcn_readfirstlane(my_img + i * nw * 1024));
    };
    x_to_image(o);
    for (int k = 0; k < per; ++k, o += nbytes) {
        const int n = n0 + k * nstep;
        if ((long long)n * slabs + slab >= units) break;
        [[maybe_unused]] const bool stamped = k == 1;            // (ablation library: time stamps of the second slab)
        GNP_STAMP(0);
        const float mu = tab[(k * gb + gl) * 2], rs = tab[(k * gb + gl) * 2 + 1];
        f32x4 xv[ITEMS], gv[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(gv[i]) : "v"(o + i * istride), "s"(dy) : "memory");
        GNP_STAMP(1);
        // the x image of this slab: everything older than the ITEMS dy loads has landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(ITEMS) : "memory");
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) xv[i] = *reinterpret_cast<const f32x4*>(my_x + (size_t)i * nw * 1024);
        GNP_STAMP(2);
        float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, s3[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(gv[i]) : "i"(ITEMS - 1 - i));       // dy item i
            const long long idx = (long long)((o + i * istride) >> 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[i][e] - mu) * rs;
                float dz = gv[i][e];
                if (drop_p > 0.f) dz = psld_dropout_keep(seed, (unsigned long long)(idx + e), drop_p) ? dz * keep_scale : 0.f;
                if (act) dz *= dsilu_f(xh * ga[e] + be[e]);
                s1[e] += dz;
                s2[e] += dz * xh;
                s3[e] += xh;
                xv[i][e] = xh;
                gv[i][e] = dz;
            }
        }
        GNP_STAMP(3);
        // this thread's part of the image has been read (the values above depend on it): the next slab's x may land there,
        // in flight during the barriers, the reduction, pass 2 and the stores
        if (k + 1 < per && (long long)(n + nstep) * slabs + slab < units) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            x_to_image(o + nbytes);
        }
        GNP_STAMP(4);
        float* my = red + ((long long)l * cq + q) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            my[e] = s1[e];
            my[4 + e] = s2[e];
        }
        if (csum_img) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red3[((long long)l * cq + q) * 4 + e] = s3[e];
        }
        lds_barrier();
        GNP_STAMP(5);
        if (tid < cq * 8) {
            double acc = 0;
            for (int ll = 0; ll < pl; ++ll) acc += (double)red[(long long)ll * cq * 8 + tid];
            const int qq = tid >> 3, kk = tid & 7;
            const int chl = qq * 4 + (kk & 3);
            const float rounded = (float)acc;
            part[(((long long)n * 2 + (kk >> 2))) * c + c0 + chl] = rounded;
            chs[chl * 2 + (kk >> 2)] = (double)rounded * (double)rgam;
            if (csum_img && kk < 4) cs1[chl] = rounded;
        } else if (csum_img && tid < cq * 12) {
            const int j = tid - cq * 8;
            double acc = 0;
            for (int ll = 0; ll < pl; ++ll) acc += (double)red3[(long long)ll * cq * 4 + j];
            cs3[j] = (float)acc;
        }
        lds_barrier();
        if (tid < gb * 2) {
            const int gg = tid >> 1, w = tid & 1;
            double t = 0;
            for (int i = 0; i < cpg; ++i) t += chs[(gg * cpg + i) * 2 + w];
            grp[gg * 2 + w] = t;
        }
        lds_barrier();
        GNP_STAMP(6);
        const float k1 = (float)((double)rs * grp[gl * 2 + 0] / cnt), k2 = (float)((double)rs * grp[gl * 2 + 1] / cnt);
        f32x4 k0;
#pragma unroll
        for (int e = 0; e < 4; ++e) k0[e] = rs * ga[e];
        if (csum_img && tid < cw) {  // one thread per channel of the slab, with the coefficients the stores below use
            const int gc = tid / cpg;
            const float rc = tab[(k * gb + gc) * 2 + 1];
            const float c0k = rc * cgam;
            const float c1k = (float)((double)rc * grp[gc * 2 + 0] / cnt), c2k = (float)((double)rc * grp[gc * 2 + 1] / cnt);
            csum_img[(long long)n * ld_img + c0 + tid] =
                (float)((double)c0k * (double)cs1[tid] - (double)hw * (double)c1k - (double)c2k * (double)cs3[tid]);
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = gn_dx(k0[e], gv[i][e], k1, xv[i][e], k2);
            *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(dx) + (o + i * istride)) = r;
        }
        GNP_STAMP(7);
        // (no barrier here: a wave that rewrites the thread sums has passed the barrier behind their last read)
    }
}

// ---- the one-pass backward on WHOLE ROWS: teams of resident workgroups -----------------------------------------------------
// Round 5 (profiles/r05/gnb_rows.md): the one-slab kernels read a 128-byte segment (32 channels) out of every 1 KB NHWC row
// and reach 4.3-4.6 TB/s on 128x32x32x256; the elementwise apply pass of the three-pass form, which reads whole rows, moves
// the same bytes at 5.9-6.4 TB/s.  The slab shape was forced by the reduction - all pixels of an (image, group) in one
// workgroup's registers.  Here a workgroup owns 64 pixels x 128 channels of an image (512-byte segments of consecutive rows:
// 256 threads = 32 channel quads x 8 pixel lanes, 8 items per thread), and the K = hw / 64 workgroups of an (image, 128-
// channel block) form a TEAM that exchanges its per-group partial sums through memory:
//   pass 1 as ever (dz, xhat in place; per-thread fp32 sums, fp64 across the 8 pixel lanes) -> the block's per-channel sums
//   go to part[image][k][2][c] (dgamma / dbeta are sums over images AND team members: psld_param_reduce*_f32 takes rows);
//   their gamma-weighted group sums are PUBLISHED as 64-bit slots {fp32 value, 32-bit tag} by relaxed agent-scope atomic
//   stores; every member polls the K x (2 x groups-in-block) slots of its team until they carry this set's tag, adds them in
//   member order (fp64) - every member gets bitwise the same group terms - and writes dx from its registers.
// A slot is one atomic object that carries its own validity: no fence, no counter, nothing to reset.  Slots are double-
// buffered by set parity: a member overwrites set s's slots only when it publishes s + 2, i.e. after it has read every
// member's s + 1 slots, which they published after reading all of set s.  Tags rise with every set of every launch: the
// launch's first tag is a counter in the slot buffer (word 1) that the LAST workgroup to finish advances (word 2 counts the
// finished ones) - nothing for the host to hand out, and a hipGraph replay of the launch gets fresh tags like any other.
// The buffer is zeroed once, when it is allocated, and belongs to one stream at a time.
// The grid is RESIDENT (teams x K workgroups <= what the device holds at once, teams walk their sets in lock step), so a
// member only ever waits for workgroups that are running or will run without anything finishing first.  The poll loop is
// bounded all the same: on a timeout (another process hogging the device for seconds) the workgroup raises an error word in
// the slot buffer and carries on with garbage rather than hang the queue.
// LANES = pixel lanes of a workgroup: 8 (256 threads, 64 pixels per set: maps up to 32x32) or 16 (512 threads, 128 pixels:
// 64x64 maps, where the alternative is the three-pass form at 20+ bytes per element; the exchange is half as large).
constexpr int TEAM_CH = 128;         // channels per workgroup
constexpr int TEAM_ITEMS = 8;
constexpr int TEAM_MAXK = 32;        // team members: hw <= 1024 at 64 pixels, hw <= 4096 at 128 pixels per member
constexpr int TEAM_SLOTS = 64;       // 2 sums x up to 32 groups in a 128-channel block
constexpr int TEAM_MAXTEAMS = 512;
constexpr long long TEAM_SYNC_BYTES = 256 + (long long)TEAM_MAXTEAMS * 2 * TEAM_MAXK * TEAM_SLOTS * 8;     // error word, tag counter + slots

template <bool THIRD, int LANES>
__global__ void __launch_bounds__(LANES * 32) gn_bwd_team_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int hw, int c, int groups, int act, float drop_p,
                                                          unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                                          int accumulate, const float* __restrict__ add, float add_scale,
                                                          float* __restrict__ dx, float* __restrict__ part,
                                                          float* __restrict__ colsum_rows, int ld_rows,
                                                          unsigned long long* __restrict__ sync, int rounds, int K, int sets,
                                                          int teams) {
    constexpr int TEAM_PX = LANES * TEAM_ITEMS, THREADS = LANES * 32;
    __shared__ float red[LANES * 32 * 8];        // [pixel lane][quad][s1 x4 | s2 x4]; later [lane][quad][4] column sums
    __shared__ double chs[TEAM_CH * 2];          // gamma-weighted channel sums of this block
    __shared__ float gp[TEAM_MAXK * TEAM_SLOTS]; // the team's group partials [member][slot]
    __shared__ double grp[TEAM_SLOTS];           // the image's group terms [group in block][2]
    if (seed_dev) seed += seed_dev[0];
    const int tid = threadIdx.x;
    const int q = tid & 31, l = tid >> 5;        // channel quad of the block, pixel lane
    const int team = blockIdx.x / K, k = blockIdx.x - team * K;
    // first tag of this launch (the counter moves only after every workgroup of the launch has finished, see the end)
    const unsigned tag0 = (unsigned)__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int cblocks = c / TEAM_CH;
    const int cpg = c / groups, gpb = TEAM_CH / cpg, nslot = 2 * gpb;
    const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const double cnt = (double)cpg * hw;
    unsigned long long* slots = sync + 32 + (long long)team * 2 * TEAM_MAXK * TEAM_SLOTS;
    for (int s = team, it = 0; s < sets; s += teams, ++it) {
        const int n = s / cblocks, cb = s - n * cblocks;
        const int c0 = cb * TEAM_CH, ch0 = c0 + q * 4;
        const int g = ch0 / cpg;                 // cpg % 4 == 0: a quad lies in one group
        const float mu = mean[n * groups + g], rs = rstd[n * groups + g];
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + ch0), be = *reinterpret_cast<const f32x4*>(beta + ch0);
        // pixels k*TEAM_PX + l + LANES i: a wave (2 pixel lanes x 32 quads) reads two 512-byte runs
        const long long e0 = (((long long)n * hw + k * TEAM_PX + l) * c + ch0);
        const long long istride = (long long)LANES * c;
        f32x4 xv[TEAM_ITEMS], gv[TEAM_ITEMS];
#pragma unroll
        for (int i = 0; i < TEAM_ITEMS; ++i) {
            xv[i] = *reinterpret_cast<const f32x4*>(x + e0 + i * istride);
            gv[i] = *reinterpret_cast<const f32x4*>(dy + e0 + i * istride);
        }
        float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < TEAM_ITEMS; ++i) {
            const long long idx = e0 + i * istride;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[i][e] - mu) * rs;
                float dz = gv[i][e];
                if (drop_p > 0.f) dz = psld_dropout_keep(seed, (unsigned long long)(idx + e), drop_p) ? dz * keep_scale : 0.f;
                if (act) dz *= dsilu_f(xh * ga[e] + be[e]);
                s1[e] += dz;
                s2[e] += dz * xh;
                xv[i][e] = xh;
                gv[i][e] = dz;
            }
        }
        float* my = red + (l * 32 + q) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            my[e] = s1[e];
            my[4 + e] = s2[e];
        }
        __syncthreads();
        if (tid < 2 * TEAM_CH) {   // 128 channels x 2 sums: the pixel lanes in lane order
            const int chl = tid >> 1, w = tid & 1;
            double acc = 0;
#pragma unroll
            for (int ll = 0; ll < LANES; ++ll) acc += (double)red[(ll * 32 + (chl >> 2)) * 8 + w * 4 + (chl & 3)];
            const float rounded = (float)acc;
            part[(((long long)n * K + k) * 2 + w) * c + c0 + chl] = rounded;
            chs[chl * 2 + w] = (double)rounded * (double)gamma[c0 + chl];
        }
        __syncthreads();
        const unsigned tag = tag0 + (unsigned)it;
        unsigned long long* buf = slots + (long long)(it & 1) * TEAM_MAXK * TEAM_SLOTS;
        if (tid < nslot) {
            const int gg = tid >> 1, w = tid & 1;
            double t = 0;
            for (int i = 0; i < cpg; ++i) t += chs[(gg * cpg + i) * 2 + w];
            const unsigned long long word = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint((float)t);
            __hip_atomic_store(buf + k * TEAM_SLOTS + tid, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // every member's slots of this set
        for (int j = tid; j < K * nslot; j += THREADS) {
            const int kk = j / nslot, sl = j - kk * nslot;
            const unsigned long long* ptr = buf + kk * TEAM_SLOTS + sl;
            unsigned long long word = __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while ((unsigned)(word >> 32) != tag) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) {       // seconds: give up rather than hang the queue
                    __hip_atomic_store(sync, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                word = __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            gp[kk * TEAM_SLOTS + sl] = __uint_as_float((unsigned)word);
        }
        __syncthreads();
        if (tid < nslot) {
            double t = 0;
            for (int kk = 0; kk < K; ++kk) t += (double)gp[kk * TEAM_SLOTS + tid];
            grp[tid] = t;
        }
        __syncthreads();
        const int gl = (q * 4) / cpg;
        const float k1 = (float)((double)rs * grp[gl * 2 + 0] / cnt), k2 = (float)((double)rs * grp[gl * 2 + 1] / cnt);
        f32x4 k0;
#pragma unroll
        for (int e = 0; e < 4; ++e) k0[e] = rs * ga[e];
#pragma unroll
        for (int i = 0; i < TEAM_ITEMS; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[i][e] = gn_dx(k0[e], gv[i][e], k1, xv[i][e], k2);
        if constexpr (THIRD) {
            if (add) {
#pragma unroll
                for (int i = 0; i < TEAM_ITEMS; ++i) xv[i] = *reinterpret_cast<const f32x4*>(add + e0 + i * istride);
#pragma unroll
                for (int i = 0; i < TEAM_ITEMS; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) gv[i][e] += add_scale * xv[i][e];
            }
            if (accumulate) {
#pragma unroll
                for (int i = 0; i < TEAM_ITEMS; ++i) xv[i] = *reinterpret_cast<const f32x4*>(dx + e0 + i * istride);
#pragma unroll
                for (int i = 0; i < TEAM_ITEMS; ++i) gv[i] += xv[i];
            }
        }
#pragma unroll
        for (int i = 0; i < TEAM_ITEMS; ++i) *reinterpret_cast<f32x4*>(dx + e0 + i * istride) = gv[i];
        if (colsum_rows) {       // column sums of the values just written, over this block's pixels (red: free since the barrier above)
            float s4[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < TEAM_ITEMS; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) s4[e] += gv[i][e];
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(l * 32 + q) * 4 + e] = s4[e];
            __syncthreads();
            if (tid < TEAM_CH) {
                double acc = 0;
#pragma unroll
                for (int ll = 0; ll < LANES; ++ll) acc += (double)red[ll * TEAM_CH + tid];
                colsum_rows[((long long)n * K + k) * ld_rows + c0 + tid] = (float)acc;
            }
        }
        __syncthreads();         // red / chs / gp / grp are rewritten by the next set
    }
    if (tid == 0) {              // the last workgroup out advances the tag counter for the next launch
        const unsigned long long done = __hip_atomic_fetch_add(sync + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
        if (done == (unsigned long long)gridDim.x) {
            __hip_atomic_store(sync + 2, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(sync + 1, (unsigned long long)rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// team size (= rows per image of `part`) and pixel lanes per member, or 0 when the team form does not take the shape
inline int gn_bwd_team_k(int batch, int hw, int c, int groups, int* lanes) {
    if (batch <= 0 || groups <= 0 || c % groups) return 0;
    const int cpg = c / groups;
    if (c % TEAM_CH || cpg % 4 || TEAM_CH % cpg || 2 * (TEAM_CH / cpg) > TEAM_SLOTS) return 0;
    const int ln = hw > 1024 ? 16 : 8, px = ln * TEAM_ITEMS;
    if (hw % px || hw / px < 2 || hw / px > TEAM_MAXK) return 0;
    *lanes = ln;
    return hw / px;
}

// resident grid: teams (each K workgroups) and the sets every team walks
inline bool gn_bwd_team_grid(int batch, int c, int K, int lanes, bool third, int* teams, int* rounds) {
    struct Slot { int per_cu[4] = {-1, -1, -1, -1}; int cus = 0; };
    static Slot slots[PSLD_MAX_DEVICES];
    Slot& sl = slots[psld_device_slot()];
    int& per_cu = sl.per_cu[(third ? 1 : 0) + (lanes == 16 ? 2 : 0)];
    if (per_cu < 0) {
        const void* fn = lanes == 16 ? (third ? reinterpret_cast<const void*>(&gn_bwd_team_kernel<true, 16>)
                                              : reinterpret_cast<const void*>(&gn_bwd_team_kernel<false, 16>))
                                     : (third ? reinterpret_cast<const void*>(&gn_bwd_team_kernel<true, 8>)
                                              : reinterpret_cast<const void*>(&gn_bwd_team_kernel<false, 8>));
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, lanes * 32, 0) != hipSuccess) per_cu = 0;
    }
    if (sl.cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&sl.cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            sl.cus <= 0)
            sl.cus = -1;
    }
    if (per_cu <= 0 || sl.cus <= 0) return false;
    const int sets = batch * (c / TEAM_CH);
    int max_teams = (int)((long long)per_cu * sl.cus / K);
    if (max_teams > TEAM_MAXTEAMS) max_teams = TEAM_MAXTEAMS;
    if (max_teams < 1) return false;
    *rounds = cdiv(sets, max_teams);
    *teams = cdiv(sets, *rounds);            // equal shares: every team walks `rounds` sets (the last ones one fewer)
    return true;
}

// groups per block / pixel lanes / items per thread of the fused backward, or false when the slab does not fit registers
inline bool gn_bwd_fused_plan(int batch, int hw, int c, int groups, int* gb, int* pl, int* items) {
    const int cpg = c / groups;
    if (cpg % 4 || cpg > 32 || (long long)batch * hw * c * 4 >= (1ll << 32)) return false;
    int g = 32 / cpg;                              // as many whole groups as fit 32 channels (whole 128-byte lines; 16-channel
                                                   // slabs measured 136 vs 112 us on 128x32x32x256) ...
    while (g > 1 && groups % g) --g;               // ... dividing the group count
    const int cq = g * cpg / 4;
    const int lanes = 512 / cq;
    const int it = cdiv(hw, lanes);
    if (it > 16) return false;
    *gb = g;
    *pl = lanes;
    *items = it <= 1 ? 1 : it <= 2 ? 2 : it <= 4 ? 4 : 16;      // (5..8 pixels per thread - non-square maps only - run the 16-item kernel half empty)
    return true;
}

// PSLD_GN_BWD_AUTO / PSLD_GN_BWD_ONE_SLAB (psld_set_gn_bwd_kernel); initial value from PSLD_GN_BWD_PIPE
inline int& gn_bwd_kind() {
    static int kind = [] { const char* v = getenv("PSLD_GN_BWD_PIPE"); return v && atoi(v) == 0 ? PSLD_GN_BWD_ONE_SLAB : PSLD_GN_BWD_AUTO; }();
    return kind;
}

// The resident form: when every workgroup would walk at least two slabs.  Returns the grid (all workgroups resident, equal
// shares) or false.  items in {4, 16} with hw == items * pl exactly (no guarded items: the counted vmcnt needs them all).
template <int ITEMS>
inline bool gn_bwd_pipe_setup(size_t lds, int threads, int* per_cu) {
    // per device and per (threads, lds) of the last query: hipFuncSetAttribute and the occupancy belong to ONE device, and
    // the occupancy to one launch shape (510 vs 512 threads, LDS sizes differ by shape)
    struct Slot { size_t configured = 0; int blocks = -1; int threads = 0; size_t lds = 0; };
    static Slot slots[PSLD_MAX_DEVICES];
    Slot& s = slots[psld_device_slot()];
    if (lds > s.configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_bwd_pipe_kernel<ITEMS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return false;
        s.configured = lds;
        s.blocks = -1;
    }
    if (s.blocks < 0 || s.threads != threads || s.lds != lds) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&s.blocks, reinterpret_cast<const void*>(&gn_bwd_pipe_kernel<ITEMS>),
                                                         threads, lds) != hipSuccess)
            s.blocks = 0;
        s.threads = threads;
        s.lds = lds;
    }
    *per_cu = s.blocks;
    return s.blocks > 0;
}

inline bool gn_bwd_pipe_plan(int batch, int hw, int c, int slabs, int gb, int threads, int pl, int items, size_t fused_lds,
                             int* grid, int* per_out, size_t* lds_out) {
    if (gn_bwd_kind() != PSLD_GN_BWD_AUTO) return false;
    constexpr int MAXPER = 8;
    if ((items != 4 && items != 16) || hw != items * pl || (long long)batch * hw * c * 4 >= (1ll << 32)) return false;
    const size_t lds = (size_t)items * ((threads + 63) / 64) * 1024 + fused_lds + (size_t)MAXPER * gb * 2 * sizeof(float);
    if (lds > 160 * 1024) return false;
    int per_cu = 0;
    const bool ok = items == 4 ? gn_bwd_pipe_setup<4>(lds, threads, &per_cu) : gn_bwd_pipe_setup<16>(lds, threads, &per_cu);
    if (!ok) return false;
    static int cu_count[PSLD_MAX_DEVICES] = {};       // per device (0 = not asked yet, -1 = the query failed)
    int& cus = cu_count[psld_device_slot()];
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus <= 0)
            cus = -1;
    }
    if (cus <= 0) return false;
    // whole images per round: the grid is a multiple of the slabs per image (a workgroup keeps its channels), everything resident
    const int images_per_round = cus * per_cu / slabs;
    if (images_per_round < 1) return false;
    const int per = cdiv(batch, images_per_round);        // slabs per workgroup
    if (per < 2 || per > MAXPER || per * gb > threads) return false;      // one slab: nothing to prefetch for - the one-slab kernel
    *grid = cdiv(batch, per) * slabs;
    *per_out = per;
    *lds_out = lds;
    return true;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" long long psld_gn_workspace_bytes(int batch, int hw, int c, int groups) {
    if (batch <= 0 || hw <= 0 || c <= 0 || c % 4) return 0;
    const Map m = make_map(batch, hw, c);
    size_t fwd = align256((size_t)batch * m.chunks * groups * 2 * sizeof(double));
    size_t bwd = align256((size_t)batch * m.chunks * 2 * c * sizeof(float)) +
                 align256((size_t)batch * 2 * c * sizeof(float)) + align256((size_t)batch * 3 * c * sizeof(float));
    return (long long)(fwd > bwd ? fwd : bwd);
}

extern "C" int psld_gn_stats_nhwc_f32(const float* x, int batch, int hw, int c, int groups, float eps,
                                      const float* gamma, const float* beta, float* mean, float* rstd,
                                      float* scale, float* shift, void* workspace, hipStream_t stream) {
    PSLD_CHECK_ARG(x && gamma && beta && mean && rstd && scale && shift && workspace, "psld_gn_stats: null pointer");
    PSLD_CHECK_ARG(c % 4 == 0 && c / 4 <= MAXT && groups <= MAXG && c % groups == 0,
                   "psld_gn_stats: unsupported C=%d groups=%d", c, groups);
    const Map m = make_map(batch, hw, c);
    double* part = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(m.chunks, batch), dim3(m.threads), 0, stream, x, hw, c, groups, m.cq,
                       m.pl, m.chunk_px, m.chunks, part);
    PSLD_CHECK_LAUNCH("gn_partial_kernel");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(batch), dim3(256), 0, stream, part, hw, c, groups, m.chunks, 1, eps,
                       gamma, beta, mean, rstd, scale, shift);
    PSLD_CHECK_LAUNCH("gn_finalize_kernel");
    return PSLD_OK;
}

extern "C" int psld_gn_stats_from_partials_f32(const double* gn_part, int fine_width, int batch, int hw, int c, int groups,
                                               float eps, const float* gamma, const float* beta, float* mean, float* rstd,
                                               float* scale, float* shift, hipStream_t stream) {
    PSLD_CHECK_ARG(gn_part && gamma && beta && mean && rstd && scale && shift, "psld_gn_stats_from_partials: null pointer");
    PSLD_CHECK_ARG((fine_width == 8 || fine_width == 4) && groups > 0 && groups <= MAXG && c % groups == 0 &&
                       (c / groups) % fine_width == 0 && hw % 64 == 0 && hw > 0,
                   "psld_gn_stats_from_partials: unsupported C=%d groups=%d hw=%d fine_width=%d", c, groups, hw, fine_width);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(batch), dim3(256), 0, stream, gn_part, hw, c, groups, hw / 64,
                       (c / groups) / fine_width, eps, gamma, beta, mean, rstd, scale, shift);
    PSLD_CHECK_LAUNCH("gn_finalize_kernel");
    return PSLD_OK;
}

extern "C" int psld_gn_apply_nhwc_f32(const float* x, const float* scale, const float* shift, float* y, int batch,
                                      int hw, int c, int act, float drop_p, unsigned long long seed,
                                      const unsigned long long* seed_dev, hipStream_t stream) {
    PSLD_CHECK_ARG(x && scale && shift && y, "psld_gn_apply: null pointer");
    PSLD_CHECK_ARG(c % 4 == 0 && c / 4 <= MAXT, "psld_gn_apply: unsupported C=%d", c);
    const Map m = make_map(batch, hw, c, true);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(m.chunks, batch), dim3(m.threads), 0, stream, x, scale, shift, y, hw, c,
                       m.cq, m.pl, m.chunk_px, act, drop_p, seed, seed_dev);
    PSLD_CHECK_LAUNCH("gn_apply_kernel");
    return PSLD_OK;
}

extern "C" int psld_gn_apply_limb_nhwc(const float* x, const float* scale, const float* shift, void* y_limb, int batch,
                                       int hw, int c, int act, float drop_p, unsigned long long seed,
                                       const unsigned long long* seed_dev, hipStream_t stream) {
    PSLD_CHECK_ARG(x && scale && shift && y_limb, "psld_gn_apply_limb: null pointer");
    PSLD_CHECK_ARG(c % 32 == 0 && c / 4 <= MAXT, "psld_gn_apply_limb: unsupported C=%d (needs a multiple of 32, <= 1024)", c);
    const Map m = make_map(batch, hw, c, true);
    hipLaunchKernelGGL(gn_apply_limb_kernel, dim3(m.chunks, batch), dim3(m.threads), 0, stream, x, scale, shift,
                       reinterpret_cast<unsigned char*>(y_limb), hw, c, m.cq, m.pl, m.chunk_px, act, drop_p, seed, seed_dev);
    PSLD_CHECK_LAUNCH("gn_apply_limb_kernel");
    return PSLD_OK;
}

extern "C" int psld_gn_bwd_nhwc_f32(const float* dy, const float* x, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, int batch, int hw, int c, int groups,
                                    int act, float drop_p, unsigned long long seed, const unsigned long long* seed_dev, float* dx,
                                    int accumulate_dx, const float* add, float add_scale, float* sums, float* colsum_img,
                                    int ld_img, void* workspace, hipStream_t stream) {
    PSLD_CHECK_ARG(dy && x && mean && rstd && gamma && beta && dx && sums && workspace, "psld_gn_bwd: null pointer");
    PSLD_CHECK_ARG(c % 4 == 0 && c / 4 <= MAXT && groups <= MAXG && c % groups == 0,
                   "psld_gn_bwd: unsupported C=%d groups=%d", c, groups);
    PSLD_CHECK_ARG(!colsum_img || ld_img >= c, "psld_gn_bwd: ld_img < c");
    int gb = 0, fpl = 0, items = 0;
    if (gn_bwd_fused_plan(batch, hw, c, groups, &gb, &fpl, &items)) {
        const int cpg = c / groups, cw = gb * cpg, cq = cw / 4;
        const dim3 grid(groups / gb, batch), block(cq * fpl);
        const size_t flds = (size_t)fpl * cq * 8 * sizeof(float) + (size_t)(cw + gb) * 2 * sizeof(double) +
                            ((size_t)fpl * cq * 4 + 2 * cw) * sizeof(float);
        int pipe_grid = 0, pipe_per = 0;
        size_t plds = 0;
        if (!add && !accumulate_dx &&       // (a third operand: the one-slab kernel, see gn_bwd_pipe_kernel)
            gn_bwd_pipe_plan(batch, hw, c, groups / gb, gb, (int)block.x, fpl, items, flds, &pipe_grid, &pipe_per, &plds)) {
            const int slabs = groups / gb;
#define PSLD_GN_PIPE(IT)                                                                                               \
    hipLaunchKernelGGL((gn_bwd_pipe_kernel<IT>), dim3(pipe_grid), block, plds, stream, dy, x, mean, rstd, gamma, beta, hw, \
                       c, groups, gb, fpl, act, drop_p, seed, seed_dev, dx, sums, slabs, slabs * batch, pipe_per, colsum_img, \
                       ld_img)
            if (items == 4) PSLD_GN_PIPE(4); else PSLD_GN_PIPE(16);
#undef PSLD_GN_PIPE
            PSLD_CHECK_LAUNCH("gn_bwd_pipe_kernel");
            return PSLD_OK;
        }
#define PSLD_GN_FUSED(IT, EARLY)                                                                                       \
    hipLaunchKernelGGL((gn_bwd_fused_kernel<IT, EARLY>), grid, block, flds, stream, dy, x, mean, rstd, gamma, beta, hw, c, \
                       groups, gb, fpl, act, drop_p, seed, seed_dev, accumulate_dx, add, add_scale, dx, sums, colsum_img,  \
                       ld_img)
        // a third operand on the larger slabs: asked for before the reduction (gn_bwd_fused_kernel<., EARLY>)
        const bool early = (add || accumulate_dx) && gn_bwd_kind() == PSLD_GN_BWD_AUTO;
        switch (items) {
            case 1: PSLD_GN_FUSED(1, false); break;
            case 2: PSLD_GN_FUSED(2, false); break;
            case 4: if (early) PSLD_GN_FUSED(4, true); else PSLD_GN_FUSED(4, false); break;
            default: if (early) PSLD_GN_FUSED(16, true); else PSLD_GN_FUSED(16, false); break;
        }
#undef PSLD_GN_FUSED
        PSLD_CHECK_LAUNCH("gn_bwd_fused_kernel");
        return PSLD_OK;
    }
    PSLD_CHECK_ARG(!colsum_img, "psld_gn_bwd: column sums of dx are not formed on the three-pass path (psld_gn_bwd_colsum_supported)");
    const Map m = make_map(batch, hw, c);
    char* ws = reinterpret_cast<char*>(workspace);
    float* part = reinterpret_cast<float*>(ws);
    ws += align256((size_t)batch * m.chunks * 2 * c * sizeof(float));
    ws += align256((size_t)batch * 2 * c * sizeof(float));
    float* coef = reinterpret_cast<float*>(ws);
    const size_t lds = (size_t)m.pl * m.cq * 8 * sizeof(float);
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(m.chunks, batch), dim3(m.threads), lds, stream, dy, x, mean, rstd,
                       gamma, beta, hw, c, groups, m.cq, m.pl, m.chunk_px, m.chunks, act, drop_p, seed, seed_dev, part);
    PSLD_CHECK_LAUNCH("gn_bwd_partial_kernel");
    hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3(batch), dim3(1024), 0, stream, part, rstd, gamma, hw, c, groups, m.chunks,
                       coef, sums);
    PSLD_CHECK_LAUNCH("gn_bwd_coef_kernel");
    const Map ma = make_map(batch, hw, c, true);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(ma.chunks, batch), dim3(ma.threads), 0, stream, dy, x, mean, rstd,
                       gamma, beta, coef, hw, c, groups, ma.cq, ma.pl, ma.chunk_px, act, drop_p, seed, seed_dev, accumulate_dx, add, add_scale,
                       dx);
    PSLD_CHECK_LAUNCH("gn_bwd_apply_kernel");
    return PSLD_OK;
}

extern "C" int psld_gn_bwd_team_rows(int batch, int hw, int c, int groups) {
    if (gn_bwd_kind() != PSLD_GN_BWD_AUTO || (long long)batch * hw * c >= (1ll << 40)) return 0;
    int lanes = 0;
    const int K = gn_bwd_team_k(batch, hw, c, groups, &lanes);
    int teams = 0, rounds = 0;
    return K > 0 && gn_bwd_team_grid(batch, c, K, lanes, true, &teams, &rounds) && gn_bwd_team_grid(batch, c, K, lanes, false, &teams, &rounds) ? K : 0;
}

extern "C" long long psld_gn_bwd_team_sync_bytes(void) { return TEAM_SYNC_BYTES; }

extern "C" int psld_gn_bwd_team_f32(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                    const float* beta, int batch, int hw, int c, int groups, int act, float drop_p,
                                    unsigned long long seed, const unsigned long long* seed_dev, float* dx, int accumulate_dx,
                                    const float* add, float add_scale, float* sums, float* colsum_rows, int ld_rows, void* sync,
                                    hipStream_t stream) {
    PSLD_CHECK_ARG(dy && x && mean && rstd && gamma && beta && dx && sums && sync, "psld_gn_bwd_team: null pointer");
    int lanes = 0;
    const int K = gn_bwd_team_k(batch, hw, c, groups, &lanes);
    PSLD_CHECK_ARG(K > 0, "psld_gn_bwd_team: unsupported shape B=%d hw=%d C=%d groups=%d (psld_gn_bwd_team_rows)", batch, hw, c, groups);
    PSLD_CHECK_ARG(!colsum_rows || ld_rows >= c, "psld_gn_bwd_team: ld_rows < c");
    PSLD_CHECK_ARG((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) |
                    reinterpret_cast<uintptr_t>(add) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(sync) % 8 == 0,
                   "psld_gn_bwd_team: unaligned pointer");
    const bool third = add != nullptr || accumulate_dx != 0;
    int teams = 0, rounds = 0;
    PSLD_CHECK_ARG(gn_bwd_team_grid(batch, c, K, lanes, third, &teams, &rounds), "psld_gn_bwd_team: no resident grid");
    const int sets = batch * (c / TEAM_CH);
    unsigned long long* sy = reinterpret_cast<unsigned long long*>(sync);
#define PSLD_GN_TEAM(THIRD, LANES)                                                                                     \
    hipLaunchKernelGGL((gn_bwd_team_kernel<THIRD, LANES>), dim3(teams * K), dim3(LANES * 32), 0, stream, dy, x, mean, rstd,  \
                       gamma, beta, hw, c, groups, act, drop_p, seed, seed_dev, accumulate_dx, add, add_scale, dx, sums,     \
                       colsum_rows, ld_rows, sy, rounds, K, sets, teams)
    if (lanes == 16) {
        if (third) PSLD_GN_TEAM(true, 16); else PSLD_GN_TEAM(false, 16);
    } else {
        if (third) PSLD_GN_TEAM(true, 8); else PSLD_GN_TEAM(false, 8);
    }
#undef PSLD_GN_TEAM
    PSLD_CHECK_LAUNCH("gn_bwd_team_kernel");
    return PSLD_OK;
}

extern "C" int psld_gn_bwd_colsum_supported(int batch, int hw, int c, int groups) {
    int gb = 0, pl = 0, items = 0;
    return batch > 0 && hw > 0 && c > 0 && c % 4 == 0 && c / 4 <= MAXT && groups > 0 && groups <= MAXG && c % groups == 0 &&
           gn_bwd_fused_plan(batch, hw, c, groups, &gb, &pl, &items);
}

extern "C" int psld_set_gn_bwd_kernel(int kind) {
    PSLD_CHECK_ARG(kind == PSLD_GN_BWD_AUTO || kind == PSLD_GN_BWD_ONE_SLAB, "psld_set_gn_bwd_kernel: unknown kind %d", kind);
    gn_bwd_kind() = kind;
    return PSLD_OK;
}
extern "C" int psld_get_gn_bwd_kernel(void) { return gn_bwd_kind(); }

#ifdef PSLD_ABLATIONS
extern "C" void psld_abl_set_gnb_debug(unsigned long long* p, int mode) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gnb_dbg), &p, sizeof(p));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gnb_mode), &mode, sizeof(mode));
}
#endif
