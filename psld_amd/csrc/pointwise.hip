// Bandwidth-bound helpers: layout changes at the network boundary, weight packing, residual /
// pyramid combines, SiLU on the time embedding, column sums (bias gradients), row softmax
// (attention) and the Fourier / positional time embedding.
#include "common.h"
#include "psld_hip.h"

namespace {

inline int grid_for(long long n, int per_thread = 1) {
    long long b = (n + 256LL * per_thread - 1) / (256LL * per_thread);
    if (b > 256LL * 16) b = 256LL * 16;
    if (b < 1) b = 1;
    return (int)b;
}

#define GRID_STRIDE(i, n) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// ---- layout -----------------------------------------------------------------------------------
// per image: [C][HW] <-> [HW][C] through a 32x33 LDS tile (coalesced on both sides)
__global__ void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int cols) {
    __shared__ float tile[32][33];
    const long long img = blockIdx.z;
    const float* xp = x + img * rows * cols;
    float* yp = y + img * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        if (r < rows && c < cols) tile[j][tx] = xp[(long long)r * cols + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (r < rows && c < cols) yp[(long long)c * rows + r] = tile[tx][j];
    }
}

// OIHW -> [co][tap][ci]
__global__ void pack_ohwi_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int taps) {
    const long long n = (long long)cout * cin * taps;
    GRID_STRIDE(i, n) {
        const int ci = (int)(i % cin);
        long long t = i / cin;
        const int tap = (int)(t % taps);
        const int co = (int)(t / taps);
        out[i] = w[((long long)co * cin + ci) * taps + tap];
    }
}
// OIHW -> [ci][taps-1-tap][co]
__global__ void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int taps) {
    const long long n = (long long)cout * cin * taps;
    GRID_STRIDE(i, n) {
        const int co = (int)(i % cout);
        long long t = i / cout;
        const int tapf = (int)(t % taps);
        const int ci = (int)(t / taps);
        out[i] = w[((long long)co * cin + ci) * taps + (taps - 1 - tapf)];
    }
}

__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int nsplit, long long n, float* __restrict__ out,
                                    int layout, int taps, int cin, float alpha) {
    GRID_STRIDE(i, n) {
        float acc = 0.f;
        for (int s = 0; s < nsplit; ++s) acc += slabs[(long long)s * n + i];
        acc *= alpha;
        long long o = i;
        if (layout == 1) {  // [co][tap][ci] -> [co][ci][tap]
            const int ci = (int)(i % cin);
            long long t = i / cin;
            const int tap = (int)(t % taps);
            const long long co = t / taps;
            o = (co * cin + ci) * taps + tap;
        }
        out[o] = acc;
    }
}

// float4 form (n, cin multiples of 4, 16-byte aligned slabs): four consecutive ci per thread, all slabs' loads
// independent; the OIHW scatter writes four floats `taps` apart
__global__ void reduce_slabs4_kernel(const float* __restrict__ slabs, int nsplit, long long n, float* __restrict__ out,
                                     int layout, int taps, int cin, float alpha) {
    const long long n4 = n >> 2;
    GRID_STRIDE(q, n4) {
        const f32x4* p = reinterpret_cast<const f32x4*>(slabs) + q;
        f32x4 acc = p[0];
        int s = 1;
        for (; s + 3 < nsplit; s += 4) {        // four independent loads in flight; the sum stays in slab order
            const f32x4 v0 = p[(long long)s * n4], v1 = p[(long long)(s + 1) * n4];
            const f32x4 v2 = p[(long long)(s + 2) * n4], v3 = p[(long long)(s + 3) * n4];
            acc += v0;
            acc += v1;
            acc += v2;
            acc += v3;
        }
        for (; s < nsplit; ++s) acc += p[(long long)s * n4];
        acc *= alpha;
        const long long i = q << 2;
        if (layout == 1) {  // [co][tap][ci] -> [co][ci][tap]
            const int ci = (int)(i % cin);
            long long t = i / cin;
            const int tap = (int)(t % taps);
            const long long co = t / taps;
            float* o = out + (co * cin + ci) * taps + tap;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[(long long)e * taps] = acc[e];
        } else {
            reinterpret_cast<f32x4*>(out)[q] = acc;
        }
    }
}

// The same reduction for MANY weight gradients in one launch: rows of a device table
// [slabs, nsplit, n, out, layout, taps, cin, alpha bits, first unit, units].  A workgroup owns one UNIT of one job, found by
// binary search on the first units (block-uniform: scalar loads):
//   layout 0:  1024 consecutive float4 of the [n] vector (four per thread, 256 apart);
//   layout 2:  the same over a column block of wider slabs: slab element (r, j) of [n / cols][cols] sits at r * ld + j
//              (cols = the `taps` field, ld = the `cin` field; the slabs pointer already points at the block's first column) -
//              the three [c][c] gradients of one [c][3c] slab set (q | k | v projections computed by one GEMM);
//   layout 1:  one output channel x up to 256 input channels x all taps.  The slabs hold [co][tap][ci]; the gradient wants
//              OIHW = [co][ci][tap].  The per-thread scatter of reduce_slabs4_kernel writes single floats `taps` apart -
//              32-byte memory transactions for 4 useful bytes - so here the summed [tap][ci] tile goes through LDS and
//              leaves as the contiguous [ci][tap] run it is in the output (16-byte stores).
// Within a float4 the slabs are added in slab order exactly like reduce_slabs4_kernel: bitwise the per-layer launches.  A
// thread's items advance through the slabs together, eight slabs per round: 24 - 32 independent 16-byte loads in flight per
// thread (unrolled by 4: 1.92-2.04 ms per step in situ; by 8: 1.76; by 16: 1.75).
constexpr int RSB_FLAT4 = 1024;     // float4 per unit of a layout-0 job
constexpr int RSB_CI = 256;         // input channels per unit of a layout-1 job
constexpr int RSB_MAXTAPS = 9;

__global__ void __launch_bounds__(256) reduce_slabs_batch_kernel(const long long* __restrict__ table, int jobs) {
    __shared__ __attribute__((aligned(16))) float tile[RSB_CI * RSB_MAXTAPS];      // [ci][tap]
    int lo = 0, hi = jobs - 1;                 // last job whose first unit <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(long long)mid * 10 + 8] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* row = table + (long long)lo * 10;
    const int unit = (int)((long long)blockIdx.x - row[8]);
    const int nsplit = (int)row[1];
    const long long n4 = row[2] >> 2;
    const f32x4* slabs = reinterpret_cast<const f32x4*>(row[0]);
    float* out = reinterpret_cast<float*>(row[3]);
    const float alpha = __builtin_bit_cast(float, (unsigned)row[7]);
    const int tid = threadIdx.x;
    if (row[4] != 1) {
        const bool block2d = row[4] == 2;
        const long long cols4 = block2d ? row[5] >> 2 : n4, ld4 = block2d ? row[6] >> 2 : n4;
        const long long stride4 = block2d ? (n4 / cols4) * ld4 : n4;                 // float4 from a slab to the next
        // the thread's four items advance through the slabs TOGETHER (their loads of a slab are independent: 4 - 16 in
        // flight per thread); per item the slabs are still added in slab order
        constexpr int NI = RSB_FLAT4 / 256;
        const f32x4* ptr[NI];
        f32x4 acc[NI];
        bool on[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const long long q = (long long)unit * RSB_FLAT4 + k * 256 + tid;
            on[k] = q < n4;
            const long long qq = on[k] ? q : 0, r = qq / cols4, j = qq - r * cols4;
            ptr[k] = slabs + r * ld4 + j;
            acc[k] = ptr[k][0];
        }
#pragma unroll 8
        for (int sidx = 1; sidx < nsplit; ++sidx) {
            f32x4 v[NI];
#pragma unroll
            for (int k = 0; k < NI; ++k) v[k] = ptr[k][(long long)sidx * stride4];
#pragma unroll
            for (int k = 0; k < NI; ++k) acc[k] += v[k];
        }
#pragma unroll
        for (int k = 0; k < NI; ++k)
            if (on[k]) {
                acc[k] *= alpha;
                reinterpret_cast<f32x4*>(out)[(long long)unit * RSB_FLAT4 + k * 256 + tid] = acc[k];
            }
        return;
    }
    const int taps = (int)row[5], cin = (int)row[6];
    const int chunks = (cin + RSB_CI - 1) / RSB_CI;
    const int co = unit / chunks, c0 = (unit - co * chunks) * RSB_CI;
    const int cw = min(RSB_CI, cin - c0), cw4 = cw >> 2;             // cin % 4 == 0
    const f32x4* src = slabs + (((long long)co * taps) * cin + c0) / 4;
    {   // up to three items per thread (9 taps x 64 quads = 576 = 2.25 x 256), advancing through the slabs together
        constexpr int NI = (RSB_MAXTAPS * (RSB_CI / 4) + 255) / 256;
        const int total = taps * cw4;
        const f32x4* ptr[NI];
        f32x4 acc[NI];
        int dst[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = min(tid + i * 256, total - 1);             // (clamped: loads stay in range; stored only when on)
            const int tap = j / cw4, q = j - tap * cw4;
            ptr[i] = src + (long long)tap * (cin >> 2) + q;
            dst[i] = (tid + i * 256 < total) ? (q * 4) * taps + tap : -1;
            acc[i] = ptr[i][0];
        }
#pragma unroll 8
        for (int sidx = 1; sidx < nsplit; ++sidx) {
            f32x4 v[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) v[i] = ptr[i][(long long)sidx * n4];
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[i] += v[i];
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (dst[i] >= 0) {
                acc[i] *= alpha;
#pragma unroll
                for (int e = 0; e < 4; ++e) tile[dst[i] + e * taps] = acc[i][e];
            }
    }
    __syncthreads();
    // out[(co*cin + c0 + ci)*taps + tap]: cw*taps contiguous floats starting at (co*cin + c0)*taps (a multiple of 4)
    f32x4* dst = reinterpret_cast<f32x4*>(out + ((long long)co * cin + c0) * taps);
    for (int j = tid; j < (cw * taps) >> 2; j += 256) dst[j] = *reinterpret_cast<const f32x4*>(tile + 4 * j);
}

// dst[col] = alpha * sum_r src[r * ld + col]: parameter gradients that are sums over the batch (GroupNorm dgamma / dbeta from
// the per-image sums of psld_gn_bwd_nhwc_f32, bias gradients from per-image column sums).  A block = 64 columns x 16 row
// lanes; every lane adds its rows r, r + 16, ... in index order (fp64), the lanes are combined in lane order.  Jobs from a
// device table (psld_param_reduce_batch_f32) or, two of one shape, from the arguments (psld_param_reduce2_f32).
struct ParamJob {
    const float* src;
    int rows, ld, c;
    float* dst1;
    float* dst2;
    float alpha;
};
__global__ void __launch_bounds__(1024) param_reduce_kernel(const long long* __restrict__ table, int jobs, ParamJob ja,
                                                             ParamJob jb, int blocks_a) {
    __shared__ double sh[16 * 64];
    const int tid = threadIdx.x;
    ParamJob job;
    int local;
    if (table) {
        int lo = 0, hi = jobs - 1;             // last job whose first block <= blockIdx.x
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[(long long)mid * 8 + 7] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        const long long* row = table + (long long)lo * 8;
        job.src = reinterpret_cast<const float*>(row[0]);
        job.rows = (int)row[1]; job.ld = (int)row[2]; job.c = (int)row[3];
        job.dst1 = reinterpret_cast<float*>(row[4]);
        job.dst2 = reinterpret_cast<float*>(row[5]);
        job.alpha = __builtin_bit_cast(float, (unsigned)row[6]);
        local = (int)blockIdx.x - (int)row[7];
    } else if ((int)blockIdx.x < blocks_a) {
        job = ja;
        local = (int)blockIdx.x;
    } else {
        job = jb;
        local = (int)blockIdx.x - blocks_a;
    }
    const int col = local * 64 + (tid & 63);
    const int lane = tid >> 6;
    double acc = 0.0;
    if (col < job.c) {
        // eight rows asked for before the first is added (a lane of a team-kernel job walks 128 rows: one memory latency
        // each otherwise); the adds stay in row order
        const float* p = job.src + col;
        const long long step = 16ll * job.ld;
        int r = lane;
        for (; r + 7 * 16 < job.rows; r += 8 * 16) {
            float v[8];
            const float* q = p + (long long)r * job.ld;
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = q[u * step];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)v[u];
        }
        for (; r < job.rows; r += 16) acc += (double)p[(long long)r * job.ld];
    }
    sh[lane * 64 + (tid & 63)] = acc;
    __syncthreads();
    if (lane == 0 && col < job.c) {
        double t = 0.0;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += sh[l * 64 + tid];
        const float v = (float)(t * (double)job.alpha);
        job.dst1[col] = v;
        if (job.dst2) job.dst2[col] = v;
    }
}

// im2col of a few-channel NHWC tensor (the 6-channel network input / output gradient) for a 3x3 convolution:
// cols[m][ch*9 + t] = x[n][oy*stride + ky - pad][ox*stride + kx - pad][ch] (0 outside), t = ky*3 + kx, zero-filled up
// to ld_out columns; flip = 1 (stride 1) mirrors the taps: the column of tap t holds tap 8 - t.  With 9*c <= 64 the
// stem / head / first pyramid convolutions become plain GEMMs with K = 64 on the fast tile engine instead of the
// scalar-gather fallback (K = 54 does not fit its 32-channel chunking).
__global__ void im2col3x3_small_kernel(const float* __restrict__ x, int ih, int iw, int c, int oh, int ow, int stride,
                                       int pad, int flip, float* __restrict__ out, int ld_out, long long total) {
    GRID_STRIDE(i, total) {          // one thread per (row m, column j)
        const int j = (int)(i % ld_out);
        const long long m = i / ld_out;
        float v = 0.f;
        if (j < 9 * c) {
            const int ch = j / 9;
            int t = j - ch * 9;
            if (flip) t = 8 - t;
            const int ky = t / 3, kx = t - ky * 3;
            const int ox = (int)(m % ow);
            const long long r = m / ow;
            const int oy = (int)(r % oh);
            const long long n = r / oh;
            const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
            if (iy >= 0 && iy < ih && ix >= 0 && ix < iw) v = x[((n * ih + iy) * iw + ix) * c + ch];
        }
        out[i] = v;
    }
}

// 3x3 im2col / col2im of a many-channel NHWC tensor (c % 4 == 0) in the tile engine's K order (tap, channel):
// cols[m][tap*c + ch] = x[n, oy*stride + ky - pad, ox*stride + kx - pad, ch].  One float4 per thread; a (row, tap)
// pair is one contiguous run of c floats on both sides.  With it the stride-2 convolution of the input pyramid
// (layerspp.py:149-163, up_or_down_sampling.py:177) and its data gradient run as pointwise limb GEMMs.
__global__ void im2col3x3_kernel(const float* __restrict__ x, int ih, int iw, int c4, int oh, int ow, int stride, int pad,
                                 float* __restrict__ cols, long long total4) {
    GRID_STRIDE(i, total4) {
        const int q = (int)(i % c4);
        long long r = i / c4;
        const int t = (int)(r % 9);
        const long long m = r / 9;
        const int ky = t / 3, kx = t - ky * 3;
        const int ox = (int)(m % ow);
        const long long rr = m / ow;
        const int oy = (int)(rr % oh);
        const long long n = rr / oh;
        const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < ih && ix >= 0 && ix < iw)
            v = reinterpret_cast<const f32x4*>(x)[((n * ih + iy) * iw + ix) * c4 + q];
        reinterpret_cast<f32x4*>(cols)[i] = v;
    }
}
// dx[n, iy, ix, ch] = sum over the (output pixel, tap) pairs that read this input pixel of dcols[m][tap*c + ch]
// (gather form: fixed summation order ky, kx; no atomics)
__global__ void col2im3x3_kernel(const float* __restrict__ dcols, int ih, int iw, int c4, int oh, int ow, int stride,
                                 int pad, float* __restrict__ dx, long long total4) {
    GRID_STRIDE(i, total4) {
        const int q = (int)(i % c4);
        long long r = i / c4;
        const int ix = (int)(r % iw);
        r /= iw;
        const int iy = (int)(r % ih);
        const long long n = r / ih;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + pad - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= oh) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + pad - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= ow) continue;
                const long long m = (n * oh + oy) * ow + ox;
                acc += reinterpret_cast<const f32x4*>(dcols)[(m * 9 + ky * 3 + kx) * c4 + q];
            }
        }
        reinterpret_cast<f32x4*>(dx)[i] = acc;
    }
}

// 3x3 stride-1 pad-1 convolution with FEW output channels (the 6-channel head, ncsnpp.py:430): a GEMM tile would be
// 95 % padding (609 us on the tile engine), so this is a dot-product kernel.  A wave owns four horizontally adjacent
// pixels; lane l holds channels 4l..4l+3 of the shared 3 x 6 input window (18 float4) and, per (tap, output), the
// matching weight float4 from LDS ([tap][out][c], OHWI order); 4 x COUT fp32 partial sums per lane (fmaf chains in tap
// order) are combined across the lanes by a fixed butterfly.  c = 256 only uses all 64 lanes; c < 256 idles the rest.
template <int COUT>
__global__ void __launch_bounds__(256) conv3x3_fewout_kernel(const float* __restrict__ x, const float* __restrict__ w_ohwi,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            int batch, int h, int w, int c) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [9][COUT][c]
    for (int i = threadIdx.x; i < 9 * COUT * c; i += blockDim.x) {
        const int ch = i % c, r = i / c;
        const int o = r % COUT, t = r / COUT;
        wl[i] = w_ohwi[((long long)o * 9 + t) * c + ch];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4 = c >> 2;
    const bool live = lane < c4;
    const int wq = (w + 3) >> 2;                                     // groups of four pixels per image row
    const long long groups = (long long)batch * h * wq;
    for (long long gidx = (long long)blockIdx.x * 4 + wave; gidx < groups; gidx += (long long)gridDim.x * 4) {
        const int gx = (int)(gidx % wq);
        const long long r = gidx / wq;
        const int oy = (int)(r % h);
        const long long n = r / h;
        const int ox0 = gx * 4;
        f32x4 win[3][6];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 6; ++dx) {
                const int iy = oy + dy - 1, ix = ox0 + dx - 1;
                const bool ok = live && iy >= 0 && iy < h && ix >= 0 && ix < w;
                win[dy][dx] = ok ? reinterpret_cast<const f32x4*>(x)[((n * h + iy) * w + ix) * c4 + lane]
                                 : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        float acc[4][COUT];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int o = 0; o < COUT; ++o) acc[p][o] = 0.f;
        if (live) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - ky * 3;
#pragma unroll
                for (int o = 0; o < COUT; ++o) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + ((t * COUT + o) * c) + lane * 4);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const f32x4 xv = win[ky][p + kx];
                        float a = acc[p][o];
                        a = fmaf(xv[0], wv[0], a);
                        a = fmaf(xv[1], wv[1], a);
                        a = fmaf(xv[2], wv[2], a);
                        a = fmaf(xv[3], wv[3], a);
                        acc[p][o] = a;
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                float v = acc[p][o];
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, 64);
                acc[p][o] = v;
            }
        if (lane < 4 && ox0 + lane < w) {
            float* yo = y + (((n * h + oy) * w + ox0 + lane) * COUT);
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                float v = lane == 0 ? acc[0][o] : lane == 1 ? acc[1][o] : lane == 2 ? acc[2][o] : acc[3][o];
                yo[o] = v + (bias ? bias[o] : 0.f);
            }
        }
    }
}

// Many contiguous float4-multiple copies in one launch.  tab[4*i ..]: src pointer, dst pointer, float4 count, first
// float4 index of entry i in the launch-wide numbering (gathers the 57 time-embedding projection weights of the
// ResBlocks into one matrix once per optimizer step).
__global__ void copy_batch_kernel(const long long* __restrict__ tab, int ntab, long long total4) {
    GRID_STRIDE(i, total4) {
        int lo = 0, hi = ntab - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tab[4 * mid + 3] <= i) lo = mid; else hi = mid - 1;
        }
        const long long* d = tab + 4 * lo;
        const long long j = i - d[3];
        reinterpret_cast<f32x4*>(d[1])[j] = reinterpret_cast<const f32x4*>(d[0])[j];
    }
}

// dst[r*ld_dst + j] = alpha * src[r*ld_src + j] for j < cols: any column count (padding / un-padding small matrices)
__global__ void scale_copy2d_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst,
                                    long long rows, int cols, float alpha) {
    GRID_STRIDE(i, rows * cols) {
        const long long r = i / cols;
        const int j = (int)(i - r * cols);
        dst[r * ld_dst + j] = alpha * src[r * ld_src + j];
    }
}

// ---- pointwise ----------------------------------------------------------------------------------
__global__ void axpby_kernel(const float* __restrict__ a, float sa, const float* __restrict__ b, float sb,
                             float* __restrict__ y, long long n4, long long n, int accumulate) {
    GRID_STRIDE(i, n4) {
        f32x4 v = reinterpret_cast<const f32x4*>(a)[i] * sa;
        if (b) v += reinterpret_cast<const f32x4*>(b)[i] * sb;
        if (accumulate) v += reinterpret_cast<const f32x4*>(y)[i];
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
    // tail
    const long long t0 = n4 * 4;
    GRID_STRIDE(j, n - t0) {
        const long long i = t0 + j;
        float v = a[i] * sa;
        if (b) v += b[i] * sb;
        if (accumulate) v += y[i];
        y[i] = v;
    }
}

// strided 2-D copy: dst[r*ld_dst + c] (+)= src[r*ld_src + c], cols % 4 == 0 (channel concat / split)
__global__ void copy2d_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst,
                              long long rows, int cols4, int accumulate) {
    const long long n = rows * cols4;
    GRID_STRIDE(i, n) {
        const long long r = i / cols4;
        const int c = (int)(i - r * cols4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(src + r * ld_src + c);
        float* d = dst + r * ld_dst + c;
        if (accumulate) v += *reinterpret_cast<const f32x4*>(d);
        *reinterpret_cast<f32x4*>(d) = v;
    }
}

__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
    GRID_STRIDE(i, n) y[i] = silu_f(x[i]);
}
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                long long n) {
    GRID_STRIDE(i, n) dx[i] = dy[i] * dsilu_f(x[i]);
}

// out[b][c] = alpha * sum_p x[(b*hw+p)*ld + c].
// Vector path (c % 4 == 0, 16-B aligned): block = (channel quads) x (pixel lanes) like the GroupNorm
// kernels, grid (chunks, batch); per-thread fp32 partials over <= 64 rows, fp64 across lanes; chunk
// partials are combined by colsum_final_kernel (deterministic, no atomics).
__global__ void colsum_partial_kernel(const float* __restrict__ x, int ld, int hw, int c, int cq, int pl,
                                      int chunk_px, int chunks, double* __restrict__ part) {
    extern __shared__ double red[];  // [pl][cq*4]
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int q = threadIdx.x % cq, l = threadIdx.x / cq;
    const int p0 = chunk * chunk_px, p1 = min(hw, p0 + chunk_px);
    const float* base = x + ((long long)b * hw) * ld + q * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int p = p0 + l; p < p1; p += pl) acc += *reinterpret_cast<const f32x4*>(base + (long long)p * ld);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(long long)l * cq * 4 + q * 4 + e] = (double)acc[e];
    __syncthreads();
    for (int i = threadIdx.x; i < cq * 4; i += blockDim.x) {
        double t = 0.0;
        for (int ll = 0; ll < pl; ++ll) t += red[(long long)ll * cq * 4 + i];
        part[((long long)b * chunks + chunk) * c + i] = t;
    }
}
__global__ void colsum_final_kernel(const double* __restrict__ part, int chunks, int c, float* __restrict__ out,
                                    float alpha, int ld_out) {
    const int b = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= c) return;
    double t = 0.0;
    for (int k = 0; k < chunks; ++k) t += part[((long long)b * chunks + k) * c + col];
    out[(long long)b * ld_out + col] = (float)(t * (double)alpha);
}
// chunk partials -> per-image sums [batch][c] (unscaled) AND the batch total, one launch: grid c/16, block 1024 =
// 16 columns x 64 image lanes (two images per lane at B=128, their chunk loads all in flight together: a version
// with 4 blocks and 128 dependent loads per thread took 57 us); every lane adds its images in index order, the lanes
// are combined in lane order (fixed order, no atomics: bitwise repeatable)
// seg > 0: the total is cut into column segments of `seg` that go to out, out1, out2 (the q | k | v bias gradients of one
// [rows][3c] gradient buffer)
__global__ void __launch_bounds__(1024) colsum_final_total_kernel(const double* __restrict__ part, int chunks, int batch,
                                                                  int c, float* __restrict__ per_image, int ld,
                                                                  float* __restrict__ out, float alpha,
                                                                  float* __restrict__ out1 = nullptr,
                                                                  float* __restrict__ out2 = nullptr, int seg = 0) {
    __shared__ double red[64][16];
    const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    double tot = 0.0;
    if (col < c)
        for (int b = bl; b < batch; b += 64) {
            const double* p = part + (long long)b * chunks * c + col;
            double t = 0.0;
#pragma unroll 8
            for (int k = 0; k < chunks; ++k) t += p[(long long)k * c];
            const float f = (float)t;                 // the total is the sum of the ROUNDED per-image sums
            per_image[(long long)b * ld + col] = f;
            tot += (double)f;
        }
    red[bl][cl] = tot;
    __syncthreads();
    if (bl == 0 && col < c) {
        double t = 0.0;
#pragma unroll
        for (int l = 0; l < 64; ++l) t += red[l][cl];
        const float v = (float)(t * (double)alpha);
        if (seg <= 0 || col < seg) out[col] = v;
        else if (col < 2 * seg) out1[col - seg] = v;
        else out2[col - 2 * seg] = v;
    }
}
// scalar fallback: grid (c/64, batch); block 256 = 64 columns x 4 row lanes
__global__ void colsum_kernel(const float* __restrict__ x, int ld, int hw, int c, float* __restrict__ out,
                              float alpha) {
    __shared__ double red[4][64];
    const int b = blockIdx.y;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    double acc = 0.0;
    if (col < c) {
        const float* p = x + ((long long)b * hw) * ld + col;
        float part = 0.f;
        int cnt = 0;
        for (int r = rl; r < hw; r += 4) {
            part += p[(long long)r * ld];
            if (++cnt == 32) {
                acc += (double)part;
                part = 0.f;
                cnt = 0;
            }
        }
        acc += (double)part;
    }
    red[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && col < c) out[(long long)b * c + col] = (float)((red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) * (double)alpha);
}

// one wave64 per row, row held in registers (L = 256 * V floats, V float4 per lane): one read, one write
template <int V>
__global__ void softmax_rows_reg_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int L) {
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const f32x4* xp = reinterpret_cast<const f32x4*>(x + row * L);
    f32x4 v[V];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        v[j] = xp[lane + 64 * j];
        mx = fmaxf(mx, fmaxf(fmaxf(v[j][0], v[j][1]), fmaxf(v[j][2], v[j][3])));
    }
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[j][e] = expf(v[j][e] - mx);
            s += v[j][e];
        }
    s = wave_sum(s);
    const float inv = 1.0f / s;
    f32x4* yp = reinterpret_cast<f32x4*>(y + row * L);
#pragma unroll
    for (int j = 0; j < V; ++j) yp[lane + 64 * j] = v[j] * inv;
}
template <int V>
__global__ void softmax_rows_bwd_reg_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                            float* __restrict__ dx, long long rows, int L) {
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const f32x4* yp = reinterpret_cast<const f32x4*>(y + row * L);
    const f32x4* gp = reinterpret_cast<const f32x4*>(dy + row * L);
    f32x4 yv[V], gv[V];
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        yv[j] = yp[lane + 64 * j];
        gv[j] = gp[lane + 64 * j];
        d += yv[j][0] * gv[j][0] + yv[j][1] * gv[j][1] + yv[j][2] * gv[j][2] + yv[j][3] * gv[j][3];
    }
    d = wave_sum(d);
    f32x4* op = reinterpret_cast<f32x4*>(dx + row * L);
#pragma unroll
    for (int j = 0; j < V; ++j) op[lane + 64 * j] = yv[j] * (gv[j] - d);
}

// generic fallback: one wave64 per row
__global__ void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int L) {
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xp = x + row * L;
    float* yp = y + row * L;
    float mx = -INFINITY;
    for (int i = lane; i < L; i += 64) mx = fmaxf(mx, xp[i]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int i = lane; i < L; i += 64) {
        const float e = expf(xp[i] - mx);
        yp[i] = e;
        s += e;
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int i = lane; i < L; i += 64) yp[i] *= inv;
}

__global__ void softmax_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                        float* __restrict__ dx, long long rows, int L) {
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* yp = y + row * L;
    const float* gp = dy + row * L;
    float d = 0.f;
    for (int i = lane; i < L; i += 64) d += yp[i] * gp[i];
    d = wave_sum(d);
    for (int i = lane; i < L; i += 64) dx[row * L + i] = yp[i] * (gp[i] - d);
}

// ---- time embedding ---------------------------------------------------------------------------------
// Precision-critical: |p| reaches 1e3..1e4 rad, so full-range sinf/cosf (no fast-math) and the
// reference's multiplication order ((log t * W) * 2) * pi, every product rounded to f32.
__global__ void time_embed_kernel(const float* __restrict__ t, const float* __restrict__ W, float* __restrict__ out,
                                  int batch, int e, int use_log) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * e) return;
    const int b = i / e, k = i % e;
    float p;
    if (use_log) {
        const float lt = logf(t[b]);
        p = __fmul_rn(__fmul_rn(__fmul_rn(lt, W[k]), 2.0f), 3.14159265358979323846f);
    } else {
        p = __fmul_rn(t[b], W[k]);
    }
    out[(long long)b * 2 * e + k] = sinf(p);
    out[(long long)b * 2 * e + e + k] = cosf(p);
}

// ---- edges of the hot path -------------------------------------------------
// Sample writer (callbacks.py:103-107 + util.py:147-158): keep the position half of the [B,2C,H,W]
// f64 state, x*0.5+0.5 (denorm), *255, clip to [0,255], truncate to uint8, NCHW -> NHWC for the encoder.
__global__ void samples_to_u8_kernel(const double* __restrict__ x, unsigned char* __restrict__ out, int batch,
                                     int c, int c_total, int hw, int denorm) {
    const long long n = (long long)batch * hw * c;
    GRID_STRIDE(i, n) {
        const int ch = (int)(i % c);
        long long t = i / c;
        const int p = (int)(t % hw);
        const int b = (int)(t / hw);
        double v = x[((long long)b * c_total + ch) * hw + p];
        if (denorm) v = v * 0.5 + 0.5;
        v = v * 255.0;
        v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
        out[i] = (unsigned char)v;
    }
}
// Data loader (util.py:25-30 + datasets/cifar10.py:33-46): uint8 HWC -> float32 CHW,
// (img / 127.5) - 1 (norm) or img / 255, evaluated in double then rounded to f32 like the reference;
// optional per-image horizontal flip (the RandomHorizontalFlip transform of util.py:80-113).
__global__ void u8_to_images_kernel(const unsigned char* __restrict__ img, float* __restrict__ out,
                                    const unsigned char* __restrict__ flip, int batch, int c, int h, int w, int norm) {
    const long long n = (long long)batch * c * h * w;
    GRID_STRIDE(i, n) {
        const int xw = (int)(i % w);
        long long t = i / w;
        const int y = (int)(t % h);
        t /= h;
        const int ch = (int)(t % c);
        const int b = (int)(t / c);
        const int sx = (flip && flip[b]) ? (w - 1 - xw) : xw;
        const double u = (double)img[(((long long)b * h + y) * w + sx) * c + ch];
        out[i] = (float)(norm ? (u / 127.5) - 1.0 : u / 255.0);
    }
}

__global__ void f64_to_f32_kernel(const double* __restrict__ x, float* __restrict__ y, long long n) {
    GRID_STRIDE(i, n) y[i] = (float)x[i];
}
__global__ void f32_to_f64_kernel(const float* __restrict__ x, double* __restrict__ y, long long n) {
    GRID_STRIDE(i, n) y[i] = (double)x[i];
}

}  // namespace

extern "C" int psld_nchw_to_nhwc_f32(const float* x, float* y, int batch, int c, int hw, hipStream_t stream) {
    PSLD_CHECK_ARG(x && y && batch > 0 && c > 0 && hw > 0, "psld_nchw_to_nhwc_f32: bad args");
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(hw, 32), cdiv(c, 32), batch), dim3(256), 0, stream, x, y, c, hw);
    PSLD_CHECK_LAUNCH("psld_nchw_to_nhwc_f32");
    return PSLD_OK;
}
extern "C" int psld_nhwc_to_nchw_f32(const float* x, float* y, int batch, int c, int hw, hipStream_t stream) {
    PSLD_CHECK_ARG(x && y && batch > 0 && c > 0 && hw > 0, "psld_nhwc_to_nchw_f32: bad args");
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(c, 32), cdiv(hw, 32), batch), dim3(256), 0, stream, x, y, hw, c);
    PSLD_CHECK_LAUNCH("psld_nhwc_to_nchw_f32");
    return PSLD_OK;
}

extern "C" int psld_pack_oihw_to_ohwi_f32(const float* w, float* out, int cout, int cin, int taps, hipStream_t stream) {
    PSLD_CHECK_ARG(w && out, "psld_pack_oihw_to_ohwi_f32: null pointer");
    const long long n = (long long)cout * cin * taps;
    hipLaunchKernelGGL(pack_ohwi_kernel, dim3(grid_for(n)), dim3(256), 0, stream, w, out, cout, cin, taps);
    PSLD_CHECK_LAUNCH("psld_pack_oihw_to_ohwi_f32");
    return PSLD_OK;
}
extern "C" int psld_pack_oihw_to_dgrad_f32(const float* w, float* out, int cout, int cin, int taps, hipStream_t stream) {
    PSLD_CHECK_ARG(w && out, "psld_pack_oihw_to_dgrad_f32: null pointer");
    const long long n = (long long)cout * cin * taps;
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(grid_for(n)), dim3(256), 0, stream, w, out, cout, cin, taps);
    PSLD_CHECK_LAUNCH("psld_pack_oihw_to_dgrad_f32");
    return PSLD_OK;
}
extern "C" int psld_reduce_slabs_f32(const float* slabs, int nsplit, long long n, float* out, int layout, int cout,
                                     int taps, int cin, float alpha, hipStream_t stream) {
    PSLD_CHECK_ARG(slabs && out && nsplit >= 1, "psld_reduce_slabs_f32: bad args");
    PSLD_CHECK_ARG(layout == 0 || n == (long long)cout * taps * cin, "psld_reduce_slabs_f32: shape mismatch");
    const bool vec = n % 4 == 0 && (layout == 0 || cin % 4 == 0) && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0 &&
                     (layout == 1 || (reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(reduce_slabs4_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, slabs, nsplit, n, out, layout,
                           taps, cin, alpha);
    else
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid_for(n)), dim3(256), 0, stream, slabs, nsplit, n, out, layout,
                           taps, cin, alpha);
    PSLD_CHECK_LAUNCH("psld_reduce_slabs_f32");
    return PSLD_OK;
}

extern "C" int psld_reduce_slabs_batch_units(long long n, int layout, int taps, int cin) {
    if (n <= 0 || n % 4) return 0;
    if (layout == 1) {
        if (taps < 1 || taps > RSB_MAXTAPS || cin < 4 || cin % 4 || n % ((long long)taps * cin)) return 0;
        return (int)(n / ((long long)taps * cin)) * ((cin + RSB_CI - 1) / RSB_CI);
    }
    if (layout == 2 && (taps < 4 || taps % 4 || cin < taps || cin % 4 || n % taps)) return 0;      // taps = cols, cin = ld
    return (int)((n / 4 + RSB_FLAT4 - 1) / RSB_FLAT4);
}

extern "C" int psld_reduce_slabs_batch_f32(const long long* table_dev, int jobs, long long units, hipStream_t stream) {
    PSLD_CHECK_ARG(table_dev && jobs > 0 && units > 0 && units < (1ll << 31), "psld_reduce_slabs_batch_f32: bad args");
    hipLaunchKernelGGL(reduce_slabs_batch_kernel, dim3((unsigned)units), dim3(256), 0, stream, table_dev, jobs);
    PSLD_CHECK_LAUNCH("psld_reduce_slabs_batch_f32");
    return PSLD_OK;
}

extern "C" int psld_param_reduce2_f32(const float* src_a, const float* src_b, int rows, int ld, int c, float* dst_a,
                                      float* dst_b, float alpha, hipStream_t stream) {
    PSLD_CHECK_ARG(src_a && dst_a && rows > 0 && c > 0 && ld >= c && (!src_b) == (!dst_b), "psld_param_reduce2_f32: bad args");
    const ParamJob ja{src_a, rows, ld, c, dst_a, nullptr, alpha}, jb{src_b, rows, ld, c, dst_b, nullptr, alpha};
    const int per = cdiv(c, 64);
    hipLaunchKernelGGL(param_reduce_kernel, dim3(src_b ? 2 * per : per), dim3(1024), 0, stream, nullptr, 0, ja, jb, per);
    PSLD_CHECK_LAUNCH("psld_param_reduce2_f32");
    return PSLD_OK;
}

extern "C" int psld_param_reduce_batch_f32(const long long* table_dev, int jobs, int blocks, hipStream_t stream) {
    PSLD_CHECK_ARG(table_dev && jobs > 0 && blocks > 0, "psld_param_reduce_batch_f32: bad args");
    const ParamJob none{nullptr, 0, 0, 0, nullptr, nullptr, 0.f};
    hipLaunchKernelGGL(param_reduce_kernel, dim3(blocks), dim3(1024), 0, stream, table_dev, jobs, none, none, 0);
    PSLD_CHECK_LAUNCH("psld_param_reduce_batch_f32");
    return PSLD_OK;
}

extern "C" int psld_im2col3x3_small_f32(const float* x, int batch, int ih, int iw, int c, int oh, int ow, int stride,
                                        int pad, int flip, float* out, int ld_out, hipStream_t stream) {
    PSLD_CHECK_ARG(x && out && batch > 0 && c > 0 && 9 * c <= ld_out && oh > 0 && ow > 0 && stride >= 1,
                   "psld_im2col3x3_small_f32: bad args (c=%d ld_out=%d)", c, ld_out);
    PSLD_CHECK_ARG(!flip || stride == 1, "psld_im2col3x3_small_f32: flip needs stride 1");
    const long long total = (long long)batch * oh * ow * ld_out;
    hipLaunchKernelGGL(im2col3x3_small_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, ih, iw, c, oh, ow, stride,
                       pad, flip, out, ld_out, total);
    PSLD_CHECK_LAUNCH("psld_im2col3x3_small_f32");
    return PSLD_OK;
}

extern "C" int psld_im2col3x3_f32(const float* x, int batch, int ih, int iw, int c, int oh, int ow, int stride, int pad,
                                  float* cols, hipStream_t stream) {
    PSLD_CHECK_ARG(x && cols && batch > 0 && c > 0 && c % 4 == 0 && oh > 0 && ow > 0 && stride >= 1 && pad >= 0 &&
                       (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(cols) & 15) == 0,
                   "psld_im2col3x3_f32: bad args (c=%d)", c);
    const long long total4 = (long long)batch * oh * ow * 9 * (c / 4);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(total4)), dim3(256), 0, stream, x, ih, iw, c / 4, oh, ow, stride, pad,
                       cols, total4);
    PSLD_CHECK_LAUNCH("psld_im2col3x3_f32");
    return PSLD_OK;
}

extern "C" int psld_col2im3x3_f32(const float* dcols, int batch, int ih, int iw, int c, int oh, int ow, int stride, int pad,
                                  float* dx, hipStream_t stream) {
    PSLD_CHECK_ARG(dcols && dx && batch > 0 && c > 0 && c % 4 == 0 && oh > 0 && ow > 0 && stride >= 1 && pad >= 0 &&
                       (reinterpret_cast<uintptr_t>(dx) & 15) == 0 && (reinterpret_cast<uintptr_t>(dcols) & 15) == 0,
                   "psld_col2im3x3_f32: bad args (c=%d)", c);
    const long long total4 = (long long)batch * ih * iw * (c / 4);
    hipLaunchKernelGGL(col2im3x3_kernel, dim3(grid_for(total4)), dim3(256), 0, stream, dcols, ih, iw, c / 4, oh, ow, stride,
                       pad, dx, total4);
    PSLD_CHECK_LAUNCH("psld_col2im3x3_f32");
    return PSLD_OK;
}

extern "C" int psld_conv3x3_fewout_supported(int cin, int cout) {
    return cin > 0 && cin % 4 == 0 && cin <= 256 && (cout == 3 || cout == 6) && 9 * cout * cin * 4 <= 64 * 1024;
}

extern "C" int psld_conv3x3_fewout_f32(const float* x, const float* w_ohwi, const float* bias, float* y, int batch,
                                       int h, int w, int cin, int cout, hipStream_t stream) {
    PSLD_CHECK_ARG(x && w_ohwi && y && batch > 0 && h > 0 && w > 0, "psld_conv3x3_fewout_f32: bad args");
    PSLD_CHECK_ARG(psld_conv3x3_fewout_supported(cin, cout), "psld_conv3x3_fewout_f32: unsupported cin=%d cout=%d", cin, cout);
    PSLD_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0, "psld_conv3x3_fewout_f32: unaligned input");
    const size_t lds = (size_t)9 * cout * cin * sizeof(float);
    const long long groups = (long long)batch * h * ((w + 3) / 4);
    long long blocks = (groups + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (cout == 6) {
        static PsldPerDeviceFlag configured_; bool& configured = configured_.here();
        if (!configured) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_fewout_kernel<6>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            configured = true;
        }
        hipLaunchKernelGGL(conv3x3_fewout_kernel<6>, dim3((unsigned)blocks), dim3(256), lds, stream, x, w_ohwi, bias, y, batch,
                           h, w, cin);
    } else {
        static PsldPerDeviceFlag configured_; bool& configured = configured_.here();
        if (!configured) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_fewout_kernel<3>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            configured = true;
        }
        hipLaunchKernelGGL(conv3x3_fewout_kernel<3>, dim3((unsigned)blocks), dim3(256), lds, stream, x, w_ohwi, bias, y, batch,
                           h, w, cin);
    }
    PSLD_CHECK_LAUNCH("psld_conv3x3_fewout_f32");
    return PSLD_OK;
}

extern "C" int psld_copy_batch_f32(const long long* table_dev, int entries, long long total4, hipStream_t stream) {
    PSLD_CHECK_ARG(table_dev && entries > 0 && total4 > 0, "psld_copy_batch_f32: bad args");
    hipLaunchKernelGGL(copy_batch_kernel, dim3(grid_for(total4)), dim3(256), 0, stream, table_dev, entries, total4);
    PSLD_CHECK_LAUNCH("psld_copy_batch_f32");
    return PSLD_OK;
}

extern "C" int psld_scale_copy2d_f32(const float* src, int ld_src, float* dst, int ld_dst, long long rows, int cols,
                                     float alpha, hipStream_t stream) {
    PSLD_CHECK_ARG(src && dst && rows >= 0 && cols > 0 && ld_src >= cols && ld_dst >= cols, "psld_scale_copy2d_f32: bad args");
    if (rows == 0) return PSLD_OK;
    hipLaunchKernelGGL(scale_copy2d_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, stream, src, ld_src, dst, ld_dst,
                       rows, cols, alpha);
    PSLD_CHECK_LAUNCH("psld_scale_copy2d_f32");
    return PSLD_OK;
}

extern "C" int psld_axpby_f32(const float* a, float sa, const float* b, float sb, float* y, long long n,
                              int accumulate, hipStream_t stream) {
    PSLD_CHECK_ARG(a && y && n >= 0, "psld_axpby_f32: bad args");
    if (n == 0) return PSLD_OK;
    const bool al = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(y) |
                      (b ? reinterpret_cast<uintptr_t>(b) : 0)) & 15) == 0;
    const long long n4 = al ? n / 4 : 0;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n, 4)), dim3(256), 0, stream, a, sa, b, sb, y, n4, n, accumulate);
    PSLD_CHECK_LAUNCH("psld_axpby_f32");
    return PSLD_OK;
}
extern "C" int psld_copy2d_f32(const float* src, int ld_src, float* dst, int ld_dst, long long rows, int cols,
                               int accumulate, hipStream_t stream) {
    PSLD_CHECK_ARG(src && dst && rows >= 0 && cols > 0, "psld_copy2d_f32: bad args");
    PSLD_CHECK_ARG(cols % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 &&
                       ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0,
                   "psld_copy2d_f32: needs 16-byte aligned rows (cols=%d ld=%d/%d)", cols, ld_src, ld_dst);
    if (rows == 0) return PSLD_OK;
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(rows * (cols / 4))), dim3(256), 0, stream, src, ld_src, dst,
                       ld_dst, rows, cols / 4, accumulate);
    PSLD_CHECK_LAUNCH("psld_copy2d_f32");
    return PSLD_OK;
}
extern "C" int psld_silu_f32(const float* x, float* y, long long n, hipStream_t stream) {
    PSLD_CHECK_ARG(x && y, "psld_silu_f32: null pointer");
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, n);
    PSLD_CHECK_LAUNCH("psld_silu_f32");
    return PSLD_OK;
}
extern "C" int psld_silu_bwd_f32(const float* x, const float* dy, float* dx, long long n, hipStream_t stream) {
    PSLD_CHECK_ARG(x && dy && dx, "psld_silu_bwd_f32: null pointer");
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, dy, dx, n);
    PSLD_CHECK_LAUNCH("psld_silu_bwd_f32");
    return PSLD_OK;
}
extern "C" long long psld_colsum_workspace_bytes(int batch, int hw, int c) {
    return (long long)batch * 64 * c * sizeof(double) + 256;
}
extern "C" int psld_colsum_f32(const float* x, int ld, int batch, int hw, int c, float* out, float alpha,
                               void* workspace, hipStream_t stream) {
    PSLD_CHECK_ARG(x && out && batch > 0 && hw > 0 && c > 0, "psld_colsum_f32: bad args");
    const bool vec = workspace && c % 4 == 0 && ld % 4 == 0 && c / 4 <= 256 && hw >= 8 &&
                     (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (vec) {
        const int cq = c / 4;
        int pl = 256 / cq;
        if (pl < 1) pl = 1;
        if (pl > hw) pl = hw;
        int chunks = cdiv(1024, batch);
        const int max_chunks = cdiv(hw, pl * 4);
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks > 16) chunks = 16;
        if (chunks < 1) chunks = 1;
        const int chunk_px = cdiv(hw, chunks);
        chunks = cdiv(hw, chunk_px);
        double* part = reinterpret_cast<double*>(workspace);
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(chunks, batch), dim3(cq * pl), (size_t)pl * cq * 4 * sizeof(double),
                           stream, x, ld, hw, c, cq, pl, chunk_px, chunks, part);
        PSLD_CHECK_LAUNCH("colsum_partial_kernel");
        hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(c, 128), batch), dim3(128), 0, stream, part, chunks, c, out,
                           alpha, c);
        PSLD_CHECK_LAUNCH("colsum_final_kernel");
        return PSLD_OK;
    }
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(c, 64), batch), dim3(256), 0, stream, x, ld, hw, c, out, alpha);
    PSLD_CHECK_LAUNCH("psld_colsum_f32");
    return PSLD_OK;
}
// out[c] = alpha * sum over (batch, hw) of x; per_image[b][c] (optional, unscaled) = sum over hw.  Two launches
// (chunk partials; per-image sums + batch total) instead of the four of two chained psld_colsum_f32 calls.
extern "C" int psld_bias_grad_f32(const float* x, int ld, int batch, int hw, int c, float* per_image, int ld_per_image,
                                  float* out, float alpha, void* workspace, hipStream_t stream) {
    PSLD_CHECK_ARG(x && out && workspace && batch > 0 && hw > 0 && c > 0, "psld_bias_grad_f32: bad args");
    PSLD_CHECK_ARG(c % 4 == 0 && ld % 4 == 0 && c / 4 <= 256 && (reinterpret_cast<uintptr_t>(x) & 15) == 0,
                   "psld_bias_grad_f32: needs c %%4 == 0 (<= 1024), ld %%4 == 0 and a 16-byte aligned input");
    const int cq = c / 4;
    int pl = 256 / cq;
    if (pl < 1) pl = 1;
    if (pl > hw) pl = hw;
    int chunks = cdiv(1024, batch);
    const int max_chunks = cdiv(hw, pl * 4);
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks > 16) chunks = 16;
    if (chunks < 1) chunks = 1;
    const int chunk_px = cdiv(hw, chunks);
    chunks = cdiv(hw, chunk_px);
    double* part = reinterpret_cast<double*>(workspace);
    // without a caller buffer the per-image sums live behind the partials (psld_colsum_workspace_bytes covers
    // 64 chunk rows per image, at most 16 are used)
    float* pim = per_image ? per_image : reinterpret_cast<float*>(part + (long long)batch * 16 * c);
    const int ldp = per_image && ld_per_image > 0 ? ld_per_image : c;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(chunks, batch), dim3(cq * pl), (size_t)pl * cq * 4 * sizeof(double),
                       stream, x, ld, hw, c, cq, pl, chunk_px, chunks, part);
    PSLD_CHECK_LAUNCH("colsum_partial_kernel");
    hipLaunchKernelGGL(colsum_final_total_kernel, dim3(cdiv(c, 16)), dim3(1024), 0, stream, part, chunks, batch, c, pim, ldp,
                       out, alpha);
    PSLD_CHECK_LAUNCH("colsum_final_total_kernel");
    return PSLD_OK;
}
// psld_bias_grad_f32 over a [batch*hw][3*seg] buffer whose three column segments belong to three parameters.
extern "C" int psld_bias_grad_seg_f32(const float* x, int ld, int batch, int hw, int seg, float* out0, float* out1,
                                      float* out2, float alpha, void* workspace, hipStream_t stream) {
    const int c = 3 * seg;
    PSLD_CHECK_ARG(x && out0 && out1 && out2 && workspace && batch > 0 && hw > 0 && seg > 0, "psld_bias_grad_seg_f32: bad args");
    PSLD_CHECK_ARG(c % 4 == 0 && ld % 4 == 0 && c / 4 <= 256 && (reinterpret_cast<uintptr_t>(x) & 15) == 0,
                   "psld_bias_grad_seg_f32: needs 3*seg %%4 == 0 (<= 1024), ld %%4 == 0 and a 16-byte aligned input");
    const int cq = c / 4;
    int pl = 256 / cq;
    if (pl < 1) pl = 1;
    if (pl > hw) pl = hw;
    int chunks = cdiv(1024, batch);
    const int max_chunks = cdiv(hw, pl * 4);
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks > 16) chunks = 16;
    if (chunks < 1) chunks = 1;
    const int chunk_px = cdiv(hw, chunks);
    chunks = cdiv(hw, chunk_px);
    double* part = reinterpret_cast<double*>(workspace);
    float* pim = reinterpret_cast<float*>(part + (long long)batch * 16 * c);     // per-image sums: scratch behind the partials
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(chunks, batch), dim3(cq * pl), (size_t)pl * cq * 4 * sizeof(double),
                       stream, x, ld, hw, c, cq, pl, chunk_px, chunks, part);
    PSLD_CHECK_LAUNCH("colsum_partial_kernel");
    hipLaunchKernelGGL(colsum_final_total_kernel, dim3(cdiv(c, 16)), dim3(1024), 0, stream, part, chunks, batch, c, pim, c,
                       out0, alpha, out1, out2, seg);
    PSLD_CHECK_LAUNCH("colsum_final_total_kernel");
    return PSLD_OK;
}
// Cross entropy of [rows][n] logits against int64 labels (nn.CrossEntropyLoss, losses.py:147-173) with its gradient:
// one block; every thread walks rows tid, tid+256, ...; the per-thread sums are combined in thread order.
__global__ void softmax_xent_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, int rows,
                                    int n, float loss_scale, float grad_scale, float* __restrict__ loss,
                                    float* __restrict__ dlogits, float* __restrict__ correct) {
    __shared__ double sl[256];
    __shared__ int sc[256];
    double lsum = 0.0;
    int csum = 0;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        const float* z = logits + (long long)r * n;
        float mx = z[0];
        int arg = 0;
        for (int j = 1; j < n; ++j)
            if (z[j] > mx) { mx = z[j]; arg = j; }
        float se = 0.f;
        for (int j = 0; j < n; ++j) se += expf(z[j] - mx);
        const int y = (int)labels[r];
        lsum += (double)(logf(se) + mx - z[y]);
        csum += arg == y;
        if (dlogits) {
            const float inv = 1.0f / se;
            for (int j = 0; j < n; ++j)
                dlogits[(long long)r * n + j] = grad_scale * (expf(z[j] - mx) * inv - (j == y ? 1.f : 0.f));
        }
    }
    sl[threadIdx.x] = lsum;
    sc[threadIdx.x] = csum;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        int c = 0;
        for (int i = 0; i < (int)blockDim.x; ++i) { t += sl[i]; c += sc[i]; }
        if (loss) *loss = (float)(t * (double)loss_scale);
        if (correct) *correct = (float)c;
    }
}

extern "C" int psld_softmax_xent_f32(const float* logits, const long long* labels, int rows, int n, float loss_scale,
                                     float grad_scale, float* loss, float* dlogits, float* correct, hipStream_t stream) {
    PSLD_CHECK_ARG(logits && labels && rows > 0 && n > 0, "psld_softmax_xent_f32: bad args");
    hipLaunchKernelGGL(softmax_xent_kernel, dim3(1), dim3(256), 0, stream, logits, labels, rows, n, loss_scale, grad_scale,
                       loss, dlogits, correct);
    PSLD_CHECK_LAUNCH("psld_softmax_xent_f32");
    return PSLD_OK;
}
extern "C" int psld_softmax_rows_f32(const float* x, float* y, long long rows, int L, hipStream_t stream) {
    PSLD_CHECK_ARG(x && y && rows >= 0 && L > 0, "psld_softmax_rows_f32: bad args");
    if (rows == 0) return PSLD_OK;
    const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (al && L == 256) {
        hipLaunchKernelGGL(softmax_rows_reg_kernel<1>, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, y, rows, L);
    } else if (al && L == 512) {
        hipLaunchKernelGGL(softmax_rows_reg_kernel<2>, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, y, rows, L);
    } else if (al && L == 1024) {
        hipLaunchKernelGGL(softmax_rows_reg_kernel<4>, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, y, rows, L);
    } else
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, y, rows, L);
    PSLD_CHECK_LAUNCH("psld_softmax_rows_f32");
    return PSLD_OK;
}
extern "C" int psld_softmax_rows_bwd_f32(const float* y, const float* dy, float* dx, long long rows, int L,
                                         hipStream_t stream) {
    PSLD_CHECK_ARG(y && dy && dx && rows >= 0 && L > 0, "psld_softmax_rows_bwd_f32: bad args");
    if (rows == 0) return PSLD_OK;
    const bool al = ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
    if (al && L == 256) {
        hipLaunchKernelGGL(softmax_rows_bwd_reg_kernel<1>, dim3(cdiv(rows, 4)), dim3(256), 0, stream, y, dy, dx, rows, L);
    } else if (al && L == 512) {
        hipLaunchKernelGGL(softmax_rows_bwd_reg_kernel<2>, dim3(cdiv(rows, 4)), dim3(256), 0, stream, y, dy, dx, rows, L);
    } else if (al && L == 1024) {
        hipLaunchKernelGGL(softmax_rows_bwd_reg_kernel<4>, dim3(cdiv(rows, 4)), dim3(256), 0, stream, y, dy, dx, rows, L);
    } else
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, y, dy, dx, rows, L);
    PSLD_CHECK_LAUNCH("psld_softmax_rows_bwd_f32");
    return PSLD_OK;
}
extern "C" int psld_time_embed_f32(const float* t, const float* W, float* out, int batch, int e, int use_log,
                                   hipStream_t stream) {
    PSLD_CHECK_ARG(t && W && out && batch > 0 && e > 0, "psld_time_embed_f32: bad args");
    hipLaunchKernelGGL(time_embed_kernel, dim3(cdiv((long long)batch * e, 256)), dim3(256), 0, stream, t, W, out,
                       batch, e, use_log);
    PSLD_CHECK_LAUNCH("psld_time_embed_f32");
    return PSLD_OK;
}
extern "C" int psld_samples_to_uint8(const double* x, unsigned char* out, int batch, int c, int c_total, int hw,
                                     int denorm, hipStream_t stream) {
    PSLD_CHECK_ARG(x && out && batch > 0 && c > 0 && c_total >= c && hw > 0, "psld_samples_to_uint8: bad args");
    hipLaunchKernelGGL(samples_to_u8_kernel, dim3(grid_for((long long)batch * c * hw)), dim3(256), 0, stream, x, out,
                       batch, c, c_total, hw, denorm);
    PSLD_CHECK_LAUNCH("psld_samples_to_uint8");
    return PSLD_OK;
}
extern "C" int psld_uint8_to_images_f32(const unsigned char* img, float* out, const unsigned char* flip, int batch,
                                        int c, int h, int w, int norm, hipStream_t stream) {
    PSLD_CHECK_ARG(img && out && batch > 0 && c > 0 && h > 0 && w > 0, "psld_uint8_to_images_f32: bad args");
    hipLaunchKernelGGL(u8_to_images_kernel, dim3(grid_for((long long)batch * c * h * w)), dim3(256), 0, stream, img,
                       out, flip, batch, c, h, w, norm);
    PSLD_CHECK_LAUNCH("psld_uint8_to_images_f32");
    return PSLD_OK;
}
extern "C" int psld_f64_to_f32(const double* x, float* y, long long n, hipStream_t stream) {
    PSLD_CHECK_ARG(x && y, "psld_f64_to_f32: null pointer");
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, n);
    PSLD_CHECK_LAUNCH("psld_f64_to_f32");
    return PSLD_OK;
}
extern "C" int psld_f32_to_f64(const float* x, double* y, long long n, hipStream_t stream) {
    PSLD_CHECK_ARG(x && y, "psld_f32_to_f64: null pointer");
    hipLaunchKernelGGL(f32_to_f64_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, n);
    PSLD_CHECK_LAUNCH("psld_f32_to_f64");
    return PSLD_OK;
}
