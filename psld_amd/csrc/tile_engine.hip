// fp32 MFMA tile engine: GEMM (NT / NN / TN), implicit-GEMM convolution (forward and
// data-gradient) and convolution weight-gradient, all on NHWC activations.
//
// Replaces, on the reference's hot path, every dense contraction that the reference runs
// through ATen: nn.Conv2d 3x3/1x1 (song_sde/layers.py:85-109), NIN (layers.py:531-540),
// nn.Linear (layerspp.py:225-228, ncsnpp.py:99-105), the attention einsums
// (layerspp.py:82-87) and the strided pyramid conv (up_or_down_sampling.py:177) — and their
// autograd backward passes.  In the default arithmetic mode (PSLD_MATH_BF16X6) the 3x3 stride-1
// convolutions and the 1x1 / NIN projections of 128-multiple widths go to conv_split.hip instead;
// this engine keeps everything else (and everything when PSLD_MATH=f32).
//
// Design (CDNA4):  one workgroup = 256 threads = 4 wave64, block tile 128x128, K step 32.
// Each wave owns a 64x64 sub-tile = 2x2 v_mfma_f32_32x32x2_f32 accumulators (64 VGPRs);
// exact fp32 (k-ordered fmaf chain), so outputs stay within fp32 rounding of the CPU
// reference.  Operands are staged global -> registers -> LDS (register prefetch of tile
// k+1 overlaps the 64 MFMAs of tile k).  K-contiguous operands sit in LDS as [row][32+4]
// and are read as ds_read_b128 (4 k-values per lane per read; lanes 0-31 take k=8j..8j+3,
// lanes 32-63 take k=8j+4..8j+7 — the k order inside a dot product is free as long as A and
// B agree); row-contiguous ("MC") operands sit as [k][128] and are read with ds_read_b32.
// The 36-float row stride makes both the b128 fragment reads and the b128 staging writes
// bank-conflict free (MI355X LDS: 64 banks, 16-lane groups for b128).
#include "common.h"
#include "psld_hip.h"
#include "tile_shared.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int KC_LD = BK + 4;   // K-contiguous LDS image: [128][36]
constexpr int MC_LD = BM;       // row-contiguous LDS image: [32][128]
constexpr int NTHREADS = 256;

enum : int { OP_KC = 0, OP_MC = 1, OP_IM2COL = 2, OP_SHIFT = 3 };

struct Operand {
    const float* p;     // base
    const float* p2;    // second source (im2col concat) or null
    long long stride_z; // per-batch element stride
    int ld;             // leading dimension (elements)
    int vec;            // 1: float4 loads legal (16B aligned base, ld%4==0, extents %4==0)
};

struct ConvGeom {
    int IH, IW, C1, C2, OH, OW, KH, KW, stride, pad, tstride;
};

using Epilogue = PsldEpilogue;

struct TileArgs {
    int M, N, K;
    Operand A, B;
    float* C;
    long long c_stride_z;      // per zb
    long long c_stride_split;  // per split slab
    int ldc;
    int nsplit;
    int kper;                  // K range per split (multiple of BK)
    ConvGeom g;
    Epilogue e;
};

struct RowInfo {  // im2col row (output pixel) decomposition
    int img, oy, ox;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// ---- operand tile -> registers -------------------------------------------------------
// KC modes (OP_KC, OP_IM2COL): thread (c4 = tid&7, r0 = tid>>3) owns rows r0+32i, k = 4*c4..4*c4+3
// MC modes (OP_MC, OP_SHIFT):  thread (c  = tid&31, kr = tid>>5) owns k rows kr+8i, m = 4c..4c+3
template <int MODE>
__device__ __forceinline__ void load_tile(f32x4 (&r)[4], const Operand& op, const ConvGeom& g,
                                          const float* base, int row0, int nrows, int k0, int kend,
                                          const RowInfo (&ri)[4], int tap_z) {
    const int tid = threadIdx.x;
    if constexpr (MODE == OP_KC) {
        const int c4 = tid & 7, r0 = tid >> 3;
        const int gk = k0 + c4 * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gr = row0 + r0 + 32 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gr < nrows && gk < kend) {
                const float* p = base + (long long)gr * op.ld + gk;
                if (op.vec) {
                    v = ld4(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (gk + e < kend) v[e] = p[e];
                }
            }
            r[i] = v;
        }
    } else if constexpr (MODE == OP_IM2COL) {
        const int c4 = tid & 7;
        const int gk = k0 + c4 * 4;
        const int Ct = g.C1 + g.C2;
        if (op.vec) {
            // 4 consecutive k share one tap and one source
            const int tap = gk / Ct;
            const int c = gk - tap * Ct;
            const int ky = tap / g.KW, kx = tap - ky * g.KW;
            const bool second = c >= g.C1;
            const float* src = second ? op.p2 : base;
            const int cs = second ? g.C2 : g.C1;
            const int cc = second ? c - g.C1 : c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ri[i].img >= 0 && gk < kend) {
                    int iy = ri[i].oy * g.stride + ky - g.pad;
                    int ix = ri[i].ox * g.stride + kx - g.pad;
                    bool ok = true;
                    if (g.tstride > 1) {
                        ok = (iy % g.tstride == 0) && (ix % g.tstride == 0) && iy >= 0 && ix >= 0;
                        iy /= g.tstride;
                        ix /= g.tstride;
                    }
                    if (ok && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW)
                        v = ld4(src + ((long long)(ri[i].img * g.IH + iy) * g.IW + ix) * cs + cc);
                }
                r[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ri[i].img >= 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = gk + e;
                        if (k >= kend) continue;
                        const int tap = k / Ct;
                        const int c = k - tap * Ct;
                        const int ky = tap / g.KW, kx = tap - ky * g.KW;
                        int iy = ri[i].oy * g.stride + ky - g.pad;
                        int ix = ri[i].ox * g.stride + kx - g.pad;
                        bool ok = true;
                        if (g.tstride > 1) {
                            ok = (iy % g.tstride == 0) && (ix % g.tstride == 0) && iy >= 0 && ix >= 0;
                            iy /= g.tstride;
                            ix /= g.tstride;
                        }
                        if (ok && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW) {
                            const bool second = c >= g.C1;
                            const float* src = second ? op.p2 : base;
                            const int cs = second ? g.C2 : g.C1;
                            const int cc = second ? c - g.C1 : c;
                            v[e] = src[((long long)(ri[i].img * g.IH + iy) * g.IW + ix) * cs + cc];
                        }
                    }
                }
                r[i] = v;
            }
        }
    } else if constexpr (MODE == OP_MC) {
        const int c = tid & 31, kr = tid >> 5;
        const int gm = row0 + c * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gk = k0 + kr + 8 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gk < kend && gm < nrows) {
                const float* p = base + (long long)gk * op.ld + gm;
                if (op.vec) {
                    v = ld4(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (gm + e < nrows) v[e] = p[e];
                }
            }
            r[i] = v;
        }
    } else {  // OP_SHIFT: k indexes OUTPUT pixels of the conv; value = X[img, oy*s+ky-p, ox*s+kx-p, m]
        const int c = tid & 31, kr = tid >> 5;
        const int gm = row0 + c * 4;
        const int ky = tap_z / g.KW, kx = tap_z - ky * g.KW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gk = k0 + kr + 8 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gk < kend && gm < nrows) {
                const int ox = gk % g.OW;
                const int t = gk / g.OW;
                const int oy = t % g.OH;
                const int img = t / g.OH;
                const int iy = oy * g.stride + ky - g.pad;
                const int ix = ox * g.stride + kx - g.pad;
                if (iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW) {
                    const float* p = base + ((long long)(img * g.IH + iy) * g.IW + ix) * op.ld + gm;
                    if (op.vec) {
                        v = ld4(p);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (gm + e < nrows) v[e] = p[e];
                    }
                }
            }
            r[i] = v;
        }
    }
}

template <int MODE>
__device__ __forceinline__ void store_tile(float* lds, const f32x4 (&r)[4]) {
    const int tid = threadIdx.x;
    if constexpr (MODE == OP_KC || MODE == OP_IM2COL) {
        const int c4 = tid & 7, r0 = tid >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(lds + (r0 + 32 * i) * KC_LD + c4 * 4) = r[i];
    } else {
        const int c = tid & 31, kr = tid >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(lds + (kr + 8 * i) * MC_LD + c * 4) = r[i];
    }
}

template <int AMODE, int BMODE>
__global__ void __launch_bounds__(NTHREADS, 2) tile_kernel(const TileArgs a) {
    constexpr bool A_KC = (AMODE == OP_KC || AMODE == OP_IM2COL);
    constexpr bool B_KC = (BMODE == OP_KC || BMODE == OP_IM2COL);
    __shared__ __attribute__((aligned(16))) float As[A_KC ? BM * KC_LD : BK * MC_LD];
    __shared__ __attribute__((aligned(16))) float Bs[B_KC ? BN * KC_LD : BK * MC_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const int tiles_n = (a.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n;
    const int tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int z = blockIdx.y;
    const int split = z % a.nsplit;
    const int zb = z / a.nsplit;
    const int kbeg = split * a.kper;
    const int kend = min(a.K, kbeg + a.kper);

    const float* Abase = a.A.p + (long long)zb * a.A.stride_z;
    const float* Bbase = a.B.p + (long long)zb * a.B.stride_z;

    RowInfo ri[4];
    if constexpr (AMODE == OP_IM2COL) {
        const int r0 = tid >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = m0 + r0 + 32 * i;
            if (gm < a.M) {
                ri[i].ox = gm % a.g.OW;
                const int t = gm / a.g.OW;
                ri[i].oy = t % a.g.OH;
                ri[i].img = t / a.g.OH;
            } else {
                ri[i].img = -1;
                ri[i].oy = ri[i].ox = 0;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) ri[i].img = ri[i].oy = ri[i].ox = 0;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    f32x4 ra[4], rb[4];
    load_tile<AMODE>(ra, a.A, a.g, Abase, m0, a.M, kbeg, kend, ri, zb);
    load_tile<BMODE>(rb, a.B, a.g, Bbase, n0, a.N, kbeg, kend, ri, zb);
    store_tile<AMODE>(As, ra);
    store_tile<BMODE>(Bs, rb);
    __syncthreads();

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK) < kend;
        if (more) {
            load_tile<AMODE>(ra, a.A, a.g, Abase, m0, a.M, k0 + BK, kend, ri, zb);
            load_tile<BMODE>(rb, a.B, a.g, Bbase, n0, a.N, k0 + BK, kend, ri, zb);
        }
#pragma unroll
        for (int j = 0; j < BK / 8; ++j) {
            float af[2][4], bf[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (A_KC) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(As + (wr * 64 + i * 32 + r) * KC_LD + 8 * j + 4 * h);
                    af[i][0] = t[0]; af[i][1] = t[1]; af[i][2] = t[2]; af[i][3] = t[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) af[i][s] = As[(8 * j + 4 * h + s) * MC_LD + wr * 64 + i * 32 + r];
                }
                if constexpr (B_KC) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(Bs + (wc * 64 + i * 32 + r) * KC_LD + 8 * j + 4 * h);
                    bf[i][0] = t[0]; bf[i][1] = t[1]; bf[i][2] = t[2]; bf[i][3] = t[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) bf[i][s] = Bs[(8 * j + 4 * h + s) * MC_LD + wc * 64 + i * 32 + r];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[n][s], acc[i][n], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_tile<AMODE>(As, ra);
            store_tile<BMODE>(Bs, rb);
            __syncthreads();
        }
    }

    // ---- epilogue: C/D layout of v_mfma_f32_32x32x2: col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5)
    float* Cb = a.C + (long long)zb * a.c_stride_z + (long long)split * a.c_stride_split;
    const float* Rb = a.e.res ? a.e.res + (long long)zb * a.e.res_stride_z : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int gn = n0 + wc * 64 + n * 32 + r;
            if (gn >= a.N) continue;
            const float bias = a.e.bias ? a.e.bias[gn] : 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int gm = m0 + wr * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (gm >= a.M) continue;
                float x = acc[i][n][v] * a.e.alpha + bias;
                if (a.e.rowbias) x += a.e.rowbias[(long long)(gm / a.e.rows_per_img) * a.e.ld_rowbias + gn];
                if (Rb) x += Rb[(long long)gm * a.e.ldres + gn];
                x *= a.e.out_scale;
                float* cp = Cb + (long long)gm * a.ldc + gn;
                if (a.e.accumulate) x += *cp;
                *cp = x;
            }
        }
    }
}

template <int AMODE, int BMODE>
int launch(const TileArgs& a, int nz, hipStream_t stream, const char* name) {
    if (a.M <= 0 || a.N <= 0 || nz <= 0) return PSLD_OK;
    const long long tiles = (long long)cdiv(a.M, BM) * cdiv(a.N, BN);
    PSLD_CHECK_ARG(tiles < (1LL << 31) && nz * a.nsplit <= 65535, "%s: grid too large", name);
    dim3 grid((unsigned)tiles, (unsigned)(nz * a.nsplit));
    hipLaunchKernelGGL((tile_kernel<AMODE, BMODE>), grid, dim3(NTHREADS), 0, stream, a);
    PSLD_CHECK_LAUNCH(name);
    return PSLD_OK;
}


// =======================================================================================
// FAST specialisation: the hot shapes (every operand float4-addressable, K % 32 == 0 per
// split, conv channels % 32 == 0, no transposed stride).  Differences from the generic kernel:
//   * double-buffered LDS, ONE barrier per K-step (loads of tile k+1 in flight under the MFMAs
//     of tile k, their ds_writes go to the other buffer);
//   * branch-free loaders (clamped address + select instead of exec-masked branches);
//   * conv K order = (channel chunk, tap, channel-in-chunk): the 9 taps of one 32-channel chunk
//     are consecutive K-steps, so 8 of 9 gathers re-hit the same cache lines in L1/L2;
//   * wgrad pixel decomposition by shifts (OH, OW powers of two).
// =======================================================================================
// 16 bytes of zeros: out-of-range lanes load from here, so no select (and no early vmcnt wait) is needed
__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

struct FastGeom {
    int taps, ow_shift, oh_shift, ct_chunks;  // ct_chunks = (C1+C2)/32
    const float* zero;                        // 16 B of zeros in global memory (filled by launch_fast)
};

template <int MODE, int NR>
__device__ __forceinline__ void fast_load(f32x4 (&r)[NR], const Operand& op, const ConvGeom& g, const FastGeom& fg,
                                          const float* base, int row0, int nrows, int kstep, int kend,
                                          const RowInfo (&ri)[NR], int tap_z, const float* zp) {
    const int tid = threadIdx.x;
    if constexpr (MODE == OP_KC) {
        const int c4 = tid & 7, r0 = tid >> 3;
        const int gk = kstep * BK + c4 * 4;
        const bool kok = gk < kend;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int gr = row0 + r0 + 32 * i;
            const bool ok = kok && gr < nrows;
            r[i] = ld4(ok ? base + ((long long)gr * op.ld + gk) : zp);
        }
    } else if constexpr (MODE == OP_IM2COL) {
        // ri[i].img = bit mask of valid taps (0 for rows beyond M), ri[i].oy = pixel index of tap (0,0)
        const int c4 = tid & 7;
        const int chunk = kstep / fg.taps;          // wave-uniform
        const int tap = kstep - chunk * fg.taps;
        const int ky = tap / g.KW, kx = tap - ky * g.KW;
        const int dpix = ky * g.IW + kx;
        const int c0 = chunk * BK;
        const bool second = c0 >= g.C1;
        const float* src = second ? op.p2 : base;
        const int cs = second ? g.C2 : g.C1;
        const int cc = (second ? c0 - g.C1 : c0) + c4 * 4;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const bool ok = (ri[i].img >> tap) & 1;
            r[i] = ld4(ok ? src + ((long long)(ri[i].oy + dpix) * cs + cc) : zp);
        }
    } else if constexpr (MODE == OP_MC) {
        const int c = tid & 31, kr = tid >> 5;
        const int gm = row0 + c * 4;
        const bool mok = gm < nrows;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gk = kstep * BK + kr + 8 * i;
            const bool ok = mok && gk < kend;
            r[i] = ld4(ok ? base + ((long long)gk * op.ld + gm) : zp);
        }
    } else {  // OP_SHIFT
        const int c = tid & 31, kr = tid >> 5;
        const int gm = row0 + c * 4;
        const bool mok = gm < nrows;
        const int ky = tap_z / g.KW, kx = tap_z - ky * g.KW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gk = kstep * BK + kr + 8 * i;
            const int ox = gk & (g.OW - 1);
            const int t = gk >> fg.ow_shift;
            const int oy = t & (g.OH - 1);
            const int img = t >> fg.oh_shift;
            const int iy = oy * g.stride + ky - g.pad;
            const int ix = ox * g.stride + kx - g.pad;
            const bool ok = mok && gk < kend && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
            r[i] = ld4(ok ? base + (((long long)(img * g.IH + iy) * g.IW + ix) * op.ld + gm) : zp);
        }
    }
}

template <int NR>
__device__ __forceinline__ void store_kc_rows(float* lds, const f32x4 (&r)[NR]) {
    const int tid = threadIdx.x;
    const int c4 = tid & 7, r0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < NR; ++i) *reinterpret_cast<f32x4*>(lds + (r0 + 32 * i) * KC_LD + c4 * 4) = r[i];
}

// TBM = 128 (default) or 64 (small-M layers: the 8x8 convs give only 128 tiles of 128 rows for 256 CUs).
// TBN = 128 (default) or 256 (K-contiguous B only): each wave then owns 64 x 128 = 8 accumulators, the A tile is
// fetched once for all 256 output channels, 0.19 instead of 0.25 fragment reads per MFMA.
template <int AMODE, int BMODE, int TBM, int NBUF, int TBN>
__global__ void __launch_bounds__(NTHREADS, (NBUF == 2 || TBN == 256) ? 2 : 3) tile_kernel_fast(const TileArgs a, const FastGeom fg) {
    constexpr bool A_KC = (AMODE == OP_KC || AMODE == OP_IM2COL);
    constexpr bool B_KC = (BMODE == OP_KC || BMODE == OP_IM2COL);
    static_assert(TBM == 128 || (TBM == 64 && A_KC), "TBM=64 needs a K-contiguous A operand");
    constexpr int MI = TBM / 64;          // 32-row MFMA tiles per wave along M
    constexpr int NRA = A_KC ? TBM / 32 : 4;  // A rows (KC) or k-rows (MC) staged per thread
    static_assert(TBN == 128 || (TBN == 256 && B_KC), "TBN=256 needs a K-contiguous B operand");
    constexpr int NI = TBN / 64;              // 32-column MFMA tiles per wave along N
    constexpr int NRB = B_KC ? TBN / 32 : 4;  // B rows (KC) or k-rows (MC) staged per thread
    constexpr int A_SZ = A_KC ? TBM * KC_LD : BK * MC_LD;
    constexpr int B_SZ = B_KC ? TBN * KC_LD : BK * MC_LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As0 = smem;
    float* Bs0 = smem + NBUF * A_SZ;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const int tiles_n = (a.N + TBN - 1) / TBN;
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (private L2 each), so give
    // every XCD a contiguous run of tiles — the N-tiles of one M-tile and neighbouring M-tiles (halo
    // rows) then share one L2.  Bijective form (cdna guide T1) for any grid size.
    int bid = blockIdx.x;
    int z = blockIdx.y;
    if constexpr (BMODE == OP_SHIFT) {
        // weight gradient: the tiles x taps blocks of one split-K range read the same dY / x pixels;
        // keep them on one XCD (shared L2) by swizzling the LINEAR block id, split slowest.
        const int per_z = gridDim.x;
        const int nwg = per_z * gridDim.y, lin = blockIdx.y * per_z + blockIdx.x;
        const int q = nwg >> 3, rem = nwg & 7, xcd = lin & 7;
        const int vid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (lin >> 3);
        bid = vid % per_z;
        const int zz = vid / per_z;               // = split * taps + tap
        const int ntap = gridDim.y / a.nsplit;
        z = (zz % ntap) * a.nsplit + zz / ntap;   // back to the (zb = tap, split) packing used below
    } else {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n;
    const int tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * TBM, n0 = tile_n * TBN;

    const int split = z % a.nsplit;
    const int zb = z / a.nsplit;
    const int kbeg = split * a.kper;
    const int kend = min(a.K, kbeg + a.kper);
    const int ks0 = kbeg / BK;
    const int nks = (kend - kbeg + BK - 1) / BK;

    const float* Abase = a.A.p + (long long)zb * a.A.stride_z;
    const float* Bbase = a.B.p + (long long)zb * a.B.stride_z;
    // opaque pointer to 16 bytes of zeros (laundered so the compiler cannot fold the loads back into
    // selects, which would force a vmcnt(0) wait ahead of the MFMA block)
    const float* zp = fg.zero;

    RowInfo ri[NRA];
    if constexpr (AMODE == OP_IM2COL) {
        const int r0 = tid >> 3;
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
            const int gm = m0 + r0 + 32 * i;
            ri[i].img = 0;
            ri[i].oy = 0;
            ri[i].ox = 0;
            if (gm < a.M) {
                const int ox = gm % a.g.OW;
                const int t = gm / a.g.OW;
                const int oy = t % a.g.OH;
                const int img = t / a.g.OH;
                const int iy0 = oy * a.g.stride - a.g.pad, ix0 = ox * a.g.stride - a.g.pad;
                unsigned mask = 0;
                for (int ky = 0; ky < a.g.KH; ++ky)
                    for (int kx = 0; kx < a.g.KW; ++kx) {
                        const int iy = iy0 + ky, ix = ix0 + kx;
                        if (iy >= 0 && iy < a.g.IH && ix >= 0 && ix < a.g.IW) mask |= 1u << (ky * a.g.KW + kx);
                    }
                ri[i].img = (int)mask;
                ri[i].oy = (img * a.g.IH + iy0) * a.g.IW + ix0;   // pixel index of tap (0,0); only used when valid
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NRA; ++i) ri[i].img = ri[i].oy = ri[i].ox = 0;
    }
    RowInfo rib[NRB];
#pragma unroll
    for (int i = 0; i < NRB; ++i) rib[i].img = rib[i].oy = rib[i].ox = 0;

    // B operand of a conv: weights [N][tap][Ct] -> column of K-step ks is tap*Ct + chunk*32
    auto b_kstep_col = [&](int ks) -> int {
        if constexpr (AMODE == OP_IM2COL) {
            const int chunk = ks / fg.taps;
            const int tap = ks - chunk * fg.taps;
            return tap * (fg.ct_chunks * BK) + chunk * BK;
        } else {
            return ks * BK;
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    f32x4 ra[NRA], rb[NRB];
    auto store_b = [&](float* dst) {
        if constexpr (B_KC) store_kc_rows<NRB>(dst, rb);
        else store_tile<BMODE>(dst, rb);
    };
    auto store_a = [&](float* dst) {
        if constexpr (A_KC) store_kc_rows<NRA>(dst, ra);
        else store_tile<AMODE>(dst, ra);
    };
    auto load_b = [&](int ks) {
        if constexpr (BMODE == OP_KC && AMODE == OP_IM2COL) {
            // weights: plain KC load at a remapped column
            const int c4 = tid & 7, r0 = tid >> 3;
            const int col = b_kstep_col(ks) + c4 * 4;
#pragma unroll
            for (int i = 0; i < NRB; ++i) {
                const int gr = n0 + r0 + 32 * i;
                const bool ok = gr < a.N;
                rb[i] = ld4(ok ? Bbase + ((long long)gr * a.B.ld + col) : zp);
            }
        } else {
            fast_load<BMODE, NRB>(rb, a.B, a.g, fg, Bbase, n0, a.N, ks, kend, rib, zb, zp);
        }
    };

    fast_load<AMODE, NRA>(ra, a.A, a.g, fg, Abase, m0, a.M, ks0, kend, ri, zb, zp);
    load_b(ks0);
    store_a(As0);
    store_b(Bs0);
    __syncthreads();

    int cur = 0;
    for (int it = 0; it < nks; ++it) {
        const bool more = (it + 1) < nks;
        if (more) {
            fast_load<AMODE, NRA>(ra, a.A, a.g, fg, Abase, m0, a.M, ks0 + it + 1, kend, ri, zb, zp);
            load_b(ks0 + it + 1);
        }
        const float* As = As0 + cur * A_SZ;
        const float* Bs = Bs0 + cur * B_SZ;
        // fragment reads for 8 k-values (4 MFMA k-pairs) at a time, software-pipelined one group ahead
        float af[2][MI][4], bf[2][NI][4];
        auto read_frags = [&](int j, float (&fa)[MI][4], float (&fb)[NI][4]) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if constexpr (A_KC) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(As + (wr * (TBM / 2) + i * 32 + r) * KC_LD + 8 * j + 4 * h);
                    fa[i][0] = t[0]; fa[i][1] = t[1]; fa[i][2] = t[2]; fa[i][3] = t[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) fa[i][s] = As[(8 * j + 4 * h + s) * MC_LD + wr * 64 + i * 32 + r];
                }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if constexpr (B_KC) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(Bs + (wc * (TBN / 2) + i * 32 + r) * KC_LD + 8 * j + 4 * h);
                    fb[i][0] = t[0]; fb[i][1] = t[1]; fb[i][2] = t[2]; fb[i][3] = t[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) fb[i][s] = Bs[(8 * j + 4 * h + s) * MC_LD + wc * 64 + i * 32 + r];
                }
            }
        };
        read_frags(0, af[0], bf[0]);
#pragma unroll
        for (int j = 0; j < BK / 8; ++j) {
            if (j + 1 < BK / 8) read_frags(j + 1, af[(j + 1) & 1], bf[(j + 1) & 1]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int n = 0; n < NI; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j & 1][i][s], bf[j & 1][n][s], acc[i][n], 0, 0, 0);
        }
        if constexpr (NBUF == 2) {
            if (more) {
                store_a(As0 + (cur ^ 1) * A_SZ);
                store_b(Bs0 + (cur ^ 1) * B_SZ);
            }
            __syncthreads();
            cur ^= 1;
        } else {
            __syncthreads();
            if (more) {
                store_a(As0);
                store_b(Bs0);
                __syncthreads();
            }
        }
    }

    float* Cb = a.C + (long long)zb * a.c_stride_z + (long long)split * a.c_stride_split;
    const float* Rb = a.e.res ? a.e.res + (long long)zb * a.e.res_stride_z : nullptr;
    const bool rb_uniform = a.e.rowbias && (a.e.rows_per_img % 32 == 0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row_base = m0 + wr * (TBM / 2) + i * 32;          // wave-uniform, 32-aligned
        const int img_u = rb_uniform ? row_base / a.e.rows_per_img : 0;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int gn = n0 + wc * (TBN / 2) + n * 32 + r;
            if (gn >= a.N) continue;
            float bias = a.e.bias ? a.e.bias[gn] : 0.f;
            // time-embedding bias: one value per (image, channel); a 32-row tile never straddles images
            // (a block of rows entirely beyond M must not touch the row of a non-existent image)
            if (rb_uniform && row_base < a.M) bias += a.e.rowbias[(long long)img_u * a.e.ld_rowbias + gn];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int gm = row_base + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (gm >= a.M) continue;
                float x = acc[i][n][v] * a.e.alpha + bias;
                if (a.e.rowbias && !rb_uniform)
                    x += a.e.rowbias[(long long)(gm / a.e.rows_per_img) * a.e.ld_rowbias + gn];
                if (Rb) x += Rb[(long long)gm * a.e.ldres + gn];
                x *= a.e.out_scale;
                float* cp = Cb + (long long)gm * a.ldc + gn;
                if (a.e.accumulate) x += *cp;
                *cp = x;
            }
        }
    }
}

inline int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

template <int AMODE, int BMODE, int TBM, int NBUF, int TBN>
int launch_fast_impl(const TileArgs& a, FastGeom fg, int nz, hipStream_t stream, const char* name) {
    fg.zero = psld_detail_zero_page(name);
    if (!fg.zero) return PSLD_ERR_LAUNCH;
    constexpr bool A_KC = (AMODE == OP_KC || AMODE == OP_IM2COL);
    constexpr bool B_KC = (BMODE == OP_KC || BMODE == OP_IM2COL);
    constexpr int A_SZ = A_KC ? TBM * KC_LD : BK * MC_LD;
    constexpr int B_SZ = B_KC ? TBN * KC_LD : BK * MC_LD;
    constexpr size_t LDS = (size_t)NBUF * (A_SZ + B_SZ) * sizeof(float);
    static PsldPerDeviceFlag configured_; bool& configured = configured_.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_kernel_fast<AMODE, BMODE, TBM, NBUF, TBN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) {
            psld_set_error("%s: hipFuncSetAttribute failed: %s", name, hipGetErrorString(e));
            return PSLD_ERR_LAUNCH;
        }
        configured = true;
    }
    const long long tiles = (long long)cdiv(a.M, TBM) * cdiv(a.N, TBN);
    PSLD_CHECK_ARG(tiles < (1LL << 31) && nz * a.nsplit <= 65535, "%s: grid too large", name);
    dim3 grid((unsigned)tiles, (unsigned)(nz * a.nsplit));
    hipLaunchKernelGGL((tile_kernel_fast<AMODE, BMODE, TBM, NBUF, TBN>), grid, dim3(NTHREADS), LDS, stream, a, fg);
    PSLD_CHECK_LAUNCH(name);
    return PSLD_OK;
}

template <int AMODE, int BMODE, int TBM = 128>
int launch_fast(const TileArgs& a, FastGeom fg, int nz, hipStream_t stream, const char* name) {
    // Measured on MI355X (tools/bench_tile.py): the conv / wgrad loaders prefer one LDS buffer at 3 workgroups
    // per CU (+1.5 % / +3 %), the plain GEMM prefers two buffers at 2 per CU (+3 %).
    constexpr int NBUF = (AMODE == OP_IM2COL || BMODE == OP_SHIFT) ? 1 : 2;
    constexpr bool B_KC = (BMODE == OP_KC || BMODE == OP_IM2COL);
    if constexpr (B_KC && TBM == 128) {
        // 128x256 block tile (wave 64x128) when N is a multiple of 256 and the grid still fills the chip:
        // measured +2..3 % on the 32x32 layers and the 4096^3 GEMM (134 TF), neutral-to-worse on smaller grids
        if (a.N % 256 == 0 && (long long)cdiv(a.M, 128) * (a.N / 256) * nz >= 512)
            return launch_fast_impl<AMODE, BMODE, TBM, 1, 256>(a, fg, nz, stream, name);
    }
    return launch_fast_impl<AMODE, BMODE, TBM, NBUF, 128>(a, fg, nz, stream, name);
}

// out = epilogue(sum_s slabs[s]) for split-K convolutions (float4 along N)
__global__ void conv_reduce_epilogue_kernel(const float* __restrict__ slabs, int nsplit, int M, int N, float* __restrict__ out,
                                            int ldc, const Epilogue e) {
    const int n4 = N >> 2;
    const long long total = (long long)M * n4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i % n4);
        const int m = (int)(i / n4);
        const long long off = (long long)m * N + q * 4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(slabs + off);
        const long long slab = (long long)M * N;
        int s = 1;
        for (; s + 3 < nsplit; s += 4) {        // four independent loads in flight; the sum stays in slab order
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(slabs + s * slab + off);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(slabs + (s + 1) * slab + off);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(slabs + (s + 2) * slab + off);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(slabs + (s + 3) * slab + off);
            acc += v0;
            acc += v1;
            acc += v2;
            acc += v3;
        }
        for (; s < nsplit; ++s) acc += *reinterpret_cast<const f32x4*>(slabs + s * slab + off);
        acc *= e.alpha;
        if (e.bias) acc += *reinterpret_cast<const f32x4*>(e.bias + q * 4);
        if (e.rowbias) acc += *reinterpret_cast<const f32x4*>(e.rowbias + (long long)(m / e.rows_per_img) * e.ld_rowbias + q * 4);
        if (e.res) acc += *reinterpret_cast<const f32x4*>(e.res + (long long)m * e.ldres + q * 4);
        acc *= e.out_scale;
        float* cp = out + (long long)m * ldc + q * 4;
        if (e.accumulate) acc += *reinterpret_cast<const f32x4*>(cp);
        *reinterpret_cast<f32x4*>(cp) = acc;
    }
}

// The same pass when the consumer of the output is a GroupNorm (round 6): it also leaves the partial sums of what it stores,
// in the layout the limb kernels' epilogues write (psld_epilogue_t.gn_part: [64-row run][fine group of gn_fine channels][sum,
// sum of squares], float64) - so a split launch no longer costs its GroupNorm a pass over the tensor (gn_partial_kernel: 54
// launches of the B=128 step, every GroupNorm of the 8x8 level).  Workgroup = one 64-row run x 64 channels; thread = float4
// column q of the slab, rows rg + 16 i.  Fixed order: rows and components in the thread, the four row groups of a wave by two
// shuffles, the four waves through LDS - bitwise repeatable.
__global__ void __launch_bounds__(256) conv_reduce_epilogue_gn_kernel(const float* __restrict__ slabs, int nsplit, int M, int N,
                                                                        float* __restrict__ out, int ldc, const Epilogue e) {
    __shared__ float red[4][16][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, rg = 4 * wave + (lane >> 4);
    const int run = blockIdx.x, n0 = blockIdx.y * 64 + q * 4;
    const long long slab = (long long)M * N;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (e.bias) bias = *reinterpret_cast<const f32x4*>(e.bias + n0);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = run * 64 + rg + 16 * i;
        const long long off = (long long)m * N + n0;
        f32x4 acc = *reinterpret_cast<const f32x4*>(slabs + off);
        int s = 1;
        for (; s + 3 < nsplit; s += 4) {        // the plain kernel's order
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(slabs + s * slab + off);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(slabs + (s + 1) * slab + off);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(slabs + (s + 2) * slab + off);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(slabs + (s + 3) * slab + off);
            acc += v0;
            acc += v1;
            acc += v2;
            acc += v3;
        }
        for (; s < nsplit; ++s) acc += *reinterpret_cast<const f32x4*>(slabs + s * slab + off);
        acc *= e.alpha;
        acc += bias;
        if (e.rowbias) acc += *reinterpret_cast<const f32x4*>(e.rowbias + (long long)(m / e.rows_per_img) * e.ld_rowbias + n0);
        if (e.res) acc += *reinterpret_cast<const f32x4*>(e.res + (long long)m * e.ldres + n0);
        acc *= e.out_scale;
        *reinterpret_cast<f32x4*>(out + (long long)m * ldc + n0) = acc;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            s1 += acc[v];
            s2 += acc[v] * acc[v];
        }
    }
    s1 += __shfl_xor(s1, 16, 64);
    s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lane < 16) {
        red[wave][lane][0] = s1;
        red[wave][lane][1] = s2;
    }
    __syncthreads();
    if (tid < 16) {
        float t1 = ((red[0][tid][0] + red[1][tid][0]) + red[2][tid][0]) + red[3][tid][0];
        float t2 = ((red[0][tid][1] + red[1][tid][1]) + red[2][tid][1]) + red[3][tid][1];
        const int c0 = blockIdx.y * 64 + tid * 4;
        if (e.gn_fine == 4) {
            double* pp = e.gn_part + ((long long)run * (N >> 2) + (c0 >> 2)) * 2;
            pp[0] = (double)t1;
            pp[1] = (double)t2;
        } else if ((tid & 1) == 0) {
            const float u1 = ((red[0][tid + 1][0] + red[1][tid + 1][0]) + red[2][tid + 1][0]) + red[3][tid + 1][0];
            const float u2 = ((red[0][tid + 1][1] + red[1][tid + 1][1]) + red[2][tid + 1][1]) + red[3][tid + 1][1];
            double* pp = e.gn_part + ((long long)run * (N >> 3) + (c0 >> 3)) * 2;
            pp[0] = (double)(t1 + u1);
            pp[1] = (double)(t2 + u2);
        }
    }
}

}  // namespace

const float* psld_detail_zero_page(const char* name) {
    static const float* zero_dev = nullptr;
    if (!zero_dev) {
        void* zptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&zptr, HIP_SYMBOL(g_zero_page));
        if (e != hipSuccess || !zptr) {
            psld_set_error("%s: hipGetSymbolAddress failed: %s", name, hipGetErrorString(e));
            return nullptr;
        }
        zero_dev = static_cast<const float*>(zptr);
    }
    return zero_dev;
}

bool psld_detail_conv_reduce_gn_ok(int M, int N, const PsldEpilogue& e) {
    return M % 64 == 0 && N % 64 == 0 && e.gn_hw > 0 && e.gn_hw % 64 == 0 && !e.accumulate;
}

int psld_detail_conv_reduce_epilogue(const float* slabs, int nsplit, int M, int N, float* y, int ldy,
                                     const PsldEpilogue& e, hipStream_t stream) {
    if (e.gn_part) {
        // callers split a launch whose epilogue forms GroupNorm sums only when psld_detail_conv_reduce_gn_ok says so
        PSLD_CHECK_ARG(psld_detail_conv_reduce_gn_ok(M, N, e), "conv_reduce_epilogue: GroupNorm sums need M, N, gn_hw %% 64 == 0, 16-byte rows and no accumulation");
        hipLaunchKernelGGL(conv_reduce_epilogue_gn_kernel, dim3((unsigned)(M / 64), (unsigned)(N / 64)), dim3(256), 0, stream, slabs, nsplit, M, N,
                           y, ldy, e);
        PSLD_CHECK_LAUNCH("conv_reduce_epilogue_gn_kernel");
        return PSLD_OK;
    }
    const long long total = (long long)M * (N / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_reduce_epilogue_kernel, dim3(blocks), dim3(256), 0, stream, slabs, nsplit, M, N, y, ldy, e);
    PSLD_CHECK_LAUNCH("conv_reduce_epilogue_kernel");
    return PSLD_OK;
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" int psld_gemm_f32(int trans_a, int trans_b, int M, int N, int K,
                             const float* A, int lda, long long stride_a,
                             const float* B, int ldb, long long stride_b,
                             float* C, int ldc, long long stride_c, int batch,
                             const psld_epilogue_t* epi, hipStream_t stream) {
    PSLD_CHECK_ARG(A && B && C, "psld_gemm_f32: null pointer");
    PSLD_CHECK_ARG(M >= 0 && N >= 0 && K > 0 && batch >= 1, "psld_gemm_f32: bad shape %d %d %d x%d", M, N, K, batch);
    TileArgs a{};
    a.M = M; a.N = N; a.K = K;
    a.A = {A, nullptr, stride_a, lda, 0};
    a.B = {B, nullptr, stride_b, ldb, 0};
    a.C = C; a.c_stride_z = stride_c; a.c_stride_split = 0; a.ldc = ldc;
    a.nsplit = 1; a.kper = cdiv(K, BK) * BK;
    a.e = make_epilogue(epi);
    PSLD_CHECK_ARG(!a.e.gn_part, "fp32 tile engine: psld_epilogue_t.gn_part is a limb-kernel feature");
    // op(A) is M x K.  trans_a == 0: A stored [M][K] (K contiguous);  1: stored [K][M].
    // op(B) is K x N.  trans_b == 0: B stored [K][N] (N contiguous);  1: stored [N][K].
    if (!trans_a) a.A.vec = aligned16(A) && lda % 4 == 0 && K % 4 == 0 && stride_a % 4 == 0;
    else          a.A.vec = aligned16(A) && lda % 4 == 0 && M % 4 == 0 && stride_a % 4 == 0;
    if (trans_b)  a.B.vec = aligned16(B) && ldb % 4 == 0 && K % 4 == 0 && stride_b % 4 == 0;
    else          a.B.vec = aligned16(B) && ldb % 4 == 0 && N % 4 == 0 && stride_b % 4 == 0;
    const bool fastok = a.A.vec && a.B.vec && K % BK == 0 && M > 0 && N > 0;
    FastGeom fg{1, 0, 0, 0, nullptr};
    if (fastok) {
        const bool small = (long long)cdiv(M, BM) * cdiv(N, BN) * batch <= 256;
        if (!trans_a && trans_b && small && M > 64)
            return launch_fast<OP_KC, OP_KC, 64>(a, fg, batch, stream, "psld_gemm_f32[NT,fast64]");
        if (!trans_a && trans_b)  return launch_fast<OP_KC, OP_KC>(a, fg, batch, stream, "psld_gemm_f32[NT,fast]");
        if (!trans_a && !trans_b) return launch_fast<OP_KC, OP_MC>(a, fg, batch, stream, "psld_gemm_f32[NN,fast]");
        if (trans_a && !trans_b)  return launch_fast<OP_MC, OP_MC>(a, fg, batch, stream, "psld_gemm_f32[TN,fast]");
    }
    if (!trans_a && trans_b)  return launch<OP_KC, OP_KC>(a, batch, stream, "psld_gemm_f32[NT]");
    if (!trans_a && !trans_b) return launch<OP_KC, OP_MC>(a, batch, stream, "psld_gemm_f32[NN]");
    if (trans_a && !trans_b)  return launch<OP_MC, OP_MC>(a, batch, stream, "psld_gemm_f32[TN]");
    return launch<OP_MC, OP_KC>(a, batch, stream, "psld_gemm_f32[TT]");
}

// Split-K TN GEMM into slabs: slabs[s][M][N] = A[ks..ke]^T B[ks..ke]; caller reduces.
extern "C" int psld_gemm_tn_splitk_f32(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                                       float* slabs, int nsplit, hipStream_t stream) {
    PSLD_CHECK_ARG(A && B && slabs && nsplit >= 1, "psld_gemm_tn_splitk_f32: bad args");
    TileArgs a{};
    a.M = M; a.N = N; a.K = K;
    a.A = {A, nullptr, 0, lda, aligned16(A) && lda % 4 == 0 && M % 4 == 0};
    a.B = {B, nullptr, 0, ldb, aligned16(B) && ldb % 4 == 0 && N % 4 == 0};
    a.C = slabs; a.c_stride_z = 0; a.c_stride_split = (long long)M * N; a.ldc = N;
    a.nsplit = nsplit; a.kper = cdiv(cdiv(K, nsplit), BK) * BK;
    a.e = make_epilogue(nullptr);
    if (a.A.vec && a.B.vec && M > 0 && N > 0) {
        FastGeom fg{1, 0, 0, 0, nullptr};
        return launch_fast<OP_MC, OP_MC>(a, fg, 1, stream, "psld_gemm_tn_splitk_f32[fast]");
    }
    return launch<OP_MC, OP_MC>(a, 1, stream, "psld_gemm_tn_splitk_f32");
}

extern "C" long long psld_conv2d_workspace_bytes(int batch, int oh, int ow, int cout) {
    return 8LL * batch * oh * ow * cout * (long long)sizeof(float);
}

// Same as psld_conv2d_nhwc_f32 plus an optional workspace: when the output grid cannot fill the chip
// (small batches, 8x8 / 16x16 layers) the K range (channel chunks x taps) is split over extra workgroups
// that write partial slabs, and one pass applies the epilogue to their sum.
extern "C" int psld_conv2d_nhwc_ws_f32(const float* x1, int c1, const float* x2, int c2,
                                       int batch, int ih, int iw,
                                       const float* w_ohwi, int cout, int kh, int kw,
                                       int stride, int pad, int transposed_stride,
                                       int oh, int ow, float* y, int ldy,
                                       const psld_epilogue_t* epi, void* workspace, long long ws_bytes,
                                       hipStream_t stream) {
    PSLD_CHECK_ARG(x1 && w_ohwi && y, "psld_conv2d_nhwc_f32: null pointer");
    PSLD_CHECK_ARG(c1 > 0 && c2 >= 0 && (c2 == 0 || x2), "psld_conv2d_nhwc_f32: bad channel split");
    PSLD_CHECK_ARG(stride >= 1 && transposed_stride >= 1, "psld_conv2d_nhwc_f32: bad stride");
    const int ct = c1 + c2;
    TileArgs a{};
    a.M = batch * oh * ow; a.N = cout; a.K = kh * kw * ct;
    a.A = {x1, x2, 0, 0, 0};
    a.A.vec = aligned16(x1) && (!x2 || aligned16(x2)) && c1 % 4 == 0 && c2 % 4 == 0;
    a.B = {w_ohwi, nullptr, 0, a.K, aligned16(w_ohwi) && a.K % 4 == 0};
    a.C = y; a.ldc = ldy; a.c_stride_z = 0; a.c_stride_split = 0;
    a.nsplit = 1; a.kper = cdiv(a.K, BK) * BK;
    a.g = {ih, iw, c1, c2, oh, ow, kh, kw, stride, pad, transposed_stride};
    a.e = make_epilogue(epi);
    PSLD_CHECK_ARG(!a.e.gn_part, "fp32 tile engine: psld_epilogue_t.gn_part is a limb-kernel feature");
    if (a.A.vec && a.B.vec && c1 % BK == 0 && c2 % BK == 0 && transposed_stride == 1 && a.M > 0 && kh * kw <= 31) {
        FastGeom fg{kh * kw, 0, 0, ct / BK, nullptr};
        const long long tiles128 = (long long)cdiv(a.M, BM) * cdiv(a.N, BN);
        const long long tiles64 = (long long)cdiv(a.M, 64) * cdiv(a.N, BN);
        const int ksteps = a.K / BK;
        // split-K when even the 64-row tiling leaves most of the 256 CUs x 3 slots empty
        int ns = 1;
        if (workspace && tiles64 <= 256 && a.M > 64 && cout % 4 == 0 && ldy % 4 == 0 && aligned16(y) &&
            (!a.e.bias || aligned16(a.e.bias)) && (!a.e.rowbias || (aligned16(a.e.rowbias) && a.e.ld_rowbias % 4 == 0)) &&
            (!a.e.res || (aligned16(a.e.res) && a.e.ldres % 4 == 0))) {
            ns = (int)(768 / tiles64);
            if (ns > 8) ns = 8;
            if (ns > ksteps / 6) ns = ksteps / 6;
            if ((long long)ns * a.M * a.N * (long long)sizeof(float) > ws_bytes) ns = 1;
        }
        if (ns >= 2) {
            TileArgs p = a;
            p.C = reinterpret_cast<float*>(workspace);
            p.ldc = a.N;
            p.c_stride_split = (long long)a.M * a.N;
            p.nsplit = ns;
            p.kper = cdiv(ksteps, ns) * BK;
            p.e = make_epilogue(nullptr);
            int st = launch_fast<OP_IM2COL, OP_KC, 64>(p, fg, 1, stream, "psld_conv2d_nhwc_f32[fast64,splitK]");
            if (st != PSLD_OK) return st;
            return psld_detail_conv_reduce_epilogue(p.C, ns, a.M, a.N, y, ldy, a.e, stream);
        }
        if (tiles128 <= 256 && a.M > 64)
            return launch_fast<OP_IM2COL, OP_KC, 64>(a, fg, 1, stream, "psld_conv2d_nhwc_f32[fast64]");
        return launch_fast<OP_IM2COL, OP_KC>(a, fg, 1, stream, "psld_conv2d_nhwc_f32[fast]");
    }
    return launch<OP_IM2COL, OP_KC>(a, 1, stream, "psld_conv2d_nhwc_f32");
}

extern "C" int psld_conv2d_nhwc_f32(const float* x1, int c1, const float* x2, int c2,
                                    int batch, int ih, int iw,
                                    const float* w_ohwi, int cout, int kh, int kw,
                                    int stride, int pad, int transposed_stride,
                                    int oh, int ow, float* y, int ldy,
                                    const psld_epilogue_t* epi, hipStream_t stream) {
    return psld_conv2d_nhwc_ws_f32(x1, c1, x2, c2, batch, ih, iw, w_ohwi, cout, kh, kw, stride, pad, transposed_stride,
                                   oh, ow, y, ldy, epi, nullptr, 0, stream);
}

// dW slabs: slabs[s][cout][kh*kw][cin_total] restricted to columns [col0, col0+cin) for input x.
extern "C" int psld_conv2d_wgrad_nhwc_f32(const float* dy, int lddy, int cout,
                                          const float* x, int cin, int batch, int ih, int iw,
                                          int kh, int kw, int stride, int pad, int oh, int ow,
                                          float* slabs, int cin_total, int col0, int nsplit,
                                          hipStream_t stream) {
    PSLD_CHECK_ARG(dy && x && slabs && nsplit >= 1, "psld_conv2d_wgrad_nhwc_f32: bad args");
    TileArgs a{};
    a.M = cout; a.N = cin; a.K = batch * oh * ow;
    a.A = {dy, nullptr, 0, lddy, aligned16(dy) && lddy % 4 == 0 && cout % 4 == 0};
    a.B = {x, nullptr, 0, cin, aligned16(x) && cin % 4 == 0};
    a.C = slabs + col0;
    a.ldc = kh * kw * cin_total;
    a.c_stride_z = cin_total;                                  // tap -> column block
    a.c_stride_split = (long long)cout * kh * kw * cin_total;  // slab
    a.nsplit = nsplit; a.kper = cdiv(cdiv(a.K, nsplit), BK) * BK;
    a.g = {ih, iw, cin, 0, oh, ow, kh, kw, stride, pad, 1};
    a.e = make_epilogue(nullptr);
    const int ows = ilog2_exact(ow), ohs = ilog2_exact(oh);
    if (a.A.vec && a.B.vec && ows >= 0 && ohs >= 0 && cout > 0 && cin > 0) {
        FastGeom fg{kh * kw, ows, ohs, 0, nullptr};
        return launch_fast<OP_MC, OP_SHIFT>(a, fg, kh * kw, stream, "psld_conv2d_wgrad_nhwc_f32[fast]");
    }
    return launch<OP_MC, OP_SHIFT>(a, kh * kw, stream, "psld_conv2d_wgrad_nhwc_f32");
}
