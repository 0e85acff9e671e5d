// 3x3 stride-1 convolutions (forward, data-gradient, weight-gradient) on bf16 MFMA with fp32 operands carried
// as three bf16 limbs ("bf16x6": PSLD_MATH_BF16X6 in include/psld_hip.h).
//
// Why: v_mfma_f32_32x32x2_f32 tops out at ~136 TFLOP/s on this chip; v_mfma_f32_32x32x16_bf16 moves 16x the
// k-depth per instruction.  Every fp32 value x is decomposed EXACTLY into hi + mid + lo (bf16 each, 8 significant
// bits apiece, 24 in total) and a product a*b is accumulated in fp32 from the six limb products of weight
// >= 2^-16: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi.  The three dropped products are < 2^-23 |a*b| —
// below the rounding of the fp32 product itself — so results agree with the fp32 MFMA path to fp32 rounding
// (tests/test_kernels_gpu.py runs every parity case in both modes).
//
// Why a direct convolution rather than the im2col tile engine: the limb split costs VALU work and LDS stores,
// so the A operand is converted ONCE per 32-channel chunk — a halo tile of (rows+2) x (W+2) pixels kept in LDS
// and read at nine shifted offsets — and the weights are pre-split once per optimizer step into MFMA fragment
// order, so a wave fetches its B fragments straight from L2 with one coalesced 16-byte load per lane: no LDS
// staging, no conversion and no barrier per tap.  Per tap a wave issues 12 global loads, 12 ds_read_b128 and
// 96 v_mfma_f32_16x16x32_bf16.
//
// Replaces nn.Conv2d 3x3 (song_sde/layers.py:85-109 via layerspp.py:29-39) forward and backward for the layers
// whose channel counts are multiples of 32 (in) / 128 (out) — every ResBlock conv of the C10 / CelebA-64 nets.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "psld_hip.h"
#include "tile_shared.h"
#include "limb.h"

namespace {

constexpr int TAP_U4 = 4 * 3 * 64;   // uint4 per (wave column, tap, 32-channel chunk): [n-block 4][limb 3][lane 64]

int g_math_mode = -1;
inline int math_mode() {
    if (g_math_mode < 0) {
        const char* e = getenv("PSLD_MATH");
        g_math_mode = (e && !strcmp(e, "f32")) ? PSLD_MATH_F32 : PSLD_MATH_BF16X6;
    }
    return g_math_mode;
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- weights -> limb fragments ---------------------------------------------------------------------------
// Operand order of v_mfma_f32_16x16x32_bf16: lane l holds B[k = 8*(l >> 4) + j][col = l & 15], j = 0..7.
// out[nt][wc][chunk][tap][nb][limb][lane] (uint4 = 8 bf16): n = nt*128 + wc*64 + nb*16 + (lane & 15),
// k = chunk*32 + (lane >> 4)*8 + j.   dgrad = 0: B[n][k] = w[co = n][ci = k][tap];
// dgrad = 1: B[n][k] = w[co = k][ci = n][8 - tap] (the data-gradient of a stride-1 pad-1 3x3 conv is the same
// conv with the taps flipped and the channel roles swapped).
// General form: B[n][k] for tap t is w[n*sn + k*sk + (flip ? taps-1-t : t)*st]; taps = 1 gives the fragments of a
// plain NT GEMM (the pointwise kernel: 1x1 convolutions, NIN projections).
// One work item = one lane slot (nt, wc, chunk, nb, lane) for ALL taps and limbs: its 8 k-values x taps source
// elements are neighbours in memory (OIHW keeps the 9 taps of a (co, ci) pair together), and split3 yields the three
// limbs at once.
// chunk0 / chunks_total: the tensor fills the 32-wide K chunks chunk0 .. chunk0 + k_in/32 of a fragment set whose K
// dimension has chunks_total chunks (several parameters concatenated along K: the attention q | k | v data gradient).
__device__ __forceinline__ void pack_frag_item(const float* __restrict__ w, u32x4* __restrict__ out, long long item,
                                               int k_in, int taps, long long sn, long long sk, long long st, int flip,
                                               int chunk0 = 0, int chunks_total = 0) {
    const int chunks = k_in / 32;
    if (chunks_total == 0) chunks_total = chunks;
    long long t = item;
    const int lane = (int)(t & 63); t >>= 6;
    const int nb = (int)(t & 3); t >>= 2;
    const int chunk = (int)(t % chunks); t /= chunks;
    const int wc = (int)(t & 1); t >>= 1;
    const int nt = (int)t;
    const int n = nt * 128 + wc * 64 + nb * 16 + (lane & 15);
    const int k0 = chunk * 32 + (lane >> 4) * 8;
    const float* src = w + n * sn + k0 * sk;
    // uint4 index of (tap, limb) for this slot: ((((nt*2 + wc)*chunks + chunk)*taps + tap)*4 + nb)*3 + limb)*64 + lane
    const long long base = ((long long)(nt * 2 + wc) * chunks_total + chunk0 + chunk) * taps;
    for (int tap = 0; tap < taps; ++tap) {
        const float* p = src + (flip ? taps - 1 - tap : tap) * st;
        unsigned hi[4], mid[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split3(p[(2 * j) * sk], p[(2 * j + 1) * sk], hi[j], mid[j], lo[j]);
        u32x4* o = out + (((base + tap) * 4 + nb) * 3) * 64 + lane;
        o[0] = u32x4{hi[0], hi[1], hi[2], hi[3]};
        o[64] = u32x4{mid[0], mid[1], mid[2], mid[3]};
        o[128] = u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
}

__global__ void pack_frag_kernel(const float* __restrict__ w, u32x4* __restrict__ out, int n_out, int k_in, int taps,
                                 long long sn, long long sk, long long st, int flip) {
    const long long items = (long long)n_out * (k_in / 32) * 4;     // n_out/128 * 2 * chunks * 4 * 64
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (long long)gridDim.x * blockDim.x)
        pack_frag_item(w, out, i, k_in, taps, sn, sk, st, flip);
}

// Many weight tensors in one launch.  tab[8*i ..]: src pointer, dst pointer, n_out, k_in | chunk0 << 20 | chunks_total << 40,
// taps | flip << 32, sn, sk, first work item of tensor i in the launch-wide numbering (st = 1; a tensor has
// n_out * k_in / 8 items; chunk0 = chunks_total = 0: the tensor is the whole K dimension of its fragment set).
__global__ void pack_frag_batch_kernel(const long long* __restrict__ tab, int ntab, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = ntab - 1;
        while (lo < hi) {                       // last entry whose first item is <= idx
            const int mid = (lo + hi + 1) >> 1;
            if (tab[8 * mid + 7] <= idx) lo = mid; else hi = mid - 1;
        }
        const long long* d = tab + 8 * lo;
        pack_frag_item(reinterpret_cast<const float*>(d[0]), reinterpret_cast<u32x4*>(d[1]), idx - d[7], (int)(d[3] & 0xfffff),
                       (int)(d[4] & 0xffffffffLL), d[5], d[6], 1, (int)(d[4] >> 32), (int)((d[3] >> 20) & 0xfffff),
                       (int)((d[3] >> 40) & 0xfffff));
    }
}

int launch_pack(const float* w, void* out, int n_out, int k_in, int taps, long long sn, long long sk, long long st,
                int flip, hipStream_t stream, const char* name) {
    const long long total = (long long)n_out * (k_in / 32) * 4;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(blocks), dim3(256), 0, stream, w, reinterpret_cast<u32x4*>(out), n_out,
                       k_in, taps, sn, sk, st, flip);
    PSLD_CHECK_LAUNCH(name);
    return PSLD_OK;
}

// ---- forward / data-gradient -------------------------------------------------------------------------------
struct DConvArgs {
    const float* x1;
    const float* x2;
    int C1, C2;
    int B, H, W;            // stride 1, pad 1: output spatial == input spatial
    const u32x4* wfrag;
    int N, M;               // cout (multiple of 128), B*H*W
    int chunks;             // (C1 + C2) / 32
    int chunks_per_split;
    float* C;
    int ldc;
    long long c_stride_split;
    int nseg, rps;          // image segments per 128-pixel tile and output rows per segment
    int pitch;              // pixel rows of the LDS halo image per image row: W + 2, or 16 for 8-wide maps on 64-row tiles (see dconv_pitch)
    PsldEpilogue e;
    const float* zero;
    int v4;                 // rows of C / residual / bias / row bias are 16-byte aligned: dwordx4 epilogue (set by plan_split)
};

// Fused epilogue of the limb kernels.  They issue the MFMAs with the WEIGHT fragment as the first operand, i.e. they
// accumulate the transposed block D^T[channel][pixel]: in the 16x16 C/D layout (col = lane & 15, row = 4*(lane >> 4) + v)
// a lane then holds FOUR CONSECUTIVE CHANNELS of ONE pixel per block - a 16-byte run of the NHWC row - so output, residual,
// previous output and biases move as dwordx4 (16 memory instructions per block row instead of 64 single-dword ones).  Rows
// of C / residual / bias / row bias must be 16-byte aligned (checked by the entry points).
// Wave tile: 128 (pixels) x 32 (channels), 8 x 2 accumulator blocks at (m0, n0 + wave*32): the four waves of a workgroup
// split the 128 output channels (each weight fragment is loaded by ONE wave: half the L2 -> CU fragment stream of a 2 x 2
// arrangement of 64 x 64 tiles).  GroupNorm partial sums per 64-row run (block rows 0-3 / 4-7).
// Wave-uniform values pinned to scalar registers.  The epilogue parameters arrive in the by-value kernel argument; left to
// itself hipcc keeps part of that structure in SCRATCH and reloads out_scale / accumulate in front of every store - and a
// scratch load is a vector-memory load, so each reload waits (vmcnt(0)) for every store issued before it: the 16 stores
// of a wave went out one memory round trip at a time (timing ablation of the pointwise kernel without its epilogue:
// 243 -> 171 us; the loop of a 128 x 256 x 512 tile is only ~25 us long).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
template <class T>
__device__ __forceinline__ T* uni(T* p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    // back through a GLOBAL address-space pointer: a bare integer -> pointer cast is a generic pointer, and its accesses
    // become flat_load / flat_store (which also tick the LDS counter and can only be waited for with vmcnt(0) lgkmcnt(0))
    typedef __attribute__((address_space(1))) T G;
    return (T*)(G*)(((unsigned long long)hi << 32) | lo);
}

template <int MBK>      // 8: 128-row tile, 4: 64-row tile (small grids)
__device__ __forceinline__ void dconv_epilogue_n32(const DConvArgs& a, f32x4v (&acc)[MBK][2], int m0, int nw0, int lane,
                                                   int split) {
    constexpr int RUNS = MBK / 4;       // 64-row runs of the tile
    const int r16 = lane & 15, kq = lane >> 4;
    const int M = uni(a.M), ldc = uni(a.ldc);
    float* Cb = uni(a.C) + (long long)split * a.c_stride_split;
    const float alpha = uni(a.e.alpha), out_scale = uni(a.e.out_scale);
    const bool accumulate = uni(a.e.accumulate) != 0;
    const float* e_bias = uni(a.e.bias);
    const float* e_rowbias = uni(a.e.rowbias);
    const float* e_res = uni(a.e.res);
    const int ldres = uni(a.e.ldres), rows_per_img = uni(a.e.rows_per_img), ld_rowbias = uni(a.e.ld_rowbias);
    const PsldEpilogue& e = a.e;
    const bool rb_uniform = e_rowbias && (rows_per_img % 16 == 0);
    const int cn0 = nw0 + 4 * kq;
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4v bias4[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
        bias4[nb] = e_bias ? *reinterpret_cast<const f32x4v*>(e_bias + cn0 + nb * 16) : zero4;
    float gs[RUNS][2], gss[RUNS][2];     // [64-row run][block column]
#pragma unroll
    for (int r = 0; r < RUNS; ++r)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) gs[r][nb] = gss[r][nb] = 0.f;
    // STRAIGHT-LINE code, no branch around a memory instruction: loads and stores share one in-order counter on this chip,
    // and behind a branch the compiler can only wait for "everything" (vmcnt(0)) - i.e. for the stores of the previous
    // block row as well.  Absent operands (no residual / previous output / row bias) are read from the 16-byte zero
    // page instead, rows beyond M are clamped for the loads and masked for the store, and the loads of block row mb + 1
    // are issued BEFORE the stores of row mb, so waiting for them does not wait for those stores.
    const float* zp = a.zero;
    const bool any_load = e_res || accumulate || e_rowbias;
    auto row_loads = [&](int mb, f32x4v (&rv)[2], f32x4v (&cv)[2], f32x4v (&tb)[2]) {
        const int row_base = m0 + mb * 16;
        const int gm = min(row_base + r16, M - 1);
        const long long coff = (long long)gm * ldc, roff = (long long)gm * ldres;
        const long long toff = (long long)((rb_uniform ? min(row_base, M - 1) : gm) / rows_per_img) * ld_rowbias;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int gn = cn0 + nb * 16;
            rv[nb] = *reinterpret_cast<const f32x4v*>(e_res ? e_res + roff + gn : zp);
            cv[nb] = *reinterpret_cast<const f32x4v*>(accumulate ? Cb + coff + gn : zp);
            tb[nb] = *reinterpret_cast<const f32x4v*>(e_rowbias ? e_rowbias + toff + gn : zp);
        }
    };
    auto row_store = [&](int mb, const f32x4v (&rv)[2], const f32x4v (&cv)[2], const f32x4v (&tb)[2]) {
        const int row_base = m0 + mb * 16;
        const int gm = min(row_base + r16, M - 1);
        const bool ok = row_base + r16 < M;
        const long long coff = (long long)gm * ldc;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int gn = cn0 + nb * 16;
            // x = ((acc*alpha + bias + rowbias) + res) * out_scale + prev: the additive terms in the scalar form's order
            f32x4v o = acc[mb][nb] * alpha + (bias4[nb] + tb[nb]);
            o += rv[nb];
            o *= out_scale;
            o += cv[nb];
            if (ok) {
                *reinterpret_cast<f32x4v*>(Cb + coff + gn) = o;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    gs[mb >> 2][nb] += o[v];
                    gss[mb >> 2][nb] += o[v] * o[v];
                }
            }
        }
    };
    if (any_load) {
        f32x4v rv[2][2], cv[2][2], tb[2][2];
        row_loads(0, rv[0], cv[0], tb[0]);
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb) {
            if (mb + 1 < MBK) row_loads(mb + 1, rv[(mb + 1) & 1], cv[(mb + 1) & 1], tb[(mb + 1) & 1]);
            row_store(mb, rv[mb & 1], cv[mb & 1], tb[mb & 1]);
        }
    } else {
        const f32x4v z2[2] = {zero4, zero4};
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb) row_store(mb, z2, z2, z2);
    }
    if (e.gn_part) {
        // a lane holds 4 channels of a block column (quad kq = lane >> 4), its 16-lane row the 16 pixels of a block row: the
        // butterfly over sft = 1..8 sums the pixels (four-channel sums, gn_fine = 4), sft = 16 adds the partner quad (eight)
        const bool fine4 = e.gn_fine == 4;
        const int fine = fine4 ? a.N >> 2 : a.N >> 3, chunks = e.gn_hw >> 6;
#pragma unroll
        for (int r = 0; r < RUNS; ++r) {
            const int row0 = m0 + r * 64;
            if (row0 >= a.M) continue;
            const int img = row0 / e.gn_hw, chunk = (row0 - img * e.gn_hw) >> 6;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                float s1 = gs[r][nb], s2 = gss[r][nb];
#pragma unroll
                for (int sft = 1; sft <= 8; sft <<= 1) {
                    s1 += __shfl_xor(s1, sft, 64);
                    s2 += __shfl_xor(s2, sft, 64);
                }
                if (fine4) {
                    if ((lane & 0xf) == 0) {
                        const int f = ((nw0 + nb * 16) >> 2) + (lane >> 4);
                        double* pp = e.gn_part + (((long long)img * chunks + chunk) * fine + f) * 2;
                        pp[0] = (double)s1;
                        pp[1] = (double)s2;
                    }
                    continue;
                }
                s1 += __shfl_xor(s1, 16, 64);
                s2 += __shfl_xor(s2, 16, 64);
                if ((lane & 0x1f) == 0) {
                    const int f = ((nw0 + nb * 16) >> 3) + (lane >> 5);
                    double* pp = e.gn_part + (((long long)img * chunks + chunk) * fine + f) * 2;
                    pp[0] = (double)s1;
                    pp[1] = (double)s2;
                }
            }
        }
    }
}

// NH = float4 staging items per thread and stage: the LDS image has NH*32 pixel rows per limb.
// TAPS = K steps (of 32 channels) served by one staged image: the 9 filter taps of a 3x3 convolution (PW = false:
// the image is the halo tile of ONE 32-channel chunk, >= nseg*(rps+2)*(W+2) rows), or, for the pointwise kernel
// (PW = true: plain NT GEMM / 1x1 convolution, the image is the tile's 128 rows), TAPS consecutive 32-channel chunks
// stored one after the other (NH = 4*TAPS).
// The four waves of a workgroup split the 128 output channels (wave tile 128 x 32: each weight fragment is loaded by one
// wave; the 2 x 2 arrangement of 64 x 64 tiles of round 1 loaded every fragment twice).
// MT: pixel rows per workgroup tile, 128 or - 3x3 only - 64 (wave tile 64 x 32): twice the workgroups for output
// grids too small to fill the chip, instead of (or with a shallower) split of the K range.
template <int NH, int TAPS, bool PW, int MT = 128>
__global__ void __launch_bounds__(256, 2) dconv_kernel(const DConvArgs a) {
    static_assert(MT == 128 || (MT == 64 && !PW), "64-row tiles exist for the 3x3 kernels");
    constexpr int MBK = MT / 16, NBK = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LIMB = NH * 32 * ROWB;
    static_assert(!PW || NH == 4 * TAPS, "pointwise staging: 128 rows x 8 quads per 32-channel chunk");

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1;
    const int c4 = tid & 7;

    const int tiles_n = a.N >> 7;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * MT, n0 = tile_n * 128;
    const int split = blockIdx.y;
    const int c_beg = split * a.chunks_per_split;               // stages: chunks (conv) or groups of TAPS chunks (PW)
    const int c_end = min(a.chunks, c_beg + a.chunks_per_split);

    const int W2 = a.pitch;          // >= W + 2 (columns beyond W + 1 are never read)
    const float* zp = a.zero;

    // source pixel of every item this thread stages (-1: zero padding / beyond the batch)
    int hoff[NH];
    if constexpr (PW) {
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int gm = m0 + (((tid + 256 * i) >> 3) & 127);
            hoff[i] = gm < a.M ? gm : -1;
        }
    } else {
        const int HW = a.H * a.W;
        const int img0 = m0 / HW;
        const int oy0 = (m0 - img0 * HW) / a.W;     // 0 when a tile holds whole images
        const int seg_px = (a.rps + 2) * W2;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int px = (tid + 256 * i) >> 3;
            const int seg = px / seg_px;
            const int rem = px - seg * seg_px;
            const int hr = rem / W2, hx = rem - hr * W2;
            const int img = img0 + seg, iy = oy0 + hr - 1, ix = hx - 1;
            const bool ok = seg < a.nseg && img < a.B && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            hoff[i] = ok ? (img * a.H + iy) * a.W + ix : -1;
        }
    }
    f32x4 hv[NH];
    auto load_halo = [&](int c) {
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int c0 = (PW ? c * TAPS + (i >> 2) : c) * 32;   // item i of a pointwise stage belongs to chunk i / 4
            const bool second = c0 >= a.C1;
            const float* src = second ? a.x2 : a.x1;
            const int cs = second ? a.C2 : a.C1;
            const int cc = (second ? c0 - a.C1 : c0) + c4 * 4;
            hv[i] = ld4(hoff[i] >= 0 ? src + ((long long)hoff[i] * cs + cc) : zp);
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            unsigned h0, m0_, l0, h1, m1, l1;
            split3(hv[i][0], hv[i][1], h0, m0_, l0);
            split3(hv[i][2], hv[i][3], h1, m1, l1);
            const int prow = (tid + 256 * i) >> 3;
            unsigned char* q = smem + prow * ROWB + (((c4 >> 1) ^ lds_swz(prow)) << 4) + (c4 & 1) * 8;
            *reinterpret_cast<u32x2*>(q) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(q + LIMB) = u32x2{m0_, m1};
            *reinterpret_cast<u32x2*>(q + 2 * LIMB) = u32x2{l0, l1};
        }
    };

    // v_mfma_f32_16x16x32_bf16 (under this load it sustains a ~15 % higher clock than 32x32x16: measured 212 vs 184
    // TFLOP/s on 256->256 @32x32): the 128x32 wave tile is 8x2 blocks, one 32-deep K step per tap and chunk.
    // Lane l holds A[row = l & 15][k = 8*(l >> 4) + j]: LDS pixel row of its rows at tap (0, 0)
    const int r16 = lane & 15, kq = lane >> 4;
    int abase[MBK];
#pragma unroll
    for (int mb = 0; mb < MBK; ++mb) {
        const int ml = mb * 16 + r16;
        if constexpr (PW) {
            abase[mb] = ml;
        } else {
            const int seg = ml / (a.rps * a.W);
            const int rem = ml - seg * (a.rps * a.W);
            const int ry = rem / a.W, ox = rem - ry * a.W;
            abase[mb] = (seg * (a.rps + 2) + ry) * W2 + ox;
        }
    }

    // B fragments of K step sigma = stage*TAPS + tap for this wave's 64 columns
    const u32x4* wp = a.wfrag + ((long long)(tile_n * 2 + wc) * a.chunks * TAPS) * TAP_U4 + lane +
                      (wave & 1) * 2 * 3 * 64;                  // a 32-column wave: blocks 2*(wave & 1), +1 of its half
    const int sig_beg = c_beg * TAPS, sig_end = c_end * TAPS;
    u32x4 bq[2][NBK][3];
    auto load_b = [&](int sigma, u32x4 (&dst)[NBK][3]) {
        const u32x4* p = wp + (long long)min(sigma, sig_end - 1) * TAP_U4;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
            for (int l = 0; l < 3; ++l) dst[nb][l] = p[(nb * 3 + l) * 64];
    };

    f32x4v acc[MBK][NBK];
#pragma unroll
    for (int i = 0; i < MBK; ++i)
#pragma unroll
        for (int j = 0; j < NBK; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

    load_halo(c_beg);
    load_b(sig_beg, bq[0]);
    store_halo();
    __syncthreads();

    int c = c_beg, tap = 0, tap_off = 0, kx = 0;     // tap_off: pixel-row offset of the current tap in the image
    constexpr int PREFETCH_TAP = TAPS >= 2 ? TAPS - 2 : 0;     // where the next stage's global loads are issued
    // one K step; pp (compile time) = which B fragment buffer it reads, the other one receives the next step's
    auto step = [&](int sigma, auto PP) {
        constexpr int pp = decltype(PP)::value;
        const bool more = (c + 1) < c_end;
        if (tap == PREFETCH_TAP && more) load_halo(c + 1);
        load_b(sigma + 1, bq[pp ^ 1]);
        u32x4 fa[MBK][3];
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb) {
            const int prow = abase[mb] + tap_off;
            const unsigned char* q = smem + prow * ROWB + ((kq ^ lds_swz(prow)) << 4);
#pragma unroll
            for (int l = 0; l < 3; ++l) fa[mb][l] = *reinterpret_cast<const u32x4*>(q + l * LIMB);
        }
        // limb products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(       // weights first: D^T (see dconv_epilogue_n32)
                        __builtin_bit_cast(bf16x8, bq[pp][nb][PB[t]]), __builtin_bit_cast(bf16x8, fa[mb][PA[t]]),
                        acc[mb][nb], 0, 0, 0);
        if (++tap == TAPS) {            // stage done: swap in the next image
            tap = 0; tap_off = 0; kx = 0;
            __syncthreads();
            if (more) {
                store_halo();
                __syncthreads();
            }
            ++c;
        } else if constexpr (PW) {
            tap_off += 128;
        } else {
            if (++kx == 3) { kx = 0; tap_off += W2 - 2; } else { tap_off += 1; }
        }
    };
    for (int sigma = sig_beg; sigma < sig_end; sigma += 2) {
        step(sigma, std::integral_constant<int, 0>{});
        if (sigma + 1 < sig_end) step(sigma + 1, std::integral_constant<int, 1>{});
    }

    dconv_epilogue_n32<MBK>(a, acc, m0, n0 + wave * 32, lane, split);
}

// ---- forward / data-gradient on pre-split activations ("limb planes") ----------------------------------------------
// The producers of a 3x3 convolution's input (GroupNorm+SiLU apply, GroupNorm backward) can write the activation
// already decomposed: bf16 limb planes [pixel][C/32 chunks][3 limbs][32 channels] (6 bytes per element instead of 4).
// A chunk of a pixel is then 3 x 64 contiguous bytes in exactly the form the LDS image wants, so the halo tile is
// staged by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no split3, no ds_write) into one of TWO images:
// the next chunk's image fills while the current chunk's nine taps run, one barrier per chunk.  The XOR slot
// swizzle of the image is applied on the SOURCE address (the DMA destination is lane-linear).
// RG = 16-row groups per image (>= halo pixels / 16); wave w moves row groups w, w + 4, ...
constexpr int LP_PIX_BYTES_PER_CH = 6;     // bytes per element of a limb-plane tensor

template <int RG, bool DB, int MT = 128>
__global__ void __launch_bounds__(256, 2) dconv_lp_kernel(const DConvArgs a) {
    static_assert(MT == 128 || MT == 64, "128- or 64-row tiles");
    // accumulator blocks per wave: 128 x 32 (1 x 4 waves) or, on 64-row tiles, 64 x 32
    constexpr int MBK = MT / 16, NBK = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LIMB = RG * 16 * ROWB;
    constexpr int BUF = 3 * LIMB;
    constexpr int TAPS = 9;
    constexpr int NRG = (RG + 3) / 4;            // row groups per wave (upper bound)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: uniform branches below
    const int wc = wave >> 1;

    const int tiles_n = a.N >> 7;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * MT, n0 = tile_n * 128;
    const int split = blockIdx.y;
    const int c_beg = split * a.chunks_per_split;
    const int c_end = min(a.chunks, c_beg + a.chunks_per_split);

    const int W2 = a.pitch;          // >= W + 2 (columns beyond W + 1 are never read)
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(a.zero);
    const unsigned char* p1 = reinterpret_cast<const unsigned char*>(a.x1);
    const unsigned char* p2 = reinterpret_cast<const unsigned char*>(a.x2);

    // items this lane moves: image row rg*16 + (lane >> 2), 16-byte slot lane & 3 (the image's XOR swizzle is applied
    // on the source side).  hpix = source pixel, or -1 for padding: such items read the 16-byte zero page three times.
    // Everything below is arithmetic on integers (no per-lane branch: a branch around a DMA makes hipcc drain vmcnt).
    int hpix[NRG];
    int sslot[NRG];
    {
        const int HW = a.H * a.W;
        const int img0 = m0 / HW;
        const int oy0 = (m0 - img0 * HW) / a.W;
        const int seg_px = (a.rps + 2) * W2;
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            const int px = (wave + 4 * i) * 16 + (lane >> 2);
            const int seg = px / seg_px;
            const int rem = px - seg * seg_px;
            const int hr = rem / W2, hx = rem - hr * W2;
            const int img = img0 + seg, iy = oy0 + hr - 1, ix = hx - 1;
            const bool ok = seg < a.nseg && img < a.B && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            hpix[i] = ok ? (img * a.H + iy) * a.W + ix : -1;
            sslot[i] = ((lane & 3) ^ lds_swz(px)) << 4;
        }
    }
    // row group i of this wave (three DMAs: one per limb) of chunk c into image buf
    auto issue_dma = [&](int c, int buf, int i) {
        const int c0 = c * 32;
        const bool second = c0 >= a.C1;
        const unsigned char* src = second ? p2 : p1;
        const long long ps = (long long)(second ? a.C2 : a.C1) * LP_PIX_BYTES_PER_CH;
        const long long coff = ((second ? c0 - a.C1 : c0) >> 5) * 192;
        const long long zdelta = zp - src;       // scalar: where the zero page sits relative to this source
        const int rg = wave + 4 * i;
        if (rg < RG) {                            // scalar branch
            const bool ok = hpix[i] >= 0;
            const long long off = ok ? hpix[i] * ps + coff + sslot[i] : zdelta;
            const long long lstep = ok ? 64 : 0;
            const unsigned char* g = src + off;
            unsigned char* d = smem + buf * BUF + rg * 1024;
#pragma unroll
            for (int l = 0; l < 3; ++l)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + l * lstep),
                                                 (__attribute__((address_space(3))) void*)(d + l * LIMB), 16, 0, 0);
        }
    };

    const int r16 = lane & 15, kq = lane >> 4;
    int abase[MBK];
#pragma unroll
    for (int mb = 0; mb < MBK; ++mb) {
        const int ml = mb * 16 + r16;
        const int seg = ml / (a.rps * a.W);
        const int rem = ml - seg * (a.rps * a.W);
        const int ry = rem / a.W, ox = rem - ry * a.W;
        abase[mb] = (seg * (a.rps + 2) + ry) * W2 + ox;
    }

    const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
    // fragment buffer: [n tile][64-column half wc][chunk][tap][16-column block nb 4][limb 3][lane]; a 32-column wave takes
    // blocks 2*(wave & 1) and 2*(wave & 1) + 1 of its half
    const u32x4* wp = a.wfrag + ((long long)(tile_n * 2 + wc) * a.chunks * TAPS) * TAP_U4 + lane + (wave & 1) * 2 * 3 * 64;
    const int sig_beg = c_beg * TAPS, sig_end = c_end * TAPS;
    u32x4 bq[2][NBK][3];
    auto load_b = [&](int sigma, u32x4 (&dst)[NBK][3]) {
        const u32x4* p = wp + (long long)min(sigma, sig_end - 1) * TAP_U4;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
            for (int l = 0; l < 3; ++l) dst[nb][l] = p[(nb * 3 + l) * 64];
    };

    f32x4v acc[MBK][NBK];
#pragma unroll
    for (int i = 0; i < MBK; ++i)
#pragma unroll
        for (int j = 0; j < NBK; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 0; i < NRG; ++i) issue_dma(c_beg, 0, i);
    load_b(sig_beg, bq[0]);
    __syncthreads();                         // vmcnt(0) + barrier: image 0 has landed for every wave

    int c = c_beg, tap = 0, tap_off = 0, kx = 0, buf = 0;
    // Next chunk's image: one row group per tap, taps ISSUE_TAP0 .. ISSUE_TAP0 + NRG - 1.  While an LDS-DMA is in flight
    // hipcc waits vmcnt(0) (not a counted vmcnt) for every B fragment, and vmcnt retires in order: a DMA is therefore
    // issued right AFTER an explicit wait for the B fragments this tap needs anyway, and has the tap's 96 MFMAs to land
    // before the next tap's wait reaches it.
    constexpr int ISSUE_TAP0 = 9 - 1 - NRG;
    auto step = [&](int sigma, auto PP) {
        constexpr int pp = decltype(PP)::value;
        const bool more = (c + 1) < c_end;
        // ONE vector-memory wait per tap, here: it retires this tap's B fragments and the row group issued a tap ago.
        // Everything issued below (a row group of the next image, the next tap's B fragments) has this tap's 96 MFMAs
        // to land.  (With a DMA in flight hipcc can only wait vmcnt(0), never a counted vmcnt: any later wait of its
        // own would drain the prefetch it sits behind.)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (DB && more && tap >= ISSUE_TAP0 && tap < ISSUE_TAP0 + NRG) issue_dma(c + 1, buf ^ 1, tap - ISSUE_TAP0);
        load_b(sigma + 1, bq[pp ^ 1]);
        // A fragments by inline-asm ds_read_b128: to hipcc an LDS read it can see may alias the image an in-flight DMA is
        // filling (same array), and it would wait vmcnt(0) in front of every one of them.  Issue order = use order
        // (limb lo, hi, mid: the six products are lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi), counted lgkmcnt waits.
        u32x4 fa[MBK][3];
        unsigned addr[MBK];
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb) {
            const int prow = abase[mb] + tap_off;
            addr[mb] = lds_base + buf * BUF + prow * ROWB + ((kq ^ lds_swz(prow)) << 4);
        }
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[mb][2]) : "v"(addr[mb]), "n"(2 * LIMB) : "memory");
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb)
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[mb][0]) : "v"(addr[mb]) : "memory");
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[mb][1]) : "v"(addr[mb]), "n"(LIMB) : "memory");
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            if (t == 0) { if constexpr (MBK == 8) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); }
            if (t == 1) { if constexpr (MBK == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); }
            if (t == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t <= 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(       // weights first: D^T (see dconv_epilogue_n32)
                        __builtin_bit_cast(bf16x8, bq[pp][nb][PB[t]]), __builtin_bit_cast(bf16x8, fa[mb][PA[t]]),
                        acc[mb][nb], 0, 0, 0);
        }
        if (++tap == TAPS) {
            tap = 0; tap_off = 0; kx = 0;
            if constexpr (DB) {
                // everyone is done with image buf; every wave's row groups of image buf^1 were issued by tap 7 and
                // retired by its own vmcnt(0) at the top of tap 8: a bare barrier (no vmcnt drain: the next tap's B
                // fragments stay in flight across it)
                __builtin_amdgcn_s_barrier();
                buf ^= 1;
            } else {
                __syncthreads();
                if (more) {
#pragma unroll
                    for (int i = 0; i < NRG; ++i) issue_dma(c + 1, 0, i);
                    __syncthreads();
                }
            }
            ++c;
        } else {
            if (++kx == 3) { kx = 0; tap_off += W2 - 2; } else { tap_off += 1; }
        }
    };
    for (int sigma = sig_beg; sigma < sig_end; sigma += 2) {
        step(sigma, std::integral_constant<int, 0>{});
        if (sigma + 1 < sig_end) step(sigma + 1, std::integral_constant<int, 1>{});
    }

    dconv_epilogue_n32<MBK>(a, acc, m0, n0 + wave * 32, lane, split);
}

// fp32 NHWC [rows][c] -> limb planes [rows][c/32][3][32] (tests, and producers without a fused writer)
__global__ void f32_to_limb_kernel(const float* __restrict__ x, long long rows, int c, unsigned char* __restrict__ y) {
    const long long n4 = rows * (c >> 2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / (c >> 2);
        const int q = (int)(i - r * (c >> 2));
        const f32x4 v = ld4(x + r * c + q * 4);
        unsigned h0, m0, l0, h1, m1, l1;
        split3(v[0], v[1], h0, m0, l0);
        split3(v[2], v[3], h1, m1, l1);
        unsigned char* d = y + r * (long long)c * LP_PIX_BYTES_PER_CH + (q >> 3) * 192 + (q & 7) * 8;
        *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(d + 64) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(d + 128) = u32x2{l0, l1};
    }
}
__global__ void limb_to_f32_kernel(const unsigned char* __restrict__ y, long long rows, int c, float* __restrict__ x) {
    const long long n = rows * c;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c;
        const int ch = (int)(i - r * c);
        const unsigned short* d = reinterpret_cast<const unsigned short*>(y + r * (long long)c * LP_PIX_BYTES_PER_CH +
                                                                          (ch >> 5) * 192) + (ch & 31);
        x[i] = (__uint_as_float((unsigned)d[0] << 16) + __uint_as_float((unsigned)d[32] << 16)) +
               __uint_as_float((unsigned)d[64] << 16);
    }
}

// ---- weight gradient ---------------------------------------------------------------------------------------
// dW[co][tap][ci] = sum_p dy[p][co] * x[p + tap][ci].  One workgroup owns a 64 (co) x 64 (ci) tile for the three
// taps of ONE filter row ky (each wave 32 x 32 x 3 taps = twelve 16x16 accumulators, 48 VGPRs, so four waves fit
// per SIMD) over a range of 32-pixel K tiles: per K tile the dy rows and the input row(s) oy + ky - 1 of x, W + 2
// pixels wide, are split into limbs once and kept in LDS as [pixel][channel] images.  Both MFMA operands need
// k (= pixel) along the register, so the fragments come from ds_read_b64_tr_b16 transposed reads; the x fragment of
// tap kx sits kx rows further.  One v_mfma_f32_16x16x32_bf16 consumes the whole K tile; its k slot (group g = lane
// >> 4, j, q) is pixel 4g + 16j + q, so the two 16-lane groups of a half-wave read eight CONSECUTIVE pixel rows,
// which the 160-byte row stride spreads over all 64 banks (conflict-free).
constexpr int WG_RS = 160;           // bytes per pixel row and limb: 64 channels bf16 + 32 pad
constexpr int WG_AROWS = 32, WG_BROWS = 48;
constexpr int WG_BLIMB = WG_BROWS * WG_RS;
constexpr int WG_NB = 3;             // x float4 items per thread (48 rows x 16 quads / 256)

struct DWgradArgs {
    const float* dy;
    int lddy;
    const float* x;
    int cin;                // channels of x (row stride)
    const float* x2;        // second source of a channel concatenation (or null): c_in tiles beyond cin read it
    int cin2;
    int B, H, W;
    int cout_tiles, cin_tiles;
    int ktiles, ktiles_per_split;
    float* slabs;
    int ld_tap;             // cin_total: stride between taps of one output channel
    long long slab_stride;
    int hw_w;               // staged row width = min(W, 32) + 2
    int hrows;              // staged rows per K tile = max(1, 32 / W)
};

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ u32x2 lds_tr16(const unsigned char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) void*)(p));
    return __builtin_bit_cast(u32x2, v);
}

// CB = 16-channel dy blocks per wave: 2 -> 64-co tile, three workgroups per CU; 4 -> 128-co tile, two per CU (every
// staged x value and every B fragment then feeds twice the MFMAs).
// XLP: x (and x2) are bf16 limb planes [pixel][c/32][3][32] (written by GroupNorm's apply pass): the x items are staged
// as plain 16-byte copies (4.5 per thread, one limb each) - no split3 for that operand, which is re-staged by
// cout_tiles * 3 workgroups (the split is 43 % of this kernel's VALU work, and VALU issue is what bounds it).
// ABL: timing-only ablations (wrong results; PSLD_DWGRAD_ABL, tools/bench_limb.py --wgrad): 1 = no limb split (raw halves are
// stored), 2 = one ds_read_b128 per fragment instead of two transposed reads, 4 = no staging at all (no split, no LDS
// stores, and with them the global loads).  Round 3 (profiles/r03/ab_dwgrad.txt, 256->256 @32 B=128): 213 TFLOP/s as
// shipped, 231 without the split, 223 with half the LDS read instructions, 233 with both, 275 with no staging at all.  An
// eight-wave form built on that reading - 128 x 128 x 3-tap tiles (29 % fewer staged bytes per MFMA), two LDS images, one
// barrier per K tile, SIMD partners staging and multiplying in opposite order as in wino_conv8s_kernel, the tile after
// next held in registers - was bitwise this kernel and exactly as fast (222 / 230 / 232 / 238 vs 222 / 232 / 234 / 234
// TFLOP/s on the four 32x32 / 16x16 shapes): per K tile a wave issues ~350 vector / LDS / memory instructions (staging
// ~200 of them, tile addressing the rest) beside 144 MFMAs, and 2 waves x (350 x 4 + 144 x 8) issue cycles exceed the
// 4608 cycles the SIMD's matrix pipe needs - the kernel is ISSUE-bound at ~0.65 pipe occupancy in both forms.  What
// would move it is operands that arrive pre-split (no producer can write them for less than the split costs here: +6 B per
// element on kernels that run at the HBM roofline).  Removed again.
template <int CB, bool XLP, int ABL = 0>
__global__ void __launch_bounds__(256, CB == 2 ? 3 : 2) dwgrad_kernel(const DWgradArgs a) {
    constexpr int CO_T = 32 * CB;                // output channels per workgroup
    constexpr int RSA = CO_T * 2 + 32;           // dy row stride: 160 / 288 B, both conflict-free for the transposed reads
    constexpr int ALIMB = WG_AROWS * RSA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                    // [3][32][RSA]
    unsigned char* Bs = smem + 3 * ALIMB;        // [3][48][160]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // blocks of one K range read the same pixels: keep them on one XCD (shared L2)
    const int tiles = a.cout_tiles * a.cin_tiles;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = vid / (tiles * 3);
    const int rest = vid - split * tiles * 3;
    const int ky = rest / tiles, tile = rest - ky * tiles;
    const int co0 = (tile / a.cin_tiles) * CO_T;
    const int ci_out = (tile % a.cin_tiles) * 64;                // column of the slab
    const bool second = ci_out >= a.cin;
    const float* xsrc = second ? a.x2 : a.x;
    const int xc = second ? a.cin2 : a.cin;                      // row stride of the source
    const int ci0 = second ? ci_out - a.cin : ci_out;            // channel within the source
    const int kt_beg = split * a.ktiles_per_split;
    const int kt_end = min(a.ktiles, kt_beg + a.ktiles_per_split);
    const int HW = a.H * a.W;

    // transposed-read lane roles: a group of 16 lanes reads 4 pixel rows x 16 channels; lane 4q + p supplies the
    // address of row q, channels 4p..4p+3 and receives channel (lane & 15) of the four rows
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
    int a_base[2], b_base[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = 4 * g + 16 * j + q;              // pixel of this lane's row in read j
        a_base[j] = k * RSA + (wr * 16 * CB + 4 * p4) * 2;
        const int ry = a.W >= 32 ? 0 : k / a.W;
        const int ox = a.W >= 32 ? k : k - ry * a.W;
        b_base[j] = (ry * a.hw_w + ox) * WG_RS + (wc * 32 + 4 * p4) * 2;
    }

    const int qa = tid & 15, ra = tid >> 4;     // staging: channel quad, pixel row (+16 per item)
    // Staged x items: everything that does not change from K tile to K tile is folded into a byte offset and a few
    // validity bits per item; per tile only a scalar base pointer and a scalar row mask move.  Loads go through a
    // raw buffer resource, whose range check returns zeros for the offset 0xFFFFFFFF given to padding items
    // (no zero-page select, ~4 VALU per item instead of ~15).  Measured: neutral - and so was a double-buffered LDS
    // variant with one barrier per K tile at two workgroups per CU (181-188 vs 200-209 TFLOP/s): the kernel runs
    // best as three single-buffered workgroups per CU overlapping each other's staging phases.  s_setprio(1) around
    // the MFMA phase: neutral.  Upper bound of removing the limb split altogether (operands pre-split in HBM): +9 %.
    constexpr int NXI = XLP ? 5 : WG_NB;        // limb planes: 48 rows x 3 limbs x 8 sixteen-byte slots = 1152 items
    unsigned xoff[NXI];     // byte offset from pixel (img, oy0 + ky - 1, ox0 - 1), channel ci0
    int xmeta[NXI];         // bits 0-4: staged row of the item (31: never valid); bit 5: column valid when the K tile
                            // starts at ox0 = 0, bit 6: at ox0 = 32 (W = 64)
    int xdst[NXI];          // limb planes: LDS byte offset of the item inside Bs
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        if constexpr (XLP) {
            const int id = tid + 256 * i;                    // item: limb l, staged row px, 16-byte slot t (8 channels)
            const int l = id / 384, rem = id - l * 384;
            const int px = rem >> 3, t = rem & 7;
            const int hr = px / a.hw_w, hc = px - hr * a.hw_w;
            xoff[i] = (unsigned)((hr * a.W + hc) * xc * 6 + ((t >> 2) * 3 + l) * 64 + (t & 3) * 16);
            xmeta[i] = ((id < 1152 && hr < a.hrows) ? hr : 31) | ((hc >= 1 && hc <= a.W) ? 32 : 0) | ((hc + 31 < a.W) ? 64 : 0);
            xdst[i] = l * WG_BLIMB + px * WG_RS + t * 16;
        } else {
            const int px = ra + 16 * i;
            const int hr = px / a.hw_w, hc = px - hr * a.hw_w;
            xoff[i] = (unsigned)(((hr * a.W + hc) * xc + qa * 4) * 4);
            xmeta[i] = (hr < a.hrows ? hr : 31) | ((hc >= 1 && hc <= a.W) ? 32 : 0) | ((hc + 31 < a.W) ? 64 : 0);
            xdst[i] = 0;
        }
    }
    unsigned aoff[CB];      // dy items: pixel ra + 16*(i & 1), channel quad qa + 16*(i >> 1)
#pragma unroll
    for (int i = 0; i < CB; ++i) aoff[i] = (unsigned)(((ra + 16 * (i & 1)) * a.lddy + (qa + 16 * (i >> 1)) * 4) * 4);
    f32x4 va[CB], vb[NXI];
    auto load_tile = [&](int kt) {
        const int p0 = kt * 32;
        const int img = p0 / HW;
        const int rem = p0 - img * HW;
        const int oy0 = rem / a.W, ox0 = rem - oy0 * a.W;
        const int iy0 = oy0 + ky - 1;
        unsigned rowmask = 0;                       // bit r: staged row r lies inside the image
        for (int rr = 0; rr < a.hrows; ++rr) rowmask |= (iy0 + rr >= 0 && iy0 + rr < a.H) ? 1u << rr : 0u;
        const int colsel = 5 + (ox0 >> 5);
        // limb planes: 6 bytes per element, the tile's first 32-channel chunk at (ci0 / 32) * 192 bytes into the pixel
        const unsigned char* xb8 = reinterpret_cast<const unsigned char*>(xsrc) +
                                   (XLP ? ((long long)((img * a.H + iy0) * a.W + ox0 - 1) * xc * 6 + (ci0 >> 5) * 192)
                                        : ((long long)((img * a.H + iy0) * a.W + ox0 - 1) * xc + ci0) * 4);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(xb8), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.dy + ((long long)p0 * a.lddy + co0)), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < CB; ++i)
            va[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, aoff[i], 0, 0));
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const bool ok = ((rowmask >> (xmeta[i] & 31)) & (unsigned)(xmeta[i] >> colsel) & 1u) != 0;
            vb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xoff[i] : 0xffffffffu, 0, 0));
        }
    };
    auto store_rows = [&](unsigned char* d, int limb_stride, const f32x4& v) {
        if constexpr ((ABL & 4) != 0) return;
        unsigned h0, m0, l0, h1, m1, l1;
        if constexpr ((ABL & 1) != 0) {
            h0 = m0 = l0 = __float_as_uint(v[0]) ^ __float_as_uint(v[1]);
            h1 = m1 = l1 = __float_as_uint(v[2]) ^ __float_as_uint(v[3]);
        } else {
            split3(v[0], v[1], h0, m0, l0);
            split3(v[2], v[3], h1, m1, l1);
        }
        *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(d + limb_stride) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(d + 2 * limb_stride) = u32x2{l0, l1};
    };
    auto frag = [&](const unsigned char* img, const int (&base)[2], int off) -> u32x4 {
        if constexpr ((ABL & 2) != 0) return *reinterpret_cast<const u32x4*>(img + ((base[0] + off) & ~15));
        const u32x2 lo = lds_tr16(img + base[0] + off), hi = lds_tr16(img + base[1] + off);
        return u32x4{lo[0], lo[1], hi[0], hi[1]};
    };

    f32x4v acc[3][CB][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

    if (kt_beg < kt_end) load_tile(kt_beg);
    for (int kt = kt_beg; kt < kt_end; ++kt) {
#pragma unroll
        for (int i = 0; i < CB; ++i) store_rows(As + (ra + 16 * (i & 1)) * RSA + (qa + 16 * (i >> 1)) * 8, ALIMB, va[i]);
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            if constexpr (XLP) {
                if (tid + 256 * i < 1152) *reinterpret_cast<f32x4*>(Bs + xdst[i]) = vb[i];
            } else {
                store_rows(Bs + (ra + 16 * i) * WG_RS + qa * 8, WG_BLIMB, vb[i]);
            }
        }
        __syncthreads();
        if (kt + 1 < kt_end) load_tile(kt + 1);
        u32x4 fa[CB][3];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int l = 0; l < 3; ++l) fa[cb][l] = frag(As, a_base, l * ALIMB + cb * 32);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
            u32x4 fb[2][3];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int l = 0; l < 3; ++l) fb[nb][l] = frag(Bs, b_base, l * WG_BLIMB + nb * 32 + tx * WG_RS);
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[tx][cb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, fa[cb][PA[u]]), __builtin_bit_cast(bf16x8, fb[nb][PB[u]]),
                            acc[tx][cb][nb], 0, 0, 0);
        }
        __syncthreads();
    }

    // C/D layout of the 16x16 MFMA: col (ci) = lane & 15, row (co) = 4*(lane >> 4) + v
    float* S = a.slabs + (long long)split * a.slab_stride;
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int co = co0 + wr * 16 * CB + cb * 16 + 4 * g + v;
                    S[((long long)co * 9 + ky * 3 + tx) * a.ld_tap + ci_out + wc * 32 + nb * 16 + i16] = acc[tx][cb][nb][v];
                }
}

// ---- weight gradient, wave-specialised ---------------------------------------------------------------
// The same 128 (c_out) x 64 (c_in) x 3-tap tile and the same staged images as dwgrad_kernel<4, false>, but in a 512-thread
// workgroup whose waves 4-7 do nothing but load, split3 and ds_write the NEXT K tile's limb images while waves 0-3 issue
// only transposed LDS reads and MFMAs on the current one: two images, ONE barrier per K tile, the raw rows of the tile after
// next in the producers' registers.  Per SIMD one consumer and one producer wave instead of two waves that each alternate
// between the two kinds of work.  Same products in the same order per output element: bitwise dwgrad_kernel.
// Measured (profiles/r04/ab_dwgrad_specialised.txt, B=128, same box, alternating): 256->256 @32 226 -> 239, 512->256 @32
// 228 -> 234, 256->256 @16 230 -> 240, 512->256 @16 231 -> 246 TFLOP/s (+3-7 %), 8x8 unchanged; faster than the limb-plane-x
// form of the old kernel.  A second register set in the producers (three tiles ahead) and s_setprio 1 for the consumers:
// no further change - a consumer wave needs ~4,000 cycles per K tile where its 144 MFMAs take 2,304: what is exposed is
// the LDS latency of its 60 transposed fragment reads behind each barrier, not the producers.  Reading the next tile's
// first fragments before the barrier (three images, producers two tiles ahead) was built too: with 6 of the 30 fragments
// 219 TFLOP/s (248 VGPRs; the fenced schedule costs more than the latency it hides), with 12 it spills (134).
__global__ void __launch_bounds__(512) dwgrad_ws_kernel(const DWgradArgs a) {
    constexpr int CB = 4;
    constexpr int CO_T = 32 * CB;
    constexpr int RSA = CO_T * 2 + 32;           // 288
    constexpr int ALIMB = WG_AROWS * RSA;
    constexpr int IMG = 3 * (ALIMB + WG_BLIMB);  // one image pair (dy | x)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int tiles = a.cout_tiles * a.cin_tiles;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = vid / (tiles * 3);
    const int rest = vid - split * tiles * 3;
    const int ky = rest / tiles, tile = rest - ky * tiles;
    const int co0 = (tile / a.cin_tiles) * CO_T;
    const int ci_out = (tile % a.cin_tiles) * 64;
    const int kt_beg = split * a.ktiles_per_split;
    const int kt_end = min(a.ktiles, kt_beg + a.ktiles_per_split);
    const int nkt = kt_end - kt_beg;

    // The two roles are two disjoint code paths with the same number of barriers (one before the loop, one per K tile):
    // their registers do not add up.
    if (wave >= 4) {
        // ---- producers: thread pt stages what thread pt of dwgrad_kernel stages ---------------------------------------------
        const bool second = ci_out >= a.cin;
        const float* xsrc = second ? a.x2 : a.x;
        const int xc = second ? a.cin2 : a.cin;
        const int ci0 = second ? ci_out - a.cin : ci_out;
        const int HW = a.H * a.W;
        const int pt = tid & 255;
        const int qa = pt & 15, ra = pt >> 4;
        unsigned xoff[WG_NB];
        int xmeta[WG_NB];
#pragma unroll
        for (int i = 0; i < WG_NB; ++i) {
            const int px = ra + 16 * i;
            const int hr = px / a.hw_w, hc = px - hr * a.hw_w;
            xoff[i] = (unsigned)(((hr * a.W + hc) * xc + qa * 4) * 4);
            xmeta[i] = (hr < a.hrows ? hr : 31) | ((hc >= 1 && hc <= a.W) ? 32 : 0) | ((hc + 31 < a.W) ? 64 : 0);
        }
        unsigned aoff[CB];
#pragma unroll
        for (int i = 0; i < CB; ++i) aoff[i] = (unsigned)(((ra + 16 * (i & 1)) * a.lddy + (qa + 16 * (i >> 1)) * 4) * 4);
        f32x4 va[CB], vb[WG_NB];
        auto load_tile = [&](int kt) {
            const int p0 = kt * 32;
            const int img = p0 / HW;
            const int rem = p0 - img * HW;
            const int oy0 = rem / a.W, ox0 = rem - oy0 * a.W;
            const int iy0 = oy0 + ky - 1;
            unsigned rowmask = 0;
            for (int rr = 0; rr < a.hrows; ++rr) rowmask |= (iy0 + rr >= 0 && iy0 + rr < a.H) ? 1u << rr : 0u;
            const int colsel = 5 + (ox0 >> 5);
            const unsigned char* xb8 = reinterpret_cast<const unsigned char*>(xsrc) + ((long long)((img * a.H + iy0) * a.W + ox0 - 1) * xc + ci0) * 4;
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(xb8), 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.dy + ((long long)p0 * a.lddy + co0)), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int i = 0; i < CB; ++i) va[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, aoff[i], 0, 0));
#pragma unroll
            for (int i = 0; i < WG_NB; ++i) {
                const bool ok = ((rowmask >> (xmeta[i] & 31)) & (unsigned)(xmeta[i] >> colsel) & 1u) != 0;
                vb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xoff[i] : 0xffffffffu, 0, 0));
            }
        };
        auto store_rows = [&](unsigned char* d, int limb_stride, const f32x4& v) {
            unsigned h0, m0, l0, h1, m1, l1;
            split3(v[0], v[1], h0, m0, l0);
            split3(v[2], v[3], h1, m1, l1);
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + limb_stride) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(d + 2 * limb_stride) = u32x2{l0, l1};
        };
        auto store_tile = [&](int buf) {
            unsigned char* As = smem + buf * IMG;
            unsigned char* Bs = As + 3 * ALIMB;
#pragma unroll
            for (int i = 0; i < CB; ++i) store_rows(As + (ra + 16 * (i & 1)) * RSA + (qa + 16 * (i >> 1)) * 8, ALIMB, va[i]);
#pragma unroll
            for (int i = 0; i < WG_NB; ++i) store_rows(Bs + (ra + 16 * i) * WG_RS + qa * 8, WG_BLIMB, vb[i]);
        };
        // the raw rows of tile i + 2 are in flight while tile i + 1 is split and stored (a second register set, i.e. three
        // tiles ahead, measured no different: the consumers, not the producers, set the pace)
        if (nkt > 0) {
            load_tile(kt_beg);
            store_tile(0);
            if (nkt > 1) load_tile(kt_beg + 1);
        }
        __syncthreads();
        for (int i = 0; i < nkt; ++i) {
            if (i + 1 < nkt) store_tile((i + 1) & 1);       // everyone left that image at the last barrier
            if (i + 2 < nkt) load_tile(kt_beg + i + 2);
            __syncthreads();
        }
        return;
    }

    // ---- consumers: dwgrad_kernel's wave tile (64 c_out x 32 c_in x 3 taps) and lane roles -----------------------------------
    const int wr = wave >> 1, wc = wave & 1;
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
    int a_base[2], b_base[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = 4 * g + 16 * j + q;
        a_base[j] = k * RSA + (wr * 16 * CB + 4 * p4) * 2;
        const int ry = a.W >= 32 ? 0 : k / a.W;
        const int ox = a.W >= 32 ? k : k - ry * a.W;
        b_base[j] = (ry * a.hw_w + ox) * WG_RS + (wc * 32 + 4 * p4) * 2;
    }
    auto frag = [&](const unsigned char* img, const int (&base)[2], int off) -> u32x4 {
        const u32x2 lo = lds_tr16(img + base[0] + off), hi = lds_tr16(img + base[1] + off);
        return u32x4{lo[0], lo[1], hi[0], hi[1]};
    };
    f32x4v acc[3][CB][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int i = 0; i < nkt; ++i) {
        const unsigned char* As = smem + (i & 1) * IMG;
        const unsigned char* Bs = As + 3 * ALIMB;
        u32x4 fa[CB][3];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int l = 0; l < 3; ++l) fa[cb][l] = frag(As, a_base, l * ALIMB + cb * 32);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
            u32x4 fb[2][3];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int l = 0; l < 3; ++l) fb[nb][l] = frag(Bs, b_base, l * WG_BLIMB + nb * 32 + tx * WG_RS);
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[tx][cb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, fa[cb][PA[u]]), __builtin_bit_cast(bf16x8, fb[nb][PB[u]]),
                            acc[tx][cb][nb], 0, 0, 0);
        }
        __syncthreads();
    }

    float* S = a.slabs + (long long)split * a.slab_stride;
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int co = co0 + wr * 16 * CB + cb * 16 + 4 * g + v;
                    S[((long long)co * 9 + ky * 3 + tx) * a.ld_tap + ci_out + wc * 32 + nb * 16 + i16] = acc[tx][cb][nb][v];
                }
}

int launch_dwgrad_ws(const DWgradArgs& a, int nsplit, hipStream_t stream) {
    constexpr size_t LDS = (size_t)2 * 3 * (WG_AROWS * (64 * 4 + 32) + WG_BLIMB);
    static PsldPerDeviceFlag configured_; bool& configured = configured_.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dwgrad_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) {
            psld_set_error("psld_conv3x3_wgrad_split_f32: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return PSLD_ERR_LAUNCH;
        }
        configured = true;
    }
    hipLaunchKernelGGL(dwgrad_ws_kernel, dim3((unsigned)(a.cout_tiles * a.cin_tiles * 3 * nsplit)), dim3(512), LDS, stream, a);
    PSLD_CHECK_LAUNCH("psld_conv3x3_wgrad_split_f32");
    return PSLD_OK;
}

// ---- pointwise weight gradient (TN GEMM) ---------------------------------------------------------------------
// C[i][j] = sum_p A[p][i] * B[p][j]: the weight gradient of a 1x1 convolution / NIN projection (A = dy, B = x or the
// other way round for NIN's [in][out] weights).  Same machinery as dwgrad_kernel without the halo: a workgroup owns
// a 128 x 128 tile of C, stages 32 rows of A and B per K tile as limb images (288-byte rows) and reads both through
// ds_read_b64_tr_b16; wave tile 64 x 64, two workgroups per CU.
struct PWgradArgs {
    const float* a;
    int lda;
    const float* b;
    int ldb;
    const float* b2;        // second source of a column concatenation of B (or null): columns >= n1 read it
    int ldb2, n1;
    int tiles_i, tiles_j;
    int ktiles, ktiles_per_split;
    float* slabs;
    int ldc;
    long long slab_stride;
};

__global__ void __launch_bounds__(256, 2) pwgrad_kernel(const PWgradArgs a) {
    constexpr int RS = 288;                      // 128 channels bf16 + 32 pad
    constexpr int LIMB = 32 * RS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                    // [3][32][288]
    unsigned char* Bs = smem + 3 * LIMB;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles = a.tiles_i * a.tiles_j;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = vid / tiles, tile = vid - split * tiles;
    const int i0 = (tile / a.tiles_j) * 128, j0 = (tile % a.tiles_j) * 128;
    const bool second = j0 >= a.n1;
    const float* bsrc = second ? a.b2 : a.b;
    const int ldb = second ? a.ldb2 : a.ldb;
    const int jb = second ? j0 - a.n1 : j0;
    const int kt_beg = split * a.ktiles_per_split;
    const int kt_end = min(a.ktiles, kt_beg + a.ktiles_per_split);

    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
    int a_base[2], b_base[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = 4 * g + 16 * j + q;
        a_base[j] = k * RS + (wr * 64 + 4 * p4) * 2;
        b_base[j] = k * RS + (wc * 64 + 4 * p4) * 2;
    }
    const int qa = tid & 15, ra = tid >> 4;     // staging item i: row ra + 16*(i & 1), channel quad qa + 16*(i >> 1)
    unsigned aoff[4], boff[4];
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ra + 16 * (i & 1), c = qa + 16 * (i >> 1);
        aoff[i] = (unsigned)((r * a.lda + c * 4) * 4);
        boff[i] = (unsigned)((r * ldb + c * 4) * 4);
        soff[i] = r * RS + c * 8;
    }
    f32x4 va[4], vb[4];
    auto load_tile = [&](int kt) {
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.a + ((long long)kt * 32 * a.lda + i0)), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(bsrc + ((long long)kt * 32 * ldb + jb)), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) va[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, aoff[i], 0, 0));
#pragma unroll
        for (int i = 0; i < 4; ++i) vb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, boff[i], 0, 0));
    };
    auto store_rows = [&](unsigned char* d, const f32x4& v) {
        unsigned h0, m0, l0, h1, m1, l1;
        split3(v[0], v[1], h0, m0, l0);
        split3(v[2], v[3], h1, m1, l1);
        *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(d + LIMB) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(d + 2 * LIMB) = u32x2{l0, l1};
    };
    auto frag = [&](const unsigned char* img, const int (&base)[2], int off) -> u32x4 {
        const u32x2 lo = lds_tr16(img + base[0] + off), hi = lds_tr16(img + base[1] + off);
        return u32x4{lo[0], lo[1], hi[0], hi[1]};
    };

    f32x4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

    if (kt_beg < kt_end) load_tile(kt_beg);
    for (int kt = kt_beg; kt < kt_end; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) store_rows(As + soff[i], va[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) store_rows(Bs + soff[i], vb[i]);
        __syncthreads();
        if (kt + 1 < kt_end) load_tile(kt + 1);
        u32x4 fa[4][3];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int l = 0; l < 3; ++l) fa[cb][l] = frag(As, a_base, l * LIMB + cb * 32);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            u32x4 fb[3];
#pragma unroll
            for (int l = 0; l < 3; ++l) fb[l] = frag(Bs, b_base, l * LIMB + nb * 32);
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[cb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fa[cb][PA[u]]), __builtin_bit_cast(bf16x8, fb[PB[u]]), acc[cb][nb], 0, 0, 0);
        }
        __syncthreads();
    }

    float* S = a.slabs + (long long)split * a.slab_stride;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = i0 + wr * 64 + cb * 16 + 4 * g + v;
                S[(long long)i * a.ldc + j0 + wc * 64 + nb * 16 + i16] = acc[cb][nb][v];
            }
}

// ---- batched GEMM with both operands split in the kernel (attention QK^T, PV and their gradients) -------------------
// C[b] = alpha * op(A[b]) op(B[b]), 128 x 128 tiles, K tiles of 32.  An operand whose K index is the row index in
// memory (AT: A stored [k][i]; !BT: B stored [k][j]) is staged like pwgrad_kernel's images and read transposed; an
// operand with contiguous K (A stored [i][k]; BT: B stored [j][k]) is staged like dconv_kernel's pixel rows (64-byte
// rows, slot swizzle) and read with ds_read_b128.  The transposed reads deliver K in the order 4g + 16j + q, so the
// row-major images store their 4-element K groups in that order too (group c at slot c & 3, half c >> 2).
struct BGemmArgs {
    const float* a;
    int lda;
    long long sa;
    const float* b;
    int ldb;
    long long sb;
    float* c;
    int ldc;
    long long sc;
    int tiles_i, tiles_j, ktiles;
    float alpha;
};

template <bool AT, bool BT>
__global__ void __launch_bounds__(256, 2) bgemm_kernel(const BGemmArgs a) {
    constexpr int MC_RS = 288, MC_LIMB = 32 * MC_RS, KC_LIMB = 128 * ROWB;
    constexpr int A_LIMB = AT ? MC_LIMB : KC_LIMB, B_LIMB = BT ? KC_LIMB : MC_LIMB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Bs = smem + 3 * A_LIMB;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles = a.tiles_i * a.tiles_j;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int batch = vid / tiles, tile = vid - batch * tiles;
    const int i0 = (tile / a.tiles_j) * 128, j0 = (tile % a.tiles_j) * 128;
    const float* Ab = a.a + batch * a.sa + (AT ? (long long)i0 : (long long)i0 * a.lda);
    const float* Bb = a.b + batch * a.sb + (BT ? (long long)j0 * a.ldb : (long long)j0);

    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
    // fragment read addresses
    int a_rd[2], b_rd[2];
    if constexpr (AT) {
#pragma unroll
        for (int j = 0; j < 2; ++j) a_rd[j] = (4 * g + 16 * j + q) * MC_RS + (wr * 64 + 4 * p4) * 2;
    } else {
        a_rd[0] = (wr * 64 + i16) * ROWB + ((g ^ lds_swz(i16)) << 4);
        a_rd[1] = 0;
    }
    if constexpr (!BT) {
#pragma unroll
        for (int j = 0; j < 2; ++j) b_rd[j] = (4 * g + 16 * j + q) * MC_RS + (wc * 64 + 4 * p4) * 2;
    } else {
        b_rd[0] = (wc * 64 + i16) * ROWB + ((g ^ lds_swz(i16)) << 4);
        b_rd[1] = 0;
    }
    // staging items: global byte offset within a K tile and LDS byte offset
    unsigned a_go[4], b_go[4];
    int a_so[4], b_so[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 4) + 16 * (i & 1), cq = (tid & 15) + 16 * (i >> 1);      // [k row][channel quad] images
        const int row = (tid >> 3) + 32 * i, c4 = tid & 7;                            // [row][k quad] images
        const int kc_so = row * ROWB + ((((c4 & 3) ^ lds_swz(row))) << 4) + (c4 >> 2) * 8;
        a_go[i] = AT ? (unsigned)((r * a.lda + cq * 4) * 4) : (unsigned)((row * a.lda + c4 * 4) * 4);
        a_so[i] = AT ? r * MC_RS + cq * 8 : kc_so;
        b_go[i] = !BT ? (unsigned)((r * a.ldb + cq * 4) * 4) : (unsigned)((row * a.ldb + c4 * 4) * 4);
        b_so[i] = !BT ? r * MC_RS + cq * 8 : kc_so;
    }
    f32x4 va[4], vb[4];
    auto load_tile = [&](int kt) {
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(Ab + (AT ? (long long)kt * 32 * a.lda : (long long)kt * 32)), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(Bb + (!BT ? (long long)kt * 32 * a.ldb : (long long)kt * 32)), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) va[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, a_go[i], 0, 0));
#pragma unroll
        for (int i = 0; i < 4; ++i) vb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, b_go[i], 0, 0));
    };
    auto store_item = [&](unsigned char* d, int limb, const f32x4& v) {
        unsigned h0, m0, l0, h1, m1, l1;
        split3(v[0], v[1], h0, m0, l0);
        split3(v[2], v[3], h1, m1, l1);
        *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(d + limb) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(d + 2 * limb) = u32x2{l0, l1};
    };
    auto frag_a = [&](int l, int blk) -> u32x4 {
        if constexpr (AT) {
            const u32x2 lo = lds_tr16(As + a_rd[0] + l * A_LIMB + blk * 32), hi = lds_tr16(As + a_rd[1] + l * A_LIMB + blk * 32);
            return u32x4{lo[0], lo[1], hi[0], hi[1]};
        } else {
            return *reinterpret_cast<const u32x4*>(As + a_rd[0] + l * A_LIMB + blk * 16 * ROWB);
        }
    };
    auto frag_b = [&](int l, int blk) -> u32x4 {
        if constexpr (!BT) {
            const u32x2 lo = lds_tr16(Bs + b_rd[0] + l * B_LIMB + blk * 32), hi = lds_tr16(Bs + b_rd[1] + l * B_LIMB + blk * 32);
            return u32x4{lo[0], lo[1], hi[0], hi[1]};
        } else {
            return *reinterpret_cast<const u32x4*>(Bs + b_rd[0] + l * B_LIMB + blk * 16 * ROWB);
        }
    };

    f32x4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

    load_tile(0);
    for (int kt = 0; kt < a.ktiles; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) store_item(As + a_so[i], A_LIMB, va[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) store_item(Bs + b_so[i], B_LIMB, vb[i]);
        __syncthreads();
        if (kt + 1 < a.ktiles) load_tile(kt + 1);
        u32x4 fa[4][3];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int l = 0; l < 3; ++l) fa[cb][l] = frag_a(l, cb);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            u32x4 fb[3];
#pragma unroll
            for (int l = 0; l < 3; ++l) fb[l] = frag_b(l, nb);
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[cb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fa[cb][PA[u]]), __builtin_bit_cast(bf16x8, fb[PB[u]]), acc[cb][nb], 0, 0, 0);
        }
        __syncthreads();
    }

    float* Cb = a.c + batch * a.sc;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = i0 + wr * 64 + cb * 16 + 4 * g + v;
                Cb[(long long)i * a.ldc + j0 + wc * 64 + nb * 16 + i16] = a.alpha * acc[cb][nb][v];
            }
}

template <bool AT, bool BT>
int launch_bgemm(const BGemmArgs& a, int batch, hipStream_t stream) {
    constexpr size_t LDS = (size_t)3 * ((AT ? 32 * 288 : 128 * ROWB) + (BT ? 128 * ROWB : 32 * 288));
    static PsldPerDeviceFlag configured_; bool& configured = configured_.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bgemm_kernel<AT, BT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) {
            psld_set_error("psld_bgemm_split_f32: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return PSLD_ERR_LAUNCH;
        }
        configured = true;
    }
    hipLaunchKernelGGL((bgemm_kernel<AT, BT>), dim3((unsigned)(a.tiles_i * a.tiles_j * batch)), dim3(256), LDS, stream, a);
    PSLD_CHECK_LAUNCH("psld_bgemm_split_f32");
    return PSLD_OK;
}

// ---- pointwise, eight waves ------------------------------------------------------------------------------------------
// The pointwise GEMMs (ResBlock shortcuts 512 -> 256, attention projections) ran at 113-161 TFLOP/s on dconv_kernel<8, 2,
// PW>: a 1x1 convolution has one K step per staged chunk where the 3x3 form has nine, so per MFMA it stages (loads, splits,
// stores to LDS) nine times the activations, each 128-channel tile re-stages its rows, and its single LDS image puts every
// stage's split3 + ds_write between two barriers.  Here one workgroup of EIGHT waves owns 128 rows x 256 channels (wave w:
// all 128 rows x channels 32 w .. 32 w + 31, the 8 x 2 accumulator blocks and epilogue of the N32 layout): the rows are
// staged once for twice the channels, by twice the threads (4 float4 per thread and 64-channel stage instead of 8), into
// TWO images - stage s + 1 is split and stored while stage s multiplies, one barrier per stage - and the raw operands of
// stage s + 2 are already in flight.  Weight fragments: the existing layout (per 64-channel half tile), wave w reads the
// two 16-channel blocks of its 32 channels.
constexpr int PW8_ROWS = 256;                       // LDS pixel rows per image: 128 rows x 2 chunks
constexpr int PW8_LIMB = PW8_ROWS * ROWB;           // bytes per limb of one image
constexpr int PW8_IMG = 3 * PW8_LIMB;
template <int ABL = 0>      // timing-only ablations (PSLD_PW8_ABL): 1 no staging after the prologue, 2 weights loaded once, 4 no epilogue
__global__ void __launch_bounds__(512) pw8_kernel(const DConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 7;
    const int tiles_n = a.N >> 8;
    const int total = ((a.M + 127) >> 7) * tiles_n;
    const int S = a.chunks;                          // stages (of two 32-channel chunks) per tile
    // PERSISTENT: the workgroup walks tiles blockIdx.x, + gridDim.x, ... as ONE stream of stages - the staging pipeline
    // (raw rows two stages ahead, the split image one stage ahead, weights one K step ahead) runs across tile boundaries,
    // and a tile's output stores drain while the next tile multiplies.  A 128 x 256 x 512 tile is only ~25 us of MFMAs:
    // with one workgroup per CU retiring per tile, its prologue and the wait for its 128 KB of stores were 30-40 % of the
    // launch (timing ablation without the epilogue: 243 -> 171 us on 512 -> 256 @32x32, B=128).
    const int nseq = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int G = nseq * S;
    const float* zp = a.zero;
    auto tile_origin = [&](int seq, int& m0, int& tn) {
        const int t = xcd_remap((int)blockIdx.x + seq * (int)gridDim.x, total);
        const int tm = t / tiles_n;
        tn = t - tm * tiles_n;
        m0 = tm * 128;
    };

    f32x4 hv[4];
    int lseq = 0, lst = 0, lm0, ltn;                 // cursor of the row loads
    tile_origin(0, lm0, ltn);
    auto load_rows = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = lm0 + (((tid >> 3) + 64 * i) & 127);
            const int c0 = (lst * 2 + (i >> 1)) * 32;          // items 0, 1: first chunk of the stage, 2, 3: second
            const bool second = c0 >= a.C1;
            const float* src = second ? a.x2 : a.x1;
            const int cs = second ? a.C2 : a.C1;
            const int cc = (second ? c0 - a.C1 : c0) + c4 * 4;
            hv[i] = ld4(gm < a.M ? src + ((long long)gm * cs + cc) : zp);
        }
        if (++lst == S) {
            lst = 0;
            if (++lseq < nseq) tile_origin(lseq, lm0, ltn);
        }
    };
    auto store_rows = [&](int img) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned h0, m0_, l0, h1, m1, l1;
            split3(hv[i][0], hv[i][1], h0, m0_, l0);
            split3(hv[i][2], hv[i][3], h1, m1, l1);
            const int prow = (tid >> 3) + 64 * i;
            unsigned char* q = smem + img * PW8_IMG + prow * ROWB + (((c4 >> 1) ^ lds_swz(prow)) << 4) + (c4 & 1) * 8;
            *reinterpret_cast<u32x2*>(q) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(q + PW8_LIMB) = u32x2{m0_, m1};
            *reinterpret_cast<u32x2*>(q + 2 * PW8_LIMB) = u32x2{l0, l1};
        }
    };

    const int r16 = lane & 15, kq = lane >> 4;
    const long long tile_u4 = (long long)a.chunks * 2 * TAP_U4;                  // fragments of one 64-channel half tile
    const u32x4* wbase = a.wfrag + (long long)(wave >> 1) * tile_u4 + lane + (wave & 1) * 2 * 3 * 64;
    u32x4 bq[2][2][3];
    auto load_b = [&](const u32x4* p, u32x4 (&dst)[2][3]) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int l = 0; l < 3; ++l) dst[nb][l] = p[(nb * 3 + l) * 64];
    };
    f32x4v acc[8][2];

    // A stage = 2 K steps (chunks) x 4 quarters of the 128 rows (two 16-row blocks each): 24 MFMAs per quarter.  The A
    // fragments of quarter j + 1 are read from LDS before the MFMAs of quarter j are issued (rolling two-deep buffer).
    // Per accumulator the six limb products keep their order, smallest first.
    u32x4 fa[2][2][3];
    auto read_q = [&](int img, int j, u32x4 (&dst)[2][3]) {          // quarter j of the stage: chunk j >> 2, blocks 2 (j & 3), +1
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int prow = (j >> 2) * 128 + ((j & 3) * 2 + mb) * 16 + r16;
            const unsigned char* q = smem + img * PW8_IMG + prow * ROWB + ((kq ^ lds_swz(prow)) << 4);
#pragma unroll
            for (int l = 0; l < 3; ++l) dst[mb][l] = *reinterpret_cast<const u32x4*>(q + l * PW8_LIMB);
        }
    };
    auto mfma_q = [&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr int pp = j >> 2, b0 = (j & 3) * 2;
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[b0 + mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, bq[pp][nb][PB[t]]), __builtin_bit_cast(bf16x8, fa[j & 1][mb][PA[t]]),
                        acc[b0 + mb][nb], 0, 0, 0);
    };

    const bool early = wave >= 4;           // staging slot of this wave within a stage (below)
    int m0, tn;
    tile_origin(0, m0, tn);
    const u32x4* wp = wbase + (long long)tn * 4 * tile_u4;
    load_rows();
    load_b(wp, bq[0]);
    store_rows(0);
    if (G > 1) load_rows();
    __syncthreads();
    int g = 0;
    for (int seq = 0; seq < nseq; ++seq) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        int nm0 = m0, ntn = tn;
        if (seq + 1 < nseq) tile_origin(seq + 1, nm0, ntn);
        const u32x4* wp_next = wbase + (long long)ntn * 4 * tile_u4;
        for (int st = 0; st < S; ++st, ++g) {
            const int img = g & 1;
            if (!(ABL & 2)) load_b(wp + (long long)(2 * st + 1) * TAP_U4, bq[1]);
            // SIMD partners (waves w, w + 4) would split / store the next image at the same moment and leave the matrix
            // pipe idle together: waves 4-7 do it at the head of the stage, waves 0-3 between the two K steps (512 -> 256
            // @32x32 B=128: 213 -> 201 us; the smaller shapes do not move)
            if (early && !(ABL & 1)) {
                if (g + 1 < G) store_rows(img ^ 1);
                if (g + 2 < G) load_rows();
                __builtin_amdgcn_sched_barrier(0);
            }
            read_q(img, 0, fa[0]);
#define PW8_Q(J)                                                       \
            if (J < 7) read_q(img, J + 1, fa[(J + 1) & 1]);            \
            __builtin_amdgcn_sched_barrier(0);                         \
            mfma_q(std::integral_constant<int, J>{});                  \
            __builtin_amdgcn_sched_barrier(0);
            PW8_Q(0) PW8_Q(1) PW8_Q(2) PW8_Q(3)
            if (!early && !(ABL & 1)) {
                if (g + 1 < G) store_rows(img ^ 1);        // everyone left image img ^ 1 at the last barrier
                if (g + 2 < G) load_rows();
            }
            // the first chunk's weights are done with: the next stage's first chunk (of the next tile after the last stage)
            if (!(ABL & 2)) load_b(st + 1 < S ? wp + (long long)(2 * st + 2) * TAP_U4 : wp_next, bq[0]);
            __builtin_amdgcn_sched_barrier(0);
            PW8_Q(4) PW8_Q(5) PW8_Q(6) PW8_Q(7)
#undef PW8_Q
            __syncthreads();
        }
        if (ABL & 4) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) t += acc[i][0][0] + acc[i][1][1];
            if (t == 1.2345f) a.C[tid] = t;
        } else {
            dconv_epilogue_n32<8>(a, acc, m0, tn * 256 + wave * 32, lane, 0);
        }
        m0 = nm0; tn = ntn; wp = wp_next;
    }
}

template <int ABL = 0>
int launch_pw8(const DConvArgs& a, hipStream_t stream, const char* name) {
    constexpr size_t LDS = (size_t)2 * PW8_IMG;
    static PsldPerDeviceFlag configured_; bool& configured = configured_.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw8_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)LDS);
        if (e != hipSuccess) {
            psld_set_error("%s: hipFuncSetAttribute failed: %s", name, hipGetErrorString(e));
            return PSLD_ERR_LAUNCH;
        }
        configured = true;
    }
    const int total = cdiv(a.M, 128) * (a.N / 256);
    hipLaunchKernelGGL(pw8_kernel<ABL>, dim3((unsigned)(total < 256 ? total : 256)), dim3(512), LDS, stream, a);
    PSLD_CHECK_LAUNCH(name);
    return PSLD_OK;
}

template <int CB, bool XLP, int ABL = 0>
int launch_dwgrad(const DWgradArgs& a, int nsplit, hipStream_t stream) {
    constexpr size_t LDS = (size_t)3 * (WG_AROWS * (64 * CB + 32) + WG_BLIMB);
    static PsldPerDeviceFlag configured_; bool& configured = configured_.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dwgrad_kernel<CB, XLP, ABL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) {
            psld_set_error("psld_conv3x3_wgrad_split_f32: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return PSLD_ERR_LAUNCH;
        }
        configured = true;
    }
    hipLaunchKernelGGL((dwgrad_kernel<CB, XLP, ABL>), dim3((unsigned)(a.cout_tiles * a.cin_tiles * 3 * nsplit)), dim3(256), LDS,
                       stream, a);
    PSLD_CHECK_LAUNCH("psld_conv3x3_wgrad_split_f32");
    return PSLD_OK;
}

template <int NH, int TAPS, bool PW, int MT = 128>
int launch_dconv(const DConvArgs& a, int nsplit, hipStream_t stream, const char* name) {
    constexpr size_t LDS = (size_t)3 * NH * 32 * ROWB;
    static PsldPerDeviceFlag configured_; bool& configured = configured_.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dconv_kernel<NH, TAPS, PW, MT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) {
            psld_set_error("%s: hipFuncSetAttribute failed: %s", name, hipGetErrorString(e));
            return PSLD_ERR_LAUNCH;
        }
        configured = true;
    }
    dim3 grid((unsigned)(cdiv(a.M, MT) * (a.N / 128)), (unsigned)nsplit);
    hipLaunchKernelGGL((dconv_kernel<NH, TAPS, PW, MT>), grid, dim3(256), LDS, stream, a);
    PSLD_CHECK_LAUNCH(name);
    return PSLD_OK;
}

template <int RG, bool DB, int MT = 128>
int launch_dconv_lp(const DConvArgs& a, int nsplit, hipStream_t stream, const char* name) {
    constexpr size_t LDS = (size_t)(DB ? 2 : 1) * 3 * RG * 16 * ROWB;
    static PsldPerDeviceFlag configured_; bool& configured = configured_.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dconv_lp_kernel<RG, DB, MT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) {
            psld_set_error("%s: hipFuncSetAttribute failed: %s", name, hipGetErrorString(e));
            return PSLD_ERR_LAUNCH;
        }
        configured = true;
    }
    dim3 grid((unsigned)(cdiv(a.M, MT) * (a.N / 128)), (unsigned)nsplit);
    hipLaunchKernelGGL((dconv_lp_kernel<RG, DB, MT>), grid, dim3(256), LDS, stream, a);
    PSLD_CHECK_LAUNCH(name);
    return PSLD_OK;
}

// split the K stages over extra workgroups when the output grid cannot fill 256 CUs x 2 slots; returns the number
// of slabs (1 = write the output directly) and fills the slab fields of `a`
int plan_split(DConvArgs& a, const PsldEpilogue& e, float* y, int ldy, void* workspace, long long ws_bytes, int mt = 128) {
    const long long tiles = (long long)cdiv(a.M, mt) * (a.N / 128);
    int ns = 1;
    if (workspace && tiles < 384 && (!e.gn_part || psld_detail_conv_reduce_gn_ok(a.M, a.N, e)) && ldy % 4 == 0 && aligned16(y) && (!e.bias || aligned16(e.bias)) &&
        (!e.rowbias || (aligned16(e.rowbias) && e.ld_rowbias % 4 == 0)) && (!e.res || (aligned16(e.res) && e.ldres % 4 == 0))) {
        ns = (int)(512 / tiles);
        if (ns > 8) ns = 8;
        // at least two 32-channel chunks per K range - one when the output has so few tiles that even eight ranges leave
        // most CUs idle (256 -> 256 @8x8, batch 16: 16 tiles; measured 34 -> 44 TFLOP/s)
        const int min_chunks = tiles <= 32 ? 1 : 2;
        if (ns > a.chunks / min_chunks) ns = a.chunks / min_chunks;
        while (ns > 1 && (long long)ns * a.M * a.N * (long long)sizeof(float) > ws_bytes) --ns;
        if (ns < 1) ns = 1;
    }
    a.chunks_per_split = cdiv(a.chunks, ns);
    ns = cdiv(a.chunks, a.chunks_per_split);
    if (ns >= 2) {
        a.C = reinterpret_cast<float*>(workspace);
        a.ldc = a.N;
        a.c_stride_split = (long long)a.M * a.N;
        a.e = make_epilogue(nullptr);
    } else {
        a.C = y;
        a.ldc = ldy;
        a.c_stride_split = 0;
        a.e = e;
    }
    a.v4 = (a.ldc % 4 == 0) && aligned16(a.C) && (a.c_stride_split % 4 == 0) &&
           (!a.e.res || (a.e.ldres % 4 == 0 && aligned16(a.e.res))) && (!a.e.bias || aligned16(a.e.bias)) &&
           (!a.e.rowbias || (a.e.ld_rowbias % 4 == 0 && aligned16(a.e.rowbias)));
    return ns;
}

// Pixel rows the LDS halo image spends per image row.  W + 2, except 8-wide maps on 64-row tiles: there a 16-row MFMA block
// spans two image rows, and with a pitch of 10 the rows b..b+7 | b+10..b+17 put two rows of the same 16-lane ds_read_b128
// group on one 16-byte slot of the bank line for every XOR swizzle that is linear in the row index (enumerated; measured
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.47-0.50, profiles/r02/pmc_tile_kernels.md).  With a pitch of 16 the second
// image row sits 16 rows further and lds_swz is conflict-free again: conflict share 0.474 / 0.500 -> 0.000, MFMA busy 0.503
// -> 0.504 / 0.516 -> 0.541, 74.2 -> 73.1 / 71.3 -> 68.8 us under the counters (profiles/r03/pmc_lds_w8_pitch{0,1}.md) and
// nothing in an interleaved A/B (profiles/r03/ab_w8.txt: 181.1 vs 181.6 TFLOP/s at B=128, 58.5 vs 60.0 at B=16): the
// conflict cycles sat in the shadow of the MFMAs; what holds these launches at half the pipe is their grid (256
// workgroups of 64 rows at B=128: one per CU).
int dconv_pitch(int w, int mt) { return (w == 8 && mt == 64) ? 16 : w + 2; }

bool dconv_geometry(int h, int w, int* nseg, int* rps, int* halo_px, int mt = 128) {
    if (w != 8 && w != 16 && w != 32 && w != 64) return false;
    const int hw = h * w;
    if (hw >= mt) {
        if (hw % mt || mt % w) return false;
        *nseg = 1;
        *rps = mt / w;
    } else {
        if (mt % hw) return false;
        *nseg = mt / hw;
        *rps = h;
    }
    *halo_px = *nseg * (*rps + 2) * dconv_pitch(w, mt);
    return *halo_px <= 9 * 32;
}

// Tile height of a 3x3 limb convolution: 64-row tiles when 128-row tiles leave the grid short of 384 workgroups - the same
// bound under which plan_split starts cutting the K range - and the geometry allows them.  (8-wide maps on 64-row tiles at
// every batch size measured slower at large batches - B=512: 236-251 vs 244-267 TFLOP/s - and are not offered.)
int dconv_tile_rows(const DConvArgs& a, int h, int w) {
    int nseg, rps, halo;
    const long long tiles128 = (long long)cdiv(a.M, 128) * (a.N / 128);
    // halo of a 64-row tile <= 160 pixel rows: the images the 64-row instances are built with (10 row groups / 5 items)
    return (tiles128 < 384 && a.M % 64 == 0 && dconv_geometry(h, w, &nseg, &rps, &halo, 64) && halo <= 160) ? 64 : 128;
}

}  // namespace

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" int psld_set_math_mode(int mode) {
    PSLD_CHECK_ARG(mode == PSLD_MATH_F32 || mode == PSLD_MATH_BF16X6, "psld_set_math_mode: unknown mode %d", mode);
    g_math_mode = mode;
    return PSLD_OK;
}

extern "C" int psld_get_math_mode(void) { return math_mode(); }

extern "C" long long psld_conv3x3_frag_bytes(int cout, int cin) { return (long long)cout * cin * 9 * 6; }

extern "C" int psld_conv3x3_split_supported(int c1, int c2, int batch, int h, int w, int cout) {
    int nseg, rps, halo;
    return c1 > 0 && c2 >= 0 && c1 % 32 == 0 && c2 % 32 == 0 && cout > 0 && cout % 128 == 0 && batch > 0 &&
           dconv_geometry(h, w, &nseg, &rps, &halo);
}

extern "C" int psld_pack_conv3x3_frag(const float* w_oihw, void* wfrag, int cout, int cin, int dgrad,
                                      hipStream_t stream) {
    PSLD_CHECK_ARG(w_oihw && wfrag, "psld_pack_conv3x3_frag: null pointer");
    const int n_out = dgrad ? cin : cout, k_in = dgrad ? cout : cin;
    PSLD_CHECK_ARG(n_out > 0 && k_in > 0 && n_out % 128 == 0 && k_in % 32 == 0,
                   "psld_pack_conv3x3_frag: needs out channels %%128 and in channels %%32 (got %d, %d)", n_out, k_in);
    if (dgrad) return launch_pack(w_oihw, wfrag, n_out, k_in, 9, 9, (long long)cin * 9, 1, 1, stream, "psld_pack_conv3x3_frag");
    return launch_pack(w_oihw, wfrag, n_out, k_in, 9, (long long)cin * 9, 9, 1, 0, stream, "psld_pack_conv3x3_frag");
}

extern "C" int psld_pack_frag_batch(const long long* table_dev, int entries, long long total_items, hipStream_t stream) {
    PSLD_CHECK_ARG(table_dev && entries > 0 && total_items > 0, "psld_pack_frag_batch: bad args");
    const long long want = (total_items + 255) / 256;
    hipLaunchKernelGGL(pack_frag_batch_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, stream,
                       table_dev, entries, total_items);
    PSLD_CHECK_LAUNCH("psld_pack_frag_batch");
    return PSLD_OK;
}

extern "C" int psld_conv3x3_split_f32(const float* x1, int c1, const float* x2, int c2, int batch, int h, int w,
                                      const void* wfrag, int cout, float* y, int ldy, const psld_epilogue_t* epi,
                                      void* workspace, long long ws_bytes, hipStream_t stream) {
    PSLD_CHECK_ARG(x1 && wfrag && y && (c2 == 0 || x2), "psld_conv3x3_split_f32: null pointer");
    PSLD_CHECK_ARG(psld_conv3x3_split_supported(c1, c2, batch, h, w, cout),
                   "psld_conv3x3_split_f32: unsupported shape c1=%d c2=%d %dx%d cout=%d", c1, c2, h, w, cout);
    PSLD_CHECK_ARG(aligned16(x1) && (!x2 || aligned16(x2)) && aligned16(wfrag), "psld_conv3x3_split_f32: unaligned pointer");
    DConvArgs a{};
    a.x1 = x1; a.x2 = x2; a.C1 = c1; a.C2 = c2;
    a.B = batch; a.H = h; a.W = w;
    a.wfrag = reinterpret_cast<const u32x4*>(wfrag);
    a.N = cout; a.M = batch * h * w;
    a.chunks = (c1 + c2) / 32;
    int halo_px = 0;
    const PsldEpilogue e = make_epilogue(epi);
    a.e = e;
    const int mt = dconv_tile_rows(a, h, w);
    dconv_geometry(h, w, &a.nseg, &a.rps, &halo_px, mt);
    a.pitch = dconv_pitch(w, mt);
    a.zero = psld_detail_zero_page("psld_conv3x3_split_f32");
    if (!a.zero) return PSLD_ERR_LAUNCH;
    PSLD_CHECK_ARG(!e.gn_part || (e.gn_hw == h * w && e.gn_hw % 64 == 0 && !e.accumulate),
                   "psld_conv3x3_split_f32: gn_part needs gn_hw = h*w, a multiple of 64, and no accumulation");
    const int ns = plan_split(a, e, y, ldy, workspace, ws_bytes, mt);
    PSLD_CHECK_ARG(a.v4, "limb kernels: y, residual, bias and rowbias need 16-byte aligned rows (pointer and row stride)");
    const int nh = cdiv((long long)halo_px * 8, 256);
    const char* name = "psld_conv3x3_split_f32";
    int st;
    if (mt == 64) st = nh <= 4 ? launch_dconv<4, 9, false, 64>(a, ns, stream, name)
                               : launch_dconv<5, 9, false, 64>(a, ns, stream, name);
    else if (nh <= 6) st = launch_dconv<6, 9, false>(a, ns, stream, name);
    else if (nh <= 7) st = launch_dconv<7, 9, false>(a, ns, stream, name);
    else st = launch_dconv<9, 9, false>(a, ns, stream, name);
    if (st != PSLD_OK) return st;
    if (ns >= 2) return psld_detail_conv_reduce_epilogue(a.C, ns, a.M, cout, y, ldy, e, stream);
    return PSLD_OK;
}

extern "C" long long psld_limb_bytes(long long rows, int c) { return rows * (long long)c * LP_PIX_BYTES_PER_CH; }

extern "C" int psld_f32_to_limb(const float* x, long long rows, int c, void* y, hipStream_t stream) {
    PSLD_CHECK_ARG(x && y && rows > 0 && c > 0 && c % 32 == 0 && aligned16(x) && aligned16(y), "psld_f32_to_limb: bad args");
    const long long n4 = rows * (c / 4);
    const int blocks = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(f32_to_limb_kernel, dim3(blocks), dim3(256), 0, stream, x, rows, c, reinterpret_cast<unsigned char*>(y));
    PSLD_CHECK_LAUNCH("psld_f32_to_limb");
    return PSLD_OK;
}

extern "C" int psld_limb_to_f32(const void* y, long long rows, int c, float* x, hipStream_t stream) {
    PSLD_CHECK_ARG(x && y && rows > 0 && c > 0 && c % 32 == 0, "psld_limb_to_f32: bad args");
    const long long n = rows * c;
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(limb_to_f32_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const unsigned char*>(y), rows, c, x);
    PSLD_CHECK_LAUNCH("psld_limb_to_f32");
    return PSLD_OK;
}

extern "C" int psld_conv3x3_limb_f32(const void* x1, int c1, const void* x2, int c2, int batch, int h, int w,
                                     const void* wfrag, int cout, float* y, int ldy, const psld_epilogue_t* epi,
                                     void* workspace, long long ws_bytes, hipStream_t stream) {
    PSLD_CHECK_ARG(x1 && wfrag && y && (c2 == 0 || x2), "psld_conv3x3_limb_f32: null pointer");
    PSLD_CHECK_ARG(psld_conv3x3_split_supported(c1, c2, batch, h, w, cout),
                   "psld_conv3x3_limb_f32: unsupported shape c1=%d c2=%d %dx%d cout=%d", c1, c2, h, w, cout);
    PSLD_CHECK_ARG(aligned16(x1) && (!x2 || aligned16(x2)) && aligned16(wfrag), "psld_conv3x3_limb_f32: unaligned pointer");
    DConvArgs a{};
    a.x1 = reinterpret_cast<const float*>(x1); a.x2 = reinterpret_cast<const float*>(x2); a.C1 = c1; a.C2 = c2;
    a.B = batch; a.H = h; a.W = w;
    a.wfrag = reinterpret_cast<const u32x4*>(wfrag);
    a.N = cout; a.M = batch * h * w;
    a.chunks = (c1 + c2) / 32;
    int halo_px = 0;
    const PsldEpilogue e = make_epilogue(epi);
    const int mt = dconv_tile_rows(a, h, w);
    dconv_geometry(h, w, &a.nseg, &a.rps, &halo_px, mt);
    a.pitch = dconv_pitch(w, mt);
    a.zero = psld_detail_zero_page("psld_conv3x3_limb_f32");
    if (!a.zero) return PSLD_ERR_LAUNCH;
    PSLD_CHECK_ARG(!e.gn_part || (e.gn_hw == h * w && e.gn_hw % 64 == 0 && !e.accumulate),
                   "psld_conv3x3_limb_f32: gn_part needs gn_hw = h*w, a multiple of 64, and no accumulation");
    const int ns = plan_split(a, e, y, ldy, workspace, ws_bytes, mt);
    PSLD_CHECK_ARG(a.v4, "limb kernels: y, residual, bias and rowbias need 16-byte aligned rows (pointer and row stride)");
    const int rg = cdiv(halo_px, 16);
    const char* name = "psld_conv3x3_limb_f32";
    int st;
    // two images of RG <= 13 row groups (79,872 B) leave room for two workgroups per CU (163,840 B of LDS)
    if (mt == 64) st = rg <= 7 ? launch_dconv_lp<7, true, 64>(a, ns, stream, name)
                     : rg <= 9 ? launch_dconv_lp<9, true, 64>(a, ns, stream, name)
                               : launch_dconv_lp<10, true, 64>(a, ns, stream, name);
    else if (rg <= 12) st = launch_dconv_lp<12, true>(a, ns, stream, name);
    else if (rg <= 13) st = launch_dconv_lp<13, true>(a, ns, stream, name);
    else st = launch_dconv_lp<18, false>(a, ns, stream, name);
    if (st != PSLD_OK) return st;
    if (ns >= 2) return psld_detail_conv_reduce_epilogue(a.C, ns, a.M, cout, y, ldy, e, stream);
    return PSLD_OK;
}

extern "C" int psld_conv3x3_wgrad_split_supported(int cout, int cin, int batch, int h, int w) {
    return cout > 0 && cin > 0 && cout % 64 == 0 && cin % 64 == 0 && batch > 0 &&
           (w == 8 || w == 16 || w == 32 || w == 64) && (h * w) % 32 == 0;
}

extern "C" int psld_conv3x3_wgrad_split_cout_tile(int cout) { return cout % 128 ? 64 : 128; }

extern "C" int psld_conv3x3_wgrad_split_f32(const float* dy, int lddy, int cout, const float* x, int cin,
                                            const float* x2, int cin2, int batch, int h, int w, float* slabs,
                                            int cin_total, int col0, int nsplit, hipStream_t stream) {
    PSLD_CHECK_ARG(dy && x && slabs && nsplit >= 1 && cin2 >= 0 && (cin2 == 0 || x2), "psld_conv3x3_wgrad_split_f32: bad args");
    PSLD_CHECK_ARG(psld_conv3x3_wgrad_split_supported(cout, cin, batch, h, w) &&
                       (cin2 == 0 || psld_conv3x3_wgrad_split_supported(cout, cin2, batch, h, w)),
                   "psld_conv3x3_wgrad_split_f32: unsupported shape cout=%d cin=%d+%d %dx%d", cout, cin, cin2, h, w);
    PSLD_CHECK_ARG(aligned16(dy) && aligned16(x) && (cin2 == 0 || aligned16(x2)) && lddy % 4 == 0,
                   "psld_conv3x3_wgrad_split_f32: unaligned operand");
    DWgradArgs a{};
    a.dy = dy; a.lddy = lddy; a.x = x; a.cin = cin; a.x2 = x2; a.cin2 = cin2;
    a.B = batch; a.H = h; a.W = w;
    const int co_tile = psld_conv3x3_wgrad_split_cout_tile(cout);
    a.cout_tiles = cout / co_tile; a.cin_tiles = (cin + cin2) / 64;
    a.ktiles = batch * h * w / 32;
    a.ktiles_per_split = cdiv(a.ktiles, nsplit);
    PSLD_CHECK_ARG(cdiv(a.ktiles, a.ktiles_per_split) == nsplit, "psld_conv3x3_wgrad_split_f32: nsplit %d leaves empty slabs", nsplit);
    a.slabs = slabs + col0;
    a.ld_tap = cin_total;
    a.slab_stride = (long long)cout * 9 * cin_total;
    a.hw_w = (w < 32 ? w : 32) + 2;
    a.hrows = w >= 32 ? 1 : 32 / w;
#ifdef PSLD_ABLATIONS      // timing-only variants (wrong results): libpsld_hip_abl.so only (make -C tools/abl), never the product library
    static const int ws = [] { const char* v = getenv("PSLD_DWGRAD_WS"); return v ? atoi(v) : 1; }();
    if (!ws && co_tile == 128) return launch_dwgrad<4, false>(a, nsplit, stream);     // round 3's kernel, for A/B
    static const int abl = [] { const char* v = getenv("PSLD_DWGRAD_ABL"); return v ? atoi(v) : 0; }();
    if (abl && co_tile == 128) {
        switch (abl) {
            case 1: return launch_dwgrad<4, false, 1>(a, nsplit, stream);
            case 2: return launch_dwgrad<4, false, 2>(a, nsplit, stream);
            case 3: return launch_dwgrad<4, false, 3>(a, nsplit, stream);
            case 4: return launch_dwgrad<4, false, 4>(a, nsplit, stream);
            case 6: return launch_dwgrad<4, false, 6>(a, nsplit, stream);
        }
    }
#endif
    // 128-channel tiles: the wave-specialised kernel (512 threads, one workgroup per CU)
    return co_tile == 128 ? launch_dwgrad_ws(a, nsplit, stream) : launch_dwgrad<2, false>(a, nsplit, stream);
}

extern "C" int psld_conv3x3_wgrad_xlimb_f32(const float* dy, int lddy, int cout, const void* x_limb, int cin,
                                            const void* x2_limb, int cin2, int batch, int h, int w, float* slabs,
                                            int cin_total, int col0, int nsplit, hipStream_t stream) {
    PSLD_CHECK_ARG(dy && x_limb && slabs && nsplit >= 1 && cin2 >= 0 && (cin2 == 0 || x2_limb), "psld_conv3x3_wgrad_xlimb_f32: bad args");
    PSLD_CHECK_ARG(psld_conv3x3_wgrad_split_supported(cout, cin, batch, h, w) &&
                       (cin2 == 0 || psld_conv3x3_wgrad_split_supported(cout, cin2, batch, h, w)),
                   "psld_conv3x3_wgrad_xlimb_f32: unsupported shape cout=%d cin=%d+%d %dx%d", cout, cin, cin2, h, w);
    PSLD_CHECK_ARG(aligned16(dy) && aligned16(x_limb) && (cin2 == 0 || aligned16(x2_limb)) && lddy % 4 == 0,
                   "psld_conv3x3_wgrad_xlimb_f32: unaligned operand");
    DWgradArgs a{};
    a.dy = dy; a.lddy = lddy; a.x = reinterpret_cast<const float*>(x_limb); a.cin = cin;
    a.x2 = reinterpret_cast<const float*>(x2_limb); a.cin2 = cin2;
    a.B = batch; a.H = h; a.W = w;
    const int co_tile = psld_conv3x3_wgrad_split_cout_tile(cout);
    a.cout_tiles = cout / co_tile; a.cin_tiles = (cin + cin2) / 64;
    a.ktiles = batch * h * w / 32;
    a.ktiles_per_split = cdiv(a.ktiles, nsplit);
    PSLD_CHECK_ARG(cdiv(a.ktiles, a.ktiles_per_split) == nsplit, "psld_conv3x3_wgrad_xlimb_f32: nsplit %d leaves empty slabs", nsplit);
    a.slabs = slabs + col0;
    a.ld_tap = cin_total;
    a.slab_stride = (long long)cout * 9 * cin_total;
    a.hw_w = (w < 32 ? w : 32) + 2;
    a.hrows = w >= 32 ? 1 : 32 / w;
    return co_tile == 128 ? launch_dwgrad<4, true>(a, nsplit, stream) : launch_dwgrad<2, true>(a, nsplit, stream);
}

// ---- pointwise weight gradient ------------------------------------------------------------------------------
extern "C" int psld_gemm_tn_split_supported(int m, int n, int k) {
    return m > 0 && n > 0 && k > 0 && m % 128 == 0 && n % 128 == 0 && k % 32 == 0;
}

extern "C" int psld_gemm_tn_split_f32(int m, int n, int k, const float* a, int lda, const float* b, int ldb,
                                      const float* b2, int ldb2, int n2, float* slabs, int ldc, int nsplit,
                                      hipStream_t stream) {
    PSLD_CHECK_ARG(a && b && slabs && nsplit >= 1 && n2 >= 0 && (n2 == 0 || b2), "psld_gemm_tn_split_f32: bad args");
    PSLD_CHECK_ARG(psld_gemm_tn_split_supported(m, n, k) && n2 % 128 == 0,
                   "psld_gemm_tn_split_f32: unsupported shape m=%d n=%d+%d k=%d", m, n, n2, k);
    PSLD_CHECK_ARG(aligned16(a) && aligned16(b) && lda % 4 == 0 && ldb % 4 == 0 && lda >= m && ldb >= n && ldc >= n + n2 &&
                       (n2 == 0 || (aligned16(b2) && ldb2 % 4 == 0 && ldb2 >= n2)),
                   "psld_gemm_tn_split_f32: unaligned operand or short row stride");
    PWgradArgs p{};
    p.a = a; p.lda = lda; p.b = b; p.ldb = ldb; p.b2 = b2; p.ldb2 = ldb2; p.n1 = n;
    n += n2;
    p.tiles_i = m / 128; p.tiles_j = n / 128;
    p.ktiles = k / 32;
    p.ktiles_per_split = cdiv(p.ktiles, nsplit);
    PSLD_CHECK_ARG(cdiv(p.ktiles, p.ktiles_per_split) == nsplit, "psld_gemm_tn_split_f32: nsplit %d leaves empty slabs", nsplit);
    p.slabs = slabs; p.ldc = ldc;
    p.slab_stride = (long long)m * ldc;
    constexpr size_t LDS = (size_t)2 * 3 * 32 * 288;
    static PsldPerDeviceFlag configured_; bool& configured = configured_.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pwgrad_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) {
            psld_set_error("psld_gemm_tn_split_f32: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return PSLD_ERR_LAUNCH;
        }
        configured = true;
    }
    hipLaunchKernelGGL(pwgrad_kernel, dim3((unsigned)(p.tiles_i * p.tiles_j * nsplit)), dim3(256), LDS, stream, p);
    PSLD_CHECK_LAUNCH("psld_gemm_tn_split_f32");
    return PSLD_OK;
}

// ---- batched GEMM, both operands fp32 activations --------------------------------------------------------------
extern "C" int psld_bgemm_split_supported(int ta, int tb, int m, int n, int k) {
    return !(ta && tb) && m > 0 && n > 0 && k > 0 && m % 128 == 0 && n % 128 == 0 && k % 32 == 0;
}

extern "C" int psld_bgemm_split_f32(int ta, int tb, int m, int n, int k, const float* a, int lda, long long stride_a,
                                    const float* b, int ldb, long long stride_b, float* c, int ldc, long long stride_c,
                                    int batch, float alpha, hipStream_t stream) {
    PSLD_CHECK_ARG(a && b && c && batch >= 1, "psld_bgemm_split_f32: bad args");
    PSLD_CHECK_ARG(psld_bgemm_split_supported(ta, tb, m, n, k), "psld_bgemm_split_f32: unsupported ta=%d tb=%d m=%d n=%d k=%d",
                   ta, tb, m, n, k);
    PSLD_CHECK_ARG(aligned16(a) && aligned16(b) && lda % 4 == 0 && ldb % 4 == 0 && stride_a % 4 == 0 && stride_b % 4 == 0,
                   "psld_bgemm_split_f32: unaligned operand");
    BGemmArgs p{};
    p.a = a; p.lda = lda; p.sa = stride_a;
    p.b = b; p.ldb = ldb; p.sb = stride_b;
    p.c = c; p.ldc = ldc; p.sc = stride_c;
    p.tiles_i = m / 128; p.tiles_j = n / 128; p.ktiles = k / 32;
    p.alpha = alpha;
    if (ta) return launch_bgemm<true, false>(p, batch, stream);
    if (tb) return launch_bgemm<false, true>(p, batch, stream);
    return launch_bgemm<false, false>(p, batch, stream);
}

// ---- pointwise (NT GEMM with pre-split B) -------------------------------------------------------------------
extern "C" long long psld_gemm_frag_bytes(int n, int k) { return (long long)n * k * 6; }

extern "C" int psld_gemm_split_supported(int k1, int k2, int m, int n) {
    return k1 > 0 && k2 >= 0 && k1 % 32 == 0 && k2 % 32 == 0 && (k1 + k2) % 64 == 0 && n > 0 && n % 128 == 0 && m > 0;
}

extern "C" int psld_pack_gemm_frag(const float* b, void* bfrag, int n, int k, long long stride_n, long long stride_k,
                                   hipStream_t stream) {
    PSLD_CHECK_ARG(b && bfrag, "psld_pack_gemm_frag: null pointer");
    PSLD_CHECK_ARG(n > 0 && k > 0 && n % 128 == 0 && k % 64 == 0, "psld_pack_gemm_frag: needs n %%128 and k %%64 (got %d, %d)", n, k);
    return launch_pack(b, bfrag, n, k, 1, stride_n, stride_k, 0, 0, stream, "psld_pack_gemm_frag");
}

extern "C" int psld_gemm_split_f32(const float* a1, int k1, const float* a2, int k2, int m, const void* bfrag, int n,
                                   float* y, int ldy, const psld_epilogue_t* epi, void* workspace, long long ws_bytes,
                                   hipStream_t stream) {
    PSLD_CHECK_ARG(a1 && bfrag && y && (k2 == 0 || a2), "psld_gemm_split_f32: null pointer");
    PSLD_CHECK_ARG(psld_gemm_split_supported(k1, k2, m, n), "psld_gemm_split_f32: unsupported shape k1=%d k2=%d m=%d n=%d", k1, k2, m, n);
    PSLD_CHECK_ARG(aligned16(a1) && (!a2 || aligned16(a2)) && aligned16(bfrag), "psld_gemm_split_f32: unaligned pointer");
    DConvArgs a{};
    a.x1 = a1; a.x2 = a2; a.C1 = k1; a.C2 = k2;
    a.B = 1; a.H = 1; a.W = 1;
    a.wfrag = reinterpret_cast<const u32x4*>(bfrag);
    a.N = n; a.M = m;
    a.chunks = (k1 + k2) / 64;          // stages of two 32-channel chunks
    a.nseg = 1; a.rps = 1; a.pitch = a.W + 2;
    a.zero = psld_detail_zero_page("psld_gemm_split_f32");
    if (!a.zero) return PSLD_ERR_LAUNCH;
    const PsldEpilogue e = make_epilogue(epi);
    PSLD_CHECK_ARG(!e.gn_part || (e.gn_hw > 0 && e.gn_hw % 64 == 0 && m % e.gn_hw == 0 && !e.accumulate),
                   "psld_gemm_split_f32: gn_part needs gn_hw (rows per image) a multiple of 64 dividing m, and no accumulation");
    // eight-wave 128 x 256 tiles from 128 of them on (below: the four-wave 128 x 128 kernel).  Half a chip of persistent
    // workgroups still beats 256 four-wave tiles split in two K ranges plus their reduction launch: B=16 step 537 / 539 ->
    // 550 / 547 images/s, B=64 903 -> 908; from 64 tiles on it does not (543 / 542).
    const bool wide = n % 256 == 0 && (long long)cdiv(m, 128) * (n / 256) >= 128;
    const int ns = plan_split(a, e, y, ldy, wide ? nullptr : workspace, ws_bytes);
    PSLD_CHECK_ARG(a.v4, "limb kernels: y, residual, bias and rowbias need 16-byte aligned rows (pointer and row stride)");
    if (wide) {
#ifdef PSLD_ABLATIONS      // timing-only variants (wrong results): libpsld_hip_abl.so only
        static const int abl = [] { const char* v = getenv("PSLD_PW8_ABL"); return v ? atoi(v) : 0; }();
        switch (abl) {
            case 1: return launch_pw8<1>(a, stream, "psld_gemm_split_f32");
            case 2: return launch_pw8<2>(a, stream, "psld_gemm_split_f32");
            case 3: return launch_pw8<3>(a, stream, "psld_gemm_split_f32");
            case 4: return launch_pw8<4>(a, stream, "psld_gemm_split_f32");
            case 7: return launch_pw8<7>(a, stream, "psld_gemm_split_f32");
        }
#endif
        return launch_pw8<0>(a, stream, "psld_gemm_split_f32");
    }
    const int st = launch_dconv<8, 2, true>(a, ns, stream, "psld_gemm_split_f32");
    if (st != PSLD_OK) return st;
    if (ns >= 2) return psld_detail_conv_reduce_epilogue(a.C, ns, m, n, y, ldy, e, stream);
    return PSLD_OK;
}
