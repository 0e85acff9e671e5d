// Weight gradient of the 3x3 stride-1 convolutions in the Winograd F(2x2, 3x3) domain (replaces the
// backward of nn.Conv2d, /root/reference/main/models/score_fn/song_sde/layers.py:103-109, for the 32x32 / 16x16 levels).
//
//   forward:   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          per 2x2 output tile / 4x4 input tile   (conv_wino.hip)
//   here:      dU[xi] = sum_tiles (A dY A^T)[xi] (x) (B^T d B)[xi]      xi = one of the 16 positions, (x) = outer product over (c_out, c_in)
//              dg     = G^T dU G
//
// i.e. 16 TN GEMMs [c_out x c_in] += M_xi^T V_xi over K = tiles = B H W / 4 instead of 9 over K = pixels: 16 products per 4
// pixels instead of 36 - 2.25x fewer matrix instructions for the same 2 M N 9 C_in.  It is a re-association of fp32
// arithmetic, not a narrower one: both transforms are fp32 adds of 2 / 4 values (A, B hold 0 / +-1), the transformed values
// are split EXACTLY into three bf16 limbs like every other limb kernel (limb.h) and every product is the same six-limb-
// product sum accumulated in fp32; G's halves are exact scalings applied once per layer by the reduction kernel.
//
// What round 5 priced as the obstacle - BOTH operands transformed and split per use, 3.6x the vector work per matrix cycle of
// dwgrad_ws_kernel - is attacked by the tile shape: a workgroup owns ONE position and 256 (c_out) x 128 (c_in) of its GEMM.
// The vector work per MFMA goes with (1 / tile_m + 1 / tile_n): one position x 256 x 128 needs 16 transformed values per MFMA
// where four positions x 128 x 128 (the same accumulator budget) would need 21 and sixteen x 64 x 64 would need 43.  What a
// position costs more than a tap is loads: a transformed value is a signed sum of 1, 2 or 4 pixels, read straight from
// global memory / L2 (the 16 x c_in-tile workgroups of one K range sit on ONE XCD and walk it in step: one HBM fetch).
//
// Shape of the kernel (the skeleton of dwgrad_ws_kernel, conv_split.hip):
//   512 threads, one workgroup per CU, 156 KB of LDS = two images of { dY-side [3 limbs][32 tiles][256 ch], x-side [3][32][128] };
//   waves 4-7 (producers): per K tile of 32 Winograd tiles, thread = (tile, channel quad) items - 4 x-items of 4 loads, 8
//     dY-items of 1 / 2 / 4 loads - combine, split3, ds_write_b64 per limb into the NEXT image; every item has a register slot
//     that is refilled with the same item of the tile after as soon as it is consumed (a whole K tile for a load to land);
//   waves 0-3 (consumers): 128 (c_out) x 64 (c_in) each = 8 x 4 blocks of v_mfma_f32_16x16x32_bf16 x 6 limb products = 192 MFMAs
//     per K tile on transposed fragment reads (ds_read_b64_tr_b16, k slot = tile 4g + 16j + q as in dwgrad_kernel), 128
//     accumulator registers;
//   one barrier per K tile; slabs[split][position][c_out][c_in] -> wwgrad_reduce_kernel: sum over the splits in fixed order,
//     G^T . G with the folded signs, written (or added) to the OIHW gradient.
// Signs: A's last row is [0, -1]: the dY-side values of positions with i = 3 or j = 3 are formed WITHOUT that sign (one load,
// no negation) and the reduction applies it (Gs = diag(1, 1, 1, -1) G).
#include "limb.h"
#include "psld_hip.h"
#include "tile_shared.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ u32x2 lds_tr16(const unsigned char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) void*)(p));
    return __builtin_bit_cast(u32x2, v);
}

constexpr int WW_CI = 128;                       // c_in per workgroup; c_out per workgroup CO = 256 (128 for layers of 128 output channels)
constexpr int WW_RSB = WW_CI * 2 + 32;           // 288: x-side row stride (bytes per tile row and limb); 32 mod 256 like pwgrad_kernel's:
constexpr int WW_BLIMB = 32 * WW_RSB;            // eight consecutive rows cover all 64 banks
template <int CO>
struct WWGeom {
    static constexpr int RSA = CO * 2 + 32;      // 544 / 288: dY-side row stride
    static constexpr int ALIMB = 32 * RSA;
    static constexpr int IMG = 3 * (ALIMB + WW_BLIMB);           // 79 872 / 55 296 bytes
    static constexpr size_t LDS = 2 * (size_t)IMG;               // 159 744 / 110 592
    static constexpr int NY = CO / 32;           // dY items per producer thread and K tile (8 / 4)
    static constexpr int CB = CO / 32;           // 16-channel c_out blocks per consumer wave (wave tile CO/2 x 64)
};

struct WWgradArgs {
    const float* dy;
    int lddy;
    const float* x;
    int cin;                // channels of x (row stride)
    const float* x2;        // second source of a channel concatenation (or null): c_in tiles beyond cin read it
    int cin2;
    int H, W;
    int lg_tw, th_mask, tw_mask;    // tiles per row = W / 2 = 1 << lg_tw; tile rows per image - 1; tiles per row - 1
    int rows_per_kt;        // tile rows of a K tile = 32 >> lg_tw
    int cout_tiles, cin_tiles;
    int ktiles, ktiles_per_split;       // the last split may be shorter (never empty)
    float* slabs;           // [split][16][cout][cin_total]
    int cout, cin_total;
    int pos_override;       // ablation library only (PSLD_WWGRAD_POS): every workgroup takes this position (-1: its own)
};

// rows / columns of the 4x4 input patch that position index i combines: V_i = d[r1] + s * d[r2]  (B^T rows)
__device__ __forceinline__ void x_combo(int i, int& r1, int& r2, float& s) {
    r1 = i == 0 ? 0 : (i == 2 ? 2 : 1);
    r2 = i == 0 ? 2 : (i == 2 ? 1 : (i == 3 ? 3 : 2));
    s = i == 1 ? 1.f : -1.f;
}
// rows / columns of the 2x2 output-gradient tile: M_i = y[r1] (+ s * y[1] when n == 2); i == 3: y[1], sign folded into the reduction
__device__ __forceinline__ void y_combo(int i, int& r1, int& n, float& s) {
    r1 = i == 3 ? 1 : 0;
    n = (i == 1 || i == 2) ? 2 : 1;
    s = i == 2 ? -1.f : 1.f;
}

// limb.h's split3 with the conversion written as a vector cast instead of inline asm: after every inline-asm
// v_cvt_pk_bf16_f32 hipcc pads an `s_nop 0` (64 of them per staged K tile pair here); in this kernel the cast form compiles
// to exactly three conversions per pair (checked in the ISA: 11 instructions per pair, no nop).  Same values bit for bit.
typedef __attribute__((ext_vector_type(2))) float ww_f32x2;
__device__ __forceinline__ unsigned ww_cvt_pk(float a, float b) {
    const ww_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void ww_split3(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = ww_cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    mid = ww_cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
    lo = ww_cvt_pk(s0, s1);
}

template <int ABL = 0>
__device__ __forceinline__ void ww_store(unsigned char* d, int limb_stride, const f32x4& v) {
    unsigned h0, m0, l0, h1, m1, l1;
    if constexpr ((ABL & 1) != 0) {
        h0 = m0 = l0 = __float_as_uint(v[0]) ^ __float_as_uint(v[1]);
        h1 = m1 = l1 = __float_as_uint(v[2]) ^ __float_as_uint(v[3]);
    } else {
        ww_split3(v[0], v[1], h0, m0, l0);
        ww_split3(v[2], v[3], h1, m1, l1);
    }
    *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(d + limb_stride) = u32x2{m0, m1};
    *reinterpret_cast<u32x2*>(d + 2 * limb_stride) = u32x2{l0, l1};
}

__device__ __forceinline__ f32x4 bload(const __amdgpu_buffer_rsrc_t& r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t brsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffff, 0x00020000);
}

// ABL: timing-only ablations (wrong results; libpsld_hip_abl.so only, PSLD_WWGRAD_ABL, tools/bench_wwgrad.py): 1 = no limb split
// (raw halves stored), 2 = no global loads (the slots keep their first contents), 4 = no MFMAs, 8 = no fragment reads, 16 = the
// product kernel with PSLD_WWGRAD_POS honoured, 32 = consumers only keep the barriers, 64 = producers only keep the barriers
// after the first tile.  Measured (profiles/r06/wwgrad_ablations.txt, 256->256 @32 B=128, kernel + reduction 381 us): consumers
// alone 298, producers alone 277, no split 326, no global loads 274; s_setprio(3) for the consumers 377-382 (nothing).  Either
// role alone runs near its own floor (the consumers at the clock the chip holds under MFMA load, the producers at their VALU +
// the 64 B/clk L2->L1 path: 192 KB of loads per K tile for a centre position); together they share each SIMD's issue port, the
// LDS and one barrier per K tile.  A form of the consumers on v_mfma_f32_32x32x16_bf16 (half as many MFMAs to issue; unpadded
// rows with the 64-byte granules swizzled by row & 3 for its 4-row transposed reads) passed the parity tests on the eight
// shapes: 372.5 -> 364.8 us on 256->256 @32, 713 -> 712 on 512->256 @32, 461 -> 437 on 128->128 @64, +-0 on the 16x16 level
// (profiles/r06/wwgrad_mfma32.txt) - the issue port alone is not the lever; removed again.
template <int ABL = 0, int CO = 256>
__global__ void __launch_bounds__(512) wwgrad_ws_kernel(const WWgradArgs a) {
    using G = WWGeom<CO>;
    constexpr int WW_CO = CO, WW_RSA = G::RSA, WW_ALIMB = G::ALIMB, WW_IMG = G::IMG, NY = G::NY, CB = G::CB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // the 16 positions x c_in tiles x c_out tiles of one K range are consecutive ids: one XCD, one L2
    const int units = 16 * a.cin_tiles * a.cout_tiles;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = vid / units;
    int rest = vid - split * units;
    int pos = rest & 15;
    if constexpr (ABL != 0) {
        if (a.pos_override >= 0) pos = a.pos_override;
    }
    rest >>= 4;
    const int ci_tile = rest % a.cin_tiles, co_tile = rest / a.cin_tiles;
    const int pi = pos >> 2, pj = pos & 3;
    const int co0 = co_tile * WW_CO, ci_out = ci_tile * WW_CI;
    const int kt_beg = split * a.ktiles_per_split;
    const int nkt = min(a.ktiles_per_split, a.ktiles - kt_beg);

    if (wave >= 4) {
        // ---- producers ------------------------------------------------------------------------------------------------------
        const bool second = ci_out >= a.cin;
        const float* xsrc = second ? a.x2 : a.x;
        const int xc = second ? a.cin2 : a.cin;
        const int ci0 = second ? ci_out - a.cin : ci_out;
        const int pt = tid & 255;
        int xr1, xr2, xc1, xc2, yr1, ynr, yc1, ync;
        float xsr, xsc, ysr, ysc;
        x_combo(pi, xr1, xr2, xsr);
        x_combo(pj, xc1, xc2, xsc);
        y_combo(pi, yr1, ynr, ysr);
        y_combo(pj, yc1, ync, ysc);
        const bool need_top = xr1 == 0, need_bot = xr2 == 3;        // the one row of this position that can leave the image
        // x items: channel quad pt & 31, tile row k = (pt >> 5) + 8 i
        const int qx = pt & 31, kx0 = pt >> 5;
        unsigned xoff[4];       // byte offset of tile k's patch origin (pixel row 2 dR, column 2 tx) from the K tile's origin
        int xdr[4];             // tile row of the item inside the K tile
        unsigned xcol[4];       // bit 0: column c1 inside the image, bit 1: column c2
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = kx0 + 8 * i;
            const int tx = k & a.tw_mask, dr = k >> a.lg_tw;
            xoff[i] = (unsigned)((((2 * dr) * a.W + 2 * tx) * xc + qx * 4) * 4);
            xdr[i] = dr;
            xcol[i] = ((xc1 == 0 && tx == 0) ? 0u : 1u) | ((xc2 == 3 && tx == a.tw_mask) ? 0u : 2u);
        }
        // dY items: channel quad pt & (CO / 4 - 1), tile row k = pt / (CO / 4) + (32 / NY) i
        constexpr int QY = CO / 4, KSTEP = 32 / NY;
        const int qy = pt & (QY - 1), ky0 = pt / QY;
        unsigned yoff[NY];
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int k = ky0 + KSTEP * i;
            const int tx = k & a.tw_mask, dr = k >> a.lg_tw;
            yoff[i] = (unsigned)((((2 * dr) * a.W + 2 * tx) * a.lddy + qy * 4) * 4);
        }
        const int lds_x = kx0 * WW_RSB + qx * 8, lds_y = ky0 * WW_RSA + qy * 8;

        // One register SLOT per item of a K tile (4 x-items of four loads, 8 dY-items of NR x NC loads): an item of tile t is
        // combined, split and stored from its slot and the slot is refilled at once with the same item of tile t + 1 - every
        // load has a whole K tile of work (~3,000 cycles) to land, whatever the L2 does under load.  (The first form kept three
        // groups of four items and gave a group one group's work to land: 486 us on 256->256 @32 B=128 where this runs 404.)
        auto stage = [&](auto nr_tag, auto nc_tag) {
            constexpr int NR = decltype(nr_tag)::value, NC = decltype(nc_tag)::value, NL = NR * NC;
            f32x4 sx[4][4];         // [item][r1c1, r1c2, r2c1, r2c2]
            f32x4 sy[NY][NL];
            auto load_tile = [&](int kt) {
                const int R0 = kt * a.rows_per_kt;                   // global tile row (image * tile rows + ty) of the K tile
                // pixel row 2 R0 - 1 + r, column -1 + c of the first tile: may lie before the tensor; never dereferenced there
                const float* xb = xsrc + ((long long)(2 * R0 - 1) * a.W - 1) * xc + ci0;
                const __amdgpu_buffer_rsrc_t r11 = brsrc(xb + (long long)(xr1 * a.W + xc1) * xc), r12 = brsrc(xb + (long long)(xr1 * a.W + xc2) * xc),
                                             r21 = brsrc(xb + (long long)(xr2 * a.W + xc1) * xc), r22 = brsrc(xb + (long long)(xr2 * a.W + xc2) * xc);
                const float* yb = a.dy + ((long long)(2 * R0) * a.W) * a.lddy + co0;
                const __amdgpu_buffer_rsrc_t q00 = brsrc(yb + (long long)(yr1 * a.W + yc1) * a.lddy), q01 = brsrc(yb + (long long)(yr1 * a.W + 1) * a.lddy),
                                             q10 = brsrc(yb + (long long)(a.W + yc1) * a.lddy), q11 = brsrc(yb + (long long)(a.W + 1) * a.lddy);
                return [=, &sx, &sy](auto is_x, int i) {
                    if constexpr (decltype(is_x)::value) {
                        const int ty = (R0 + xdr[i]) & a.th_mask;
                        const bool row1 = !(need_top && ty == 0), row2 = !(need_bot && ty == a.th_mask);
                        const bool c1 = (xcol[i] & 1u) != 0, c2 = (xcol[i] & 2u) != 0;
                        sx[i][0] = bload(r11, (row1 && c1) ? xoff[i] : 0xffffffffu);
                        sx[i][1] = bload(r12, (row1 && c2) ? xoff[i] : 0xffffffffu);
                        sx[i][2] = bload(r21, (row2 && c1) ? xoff[i] : 0xffffffffu);
                        sx[i][3] = bload(r22, (row2 && c2) ? xoff[i] : 0xffffffffu);
                    } else {
                        sy[i][0] = bload(q00, yoff[i]);
                        if constexpr (NC == 2) sy[i][1] = bload(q01, yoff[i]);
                        if constexpr (NR == 2) {
                            sy[i][NC] = bload(q10, yoff[i]);
                            if constexpr (NC == 2) sy[i][NC + 1] = bload(q11, yoff[i]);
                        }
                    }
                };
            };
            using TX = std::integral_constant<bool, true>;
            using TY = std::integral_constant<bool, false>;
            auto put_x = [&](unsigned char* img, int i) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = (sx[i][0][e] + xsc * sx[i][1][e]) + xsr * (sx[i][2][e] + xsc * sx[i][3][e]);
                ww_store<ABL>(img + 3 * WW_ALIMB + lds_x + i * 8 * WW_RSB, WW_BLIMB, v);
            };
            auto put_y = [&](unsigned char* img, int i) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = sy[i][0][e];
                    if constexpr (NC == 2) t += ysc * sy[i][1][e];
                    if constexpr (NR == 2) {
                        float u = sy[i][NC][e];
                        if constexpr (NC == 2) u += ysc * sy[i][NC + 1][e];
                        t += ysr * u;
                    }
                    v[e] = t;
                }
                ww_store<ABL>(img + lds_y + i * KSTEP * WW_RSA, WW_ALIMB, v);
            };
            {
                auto ld = load_tile(kt_beg);
#pragma unroll
                for (int i = 0; i < 4; ++i) ld(TX{}, i);
#pragma unroll
                for (int i = 0; i < NY; ++i) ld(TY{}, i);
            }
            for (int t = 0; t < nkt; ++t) {
                unsigned char* img = smem + (t & 1) * WW_IMG;
                if constexpr ((ABL & 64) != 0) {
                    if (t > 0) {
                        __syncthreads();
                        continue;
                    }
                }
                if (t + 1 < nkt && (ABL & 2) == 0) {
                    auto ld = load_tile(kt_beg + t + 1);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        put_x(img, i);
                        __builtin_amdgcn_sched_barrier(0);      // hipcc otherwise gathers the 48 loads at the end of the iteration
                        ld(TX{}, i);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int i = 0; i < NY; ++i) {
                        put_y(img, i);
                        __builtin_amdgcn_sched_barrier(0);
                        ld(TY{}, i);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) put_x(img, i);
#pragma unroll
                    for (int i = 0; i < NY; ++i) put_y(img, i);
                }
                __syncthreads();        // tile t is in its image; everyone has left the other image (tile t - 1)
            }
            __syncthreads();            // the consumers' last barrier
        };
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if (ynr == 2) {
            if (ync == 2) stage(I2{}, I2{}); else stage(I2{}, I1{});
        } else {
            if (ync == 2) stage(I1{}, I2{}); else stage(I1{}, I1{});
        }
        return;
    }

    // ---- consumers: 128 (c_out) x 64 (c_in) per wave ----------------------------------------------------------------------------
    const int wr = wave >> 1, wc = wave & 1;
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
    int a_base[2], b_base[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = 4 * g + 16 * j + q;              // tile (K slot) of this lane's row in read j
        a_base[j] = k * WW_RSA + (wr * (CO / 2) + 4 * p4) * 2;
        b_base[j] = 3 * WW_ALIMB + k * WW_RSB + (wc * 64 + 4 * p4) * 2;
    }
    auto frag = [&](const unsigned char* img, const int (&base)[2], int off) -> u32x4 {
        const u32x2 lo = lds_tr16(img + base[0] + off), hi = lds_tr16(img + base[1] + off);
        return u32x4{lo[0], lo[1], hi[0], hi[1]};
    };
    f32x4v acc[CB][4];
#pragma unroll
    for (int i = 0; i < CB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int i = 0; i < nkt; ++i) {
        const unsigned char* img = smem + (((ABL & 8) != 0 ? 0 : i) & 1) * WW_IMG;
        if constexpr ((ABL & 32) != 0) {
            __syncthreads();
            continue;
        }
        if constexpr ((ABL & 8) != 0) {
            if (i > 0) {            // fragments of the first tile, read once: the MFMAs below run on whatever the registers hold
                u32x4 f = frag(img, a_base, 0);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int u = 0; u < 6; ++u)
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb)
                            acc[cb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f), __builtin_bit_cast(bf16x8, f), acc[cb][nb], 0, 0, 0);
                __syncthreads();
                continue;
            }
        }
        u32x4 fb[4][3];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int l = 0; l < 3; ++l) fb[nb][l] = frag(img, b_base, l * WW_BLIMB + nb * 32);
        u32x4 fa[2][3];
#pragma unroll
        for (int l = 0; l < 3; ++l) fa[0][l] = frag(img, a_base, l * WW_ALIMB);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if (cb + 1 < CB) {
#pragma unroll
                for (int l = 0; l < 3; ++l) fa[(cb + 1) & 1][l] = frag(img, a_base, l * WW_ALIMB + (cb + 1) * 32);
            }
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            if constexpr ((ABL & 4) != 0) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int l = 0; l < 3; ++l)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[cb][nb][e] += __uint_as_float(fa[cb & 1][l][e] ^ fb[nb][l][e]);
            } else {
#pragma unroll
                for (int u = 0; u < 6; ++u)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        acc[cb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, fa[cb & 1][PA[u]]), __builtin_bit_cast(bf16x8, fb[nb][PB[u]]), acc[cb][nb], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // C/D layout of the 16x16 MFMA: col (ci) = lane & 15, row (co) = 4*(lane >> 4) + v
    float* S = a.slabs + ((long long)split * 16 + pos) * a.cout * a.cin_total;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int co = co0 + wr * (CO / 2) + cb * 16 + 4 * g + v;
                S[(long long)co * a.cin_total + ci_out + wc * 64 + nb * 16 + i16] = acc[cb][nb][v];
            }
}

// dg[co][ci][ky][kx] = sum_ij Gs[i][ky] Gs[j][kx] sum_split slabs[split][4 i + j][co][ci],  Gs = diag(1, 1, 1, -1) G
// (the -1: the dY-side transform of the positions with i = 3 / j = 3 was formed without A's sign).  One thread per (co, ci):
// 16 x nsplit coalesced reads, nine values written as one contiguous 36-byte run of the OIHW gradient.
__global__ void __launch_bounds__(64) wwgrad_reduce_kernel(const float* __restrict__ slabs, int nsplit, long long n, float* __restrict__ dw,
                                                           int accumulate, float alpha) {
    const long long idx = (long long)blockIdx.x * 64 + threadIdx.x;
    if (idx >= n) return;
    // splits in order (a fixed sum per position); the 16 positions of a split are 16 independent loads in flight per thread,
    // one-wave workgroups so that the ~1000 of them cover every CU several times (the first form - 256-thread blocks, one
    // position at a time - moved its 33 MB at 1.1 TB/s: 30 us per layer, 2.2 ms per step)
    float m[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) m[p] = 0.f;
    const float* src = slabs + idx;
    for (int sp = 0; sp < nsplit; ++sp) {
        float v[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) v[p] = __builtin_nontemporal_load(src + ((long long)sp * 16 + p) * n);
#pragma unroll
        for (int p = 0; p < 16; ++p) m[p] += v[p];
    }
    // t[i][kx] = sum_j Gs[j][kx] m[i][j]:  kx=0: m0 + (m1 + m2)/2;  kx=1: (m1 - m2)/2;  kx=2: (m1 + m2)/2 - m3
    float t[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float h = 0.5f * (m[4 * i + 1] + m[4 * i + 2]), d = 0.5f * (m[4 * i + 1] - m[4 * i + 2]);
        t[i][0] = m[4 * i] + h;
        t[i][1] = d;
        t[i][2] = h - m[4 * i + 3];
    }
    float o[9];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const float h = 0.5f * (t[1][kx] + t[2][kx]), d = 0.5f * (t[1][kx] - t[2][kx]);
        o[kx] = t[0][kx] + h;
        o[3 + kx] = d;
        o[6 + kx] = h - t[3][kx];
    }
    float* out = dw + idx * 9;
#pragma unroll
    for (int e = 0; e < 9; ++e) out[e] = accumulate ? out[e] + alpha * o[e] : alpha * o[e];
}

inline int ww_co_tile(int cout) { return cout % 256 ? 128 : 256; }     // c_out per workgroup

int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

extern "C" int psld_conv3x3_wgrad_wino_supported(int cout, int cin, int cin2, int batch, int h, int w) {
    if (cout <= 0 || cin <= 0 || cin2 < 0 || batch <= 0 || cout % 128 || cin % WW_CI || cin2 % WW_CI) return 0;
    if (h != w || !(w == 8 || w == 16 || w == 32 || w == 64)) return 0;
    return ((long long)batch * h * w / 4) % 32 == 0;
}

// K splits that fill the chip with one round of one-workgroup-per-CU tiles (0: shape not taken)
extern "C" int psld_conv3x3_wgrad_wino_nsplit(int cout, int cin_total, int batch, int h, int w) {
    if (cout % 128 || cin_total % WW_CI) return 0;
    const int units = 16 * (cout / ww_co_tile(cout)) * (cin_total / WW_CI);
    const long long ktiles = (long long)batch * h * w / 4 / 32;
    static int cus[PSLD_MAX_DEVICES] = {};
    int& n = cus[psld_device_slot()];
    if (n == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    int ns = n / units;
    if (ns < 1) ns = 1;
    if (ns > ktiles) ns = (int)ktiles;
    const long long per = (ktiles + ns - 1) / ns;
    return (int)((ktiles + per - 1) / per);       // every split non-empty
}

extern "C" long long psld_conv3x3_wgrad_wino_ws_bytes(int cout, int cin_total, int nsplit) {
    return (long long)nsplit * 16 * cout * cin_total * 4;
}

extern "C" int psld_conv3x3_wgrad_wino_f32(const float* dy, int lddy, int cout, const float* x, int cin, const float* x2, int cin2,
                                           int batch, int h, int w, float* slabs, int nsplit, float* dw_oihw, int accumulate,
                                           float alpha, hipStream_t stream) {
    PSLD_CHECK_ARG(dy && x && slabs && dw_oihw && nsplit >= 1 && (cin2 == 0 || x2), "psld_conv3x3_wgrad_wino_f32: bad args");
    PSLD_CHECK_ARG(psld_conv3x3_wgrad_wino_supported(cout, cin, cin2, batch, h, w),
                   "psld_conv3x3_wgrad_wino_f32: unsupported shape cout=%d cin=%d+%d B=%d %dx%d", cout, cin, cin2, batch, h, w);
    PSLD_CHECK_ARG(aligned16(dy) && aligned16(x) && (cin2 == 0 || aligned16(x2)) && lddy % 4 == 0 && lddy >= cout,
                   "psld_conv3x3_wgrad_wino_f32: unaligned operand");
    const long long ktiles = (long long)batch * h * w / 4 / 32;
    const long long per_split = (ktiles + nsplit - 1) / nsplit;
    PSLD_CHECK_ARG((ktiles + per_split - 1) / per_split == nsplit, "psld_conv3x3_wgrad_wino_f32: %d splits of %lld K tiles leave empty slabs", nsplit, ktiles);
    WWgradArgs a{};
    a.dy = dy; a.lddy = lddy; a.x = x; a.cin = cin; a.x2 = x2; a.cin2 = cin2;
    a.H = h; a.W = w;
    a.lg_tw = ilog2(w / 2); a.tw_mask = w / 2 - 1; a.th_mask = h / 2 - 1;
    a.rows_per_kt = 32 >> a.lg_tw;
    const int co_tile = ww_co_tile(cout);
    a.cout_tiles = cout / co_tile; a.cin_tiles = (cin + cin2) / WW_CI;
    a.ktiles = (int)ktiles; a.ktiles_per_split = (int)per_split;
    a.slabs = slabs; a.cout = cout; a.cin_total = cin + cin2;
    a.pos_override = -1;
    const dim3 grid((unsigned)(16 * a.cout_tiles * a.cin_tiles * nsplit));
#ifdef PSLD_ABLATIONS      // timing-only variants (wrong results): libpsld_hip_abl.so only (make -C tools/abl), never the product library
    static const int abl = [] { const char* v = getenv("PSLD_WWGRAD_ABL"); return v ? atoi(v) : 0; }();
    static const int pos_ov = [] { const char* v = getenv("PSLD_WWGRAD_POS"); return v ? atoi(v) : -1; }();
    a.pos_override = pos_ov;
#define WW_ABL_CASE(N)                                                                                                              \
    case N:                                                                                                                         \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wwgrad_ws_kernel<N, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WWGeom<256>::LDS); \
        hipLaunchKernelGGL((wwgrad_ws_kernel<N, 256>), grid, dim3(512), WWGeom<256>::LDS, stream, a);                                \
        PSLD_CHECK_LAUNCH("wwgrad_ws_kernel (ablation)");                                                                           \
        launched = true;                                                                                                            \
        break;
    bool launched = false;
    switch (co_tile == 256 ? abl : 0) {
        WW_ABL_CASE(1) WW_ABL_CASE(2) WW_ABL_CASE(3) WW_ABL_CASE(4) WW_ABL_CASE(8) WW_ABL_CASE(12) WW_ABL_CASE(16) WW_ABL_CASE(32) WW_ABL_CASE(64)
        default: break;
    }
#undef WW_ABL_CASE
    if (!launched)
#endif
    {
    static PsldPerDeviceFlag configured_; bool& configured = configured_.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wwgrad_ws_kernel<0, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WWGeom<256>::LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wwgrad_ws_kernel<0, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WWGeom<128>::LDS);
        if (e != hipSuccess) {
            psld_set_error("psld_conv3x3_wgrad_wino_f32: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return PSLD_ERR_LAUNCH;
        }
        configured = true;
    }
    if (co_tile == 256)
        hipLaunchKernelGGL((wwgrad_ws_kernel<0, 256>), grid, dim3(512), WWGeom<256>::LDS, stream, a);
    else
        hipLaunchKernelGGL((wwgrad_ws_kernel<0, 128>), grid, dim3(512), WWGeom<128>::LDS, stream, a);
    PSLD_CHECK_LAUNCH("wwgrad_ws_kernel");
    }
    const long long n = (long long)cout * a.cin_total;
    hipLaunchKernelGGL(wwgrad_reduce_kernel, dim3((unsigned)cdiv(n, 64)), dim3(64), 0, stream, slabs, nsplit, n, dw_oihw, accumulate, alpha);
    PSLD_CHECK_LAUNCH("wwgrad_reduce_kernel");
    return PSLD_OK;
}
