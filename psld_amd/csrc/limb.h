// Pieces shared by the bf16 limb MFMA kernels (conv_split.hip: direct 3x3 / pointwise / weight gradient;
// conv_wino.hip: Winograd F(2x2, 3x3) forward / data gradient).
#pragma once
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int ROWB = 64;        // bytes per pixel and limb in the LDS image: 32 bf16, no padding.  The 16-byte slot s
                                // (8 channels) of pixel row p sits at slot s ^ lds_swz(p): with it the ds_read_b128 of
                                // the 16x16x32 A operand (16 consecutive rows x 4 slots per wave) is bank-conflict
                                // free for every row offset, i.e. for all nine taps (a linear padded image is 2-way)
__device__ __forceinline__ int lds_swz(int p) { return (p >> 1) & 2; }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Exact three-limb decomposition of two fp32 values into packed bf16 pairs (x0 in the low half):
// hi = rne_bf16(x), mid = rne_bf16(x - hi), lo = x - hi - mid (at most 8 significant bits left: exact).
// v_cvt_pk_bf16_f32 as an opaque (pure) instruction: written with bf16 vector casts, hipcc may re-derive "hi << 16" from a
// second, single-value conversion (5 conversions per pair instead of 3 - seen under -fno-slp-vectorize).
__device__ __forceinline__ unsigned cvt_pk_bf16(float x0, float x1) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x0), "v"(x1));
    return r;
}
__device__ __forceinline__ void split3(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    mid = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
    lo = cvt_pk_bf16(s0, s1);
}

typedef float f32x4v __attribute__((ext_vector_type(4)));


}  // namespace
