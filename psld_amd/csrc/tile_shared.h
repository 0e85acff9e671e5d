// Pieces shared by the MFMA kernels of tile_engine.hip (fp32 MFMA) and conv_split.hip (bf16 limb MFMA).
#pragma once
#include "common.h"
#include "psld_hip.h"

// fused epilogue (include/psld_hip.h: psld_epilogue_t) as the kernels take it
struct PsldEpilogue {
    float alpha;
    const float* bias;      // [N] or null
    const float* rowbias;   // [M/rows_per_img][ld_rowbias] or null (time-embedding bias)
    int ld_rowbias;
    int rows_per_img;
    const float* res;       // residual [M][ldres] or null
    int ldres;
    long long res_stride_z;
    float out_scale;
    int accumulate;         // C += result
    double* gn_part;        // limb kernels only: GroupNorm partial sums of the output (psld_epilogue_t.gn_part)
    int gn_hw;
    int gn_fine;            // channels per partial sum: 8, or 4 (psld_epilogue_t.gn_fine)
};

static inline PsldEpilogue make_epilogue(const psld_epilogue_t* e) {
    PsldEpilogue o;
    o.alpha = 1.f; o.bias = nullptr; o.rowbias = nullptr; o.ld_rowbias = 0; o.rows_per_img = 1;
    o.res = nullptr; o.ldres = 0; o.res_stride_z = 0; o.out_scale = 1.f; o.accumulate = 0;
    o.gn_part = nullptr; o.gn_hw = 0; o.gn_fine = 8;
    if (e) {
        o.alpha = e->alpha; o.bias = e->bias; o.rowbias = e->rowbias; o.ld_rowbias = e->ld_rowbias;
        o.rows_per_img = e->rows_per_img > 0 ? e->rows_per_img : 1;
        o.res = e->residual; o.ldres = e->ld_residual; o.res_stride_z = e->residual_stride_batch;
        o.out_scale = e->out_scale; o.accumulate = e->accumulate;
        o.gn_part = e->gn_part; o.gn_hw = e->gn_hw; o.gn_fine = e->gn_fine == 4 ? 4 : 8;
    }
    return o;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// 16 bytes of zeros in device memory (out-of-range lanes load from it instead of branching); null on failure
const float* psld_detail_zero_page(const char* name);
// out = epilogue(sum_s slabs[s][M][N]) (float4 along N)
int psld_detail_conv_reduce_epilogue(const float* slabs, int nsplit, int M, int N, float* y, int ldy,
                                     const PsldEpilogue& e, hipStream_t stream);
// ... which also forms the GroupNorm partial sums of the output (e.gn_part) when this holds
bool psld_detail_conv_reduce_gn_ok(int M, int N, const PsldEpilogue& e);

// XCD-aware block remap: workgroups are dealt round-robin over the 8 XCDs (private L2 each); give every XCD a
// contiguous run of logical tiles.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
}

// Store one 32x32 MFMA accumulator block (C/D layout of v_mfma_f32_32x32x*: col = lane & 31,
// row = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)) through the fused epilogue.
__device__ __forceinline__ void epilogue_store_block(const f32x16& acc, int row_base, int gn, int h, int M, int N,
                                                     float* Cb, int ldc, const float* Rb, const PsldEpilogue& e) {
    if (gn >= N) return;
    const bool rb_uniform = e.rowbias && (e.rows_per_img % 32 == 0);
    float bias = e.bias ? e.bias[gn] : 0.f;
    // time-embedding bias: one value per (image, channel); a 32-row block never straddles images
    if (rb_uniform && row_base < M) bias += e.rowbias[(long long)(row_base / e.rows_per_img) * e.ld_rowbias + gn];
    // residual / previous-output values first, all 16 in flight: a load issued after a store the compiler cannot tell
    // apart from it waits out its own latency (see dconv_epilogue in conv_split.hip)
    float rv[16], cv[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int gm = row_base + (v & 3) + 8 * (v >> 2) + 4 * h;
        const bool ok = gm < M;
        rv[v] = (Rb && ok) ? Rb[(long long)gm * e.ldres + gn] : 0.f;
        cv[v] = (e.accumulate && ok) ? Cb[(long long)gm * ldc + gn] : 0.f;
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int gm = row_base + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (gm >= M) continue;
        float x = acc[v] * e.alpha + bias;
        if (e.rowbias && !rb_uniform) x += e.rowbias[(long long)(gm / e.rows_per_img) * e.ld_rowbias + gn];
        if (Rb) x += rv[v];
        x *= e.out_scale;
        if (e.accumulate) x += cv[v];
        Cb[(long long)gm * ldc + gn] = x;
    }
}
