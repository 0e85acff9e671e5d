// Fused single-head spatial self-attention, forward: O = softmax(Q K^T * scale) V in ONE kernel, the HW x HW score
// matrix never leaves the CU (registers: the probabilities of a wave's 16 queries stay in the accumulator layout and feed
// the second contraction directly).  Replaces, for the forward pass, the two batched limb GEMMs and the softmax pass
// between them (AttnBlockpp.forward: einsum -> softmax -> einsum, song_sde/layerspp.py:82-86); same arithmetic as those
// kernels: fp32 operands split exactly into three bf16 limbs, six limb products per product on v_mfma_f32_16x16x32_bf16,
// fp32 accumulation, expf / fp32 softmax.
//
// One workgroup = one image x 128 queries (8 waves, 16x16 maps) or x 64 queries (4 waves, 8x8 maps); wave w owns queries
// 16w .. 16w+15 against ALL keys (HW = 256 or 64).  The global loads of the next 32-channel chunk / 32-key step fly under
// the MFMAs of the current one (round 3's kernel loaded, waited, staged, multiplied - one memory latency per chunk - and
// kept its probabilities in scratch: the key-step loop was not unrolled, acc_s[2 ks] a dynamic index).
//   phase 1  S^T[key][q] = sum_c K[key][c] Q[q][c]: per 32-channel chunk the K rows (HW x 32) and Q rows (64 x 32) are split
//            into limbs and staged as [limb][row][64 B] images (the direct convolution's layout and swizzle); the MFMAs
//            take the K fragment as the first operand, so a lane ends up with the scores of ONE query (column l & 15)
//            against 4 consecutive keys per 16-key block: the softmax row reduction is in-lane plus two cross-lane steps.
//   phase 2  p = softmax(scale * s) in registers; optionally written out ([B][HW][HW], what the backward reads).
//   phase 3  O^T[c][q] = sum_key V[key][c] P[q][key]: the MFMA's k slots of a 32-key step are ASSIGNED to the keys a lane
//            already holds (slots 0-3 = keys 16 kb0 + 4g .. +3, slots 4-7 = 16 kb1 + 4g .. +3), so the P fragments are just
//            the split accumulator values - no exchange - and the V fragments (first operand, rows = channels) come from a
//            [limb][key][channel] image through two ds_read_b64_tr_b16 with exactly that row assignment.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "psld_hip.h"
#include "tile_shared.h"
#include "limb.h"

namespace {

struct AttnArgs {
    const float* q;
    const float* k;
    const float* v;
    int ld;                 // row stride of q / k / v (3C for the fused q|k|v buffer)
    float scale;
    float* o;
    int ldo;
    float* p;               // [B][HW][HW] or null
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ u32x2 lds_tr16(const unsigned char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) void*)(p));
    return __builtin_bit_cast(u32x2, v);
}

template <int HW, int C, int QW>
__global__ void __launch_bounds__(64 * QW, 2) attn_fwd_kernel(const AttnArgs a) {
    constexpr int NT = 64 * QW;                 // threads: QW waves, wave w owns queries 16 w .. 16 w + 15 of the tile
    constexpr int QR = 16 * QW;                 // queries per workgroup
    constexpr int KB = HW / 16;                 // 16-key blocks
    constexpr int CB = C / 16;                  // 16-channel blocks of the output
    constexpr int KROWS = HW * ROWB;            // bytes of one limb of the K image
    constexpr int QROWS = QR * ROWB;
    constexpr int VRS = C * 2 + 32;             // V image: bytes per key row and limb (544 for C = 256: rows 8 banks apart)
    constexpr int VLIMB = 32 * VRS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // max(3 (KROWS + QROWS), 3 VLIMB)
    unsigned char* Ks = smem;
    unsigned char* Qs = smem + 3 * KROWS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    constexpr int QTILES = HW / QR;
    const int b = blockIdx.x / QTILES, qt = blockIdx.x - b * QTILES;
    const long long row0 = (long long)b * HW;               // first row of this image in q / k / v / o
    const int q0 = qt * QR;

    // staging items (float4): K rows x 8 channel quads and Q rows x 8 quads per 32-channel chunk, 32 key rows x C / 4 quads per
    // 32-key step of V.  The global loads of chunk / step i + 1 are issued right behind the barrier that publishes the images
    // of i and land while the MFMAs of i run (they are consumed behind the next barrier); V's first step is asked for during
    // the last chunk of the scores.
    constexpr int KI = HW * 8 / NT, QI = QR * 8 / NT, VI = 32 * (C / 4) / NT, QPR = C / 4, RSTEP = NT / 8;
    const int c4 = tid & 7, srow = tid >> 3;    // item i: row srow + RSTEP i, quad c4
    f32x4 kv[KI], qv[QI], vv[VI];
    auto load_kq = [&](int ch) {
#pragma unroll
        for (int i = 0; i < KI; ++i) kv[i] = ld4(a.k + (row0 + srow + RSTEP * i) * a.ld + ch * 32 + c4 * 4);
#pragma unroll
        for (int i = 0; i < QI; ++i) qv[i] = ld4(a.q + (row0 + q0 + srow + RSTEP * i) * a.ld + ch * 32 + c4 * 4);
    };
    auto load_v = [&](int ks) {
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            const int it = tid + NT * i;
            const int row = it / QPR, qd = it - row * QPR;
            vv[i] = ld4(a.v + (row0 + ks * 32 + row) * a.ld + qd * 4);
        }
    };
    auto put = [&](unsigned char* img, int limb_bytes, int row, const f32x4& v) {
        unsigned h0, m0, l0, h1, m1, l1;
        split3(v[0], v[1], h0, m0, l0);
        split3(v[2], v[3], h1, m1, l1);
        unsigned char* d = img + row * ROWB + (((c4 >> 1) ^ lds_swz(row)) << 4) + (c4 & 1) * 8;
        *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(d + limb_bytes) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(d + 2 * limb_bytes) = u32x2{l0, l1};
    };

    // ---- phase 1: scores ------------------------------------------------------------------------------------------------
    f32x4v acc_s[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) acc_s[kb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    load_kq(0);
    for (int ch = 0; ch < C / 32; ++ch) {
        if (ch) __syncthreads();                // the previous chunk's fragments have been read
#pragma unroll
        for (int i = 0; i < KI; ++i) put(Ks, KROWS, srow + RSTEP * i, kv[i]);
#pragma unroll
        for (int i = 0; i < QI; ++i) put(Qs, QROWS, srow + RSTEP * i, qv[i]);
        __syncthreads();
        if (ch + 1 < C / 32)
            load_kq(ch + 1);
        else
            load_v(0);
        u32x4 fq[3];
        {
            const int row = wave * 16 + r16;
            const unsigned char* p = Qs + row * ROWB + ((g ^ lds_swz(row)) << 4);
#pragma unroll
            for (int l = 0; l < 3; ++l) fq[l] = *reinterpret_cast<const u32x4*>(p + l * QROWS);
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int row = kb * 16 + r16;
            const unsigned char* p = Ks + row * ROWB + ((g ^ lds_swz(row)) << 4);
            u32x4 fk[3];
#pragma unroll
            for (int l = 0; l < 3; ++l) fk[l] = *reinterpret_cast<const u32x4*>(p + l * KROWS);
            // limb products, smallest first: [first operand K, second operand Q] limbs (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
                acc_s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fk[PA[t]]),
                                                                    __builtin_bit_cast(bf16x8, fq[PB[t]]), acc_s[kb], 0, 0, 0);
        }
    }

    // ---- phase 2: softmax over the keys of this lane's query (column r16): 4 KB values here, the rest in lanes r16 + 16 g' ----
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            acc_s[kb][v] *= a.scale;
            mx = fmaxf(mx, acc_s[kb][v]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            acc_s[kb][v] = expf(acc_s[kb][v] - mx);
            sum += acc_s[kb][v];
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    const long long qrow = row0 + q0 + wave * 16 + r16;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        acc_s[kb] *= inv;
        if (a.p) *reinterpret_cast<f32x4v*>(a.p + qrow * HW + kb * 16 + 4 * g) = acc_s[kb];
    }

    // ---- phase 3: O^T = V^T P^T over 32-key steps ---------------------------------------------------------------------------
    f32x4v acc_o[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc_o[cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // transposed-read roles: lane 4q + p of a 16-lane group supplies row q, channels 4p..4p+3 (conv_split.hip: dwgrad_kernel)
    const int tq = r16 >> 2, tp = r16 & 3;
    const int vbase0 = (4 * g + tq) * VRS + tp * 8, vbase1 = (16 + 4 * g + tq) * VRS + tp * 8;
#pragma unroll                                  // (acc_s[2 ks] must be a register, not a scratch slot)
    for (int ks = 0; ks < HW / 32; ++ks) {
        __syncthreads();                        // phase 1's images / the previous step's V image have been read
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            const int it = tid + NT * i;
            const int row = it / QPR, qd = it - row * QPR;
            unsigned h0, m0, l0, h1, m1, l1;
            split3(vv[i][0], vv[i][1], h0, m0, l0);
            split3(vv[i][2], vv[i][3], h1, m1, l1);
            unsigned char* d = smem + row * VRS + qd * 8;
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + VLIMB) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(d + 2 * VLIMB) = u32x2{l0, l1};
        }
        __syncthreads();
        if (ks + 1 < HW / 32) load_v(ks + 1);
        // P fragments of this step: k slots 0-3 = block 2 ks, slots 4-7 = block 2 ks + 1 (the keys this lane holds)
        u32x4 fp[3];
        {
            unsigned h[4], m[4], l[4];
            split3(acc_s[2 * ks][0], acc_s[2 * ks][1], h[0], m[0], l[0]);
            split3(acc_s[2 * ks][2], acc_s[2 * ks][3], h[1], m[1], l[1]);
            split3(acc_s[2 * ks + 1][0], acc_s[2 * ks + 1][1], h[2], m[2], l[2]);
            split3(acc_s[2 * ks + 1][2], acc_s[2 * ks + 1][3], h[3], m[3], l[3]);
            fp[0] = u32x4{h[0], h[1], h[2], h[3]};
            fp[1] = u32x4{m[0], m[1], m[2], m[3]};
            fp[2] = u32x4{l[0], l[1], l[2], l[3]};
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            u32x4 fv[3];
#pragma unroll
            for (int lm = 0; lm < 3; ++lm) {
                const u32x2 lo = lds_tr16(smem + lm * VLIMB + vbase0 + cb * 32), hi = lds_tr16(smem + lm * VLIMB + vbase1 + cb * 32);
                fv[lm] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
                acc_o[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fv[PA[t]]),
                                                                    __builtin_bit_cast(bf16x8, fp[PB[t]]), acc_o[cb], 0, 0, 0);
        }
    }
    // rows of the accumulator = channels cb*16 + 4g + v, column = query r16: 16-byte stores
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) *reinterpret_cast<f32x4v*>(a.o + qrow * a.ldo + cb * 16 + 4 * g) = acc_o[cb];
}

template <int HW, int C, int QW>
int launch_attn(const AttnArgs& a, int batch, hipStream_t stream) {
    constexpr size_t PH1 = (size_t)3 * (HW + 16 * QW) * ROWB, PH3 = (size_t)3 * 32 * (C * 2 + 32), LDS = PH1 > PH3 ? PH1 : PH3;
    static PsldPerDeviceFlag configured_; bool& configured = configured_.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<HW, C, QW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) {
            psld_set_error("psld_attn_fwd_split_f32: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return PSLD_ERR_LAUNCH;
        }
        configured = true;
    }
    hipLaunchKernelGGL((attn_fwd_kernel<HW, C, QW>), dim3((unsigned)(batch * (HW / (16 * QW)))), dim3(64 * QW), LDS, stream, a);
    PSLD_CHECK_LAUNCH("psld_attn_fwd_split_f32");
    return PSLD_OK;
}

}  // namespace

extern "C" int psld_attn_fwd_split_supported(int hw, int c) {
    return (hw == 256 || hw == 64) && (c == 256 || c == 128);
}

extern "C" int psld_attn_fwd_split_f32(const float* q, const float* k, const float* v, int ld, int batch, int hw, int c,
                                       float scale, float* o, int ldo, float* p, hipStream_t stream) {
    PSLD_CHECK_ARG(q && k && v && o && batch > 0, "psld_attn_fwd_split_f32: null pointer / empty batch");
    PSLD_CHECK_ARG(psld_attn_fwd_split_supported(hw, c), "psld_attn_fwd_split_f32: unsupported hw=%d c=%d", hw, c);
    PSLD_CHECK_ARG(ld % 4 == 0 && ldo % 4 == 0 && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o) && (!p || aligned16(p)),
                   "psld_attn_fwd_split_f32: 16-byte aligned rows needed");
    AttnArgs a{q, k, v, ld, scale, o, ldo, p};
    // 16x16 maps: 128 queries per workgroup, eight waves (K / V are split into limbs once per 128 queries); 8x8 maps: the
    // whole image (64 queries), four waves
#ifdef PSLD_ABLATIONS
    static const int qw4 = [] { const char* v = getenv("PSLD_ATTN_QW4"); return v ? atoi(v) : 0; }();
    if (qw4 && hw == 256 && c == 256) return launch_attn<256, 256, 4>(a, batch, stream);
#endif
    if (hw == 256) return c == 256 ? launch_attn<256, 256, 8>(a, batch, stream) : launch_attn<256, 128, 8>(a, batch, stream);
    return c == 256 ? launch_attn<64, 256, 4>(a, batch, stream) : launch_attn<64, 128, 4>(a, batch, stream);
}
