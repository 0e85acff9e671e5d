// Winograd F(2x2, 3x3) form of the 3x3 stride-1 pad-1 convolutions (forward and data gradient) on the bf16 limb MFMA
// path of conv_split.hip: 2.25x fewer matrix instructions for the same fp32-equivalent result.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 4x4 input tile, g: 3x3 filter, Y: 2x2 output tile (Lavin & Gray)
//
// It is a re-association of the same fp32 arithmetic, not a narrower one: the transformed operands V = B^T d B and
// U = G g G^T are formed in fp32, each is decomposed EXACTLY into three bf16 limbs (limb.h: split3) and every product
// is accumulated in fp32 from the six limb products of weight >= 2^-16, as in the direct kernels.  What the reference's
// GPU path may run for exactly these layers (nn.Conv2d 3x3 fp32 -> cuDNN Winograd; song_sde/layers.py:103-109).
//
// Shape of the kernel (one 512-thread workgroup per CU, two waves per SIMD):
//   * a workgroup owns 32 tiles (= 128 consecutive output pixels, the direct kernel's tile) x 128 output channels for
//     ALL 16 transform positions; wave w owns the 16 channels n0 + 16w.. for the 32 tiles: 16 positions x 2 tile blocks
//     of v_mfma_f32_16x16x32_bf16 accumulators = 128 VGPRs, so the output transform A^T m A is register arithmetic in
//     the lane that holds (tile, 4 channels) - no exchange between waves;
//   * per 32-channel chunk the raw fp32 halo tile ((rows+2) x (W+2) pixels, <= 36 KB) goes global -> registers (loaded
//     a whole MFMA phase ahead) -> LDS; every thread then transforms (tile, 4 channels, 8 of the 16 positions):
//     2 adds per V element, split3, and writes the limb image V[pos][limb][tile][32 ch] (96 KB) - the split is paid
//     once per V element and workgroup;
//   * U comes pre-transformed and pre-split in MFMA operand order (psld_pack_conv3x3_wino, once per optimizer step,
//     16/9 the bytes of the direct fragments) straight from L2, each fragment loaded by exactly one wave;
//   * per chunk and wave: 16 positions x (6 ds_read_b128 + 3 global loads + 12 MFMAs).
// Epilogue = the direct kernels' (bias / time-embedding row bias / residual / scale / accumulate / GroupNorm partial
// sums of the output), applied to the four output pixels of a tile.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "psld_hip.h"
#include "tile_shared.h"
#include "limb.h"

namespace {

constexpr int WT = 32;                    // tiles per workgroup
constexpr int VPLANE = WT * ROWB;         // bytes of one (position, limb) plane of the V image: 32 tiles x 64 B
constexpr int VBYTES = 16 * 3 * VPLANE;   // 98,304
constexpr int WINO_THREADS = 512;

struct WinoArgs {
    const float* x1;
    const float* x2;
    int C1, C2;
    int B, H, W;
    const u32x4* ufrag;     // [N/128][8 waves][chunk][16 pos][3 limbs][64 lanes]
    int N, M;               // cout (multiple of 128), B*H*W
    int chunks;             // (C1 + C2) / 32
    float* C;
    int ldc;
    int nseg, rps;          // image segments per 128-pixel tile and output rows per segment (dconv_geometry, mt = 128)
    int cw;                 // wino_conv8s_kernel: width of a workgroup's pixel block (W, or 32 on 64-wide maps: rps = 4)
    PsldEpilogue e;
    const float* zero;
    int nmajor;
    int tiles_m;            // wino_conv8p_kernel: pixel tiles of the launch (cdiv(M, 128))
    int stagger;
    // wino_conv8s_kernel<., GNF = true>: GroupNorm apply (+ SiLU) of the input inside the raw staging - per-(image, channel)
    // scale / shift rows of source 1 and 2 ([B][C1], [B][C2]: psld_gn_stats_*), act = 1: SiLU
    const float *gsc1, *gsh1, *gsc2, *gsh2;
    int gn_act;
    unsigned long long* dbg;    // ablation library: s_memtime stamps of one chunk (ABL & 64), [workgroup][wave][8]
    int lg_tiles_x, lg_tps; // wino_conv8p_kernel: log2 of the tiles per tile row (cw / 2) and per image segment
    // wino_conv8s_kernel, small grids (round 6): the channel chunks split over ksplit workgroups per tile, each writing its plain
    // partial output to C + split * slab_stride (no epilogue); psld_detail_conv_reduce_epilogue sums them.  ksplit = 0 / 1: off
    int ksplit;
    long long slab_stride;
};

// raw halo image: pixel hp, 16-byte slot q (4 channels) -> byte offset.  Pixels sit pairwise in 256-byte rows and the
// half of the row a pixel takes alternates with the pair index: the transform phase reads pixels 2 apart (tiles are two
// pixels wide), which a linear [pixel][128 B] image would put on half of the banks (2-way conflict on ds_read_b128).
__device__ __forceinline__ int raw_off(int hp, int q) {
    return (hp >> 1) * 256 + (((((hp & 1) ^ ((hp >> 1) & 1)) << 3) + q) << 4);
}

// ---- weights -> transformed limb fragments -----------------------------------------------------------------
// One work item = lane slot (nt, wave, chunk, lane): n = nt*128 + wave*16 + (lane & 15), k = chunk*32 + (lane >> 4)*8 + j.
// dgrad = 0: g = w[co = n][ci = k][:, :]; dgrad = 1: g = w[co = k][ci = n] rotated by 180 degrees (conv_split.hip).
__device__ __forceinline__ void wino_pack_item(const float* __restrict__ w, u32x4* __restrict__ out, long long it, int k_in,
                                               long long sn, long long sk, int flip) {
    const int chunks = k_in / 32;
    long long t = it;
    const int lane = (int)(t & 63); t >>= 6;
    const int chunk = (int)(t % chunks); t /= chunks;
    const int nblk = (int)t;                            // nt*8 + wave
    const int n = nblk * 16 + (lane & 15);
    const int k0 = chunk * 32 + (lane >> 4) * 8;
    float U[8][16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float* p = w + n * sn + (k0 + j) * sk;
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = flip ? p[8 - (a * 3 + b)] : p[a * 3 + b];
        float t4[4][3];                                 // G g
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t4[0][b] = g[0][b];
            t4[1][b] = 0.5f * ((g[0][b] + g[2][b]) + g[1][b]);
            t4[2][b] = 0.5f * ((g[0][b] + g[2][b]) - g[1][b]);
            t4[3][b] = g[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {                   // (G g) G^T
            U[j][a * 4 + 0] = t4[a][0];
            U[j][a * 4 + 1] = 0.5f * ((t4[a][0] + t4[a][2]) + t4[a][1]);
            U[j][a * 4 + 2] = 0.5f * ((t4[a][0] + t4[a][2]) - t4[a][1]);
            U[j][a * 4 + 3] = t4[a][2];
        }
    }
    u32x4* o = out + ((long long)nblk * chunks + chunk) * (16 * 3 * 64) + lane;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        unsigned hi[4], mid[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split3(U[2 * j][p], U[2 * j + 1][p], hi[j], mid[j], lo[j]);
        o[(p * 3 + 0) * 64] = u32x4{hi[0], hi[1], hi[2], hi[3]};
        o[(p * 3 + 1) * 64] = u32x4{mid[0], mid[1], mid[2], mid[3]};
        o[(p * 3 + 2) * 64] = u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
}

__global__ void wino_pack_kernel(const float* __restrict__ w, u32x4* __restrict__ out, int n_out, int k_in,
                                 long long sn, long long sk, int flip) {
    const long long items = (long long)(n_out / 16) * (k_in / 32) * 64;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x)
        wino_pack_item(w, out, it, k_in, sn, sk, flip);
}

// Many weight tensors in one launch (as pack_frag_batch_kernel of conv_split.hip).  tab[8*i ..]: src pointer, dst
// pointer, n_out, k_in, flip, sn, sk, first work item of tensor i (a tensor has n_out * k_in / 8 items).
__global__ void wino_pack_batch_kernel(const long long* __restrict__ tab, int ntab, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = ntab - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tab[8 * mid + 7] <= idx) lo = mid; else hi = mid - 1;
        }
        const long long* d = tab + 8 * lo;
        wino_pack_item(reinterpret_cast<const float*>(d[0]), reinterpret_cast<u32x4*>(d[1]), idx - d[7], (int)d[3], d[5], d[6],
                       (int)d[4]);
    }
}

// ---- eight waves, staggered roles: the transform of one SIMD partner runs under the MFMAs of the other ----------------
// Same tile, same wave tile as wino_conv_kernel (wave w: 32 tiles x 16 channels x 16 positions, 128 accumulator VGPRs,
// two waves per SIMD) and the half-image schedule of wino_conv4_kernel (HP0(c): MFMAs on V rows 0,1 of chunk c, V rows 2,3
// of chunk c being produced; HP1(c): MFMAs on rows 2,3, rows 0,1 of chunk c + 1 being produced; one barrier after each).
// Inside a half-phase every wave has two blocks of work - its share of the transform (one V row of one (tile, channel
// quad) item: 8 ds_read_b128, 32 adds, 8 split3 pairs, 12 ds_write_b64) and 8 positions x 12 MFMAs - and the two waves
// that share a SIMD (w and w + 4) run them in OPPOSITE order: while one issues vector ALU work the other owns the matrix
// pipe, with no instruction-level interleaving to get right (the hardware arbitrates between the two waves) and the
// partner's MFMAs covering each wave's memory latencies, which one wave per SIMD has to cover by itself.
// (Output tile through LDS so that every wave stores whole 512-byte pixel rows instead of 64-byte runs: built, correct,
// 2.5 % slower - 454 vs 443 us - although skipping the stores altogether saves 8 %: the exposed part of the epilogue is the
// drain of 64 KB per workgroup, not the segment size.)
// (Two persistent forms of this kernel were built and measured slower: workgroups walking (pixel tile, channel tile) work
// lists, 480 vs 454 us on 256->256 @32x32 B=128, and one workgroup per pixel tile looping over the channel tiles with the
// half-phase pipeline running across the passes, 522 vs 451 us - hipcc's code for the accumulators degrades once the
// epilogue sits inside a loop.  One workgroup per (pixel tile, channel tile) it is.)
// GNF: the input is the RAW tensor a GroupNorm (+ SiLU) is to be applied to, and the apply pass runs here, on the float4 a
// thread has just loaded for the raw image: a = silu(x * scale[img][c] + shift[img][c]), zero outside the image (the
// convolution pads the ACTIVATED tensor).  The inference forward (EM / SSCS sampling) needs the activated tensor nowhere
// else, so psld_gn_apply_nhwc_f32's round trip through HBM disappears; one image per region only (maps of >= 128 pixels).
// The scale / shift quad is loaded where it is used (L1 / L2 hits; holding it in registers through the MFMA phase, or
// staging it through LDS two half-phases ahead, spills: the kernel sits at 256 VGPRs).  Measured (profiles/r04/
// wino_fused_gn.txt): the launch gets 7-10 % longer - ~110 vector instructions per thread and chunk, two of them
// transcendental per element, in a kernel that is short of vector issue slots - against the apply pass it replaces: 3-5 %
// less time for the pair on the 32x32 level at B=512, nothing on 16x16.  Reference: GroupNorm_0/1 + act in front of
// Conv_0/1, layerspp.py:245-263.
// ERAW: the raw halo of chunk c + 1 is requested right after the MFMAs of HP1(c - 1) instead of at the head of HP0(c).
// Vector-memory operations retire in issue order (one vmcnt): at the head of HP0 the four halo loads (HBM latency) sit in
// front of every weight-fragment load of the half-phase, and the first s_waitcnt of the MFMA stream waits for them.
// Issued behind the last fragment wait of the previous half-phase they have that phase's transform and the barrier to
// land before anything younger is waited for.
template <int ABL = 0, bool GNF = false, bool ERAW = false, int LA = 2>
__global__ void __launch_bounds__(WINO_THREADS) wino_conv8s_kernel(const WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NI = 4;                     // raw image: 256 halo pixels
    constexpr int RAWB = NI * 64 * 128;
    unsigned char* Vs = smem;
    unsigned char* Rs = smem + VBYTES;        // two raw images

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = a.N >> 7;
    const int nsp = a.ksplit > 1 ? a.ksplit : 1;
    const int per_range = (int)gridDim.x / nsp;                  // workgroups of one channel-chunk range
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int split = vb / per_range;
    const int bid = vb - split * per_range;
    const int kch = a.chunks / nsp, ch0 = split * kch;           // this workgroup's chunks [ch0, ch0 + kch)
    float* const Cw = a.C + split * a.slab_stride;
    const int tiles_m = per_range / tiles_n;
    const int tile_n = a.nmajor ? bid / tiles_m : bid % tiles_n;
    const int tile_m = a.nmajor ? bid - tile_n * tiles_m : bid / tiles_n;
    const int n0 = tile_n * 128;

    // Region of this workgroup: nseg images (maps smaller than 128 pixels) or one rps x cw block of an image - cw = W, or 32
    // for 64-wide maps, whose full-width tile (2 rows x 64) would need a 264-pixel halo where the two raw images hold 256.
    const int HW = a.H * a.W;
    const int W2 = a.cw + 2;
    const int tiles_x = a.cw >> 1;
    const int tps = (a.rps >> 1) * tiles_x;
    const int rpi = HW >= 128 ? HW >> 7 : 1;                    // regions per image
    const int xblocks = a.W / a.cw;
    const int reg = tile_m % rpi;
    const int img0 = HW >= 128 ? tile_m / rpi : tile_m * a.nseg;
    const int oy0 = (reg / xblocks) * a.rps, ox0 = (reg % xblocks) * a.cw;
    const float* zp = a.zero;

    const int c4 = tid & 7;
    int hoff[NI], rdst[NI];
    {
        const int seg_px = (a.rps + 2) * W2;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int px = (tid + WINO_THREADS * i) >> 3;
            const int seg = px / seg_px;
            const int rem = px - seg * seg_px;
            const int hr = rem / W2, hx = rem - hr * W2;
            const int img = img0 + seg, iy = oy0 + hr - 1, ix = ox0 + hx - 1;
            const bool ok = seg < a.nseg && img < a.B && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            hoff[i] = ok ? (img * a.H + iy) * a.W + ix : -1;
            rdst[i] = raw_off(px, c4);
        }
    }
    f32x4 hv[NI];
    auto load_raw = [&](int c) {
        const int c0 = c * 32;
        const bool second = c0 >= a.C1;
        const float* src = second ? a.x2 : a.x1;
        const int cs = second ? a.C2 : a.C1;
        const int cc = (second ? c0 - a.C1 : c0) + c4 * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) hv[i] = ld4(hoff[i] >= 0 ? src + ((long long)hoff[i] * cs + cc) : zp);
    };
    auto store_raw = [&](int buf, int c) {
        if constexpr (GNF) {
            const int c0 = c * 32;
            const bool second = c0 >= a.C1;
            const int cs = second ? a.C2 : a.C1;
            const long long go = (long long)img0 * cs + (second ? c0 - a.C1 : c0) + c4 * 4;
            const f32x4 sc = ld4((second ? a.gsc2 : a.gsc1) + go), sh = ld4((second ? a.gsh2 : a.gsh1) + go);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float z = hv[i][e] * sc[e] + sh[e];
                    const float v = a.gn_act ? silu_f(z) : z;
                    o[e] = hoff[i] >= 0 ? v : 0.f;          // the convolution pads the ACTIVATED tensor with zeros
                }
                *reinterpret_cast<f32x4*>(Rs + buf * RAWB + rdst[i]) = o;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(Rs + buf * RAWB + rdst[i]) = hv[i];
        }
    };

    // ---- transform share of this thread: item (tile, channel quad) = tid & 255, V row vr = wave >> 2 of the half -------------
    const int vr = wave >> 2;
    const int t_tile = (tid & 255) >> 3, t_q = tid & 7;
    int t_src;      // halo pixel of d[0][0]
    {
        const int seg = t_tile / tps, rem = t_tile - seg * tps;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        t_src = (seg * (a.rps + 2) + 2 * ty) * W2 + 2 * tx;
    }
    const int t_dst = t_tile * ROWB + (((t_q >> 1) ^ lds_swz(t_tile)) << 4) + (t_q & 1) * 8;
    // V row 2 th + vr from two d rows: th = 0: (0: d0 - d2) (1: d1 + d2); th = 1: (2: d2 - d1) (3: d1 - d3)
    auto transform = [&](auto TH, auto VR, int rbuf) {
        constexpr int th = decltype(TH)::value, v_r = decltype(VR)::value;
        constexpr int ra_ = th == 0 ? (v_r == 0 ? 0 : 1) : (v_r == 0 ? 2 : 1);
        constexpr int rb_ = th == 0 ? 2 : (v_r == 0 ? 1 : 3);
        constexpr bool plus = th == 0 && v_r == 1;
        constexpr int vrow = 2 * th + v_r;
        const unsigned char* Rb = Rs + rbuf * RAWB;
        f32x4 ea[4], eb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ea[c] = *reinterpret_cast<const f32x4*>(Rb + raw_off(t_src + ra_ * W2 + c, t_q));
            eb[c] = *reinterpret_cast<const f32x4*>(Rb + raw_off(t_src + rb_ * W2 + c, t_q));
        }
        f32x4 r[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = plus ? ea[c] + eb[c] : ea[c] - eb[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = j == 0 ? r[0] - r[2] : j == 1 ? r[1] + r[2] : j == 2 ? r[2] - r[1] : r[1] - r[3];
            unsigned h0, m0_, l0, h1, m1, l1;
            split3(v[0], v[1], h0, m0_, l0);
            split3(v[2], v[3], h1, m1, l1);
            unsigned char* q = Vs + (vrow * 4 + j) * 3 * VPLANE + t_dst;
            *reinterpret_cast<u32x2*>(q) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(q + VPLANE) = u32x2{m0_, m1};
            *reinterpret_cast<u32x2*>(q + 2 * VPLANE) = u32x2{l0, l1};
        }
    };

    // ---- MFMA role: 32 tiles x 16 channels x 16 positions ------------------------------------------------------------
    const int r16 = lane & 15, kq = lane >> 4;
    int aoff[2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int row = tb * 16 + r16;
        aoff[tb] = row * ROWB + ((kq ^ lds_swz(row)) << 4);
    }
    // ABL & 128 (timing only, wrong results): SIMD partners w, w + 4 stream the SAME fragments - what the L1 merges
    const u32x4* ub = a.ufrag + ((long long)(tile_n * 8 + ((ABL & 128) ? (wave & 3) : wave)) * a.chunks + ch0) * (16 * 3 * 64);      // wave-uniform
    u32x4 bq[4][3];
    auto load_b = [&](int sigma, u32x4 (&dst)[3]) {
        const u32x4* p = ub + (long long)sigma * (3 * 64);
#pragma unroll
        for (int l = 0; l < 3; ++l) dst[l] = p[l * 64 + lane];
    };
    f32x4v acc[16][2];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) acc[p][tb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    u32x4 fa[3][2];          // [limb hi | mid | lo][tile block]
    auto read_a = [&](int p, int l) {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) fa[l][tb] = *reinterpret_cast<const u32x4*>(Vs + (p * 3 + l) * VPLANE + aoff[tb]);
    };
    auto mm = [&](int p, int la, int lb) {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
            acc[p][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(      // weights first: D^T[channel][tile]
                __builtin_bit_cast(bf16x8, bq[p & 3][lb]), __builtin_bit_cast(bf16x8, fa[la][tb]), acc[p][tb], 0, 0, 0);
    };
    // the eight positions of half h of chunk c (see wino_conv_kernel for the group order and the in-place A prefetch)
    auto mfma_half = [&](auto HH, int c) {
        constexpr int h = decltype(HH)::value;
        if constexpr ((ABL & 256) != 0) __builtin_amdgcn_s_setprio(2);       // experiment: the multiplying wave outranks its partner
        read_a(8 * h, 2);
        read_a(8 * h, 1);
        read_a(8 * h, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = 8 * h + i;
            if (!(ABL & 2)) load_b(c * 16 + p + LA, bq[(p + LA) & 3]);
            __builtin_amdgcn_sched_barrier(0);
            mm(p, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (i < 7) read_a(p + 1, 2);
            __builtin_amdgcn_sched_barrier(0);
            mm(p, 1, 1);
            mm(p, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (i < 7) read_a(p + 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(p, 0, 2);
            mm(p, 0, 1);
            mm(p, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (i < 7) read_a(p + 1, 0);
        }
        if constexpr ((ABL & 256) != 0) __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // Static priority for the second-dispatched half (waves 4-7: the arbitration loser of every SIMD pair - MI355X_MICROARCH.md,
    // "Two waves per SIMD" item 4): 454 -> 443 us on 256->256 @32x32, 206 -> 201 us on 512->256 @16x16 (ABL & 4 switches it off).
    if constexpr ((ABL & 4) == 0 && (ABL & 256) == 0) { if (vr != 0) __builtin_amdgcn_s_setprio(1); }

    load_raw(ch0);
    load_b(0, bq[0]);
    load_b(1, bq[1]);
    if (LA > 2) load_b(2, bq[2]);
    store_raw(0, ch0);
    if constexpr (ERAW) load_raw(ch0 + min(1, kch - 1));
    __syncthreads();
    if (vr == 0) transform(I0{}, I0{}, 0); else transform(I0{}, I1{}, 0);      // V rows 0,1 of chunk 0
    __syncthreads();

    // diagnostic build (ABL & 64): where a wave's time goes inside chunk 3 - stamps go to a buffer nothing else reads
    auto stamp = [&](int c, int idx) {
        if constexpr ((ABL & 64) != 0) {
            if (c == 3 && a.dbg) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) a.dbg[((long long)blockIdx.x * 8 + wave) * 8 + idx] = t;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int c = 0; c < kch; ++c) {
        // HP0: MFMAs on V rows 0,1 of chunk c || V rows 2,3 of chunk c (raw image c & 1)
        stamp(c, 0);
        if constexpr (!ERAW) load_raw(ch0 + min(c + 1, kch - 1));
        if (vr == 0 && !(ABL & 1)) transform(I1{}, I0{}, c & 1);             // waves 0-3: transform, then MFMAs
        __builtin_amdgcn_sched_barrier(0);
        if (vr == 0) stamp(c, 1);
        mfma_half(I0{}, c);
        __builtin_amdgcn_sched_barrier(0);
        if (vr != 0) stamp(c, 1);
        if (vr != 0 && !(ABL & 1)) transform(I1{}, I1{}, c & 1);             // waves 4-7: MFMAs, then transform
        stamp(c, 2);
        store_raw((c + 1) & 1, ch0 + min(c + 1, kch - 1));
        stamp(c, 3);
        __syncthreads();
        // HP1: MFMAs on V rows 2,3 of chunk c || V rows 0,1 of chunk c + 1 (raw image (c + 1) & 1; stale for the last chunk)
        stamp(c, 4);
        if (vr == 0 && !(ABL & 1)) transform(I0{}, I0{}, (c + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (vr == 0) stamp(c, 5);
        mfma_half(I1{}, c);
        __builtin_amdgcn_sched_barrier(0);
        if (vr != 0) stamp(c, 5);
        if constexpr (ERAW) load_raw(ch0 + min(c + 2, kch - 1));
        if (vr != 0 && !(ABL & 1)) transform(I0{}, I1{}, (c + 1) & 1);
        stamp(c, 6);
        __syncthreads();
        stamp(c, 7);
    }

    // ---- output transform + fused epilogue (as wino_conv_kernel) -----------------------------------------------------
    const PsldEpilogue& e = a.e;
    const int cn = n0 + wave * 16 + 4 * kq;
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4v bias4 = e.bias ? *reinterpret_cast<const f32x4v*>(e.bias + cn) : zero4;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int tile = tb * 16 + r16;
        const int seg = tile / tps, rem = tile - seg * tps;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int gm00 = ((img0 + seg) * a.H + oy0 + 2 * ty) * a.W + ox0 + 2 * tx;
        const bool ok = img0 + seg < a.B;                   // whole images only: a tile is in range or not
        f32x4v s[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[i][0] = (acc[4 * i][tb] + acc[4 * i + 1][tb]) + acc[4 * i + 2][tb];
            s[i][1] = (acc[4 * i + 1][tb] - acc[4 * i + 2][tb]) - acc[4 * i + 3][tb];
        }
        f32x4v y[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            y[0][b] = (s[0][b] + s[1][b]) + s[2][b];
            y[1][b] = (s[1][b] - s[2][b]) - s[3][b];
        }
        float gs = 0.f, gss = 0.f;
        const int gmc = ok ? gm00 : 0;
        f32x4v rv[2][2], cv[2][2], tbv[2][2];
#pragma unroll
        for (int ya = 0; ya < 2; ++ya)
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
                const int gm = gmc + ya * a.W + xb;
                rv[ya][xb] = e.res ? *reinterpret_cast<const f32x4v*>(e.res + (long long)gm * e.ldres + cn) : zero4;
                cv[ya][xb] = e.accumulate ? *reinterpret_cast<const f32x4v*>(Cw + (long long)gm * a.ldc + cn) : zero4;
                tbv[ya][xb] = e.rowbias ? *reinterpret_cast<const f32x4v*>(e.rowbias + (long long)(gm / e.rows_per_img) * e.ld_rowbias + cn)
                                        : zero4;
            }
#pragma unroll
        for (int ya = 0; ya < 2; ++ya)
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
                const int gm = gmc + ya * a.W + xb;
                f32x4v o = y[ya][xb] * e.alpha + (bias4 + tbv[ya][xb]);
                if (e.res) o += rv[ya][xb];
                o *= e.out_scale;
                if (e.accumulate) o += cv[ya][xb];
                if (ok && !((ABL & 8) && gm != 0)) {
                    *reinterpret_cast<f32x4v*>(Cw + (long long)gm * a.ldc + cn) = o;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        gs += o[v];
                        gss += o[v] * o[v];
                    }
                }
            }
        if (e.gn_part) {
            float s1 = gs, s2 = gss;
#pragma unroll
            for (int sft = 1; sft <= 8; sft <<= 1) {
                s1 += __shfl_xor(s1, sft, 64);
                s2 += __shfl_xor(s2, sft, 64);
            }
            const float q1 = s1, q2 = s2;                // four-channel sums of this lane's quad (gn_fine = 4)
            s1 += __shfl_xor(s1, 16, 64);
            s2 += __shfl_xor(s2, 16, 64);
            // tile block tb = 64 pixels of one image = one run of the partial-sum table (any fixed partition of an image
            // into hw/64 runs serves: the consumer sums all of them): run 2 reg + tb, or image img0 + tb of a two-image region
            const int img = a.nseg == 1 ? img0 : img0 + tb, chunk = a.nseg == 1 ? 2 * reg + tb : 0, chunks = e.gn_hw >> 6;
            if (e.gn_fine == 4) {
                if ((lane & 0xf) == 0 && img < a.B) {
                    const int f = ((n0 + wave * 16) >> 2) + (lane >> 4);
                    double* pp = e.gn_part + (((long long)img * chunks + chunk) * (a.N >> 2) + f) * 2;
                    pp[0] = (double)q1;
                    pp[1] = (double)q2;
                }
            } else if ((lane & 0x1f) == 0 && img < a.B) {
                const int f = ((n0 + wave * 16) >> 3) + (lane >> 5);
                double* pp = e.gn_part + (((long long)img * chunks + chunk) * (a.N >> 3) + f) * 2;
                pp[0] = (double)s1;
                pp[1] = (double)s2;
            }
        }
    }
}

#ifdef PSLD_ABLATIONS
#include "conv_wino_abl.inc"
#endif

template <int ABL = 0, bool GNF = false, bool ERAW = false, int LA = 2>
int launch_wino8s(const WinoArgs& a, hipStream_t stream, const char* name) {
    constexpr size_t LDS = (size_t)VBYTES + 2 * (size_t)4 * 64 * 128;
    static_assert(LDS <= 163840, "LDS budget");
    static PsldPerDeviceFlag configured_; bool& configured = configured_.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv8s_kernel<ABL, GNF, ERAW, LA>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) {
            psld_set_error("%s: hipFuncSetAttribute failed: %s", name, hipGetErrorString(e));
            return PSLD_ERR_LAUNCH;
        }
        configured = true;
    }
    dim3 grid((unsigned)(cdiv(a.M, 128) * (a.N / 128) * (a.ksplit > 1 ? a.ksplit : 1)));
    hipLaunchKernelGGL((wino_conv8s_kernel<ABL, GNF, ERAW, LA>), grid, dim3(WINO_THREADS), LDS, stream, a);
    PSLD_CHECK_LAUNCH(name);
    return PSLD_OK;
}

int wino_cu_count() {
    static int n[PSLD_MAX_DEVICES] = {};      // per device (0 = not asked yet)
    int& cu = n[psld_device_slot()];
    if (cu <= 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cu <= 0)
            cu = 256;
    }
    return cu;
}

bool wino_geometry(int h, int w, int* nseg, int* rps, int* halo_px) {
    if (w != 8 && w != 16 && w != 32 && w != 64) return false;
    if (h % 2) return false;
    const int hw = h * w;
    if (hw >= 128) {
        if (hw % 128) return false;
        *nseg = 1;
        *rps = 128 / w;
    } else {
        if (128 % hw) return false;
        *nseg = 128 / hw;
        *rps = h;
    }
    if (*rps % 2) return false;
    *halo_px = *nseg * (*rps + 2) * (w + 2);
    // wino_conv8s_kernel: the halo of a workgroup's region in one 256-pixel raw image; 64-wide maps run as 4 x 32 blocks
    return w == 64 ? h % 4 == 0 : *halo_px <= 256;
}

}  // namespace

// + 16 KB: the kernels prefetch up to three positions (3 x 3 KB per 16-channel block) past the last fragment
extern "C" long long psld_conv3x3_wino_frag_bytes(int cout, int cin) { return (long long)cout * cin * 16 * 6 + 16384; }

extern "C" int psld_conv3x3_wino_supported(int c1, int c2, int batch, int h, int w, int cout) {
    int nseg, rps, halo;
    return c1 > 0 && c2 >= 0 && c1 % 32 == 0 && c2 % 32 == 0 && cout > 0 && cout % 128 == 0 && batch > 0 &&
           wino_geometry(h, w, &nseg, &rps, &halo);
}

extern "C" int psld_pack_conv3x3_wino(const float* w_oihw, void* ufrag, int cout, int cin, int dgrad, hipStream_t stream) {
    PSLD_CHECK_ARG(w_oihw && ufrag && aligned16(ufrag), "psld_pack_conv3x3_wino: null / unaligned pointer");
    const int n_out = dgrad ? cin : cout, k_in = dgrad ? cout : cin;
    PSLD_CHECK_ARG(n_out > 0 && k_in > 0 && n_out % 128 == 0 && k_in % 32 == 0,
                   "psld_pack_conv3x3_wino: needs out channels %%128 and in channels %%32 (got %d, %d)", n_out, k_in);
    const long long items = (long long)(n_out / 16) * (k_in / 32) * 64;
    const int blocks = (int)((items + 63) / 64 < 16384 ? (items + 63) / 64 : 16384);
    if (dgrad) hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(64), 0, stream, w_oihw, reinterpret_cast<u32x4*>(ufrag),
                                  n_out, k_in, 9LL, (long long)cin * 9, 1);
    else hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(64), 0, stream, w_oihw, reinterpret_cast<u32x4*>(ufrag),
                            n_out, k_in, (long long)cin * 9, 9LL, 0);
    PSLD_CHECK_LAUNCH("psld_pack_conv3x3_wino");
    return PSLD_OK;
}

extern "C" int psld_pack_wino_batch(const long long* table_dev, int entries, long long total_items, hipStream_t stream) {
    PSLD_CHECK_ARG(table_dev && entries > 0 && total_items > 0, "psld_pack_wino_batch: bad args");
    const long long want = (total_items + 63) / 64;
    hipLaunchKernelGGL(wino_pack_batch_kernel, dim3((unsigned)(want < 32768 ? want : 32768)), dim3(64), 0, stream,
                       table_dev, entries, total_items);
    PSLD_CHECK_LAUNCH("psld_pack_wino_batch");
    return PSLD_OK;
}

#ifdef PSLD_ABLATIONS
static unsigned long long* g_wino_dbg = nullptr;
// ablation library only: device buffer for the s_memtime stamps of PSLD_WINO_ABL=64 ([workgroups][8 waves][8] uint64)
extern "C" void psld_abl_set_wino_debug(unsigned long long* p) { g_wino_dbg = p; }
#endif

namespace {
struct WinoGn {
    const float *sc1, *sh1, *sc2, *sh2;
    int act;
};
int wino_conv(const float* x1, int c1, const float* x2, int c2, int batch, int h, int w, const void* ufrag, int cout, float* y,
              int ldy, const psld_epilogue_t* epi, const WinoGn* gn, void* workspace, long long ws_bytes, hipStream_t stream);
}  // namespace

extern "C" int psld_conv3x3_wino_f32(const float* x1, int c1, const float* x2, int c2, int batch, int h, int w,
                                     const void* ufrag, int cout, float* y, int ldy, const psld_epilogue_t* epi,
                                     hipStream_t stream) {
    return wino_conv(x1, c1, x2, c2, batch, h, w, ufrag, cout, y, ldy, epi, nullptr, nullptr, 0, stream);
}

// K splits a launch of this shape takes when it is given a workspace (1: none) and the bytes that workspace needs
extern "C" int psld_conv3x3_wino_ksplit(int c1, int c2, int batch, int h, int w, int cout) {
    if (!psld_conv3x3_wino_supported(c1, c2, batch, h, w, cout)) return 1;
    const int tiles = cdiv((long long)batch * h * w, 128) * (cout / 128), chunks = (c1 + c2) / 32;
    const int cus = wino_cu_count();
    int ks = 1;
    // one workgroup per CU: a grid below half a round leaves CUs idle - split the channel chunks so that it fills one round
    while (tiles * ks * 2 <= cus && chunks % (ks * 2) == 0 && chunks / (ks * 2) >= 2) ks *= 2;
    return ks;
}
extern "C" long long psld_conv3x3_wino_ws_bytes(int c1, int c2, int batch, int h, int w, int cout) {
    const int ks = psld_conv3x3_wino_ksplit(c1, c2, batch, h, w, cout);
    return ks > 1 ? (long long)ks * batch * h * w * cout * 4 : 0;
}

extern "C" int psld_conv3x3_wino_ws_f32(const float* x1, int c1, const float* x2, int c2, int batch, int h, int w,
                                        const void* ufrag, int cout, float* y, int ldy, const psld_epilogue_t* epi,
                                        void* workspace, long long ws_bytes, hipStream_t stream) {
    return wino_conv(x1, c1, x2, c2, batch, h, w, ufrag, cout, y, ldy, epi, nullptr, workspace, ws_bytes, stream);
}

extern "C" int psld_conv3x3_wino_gn_supported(int c1, int c2, int batch, int h, int w, int cout) {
    return psld_conv3x3_wino_supported(c1, c2, batch, h, w, cout) && h * w >= 128;      // one image per workgroup region
}

extern "C" int psld_conv3x3_wino_gn_f32(const float* x1, int c1, const float* scale1, const float* shift1, const float* x2,
                                        int c2, const float* scale2, const float* shift2, int act, int batch, int h, int w,
                                        const void* ufrag, int cout, float* y, int ldy, const psld_epilogue_t* epi,
                                        hipStream_t stream) {
    PSLD_CHECK_ARG(scale1 && shift1 && (c2 == 0 || (scale2 && shift2)), "psld_conv3x3_wino_gn_f32: null scale / shift");
    PSLD_CHECK_ARG(psld_conv3x3_wino_gn_supported(c1, c2, batch, h, w, cout),
                   "psld_conv3x3_wino_gn_f32: unsupported shape c1=%d c2=%d %dx%d cout=%d (needs h*w >= 128)", c1, c2, h, w, cout);
    PSLD_CHECK_ARG(aligned16(scale1) && aligned16(shift1) && (c2 == 0 || (aligned16(scale2) && aligned16(shift2))),
                   "psld_conv3x3_wino_gn_f32: unaligned scale / shift");
    const WinoGn gn{scale1, shift1, scale2, shift2, act};
    return wino_conv(x1, c1, x2, c2, batch, h, w, ufrag, cout, y, ldy, epi, &gn, nullptr, 0, stream);
}

extern "C" int psld_conv3x3_wino_gn_ws_f32(const float* x1, int c1, const float* scale1, const float* shift1, const float* x2,
                                           int c2, const float* scale2, const float* shift2, int act, int batch, int h, int w,
                                           const void* ufrag, int cout, float* y, int ldy, const psld_epilogue_t* epi,
                                           void* workspace, long long ws_bytes, hipStream_t stream) {
    PSLD_CHECK_ARG(scale1 && shift1 && (c2 == 0 || (scale2 && shift2)), "psld_conv3x3_wino_gn_ws_f32: null scale / shift");
    PSLD_CHECK_ARG(psld_conv3x3_wino_gn_supported(c1, c2, batch, h, w, cout),
                   "psld_conv3x3_wino_gn_ws_f32: unsupported shape c1=%d c2=%d %dx%d cout=%d (needs h*w >= 128)", c1, c2, h, w, cout);
    PSLD_CHECK_ARG(aligned16(scale1) && aligned16(shift1) && (c2 == 0 || (aligned16(scale2) && aligned16(shift2))),
                   "psld_conv3x3_wino_gn_ws_f32: unaligned scale / shift");
    const WinoGn gn{scale1, shift1, scale2, shift2, act};
    return wino_conv(x1, c1, x2, c2, batch, h, w, ufrag, cout, y, ldy, epi, &gn, workspace, ws_bytes, stream);
}

namespace {
int wino_conv(const float* x1, int c1, const float* x2, int c2, int batch, int h, int w, const void* ufrag, int cout, float* y,
              int ldy, const psld_epilogue_t* epi, const WinoGn* gn, void* workspace, long long ws_bytes, hipStream_t stream) {
    PSLD_CHECK_ARG(x1 && ufrag && y && (c2 == 0 || x2), "psld_conv3x3_wino_f32: null pointer");
    PSLD_CHECK_ARG(psld_conv3x3_wino_supported(c1, c2, batch, h, w, cout),
                   "psld_conv3x3_wino_f32: unsupported shape c1=%d c2=%d %dx%d cout=%d", c1, c2, h, w, cout);
    PSLD_CHECK_ARG(aligned16(x1) && (!x2 || aligned16(x2)) && aligned16(ufrag), "psld_conv3x3_wino_f32: unaligned pointer");
    WinoArgs a{};
    a.x1 = x1; a.x2 = x2; a.C1 = c1; a.C2 = c2;
    a.B = batch; a.H = h; a.W = w;
    a.ufrag = reinterpret_cast<const u32x4*>(ufrag);
    a.N = cout; a.M = batch * h * w;
    a.chunks = (c1 + c2) / 32;
    a.C = y; a.ldc = ldy;
    int halo_px = 0;
    wino_geometry(h, w, &a.nseg, &a.rps, &halo_px);
    a.e = make_epilogue(epi);
    const PsldEpilogue& e = a.e;
    PSLD_CHECK_ARG(!e.gn_part || (e.gn_hw == h * w && e.gn_hw % 64 == 0 && !e.accumulate),
                   "psld_conv3x3_wino_f32: gn_part needs gn_hw = h*w, a multiple of 64, and no accumulation");
    PSLD_CHECK_ARG(ldy % 4 == 0 && aligned16(y) && (!e.res || (e.ldres % 4 == 0 && aligned16(e.res))) &&
                       (!e.bias || aligned16(e.bias)) && (!e.rowbias || (e.ld_rowbias % 4 == 0 && aligned16(e.rowbias))),
                   "limb kernels: y, residual, bias and rowbias need 16-byte aligned rows (pointer and row stride)");
    a.zero = psld_detail_zero_page("psld_conv3x3_wino_f32");
    if (!a.zero) return PSLD_ERR_LAUNCH;
    a.nmajor = 1;       // channel-tile-major: an XCD streams ONE 128-channel slice of U at a time (pixel-tile-major measured 1-2 % slower)
    const char* name = "psld_conv3x3_wino_f32";
    a.cw = w;
    if (w == 64) {      // 4 x 32 pixel blocks: the 32x32 level's halo (6 x 34 = 204 pixels)
        a.cw = 32; a.rps = 4; a.nseg = 1;
        halo_px = 6 * 34;
    }
    if (gn) {
        a.gsc1 = gn->sc1; a.gsh1 = gn->sh1; a.gsc2 = gn->sc2; a.gsh2 = gn->sh2; a.gn_act = gn->act;
    }
    // Small grids (the 8x8 level at training batches: 128 workgroups for 256 CUs): split the channel chunks over ksplit
    // workgroups per tile, plain partial outputs into the workspace, summed + the whole epilogue by conv_reduce_epilogue -
    // what the direct kernels do for the same shapes (GroupNorm partial sums of the output: formed by that pass).  The GroupNorm-
    // fused form splits the same way (same chunk ranges, same reduction: bitwise the unfused pair).
    const int ks = workspace ? psld_conv3x3_wino_ksplit(c1, c2, batch, h, w, cout) : 1;
    if (ks > 1 && (!e.gn_part || psld_detail_conv_reduce_gn_ok(a.M, cout, e)) && ws_bytes >= (long long)ks * a.M * cout * 4 && aligned16(workspace)) {
        WinoArgs s = a;
        s.ksplit = ks;
        s.slab_stride = (long long)a.M * cout;
        s.C = reinterpret_cast<float*>(workspace);
        s.ldc = cout;
        s.e = make_epilogue(nullptr);
        const int rc = gn ? launch_wino8s<0, true>(s, stream, "psld_conv3x3_wino_gn_f32") : launch_wino8s<0>(s, stream, name);
        if (rc != PSLD_OK) return rc;
        return psld_detail_conv_reduce_epilogue(s.C, ks, a.M, cout, y, ldy, e, stream);
    }
    if (gn) return launch_wino8s<0, true>(a, stream, "psld_conv3x3_wino_gn_f32");
#ifdef PSLD_ABLATIONS      // libpsld_hip_abl.so only: the variants of conv_wino_abl.inc and the timing-only ablations (wrong results)
#include "conv_wino_abl_dispatch.inc"
#endif
    return launch_wino8s<0>(a, stream, name);
}
}  // namespace
