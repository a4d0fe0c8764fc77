// fp16-split 3x3 forward / input-gradient kernels tiled the way the same-arithmetic ceiling probe says pays (round 6;
// tools/split_gemm_ceiling.hip, profiles/r06_split_gemm_ceiling.txt): ONE workgroup covers a full-width band of
// the map and ALL (or 128) output channels of the layer, so every input element is staged once per layer instead of once per 32- or
// 64-channel block; chunks are 32 input channels (one v_mfma_f32_16x16x32_f16 depth per tap, no padded k-group) with ONE barrier per
// chunk (double-buffered LDS image); the pre-split packed weights never pass through LDS -- a wave loads its B fragments straight from
// the packed buffer (L2-resident; the layout [piece][tap][channel group][Cout][8] is already fragment-shaped: 16 lanes read 256
// contiguous bytes), one tap ahead of the products that use them.
//
// Same contract as conv_s32_body (conv_split.hpp): NCHW fp32 tensors, two-source inputs (the UpBlock's never-materialised torch.cat),
// two-tensor outputs (its input gradient), the staging-time BatchNorm + LeakyReLU (XF), operand scales from device-resident bounds,
// BatchNorm partial sums in the epilogue, the non-finite check.  Replaces nn.Conv2d forward / input gradient of
// utilities/UAPS_unet.py:36-44 on the 32 x 32 maps (128 output channels: down3, up1) of the 256 x 256 step.
//
// Geometry (template): the workgroup's tile is TH rows x TW columns with TW == W (no column halo: the image edge is a zero unit either
// side of every LDS row); WM x WN waves, a wave owns MT M-tiles of 16 consecutive pixels and NT N-tiles of 16 output channels.
//   G128: TW 32, TH 4, 2 x 2 waves, 64 pixels x 64 channels per wave, 128 channels per workgroup; B = 32 at 32 x 32: 256 workgroups.
// LDS image: [buffer][piece][k-group][row][TW + 2] units of 16 bytes (8 channels of one pixel, two fp16 pieces): the A fragment of 16
// consecutive pixels at any tap offset is one ds_read_b128 over 256 contiguous bytes.
#pragma once
#include "conv_split.hpp"

namespace uaps {

// NLW: loader waves.  Vector-memory loads return IN ORDER (one vmcnt): a wave that has the next chunk's activation loads (HBM latency)
// in flight and then waits for a weight fragment issued after them (L2 latency, needed one tap later) waits for the activations too --
// the prefetch of a whole chunk overlaps ONE tap of matrix work.  With NLW > 0 the activation path (fetch, BatchNorm transform, split,
// LDS stores into the other buffer) belongs to NLW extra waves whose waits stall nobody; the WM x WN compute waves only ever wait for
// weight fragments and LDS.  One barrier per chunk joins the two roles.
template <int TW, int TH, int WM, int WN, int MT, int NT, bool XF, int NLW, bool DEEPB = false>
__device__ __forceinline__ void conv_hg_body(const ConvFwdArgs& a) {
    constexpr int NCW = WM * WN, NTHR = (NCW + NLW) * 64, STHR = NLW ? NLW * 64 : NTHR, BN = WN * NT * 16, MPR = TW / 16;
    static_assert(WM * MT == TH * MPR, "the waves' M-tiles cover the tile exactly");
    constexpr int ROWU = TW + 2, IH = TH + 2, PLANE = (IH * ROWU + 15) / 16 * 16;
    constexpr int QPR = TW / 4, NITEMS = 4 * IH * QPR, ITEMS = (NITEMS + STHR - 1) / STHR;
    constexpr int NPART = TW / 32;                    // BatchNorm partial-sum parts per tile (32 columns each)
    constexpr int BUF = 2 * 4 * PLANE;                // units of one buffer
    extern __shared__ __attribute__((aligned(16))) u32x4 g_lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = (wave % NCW) / WN, wn = wave % WN;
    const bool loader = NLW > 0 && wave >= NCW;       // (wave-uniform)
    const bool stager = NLW == 0 || loader;
    const int stid = NLW ? tid - NCW * 64 : tid;      // index among the staging threads (negative on compute waves: never used there)
    int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    if (bid >= a.B * a.tiles_y * a.nblk) return;
    const int nb = bid % a.nblk; bid /= a.nblk;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int y0 = ty * TH, co0 = nb * BN;
    const int HW = a.H * a.W;
    const uint32_t HW4 = (uint32_t)HW * 4u;

    const f32x2 sc = h16_scale(__builtin_fmaxf(bound_of(a.in_bound, a.in_mul), bound_of(a.in2_bound, a.in2_mul)));
    // ONE output scale (the product of two powers of two whose exponents h16_scale clamps so that it stays normal): two separate packed
    // multiplies made hipcc pair the scalars and select the second by op_sel on the low half (tools/isa_lint.py keeps that form out)
    const float in_scale = sc.x, out_scale = sc.y * a.wscale[1];

    // the image edge: one zero unit either side of every row of every plane, written once (staging never touches them)
    for (int e = tid; e < 2 * 2 * 4 * IH * 2; e += NTHR) {
        const int side = e & 1, r = (e >> 1) % IH, pl = (e >> 1) / IH;      // pl = buffer, piece, k-group
        g_lds[pl * PLANE + r * ROWU + (side ? TW + 1 : 0)] = u32x4{0u, 0u, 0u, 0u};
    }

    // ---- staging plan: an item = 4 consecutive pixels of one row x the 8 channels of one k-group ----
    bool uin[ITEMS];
    uint32_t ugoff[ITEMS];
    int uloff[ITEMS], ukg[ITEMS];
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
        const int item = (stid < 0 ? 0 : stid) + t * STHR;
        const int kg = item / (IH * QPR), rem = item % (IH * QPR), r = rem / QPR, q = rem % QPR;
        const int gy = y0 - 1 + r;
        uin[t] = stager && item < NITEMS && (unsigned)gy < (unsigned)a.H;
        ukg[t] = stager && item < NITEMS ? kg : -1;
        ugoff[t] = (uint32_t)((kg * 8) * HW + gy * a.W + q * 4) * 4u;
        uloff[t] = kg * PLANE + r * ROWU + 1 + q * 4;
    }
    const float* in_b = a.in + (size_t)b * a.Csplit * HW;
    const float* in2_b = a.in2 + (size_t)b * (a.Cin - a.Csplit) * HW;
    const __amdgpu_buffer_rsrc_t rs_xf = XF ? make_rsrc(a.xf + (size_t)(b / (XF ? a.xf_Bg : 1)) * a.Cin, (uint32_t)a.Cin * 8u)
                                            : make_rsrc(a.wp, 0);

    f32x4 st[ITEMS][8];
    f32x2 rxf[XF ? ITEMS : 1][XF ? 8 : 1];
    auto fetch = [&](int ci0) {
        const bool second = ci0 >= a.Csplit;          // a chunk lies in one source (Csplit % 32 == 0)
        const __amdgpu_buffer_rsrc_t rs_in = second ? make_rsrc(in2_b + (size_t)(ci0 - a.Csplit) * HW, 32u * HW4)
                                                    : make_rsrc(in_b + (size_t)ci0 * HW, 32u * HW4);
#pragma unroll
        for (int t = 0; t < ITEMS; ++t) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                st[t][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, uin[t] ? (int)(ugoff[t] + (uint32_t)c * HW4) : (int)kOob, 0, 0));
            if constexpr (XF) {                       // (scale, shift) of the item's channels; zeros for a row outside the image
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    rxf[t][c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(
                        rs_xf, uin[t] ? (int)((uint32_t)(ci0 + ukg[t] * 8 + c) * 8u) : (int)kOob, 0, 0));
            }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int t = 0; t < ITEMS; ++t) {
            if (ukg[t] >= 0) {                        // (wave-uniform: NITEMS is a multiple of 64)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u32x4 hi, lo;
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2) {
                        float v0 = st[t][2 * c2][j], v1 = st[t][2 * c2 + 1][j];
                        if constexpr (XF) {
                            const float z0 = __builtin_fmaf(v0, rxf[t][2 * c2].x, rxf[t][2 * c2].y), z1 = __builtin_fmaf(v1, rxf[t][2 * c2 + 1].x, rxf[t][2 * c2 + 1].y);
                            v0 = __builtin_fmaxf(z0, z0 * a.xf_slope); v1 = __builtin_fmaxf(z1, z1 * a.xf_slope);
                        }
                        unsigned q0, q1;
                        conv_split2h(v0 * in_scale, v1 * in_scale, q0, q1);
                        hi[c2] = q0; lo[c2] = q1;
                    }
                    g_lds[buf * BUF + uloff[t] + j] = hi;
                    g_lds[buf * BUF + 4 * PLANE + uloff[t] + j] = lo;
                }
            }
        }
    };
    // B fragment of (tap t, piece p, N-tile j) for the chunk at channel ci0: lane (n = l % 16, k-group l / 16)
    const u32x4* wq = reinterpret_cast<const u32x4*>(a.wp);
    const int CGP = a.CinP;
    const size_t wlane = (size_t)(lane >> 4) * a.CoutP + co0 + wn * NT * 16 + (lane & 15);
    auto bload = [&](int ci0, int t, u32x4 (&bf)[NT][2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const size_t base = ((size_t)(p * 9 + t) * CGP + ci0 / 8) * a.CoutP + wlane;
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j][p] = wq[base + j * 16];
        }
    };

    const int nchunks = a.Cin / 32;
    if constexpr (NLW > 0) {
        if (loader) {      // a code path of its own (no accumulators live in it): as many barriers as the compute path below executes
            fetch(0);
            stage(0);
            __syncthreads();
            for (int ch = 0; ch < nchunks; ++ch) {
                if (ch + 1 < nchunks) {                // the whole activation path of the next chunk, into the buffer nobody reads now
                    fetch((ch + 1) * 32);
                    stage((ch & 1) ^ 1);
                }
                __syncthreads();
            }
            if (a.stats != nullptr) __syncthreads();
            return;
        }
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int abase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int g = wm * MT + i;
        abase[i] = (lane >> 4) * PLANE + (g / MPR) * ROWU + (g % MPR) * 16 + (lane & 15);
    }

    if constexpr (!DEEPB) {
    // Shipped form: the weight fragments of the NEXT tap are fetched under the products of the current one (64 registers); the next
    // chunk's activations are fetched at the chunk's start.  Vector-memory loads return in order, so the first wait for a weight
    // fragment behind the fetch also waits for the fetch: on operands that come from HBM every chunk stalls about a tap's length --
    // visible when the kernel has the chip to itself (53.5 us per launch in the single-stream trace against 49 us of the tile
    // kernels), invisible beside the other decoders' launches, where this form measured best (profiles/r06_hg128_ab.txt).
    u32x4 bcur[NT][2], bnext[NT][2];
    if constexpr (NLW == 0) fetch(0);
    bload(0, 0, bcur);
    if constexpr (NLW == 0) stage(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < nchunks;
        if constexpr (NLW == 0) { if (more) fetch((ch + 1) * 32); }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) bload(ch * 32, t + 1, bnext);
            else if (more) bload((ch + 1) * 32, 0, bnext);
            const int toff = buf * BUF + (t / 3) * ROWU + (t % 3);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const f16x8 ahi = __builtin_bit_cast(f16x8, g_lds[toff + abase[i]]);
                const f16x8 alo = __builtin_bit_cast(f16x8, g_lds[toff + 4 * PLANE + abase[i]]);
#pragma unroll
                for (int j = 0; j < NT; ++j) {       // smallest partial products first
                    f32x4 v = acc[i][j];
                    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, __builtin_bit_cast(f16x8, bcur[j][0]), v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, __builtin_bit_cast(f16x8, bcur[j][1]), v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, __builtin_bit_cast(f16x8, bcur[j][0]), v, 0, 0, 0);
                    acc[i][j] = v;
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) { bcur[j][0] = bnext[j][0]; bcur[j][1] = bnext[j][1]; }
        }
        if constexpr (NLW == 0) { if (more) stage(buf ^ 1); }
        __syncthreads();
    }
    } else {
    // Weight fragments of all nine taps live in registers (288 at NT = 4: one wave per SIMD owns the whole register file), loaded in two
    // batches placed around the activation fetch so that NO wait for a weight fragment ever includes the fetch (vector-memory loads
    // return in order): taps 5-8 of this chunk are issued at its start IN FRONT of the next chunk's fetch and first used at tap 5;
    // taps 0-4 of the next chunk are issued at tap 5 (behind the fetch, into the registers taps 0-4 have just released) and are first
    // used a chunk boundary later, behind the staging that needs the fetch anyway.  The fetch is waited for only by the staging at
    // the chunk's end: a whole chunk of matrix work (~3 us) covers it.  (First form of this kernel: one tap of weights ahead, fetch
    // in front -- every chunk stalled one tap behind its fetch for the HBM latency: 53.5 us per launch in the step against 49 us of
    // the tile kernels, although it was 18 % faster on cache-resident operands.  profiles/r06_hg128_ab.txt)
    // Measured: 48.3 us per launch single-stream (the tile kernels: 49.7), but no gain in the headline mode, where the simpler form
    // above won 7 of 7 interleaved pairs; kept behind UAPS_TUNE_G_DEEP.
    u32x4 bf[9][NT][2];
    if constexpr (NLW == 0) fetch(0);
#pragma unroll
    for (int t = 0; t < 5; ++t) bload(0, t, bf[t]);
    if constexpr (NLW == 0) stage(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < nchunks;
#pragma unroll
        for (int t = 5; t < 9; ++t) bload(ch * 32, t, bf[t]);
        if constexpr (NLW == 0) { if (more) fetch((ch + 1) * 32); }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t == 5 && more) {
#pragma unroll
                for (int tt = 0; tt < 5; ++tt) bload((ch + 1) * 32, tt, bf[tt]);
            }
            const int toff = buf * BUF + (t / 3) * ROWU + (t % 3);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const f16x8 ahi = __builtin_bit_cast(f16x8, g_lds[toff + abase[i]]);
                const f16x8 alo = __builtin_bit_cast(f16x8, g_lds[toff + 4 * PLANE + abase[i]]);
#pragma unroll
                for (int j = 0; j < NT; ++j) {       // smallest partial products first
                    f32x4 v = acc[i][j];
                    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, __builtin_bit_cast(f16x8, bf[t][j][0]), v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, __builtin_bit_cast(f16x8, bf[t][j][1]), v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, __builtin_bit_cast(f16x8, bf[t][j][0]), v, 0, 0, 0);
                    acc[i][j] = v;
                }
            }
        }
        if constexpr (NLW == 0) { if (more) stage(buf ^ 1); }
        __syncthreads();
    }

    }

    // ---- epilogue.  C/D of 16x16: lane l holds pixels 4 (l / 16) .. + 3 of channel l % 16 ----
    float st_s[NT][NPART], st_q[NT][NPART];
    float chk = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int p = 0; p < NPART; ++p) { st_s[j][p] = 0.f; st_q[j][p] = 0.f; }
        const int co = co0 + (wn * NT + j) * 16 + (lane & 15);
        const float bv = a.bias ? a.bias[co] : 0.f;
        const float sh = stats_shift(a, co, true);
        float* out_c = co < a.Osplit ? a.out + ((size_t)b * a.Osplit + co) * HW
                                     : a.out2 + ((size_t)b * (a.Cout - a.Osplit) + (co - a.Osplit)) * HW;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int g = wm * MT + i;
            const int gy = y0 + g / MPR, gx = (g % MPR) * 16 + (lane >> 4) * 4;
            f32x4 v = acc[i][j];
            v *= out_scale;                           // exact: a power of two
            v += bv;
            note_nonfinite(chk, v);
            *reinterpret_cast<f32x4*>(out_c + (size_t)gy * a.W + gx) = v;
            const f32x4 d = v - sh;
            const int part = ((g % MPR) * 16) / 32;
            st_s[j][part] += (d.x + d.y) + (d.z + d.w);
            st_q[j][part] += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
    }
    report_nonfinite(a.err, chk, UAPS_ERR_CONV_NONFINITE);
    if (a.stats != nullptr) {
        // per channel: the four 16-lane groups of a wave (fixed order), then the WM waves of the channel's column (fixed order)
        float* red = reinterpret_cast<float*>(g_lds);      // [wm][BN][NPART][2]; every fragment read is behind the loop's last barrier
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int p = 0; p < NPART; ++p) {
                float s = st_s[j][p], q = st_q[j][p];
                s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
                s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
                if (lane < 16) {
                    const int cb = (wn * NT + j) * 16 + lane;
                    red[((wm * BN + cb) * NPART + p) * 2 + 0] = s;
                    red[((wm * BN + cb) * NPART + p) * 2 + 1] = q;
                }
            }
        __syncthreads();
        for (int e = tid; e < BN * NPART; e += NCW * 64) {      // (the compute waves: tid < NCW * 64 here)
            const int cb = e / NPART, p = e % NPART;
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int k = 0; k < WM; ++k) { s += red[((k * BN + cb) * NPART + p) * 2]; q += red[((k * BN + cb) * NPART + p) * 2 + 1]; }
            // parts of an image: tiles_y x NPART (G128: one part per 4-row tile -- the layout uaps_conv_fwd_stats_parts reports for
            // UAPS_CONV_BOUNDED calls of this form)
            a.stats[((size_t)(co0 + cb) * a.B + b) * (a.tiles_y * NPART) + ty * NPART + p] = make_float2(s, q);
        }
    }
}

constexpr int kHg128Lds = 2 * 2 * 4 * ((6 * 34 + 15) / 16 * 16) * 16;      // two buffers x two pieces x four k-groups x plane, bytes

// Shipped: four waves, one per SIMD (the whole 512-register file per wave: the compiler then hoists the weight-fragment loads and LDS
// fragment reads far ahead of their products; bounded to two workgroups per CU -- 210 registers -- the same source runs 35 % slower,
// and so does the loader-wave form, whose six waves leave 256 registers per wave: 166 used, 26 % slower.  profiles/r06_hg128_ab.txt)
constexpr int kHg128Threads = 2 * 2 * 64;
__global__ __launch_bounds__(kHg128Threads, 1) void conv_hg128_kernel(ConvFwdArgs a) { conv_hg_body<32, 4, 2, 2, 4, 4, false, 0>(a); }
__global__ __launch_bounds__(kHg128Threads, 1) void conv_hg128_bn_kernel(ConvFwdArgs a) { conv_hg_body<32, 4, 2, 2, 4, 4, true, 0>(a); }
__global__ __launch_bounds__(kHg128Threads, 1) void conv_hg128_deep_kernel(ConvFwdArgs a) { conv_hg_body<32, 4, 2, 2, 4, 4, false, 0, true>(a); }

}  // namespace uaps
