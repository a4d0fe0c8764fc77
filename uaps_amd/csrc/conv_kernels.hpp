// Direct (implicit-GEMM) fp32 convolution kernels of the UAPS U-Net for gfx950, on the exact-f32
// matrix instruction v_mfma_f32_16x16x4_f32 (64 FLOP/clk/SIMD, bit-for-bit an fmaf chain).
//
// They replace every nn.Conv2d contraction of the reference model (utilities/UAPS_unet.py:36-44
// ConvBlock 3x3, :73 UpBlock conv1x1, :138 Decoder.out_conv) in forward, input-gradient and
// weight-gradient form.  Tensors stay in the reference's dense NCHW fp32 layout.
//
// GEMM view, forward / input-gradient:  M = pixels, N = output channels, K = (input channel, tap).
//   * a workgroup (4 waves) owns a TH x TW pixel tile of one image and BN output channels;
//   * per K-chunk of CK input channels the haloed input tile and the weight chunk [taps][CK][BN] are
//     staged in LDS; the next chunk is fetched into registers (branch-free buffer loads whose range
//     check supplies the zero padding) while the MFMAs of the current chunk run;
//   * staged rows start 4 floats left of the tile so that every global load and LDS store is 16 B;
//   * NCHW makes the MFMA A operand (16 consecutive pixels of one input-channel plane, shifted by the
//     tap) a conflict-free ds_read_b32: lanes 0-15 walk one row of the plane, the four 16-lane
//     groups take four consecutive channels, plane stride == 16 (mod 32) dwords;
//   * weights are pre-packed [tap][Cin][Cout] (zero padded) so the B operand is 16 consecutive
//     output channels of one (tap, channel) row;
//   * the accumulator tile has 4 consecutive pixels of one output channel per lane -> 16-byte stores.
// The input-gradient of a stride-1 "same" convolution is the same kernel on dY with the weights
// packed transposed and tap-flipped.
//
// Weight gradient: M = output channels, N = input channels, K = pixels, one accumulator tile per tap;
// workgroups split the pixels, write per-split partial slabs, a second kernel sums the slabs in a
// fixed order (no float atomics: bitwise reproducible).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uaps_hip.h"
#include "hints.hpp"

namespace uaps {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// {w, w} for a packed fp32 instruction, with w pinned to the LOW register of an aligned pair: the consumer then broadcasts
// with `op_sel_hi:[..0..]` (high half from the low register) and never with the low-half-from-the-high-register form that
// misbehaves beside 16x16x32 matrix instructions on MI355X (the packed-operand rule of conv_small.hpp).  The empty asm hides
// where w came from; the pair's high register is left undefined.  Costs at most one v_mov_b32.
__device__ __forceinline__ f32x2 bcast_lo(float w) {
    f32x2 t;
    t.x = w;
    t.y = __builtin_nondeterministic_value(w);
    asm("" : "+v"(t));
    return f32x2{t.x, t.x};
}
// A packed fp32 operand held as the two halves of ONE aligned register pair in natural order (no op_sel can be folded into
// its consumer); at most two v_mov_b32.
__device__ __forceinline__ f32x2 natural_pair(f32x2 v) { asm("" : "+v"(v)); return v; }

constexpr int kConvThreads = 256;
#ifndef UAPS_XF_S0
#define UAPS_XF_S0 2      // the staging-time BatchNorm work starts after UAPS_XF_S0 / 4 of a chunk's k-steps
#endif
constexpr uint32_t kOob = 0x80000000u;   // byte offset no tensor of < 2 GiB reaches: buffer loads return 0

// smallest s >= n with s % 32 == r
constexpr int pad_to_mod32(int n, int r) { return n + ((r - n % 32) + 32) % 32; }

// Geometry of a staged (haloed) pixel tile: rows start XOFF floats left of the tile so that the row
// base is 16-byte aligned in global memory (tile origins are multiples of 16 pixels).
// DIL = dilation of a 3x3 kernel (padding = DIL, the "same" form of the reference's dilated ResNet stages,
// utilities/resnet.py:8-10, 201-203); the halo then is DIL rows / columns (DIL <= 4 keeps the 4-float left margin).
template <int KS, int TH, int TW, int DIL = 1> struct TileGeom {
    static_assert(DIL >= 1 && DIL <= 4, "dilation 1..4");
    static constexpr int PAD = (KS / 2) * DIL;
    static constexpr int XOFF = PAD ? 4 : 0;
    static constexpr int IH = TH + 2 * PAD;
    static constexpr int IW = TW + 2 * XOFF;
    static constexpr int PLANE = IH * IW;
};

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Map the hardware
// block id to a logical id such that each XCD works on one contiguous range of logical ids: tiles
// that share halo rows / cache lines then meet in one L2.  The launch grid is rounded up to a
// multiple of 8; logical ids >= total exit.  Speed only, never correctness.
__device__ __forceinline__ int xcd_swizzle(int bid, int grid8) { return (bid % 8) * (grid8 / 8) + bid / 8; }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <int VEC> __device__ __forceinline__ void buf_load(__amdgpu_buffer_rsrc_t rs, uint32_t off, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
    }
}
// LDS store of VEC floats; ALIGN16: the address is 16-byte aligned (else 8-byte for VEC == 4)
template <int VEC, bool ALIGN16> __device__ __forceinline__ void lds_store(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        if constexpr (ALIGN16) {
            *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            *reinterpret_cast<f32x2*>(p) = f32x2{v[0], v[1]};
            *reinterpret_cast<f32x2*>(p + 2) = f32x2{v[2], v[3]};
        }
    } else {
        p[0] = v[0];
    }
}

// Per-thread plan for staging NPL planes of a TileGeom tile in units of VEC floats:
// unit u = tid + n*256 -> (plane c, row r, unit xu).  goff = byte offset inside the image's channel
// block (or kOob for padding / beyond the tile), loff = LDS float offset (or -1).
template <class G, int NPL, int VEC, int PS> struct StagePlan {
    static constexpr int UPR = G::IW / VEC, UPP = G::IH * UPR, NUNITS = NPL * UPP;
    static constexpr int NT = (NUNITS + kConvThreads - 1) / kConvThreads;
    int pos[NT];       // tile-independent: (plane c) << 20 | (row r) << 10 | (column xu*VEC); -1 = no unit
    int loff[NT];      // LDS float offset, -1 = no unit
    uint32_t goff[NT]; // per tile: byte offset inside the image's channel block, kOob = zero padding
    __device__ __forceinline__ void init(int tid) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int u = tid + n * kConvThreads;
            const int c = u / UPP, rem = u % UPP, r = rem / UPR, xu = rem % UPR;
            pos[n] = u < NUNITS ? (c << 20) | (r << 10) | (xu * VEC) : -1;
            loff[n] = u < NUNITS ? c * PS + r * G::IW + xu * VEC : -1;
        }
    }
    __device__ __forceinline__ void place(int y0, int x0, int H, int W) {
        const int HW = H * W, ty = y0 - G::PAD, tx = x0 - G::XOFF;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int c = pos[n] >> 20, gy = ty + ((pos[n] >> 10) & 1023), gx = tx + (pos[n] & 1023);
            const bool inside = pos[n] >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            goff[n] = inside ? (uint32_t)(c * HW + gy * W + gx) * 4u : kOob;
        }
    }
};

// The input may be the channel concatenation of two tensors that is never materialised (UpBlock: torch.cat([skip,
// up], dim=1), UAPS_unet.py:84): channels [0, Csplit) come from `in` [B,Csplit,H,W], the rest from `in2`
// [B,Cin-Csplit,H,W]; Csplit is a multiple of the channel chunk.  Likewise the output (the input gradient of such a
// conv) may be split: channels [0, Osplit) go to `out`, the rest to `out2`.  Csplit = Cin / Osplit = Cout: one tensor.
struct ConvFwdArgs {
    const float* in;    // [B, Csplit, H, W]
    const float* in2;   // [B, Cin - Csplit, H, W] or nullptr
    const float* wp;    // packed [taps][CinP][CoutP]
    const float* bias;  // [Cout] or nullptr
    float* out;         // [B, Osplit, H, W]
    float* out2;        // [B, Cout - Osplit, H, W] or nullptr
    int Csplit, Osplit;
    float2* stats;      // optional [Cout][B][tiles_y*tiles_x] per-tile (sum, sum of squares) of the output, or nullptr
    // the sums are formed about a per-channel shift s = stats_mean[c] - stats_bias[c] (either may be null = 0): sum(v - s),
    // sum((v - s)^2) -- with s near the channel mean (the BatchNorm's running mean minus the conv bias) the variance
    // E[d^2] - E[d]^2 does not cancel when |mean| >> std; the consumer of the partials must be given the same pointers
    const float* stats_mean; const float* stats_bias;
    int B, Cin, Cout, H, W;
    int CinP, CoutP;
    int tiles_x, tiles_y, nblk;
    // conv_fwd_bn_kernel only: the input is the raw output y of a previous conv and this kernel applies the train-mode
    // BatchNorm + LeakyReLU that sits between the two (UAPS_unet.py:38-39) while staging, so the activated tensor is
    // never written: xf[g][c] = (scale, shift) = (gamma*invstd, beta - mean*gamma*invstd) per statistics group
    // g = image / xf_Bg and channel c; the staged value is leaky_relu(fma(y, scale, shift))
    const float2* xf;
    float xf_slope;
    int xf_Bg;
    // fp16-split kernels only (conv_split.hpp): device scalars with an upper bound of |in| (after the XF transform) and of |in2|,
    // each times a host factor, and the packed weights' header {scale, 1 / scale}
    const float* in_bound; const float* in2_bound; const float* wscale;
    float in_mul, in2_mul;
    unsigned* err;      // fp16-split kernels: sticky device error word (uaps_set_error_word) or nullptr
    // UP2 forms of the full-width-row kernels (up2_staging.hpp): in2 is [B, Cin - Csplit, H / 2, W / 2] and is up-sampled x2 while
    // staging; (H/2 - 1) / (H - 1) and (W/2 - 1) / (W - 1) as fp32, computed on the host like uaps_up_cat_fwd does
    float up_rh, up_rw;
    float* amax;        // fp32 kernels: raise this bound (uaps_call_hints::out_amax) to max|output|, or nullptr
    // BS form of the full-width-row kernel (uaps_call_hints::bsum_*): the output IS d(activation) of a train-mode BatchNorm +
    // LeakyReLU whose raw input is bs_y [B, Cout, H, W]; the epilogue forms that BatchNorm's backward sums (sum d, sum d x_hat per
    // 8 x 32-pixel tile into `stats`, the layout of the forward statistics) and raises bs_max[0 / 1] to max|d| / max|x_hat|
    const float* bs_y; const float* bs_mean; const float* bs_invstd; const float* bs_gamma; const float* bs_beta;
    float bs_slope; int bs_Bg;
    float* bs_max;
};

// A magnitude bound that was too small lets a scaled operand overflow fp16: the pieces become +-inf and every output they
// touch NaN (inf - inf in the second piece) -- never a plausible wrong number.  The fp16-split kernels therefore check what
// they store: v * 0 is 0 for a finite v and NaN otherwise (one fma per element in the epilogue), and a wave that saw a
// non-finite output ORs UAPS_ERR_* into the caller's sticky error word.  Non-finite INPUTS raise the same flag; the reader
// (UAPSTrainer.check_errors) says so.
__device__ __forceinline__ void note_nonfinite(float& chk, const f32x4& v) {
    chk = __builtin_fmaf(v.x, 0.f, chk); chk = __builtin_fmaf(v.y, 0.f, chk);
    chk = __builtin_fmaf(v.z, 0.f, chk); chk = __builtin_fmaf(v.w, 0.f, chk);
}
__device__ __forceinline__ void report_nonfinite(unsigned* err, float chk, unsigned code) {
    if (err == nullptr) return;
    const unsigned long long bad = __builtin_amdgcn_ballot_w64(!(chk == 0.f));
    if (bad != 0ull && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(bad)) atomicOr(err, code);
}

__device__ __forceinline__ float stats_shift(const ConvFwdArgs& a, int co, bool co_ok) {
    if (!co_ok || a.stats == nullptr) return 0.f;
    return (a.stats_mean ? a.stats_mean[co] : 0.f) - (a.stats_bias ? a.stats_bias[co] : 0.f);
}

template <int KS, int TH, int TW, int BN, int CK, int VEC, int DIL, bool XF>
__device__ __forceinline__ void conv_fwd_body(const ConvFwdArgs& a) {
    using G = TileGeom<KS, TH, TW, DIL>;
    constexpr int TAPS = KS * KS, IW = G::IW, XS = G::XOFF - G::PAD;
    constexpr int PS = pad_to_mod32(G::PLANE, 16);     // input plane stride in LDS (dwords)
    constexpr int BNS = pad_to_mod32(BN, 16);           // weight row stride in LDS (dwords)
    constexpr int MT = TH * TW / 16, MW = MT / 4, NW = BN / 16, XB = TW / 16;
    constexpr int NWT4 = TAPS * CK * BN / 4;            // float4 units of a weight chunk
    constexpr int NWT_T = (NWT4 + kConvThreads - 1) / kConvThreads;
    static_assert(MT % 4 == 0 && CK % 4 == 0 && BN % 16 == 0 && PS % 4 == 0, "tile shape");
    static_assert(CK * PS >= 32 * BN, "the statistics epilogue reuses sIn");
    using Plan = StagePlan<G, CK, VEC, PS>;

    __shared__ __attribute__((aligned(16))) float sIn[CK * PS];
    __shared__ __attribute__((aligned(16))) float sW[TAPS * CK * BNS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
#ifdef UAPS_CLK_DEBUG      // diagnostic build: shader clock (clock64) against the 100 MHz wall clock over one workgroup's life
    const long long dbg_c0 = clock64(), dbg_w0 = wall_clock64();
#endif

    int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    if (bid >= a.B * a.tiles_x * a.tiles_y * a.nblk) return;
    const int nb = bid % a.nblk; bid /= a.nblk;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, co0 = nb * BN;
    const int HW = a.H * a.W;

    Plan plan;
    plan.init(tid);
    plan.place(y0, x0, a.H, a.W);
    // weight chunk plan: float4 unit e4 -> (row = tap*CK + c, co4)
    uint32_t wgoff[NWT_T];
    int wloff[NWT_T];
#pragma unroll
    for (int n = 0; n < NWT_T; ++n) {
        const int e4 = tid + n * kConvThreads;
        const int co4 = e4 % (BN / 4), row = e4 / (BN / 4);
        const int c = row % CK, tap = row / CK;
        const bool ok = e4 < NWT4;
        wgoff[n] = ok ? (uint32_t)((tap * a.CinP + c) * a.CoutP + co0 + co4 * 4) * 4u : kOob;
        wloff[n] = ok ? row * BNS + co4 * 4 : -1;
    }
    const float* in_b = a.in + (size_t)b * a.Csplit * HW;
    const float* in2_b = a.in2 + (size_t)b * (a.Cin - a.Csplit) * HW;     // only used when Csplit < Cin
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(a.wp, (uint32_t)TAPS * a.CinP * a.CoutP * 4u);

    float rin[Plan::NT][VEC];
    float rw[NWT_T][4];
    f32x2 rxf[XF ? Plan::NT : 1];                // XF: (scale, shift) of each staged unit's channel, (0, 0) for padding
    const __amdgpu_buffer_rsrc_t rs_xf = XF ? make_rsrc(a.xf + (size_t)(b / (XF ? a.xf_Bg : 1)) * a.Cin, (uint32_t)a.Cin * 8u)
                                            : make_rsrc(a.wp, 0);

    auto load_chunk = [&](int ci0) {
        // the chunk lies in one of the two sources (Csplit % CK == 0); channels past the source's end read as zero
        const bool second = ci0 >= a.Csplit;
        const __amdgpu_buffer_rsrc_t rs_in = second
            ? make_rsrc(in2_b + (size_t)(ci0 - a.Csplit) * HW, (uint32_t)(a.Cin - ci0) * HW * 4u)
            : make_rsrc(in_b + (size_t)ci0 * HW, (uint32_t)(a.Csplit - ci0) * HW * 4u);
#pragma unroll
        for (int n = 0; n < Plan::NT; ++n) buf_load<VEC>(rs_in, plan.goff[n], rin[n]);
        if constexpr (XF) {                          // padding pixels and channels past Cin read (0, 0): the unit stays zero
#pragma unroll
            for (int n = 0; n < Plan::NT; ++n)
                rxf[n] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(
                    rs_xf, plan.goff[n] != kOob ? (int)((uint32_t)(ci0 + (plan.pos[n] >> 20)) * 8u) : (int)kOob, 0, 0));
        }
        const uint32_t wbase = (uint32_t)ci0 * a.CoutP * 4u;
#pragma unroll
        for (int n = 0; n < NWT_T; ++n) buf_load<4>(rs_w, wgoff[n] + wbase, rw[n]);
    };
    // XF: leaky_relu((y - mean) * scale + shift) on the fetched registers, padding stays zero.  Called BEFORE the barrier
    // that ends a chunk's MFMA phase, so the VALU work overlaps the matrix pipe instead of sitting between two barriers.
    auto transform_unit = [&](int n) {           // leaky_relu(z) = max(z, slope * z) for 0 <= slope <= 1
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float z = __builtin_fmaf(rin[n][k], rxf[n].x, rxf[n].y);
            rin[n][k] = __builtin_fmaxf(z, z * a.xf_slope);
        }
    };
    auto transform_chunk = [&]() {
        if constexpr (XF) {
#pragma unroll
            for (int n = 0; n < Plan::NT; ++n) transform_unit(n);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int n = 0; n < Plan::NT; ++n)
            if (plan.loff[n] >= 0) lds_store<VEC, true>(&sIn[plan.loff[n]], rin[n]);
#pragma unroll
        for (int n = 0; n < NWT_T; ++n)
            if (wloff[n] >= 0) lds_store<4, true>(&sW[wloff[n]], rw[n]);
    };

    f32x4 acc[MW][NW];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A operand base: plane kq, pixel j of this wave's M tiles; B operand base: row kq, column j
    int aoff[MW];
#pragma unroll
    for (int m = 0; m < MW; ++m) {
        const int mt = wave * MW + m;
        aoff[m] = kq * PS + (mt / XB) * IW + (mt % XB) * 16 + j + XS;
    }
    const int boff = kq * BNS + j;

    const int nchunks = a.CinP / CK;
    load_chunk(0);
    transform_chunk();
    store_chunk();
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        if (more) load_chunk((ch + 1) * CK);
        // k-steps (tap, 4 channels); the fragments of step s+1 are read from LDS before the MFMAs of step s issue
        constexpr int NSTEP = TAPS * (CK / 4);
        float af[2][MW], bf[2][NW];
        auto read_frags = [&](int s, float (&a_)[MW], float (&b_)[NW]) {
            const int tap = s / (CK / 4), c4 = s % (CK / 4);
#pragma unroll
            for (int n = 0; n < NW; ++n) b_[n] = sW[boff + (tap * CK + c4 * 4) * BNS + n * 16];
#pragma unroll
            for (int m = 0; m < MW; ++m) a_[m] = sIn[aoff[m] + c4 * 4 * PS + (tap / KS) * DIL * IW + (tap % KS) * DIL];
        };
        read_frags(0, af[0], bf[0]);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s + 1 < NSTEP) read_frags(s + 1, af[(s + 1) & 1], bf[(s + 1) & 1]);
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s & 1][m], bf[s & 1][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, MW + NW, 0);   // next step's DS reads first ...
            __builtin_amdgcn_sched_group_barrier(0x008, MW * NW, 0);   // ... then this step's MFMAs
            if constexpr (XF) {      // the next chunk's units are normalised one per k-step in the second half of the
                                     // MFMA phase (the loads were issued NSTEP/2 steps ago): VALU work in the matrix pipe's shadow
                constexpr int S0 = NSTEP * UAPS_XF_S0 / 4;
#pragma unroll
                for (int n = 0; n < Plan::NT; ++n)
                    if (s == S0 + n * (NSTEP - S0) / Plan::NT) transform_unit(n);   // unconditional (stale registers when
                                                                                     // there is no next chunk): straight-line code
            }
        }
        __syncthreads();
        if (more) store_chunk();
        __syncthreads();
    }

    // ---- epilogue: lane (j, kq) holds pixels kq*4..kq*4+3 of channel j of every tile ------------
    float st_s[NW], st_q[NW];
    float am = 0.f;                              // max|stored output| of this thread (ConvFwdArgs::amax)
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        st_s[n] = 0.f; st_q[n] = 0.f;
        const int co = co0 + n * 16 + j;
        const bool co_ok = co < a.Cout;
        const float bv = (a.bias && co_ok) ? a.bias[co] : 0.f;
        const float sh = stats_shift(a, co, co_ok);
        const int coc = co_ok ? co : 0;
        float* out_c = coc < a.Osplit ? a.out + ((size_t)b * a.Osplit + coc) * HW
                                      : a.out2 + ((size_t)b * (a.Cout - a.Osplit) + (coc - a.Osplit)) * HW;
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int mt = wave * MW + m;
            const int gy = y0 + mt / XB, gx = x0 + (mt % XB) * 16 + kq * 4;
            f32x4 v = acc[m][n];
            v.x += bv; v.y += bv; v.z += bv; v.w += bv;
            float* p = out_c + (size_t)gy * a.W + gx;
            const bool row_ok = co_ok && gy < a.H;
            if (VEC == 4) {                       // W % 4 == 0: the 4 pixels are all inside or all outside
                const bool ok = row_ok && gx < a.W;
                if (ok) *reinterpret_cast<f32x4*>(p) = v;
                if (ok) am = __builtin_fmaxf(__builtin_fmaxf(am, __builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y))), __builtin_fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
                if (ok) { const f32x4 d = v - sh; st_s[n] += (d.x + d.y) + (d.z + d.w); st_q[n] += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w); }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row_ok && gx + r < a.W) { p[r] = v[r]; am = __builtin_fmaxf(am, __builtin_fabsf(v[r])); const float d = v[r] - sh; st_s[n] += d; st_q[n] += d * d; }
            }
        }
    }
    // ---- optional BatchNorm statistics of this tile (UAPS_unet.py:38,42 batch statistics, first pass): per channel
    // the 16 partial sums (4 waves x 4 lane groups) are combined through LDS in a fixed order -------------------------
    if (a.stats != nullptr) {                    // wave-uniform
        float* red = sIn;                        // free: the chunk loop ended with a barrier
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            red[((wave * 4 + kq) * BN + n * 16 + j) * 2 + 0] = st_s[n];
            red[((wave * 4 + kq) * BN + n * 16 + j) * 2 + 1] = st_q[n];
        }
        __syncthreads();
        if (tid < BN && co0 + tid < a.Cout) {
            float s0 = 0.f, q0 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0 += red[(r * BN + tid) * 2]; q0 += red[(r * BN + tid) * 2 + 1]; }
            const int tpi = a.tiles_x * a.tiles_y;
            a.stats[((size_t)(co0 + tid) * a.B + b) * tpi + ty * a.tiles_x + tx] = make_float2(s0, q0);
        }
    }
    if (a.amax != nullptr) {                     // wave-uniform: the bound of a later convolution's operand (uaps_call_hints::out_amax)
        __shared__ float s_am[16];
        block_amax_to(a.amax, am, s_am);
    }
#ifdef UAPS_CLK_DEBUG
    if (tid == 0 && gridDim.x == 512 && (blockIdx.x % 37 == 0 || blockIdx.x == 511)) {
        const long long dc = clock64() - dbg_c0, w1 = wall_clock64();
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
        printf("wg %3d xcc %u se %u cu %2u: start %lld end %lld (x10ns) len %lld  %.0f MHz\n", (int)blockIdx.x, xcc & 15, (hw >> 13) & 7, (hw >> 8) & 15,
               dbg_w0 % 1000000, w1 % 1000000, w1 - dbg_w0, (double)dc / (double)(w1 - dbg_w0) * 100.0);
    }
#endif
}

template <int KS, int TH, int TW, int BN, int CK, int VEC, int DIL = 1>
__global__ __launch_bounds__(kConvThreads, 2) void conv_fwd_kernel(ConvFwdArgs a) {
    conv_fwd_body<KS, TH, TW, BN, CK, VEC, DIL, false>(a);
}
// the same with BatchNorm(train) + LeakyReLU of the (single) input applied while staging
template <int KS, int TH, int TW, int BN, int CK, int VEC, int DIL = 1>
__global__ __launch_bounds__(kConvThreads, 2) void conv_fwd_bn_kernel(ConvFwdArgs a) {
    conv_fwd_body<KS, TH, TW, BN, CK, VEC, DIL, true>(a);
}

// -------------------------------------------------------------------------------------------------
// Weight gradient.  dw[co][ci][tap] = sum_{b,y,x} dout[b][co][y][x] * in[b][ci][y+ky-PAD][x+kx-PAD].
// A workgroup owns a (16*WCO x 16*WCI) channel block and every `nsplit`-th pixel tile.  Its 4 waves
// are arranged WCO x WCI x WK: wave (wco, wci, wk) accumulates the 16x16 block (wco, wci) for all
// taps over the tile rows wk, wk+WK, ...; the WK partial accumulators are summed through LDS at the
// end and written to slab[split][tap][CoutS][CinS].  conv_wrw_reduce_kernel sums the splits.
// LDS: sD[16*WCO][TH*TW] and sI[16*WCI][haloed tile], plane strides == 2 (mod 32) so that the
// 16 channels x 2 pixels of a 32-lane group hit 32 different banks.
// -------------------------------------------------------------------------------------------------
struct ConvWrwArgs {
    const float* dout;  // [B, Cout, H, W]
    const float* in;    // [B, Csplit, H, W]   (channels [0, Csplit) of the conv input; Csplit % 16 == 0 or == Cin)
    const float* in2;   // [B, Cin - Csplit, H, W] or nullptr: the rest of a never-materialised channel concatenation
    int Csplit;
    float* slab;        // [nsplit][taps][CoutS][CinS]
    float* bslab;       // [nsplit][CoutS] (bias gradient partials) or nullptr
    int B, Cin, Cout, H, W;
    int CoutS, CinS;
    int tiles_x, tiles_y, ncob, ncib, nsplit;
    // fp16-split kernels only: bounds of |dout| and |in| (|in2|) as in ConvFwdArgs
    const float* dy_bound; const float* in_bound; const float* in2_bound;
    float dy_mul, in_mul, in2_mul;
    int col_major;      // split kernels: consecutive tiles of a workgroup run down a 32-pixel column strip (halo rows re-read from L2)
    // conv_wrw_bn_kernel only: `in` is the raw conv output y in front of BatchNorm(train) + LeakyReLU; the activation
    // is recomputed while staging (see ConvFwdArgs::xf)
    const float2* xf;
    float xf_slope;
    int xf_Bg;
    unsigned* err;      // fp16-split kernels: sticky device error word (uaps_set_error_word) or nullptr
    // DT forms (uaps_call_hints::dyt_*): `dout` is the gradient behind the BatchNorm + LeakyReLU that follows this convolution; dy is
    // formed from it, dt_y (the raw conv output) and dt_coef [group][Cout][8] while staging and written through to dt_out
    const float* dt_y; const float* dt_coef; float* dt_out;
    float dt_slope;
    int dt_Bg;
    // UP2 forms of the full-width-row kernels (up2_staging.hpp): in2 is [B, Cin - Csplit, H / 2, W / 2], up-sampled x2 while staged
    float up_rh, up_rw;
};
constexpr int kWrwMaxGroups = 8;     // statistics groups a conv_wrw_bn_kernel keeps coefficients for (norm_act.hip kMaxGroups)

template <int KS, int TH, int TW, int WCO, int WCI, int VEC, int DIL = 1> struct WrwCfg {
    using G = TileGeom<KS, TH, TW, DIL>;
    using GD = TileGeom<1, TH, TW>;
    static constexpr int WK = 4 / (WCO * WCI);
    static constexpr int TAPS = KS * KS;
    static constexpr int BCO = 16 * WCO, BCI = 16 * WCI;
    static constexpr int PSD = pad_to_mod32(TH * TW, 2);
    static constexpr int PSI = pad_to_mod32(G::PLANE, 2);
    static constexpr int STAGE_FLOATS = BCO * PSD + BCI * PSI;
    static constexpr int RED_FLOATS = (WK / 2) * WCO * WCI * (TAPS + 1) * 256;
    static constexpr int LDS_FLOATS = STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS;
};

template <int KS, int TH, int TW, int WCO, int WCI, int VEC, int DIL, bool XF>
__device__ __forceinline__ void conv_wrw_body(const ConvWrwArgs& a) {
    using Cfg = WrwCfg<KS, TH, TW, WCO, WCI, VEC, DIL>;
    using G = typename Cfg::G;
    using GD = typename Cfg::GD;
    constexpr int WK = Cfg::WK, TAPS = Cfg::TAPS, IW = G::IW, XS = G::XOFF - G::PAD;
    constexpr int BCO = Cfg::BCO, BCI = Cfg::BCI, PSD = Cfg::PSD, PSI = Cfg::PSI;
    static_assert(WCO * WCI * WK == 4 && TH % WK == 0 && TW % 4 == 0, "wave arrangement");
    // staging is done in groups of 16 channel planes (one MFMA tile of channels): all groups share one plan, each
    // has its own buffer descriptor, so a group can come from either tensor of a split input
    using PlanD = StagePlan<GD, 16, VEC, PSD>;
    using PlanI = StagePlan<G, 16, VEC, PSI>;

    constexpr int LDSF = (Cfg::LDS_FLOATS + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) float smem[LDSF + (XF ? (kWrwMaxGroups * BCI + 1) * 2 : 0)];
    float* sD = smem;
    float* sI = smem + BCO * PSD;
    // XF: [group][BCI] (scale, shift) of this block's input channels, then one (0, 0) entry that padding units read
    f32x2* sXf = reinterpret_cast<f32x2*>(smem + LDSF);
    constexpr int XF_ZERO = kWrwMaxGroups * BCI;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int wk = wave % WK, wci = (wave / WK) % WCI, wco = wave / (WK * WCI);

    int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    if (bid >= a.nsplit * a.ncob * a.ncib) return;
    const int cib = bid % a.ncib; bid /= a.ncib;
    const int cob = bid % a.ncob;
    const int split = bid / a.ncob;
    const int co0 = cob * BCO, ci0 = cib * BCI;
    const int HW = a.H * a.W;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int ntiles = a.B * tiles_per_img;
    // this split's contiguous tile range (neighbouring tiles share halo cache lines)
    const int t_begin = (int)((long)ntiles * split / a.nsplit), t_end = (int)((long)ntiles * (split + 1) / a.nsplit);
    const bool want_bias = a.bslab != nullptr && cib == 0;

    f32x4 acc[TAPS];
    f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    PlanD pd;
    PlanI pi;
    pd.init(tid);
    pi.init(tid);
    float rd[WCO][PlanD::NT][VEC];
    float ri[WCI][PlanI::NT][VEC];

    int gs_loaded = 0;                       // statistics group of the tile held in ri (XF)
    if constexpr (XF) {
        const int G = a.B / a.xf_Bg;
        for (int i = tid; i < G * BCI; i += kConvThreads) {
            const int g = i / BCI, ch = ci0 + i % BCI;
            f32x2 v = f32x2{0.f, 0.f};               // channels past Cin stay zero
            if (ch < a.Cin) { const float2 t = a.xf[(size_t)g * a.Cin + ch]; v = f32x2{t.x, t.y}; }
            sXf[i] = v;
        }
        if (tid == 0) sXf[XF_ZERO] = f32x2{0.f, 0.f};
        __syncthreads();
    }

    auto load_tile = [&](int t) {
        const int b = t / tiles_per_img, tt = t % tiles_per_img;
        const int y0 = (tt / a.tiles_x) * TH, x0 = (tt % a.tiles_x) * TW;
        if constexpr (XF) gs_loaded = b / a.xf_Bg;
        pd.place(y0, x0, a.H, a.W);
        pi.place(y0, x0, a.H, a.W);
#pragma unroll
        for (int g = 0; g < WCO; ++g) {
            const int c0 = co0 + g * 16;              // channels past Cout read as zero (range check)
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.dout + ((size_t)b * a.Cout + c0) * HW,
                                                        (uint32_t)(c0 < a.Cout ? a.Cout - c0 : 0) * HW * 4u);
#pragma unroll
            for (int n = 0; n < PlanD::NT; ++n) buf_load<VEC>(rs, pd.goff[n], rd[g][n]);
        }
#pragma unroll
        for (int g = 0; g < WCI; ++g) {
            const int c0 = ci0 + g * 16;
            const bool second = c0 >= a.Csplit;
            const int cs = second ? c0 - a.Csplit : c0, cn = second ? a.Cin - a.Csplit : a.Csplit;   // channel in / size of its source
            const float* src = second ? a.in2 : a.in;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(src + ((size_t)b * cn + cs) * HW, (uint32_t)(cs < cn ? cn - cs : 0) * HW * 4u);
#pragma unroll
            for (int n = 0; n < PlanI::NT; ++n) buf_load<VEC>(rs, pi.goff[n], ri[g][n]);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int g = 0; g < WCO; ++g)
#pragma unroll
            for (int n = 0; n < PlanD::NT; ++n)
                if (pd.loff[n] >= 0) lds_store<VEC, false>(&sD[g * 16 * PSD + pd.loff[n]], rd[g][n]);
    };
    // XF: leaky_relu((y - mean) * scale + shift) on the fetched registers, padding stays zero; called before the barrier
    // that ends a tile's MFMA phase so that the VALU work overlaps the matrix pipe
    auto transform_unit = [&](int g, int n) {    // leaky_relu(z) = max(z, slope * z) for 0 <= slope <= 1
        const f32x2 cf = sXf[pi.goff[n] == kOob ? XF_ZERO : gs_loaded * BCI + g * 16 + (pi.pos[n] >> 20)];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float z = __builtin_fmaf(ri[g][n][k], cf.x, cf.y);
            ri[g][n][k] = __builtin_fmaxf(z, z * a.xf_slope);
        }
    };
    auto transform_tile = [&]() {
        if constexpr (XF) {
#pragma unroll
            for (int g = 0; g < WCI; ++g)
#pragma unroll
                for (int n = 0; n < PlanI::NT; ++n) transform_unit(g, n);
        }
    };
    auto store_tile_in = [&]() {
#pragma unroll
        for (int g = 0; g < WCI; ++g)
#pragma unroll
            for (int n = 0; n < PlanI::NT; ++n)
                if (pi.loff[n] >= 0) lds_store<VEC, false>(&sI[g * 16 * PSI + pi.loff[n]], ri[g][n]);
    };

    const float* pa0 = sD + (wco * 16 + j) * PSD + kq;
    const float* pb0 = sI + (wci * 16 + j) * PSI + kq + XS;

    if (t_begin < t_end) {
        load_tile(t_begin);
        transform_tile();
        store_tile();
        store_tile_in();
    }
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const bool more = t + 1 < t_end;
        if (more) load_tile(t + 1);
        // k-steps of this wave: (row rr, x4); the fragments of step s+1 are read from LDS before the MFMAs of
        // step s issue, so the LDS latency hides under 9 MFMAs instead of stalling the head of every step
        constexpr int NSTEP = (TH / WK) * (TW / 4);
        float af[2], bf[2][TAPS];
        auto read_frags = [&](int s, float& a_, float (&b_)[TAPS]) {
            const int row = wk + (s / (TW / 4)) * WK, x4 = s % (TW / 4);
            a_ = pa0[row * TW + x4 * 4];
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) b_[tap] = pb0[(row + (tap / KS) * DIL) * IW + x4 * 4 + (tap % KS) * DIL];
        };
        read_frags(0, af[0], bf[0]);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s + 1 < NSTEP) read_frags(s + 1, af[(s + 1) & 1], bf[(s + 1) & 1]);
            if (want_bias) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s & 1], 1.0f, accb, 0, 0, 0);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s & 1], bf[s & 1][tap], acc[tap], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TAPS + 1, 0);   // next step's DS reads first ...
            __builtin_amdgcn_sched_group_barrier(0x008, TAPS, 0);       // ... then this step's MFMAs
            if constexpr (XF) {      // the next tile's input units are normalised a few per k-step in the second half of
                                     // the MFMA phase: VALU work in the matrix pipe's shadow
                constexpr int S0 = NSTEP * UAPS_XF_S0 / 4, U = WCI * PlanI::NT;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (s == S0 + u * (NSTEP - S0) / U) transform_unit(u / PlanI::NT, u % PlanI::NT);   // also when no tile follows
            }
        }
        __syncthreads();
        if (more) { store_tile(); store_tile_in(); }
        __syncthreads();
    }

    // ---- sum the WK row-split partials of each (wco, wci) block through LDS ------------------------
    if constexpr (WK > 1) {
        float* red = smem;   // [slot][TAPS + 1][4][64], slot = (wk - s) * WCO*WCI + wco*WCI + wci
#pragma unroll
        for (int s = WK / 2; s >= 1; s >>= 1) {
            if (wk >= s && wk < 2 * s) {
                float* p = red + (size_t)(((wk - s) * WCO + wco) * WCI + wci) * (TAPS + 1) * 256 + lane;
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[(t * 4 + r) * 64] = acc[t][r];
#pragma unroll
                for (int r = 0; r < 4; ++r) p[(TAPS * 4 + r) * 64] = accb[r];
            }
            __syncthreads();
            if (wk < s) {
                const float* p = red + (size_t)((wk * WCO + wco) * WCI + wci) * (TAPS + 1) * 256 + lane;
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][r] += p[(t * 4 + r) * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) accb[r] += p[(TAPS * 4 + r) * 64];
            }
            __syncthreads();
        }
    }
    if (wk != 0) return;
    // lane (j, kq), register r: co = co0 + wco*16 + kq*4 + r, ci = ci0 + wci*16 + j
    float* slab = a.slab + (size_t)split * TAPS * a.CoutS * a.CinS;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wco * 16 + kq * 4 + r, ci = ci0 + wci * 16 + j;
            slab[((size_t)t * a.CoutS + co) * a.CinS + ci] = acc[t][r];
        }
    if (want_bias && wci == 0 && j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a.bslab[(size_t)split * a.CoutS + co0 + wco * 16 + kq * 4 + r] = accb[r];
    }
}

template <int KS, int TH, int TW, int WCO, int WCI, int VEC, int DIL = 1>
__global__ __launch_bounds__(kConvThreads, 2) void conv_wrw_kernel(ConvWrwArgs a) {
    conv_wrw_body<KS, TH, TW, WCO, WCI, VEC, DIL, false>(a);
}
// the same with BatchNorm(train) + LeakyReLU of the input recomputed while staging
template <int KS, int TH, int TW, int WCO, int WCI, int VEC, int DIL = 1>
__global__ __launch_bounds__(kConvThreads, 2) void conv_wrw_bn_kernel(ConvWrwArgs a) {
    conv_wrw_body<KS, TH, TW, WCO, WCI, VEC, DIL, true>(a);
}

// dw[co][ci][tap] = sum_s slab[s][tap][co][ci]; db[co] = sum_s bslab[s][co].  A block owns EL consecutive
// slab elements; its 256/EL split lanes each sum every (256/EL)-th split with 4 independent chains, then
// the lanes are combined through LDS in a fixed order -> deterministic, and enough loads in flight to
// stream the slabs at HBM/L2 rate even when the gradient itself is tiny (2304 floats for 16x16x3x3).
template <int EL>
__global__ __launch_bounds__(256) void conv_wrw_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                                              float* __restrict__ dw, float* __restrict__ db, int nsplit, int taps,
                                                              int Cout, int Cin, int CoutS, int CinS) {
    constexpr int SL = 256 / EL;
    __shared__ float red[SL][EL];
    const long n = (long)taps * CoutS * CinS;        // slab elements; the bias partials follow as n .. n+CoutS
    const int el = threadIdx.x % EL, sl = threadIdx.x / EL;
    const long e = (long)blockIdx.x * EL + el;
    const bool is_w = e < n, is_b = !is_w && bslab != nullptr && e - n < CoutS;
    const float* src = is_w ? slab + e : (is_b ? bslab + (e - n) : nullptr);
    const size_t stride = is_w ? (size_t)n : (size_t)CoutS;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (src) {
        int k = sl;
        for (; k + 3 * SL < nsplit; k += 4 * SL) {
            s0 += src[(size_t)k * stride]; s1 += src[(size_t)(k + SL) * stride];
            s2 += src[(size_t)(k + 2 * SL) * stride]; s3 += src[(size_t)(k + 3 * SL) * stride];
        }
        for (; k < nsplit; k += SL) s0 += src[(size_t)k * stride];
    }
    red[sl][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl != 0) return;
    float s = red[0][el];
#pragma unroll
    for (int q = 1; q < SL; ++q) s += red[q][el];
    if (is_w) {
        const int ci = (int)(e % CinS); const long r = e / CinS;
        const int co = (int)(r % CoutS), t = (int)(r / CoutS);
        if (co < Cout && ci < Cin) dw[((long)co * Cin + ci) * taps + t] = s;
    } else if (is_b && db && e - n < Cout) {
        db[e - n] = s;
    }
}

// The same reduction for up to kReduceBatch weight gradients by ONE launch (uaps_conv_bwd_weight_reduce_batch): a step has ~60 of
// them, each a 5-6 us launch of mostly latency.  Block -> (item, block of the item) through the prefix table; an item is summed
// exactly as conv_wrw_reduce_kernel<el> sums it (same lanes, same order): bit-identical results.
constexpr int kReduceBatch = 28;
struct ReduceDesc {
    const float* slab; const float* bslab; float* dw; float* db;
    int nsplit, taps, Cout, Cin, CoutS, CinS;
    int el;                 // 16 or 64: elements per block, as the single launch chooses; 256: a 64-element item in the 16-byte form
    unsigned first;         // first block of this item
};
struct ReduceBatch { ReduceDesc d[kReduceBatch]; int n; unsigned blocks; };

static __global__ __launch_bounds__(256) void conv_wrw_reduce_batch_kernel(ReduceBatch rb) {
    __shared__ __attribute__((aligned(16))) float red[1024];
    if (blockIdx.x >= rb.blocks) return;
    int i = 0;
    while (i + 1 < rb.n && rb.d[i + 1].first <= blockIdx.x) ++i;      // uniform: <= kReduceBatch scalar compares
    const ReduceDesc& q = rb.d[i];
    const long n = (long)q.taps * q.CoutS * q.CinS;
    const unsigned blk = blockIdx.x - q.first;
    if (q.el == 256) {
        // 64-element items, 16 bytes per lane (round 6): a block owns 256 consecutive slab elements, its 4 split lanes x 64 threads
        // each sum 4 adjacent elements with EXACTLY the chains of the 4-byte form (per element: the same splits in the same order,
        // the same (s0 + s1) + (s2 + s3), the same lane order) -- bit-identical, one KiB per wave-load instead of 256 bytes.  The
        // bias partials follow in 64-element blocks of the 4-byte form below.
        const unsigned wblocks = (unsigned)((n + 255) / 256);
        if (blk < wblocks) {
            constexpr int SL = 4;
            const int el4 = threadIdx.x & 63, sl = threadIdx.x >> 6;
            const long e = (long)blk * 256 + el4 * 4;
            f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
            const bool in = e < n;                                      // n % 4 == 0: four elements are all inside or all outside
            if (in) {
                const float* src = q.slab + e;
                int k = sl;
                for (; k + 3 * SL < q.nsplit; k += 4 * SL) {
                    s0 += *reinterpret_cast<const f32x4*>(src + (size_t)k * n); s1 += *reinterpret_cast<const f32x4*>(src + (size_t)(k + SL) * n);
                    s2 += *reinterpret_cast<const f32x4*>(src + (size_t)(k + 2 * SL) * n); s3 += *reinterpret_cast<const f32x4*>(src + (size_t)(k + 3 * SL) * n);
                }
                for (; k < q.nsplit; k += SL) s0 += *reinterpret_cast<const f32x4*>(src + (size_t)k * n);
            }
            const f32x4 t = (s0 + s1) + (s2 + s3);
            *reinterpret_cast<f32x4*>(&red[(sl * 64 + el4) * 4]) = t;
            __syncthreads();
            if (sl != 0 || !in) return;
            f32x4 s = t;
#pragma unroll
            for (int r = 1; r < SL; ++r) s += *reinterpret_cast<const f32x4*>(&red[(r * 64 + el4) * 4]);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const long ev = e + v;
                const int ci = (int)(ev % q.CinS); const long r = ev / q.CinS;
                const int co = (int)(r % q.CoutS), tp = (int)(r / q.CoutS);
                if (co < q.Cout && ci < q.Cin) q.dw[((long)co * q.Cin + ci) * q.taps + tp] = s[v];
            }
            return;
        }
        // bias partials: 64 elements per block, the 4-byte form
        const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
        const long eb = (long)(blk - wblocks) * 64 + el;
        const bool is_b = q.bslab != nullptr && eb < q.CoutS;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (is_b) {
            const float* src = q.bslab + eb;
            const size_t stride = (size_t)q.CoutS;
            int k = sl;
            for (; k + 3 * 4 < q.nsplit; k += 16) {
                s0 += src[(size_t)k * stride]; s1 += src[(size_t)(k + 4) * stride];
                s2 += src[(size_t)(k + 8) * stride]; s3 += src[(size_t)(k + 12) * stride];
            }
            for (; k < q.nsplit; k += 4) s0 += src[(size_t)k * stride];
        }
        red[sl * 64 + el] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (sl != 0) return;
        float s = red[el];
        for (int r = 1; r < 4; ++r) s += red[r * 64 + el];
        if (is_b && q.db && eb < q.Cout) q.db[eb] = s;
        return;
    }
    const int EL = q.el, SL = 256 / EL;
    const int el = threadIdx.x % EL, sl = threadIdx.x / EL;
    const long e = (long)blk * EL + el;
    const bool is_w = e < n, is_b = !is_w && q.bslab != nullptr && e - n < q.CoutS;
    const float* src = is_w ? q.slab + e : (is_b ? q.bslab + (e - n) : nullptr);
    const size_t stride = is_w ? (size_t)n : (size_t)q.CoutS;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (src) {
        int k = sl;
        for (; k + 3 * SL < q.nsplit; k += 4 * SL) {
            s0 += src[(size_t)k * stride]; s1 += src[(size_t)(k + SL) * stride];
            s2 += src[(size_t)(k + 2 * SL) * stride]; s3 += src[(size_t)(k + 3 * SL) * stride];
        }
        for (; k < q.nsplit; k += SL) s0 += src[(size_t)k * stride];
    }
    red[sl * EL + el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl != 0) return;
    float s = red[el];
    for (int r = 1; r < SL; ++r) s += red[r * EL + el];
    if (is_w) {
        const int ci = (int)(e % q.CinS); const long r = e / q.CinS;
        const int co = (int)(r % q.CoutS), t = (int)(r / q.CoutS);
        if (co < q.Cout && ci < q.Cin) q.dw[((long)co * q.Cin + ci) * q.taps + t] = s;
    } else if (is_b && q.db && e - n < q.Cout) {
        q.db[e - n] = s;
    }
}

// (weight packing, exact and split layouts: conv_split.hpp)

}  // namespace uaps
