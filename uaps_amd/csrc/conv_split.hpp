// fp32 convolutions on the bf16 matrix pipe: every fp32 operand is split EXACTLY into three bf16 pieces
// (x = x0 + x1 + x2, 8 + 8 + 8 significant bits) and each product a*b is evaluated as the six partial products of weight
// >= 2^-16 (a0b0, a0b1, a1b0, a1b1, a0b2, a2b0; the dropped a1b2 + a2b1 + a2b2 are <= 3 * 2^-24 |ab|, the size of one fp32
// rounding), all accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  The result carries the error of an fp32 fmaf chain
// (tests/test_gpu_conv.py compares both paths with a float64 reference), while six bf16 MFMAs of depth 32 take 96 cycles
// where eight fp32 MFMAs of depth 4 take 256: 2.67x the contraction rate of v_mfma_f32_16x16x4_f32.
//
// Same contract as conv_fwd_kernel (conv_kernels.hpp): NCHW fp32 tensors in and out, replaces the nn.Conv2d forward /
// input gradient of utilities/UAPS_unet.py:36-44, 73, 138.  GEMM view M = pixels, N = output channels, K = (tap, channel).
//   * a k-group (the 8 consecutive k of one lane, `8*(lane>>4)+j`) is 8 input channels at one tap; an MFMA step is 4
//     k-groups.  LDS image of the haloed input tile: [piece][channel group][row][column] of 16-byte units (8 channels
//     of one pixel), so the A fragment of 16 consecutive pixels is one conflict-free ds_read_b128 per piece: the four
//     16-lane groups the hardware services together read 256 contiguous bytes when their bases agree modulo 256 B,
//     which holds for k-groups that differ in the channel group only (plane size is a multiple of 256 B);
//   * weights are pre-split and pre-packed [piece][tap][channel group][Cout][8] once per optimizer step, a chunk's
//     slice is a plain copy into LDS [piece][k-group][BN];
//   * the split happens once per staged element (5.5 VALU operations, conv_split3) and is reused by all BN output
//     channels and all taps.
#pragma once
#include "conv_kernels.hpp"
#include "hints.hpp"
#include "stamps.hpp"

// UAPS_ABLATE (diagnostic builds only, with UAPS_STAMPS: `make stamps`; 0 = the shipped kernel): parts of conv_s32_body removed
// behind the first chunk -- 5: no global loads, splits or LDS stores; 6: also no barriers; 7: also no LDS fragment reads (a pure
// MFMA stream on register operands).  Results are garbage; only the time is of interest.
#ifndef UAPS_ABLATE
#define UAPS_ABLATE 0
#endif

namespace uaps {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// (a, b) -> three dwords of packed bf16 pairs (low half = a's piece, high half = b's) with a = a0 + a1 + a2 exactly
// (round-to-nearest pieces: the residual of an 8-bit rounding has at most 16 significant bits, the next at most 8).
__device__ __forceinline__ void conv_split3(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    const float ra = a - __builtin_bit_cast(float, p0 << 16), rb = b - __builtin_bit_cast(float, p0 & 0xffff0000u);
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, bf16x2));
    const float sa = ra - __builtin_bit_cast(float, p1 << 16), sb = rb - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{sa, sb}, bf16x2));
}

// fp16 form: (a, b) -> two dwords of packed fp16 pairs with a = a0 + a1 up to 2^-23 |a| (11 + 11 significant bits).  The caller
// scales the operands into fp16's range first.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void conv_split2h(float a, float b, unsigned& p0, unsigned& p1) {
    const f16x2 h0 = __builtin_convertvector(f32x2{a, b}, f16x2);
    const float ra = a - (float)h0[0], rb = b - (float)h0[1];
    const f16x2 h1 = __builtin_convertvector(f32x2{ra, rb}, f16x2);
    p0 = __builtin_bit_cast(unsigned, h0); p1 = __builtin_bit_cast(unsigned, h1);
}

// Power-of-two scale s (and 1 / s) that maps |x| <= bound to |s x| < 2^15, the upper end of fp16's range (max 65504): the
// two-piece split then keeps 22 significant bits of every element down to 2^-18 of the bound and an absolute error of
// 2^-40 bound below that.  bound = 0, subnormal or not finite: no scaling.  The scale exponent is clamped to +-100 so that
// s, 1 / s and the product of two inverse scales stay normal numbers.
__device__ __forceinline__ f32x2 h16_scale(float bound) {
    const int e = (int)((__builtin_bit_cast(unsigned, bound) >> 23) & 0xffu);      // 2^(e - 127) <= bound < 2^(e - 126)
    if (e == 0 || e == 255) return f32x2{1.f, 1.f};
    int se = 127 + 14 - (e - 127);
    se = se < 27 ? 27 : (se > 227 ? 227 : se);
    return f32x2{__builtin_bit_cast(float, (unsigned)se << 23), __builtin_bit_cast(float, (unsigned)(254 - se) << 23)};
}
__device__ __forceinline__ float bound_of(const float* p, float mul) { return p ? bound_max(p) * mul : 0.f; }

// number of 8-channel groups of the packed split weights for C contraction channels: padded to a multiple of 4 groups so
// that any channel chunk (8, 16 or 32 channels) stays inside one tap's rows and reads zeros past the last channel
__host__ __device__ constexpr int split_cgroups(int C) { return ((C + 7) / 8 + 3) / 4 * 4; }

// Two workgroups share a CU (one wave of each per SIMD) and run the same program from the same start: with fair
// arbitration they stay in lockstep -- both in their MFMA phase (sharing the matrix pipe), then both in their staging phase
// (pipe idle; measured: 49 % MFMA busy, SQ_WAIT_INST_ANY 63 % of a wave's life).  A static priority for the wave in the odd
// hardware slot of each SIMD lets that workgroup run its MFMA phase at full rate while the other takes the pipe during the
// first one's staging phase: the two fall into opposite phases.  Speed only.
__device__ __forceinline__ void stagger_by_wave_slot() {
    const unsigned wave_slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);     // HW_REG_HW_ID[3:0] = WAVE_ID
    if (wave_slot & 1) __builtin_amdgcn_s_setprio(1);
}

template <int KS, int TH, int TW> struct SplitGeom {
    static constexpr int PAD = KS / 2;
    static constexpr int XOFF = PAD ? 4 : 0;
    static constexpr int IH = TH + 2 * PAD, IW = TW + 2 * XOFF;
    static constexpr int PLANE = IH * IW;                 // 16-byte units of one (piece, channel group) plane
    static constexpr int UPR = IW / 4, NUNITS_PER_CG = IH * UPR;
    static_assert(PLANE % 16 == 0, "planes must be multiples of 256 bytes");
};

// CK = channels per chunk (8, 16, 32); BN = output channels per workgroup (16, 32); XF as in conv_fwd_body
template <int KS, int TH, int TW, int BN, int CK, bool XF, bool H16 = false>
__device__ __forceinline__ void conv_sfwd_body(const ConvFwdArgs& a) {
    using G = SplitGeom<KS, TH, TW>;
    constexpr int NP = H16 ? 2 : 3;                   // pieces per operand: three bf16 (six products) or two fp16 (three products)
    constexpr int TAPS = KS * KS, NCG = CK / 8, NQ = TAPS * NCG, NSTEP = (NQ + 3) / 4, NQP = NSTEP * 4;
    constexpr int IW = G::IW, PLANE = G::PLANE, XS = G::XOFF - G::PAD;
    constexpr int MT = TH * TW / 16, MW = MT / 4, NW = BN / 16, XB = TW / 16;
    constexpr int NUNITS = NCG * G::NUNITS_PER_CG;
    constexpr int NWU = NP * NQP * BN, NWT = (NWU + kConvThreads - 1) / kConvThreads;
    static_assert(NUNITS <= kConvThreads, "one staging unit (4 pixels x 8 channels) per thread");
    static_assert(MT % 4 == 0 && BN % 16 == 0, "tile shape");
    static_assert(NP * NCG * PLANE * 16 >= 16 * BN * 2 * 4, "the statistics epilogue reuses sIn");

    __shared__ __attribute__((aligned(16))) u32x4 sIn[NP * NCG * PLANE];
    __shared__ __attribute__((aligned(16))) u32x4 sW[NP * NQP * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;

    int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    if (bid >= a.B * a.tiles_x * a.tiles_y * a.nblk) return;
    stagger_by_wave_slot();
    const int nb = bid % a.nblk; bid /= a.nblk;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, co0 = nb * BN;
    const int HW = a.H * a.W;
    const uint32_t HW4 = (uint32_t)HW * 4u;

    float in_scale = 1.f, out_scale_a = 1.f, out_scale_w = 1.f;
    if constexpr (H16) {
        const f32x2 sc = h16_scale(__builtin_fmaxf(bound_of(a.in_bound, a.in_mul), bound_of(a.in2_bound, a.in2_mul)));
        in_scale = sc.x; out_scale_a = sc.y; out_scale_w = a.wscale[1];
    }

    // ---- staging plan: this thread's unit = 4 consecutive pixels x the 8 channels of one channel group ----
    const bool has_unit = tid < NUNITS;
    const int ucg = tid / G::NUNITS_PER_CG, urem = tid % G::NUNITS_PER_CG, ur = urem / G::UPR, ucu = urem % G::UPR;
    const int ugy = y0 - G::PAD + ur, ugx = x0 - G::XOFF + ucu * 4;
    const bool uin = has_unit && (unsigned)ugy < (unsigned)a.H && (unsigned)ugx < (unsigned)a.W;
    const uint32_t ugoff = (uint32_t)(ucg * 8 * HW + ugy * a.W + ugx) * 4u;
    const int uloff = (ucg * G::IH + ur) * IW + ucu * 4;
    // weights: unit e = (piece, k-group q, n); the chunk advances the channel group by NCG rows of CoutP units
    const int CGP = a.CinP;                           // for the split kernels ConvFwdArgs::CinP carries the padded group count
    uint32_t wgoff[NWT];
    int wloff[NWT];
#pragma unroll
    for (int n = 0; n < NWT; ++n) {
        const int e = tid + n * kConvThreads;
        const int col = e % BN, row = e / BN, piece = row / NQP, q = row % NQP, tap = q / NCG, cg = q % NCG;
        const bool ok = e < NWU && q < NQ;
        wgoff[n] = ok ? (uint32_t)(((piece * TAPS + tap) * CGP + cg) * a.CoutP + co0 + col) * 16u : kOob;
        wloff[n] = e < NWU ? e : -1;
    }
    const float* in_b = a.in + (size_t)b * a.Csplit * HW;
    const float* in2_b = a.in2 + (size_t)b * (a.Cin - a.Csplit) * HW;
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(a.wp, (uint32_t)(NP * TAPS) * CGP * a.CoutP * 16u);
    const __amdgpu_buffer_rsrc_t rs_xf = XF ? make_rsrc(a.xf + (size_t)(b / (XF ? a.xf_Bg : 1)) * a.Cin, (uint32_t)a.Cin * 8u)
                                            : make_rsrc(a.wp, 0);

    float rin[8][4];
    u32x4 rw[NWT];
    f32x2 rxf[XF ? 8 : 1];

    auto load_chunk = [&](int ci0) {
        const bool second = ci0 >= a.Csplit;        // the chunk lies in one source (Csplit % CK == 0)
        const __amdgpu_buffer_rsrc_t rs_in = second
            ? make_rsrc(in2_b + (size_t)(ci0 - a.Csplit) * HW, (uint32_t)(a.Cin - ci0) * HW4)
            : make_rsrc(in_b + (size_t)ci0 * HW, (uint32_t)(a.Csplit - ci0) * HW4);
#pragma unroll
        for (int c = 0; c < 8; ++c) buf_load<4>(rs_in, uin ? ugoff + (uint32_t)c * HW4 : kOob, rin[c]);
        if constexpr (XF) {                          // padding pixels and channels past Cin read (0, 0): the unit stays zero
#pragma unroll
            for (int c = 0; c < 8; ++c)
                rxf[c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(
                    rs_xf, uin ? (int)((uint32_t)(ci0 + ucg * 8 + c) * 8u) : (int)kOob, 0, 0));
        }
        const uint32_t wbase = (uint32_t)(ci0 / 8) * a.CoutP * 16u;
#pragma unroll
        for (int n = 0; n < NWT; ++n)
            rw[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wgoff[n] == kOob ? kOob : wgoff[n] + wbase), 0, 0));
    };
    // split of the fetched unit into packed pieces pk[piece][pixel] (16 bytes = 8 channels each); one call handles the channel
    // pair c2 of pixel p.  XF: leaky_relu(fma(y, scale, shift)) first; leaky_relu(z) = max(z, slope * z)
    u32x4 pk[NP][4];
    auto split_pair = [&](int idx) {
        const int p = idx / 4, c2 = idx % 4;
        float v0 = rin[2 * c2][p], v1 = rin[2 * c2 + 1][p];
        if constexpr (XF) {
            const float z0 = __builtin_fmaf(v0, rxf[2 * c2].x, rxf[2 * c2].y), z1 = __builtin_fmaf(v1, rxf[2 * c2 + 1].x, rxf[2 * c2 + 1].y);
            v0 = __builtin_fmaxf(z0, z0 * a.xf_slope); v1 = __builtin_fmaxf(z1, z1 * a.xf_slope);
        }
        if constexpr (H16) {
            unsigned q0, q1;
            conv_split2h(v0 * in_scale, v1 * in_scale, q0, q1);
            pk[0][p][c2] = q0; pk[1][p][c2] = q1;
        } else {
            unsigned q0, q1, q2;
            conv_split3(v0, v1, q0, q1, q2);
            pk[0][p][c2] = q0; pk[1][p][c2] = q1; pk[2][p][c2] = q2;
        }
    };
    auto store_chunk = [&]() {
        if (has_unit) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
#pragma unroll
                for (int q = 0; q < NP; ++q) sIn[q * NCG * PLANE + uloff + p] = pk[q][p];
            }
        }
#pragma unroll
        for (int n = 0; n < NWT; ++n)
            if (wloff[n] >= 0) sW[wloff[n]] = rw[n];
    };

    f32x4 acc[MW][NW];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A operand: lane (pixel j, k-group kq) of M tile mt reads unit abase[m] + astep[s] (+ piece plane)
    int abase[MW];
#pragma unroll
    for (int m = 0; m < MW; ++m) {
        const int mt = wave * MW + m;
        abase[m] = (mt / XB) * IW + (mt % XB) * 16 + j + XS;
    }
    int astep[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int q = 4 * s + kq, tap = q / NCG, cg = q % NCG;
        astep[s] = q < NQ ? cg * PLANE + (tap / KS) * IW + (tap % KS) : 0;     // padded k-groups meet zero weights
    }
    const int boff = kq * BN + j;

    // The MFMA phase of a chunk is NSTEP * MW units (step s, M tile m) of 6 * NW MFMAs.  The A fragments of unit u + 1 (and,
    // at a step boundary, the B fragments of the next step) are read from LDS before the MFMAs of unit u issue, and the split
    // of the NEXT chunk's fetched unit (16 channel pairs) is spread over the units behind step 0, in the matrix pipe's shadow.
    constexpr int NU = NSTEP * MW;
    constexpr int U0 = NSTEP > 1 ? MW : NU;          // first unit that carries split work (loads were issued NU - U0 units ago)
    constexpr bool SPLIT_IN_LOOP = NSTEP > 1;
    const int nchunks = (a.Cin + CK - 1) / CK;
    load_chunk(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) split_pair(i);
    store_chunk();
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        if (more) load_chunk((ch + 1) * CK);
        bf16x8 af[2][NP], bfr[2][NW][NP];
        auto read_a = [&](int u, bf16x8 (&dst)[NP]) {
            const int s = u / MW, m = u % MW;
#pragma unroll
            for (int p = 0; p < NP; ++p) dst[p] = __builtin_bit_cast(bf16x8, sIn[p * NCG * PLANE + abase[m] + astep[s]]);
        };
        auto read_b = [&](int s, bf16x8 (&dst)[NW][NP]) {
#pragma unroll
            for (int n = 0; n < NW; ++n)
#pragma unroll
                for (int p = 0; p < NP; ++p) dst[n][p] = __builtin_bit_cast(bf16x8, sW[(p * NQP + 4 * s) * BN + boff + n * 16]);
        };
        read_b(0, bfr[0]);
        read_a(0, af[0]);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int s = u / MW, m = u % MW;
            const bool new_b = u + 1 < NU && (u + 1) % MW == 0;
            if (u + 1 < NU) read_a(u + 1, af[(u + 1) & 1]);
            if (new_b) read_b(s + 1, bfr[(s + 1) & 1]);
#pragma unroll
            for (int n = 0; n < NW; ++n) {           // smallest partial products first
                f32x4 c = acc[m][n];
                if constexpr (H16) {
                    const auto H = [](const bf16x8& v) { return __builtin_bit_cast(f16x8, v); };
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(af[u & 1][1]), H(bfr[s & 1][n][0]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(af[u & 1][0]), H(bfr[s & 1][n][1]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(af[u & 1][0]), H(bfr[s & 1][n][0]), c, 0, 0, 0);
                } else {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][2], bfr[s & 1][n][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][0], bfr[s & 1][n][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][1], bfr[s & 1][n][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][1], bfr[s & 1][n][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][0], bfr[s & 1][n][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][0], bfr[s & 1][n][0], c, 0, 0, 0);
                }
                acc[m][n] = c;
            }
            if (new_b) __builtin_amdgcn_sched_group_barrier(0x100, NP + NP * NW, 0);                          // next unit's DS reads first ...
            else if (u + 1 < NU) __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, (H16 ? 3 : 6) * NW, 0);                              // ... then this unit's MFMAs
            if constexpr (SPLIT_IN_LOOP) {           // unconditional (stale registers when no chunk follows): straight-line code
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (u == U0 + i * (NU - U0) / 16) split_pair(i);
            }
        }
        if constexpr (!SPLIT_IN_LOOP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) split_pair(i);
        }
        __syncthreads();
        if (more) store_chunk();
        __syncthreads();
    }

    // ---- epilogue: lane (j, kq) holds pixels kq*4..kq*4+3 of channel j of every tile (same C/D map as 16x16x4) ----
    float st_s[NW], st_q[NW];
    float chk = 0.f;
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
            if constexpr (H16) { v *= out_scale_a; v *= out_scale_w; }      // exact: powers of two
            v.x += bv; v.y += bv; v.z += bv; v.w += bv;
            const bool ok = co_ok && gy < a.H && gx < a.W;      // W % 4 == 0: the 4 pixels are all inside or all outside
            if (ok) {
                if constexpr (H16) note_nonfinite(chk, v);
                *reinterpret_cast<f32x4*>(out_c + (size_t)gy * a.W + gx) = v;
                const f32x4 d = v - sh;
                st_s[n] += (d.x + d.y) + (d.z + d.w);
                st_q[n] += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        }
    }
    if constexpr (H16) report_nonfinite(a.err, chk, UAPS_ERR_CONV_NONFINITE);
    if (a.stats != nullptr) {                    // per-tile BatchNorm partial sums, fixed order (see conv_fwd_body)
        float* red = reinterpret_cast<float*>(sIn);
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
}

template <int KS, int TH, int TW, int BN, int CK>
__global__ __launch_bounds__(kConvThreads, 2) void conv_sfwd_kernel(ConvFwdArgs a) {
    conv_sfwd_body<KS, TH, TW, BN, CK, false>(a);
}
template <int KS, int TH, int TW, int BN, int CK>
__global__ __launch_bounds__(kConvThreads, 2) void conv_sfwd_bn_kernel(ConvFwdArgs a) {
    conv_sfwd_body<KS, TH, TW, BN, CK, true>(a);
}
template <int KS, int TH, int TW, int BN, int CK>
__global__ __launch_bounds__(kConvThreads, 2) void conv_hfwd_kernel(ConvFwdArgs a) {
    conv_sfwd_body<KS, TH, TW, BN, CK, false, true>(a);
}
template <int KS, int TH, int TW, int BN, int CK>
__global__ __launch_bounds__(kConvThreads, 2) void conv_hfwd_bn_kernel(ConvFwdArgs a) {
    conv_sfwd_body<KS, TH, TW, BN, CK, true, true>(a);
}

// -------------------------------------------------------------------------------------------------
// 3x3 forward / input gradient for >= 32 output channels on v_mfma_f32_32x32x16_bf16 (same split arithmetic).
// The limit of the 16x16x32 kernel above is not the matrix pipe but the SIMD's instruction issue: each MFMA holds it for 8
// of its 16 cycles and the LDS fragment reads, the split VALU work and the staging stores of two co-resident waves fill the
// rest.  This form issues half as many MFMAs for the same work (32 cycles each), stages 8 channels per chunk (the 9 taps
// are 9 k-groups = 4.5 -> 5 MFMA depths of 2 k-groups, 90 % of the pipe instead of 75 %), gives every wave a 64-pixel x BN
// block (BN = 64: 0.5 LDS fragment reads per MFMA instead of 0.75) and lets two threads share a staging unit (88 split
// operations per wave and chunk instead of 176).  M tile = the 32 consecutive pixels of one tile row: a fragment read is
// 512 contiguous bytes per k-group, conflict-free for any tap offset.
// Tile 8 x 32 pixels, 4 waves = 2 rows each; LDS 19.2 KB input + 3 * 10 * BN * 16 B weights (30.7 KB at BN = 64).
// -------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// DIL = dilation of the 3x3 kernel with padding = DIL (the dilated stages of utilities/resnet.py:8-10, 201-203; round 3): the
// halo is DIL rows and DIL <= 4 columns inside the 4-float margin, a tap is DIL pixels away.
// MR = output rows per wave (2, or 4 = 16-row tiles for the 32-channel layers, round 3: half the workgroups, twice the matrix
// work per staged weight fragment and per barrier; the BatchNorm statistics are still written per 8-row half).
template <int BN, bool XF, bool H16 = false, int DIL = 1, int MR = 2>
__device__ __forceinline__ void conv_s32_body(const ConvFwdArgs& a) {
    constexpr int NP = H16 ? 2 : 3;                   // pieces per operand: three bf16 (six products) or two fp16 (three products)
    constexpr int TH = 4 * MR, TW = 32, IH = TH + 2 * DIL, IW = TW + 8, PLANE = IH * IW, XS = 4 - DIL;     // rows start 4 floats left of the tile
    constexpr int NQ = 9, NKS = 5, NQP = 2 * NKS;
    constexpr int NT = BN / 32;                       // 32-channel N tiles per wave (every wave covers all BN channels)
    constexpr int NHU = IH * (IW / 2);                // staging half-units: 2 consecutive pixels x 8 channels
    constexpr int NHT = (NHU + kConvThreads - 1) / kConvThreads;      // half-units per thread: 1 (2 for dilation 4)
    constexpr int NWU = NP * NQ * BN, NWT = (NWU + kConvThreads - 1) / kConvThreads;
    static_assert(DIL >= 1 && DIL <= 4, "dilation 1..4");

    __shared__ __attribute__((aligned(16))) u32x4 sIn[NP * PLANE];
    __shared__ __attribute__((aligned(16))) u32x4 sW[NP * NQP * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    if (bid >= a.B * a.tiles_x * a.tiles_y * a.nblk) return;
    UAPS_STAMP_DECL;                                  // phases (diagnostic build only, stamps.hpp): 0 prologue, 1 first chunk fetched + stored,
                                                      // per chunk: 2 load issue, 3 matrix loop, 4 barrier, 5 LDS stores, 6 barrier; 7 epilogue stores, 8 statistics
    stagger_by_wave_slot();
    const int nb = bid % a.nblk; bid /= a.nblk;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, co0 = nb * BN;
    const int HW = a.H * a.W;
    const uint32_t HW4 = (uint32_t)HW * 4u;

    float in_scale = 1.f, out_scale_a = 1.f, out_scale_w = 1.f;
    if constexpr (H16) {
        const f32x2 sc = h16_scale(__builtin_fmaxf(bound_of(a.in_bound, a.in_mul), bound_of(a.in2_bound, a.in2_mul)));
        in_scale = sc.x; out_scale_a = sc.y; out_scale_w = a.wscale[1];
    }

    // ---- staging plan ----
    bool has_unit[NHT], uin[NHT];
    uint32_t ugoff[NHT];
    int uloff[NHT];
#pragma unroll
    for (int t = 0; t < NHT; ++t) {
        const int u = tid + t * kConvThreads;
        has_unit[t] = u < NHU;
        const int ur = u / (IW / 2), uc = u % (IW / 2);
        const int ugy = y0 - DIL + ur, ugx = x0 - 4 + uc * 2;
        uin[t] = has_unit[t] && (unsigned)ugy < (unsigned)a.H && (unsigned)ugx < (unsigned)a.W;      // W % 4 == 0: both pixels in or out
        ugoff[t] = (uint32_t)(ugy * a.W + ugx) * 4u;
        uloff[t] = ur * IW + uc * 2;
    }
    const int CGP = a.CinP;
    uint32_t wgoff[NWT];
    int wloff[NWT];
#pragma unroll
    for (int n = 0; n < NWT; ++n) {
        const int e = tid + n * kConvThreads;
        const int col = e % BN, row = e / BN, piece = row / NQ, q = row % NQ;
        wgoff[n] = e < NWU ? (uint32_t)(((piece * NQ + q) * CGP) * a.CoutP + co0 + col) * 16u : kOob;
        wloff[n] = e < NWU ? (piece * NQP + q) * BN + col : -1;
    }
    // the padded k-group (slot 9 of every piece) meets zero weights: written once
    for (int e = tid; e < NP * BN; e += kConvThreads) sW[((e / BN) * NQP + NQ) * BN + e % BN] = u32x4{0u, 0u, 0u, 0u};

    const float* in_b = a.in + (size_t)b * a.Csplit * HW;
    const float* in2_b = a.in2 + (size_t)b * (a.Cin - a.Csplit) * HW;
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(a.wp, (uint32_t)(NP * NQ) * CGP * a.CoutP * 16u);
    const __amdgpu_buffer_rsrc_t rs_xf = XF ? make_rsrc(a.xf + (size_t)(b / (XF ? a.xf_Bg : 1)) * a.Cin, (uint32_t)a.Cin * 8u)
                                            : make_rsrc(a.wp, 0);

    f32x2 rin[NHT][8];
    u32x4 rw[NWT];
    f32x2 rxf[XF ? NHT : 1][XF ? 8 : 1];
    u32x4 pk[NHT][NP][2];

    auto load_chunk = [&](int ci0) {
        const bool second = ci0 >= a.Csplit;        // the chunk lies in one source (Csplit % 8 == 0)
        const __amdgpu_buffer_rsrc_t rs_in = second
            ? make_rsrc(in2_b + (size_t)(ci0 - a.Csplit) * HW, (uint32_t)(a.Cin - ci0) * HW4)
            : make_rsrc(in_b + (size_t)ci0 * HW, (uint32_t)(a.Csplit - ci0) * HW4);
#pragma unroll
        for (int t = 0; t < NHT; ++t)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                rin[t][c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_in, uin[t] ? (int)(ugoff[t] + (uint32_t)c * HW4) : (int)kOob, 0, 0));
        if constexpr (XF) {                           // (scale, shift) of the chunk's channels; zeros for a half-unit outside the image
#pragma unroll
            for (int t = 0; t < NHT; ++t)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    rxf[t][c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(
                        rs_xf, uin[t] ? (int)((uint32_t)(ci0 + c) * 8u) : (int)kOob, 0, 0));
        }
        const uint32_t wbase = (uint32_t)(ci0 / 8) * a.CoutP * 16u;
#pragma unroll
        for (int n = 0; n < NWT; ++n)
            rw[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wgoff[n] == kOob ? kOob : wgoff[n] + wbase), 0, 0));
    };
    auto split_pair_t = [&](int t, int idx) {        // channel pair c2 of pixel p of this thread's half-unit t
        const int p = idx / 4, c2 = idx % 4;
        float v0 = rin[t][2 * c2][p], v1 = rin[t][2 * c2 + 1][p];
        if constexpr (XF) {
            const float z0 = __builtin_fmaf(v0, rxf[t][2 * c2].x, rxf[t][2 * c2].y), z1 = __builtin_fmaf(v1, rxf[t][2 * c2 + 1].x, rxf[t][2 * c2 + 1].y);
            v0 = __builtin_fmaxf(z0, z0 * a.xf_slope); v1 = __builtin_fmaxf(z1, z1 * a.xf_slope);
        }
        if constexpr (H16) {
            unsigned q0, q1;
            conv_split2h(v0 * in_scale, v1 * in_scale, q0, q1);
            pk[t][0][p][c2] = q0; pk[t][1][p][c2] = q1;
        } else {
            unsigned q0, q1, q2;
            conv_split3(v0, v1, q0, q1, q2);
            pk[t][0][p][c2] = q0; pk[t][1][p][c2] = q1; pk[t][2][p][c2] = q2;
        }
    };
    auto split_pair = [&](int idx) { split_pair_t(0, idx); };
    constexpr int NU = NKS * MR;                      // units (k-step, row) of 6 * NT MFMAs
    constexpr bool SPLIT_IN_LOOP = NHT == 2 && NU >= 18;      // both half-units are split behind the matrix units of the previous chunk
    auto store_chunk = [&]() {
#pragma unroll
        for (int t = 0; t < NHT; ++t) {
            if constexpr (NHT > 1 && !SPLIT_IN_LOOP) {      // (the second half-unit is split here, between the barriers)
                if (t > 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) split_pair_t(t, i);
                }
            }
            if (has_unit[t]) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) sIn[q * PLANE + uloff[t] + p] = pk[t][q][p];
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NWT; ++n)
            if (wloff[n] >= 0) sW[wloff[n]] = rw[n];
    };

    f32x16 acc[MR][NT];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    // A: lane (pixel r, half h) of the wave's row m reads unit abase[m] + kstep[ks]; k-group q = 2 ks + h is tap q
    int abase[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) abase[m] = (wave * MR + m) * IW + r + XS;
    int kstep[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int q = 2 * ks + h;
        kstep[ks] = q < NQ ? (q / 3) * DIL * IW + (q % 3) * DIL : 0;
    }
    const int boff = h * BN + r;

    const int nchunks = (a.Cin + 7) / 8;
    UAPS_STAMP(0);
    load_chunk(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) split_pair(i);
    if constexpr (SPLIT_IN_LOOP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) split_pair_t(1, i);
    }
    store_chunk();
    __syncthreads();
    UAPS_STAMP(1);
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        if (more && UAPS_ABLATE < 5) load_chunk((ch + 1) * 8);
        UAPS_STAMP(2);
        UAPS_STAMP_FIRST_MFMA();
        bf16x8 af[2][NP], bfr[2][NT][NP];
        auto read_a = [&](int u, bf16x8 (&dst)[NP]) {
#pragma unroll
            for (int p = 0; p < NP; ++p) dst[p] = __builtin_bit_cast(bf16x8, sIn[p * PLANE + abase[u % MR] + kstep[u / MR]]);
        };
        auto read_b = [&](int ks, bf16x8 (&dst)[NT][NP]) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int p = 0; p < NP; ++p) dst[n][p] = __builtin_bit_cast(bf16x8, sW[(p * NQP + 2 * ks) * BN + boff + n * 32]);
        };
        if (UAPS_ABLATE < 7 || ch == 0) {
            read_b(0, bfr[0]);
            read_a(0, af[0]);
            if constexpr (UAPS_ABLATE >= 7) { read_b(1, bfr[1]); read_a(1, af[1]); }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int ks = u / MR, m = u % MR;
            const bool new_b = UAPS_ABLATE < 7 && u + 1 < NU && (u + 1) % MR == 0;
            if (UAPS_ABLATE < 7 && u + 1 < NU) read_a(u + 1, af[(u + 1) & 1]);
            if (new_b) read_b(ks + 1, bfr[(ks + 1) & 1]);
#pragma unroll
            for (int n = 0; n < NT; ++n) {           // smallest partial products first
                f32x16 c = acc[m][n];
                if constexpr (H16) {
                    const auto H = [](const bf16x8& v) { return __builtin_bit_cast(f16x8, v); };
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(H(af[u & 1][1]), H(bfr[ks & 1][n][0]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(H(af[u & 1][0]), H(bfr[ks & 1][n][1]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(H(af[u & 1][0]), H(bfr[ks & 1][n][0]), c, 0, 0, 0);
                } else {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u & 1][2], bfr[ks & 1][n][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u & 1][0], bfr[ks & 1][n][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u & 1][1], bfr[ks & 1][n][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u & 1][1], bfr[ks & 1][n][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u & 1][0], bfr[ks & 1][n][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u & 1][0], bfr[ks & 1][n][0], c, 0, 0, 0);
                }
                acc[m][n] = c;
            }
            if (new_b) __builtin_amdgcn_sched_group_barrier(0x100, NP + NP * NT, 0);
            else if (u + 1 < NU) __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, (H16 ? 3 : 6) * NT, 0);
            if (UAPS_ABLATE < 5 && u >= 2 && u < 10) split_pair(u - 2);  // the next chunk's 8 channel pairs, one per unit behind the first k-step
            if constexpr (SPLIT_IN_LOOP && UAPS_ABLATE < 5) { if (u >= 10 && u < 18) split_pair_t(1, u - 10); }
        }
        UAPS_STAMP(3);
        if (UAPS_ABLATE < 6) __syncthreads();
        UAPS_STAMP(4);
        if (more && UAPS_ABLATE < 5) store_chunk();
        UAPS_STAMP(5);
        if (UAPS_ABLATE < 6) __syncthreads();
        UAPS_STAMP(6);
    }

    // ---- epilogue.  C/D of 32x32: lane (n = r, h) register i holds pixel 8 (i >> 2) + 4 h + (i & 3) of channel n ----
    float st_s[NT], st_q[NT];
    float chk = 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        st_s[n] = 0.f; st_q[n] = 0.f;
        const int co = co0 + n * 32 + r;
        const bool co_ok = co < a.Cout;
        const float bv = (a.bias && co_ok) ? a.bias[co] : 0.f;
        const float sh = stats_shift(a, co, co_ok);
        const int coc = co_ok ? co : 0;
        float* out_c = coc < a.Osplit ? a.out + ((size_t)b * a.Osplit + coc) * HW
                                      : a.out2 + ((size_t)b * (a.Cout - a.Osplit) + (coc - a.Osplit)) * HW;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int gy = y0 + wave * MR + m;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int gx = x0 + 8 * g + 4 * h;
                f32x4 v = f32x4{acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
                if constexpr (H16) { v *= out_scale_a; v *= out_scale_w; }      // exact: powers of two
                v += bv;
                if (co_ok && gy < a.H && gx < a.W) {
                    if constexpr (H16) note_nonfinite(chk, v);
                    *reinterpret_cast<f32x4*>(out_c + (size_t)gy * a.W + gx) = v;
                    const f32x4 d = v - sh;
                    st_s[n] += (d.x + d.y) + (d.z + d.w);
                    st_q[n] += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
                }
            }
        }
    }
    if constexpr (H16) report_nonfinite(a.err, chk, UAPS_ERR_CONV_NONFINITE);
    UAPS_STAMP(7);
    if (a.stats != nullptr) {                    // per-tile BatchNorm partial sums: 8 partials (4 waves x 2 halves) per channel, fixed order
        float* red = reinterpret_cast<float*>(sIn);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            red[((wave * 2 + h) * BN + n * 32 + r) * 2 + 0] = st_s[n];
            red[((wave * 2 + h) * BN + n * 32 + r) * 2 + 1] = st_q[n];
        }
        __syncthreads();
        if constexpr (MR == 2) {
            if (tid < BN && co0 + tid < a.Cout) {
                float s0 = 0.f, q0 = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) { s0 += red[(k * BN + tid) * 2]; q0 += red[(k * BN + tid) * 2 + 1]; }
                const int tpi = a.tiles_x * a.tiles_y;
                a.stats[((size_t)(co0 + tid) * a.B + b) * tpi + ty * a.tiles_x + tx] = make_float2(s0, q0);
            }
        } else {                                  // 16-row tile: one part per 8-row half (waves 0-1 / 2-3), the layout of the 8-row kernels
            static_assert(MR == 2 || MR == 4, "rows per wave");
            const int half = tid / BN, ch = tid % BN, ty8 = 2 * ty + half, tiles8_y = (a.H + 7) / 8;
            if (tid < 2 * BN && co0 + ch < a.Cout && ty8 < tiles8_y) {
                float s0 = 0.f, q0 = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) { s0 += red[((half * 4 + k) * BN + ch) * 2]; q0 += red[((half * 4 + k) * BN + ch) * 2 + 1]; }
                a.stats[((size_t)(co0 + ch) * a.B + b) * (a.tiles_x * tiles8_y) + ty8 * a.tiles_x + tx] = make_float2(s0, q0);
            }
        }
    }
    UAPS_STAMP(8);
    UAPS_STAMP_FLUSH();
}

template <int BN>
__global__ __launch_bounds__(kConvThreads, 2) void conv_s32_kernel(ConvFwdArgs a) { conv_s32_body<BN, false>(a); }
template <int BN>
__global__ __launch_bounds__(kConvThreads, 2) void conv_s32_bn_kernel(ConvFwdArgs a) { conv_s32_body<BN, true>(a); }
template <int BN>
__global__ __launch_bounds__(kConvThreads, 2) void conv_h32_kernel(ConvFwdArgs a) { conv_s32_body<BN, false, true>(a); }
template <int BN>
__global__ __launch_bounds__(kConvThreads, 2) void conv_h32_bn_kernel(ConvFwdArgs a) { conv_s32_body<BN, true, true>(a); }
// dilated 3x3 (dilation 2 / 4, padding = dilation), no staging-time BatchNorm
// 16-row tiles (4 rows per wave) for the 32-channel layers
template <int BN>
__global__ __launch_bounds__(kConvThreads, 2) void conv_s32t_kernel(ConvFwdArgs a) { conv_s32_body<BN, false, false, 1, 4>(a); }
template <int BN>
__global__ __launch_bounds__(kConvThreads, 2) void conv_s32t_bn_kernel(ConvFwdArgs a) { conv_s32_body<BN, true, false, 1, 4>(a); }
template <int BN>
__global__ __launch_bounds__(kConvThreads, 2) void conv_h32t_kernel(ConvFwdArgs a) { conv_s32_body<BN, false, true, 1, 4>(a); }
template <int BN>
__global__ __launch_bounds__(kConvThreads, 2) void conv_h32t_bn_kernel(ConvFwdArgs a) { conv_s32_body<BN, true, true, 1, 4>(a); }
template <int BN, int DIL>
__global__ __launch_bounds__(kConvThreads, 2) void conv_s32d_kernel(ConvFwdArgs a) { conv_s32_body<BN, false, false, DIL>(a); }
template <int BN, int DIL>
__global__ __launch_bounds__(kConvThreads, 2) void conv_h32d_kernel(ConvFwdArgs a) { conv_s32_body<BN, false, true, DIL>(a); }

// -------------------------------------------------------------------------------------------------
// Split weight packing: w [Cout][Cin][KS][KS] fp32 ->
//   sf [piece][tap][CGP(Cin)][CoutP][8]     sf[p][t][g][co][j] = piece p of w[co][8g+j][t]            (forward)
//   sb [piece][tap][CGP(Cout)][CinPn][8]    sb[p][T-1-t][g][ci][j] = piece p of w[8g+j][ci][t]       (input gradient)
// zero padded; 16-byte units of 8 bf16 (three pieces) and, behind a header of kH16Header floats (16 partial maxima of |w| written
// by conv_weight_scale_kernel before the pack kernel runs, then {s, 1 / s, max|w|} with s = h16_scale(max|w|)), the same
// layouts with two fp16 pieces of s * w.
// One thread per (unit, pair of channels).
// -------------------------------------------------------------------------------------------------
constexpr int kH16Header = 32, kH16Parts = 16;      // floats in front of the fp16 pieces; [0, 16) partial maxima, [16] s, [17] 1 / s, [18] max|w|
struct SplitPackDesc { const float* w; unsigned* sf; unsigned* sb; int Cout, Cin, taps, CGf, CoutP, CGb, CinPn; };
template <bool H16>
__device__ __forceinline__ void conv_pack_split_elem(const SplitPackDesc& q, long e) {
    // dword index space: forward part [tap][CGf][CoutP][4 dwords], then backward part [tap][CGb][CinPn][4]
    const long nf = (long)q.taps * q.CGf * q.CoutP * 4, nbk = (long)q.taps * q.CGb * q.CinPn * 4;
    float v0 = 0.f, v1 = 0.f;
    unsigned* dst;
    long piece_stride, idx;
    if (e < nf) {
        if (!q.sf) return;
        const int d = (int)(e % 4); long r = e / 4;
        const int co = (int)(r % q.CoutP); r /= q.CoutP;
        const int g = (int)(r % q.CGf), t = (int)(r / q.CGf);
        const int ci = g * 8 + d * 2;
        if (co < q.Cout && ci < q.Cin) v0 = q.w[((long)co * q.Cin + ci) * q.taps + t];
        if (co < q.Cout && ci + 1 < q.Cin) v1 = q.w[((long)co * q.Cin + ci + 1) * q.taps + t];
        dst = q.sf; piece_stride = nf; idx = e;
    } else if (e < nf + nbk) {
        if (!q.sb) return;
        const long f = e - nf;
        const int d = (int)(f % 4); long r = f / 4;
        const int ci = (int)(r % q.CinPn); r /= q.CinPn;
        const int g = (int)(r % q.CGb), t = (int)(r / q.CGb);
        const int co = g * 8 + d * 2, tt = q.taps - 1 - t;
        if (ci < q.Cin && co < q.Cout) v0 = q.w[((long)co * q.Cin + ci) * q.taps + tt];
        if (ci < q.Cin && co + 1 < q.Cout) v1 = q.w[((long)(co + 1) * q.Cin + ci) * q.taps + tt];
        dst = q.sb; piece_stride = nbk; idx = f;
    } else {
        return;
    }
    if constexpr (H16) {                          // dst points behind the header
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < kH16Parts; ++i) m = __builtin_fmaxf(m, __builtin_bit_cast(float, dst[i - kH16Header]));
        const f32x2 s2 = h16_scale(m);
        const float sc = s2[0], inv = s2[1];
        if (e == 0 || e == nf) {                  // one thread per direction completes the header for the convolution kernels
            float* hdr = reinterpret_cast<float*>(dst) - kH16Header;
            hdr[16] = sc; hdr[17] = inv; hdr[18] = m;
        }
        unsigned p0, p1;
        conv_split2h(v0 * sc, v1 * sc, p0, p1);
        dst[idx] = p0; dst[piece_stride + idx] = p1;
    } else {
        unsigned p0, p1, p2;
        conv_split3(v0, v1, p0, p1, p2);
        dst[idx] = p0; dst[piece_stride + idx] = p1; dst[2 * piece_stride + idx] = p2;
    }
}

// -------------------------------------------------------------------------------------------------
// Weight packing, all layouts of one convolution by one kernel: w [Cout][Cin][KS][KS] (nn.Conv2d.weight) ->
//   wf [tap][CinP][CoutP]            wf[t][ci][co] = w[co][ci][t]            (exact forward)
//   wb [tap][CoutPk][CinPn]          wb[T-1-t][co][ci] = w[co][ci][t]        (exact input gradient)
//   sf / sb, header + hf / hb        the split layouts above; they FOLLOW wf / wb in the same buffers
// zero padded; wf / wb may be null (then nothing of that direction is written).
// -------------------------------------------------------------------------------------------------
struct PackDesc { const float* w; float* wf; float* wb; int Cout, Cin, taps, CinP, CoutP, CoutPk, CinPn, CGf, CGb; };
// offsets (in floats / dwords) of the parts of the two buffers
__host__ __device__ inline long pack_nf(const PackDesc& q) { return (long)q.taps * q.CinP * q.CoutP; }
__host__ __device__ inline long pack_nbk(const PackDesc& q) { return (long)q.taps * q.CoutPk * q.CinPn; }
__host__ __device__ inline long pack_nsf(const PackDesc& q) { return (long)q.taps * q.CGf * q.CoutP * 4; }      // dwords of ONE piece
__host__ __device__ inline long pack_nsb(const PackDesc& q) { return (long)q.taps * q.CGb * q.CinPn * 4; }
__host__ __device__ inline long pack_fwd_floats(const PackDesc& q) { return pack_nf(q) + 3 * pack_nsf(q) + kH16Header + 2 * pack_nsf(q); }
__host__ __device__ inline long pack_bwd_floats(const PackDesc& q) { return pack_nbk(q) + 3 * pack_nsb(q) + kH16Header + 2 * pack_nsb(q); }
__host__ __device__ inline long pack_h16_fwd_off(const PackDesc& q) { return pack_nf(q) + 3 * pack_nsf(q); }      // the header; the pieces follow it
__host__ __device__ inline long pack_h16_bwd_off(const PackDesc& q) { return pack_nbk(q) + 3 * pack_nsb(q); }

__device__ __forceinline__ void conv_pack_elem(const PackDesc& q, long e) {
    const long nf = pack_nf(q), nbk = pack_nbk(q), ns = pack_nsf(q) + pack_nsb(q);
    if (e < nf) {
        if (!q.wf) return;
        const int co = (int)(e % q.CoutP); const long r = e / q.CoutP;
        const int ci = (int)(r % q.CinP), t = (int)(r / q.CinP);
        q.wf[e] = (co < q.Cout && ci < q.Cin) ? q.w[((long)co * q.Cin + ci) * q.taps + t] : 0.f;
    } else if (e < nf + nbk) {
        if (!q.wb) return;
        const long f = e - nf;
        const int ci = (int)(f % q.CinPn); const long r = f / q.CinPn;
        const int co = (int)(r % q.CoutPk), t = (int)(r / q.CoutPk);
        q.wb[f] = (co < q.Cout && ci < q.Cin) ? q.w[((long)co * q.Cin + ci) * q.taps + (q.taps - 1 - t)] : 0.f;
    } else if (e < nf + nbk + ns) {
        SplitPackDesc sp{q.w, q.wf ? reinterpret_cast<unsigned*>(q.wf + nf) : nullptr, q.wb ? reinterpret_cast<unsigned*>(q.wb + nbk) : nullptr,
                         q.Cout, q.Cin, q.taps, q.CGf, q.CoutP, q.CGb, q.CinPn};
        conv_pack_split_elem<false>(sp, e - nf - nbk);
    } else {
        SplitPackDesc sp{q.w, q.wf ? reinterpret_cast<unsigned*>(q.wf + pack_h16_fwd_off(q) + kH16Header) : nullptr,
                         q.wb ? reinterpret_cast<unsigned*>(q.wb + pack_h16_bwd_off(q) + kH16Header) : nullptr,
                         q.Cout, q.Cin, q.taps, q.CGf, q.CoutP, q.CGb, q.CinPn};
        conv_pack_split_elem<true>(sp, e - nf - nbk - ns);
    }
}
__host__ __device__ inline long conv_pack_elems(const PackDesc& q) {
    return pack_nf(q) + pack_nbk(q) + 2 * (pack_nsf(q) + pack_nsb(q));
}
// partial maxima of |w| -> header slots [0, kH16Parts) of both buffers' fp16 parts: kH16Parts workgroups per convolution, fixed order
__device__ __forceinline__ void conv_weight_scale_block(const PackDesc& q, int part) {
    __shared__ float red[256];
    const long n = (long)q.Cout * q.Cin * q.taps;
    float m = 0.f;
    for (long i = (long)part * blockDim.x + threadIdx.x; i < n; i += (long)kH16Parts * blockDim.x)
        m = __builtin_fmaxf(m, __builtin_fabsf(q.w[i]));                 // NaN weights: fmaxf drops them, the products stay NaN
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = __builtin_fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (q.wf) q.wf[pack_h16_fwd_off(q) + part] = red[0];
        if (q.wb) q.wb[pack_h16_bwd_off(q) + part] = red[0];
    }
}
static __global__ __launch_bounds__(256) void conv_weight_scale_kernel(PackDesc q) { conv_weight_scale_block(q, (int)blockIdx.x); }
static __global__ void conv_pack_weights_kernel(PackDesc q) {
    const long n = conv_pack_elems(q);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) conv_pack_elem(q, e);
}

// One launch for up to kPackBatch convolutions (the whole U-Net has 62): blockIdx.y selects the descriptor.
constexpr int kPackBatch = 48;
struct PackBatch { PackDesc d[kPackBatch]; };
static __global__ __launch_bounds__(256) void conv_weight_scale_batch_kernel(PackBatch pb) { conv_weight_scale_block(pb.d[blockIdx.y], (int)blockIdx.x); }
static __global__ void conv_pack_weights_batch_kernel(PackBatch pb) {
    const PackDesc& q = pb.d[blockIdx.y];
    const long n = conv_pack_elems(q);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) conv_pack_elem(q, e);
}

}  // namespace uaps
