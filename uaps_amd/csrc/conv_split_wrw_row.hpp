// Weight gradient of the 3x3 convolutions with <= 16 output channels and 16 / 32 input channels on 256-pixel-wide maps (the top
// level of the U-Net at the metric's image size), two-piece fp16 form of conv_split_wrw.hpp, FULL-WIDTH ROWS (round 4).
//
// Same reason as conv_split_row16.hpp: these launches are HBM-bound, and the 4 / 8-row x 32-pixel tiles of conv_swrw_body read
// every channel's tile as 128-byte row pieces 1 KiB apart plus two neighbour pixels per piece -- three line requests per piece; the
// access pattern, not the arithmetic, set their time (2.3 TB/s at 32 -> 16 channels, B = 128).  Here a workgroup owns runs of 16
// consecutive rows of one image at full width and walks down them one row per step: per step it fetches ONE new input row of
// every channel and ONE dy row (every wave-instruction a 1-KiB row of one channel), keeps three input rows and the dy row in LDS
// ([piece][channel][slot][34 groups of 8 pixels]: a zero group on either side is the image edge) and contracts dy row y with
// the input rows y - 1, y, y + 1 over the row's 8 segments of 32 pixels.  The column taps come from the aligned fragment and the
// neighbour groups' boundary dwords (v_alignbit), as in the dilated form of conv_swrw_body -- no edge loads at all.  A wave owns
// two segments (the K dimension is split over the waves), the nine tap accumulators are summed over the waves at the end and go to
// the same per-split slabs conv_swrw_body writes: plan, workspace and conv_wrw_reduce_kernel are unchanged.
// Same arithmetic as conv_swrw_body<.., H16 = true> (fp32 accumulation of the three partial products); the order in which pixels
// enter the sums differs, so the results agree to rounding, not bit for bit.
#pragma once
#include "conv_split_wrw.hpp"
#include "up2_staging.hpp"

namespace uaps {

// WCI = input channels / 16 (1 or 2); threads = 256 * WCI; XF: the input is a raw conv output, BatchNorm + LeakyReLU applied while staging
// STRIP: maps wider than 256 pixels (W % 256 == 0) as 256-wide column strips (conv_hr16_body): a run is 16 rows of one strip; the one
// real pixel either side of a strip goes into the boundary dwords of the two margin groups, which the 256-wide form leaves zero
// DT: a.dout is the gradient behind the BatchNorm + LeakyReLU that follows this convolution; the dy row is formed from it and the
// convolution's raw output (a.dt_y, a.dt_coef: uaps_bn_act_bwd_prepare) while it is staged, and written through to a.dt_out --
// every dy row is fetched exactly once, by one wave
// DEPTH: rows in flight ahead of the row being contracted (round 5).  1: the rows step y's end needs are fetched at its start
// (one register set; the load latency is exposed behind the step's ~0.5 us of matrix work).  2: they were fetched a step earlier
// (two alternating register sets, the row loop unrolled by two so that every set has ONE issue point and ONE consumption point):
// twice the bytes in flight per CU.  Same arithmetic and summation order, bit-identical slabs.  Not with DT (register budget).
// UP2 (round 5): the second source is the LOW-resolution tensor [B, Cin - Csplit, H / 2, W / 2], up-sampled x2 while its rows are
// staged (up2_staging.hpp; conv_hr16_body's UP2 form is the forward of the same layer): an up-sampling wave fetches 8 bytes per lane
// of two low rows per channel into the registers of a plain wave's one full-row load.
template <int WCI, bool XF, bool STRIP = false, bool DT = false, int DEPTH = 1, bool UP2 = false>
__device__ __forceinline__ void conv_hrwrw_body(const ConvWrwArgs& a) {
    constexpr int WIDTH = 256, NG = WIDTH / 8, XG = NG + 2, NSLOT = 3, ROWS = 16;
    constexpr int CI = 16 * WCI, NWV = 4 * WCI, NTHR = 64 * NWV;
    constexpr int NXL = CI / NWV, NDL = 16 / NWV;        // input / dy channels a wave fetches per row: 4 and 4 (WCI 1) or 4 and 2 (WCI 2)
    constexpr int XCH = NSLOT * XG;                       // units per input channel
    constexpr int X_UNITS = 2 * CI * XCH, D_UNITS = 2 * 16 * NG;
    constexpr int RED_FLOATS = 2 * WCI * 10 * 256;        // cross-wave reduction scratch (aliases the staging image)
    static_assert(RED_FLOATS / 4 <= X_UNITS, "the reduction scratch fits the input image");
    static_assert(WCI == 1 || WCI == 2, "16 or 32 input channels");
    static_assert(DEPTH == 1 || (DEPTH == 2 && !DT && ROWS % DEPTH == 0), "two rows ahead: plain dy only");
    static_assert(!UP2 || (!XF && !STRIP && DEPTH == 1), "up-sampled second source: plain 256-wide form");

    __shared__ __attribute__((aligned(16))) u32x4 sX[X_UNITS];      // [piece][ci][slot][group]
    __shared__ __attribute__((aligned(16))) u32x4 sD[D_UNITS];      // [piece][co][group]

    const f32x2 sd = h16_scale(bound_of(a.dy_bound, a.dy_mul));
    const f32x2 sx = h16_scale(__builtin_fmaxf(bound_of(a.in_bound, a.in_mul), bound_of(a.in2_bound, a.in2_mul)));
    const float sc_d = sd.x, inv_d = sd.y, sc_x = sx.x, inv_x = sx.y;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int sp = wave_u % 4, wci = wave_u / 4;          // segment pair (K split) and input-channel block of this wave's matrix work
    const int HW = a.H * a.W;
    const uint32_t HW4 = (uint32_t)HW * 4u;

    const int split = xcd_swizzle(blockIdx.x, gridDim.x);
    if (split >= a.nsplit) return;
    const int rps = a.H / ROWS, rpi = rps * (STRIP ? a.W / WIDTH : 1), nruns = a.B * rpi;      // run -> (image, strip, 16-row band)
    const int run_begin = (int)((long)nruns * split / a.nsplit), run_end = (int)((long)nruns * (split + 1) / a.nsplit);
    const bool want_bias = a.bslab != nullptr;

    // zero halo groups (image columns -8 .. -1 and 256 .. 263) of every (piece, channel, slot)
    for (int e = tid; e < 2 * CI * NSLOT * 2; e += NTHR) sX[(e / 2) * XG + (e % 2) * (XG - 1)] = u32x4{0u, 0u, 0u, 0u};
    if constexpr (STRIP) __syncthreads();                // the strips' neighbour pixels go into the margin groups: not before the zeros are in

    // ---- staging: lane = pixels 4 lane .. 4 lane + 3 (half of an 8-pixel unit) of NXL input channels and NDL dy channels ----
    // (fetched rows live in 128-bit vector variables: a set that crosses the row loop's back edge is then ONE loop-carried value per
    // load, which the register allocator keeps in place; as scalars it copied them behind the loads, i.e. waited for them at once)
    f32x4 rx[NXL], rd[NDL];
    f32x4 rxb[1];                                        // (unused: an up-sampling wave keeps both low source rows in its set, 8 bytes per lane each)
    float rxw[2] = {0.f, 0.f};                           // UP2: (lh0, lh1), the row weights of the set's pair of source rows
    Up2Lane up_l{};
    if constexpr (UP2) up_l = up2_lane(a.up_rw, lane);
    f32x4 rx2[DEPTH > 1 ? NXL : 1], rd2[DEPTH > 1 ? NDL : 1];      // DEPTH 2: the second set
    float rhx2[1] = {0.f};
    bool x_in2 = false, d_in2 = false, h_in2 = false;
    f32x4 ry[DT ? NDL : 1];                              // DT: the raw conv output beside the gradient row
    float dtc[DT ? NDL : 1][6];                          // DT: (mean, invstd, gamma invstd, beta, mean(d), mean(d x_hat)) of this wave's dy channels
    float rhx[1] = {0.f};                                // STRIP: lanes 0 .. NXL - 1 the pixel left of the strip, NXL .. 2 NXL - 1 the pixel right of it
    f32x2 cf[XF ? NXL : 1];
    f32x2 cf_h = f32x2{0.f, 0.f};                        // STRIP && XF: the coefficients of this lane's margin channel
    bool x_in = false, d_in = false, h_in = false;
    const int xc0 = wave_u * NXL;                        // this wave's input channels xc0 .. xc0 + NXL - 1 lie in one source (Csplit % 4 == 0)
    const bool second = xc0 >= a.Csplit;
    int x0 = 0;                                          // first column of the run's strip
    auto ld4 = [](__amdgpu_buffer_rsrc_t rs, uint32_t off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0)); };
    auto load_x = [&](int b, int gy, f32x4 (&dst)[NXL], bool& ok, float (&hdst)[1], bool& hok, f32x4 (&dstb)[1], float (&lh)[2]) {
        ok = (unsigned)gy < (unsigned)a.H;
        if constexpr (UP2) {
            // one instruction stream for both kinds of wave (conv_hr16_body's UP2 form says why): two 8-byte loads per channel -- the two
            // halves of a plain wave's 16 bytes of the row, or an up-sampling wave's low columns 2 lane, 2 lane + 1 of the source rows
            // h0 and h1 (dst[i] = (row h0: x, y; row h1: z, w); up2_row)
            const int h = a.H / 2, w = a.W / 2;
            const float sy = mul_rn(a.up_rh, (float)gy);
            const int h0 = (int)sy, h1 = h0 + (h0 < h - 1 ? 1 : 0);
            lh[1] = sy - (float)h0; lh[0] = 1.f - lh[1];
            const uint32_t plane4 = second ? (uint32_t)(h * w) * 4u : HW4;
            const __amdgpu_buffer_rsrc_t rs = second ? make_rsrc(a.in2 + (size_t)b * (a.Cin - a.Csplit) * h * w, (uint32_t)(a.Cin - a.Csplit) * plane4)
                                                     : make_rsrc(a.in + (size_t)b * a.Csplit * HW, (uint32_t)a.Csplit * HW4);
            const uint32_t oa = second ? (uint32_t)((xc0 - a.Csplit) * h * w + h0 * w + 2 * lane) * 4u : (uint32_t)(xc0 * HW + gy * a.W + x0 + lane * 4) * 4u;
            const uint32_t ob = second ? (uint32_t)((xc0 - a.Csplit) * h * w + h1 * w + 2 * lane) * 4u : oa + 8u;
#pragma unroll
            for (int i = 0; i < NXL; ++i) {
                const bool v = ok && xc0 + i < a.Cin;
                const f32x2 t0 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(v ? oa + (uint32_t)i * plane4 : kOob), 0, 0));
                const f32x2 t1 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(v ? ob + (uint32_t)i * plane4 : kOob), 0, 0));
                dst[i] = f32x4{t0.x, t0.y, t1.x, t1.y};
            }
            return;
        }
        const __amdgpu_buffer_rsrc_t rs = second ? make_rsrc(a.in2 + (size_t)b * (a.Cin - a.Csplit) * HW, (uint32_t)(a.Cin - a.Csplit) * HW4)
                                                 : make_rsrc(a.in + (size_t)b * a.Csplit * HW, (uint32_t)a.Csplit * HW4);
        const uint32_t off = (uint32_t)((second ? xc0 - a.Csplit : xc0) * HW + gy * a.W + x0 + lane * 4) * 4u;
#pragma unroll
        for (int i = 0; i < NXL; ++i) dst[i] = ld4(rs, (ok && xc0 + i < a.Cin) ? off + (uint32_t)i * HW4 : kOob);
        if constexpr (STRIP) {
            const int hc = lane % NXL, hx = lane < NXL ? x0 - 1 : x0 + WIDTH;
            hok = ok && lane < 2 * NXL && (unsigned)hx < (unsigned)a.W && xc0 + hc < a.Cin;
            buf_load<1>(rs, hok ? (uint32_t)(((second ? xc0 - a.Csplit : xc0) + hc) * HW + gy * a.W + hx) * 4u : kOob, hdst);
        }
    };
    int d_b = 0, d_gy = 0;                               // image and row of the dy row the registers hold (DT: where it is written through)
    auto load_d = [&](int b, int gy, f32x4 (&dst)[NDL], bool& ok) {
        ok = (unsigned)gy < (unsigned)a.H;
        d_b = b; d_gy = gy;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.dout + (size_t)b * a.Cout * HW, (uint32_t)a.Cout * HW4);
#pragma unroll
        for (int i = 0; i < NDL; ++i) {
            const int c = wave_u * NDL + i;
            dst[i] = ld4(rs, (ok && c < a.Cout) ? (uint32_t)(c * HW + gy * a.W + x0 + lane * 4) * 4u : kOob);
        }
        if constexpr (DT) {
            const __amdgpu_buffer_rsrc_t rsy = make_rsrc(a.dt_y + (size_t)b * a.Cout * HW, (uint32_t)a.Cout * HW4);
#pragma unroll
            for (int i = 0; i < NDL; ++i) {
                const int c = wave_u * NDL + i;
                ry[i] = ld4(rsy, (ok && c < a.Cout) ? (uint32_t)(c * HW + gy * a.W + x0 + lane * 4) * 4u : kOob);
            }
        }
    };
    auto store_x = [&](f32x4 (&src)[NXL], bool ok, int slot, float (&hsrc)[1], bool hok, f32x4 (&srcb)[1], float (&lh)[2]) {
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            asm volatile("" : "+v"(src[i]));          // first touch (conv_hp16_body)
            if constexpr (UP2) {
                if (second) {                         // the four up-sampled pixels replace the fetched source pixels (zero outside the image)
                    const up2_f32x4 u = up2_row(up2_f32x2{src[i][0], src[i][1]}, up2_f32x2{src[i][2], src[i][3]}, up_l, lh[0], lh[1]);
                    src[i] = u;                       // (a row outside the image was fetched as zeros: its interpolation is zero)
                }
            }
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] = src[i][k];
                if constexpr (XF) {                    // leaky_relu(fma(y, scale, shift)); rows outside the image stay zero
                    const float z = __builtin_fmaf(v[k], cf[i].x, cf[i].y);
                    v[k] = ok ? __builtin_fmaxf(z, z * a.xf_slope) : 0.f;
                }
            }
            unsigned a0, a1, b0, b1;
            conv_split2h(v[0] * sc_x, v[1] * sc_x, a0, a1);
            conv_split2h(v[2] * sc_x, v[3] * sc_x, b0, b1);
            const int c = wave_u * NXL + i;
            unsigned* p0 = reinterpret_cast<unsigned*>(&sX[(c * NSLOT + slot) * XG + 1]) + lane * 2;      // 8 bytes per lane, consecutive
            unsigned* p1 = reinterpret_cast<unsigned*>(&sX[((CI + c) * NSLOT + slot) * XG + 1]) + lane * 2;
            *reinterpret_cast<uint2*>(p0) = make_uint2(a0, b0);
            *reinterpret_cast<uint2*>(p1) = make_uint2(a1, b1);
        }
        if constexpr (STRIP) {                         // pixel x0 - 1 -> high half of the left margin group's last dword, x0 + 256 -> low half of the right one's first
            asm volatile("" : "+v"(hsrc[0]));
            float v = hsrc[0];
            if constexpr (XF) {
                const float z = __builtin_fmaf(v, cf_h.x, cf_h.y);
                v = hok ? __builtin_fmaxf(z, z * a.xf_slope) : 0.f;
            }
            const bool left = lane < NXL;
            unsigned q0, q1;
            conv_split2h(left ? 0.f : v * sc_x, left ? v * sc_x : 0.f, q0, q1);
            if (lane < 2 * NXL) {
                const int c = wave_u * NXL + lane % NXL;
                const int g = (c * NSLOT + slot) * XG + (left ? 0 : XG - 1), d = left ? 3 : 0;
                reinterpret_cast<unsigned*>(&sX[g])[d] = q0;
                reinterpret_cast<unsigned*>(&sX[g + CI * NSLOT * XG])[d] = q1;
            }
        }
    };
    auto store_d = [&](f32x4 (&src)[NDL]) {
#pragma unroll
        for (int i = 0; i < NDL; ++i) {
            asm volatile("" : "+v"(src[i]));
            float v[4] = {src[i][0], src[i][1], src[i][2], src[i][3]};
            if constexpr (DT) {                        // dy = sc (d - k2 - x_hat k3), d = g or slope g by the sign of the BatchNorm output (norm_act.hip: dpre)
                asm volatile("" : "+v"(ry[i]));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float yc = ry[i][k] - dtc[i][0];
                    const float z = yc * dtc[i][2] + dtc[i][3];
                    const float d = z > 0.f ? v[k] : v[k] * a.dt_slope;
                    v[k] = dtc[i][2] * (d - dtc[i][4] - (yc * dtc[i][1]) * dtc[i][5]);
                }
                const int c = wave_u * NDL + i;
                const __amdgpu_buffer_rsrc_t rso = make_rsrc(a.dt_out + (size_t)d_b * a.Cout * HW, (uint32_t)a.Cout * HW4);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), rso,
                                                       (int)(c < a.Cout ? (uint32_t)(c * HW + d_gy * a.W + x0 + lane * 4) * 4u : kOob), 0, 0);
            }
            unsigned a0, a1, b0, b1;
            conv_split2h(v[0] * sc_d, v[1] * sc_d, a0, a1);
            conv_split2h(v[2] * sc_d, v[3] * sc_d, b0, b1);
            const int c = wave_u * NDL + i;
            unsigned* p0 = reinterpret_cast<unsigned*>(&sD[c * NG]) + lane * 2;
            unsigned* p1 = reinterpret_cast<unsigned*>(&sD[(16 + c) * NG]) + lane * 2;
            *reinterpret_cast<uint2*>(p0) = make_uint2(a0, b0);
            *reinterpret_cast<uint2*>(p1) = make_uint2(a1, b1);
        }
    };

    f32x4 acc[9];
    f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr unsigned kOnes = 0x3C003C00u;               // two fp16 1.0
    const f16x8 ones = __builtin_bit_cast(f16x8, u32x4{kOnes, kOnes, kOnes, kOnes});

    for (int run = run_begin; run < run_end; ++run) {
        const int b = run / rpi, rr = run % rpi, r0 = (rr % rps) * ROWS;
        x0 = (rr / rps) * WIDTH;
        if constexpr (DT) {
#pragma unroll
            for (int i = 0; i < NDL; ++i) {
                const int c = wave_u * NDL + i;
#pragma unroll
                for (int k = 0; k < 6; ++k) dtc[i][k] = c < a.Cout ? a.dt_coef[((size_t)(b / a.dt_Bg) * a.Cout + c) * 8 + k] : 0.f;
            }
        }
        if constexpr (XF && STRIP) {
            const int c = wave_u * NXL + lane % NXL;
            cf_h = f32x2{0.f, 0.f};
            if (c < a.Cin) { const float2 t = a.xf[(size_t)(b / a.xf_Bg) * a.Cin + c]; cf_h = f32x2{t.x, t.y}; }
        }
        if constexpr (XF) {
#pragma unroll
            for (int i = 0; i < NXL; ++i) {
                const int c = wave_u * NXL + i;
                f32x2 v = f32x2{0.f, 0.f};
                if (c < a.Cin) { const float2 t = a.xf[(size_t)(b / a.xf_Bg) * a.Cin + c]; v = f32x2{t.x, t.y}; }
                cf[i] = v;
            }
        }
        // rows r0 - 1 and r0 into slots 0 and 1, then row r0 + 1 (slot 2) and dy row r0: two fetch rounds, both in flight together
        {
            f32x4 ra[NXL], rb[NXL], rab[1], rbb[1];
            float ha[1] = {0.f}, hb[1] = {0.f}, wa[2] = {0.f, 0.f}, wb2[2] = {0.f, 0.f};
            bool oka, okb, hoka = false, hokb = false;
            load_x(b, r0 - 1, ra, oka, ha, hoka, rab, wa);
            load_x(b, r0, rb, okb, hb, hokb, rbb, wb2);
            load_x(b, r0 + 1, rx, x_in, rhx, h_in, rxb, rxw);
            load_d(b, r0, rd, d_in);
            __builtin_amdgcn_sched_barrier(0);
            store_x(ra, oka, 0, ha, hoka, rab, wa);
            store_x(rb, okb, 1, hb, hokb, rbb, wb2);
            store_x(rx, x_in, 2, rhx, h_in, rxb, rxw);
            store_d(rd);
        }
        __syncthreads();
        int s0 = 0;                                       // slot of input row y - 1; rows y, y + 1 follow cyclically
        auto contract = [&]() {                           // dy row (sD) against the three live input rows
#pragma unroll
            for (int sgi = 0; sgi < 2; ++sgi) {
                const int gq = (sp * 2 + sgi) * 4 + kq;  // this lane's 8-pixel group of the row
                f16x8 af[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) af[p] = __builtin_bit_cast(f16x8, sD[(p * 16 + j) * NG + gq]);
                if (want_bias && wci == 0) {              // every dy element exactly once
                    accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], ones, accb, 0, 0, 0);
                    accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], ones, accb, 0, 0, 0);
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    int slot = s0 + ky; slot = slot >= NSLOT ? slot - NSLOT : slot;
                    f16x8 bf[3][2];                        // [shift kx][piece]
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int u = ((p * CI + wci * 16 + j) * NSLOT + slot) * XG + gq + 1;
                        const u32x4 c = sX[u];
                        const unsigned cl3 = reinterpret_cast<const unsigned*>(&sX[u - 1])[3];      // pixels x - 2, x - 1 of the group's first pixel
                        const unsigned cr0 = reinterpret_cast<const unsigned*>(&sX[u + 1])[0];      // pixels x + 8, x + 9
                        const unsigned t01 = __builtin_amdgcn_alignbit(c[1], c[0], 16), t12 = __builtin_amdgcn_alignbit(c[2], c[1], 16);
                        const unsigned t23 = __builtin_amdgcn_alignbit(c[3], c[2], 16);
                        const unsigned tE0 = __builtin_amdgcn_alignbit(c[0], cl3, 16), t3E = __builtin_amdgcn_alignbit(cr0, c[3], 16);
                        bf[0][p] = __builtin_bit_cast(f16x8, u32x4{tE0, t01, t12, t23});           // pixels x - 1
                        bf[1][p] = __builtin_bit_cast(f16x8, c);
                        bf[2][p] = __builtin_bit_cast(f16x8, u32x4{t01, t12, t23, t3E});           // pixels x + 1
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        f32x4 c = acc[ky * 3 + kx];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], bf[kx][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bf[kx][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bf[kx][0], c, 0, 0, 0);
                        acc[ky * 3 + kx] = c;
                    }
                }
            }
        };
        if constexpr (DEPTH == 1) {
#pragma unroll 1
            for (int y = r0; y < r0 + ROWS; ++y) {
                const bool more = y + 1 < r0 + ROWS;
                if (more) { load_x(b, y + 2, rx, x_in, rhx, h_in, rxb, rxw); load_d(b, y + 1, rd, d_in); }
                contract();
                __syncthreads();                          // every wave is done with input row y - 1 and dy row y
                if (more) { store_x(rx, x_in, s0, rhx, h_in, rxb, rxw); store_d(rd); }
                s0 = s0 + 1 >= NSLOT ? 0 : s0 + 1;
                __syncthreads();
            }
        } else {
            // set 1 (rx, rd) serves the even steps of the run, set 2 the odd ones; a set is fetched right behind its own consumption, two
            // steps ahead of the step whose end needs it.  BRANCH-FREE: a row the run does not need is fetched from row -1 (out of
            // range: zeros, no memory access) and stored all the same -- with a load or a store inside a branch the compiler's vmcnt
            // bookkeeping merges the paths and waits for BOTH sets in front of the first store (ISA: tools/diag/isa_outline.py).
            const int rend = r0 + ROWS;
            load_x(b, r0 + 2, rx, x_in, rhx, h_in, rxb, rxw); load_d(b, r0 + 1, rd, d_in);
            __builtin_amdgcn_sched_barrier(0);            // (the two sets' loads stay in this order)
            load_x(b, r0 + 3, rx2, x_in2, rhx2, h_in2, rxb, rxw); load_d(b, r0 + 2, rd2, d_in2);
#pragma unroll 1
            for (int y = r0; y < rend; y += 2) {
                contract();
                __syncthreads();
                store_x(rx, x_in, s0, rhx, h_in, rxb, rxw); store_d(rd);          // input row y + 2, dy row y + 1
                s0 = s0 + 1 >= NSLOT ? 0 : s0 + 1;
                __syncthreads();
                { const bool v = y + 3 < rend; load_x(b, v ? y + 4 : -1, rx, x_in, rhx, h_in, rxb, rxw); load_d(b, v ? y + 3 : -1, rd, d_in); }
                contract();
                __syncthreads();
                store_x(rx2, x_in2, s0, rhx2, h_in2, rxb, rxw); store_d(rd2);     // input row y + 3, dy row y + 2 (zeros behind the run's last step)
                s0 = s0 + 1 >= NSLOT ? 0 : s0 + 1;
                __syncthreads();
                { const bool v = y + 4 < rend; load_x(b, v ? y + 5 : -1, rx2, x_in2, rhx2, h_in2, rxb, rxw); load_d(b, v ? y + 4 : -1, rd2, d_in2); }
            }
        }
    }

    // ---- sum the four K-split partials of each input-channel block through LDS (fixed order), as conv_swrw_body ----
    float* red = reinterpret_cast<float*>(sX);             // [slot][wci][10][4][64]; the staging image is dead (the loop ended with a barrier)
#pragma unroll
    for (int s = 2; s >= 1; s >>= 1) {
        if (sp >= s && sp < 2 * s) {
            float* p = red + (size_t)((sp - s) * WCI + wci) * 10 * 256 + lane;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) p[(t * 4 + r) * 64] = acc[t][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) p[(36 + r) * 64] = accb[r];
        }
        __syncthreads();
        if (sp < s) {
            const float* p = red + (size_t)(sp * WCI + wci) * 10 * 256 + lane;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] += p[(t * 4 + r) * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) accb[r] += p[(36 + r) * 64];
        }
        __syncthreads();
    }
    if (sp != 0) return;
    // lane (j, kq), register r: co = kq * 4 + r, ci = wci * 16 + j; the slab may be narrower than 16 output channels (CoutS = 4: the
    // exact-N plan of the class convolution)
    float* slab = a.slab + (size_t)split * 9 * a.CoutS * a.CinS;
    float chk = 0.f;
    const float inv = inv_d * inv_x;                       // exact: powers of two
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        note_nonfinite(chk, acc[t]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = kq * 4 + r, ci = wci * 16 + j;
            if (co < a.CoutS && ci < a.CinS) slab[((size_t)t * a.CoutS + co) * a.CinS + ci] = acc[t][r] * inv;
        }
    }
    report_nonfinite(a.err, chk, UAPS_ERR_WRW_NONFINITE);
    if (want_bias && wci == 0 && j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (kq * 4 + r < a.CoutS) a.bslab[(size_t)split * a.CoutS + kq * 4 + r] = accb[r] * inv_d;
    }
}

// (the *2 kernels keep two rows in flight ahead of the contraction, DEPTH 2: measured SLOWER in the step -- hrwrw<1> 60.0 -> 71.2 us,
// hrwrw_bn<1> 62.6 -> 68.1 us, profiles/r05_row_kernels_ab.txt -- and launched only under UAPS_TUNE_DEEP_ROWS; bit-identical slabs)
template <int WCI>
__global__ __launch_bounds__(256 * WCI, 2) void conv_hrwrw_kernel(ConvWrwArgs a) { conv_hrwrw_body<WCI, false>(a); }
template <int WCI>
__global__ __launch_bounds__(256 * WCI, 2) void conv_hrwrw_bn_kernel(ConvWrwArgs a) { conv_hrwrw_body<WCI, true>(a); }
template <int WCI>
__global__ __launch_bounds__(256 * WCI, 2) void conv_hrwrw2_kernel(ConvWrwArgs a) { conv_hrwrw_body<WCI, false, false, false, 2>(a); }
template <int WCI>
__global__ __launch_bounds__(256 * WCI, 2) void conv_hrwrw2_bn_kernel(ConvWrwArgs a) { conv_hrwrw_body<WCI, true, false, false, 2>(a); }
// the forms that turn d(activation) into dy while staging (DT, uaps_call_hints::dyt_*)
template <int WCI>
__global__ __launch_bounds__(256 * WCI, 2) void conv_hrwrw_dt_kernel(ConvWrwArgs a) { conv_hrwrw_body<WCI, false, false, true>(a); }
template <int WCI>
__global__ __launch_bounds__(256 * WCI, 2) void conv_hrwrw_bn_dt_kernel(ConvWrwArgs a) { conv_hrwrw_body<WCI, true, false, true>(a); }
template <int WCI>
__global__ __launch_bounds__(256 * WCI, 2) void conv_hrwrww_dt_kernel(ConvWrwArgs a) { conv_hrwrw_body<WCI, false, true, true>(a); }
template <int WCI>
__global__ __launch_bounds__(256 * WCI, 2) void conv_hrwrww_bn_dt_kernel(ConvWrwArgs a) { conv_hrwrw_body<WCI, true, true, true>(a); }
// 16 + 16 input channels with the second 16 up-sampled x2 from the low-resolution tensor while staging (UP2; up4's first convolution)
__global__ __launch_bounds__(512, 2) void conv_hrwrw_up_kernel(ConvWrwArgs a) { conv_hrwrw_body<2, false, false, false, 1, true>(a); }
__global__ __launch_bounds__(512, 2) void conv_hrwrw_up_dt_kernel(ConvWrwArgs a) { conv_hrwrw_body<2, false, false, true, 1, true>(a); }
// the column-strip forms for maps wider than 256 pixels (W % 256 == 0)
template <int WCI>
__global__ __launch_bounds__(256 * WCI, 2) void conv_hrwrww_kernel(ConvWrwArgs a) { conv_hrwrw_body<WCI, false, true>(a); }
template <int WCI>
__global__ __launch_bounds__(256 * WCI, 2) void conv_hrwrww_bn_kernel(ConvWrwArgs a) { conv_hrwrw_body<WCI, true, true>(a); }

}  // namespace uaps
