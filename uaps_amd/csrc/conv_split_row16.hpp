// 3x3 forward / input-gradient convolutions with <= 16 output channels on 256-pixel-wide maps (the top level of the U-Net at the
// metric's image size: in_conv, up4, the input gradients that end in 16 channels; utilities/UAPS_unet.py:36-44, 110, 152), two-piece
// fp16 form of conv_split.hpp, FULL-WIDTH ROWS (round 4).
//
// Why.  These layers are HBM-bound (arithmetic intensity ~36 flop / byte), and the 8 x 32-pixel tiles of conv_hp16_body fetch every
// channel's tile as 10 row pieces of 160 bytes 1 KiB apart, each a full cache line plus two 16-byte halo pieces of its neighbours:
// ~20 line requests per KiB.  That pattern alone tops out at 2.1-3.0 TB/s (tools/diag/tile_probe.hip: a copy kernel with the tile
// kernel's loads and stores, no arithmetic), and in-kernel stamps showed the waves of conv_hp16 parked on the ISSUE of their
// loads and stores (5.4 k of 17 k cycles per tile each).  The same bytes as whole rows -- every wave-instruction one 1-KiB row of
// one channel, every row fetched once -- stream at 5.4-5.6 TB/s in the same probe.
//
// How.  A workgroup owns ROWS = 16 consecutive rows of one image at full width and walks down them two rows per step.  The staged
// input lives in an LDS ring of four row slots per channel group (row r in slot (r + 1) & 3), laid out like conv_hp16's image
// ([piece][channel group][slot][column] of 16-byte units = 8 channels of one pixel, 4 zero units of margin on either side: the image
// edge), so the A fragment of 16 consecutive pixels is the same conflict-free ds_read_b128.  Per step the two rows below the
// ring's live rows are fetched into registers (wave = one (row, channel group): 8 loads of 1 KiB), the matrix work of the two output
// rows runs on the four live rows (M tiles of 16 pixels, K = (tap, channel), weight fragments resident in registers as in conv_hp16),
// and the fetched rows replace the two rows that fell out of reach.  No row is fetched twice inside a run (two warm-up rows per run
// of 16), no halo columns exist.  Same arithmetic, k order and results as conv_hp16_body / conv_sfwd_body<3, 8, 32, 16, 16, XF, true>;
// the BatchNorm partial sums keep the 8 x 32-tile layout of those kernels (uaps_conv_fwd_stats_parts does not depend on which runs).
#pragma once
#include "conv_split.hpp"
#include "up2_staging.hpp"

namespace uaps {

// NCG = input channels / 8 (2 or 4); threads = 128 * NCG (one wave per (row of the pair, channel group)); XF as in conv_fwd_body;
// NT = 16-channel output tiles (2: the input gradient of the two-tensor convolution of up4, 16 -> 16 + 16 channels written as two
// tensors, a.out / a.out2 with Osplit = 16; no statistics epilogue in that form)
// STRIP: maps wider than 256 pixels (W % 256 == 0) as 256-wide column strips: a run is 16 rows of one strip, and the one real pixel
// either side of a strip (zero at the image edge) is fetched per row into the margin units that the 256-wide form leaves zero
// UP2 (round 5): the second source (channel groups >= Csplit / 8) is the LOW-resolution tensor [B, Cin - Csplit, H / 2, W / 2]; its rows
// are bilinearly up-sampled x2 while they are staged (up2_staging.hpp) -- the up-sampled half of up4's concatenation is never
// written or re-read.  An up-sampling wave fetches 8 bytes per lane of two low rows per channel (16 loads: the registers of the
// 8 full-row loads of a plain wave).  Bit-identical to the materialised operand.
// BS (round 5): this launch is an input gradient whose output is d(activation) of a BatchNorm(train) + LeakyReLU; instead of the
// forward statistics its epilogue forms that BatchNorm's BACKWARD sums (sum d, sum d x_hat with d = the LeakyReLU-masked gradient:
// norm_act.hip's bn_bwd_sums_max_kernel, a pass of its own over (gradient, y) until now) per 8 x 32-pixel tile into `stats`, and
// the two maxima the finalize needs.  The raw BatchNorm input y is fetched at the top of the step, beside the next row pair.
template <int NCG, bool XF, int NT = 1, bool STRIP = false, bool UP2 = false, bool BS = false>
__device__ __forceinline__ void conv_hr16_body(const ConvFwdArgs& a) {
    constexpr int WIDTH = 256, IW = WIDTH + 8, NSLOT = 4, ROWS = 16, XS = 3;
    constexpr int NWV = 2 * NCG, NTHR = 64 * NWV, HALF = NWV / 2;
    constexpr int NQ = 9 * NCG, NSTEP = (NQ + 3) / 4;
    constexpr int MW = 32 / NWV;                      // M tiles (16 pixels) per wave and step: 2 output rows x 16 tiles over the waves
    constexpr int NTC = MW / 2;                       // 32-pixel statistics columns per wave
    constexpr int CGU = NSLOT * IW, PIECE = NCG * CGU;       // 16-byte units between channel groups / pieces
    static_assert(NCG == 2 || NCG == 4, "16 or 32 input channels");
    static_assert(NT == 1 || (NT == 2 && NCG == 2 && !XF), "two output tiles: 16 input channels, plain input");
    static_assert(!UP2 || (!XF && !STRIP && NT == 1), "up-sampled second source: plain 256-wide single-tile form");
    static_assert(!BS || (!XF && !STRIP && !UP2 && NT == 1), "BatchNorm-backward sums: plain 256-wide single-tile form");

    __shared__ __attribute__((aligned(16))) u32x4 sIn[2 * PIECE];      // [piece][channel group][slot][column]: 67.6 KB (NCG 2), 135 KB (NCG 4)
    __shared__ float sRed[NT == 1 ? NWV * 4 * NTC * 16 * 2 : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int HW = a.H * a.W;
    const uint32_t HW4 = (uint32_t)HW * 4u;
    const int CGP = a.CinP;                           // padded channel groups of the packed split weights

    const f32x2 sc = h16_scale(__builtin_fmaxf(bound_of(a.in_bound, a.in_mul), bound_of(a.in2_bound, a.in2_mul)));
    const float in_scale = sc.x, out_scale_a = sc.y, out_scale_w = a.wscale[1];

    const int rps = a.H / ROWS;                       // runs per strip (H % 16 == 0: the launcher checks)
    const int nstrips = STRIP ? a.W / WIDTH : 1, rpi = rps * nstrips, nruns = a.B * rpi;      // run -> (image, strip, 16-row band): a strip's bands are consecutive
    const int nblk = gridDim.x;
    int run = xcd_swizzle(blockIdx.x, gridDim.x);     // consecutive runs share their two boundary rows: one XCD's L2
    if (run >= nruns) return;

    // ---- weight fragments, once: lane (n = j, k-group kq) of step s holds k-group q = 4 s + kq = (tap q / NCG, channel group q % NCG) ----
    bf16x8 bfr[NSTEP][NT][2];
    {
        const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(a.wp, (uint32_t)(2 * 9) * CGP * a.CoutP * 16u);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int q = 4 * s + kq, tap = q / NCG, cg = q % NCG;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const uint32_t off = q < NQ ? (uint32_t)(((p * 9 + tap) * CGP + cg) * a.CoutP + n * 16 + j) * 16u : kOob;
                    bfr[s][n][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)off, 0, 0));
                }
        }
    }
    bool co_ok[NT];
    float bv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        co_ok[n] = n * 16 + j < a.Cout;
        bv[n] = (a.bias && co_ok[n]) ? a.bias[n * 16 + j] : 0.f;
    }
    const float sh = stats_shift(a, j, co_ok[0]);

    // the margins of every row slot (image columns -4 .. -1 and 256 .. 259) are zero for the kernel's life
    for (int e = tid; e < 2 * NCG * NSLOT * 8; e += NTHR) {
        const int rowslot = e / 8, c = e % 8;
        sIn[rowslot * IW + (c < 4 ? c : IW - 8 + c)] = u32x4{0u, 0u, 0u, 0u};
    }
    if constexpr (STRIP) __syncthreads();             // the strips' neighbour pixels go into margin units: not before the zeros are in

    // ---- staging: this wave stages channel group scg of row (pair base + srr); lane = pixels 4 lane .. 4 lane + 3 ----
    const int scg = wave_u % NCG, srr = wave_u / NCG;
    float rin[8][4];
    float up_lh0 = 0.f, up_lh1 = 0.f;                 // UP2: the row weights of the pair of source rows the registers hold
    Up2Lane up_l{};
    if constexpr (UP2) up_l = up2_lane(a.up_rw, lane);
    float rh[1] = {0.f};                              // STRIP: lanes 0..7 the pixel left of the strip, lanes 8..15 the pixel right of it, channel lane & 7
    f32x2 rxf[XF ? 8 : 1];
    f32x2 rxf_h = f32x2{0.f, 0.f};                    // STRIP && XF: the coefficients of channel lane & 7
    bool uin = false, hin = false;                    // hin (per lane): rh holds a pixel of the image
    int urow = 0;                                     // the row the registers hold
    const int c0 = scg * 8;
    const bool second = c0 >= a.Csplit;               // wave-uniform: a channel group lies in one source (Csplit % 8 == 0)

    auto load_pair = [&](int b, int y1, int x0) {     // rows y1 and y1 + 1 of image b, columns x0 .. x0 + 255
        const int gy = y1 + srr;
        if constexpr (UP2) {
            // ONE instruction stream for both kinds of wave (a load inside a wave-uniform branch comes back through a phi of differently
            // shaped registers, and the copies behind it wait for the prefetch in front of the matrix loop): every wave issues two
            // 8-byte loads per channel -- a plain wave the two halves of its 16 bytes of the row, an up-sampling wave its two low
            // columns 2 lane, 2 lane + 1 of the source rows h0 and h1 (rin[c][0..1] <- row h0, rin[c][2..3] <- row h1; up2_row)
            const int h = a.H / 2, w = a.W / 2;
            urow = gy;
            uin = (unsigned)gy < (unsigned)a.H && c0 < a.Cin;
            const float sy = mul_rn(a.up_rh, (float)gy);
            const int h0 = (int)sy, h1 = h0 + (h0 < h - 1 ? 1 : 0);
            up_lh1 = sy - (float)h0; up_lh0 = 1.f - up_lh1;
            const uint32_t plane4 = second ? (uint32_t)(h * w) * 4u : HW4;
            const __amdgpu_buffer_rsrc_t rs = second ? make_rsrc(a.in2 + (size_t)b * (a.Cin - a.Csplit) * h * w, (uint32_t)(a.Cin - a.Csplit) * plane4)
                                                     : make_rsrc(a.in + (size_t)b * a.Csplit * HW, (uint32_t)a.Csplit * HW4);
            const uint32_t oa = second ? (uint32_t)((c0 - a.Csplit) * h * w + h0 * w + 2 * lane) * 4u : (uint32_t)(c0 * HW + gy * a.W + x0 + lane * 4) * 4u;
            const uint32_t ob = second ? (uint32_t)((c0 - a.Csplit) * h * w + h1 * w + 2 * lane) * 4u : oa + 8u;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x2 t0 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(uin ? oa + (uint32_t)c * plane4 : kOob), 0, 0));
                const f32x2 t1 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(uin ? ob + (uint32_t)c * plane4 : kOob), 0, 0));
                rin[c][0] = t0.x; rin[c][1] = t0.y; rin[c][2] = t1.x; rin[c][3] = t1.y;
            }
            return;
        }
        const __amdgpu_buffer_rsrc_t rs = second ? make_rsrc(a.in2 + (size_t)b * (a.Cin - a.Csplit) * HW, (uint32_t)(a.Cin - a.Csplit) * HW4)
                                                 : make_rsrc(a.in + (size_t)b * a.Csplit * HW, (uint32_t)a.Csplit * HW4);
        urow = gy;
        uin = (unsigned)gy < (unsigned)a.H && c0 < a.Cin;
        const uint32_t off = (uint32_t)((second ? c0 - a.Csplit : c0) * HW + gy * a.W + x0 + lane * 4) * 4u;
#pragma unroll
        for (int c = 0; c < 8; ++c) buf_load<4>(rs, uin ? off + (uint32_t)c * HW4 : kOob, rin[c]);
        if constexpr (STRIP) {
            const int hx = lane < 8 ? x0 - 1 : x0 + WIDTH;
            const bool hok = uin && lane < 16 && (unsigned)hx < (unsigned)a.W && c0 + (lane & 7) < a.Cin;
            hin = hok;
            buf_load<1>(rs, hok ? (uint32_t)(((second ? c0 - a.Csplit : c0) + (lane & 7)) * HW + gy * a.W + hx) * 4u : kOob, rh);
        }
    };
    auto load_xf = [&](int b) {
        if constexpr (XF) {
            const __amdgpu_buffer_rsrc_t rs_xf = make_rsrc(a.xf + (size_t)(b / a.xf_Bg) * a.Cin, (uint32_t)a.Cin * 8u);
#pragma unroll
            for (int c = 0; c < 8; ++c)               // channels past Cin read (0, 0)
                rxf[c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_xf, (int)((uint32_t)(c0 + c) * 8u), 0, 0));
            if constexpr (STRIP)
                rxf_h = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_xf, (int)((uint32_t)(c0 + (lane & 7)) * 8u), 0, 0));
        }
    };
    // fetched row -> two fp16 pieces per element (XF: leaky_relu(fma(y, scale, shift)) first; rows outside the image stay zero), into its slot
    auto store_pair = [&]() {
        // first touch of the fetched registers (see conv_hp16_body: keeps the re-arrangement for the packed arithmetic behind the matrix loop)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(rin[c][0]), "+v"(rin[c][1]), "+v"(rin[c][2]), "+v"(rin[c][3]));
        if constexpr (UP2) {
            if (second) {                             // the four up-sampled pixels of every channel replace the fetched source pixels
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const up2_f32x4 v = up2_row(up2_f32x2{rin[c][0], rin[c][1]}, up2_f32x2{rin[c][2], rin[c][3]}, up_l, up_lh0, up_lh1);
                    rin[c][0] = v[0]; rin[c][1] = v[1]; rin[c][2] = v[2]; rin[c][3] = v[3];      // (a row outside the image was fetched as zeros: its interpolation is zero)
                }
            }
        }
        const int base = (scg * NSLOT + ((urow + 1) & 3)) * IW + 4 + lane * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            u32x4 p0, p1;
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                float v0 = rin[2 * c2][p], v1 = rin[2 * c2 + 1][p];
                if constexpr (XF) {
                    const float z0 = __builtin_fmaf(v0, rxf[2 * c2].x, rxf[2 * c2].y), z1 = __builtin_fmaf(v1, rxf[2 * c2 + 1].x, rxf[2 * c2 + 1].y);
                    v0 = uin ? __builtin_fmaxf(z0, z0 * a.xf_slope) : 0.f; v1 = uin ? __builtin_fmaxf(z1, z1 * a.xf_slope) : 0.f;
                }
                unsigned q0, q1;
                conv_split2h(v0 * in_scale, v1 * in_scale, q0, q1);
                p0[c2] = q0; p1[c2] = q1;
            }
            sIn[base + p] = p0;
            sIn[PIECE + base + p] = p1;
        }
        if constexpr (STRIP) {                        // the strip's two neighbour pixels: channel pair (lane & 6, + 1) -> one dword of the margin unit
            asm volatile("" : "+v"(rh[0]));
            float v = rh[0];
            if constexpr (XF) {
                const float z = __builtin_fmaf(v, rxf_h.x, rxf_h.y);
                v = hin ? __builtin_fmaxf(z, z * a.xf_slope) : 0.f;      // pixels outside the image stay zero
            }
            const float vo = __shfl_xor(v, 1, 64);
            unsigned q0, q1;
            conv_split2h(v * in_scale, vo * in_scale, q0, q1);
            if (lane < 16 && !(lane & 1)) {
                const int unit = (scg * NSLOT + ((urow + 1) & 3)) * IW + (lane < 8 ? 3 : 4 + WIDTH);
                reinterpret_cast<unsigned*>(&sIn[unit])[(lane & 7) >> 1] = q0;
                reinterpret_cast<unsigned*>(&sIn[PIECE + unit])[(lane & 7) >> 1] = q1;
            }
        }
    };

    // ---- matrix work: this wave's output row orr of the pair and its M tiles [tb, tb + MW) ----
    const int orr = wave_u / HALF, tb = (wave_u % HALF) * MW;
    int abase[MW], kxo[NSTEP], kyv[NSTEP];
#pragma unroll
    for (int m = 0; m < MW; ++m) abase[m] = (tb + m) * 16 + j + XS;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int q = 4 * s + kq, qq = q < NQ ? q : 0, tap = qq / NCG, cg = qq % NCG;      // padded k-groups meet zero weights
        kxo[s] = cg * CGU + tap % 3; kyv[s] = tap / 3;
    }
    float st_s[NTC], st_q[NTC];
#pragma unroll
    for (int i = 0; i < NTC; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }
    float bs_md = 0.f, bs_mx = 0.f;                   // BS: max|d|, max|x_hat| of this thread
    float bs_mu = 0.f, bs_is = 0.f, bs_sc = 0.f, bs_sh = 0.f;      // BS: (mean, invstd, gamma invstd, beta) of channel j in the run's statistics group
    float ybuf[BS ? MW : 1][4];                       // BS: the raw BatchNorm input at this step's output pixels
    const int tiles8 = a.H / 8;                       // statistics parts per image: 8-row x 32-pixel tiles, as the tile kernels write them

    const int txs = a.W / 32;                         // statistics tiles per tile row
    {
        const int b = run / rpi, rr = run % rpi;
        load_pair(b, (rr % rps) * ROWS - 1, (rr / rps) * WIDTH);
    }
    for (; run < nruns; run += nblk) {
        const int b = run / rpi, rr = run % rpi, r0 = (rr % rps) * ROWS, x0 = (rr / rps) * WIDTH;
        const bool next_run = run + nblk < nruns;
        load_xf(b);
        if constexpr (BS) {
            const int g = b / a.bs_Bg;
            const bool ok = co_ok[0];
            bs_mu = ok ? a.bs_mean[g * a.Cout + j] : 0.f; bs_is = ok ? a.bs_invstd[g * a.Cout + j] : 0.f;
            bs_sc = ok ? a.bs_gamma[j] * bs_is : 0.f; bs_sh = ok ? a.bs_beta[j] : 0.f;
        }
        store_pair();                                 // rows r0 - 1, r0
        load_pair(b, r0 + 1, x0);
        store_pair();                                 // rows r0 + 1, r0 + 2
        __syncthreads();
        // branch-free output stores (lanes of padded channels store out of range): a store inside a branch makes the compiler's
        // vmcnt bookkeeping conservative, and the wait for the prefetched rows then also waits for this step's stores
        // (channels [0, Osplit) live in a.out, the rest in a.out2; Osplit is 16 or Cout)
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(a.out + (size_t)b * a.Osplit * HW, (uint32_t)a.Osplit * HW4);
        const __amdgpu_buffer_rsrc_t rs_out2 = NT == 2 && a.Osplit < a.Cout
            ? make_rsrc(a.out2 + (size_t)b * (a.Cout - a.Osplit) * HW, (uint32_t)(a.Cout - a.Osplit) * HW4) : rs_out;
        uint32_t out_c[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = n * 16 + j;
            out_c[n] = co_ok[n] ? (uint32_t)(co < a.Osplit ? co : co - a.Osplit) * HW4 : kOob;
        }
#pragma unroll 1
        for (int k = 0; k < ROWS / 2; ++k) {
            const int y = r0 + 2 * k;                 // output rows y, y + 1 from input rows y - 1 .. y + 2
            if constexpr (UP2) {
                // ONE call site (two made the compiler split the load sequence over the branches and wait for all of it in front of
                // the matrix loop): the next pair of this run, the next run's first pair, or -- behind the last run -- rows outside the
                // image (nothing is read)
                const bool more = k + 1 < ROWS / 2;
                const int nr = run + nblk, nrr = nr % rpi;
                load_pair(more ? b : (next_run ? nr / rpi : b), more ? y + 3 : (next_run ? (nrr % rps) * ROWS - 1 : -4), more ? x0 : (nrr / rps) * WIDTH);
                __builtin_amdgcn_sched_barrier(0);    // the loads leave HERE, in front of the matrix loop (straight-line, the scheduler sinks them behind it)
            } else {
            if (k + 1 < ROWS / 2) load_pair(b, y + 3, x0);
            else if (next_run) { const int nr = run + nblk, nrr = nr % rpi; load_pair(nr / rpi, (nrr % rps) * ROWS - 1, (nrr / rps) * WIDTH); }
            }

            if constexpr (BS) {                       // y at this step's output pixels: in flight over the matrix loop, used in the epilogue
                const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(a.bs_y + (size_t)b * a.Cout * HW, (uint32_t)a.Cout * HW4);
#pragma unroll
                for (int m = 0; m < MW; ++m)
                    buf_load<4>(rs_y, out_c[0] == kOob ? kOob : out_c[0] + (uint32_t)((y + orr) * a.W + x0 + (tb + m) * 16 + kq * 4) * 4u, ybuf[m]);
            }
            int aoff[NSTEP];
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) aoff[s] = kxo[s] + ((y + orr + kyv[s]) & 3) * IW;     // input row y + orr + ky - 1 -> its slot
            f32x4 acc[MW][NT];
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            constexpr int NUN = NSTEP * MW;
            bf16x8 af[2][2];
            auto read_a = [&](int u, bf16x8 (&dst)[2]) {
                const int s = u / MW, m = u % MW;
#pragma unroll
                for (int p = 0; p < 2; ++p) dst[p] = __builtin_bit_cast(bf16x8, sIn[p * PIECE + abase[m] + aoff[s]]);
            };
            read_a(0, af[0]);
#pragma unroll
            for (int u = 0; u < NUN; ++u) {
                const int s = u / MW, m = u % MW;
                if (u + 1 < NUN) read_a(u + 1, af[(u + 1) & 1]);
                const auto H = [](const bf16x8& v) { return __builtin_bit_cast(f16x8, v); };
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    f32x4 c = acc[m][n];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(af[u & 1][1]), H(bfr[s][n][0]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(af[u & 1][0]), H(bfr[s][n][1]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(af[u & 1][0]), H(bfr[s][n][0]), c, 0, 0, 0);
                    acc[m][n] = c;
                }
                if (u + 1 < NUN) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // next unit's DS reads first ...
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NT, 0);                  // ... then this unit's MFMAs
            }

            // ---- epilogue: lane (j, kq) holds pixels kq*4..kq*4+3 of channel j of every M tile ----
            const int gy = y + orr;
            float chk = 0.f;
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const int gx = x0 + (tb + m) * 16 + kq * 4;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    f32x4 v = acc[m][n];
                    v *= out_scale_a; v *= out_scale_w;   // exact: powers of two
                    v.x += bv[n]; v.y += bv[n]; v.z += bv[n]; v.w += bv[n];
                    note_nonfinite(chk, v);               // (padded channels hold exact zeros: zero weights, no bias)
                    const bool second_out = NT == 2 && n * 16 >= a.Osplit;      // wave-uniform
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), second_out ? rs_out2 : rs_out,
                                                           (int)(out_c[n] + (uint32_t)(gy * a.W + gx) * 4u), 0, 0);
                    if constexpr (BS) {               // norm_act.hip: dpre / bn_bwd_sums_max_kernel's arithmetic per element; ONE pair of sums per run
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float yc = ybuf[m][e] - bs_mu, z = yc * bs_sc + bs_sh;
                            const float d = z > 0.f ? v[e] : v[e] * a.bs_slope, xh = yc * bs_is;
                            st_s[0] += d; st_q[0] += d * xh;
                            bs_md = __builtin_fmaxf(bs_md, __builtin_fabsf(d)); bs_mx = __builtin_fmaxf(bs_mx, __builtin_fabsf(xh));
                        }
                    } else if constexpr (NT == 1) {
                        const f32x4 d = v - sh;
                        st_s[m / 2] += (d.x + d.y) + (d.z + d.w);
                        st_q[m / 2] += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
                    }
                }
            }
            report_nonfinite(a.err, chk, UAPS_ERR_CONV_NONFINITE);
            const bool band_end = NT == 1 && (BS ? k == ROWS / 2 - 1 : (k & 3) == 3);       // an 8-row band of statistics tiles (BS: the run) is complete
            if (a.stats != nullptr && band_end) {
#pragma unroll
                for (int i = 0; i < NTC; ++i) {
                    sRed[(((wave * 4 + kq) * NTC + i) * 16 + j) * 2 + 0] = st_s[i];
                    sRed[(((wave * 4 + kq) * NTC + i) * 16 + j) * 2 + 1] = st_q[i];
                    st_s[i] = 0.f; st_q[i] = 0.f;
                }
            }
            __syncthreads();                          // every wave is done with the rows that fall out of reach; the partial sums are visible
            if (k + 1 < ROWS / 2) store_pair();
            if constexpr (BS) {
                // one part per (channel, run): the run's sums over all waves and lane groups (only slot i = 0 carries sums), fixed order
                if (a.stats != nullptr && band_end && tid < 16 && tid < a.Cout) {
                    float s0 = 0.f, q0 = 0.f;
#pragma unroll
                    for (int wv = 0; wv < NWV; ++wv)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int e = (((wv * 4 + g) * NTC + 0) * 16 + tid) * 2;
                            s0 += sRed[e]; q0 += sRed[e + 1];
                        }
                    a.stats[((size_t)tid * a.B + b) * (a.H / ROWS) + r0 / ROWS] = make_float2(s0, q0);
                }
            } else
            if (a.stats != nullptr && band_end && tid < 128 && (tid & 15) < a.Cout) {
                // per-tile BatchNorm partial sums, fixed order: the tile's column tc lies with the waves (orr 0 / 1, tc / NTC), 4 lane groups each
                const int tc = tid >> 4, ch = tid & 15, wv = tc / NTC, i = tc % NTC;
                float s0 = 0.f, q0 = 0.f;
#pragma unroll
                for (int o = 0; o < 2; ++o)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int e = ((((o * HALF + wv) * 4 + g) * NTC + i) * 16 + ch) * 2;
                        s0 += sRed[e]; q0 += sRed[e + 1];
                    }
                a.stats[((size_t)ch * a.B + b) * (tiles8 * txs) + ((r0 + 2 * k) / 8) * txs + x0 / 32 + tc] = make_float2(s0, q0);
            }
            __syncthreads();
        }
    }
    if constexpr (BS) {                               // every thread of the workgroup gets here (a workgroup without a run left at the top)
        __shared__ float s_bs[2][16];
        if (!co_ok[0]) { bs_md = 0.f; bs_mx = 0.f; }
        block_amax_to(a.bs_max, bs_md, s_bs[0]);
        block_amax_to(a.bs_max + UAPS_BOUND_FLOATS, bs_mx, s_bs[1]);
    }
}

// Two waves per SIMD for every form: left unbounded the two-output-tile and column-strip forms took 300-400 registers, i.e. ONE
// workgroup per CU where LDS allows two (round 5: conv_hr16x2_kernel 376 -> 228 registers, 113.5 -> 95.3 us in the step).
#define UAPS_HR16_BOUNDS(n) __launch_bounds__(n, 2)
template <int NCG>
__global__ UAPS_HR16_BOUNDS(128 * NCG) void conv_hr16_kernel(ConvFwdArgs a) { conv_hr16_body<NCG, false>(a); }
template <int NCG>
__global__ UAPS_HR16_BOUNDS(128 * NCG) void conv_hr16_bn_kernel(ConvFwdArgs a) { conv_hr16_body<NCG, true>(a); }
// 16 -> 32 channels (two 16-channel output tiles, one or two output tensors), no statistics
__global__ UAPS_HR16_BOUNDS(256) void conv_hr16x2_kernel(ConvFwdArgs a) { conv_hr16_body<2, false, 2>(a); }
// the input gradient that also forms the BatchNorm-backward sums of the layer in front (BS)
__global__ UAPS_HR16_BOUNDS(256) void conv_hr16_bs_kernel(ConvFwdArgs a) { conv_hr16_body<2, false, 1, false, false, true>(a); }
// 16 + 16 -> 16 channels with the second 16 up-sampled x2 from the low-resolution tensor while staging (UP2; up4's first convolution)
__global__ UAPS_HR16_BOUNDS(512) void conv_hr16_up_kernel(ConvFwdArgs a) { conv_hr16_body<4, false, 1, false, true>(a); }
// the column-strip forms for maps wider than 256 pixels (W % 256 == 0)
template <int NCG>
__global__ UAPS_HR16_BOUNDS(128 * NCG) void conv_hr16w_kernel(ConvFwdArgs a) { conv_hr16_body<NCG, false, 1, true>(a); }
template <int NCG>
__global__ UAPS_HR16_BOUNDS(128 * NCG) void conv_hr16w_bn_kernel(ConvFwdArgs a) { conv_hr16_body<NCG, true, 1, true>(a); }
__global__ UAPS_HR16_BOUNDS(256) void conv_hr16wx2_kernel(ConvFwdArgs a) { conv_hr16_body<2, false, 2, true>(a); }

}  // namespace uaps
