// 3x3 forward / input-gradient convolutions with <= 16 output channels on large maps (the 256 x 256 level of the U-Net:
// in_conv, up4, out_conv of utilities/UAPS_unet.py:36-44, 110, 138-139, 152 and the input gradients that end in 16 channels),
// two-piece fp16 form of conv_split.hpp.
//
// With 16 output channels a tile's work is 60-108 MFMAs per wave -- a microsecond -- and conv_sfwd_body's one-tile workgroups
// spend their life in the prologue (index arithmetic, weight staging), the exposed latency of their only global-load round and
// the epilogue: measured 17 % matrix-pipe utilisation and 2.5x the HBM time at 16 -> 16 channels, 256 x 256, B = 32.  Here a
// workgroup is persistent over a contiguous run of tiles (walking down 32-pixel column strips, so halo rows are re-read from
// L2), keeps ALL weight fragments of its 16 output channels in registers for its whole life (K = 9 * Cin <= 288: 5 or 9 depth-32
// steps x 2 pieces), fetches the next tile's input while the matrix pipe works on the current one, and touches LDS only for
// the input image (A fragments).  Same arithmetic, layouts and results as conv_sfwd_body<3, 8, 32, 16, 16, XF, true>.
#pragma once
#include "conv_split.hpp"

namespace uaps {

// NCG = input channels / 8 (2 or 4); XF as in conv_fwd_body
template <int NCG, bool XF>
__device__ __forceinline__ void conv_hp16_body(const ConvFwdArgs& a) {
    constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 8, PLANE = IH * IW, XS = 3;
    constexpr int NQ = 9 * NCG, NSTEP = (NQ + 3) / 4;
    // staging: a WAVE stages one channel group (WPC waves share a group's UPC units), so that the source tensor of a two-tensor
    // input and the BatchNorm coefficients are wave-uniform -- with units dealt out by thread id the buffer descriptor differed
    // between the lanes of a wave and every global load became a waterfall loop whose results were waited for at once
    // (round 4: the prefetch of the next tile was not a prefetch at all)
    constexpr int UPR = IW / 4, UPC = IH * UPR, WPC = 4 / NCG, NU = (UPC + 64 * WPC - 1) / (64 * WPC);
    static_assert(NCG == 2 || NCG == 4, "one or two waves per channel group");
    constexpr int MW = 4;                             // M tiles (16 pixels) per wave: 2 rows x 2 halves

    __shared__ __attribute__((aligned(16))) u32x4 sIn[2 * NCG * PLANE];      // [piece][channel group][row][column]
    __shared__ float sRed[16 * 16 * 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int HW = a.H * a.W;
    const uint32_t HW4 = (uint32_t)HW * 4u;
    const int CGP = a.CinP;                           // padded channel groups of the packed split weights

    const f32x2 sc = h16_scale(__builtin_fmaxf(bound_of(a.in_bound, a.in_mul), bound_of(a.in2_bound, a.in2_mul)));
    const float in_scale = sc.x, out_scale_a = sc.y, out_scale_w = a.wscale[1];

    // ---- this workgroup's run of tiles (column-major inside an image) ----
    const int tpi = a.tiles_x * a.tiles_y, ntiles = a.B * tpi;
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x), nblk = gridDim.x;
    const int t_begin = (int)((long)ntiles * bid / nblk), t_end = (int)((long)ntiles * (bid + 1) / nblk);
    if (t_begin >= t_end) return;
    UAPS_STAMP_DECL;      // phases (diagnostic build only): 0 prologue + weight fragments, 1 first tile fetched + stored, per tile: 2 load issue,
                          // 3 matrix loop, 4 epilogue stores, 5 barrier, 6 wait + split + LDS stores, 7 barrier

    // ---- weight fragments, once: lane (n = j, k-group kq) of step s holds k-group q = 4 s + kq = (tap q / NCG, channel group q % NCG) ----
    bf16x8 bfr[NSTEP][2];
    {
        const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(a.wp, (uint32_t)(2 * 9) * CGP * a.CoutP * 16u);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int q = 4 * s + kq, tap = q / NCG, cg = q % NCG;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const uint32_t off = q < NQ ? (uint32_t)(((p * 9 + tap) * CGP + cg) * a.CoutP + j) * 16u : kOob;
                bfr[s][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)off, 0, 0));
            }
        }
    }
    const int co = j;
    const bool co_ok = co < a.Cout;
    const float bv = (a.bias && co_ok) ? a.bias[co] : 0.f;
    const float sh = stats_shift(a, co, co_ok);

    // ---- staging units: 4 consecutive pixels x the 8 channels of this wave's channel group ----
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int ucg = wave_u / WPC;                     // wave-uniform
    int ur[NU], ucu[NU], uloff[NU];
    bool has[NU];
#pragma unroll
    for (int n = 0; n < NU; ++n) {
        const int u = (n * WPC + wave_u % WPC) * 64 + lane;
        has[n] = u < UPC;
        ur[n] = u / UPR; ucu[n] = u % UPR;
        uloff[n] = (ucg * IH + ur[n]) * IW + ucu[n] * 4;
    }
    float rin[NU][8][4];
    f32x2 rxf[XF ? 8 : 1];                   // (scale, shift) of this wave's channels: reloaded only when the statistics group changes
    int xf_group = -1;
    bool uin[NU], uin_next[NU];

    auto tile_of = [&](int t, int& b, int& y0, int& x0, int& part) {
        b = t / tpi;
        const int tt = t - b * tpi, tx = tt / a.tiles_y, ty = tt - tx * a.tiles_y;
        y0 = ty * TH; x0 = tx * TW; part = ty * a.tiles_x + tx;
    };
    auto load_tile = [&](int t) {
        int b, y0, x0, part;
        tile_of(t, b, y0, x0, part);
        const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(a.in + (size_t)b * a.Csplit * HW, (uint32_t)a.Csplit * HW4);
        const __amdgpu_buffer_rsrc_t rs2 = a.Csplit < a.Cin ? make_rsrc(a.in2 + (size_t)b * (a.Cin - a.Csplit) * HW, (uint32_t)(a.Cin - a.Csplit) * HW4) : rs1;
        if constexpr (XF) {
            const int g = b / a.xf_Bg;
            if (g != xf_group) {                      // workgroup-uniform; a run of tiles crosses a group boundary at most once
                xf_group = g;
                const __amdgpu_buffer_rsrc_t rs_xf = make_rsrc(a.xf + (size_t)g * a.Cin, (uint32_t)a.Cin * 8u);
#pragma unroll
                for (int c = 0; c < 8; ++c)           // channels past Cin read (0, 0)
                    rxf[c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_xf, (int)((uint32_t)(ucg * 8 + c) * 8u), 0, 0));
            }
        }
        const int c0 = ucg * 8;
        const bool second = c0 >= a.Csplit;           // wave-uniform: a channel group lies in one source (Csplit % 8 == 0)
        const __amdgpu_buffer_rsrc_t rs = second ? rs2 : rs1;
        const uint32_t cbase = (uint32_t)((second ? c0 - a.Csplit : c0) * HW) * 4u;
#pragma unroll
        for (int n = 0; n < NU; ++n) {
            const int gy = y0 - 1 + ur[n], gx = x0 - 4 + ucu[n] * 4;
            uin_next[n] = has[n] && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W && c0 < a.Cin;      // W % 4 == 0: all 4 pixels in or out
            const uint32_t off = cbase + (uint32_t)(gy * a.W + gx) * 4u;
#pragma unroll
            for (int c = 0; c < 8; ++c) buf_load<4>(rs, uin_next[n] ? off + (uint32_t)c * HW4 : kOob, rin[n][c]);
        }
    };
    // fetched unit -> two fp16 pieces per element (XF: leaky_relu(fma(y, scale, shift)) first; padding stays zero), into LDS
    auto store_tile = [&]() {
#pragma unroll
        for (int n = 0; n < NU; ++n) uin[n] = uin_next[n];       // the tile whose registers are being split
        // the fetched registers are first TOUCHED here: without this the compiler re-arranges them for the packed split arithmetic
        // right behind the loads (v_mov + s_waitcnt vmcnt(0) in front of the matrix loop: the prefetch was waited for at once)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NU; ++n)
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(rin[n][c][0]), "+v"(rin[n][c][1]), "+v"(rin[n][c][2]), "+v"(rin[n][c][3]));
#pragma unroll
        for (int n = 0; n < NU; ++n) {
            if (!has[n]) continue;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                u32x4 p0, p1;
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) {
                    float v0 = rin[n][2 * c2][p], v1 = rin[n][2 * c2 + 1][p];
                    if constexpr (XF) {               // padding pixels stay zero (the coefficients are not zeroed per pixel any more)
                        const float z0 = __builtin_fmaf(v0, rxf[2 * c2].x, rxf[2 * c2].y), z1 = __builtin_fmaf(v1, rxf[2 * c2 + 1].x, rxf[2 * c2 + 1].y);
                        v0 = uin[n] ? __builtin_fmaxf(z0, z0 * a.xf_slope) : 0.f; v1 = uin[n] ? __builtin_fmaxf(z1, z1 * a.xf_slope) : 0.f;
                    }
                    unsigned q0, q1;
                    conv_split2h(v0 * in_scale, v1 * in_scale, q0, q1);
                    p0[c2] = q0; p1[c2] = q1;
                }
                sIn[uloff[n] + p] = p0;
                sIn[NCG * PLANE + uloff[n] + p] = p1;
            }
        }
    };

    // A operand: lane (pixel j, k-group kq) of M tile m reads unit abase[m] + astep[s] (+ piece plane)
    int abase[MW], astep[NSTEP];
#pragma unroll
    for (int m = 0; m < MW; ++m) {
        const int mt = wave * MW + m;
        abase[m] = (mt / 2) * IW + (mt % 2) * 16 + j + XS;
    }
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int q = 4 * s + kq, tap = q / NCG, cg = q % NCG;
        astep[s] = q < NQ ? cg * PLANE + (tap / 3) * IW + (tap % 3) : 0;         // padded k-groups meet zero weights
    }

    UAPS_STAMP(0);
    load_tile(t_begin);
    store_tile();
    __syncthreads();
    UAPS_STAMP(1);
    for (int t = t_begin; t < t_end; ++t) {
        const bool more = t + 1 < t_end;
        if (more) load_tile(t + 1);
        UAPS_STAMP(2);
        UAPS_STAMP_FIRST_MFMA();

        f32x4 acc[MW];
#pragma unroll
        for (int m = 0; m < MW; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NUN = NSTEP * MW;
        bf16x8 af[2][2];
        auto read_a = [&](int u, bf16x8 (&dst)[2]) {
            const int s = u / MW, m = u % MW;
#pragma unroll
            for (int p = 0; p < 2; ++p) dst[p] = __builtin_bit_cast(bf16x8, sIn[p * NCG * PLANE + abase[m] + astep[s]]);
        };
        read_a(0, af[0]);
#pragma unroll
        for (int u = 0; u < NUN; ++u) {
            const int s = u / MW, m = u % MW;
            if (u + 1 < NUN) read_a(u + 1, af[(u + 1) & 1]);
            const auto H = [](const bf16x8& v) { return __builtin_bit_cast(f16x8, v); };
            f32x4 c = acc[m];
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(af[u & 1][1]), H(bfr[s][0]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(af[u & 1][0]), H(bfr[s][1]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(af[u & 1][0]), H(bfr[s][0]), c, 0, 0, 0);
            acc[m] = c;
            if (u + 1 < NUN) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // next unit's DS reads first ...
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                       // ... then this unit's MFMAs
        }

        UAPS_STAMP(3);
        // ---- epilogue: lane (j, kq) holds pixels kq*4..kq*4+3 of channel j of every M tile ----
        int b, y0, x0, part;
        tile_of(t, b, y0, x0, part);
        float st_s = 0.f, st_q = 0.f, chk = 0.f;
        float* out_c = a.out + ((size_t)b * a.Cout + (co_ok ? co : 0)) * HW;
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int mt = wave * MW + m;
            const int gy = y0 + mt / 2, gx = x0 + (mt % 2) * 16 + kq * 4;
            f32x4 v = acc[m];
            v *= out_scale_a; v *= out_scale_w;       // exact: powers of two
            v.x += bv; v.y += bv; v.z += bv; v.w += bv;
            if (co_ok && gy < a.H && gx < a.W) {      // W % 4 == 0: the 4 pixels are all inside or all outside
                note_nonfinite(chk, v);
                *reinterpret_cast<f32x4*>(out_c + (size_t)gy * a.W + gx) = v;
                const f32x4 d = v - sh;
                st_s += (d.x + d.y) + (d.z + d.w);
                st_q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        }
        report_nonfinite(a.err, chk, UAPS_ERR_CONV_NONFINITE);
        if (a.stats != nullptr) {
            sRed[((wave * 4 + kq) * 16 + j) * 2 + 0] = st_s;
            sRed[((wave * 4 + kq) * 16 + j) * 2 + 1] = st_q;
        }
        UAPS_STAMP(4);
        __syncthreads();                              // every wave is done with the LDS image; the partial sums are visible
        UAPS_STAMP(5);
        if (more) store_tile();
        UAPS_STAMP(6);
        if (a.stats != nullptr && tid < 16 && tid < a.Cout) {      // per-tile BatchNorm partial sums, fixed order (see conv_fwd_body)
            float s0 = 0.f, q0 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0 += sRed[(r * 16 + tid) * 2]; q0 += sRed[(r * 16 + tid) * 2 + 1]; }
            a.stats[((size_t)tid * a.B + b) * tpi + part] = make_float2(s0, q0);
        }
        __syncthreads();
        UAPS_STAMP(7);
    }
    UAPS_STAMP_FLUSH();
}

template <int NCG>
__global__ __launch_bounds__(kConvThreads, 2) void conv_hp16_kernel(ConvFwdArgs a) { conv_hp16_body<NCG, false>(a); }
// NCG = 2: three workgroups per CU, as launch_hp16 sizes the grid (the BatchNorm coefficients would otherwise push the kernel
// two registers past the 168 that three waves per SIMD allow -- a third of the grid then waited for a second round)
template <int NCG>
__global__ __launch_bounds__(kConvThreads) __attribute__((amdgpu_waves_per_eu(NCG == 2 ? 3 : 2, NCG == 2 ? 3 : 2)))
void conv_hp16_bn_kernel(ConvFwdArgs a) { conv_hp16_body<NCG, true>(a); }

}  // namespace uaps
