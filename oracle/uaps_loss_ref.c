/* CPU oracle (plain C, double precision) for the UAPS loss block.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Restates UAPS_train.py:186-282 of the reference (softmax per head, mean prediction, per-pixel
 * KL uncertainty, Dirichlet-mixed arg-max pseudo-label, CE + Dice pseudo-supervision weighted by
 * mean(exp(-KL)), uncertainty minimisation) and utilities/pytorch_losses.py:54-89 (dice_loss),
 * plus the closed-form gradient of that block, which the reference gets
 * from autograd.  Pinned by tests/test_oracle_golden.py against tests/golden/g1_*.npz, fixtures
 * produced by importing the reference (tools/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Layouts: logits[k] -> float [B,C,H,W] contiguous; pseudo/labels int64 [B,H,W];
 *          var float [D,B,H,W]; stats double:
 *            unsup: CE[D] | I[D*C] | P[D*C] | cnt[C] | E[D] | V[D]      (raw sums over pixels)
 *            sup:   CE[D] | I[D*C] | P[D*C] | cnt[C]
 * Deviation from the reference, on purpose: where the mean probability m_c underflows to exactly
 * 0 the reference's autograd returns NaN (0/0 in xlogy's derivative); here those terms contribute 0,
 * which is their limit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXD 8
#define MAXC 16

static void softmax_px(const float* base, long cstride, int C, double* p, double* logp) {
    double mx = base[0];
    for (int c = 1; c < C; ++c) if (base[c * cstride] > mx) mx = base[c * cstride];
    double s = 0.0;
    for (int c = 0; c < C; ++c) { p[c] = exp((double)base[c * cstride] - mx); s += p[c]; }
    double ls = log(s);
    for (int c = 0; c < C; ++c) { logp[c] = (double)base[c * cstride] - mx - ls; p[c] /= s; }
}

size_t uaps_ref_unsup_nstats(int D, int C) { return (size_t)(D + 2 * D * C + C + 2 * D); }
size_t uaps_ref_sup_nstats(int D, int C) { return (size_t)(D + 2 * D * C + C); }

/* UAPS_train.py:186-189, 223-255 and the pixel sums needed by 259-277. */
int uaps_ref_unsup_fwd(const float* const* logits, const double* w, int D, int B, int C, int H, int W,
                       int64_t* pseudo, float* var, float* mixed_out, double* stats) {
    if (D < 1 || D > MAXD || C < 1 || C > MAXC) return 1;
    const long HW = (long)H * W, N = (long)B * HW;
    double *CE = stats, *I = CE + D, *P = I + D * C, *cnt = P + D * C, *E = cnt + C, *V = E + D;
    memset(stats, 0, sizeof(double) * uaps_ref_unsup_nstats(D, C));
    for (long n = 0; n < N; ++n) {
        const long b = n / HW, hw = n % HW;
        double p[MAXD][MAXC], lp[MAXD][MAXC], m[MAXC];
        for (int c = 0; c < C; ++c) m[c] = 0.0;
        for (int k = 0; k < D; ++k) {
            softmax_px(logits[k] + (b * C) * HW + hw, HW, C, p[k], lp[k]);
        }
        /* the reference forms fp32 sums left to right; mimic that rounding for the mixture that
           feeds arg-max, keep double for everything else */
        for (int c = 0; c < C; ++c) { double s = 0; for (int k = 0; k < D; ++k) s += p[k][c]; m[c] = s / D; }
        int y = 0; float best = 0.f;
        for (int c = 0; c < C; ++c) {
            float mix = (float)w[0] * (float)p[0][c];
            for (int k = 1; k < D; ++k) mix = mix + (float)w[k] * (float)p[k][c];
            if (mixed_out) mixed_out[(b * C + c) * HW + hw] = mix;
            if (c == 0 || mix > best) { best = mix; y = c; }
        }
        pseudo[n] = y;
        cnt[y] += 1.0;
        for (int k = 0; k < D; ++k) {
            double v = 0.0;
            for (int c = 0; c < C; ++c) {
                if (m[c] > 0.0) v += m[c] * log(m[c]) - m[c] * lp[k][c];
                P[k * C + c] += p[k][c];
            }
            if (var) var[(long)k * N + n] = (float)v;
            V[k] += v;
            E[k] += exp(-v);
            CE[k] += -lp[k][y];
            I[k * C + y] += p[k][y];
        }
    }
    return 0;
}

/* Turns the raw sums into the scalar losses of UAPS_train.py:259-277 / pytorch_losses.py:85-89.
   out: ce[D] | dice[D] | s[D] | Emean[D] | ps_loss | l_uncert */
void uaps_ref_unsup_losses(const double* stats, int D, int C, long N, double eps, double* out) {
    const double *CE = stats, *I = CE + D, *P = I + D * C, *cnt = P + D * C, *E = cnt + C, *V = E + D;
    double ps = 0, lu = 0;
    for (int k = 0; k < D; ++k) {
        double dsum = 0;
        for (int c = 0; c < C; ++c) dsum += 2.0 * I[k * C + c] / (P[k * C + c] + cnt[c] + eps);
        out[k] = CE[k] / N;
        out[D + k] = 1.0 - dsum / C;
        out[2 * D + k] = 0.5 * (out[k] + out[D + k]);
        out[3 * D + k] = E[k] / N;
        ps += out[2 * D + k] * out[3 * D + k];
        lu += V[k];
    }
    out[4 * D] = ps / D;
    out[4 * D + 1] = lu / ((double)N * D);
}

/* d( gscale * (cw1*ps_loss + cw2*l_uncert) ) / d logits, SURVEY.md section 3.4. */
int uaps_ref_unsup_bwd(const float* const* logits, const int64_t* pseudo, const double* stats,
                       double cw1, double cw2, double gscale, double eps,
                       int D, int B, int C, int H, int W, float* const* dlogits) {
    if (D < 1 || D > MAXD || C < 1 || C > MAXC) return 1;
    const long HW = (long)H * W, N = (long)B * HW;
    const double *I = stats + D, *P = I + D * C, *cnt = P + D * C;
    double lo[4 * MAXD + 2];
    uaps_ref_unsup_losses(stats, D, C, N, eps, lo);
    const double *s = lo + 2 * D, *Em = lo + 3 * D;
    for (long n = 0; n < N; ++n) {
        const long b = n / HW, hw = n % HW;
        double p[MAXD][MAXC], lp[MAXD][MAXC], m[MAXC], g[MAXD], h[MAXC];
        for (int k = 0; k < D; ++k) softmax_px(logits[k] + (b * C) * HW + hw, HW, C, p[k], lp[k]);
        for (int c = 0; c < C; ++c) { double t = 0; for (int k = 0; k < D; ++k) t += p[k][c]; m[c] = t / D; }
        const int y = (int)pseudo[n];
        for (int k = 0; k < D; ++k) {
            double v = 0;
            for (int c = 0; c < C; ++c) if (m[c] > 0.0) v += m[c] * log(m[c]) - m[c] * lp[k][c];
            g[k] = cw2 / ((double)N * D) - (cw1 / D) * s[k] * exp(-v) / N;
        }
        for (int c = 0; c < C; ++c) {
            double t = 0;
            if (m[c] > 0.0) for (int k = 0; k < D; ++k) t += g[k] * (log(m[c]) + 1.0 - lp[k][c]);
            h[c] = t / D;
        }
        for (int j = 0; j < D; ++j) {
            double a[MAXC], pa = 0, ph = 0;
            for (int c = 0; c < C; ++c) {
                const double card = P[j * C + c] + cnt[c] + eps;
                a[c] = -(1.0 / C) * (2.0 * (y == c) / card - 2.0 * I[j * C + c] / (card * card));
                pa += p[j][c] * a[c];
                ph += p[j][c] * h[c];
            }
            for (int c = 0; c < C; ++c) {
                double gr = (cw1 / D) * Em[j] * 0.5 * ((p[j][c] - (y == c)) / N + p[j][c] * (a[c] - pa))
                          - g[j] * (m[c] - p[j][c])
                          + p[j][c] * (h[c] - ph);
                dlogits[j][(b * C + c) * HW + hw] = (float)(gscale * gr);
            }
        }
    }
    return 0;
}

/* UAPS_train.py:194-204 pixel sums against ground-truth labels. */
int uaps_ref_sup_fwd(const float* const* logits, const int64_t* labels, int D, int B, int C, int H, int W,
                     double* stats) {
    if (D < 1 || D > MAXD || C < 1 || C > MAXC) return 1;
    const long HW = (long)H * W, N = (long)B * HW;
    double *CE = stats, *I = CE + D, *P = I + D * C, *cnt = P + D * C;
    memset(stats, 0, sizeof(double) * uaps_ref_sup_nstats(D, C));
    for (long n = 0; n < N; ++n) {
        const long b = n / HW, hw = n % HW;
        const int y = (int)labels[n];
        if (y < 0 || y >= C) return 2;
        cnt[y] += 1.0;
        for (int k = 0; k < D; ++k) {
            double p[MAXC], lp[MAXC];
            softmax_px(logits[k] + (b * C) * HW + hw, HW, C, p, lp);
            for (int c = 0; c < C; ++c) P[k * C + c] += p[c];
            CE[k] += -lp[y];
            I[k * C + y] += p[y];
        }
    }
    return 0;
}

/* out: ce[D] | dice[D] | sup   (UAPS_train.py:208-218) */
void uaps_ref_sup_losses(const double* stats, int D, int C, long N, double eps, double* out) {
    const double *CE = stats, *I = CE + D, *P = I + D * C, *cnt = P + D * C;
    double sup = 0;
    for (int k = 0; k < D; ++k) {
        double dsum = 0;
        for (int c = 0; c < C; ++c) dsum += 2.0 * I[k * C + c] / (P[k * C + c] + cnt[c] + eps);
        out[k] = CE[k] / N;
        out[D + k] = 1.0 - dsum / C;
        sup += 0.5 * (out[k] + out[D + k]);
    }
    out[2 * D] = sup / D;
}

int uaps_ref_sup_bwd(const float* const* logits, const int64_t* labels, const double* stats, double gscale,
                     double eps, int D, int B, int C, int H, int W, float* const* dlogits) {
    if (D < 1 || D > MAXD || C < 1 || C > MAXC) return 1;
    const long HW = (long)H * W, N = (long)B * HW;
    const double *I = stats + D, *P = I + D * C, *cnt = P + D * C;
    for (long n = 0; n < N; ++n) {
        const long b = n / HW, hw = n % HW;
        const int y = (int)labels[n];
        for (int j = 0; j < D; ++j) {
            double p[MAXC], lp[MAXC], a[MAXC], pa = 0;
            softmax_px(logits[j] + (b * C) * HW + hw, HW, C, p, lp);
            for (int c = 0; c < C; ++c) {
                const double card = P[j * C + c] + cnt[c] + eps;
                a[c] = -(1.0 / C) * (2.0 * (y == c) / card - 2.0 * I[j * C + c] / (card * card));
                pa += p[c] * a[c];
            }
            for (int c = 0; c < C; ++c) {
                double gr = (0.5 / D) * ((p[c] - (y == c)) / N + p[c] * (a[c] - pa));
                dlogits[j][(b * C + c) * HW + hw] = (float)(gscale * gr);
            }
        }
    }
    return 0;
}

/* utilities/metrics.py:8-61 as one confusion matrix: counts[t*C + p]. arg-max takes the first maximum. */
int uaps_ref_confusion(const float* logits, const int64_t* labels, int B, int C, int H, int W, int64_t* counts) {
    const long HW = (long)H * W, N = (long)B * HW;
    memset(counts, 0, sizeof(int64_t) * C * C);
    for (long n = 0; n < N; ++n) {
        const long b = n / HW, hw = n % HW;
        const float* z = logits + (b * C) * HW + hw;
        int pr = 0;
        for (int c = 1; c < C; ++c) if (z[c * HW] > z[pr * HW]) pr = c;
        const int y = (int)labels[n];
        if (y < 0 || y >= C) return 2;
        counts[y * C + pr] += 1;
    }
    return 0;
}
