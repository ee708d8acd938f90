/*
 * ORACLE (test infrastructure, NOT product code): plain-C float32 restatement of the residual
 * policy-value network forward, one position at a time -- the shape of the reference's
 * evaluator call (batch 1 per playout, policy_value_net_mxnet.py:261-280).
 *
 * PARITY UNPINNED for the arithmetic (MXNet 1.6.0 is not in /root/reference, no weights or
 * recorded outputs exist); it is cross-checked against oracle/net_ref.py (float64) in
 * tests/test_oracle_net.py and used as the `cpu_baseline` ("port", 1 core) of bench.py.
 *
 * Graph: policy_value_net_mxnet.py:70-102.  BatchNorm is applied unfolded:
 *   y = (x - mean) / sqrt(var + 1e-3) * gamma + beta,  gamma := 1 where fix_gamma (stem, heads).
 *
 * Parameter blob layout (float32, contiguous, in this order):
 *   stem:   W[F][C][3][3] b[F] gamma[F] beta[F] mean[F] var[F]
 *   block i (A then B): W[F][F][3][3] b[F] gamma[F] beta[F] mean[F] var[F]
 *   policy: W[4][F] b[4] gamma[4] beta[4] mean[4] var[4]  fcW[HW][4HW] fcb[HW]
 *   value:  W[2][F] b[2] gamma[2] beta[2] mean[2] var[2]  fcW[2HW] fcb[1]
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BN_EPS 1e-3f

static void conv3x3(const float *x, int cin, int cout, int H, int W, const float *w, const float *b, float *y,
                    float *pad) {
    const int PW = W + 2, PH = H + 2;
    memset(pad, 0, sizeof(float) * (size_t)cin * PH * PW);
    for (int c = 0; c < cin; c++)
        for (int i = 0; i < H; i++) memcpy(pad + ((size_t)c * PH + i + 1) * PW + 1, x + ((size_t)c * H + i) * W, sizeof(float) * W);
    for (int o = 0; o < cout; o++) {
        float *yo = y + (size_t)o * H * W;
        for (int i = 0; i < H * W; i++) yo[i] = b[o];
        for (int c = 0; c < cin; c++) {
            const float *wk = w + ((size_t)o * cin + c) * 9;
            const float *pc = pad + (size_t)c * PH * PW;
            for (int ky = 0; ky < 3; ky++)
                for (int kx = 0; kx < 3; kx++) {
                    const float wv = wk[ky * 3 + kx];
                    for (int i = 0; i < H; i++) {
                        const float *pr = pc + (size_t)(i + ky) * PW + kx;
                        float *yr = yo + (size_t)i * W;
                        for (int j = 0; j < W; j++) yr[j] += wv * pr[j];
                    }
                }
        }
    }
}

static void bn(float *y, int c, int hw, const float *gamma, const float *beta, const float *mean, const float *var,
               int fix_gamma) {
    for (int o = 0; o < c; o++) {
        const float g = fix_gamma ? 1.0f : gamma[o];
        const float inv = 1.0f / sqrtf(var[o] + BN_EPS);
        float *yo = y + (size_t)o * hw;
        for (int i = 0; i < hw; i++) yo[i] = (yo[i] - mean[o]) * inv * g + beta[o];
    }
}

static void relu(float *y, int n) {
    for (int i = 0; i < n; i++) y[i] = y[i] > 0.f ? y[i] : 0.f;
}

static const float *take(const float **p, size_t n) {
    const float *r = *p;
    *p += n;
    return r;
}

/* planes [C][H][W] -> probs [HW], value [1]; logits / vlogit optional (may be NULL).
 * returns 0, or -1 on allocation failure. */
int ref_net_forward(const float *params, int C, int F, int n_blocks, int H, int W, const float *planes, float *probs,
                    float *value, float *logits, float *vlogit) {
    const int hw = H * W;
    const int cmax = F > C ? F : C;
    float *x = (float *)malloc(sizeof(float) * (size_t)F * hw);
    float *t = (float *)malloc(sizeof(float) * (size_t)F * hw);
    float *y = (float *)malloc(sizeof(float) * (size_t)F * hw);
    float *pad = (float *)malloc(sizeof(float) * (size_t)cmax * (H + 2) * (W + 2));
    float *feat = (float *)malloc(sizeof(float) * (size_t)4 * hw);
    float *lg = (float *)malloc(sizeof(float) * (size_t)hw);
    if (!x || !t || !y || !pad || !feat || !lg) {
        free(x); free(t); free(y); free(pad); free(feat); free(lg);
        return -1;
    }
    const float *p = params;
    {   /* stem: conv_act, fix_gamma default True */
        const float *w = take(&p, (size_t)F * C * 9), *b = take(&p, F), *g = take(&p, F), *be = take(&p, F);
        const float *m = take(&p, F), *v = take(&p, F);
        conv3x3(planes, C, F, H, W, w, b, x, pad);
        bn(x, F, hw, g, be, m, v, 1);
        relu(x, F * hw);
    }
    for (int i = 0; i < n_blocks; i++) {
        const float *wa = take(&p, (size_t)F * F * 9), *ba = take(&p, F), *ga = take(&p, F), *bea = take(&p, F);
        const float *ma = take(&p, F), *va = take(&p, F);
        conv3x3(x, F, F, H, W, wa, ba, t, pad);
        bn(t, F, hw, ga, bea, ma, va, 0);
        relu(t, F * hw);
        const float *wb = take(&p, (size_t)F * F * 9), *bb = take(&p, F), *gb = take(&p, F), *beb = take(&p, F);
        const float *mb = take(&p, F), *vb = take(&p, F);
        conv3x3(t, F, F, H, W, wb, bb, y, pad);
        bn(y, F, hw, gb, beb, mb, vb, 0);
        for (int k = 0; k < F * hw; k++) y[k] += x[k];
        relu(y, F * hw);
        float *sw = x; x = y; y = sw;
    }
    /* policy head */
    {
        const float *w = take(&p, (size_t)4 * F), *b = take(&p, 4), *g = take(&p, 4), *be = take(&p, 4);
        const float *m = take(&p, 4), *v = take(&p, 4);
        for (int o = 0; o < 4; o++)
            for (int i = 0; i < hw; i++) {
                float s = b[o];
                for (int c = 0; c < F; c++) s += w[o * F + c] * x[(size_t)c * hw + i];
                feat[o * hw + i] = s;
            }
        bn(feat, 4, hw, g, be, m, v, 1);
        relu(feat, 4 * hw);
        const float *fw = take(&p, (size_t)hw * 4 * hw), *fb = take(&p, hw);
        float mx = -INFINITY;
        for (int o = 0; o < hw; o++) {
            float s = fb[o];
            const float *row = fw + (size_t)o * 4 * hw;
            for (int k = 0; k < 4 * hw; k++) s += row[k] * feat[k];
            lg[o] = s;
            if (s > mx) mx = s;
        }
        float sum = 0.f;
        for (int o = 0; o < hw; o++) { probs[o] = expf(lg[o] - mx); sum += probs[o]; }
        for (int o = 0; o < hw; o++) probs[o] /= sum;
        if (logits) memcpy(logits, lg, sizeof(float) * hw);
    }
    /* value head */
    {
        const float *w = take(&p, (size_t)2 * F), *b = take(&p, 2), *g = take(&p, 2), *be = take(&p, 2);
        const float *m = take(&p, 2), *v = take(&p, 2);
        for (int o = 0; o < 2; o++)
            for (int i = 0; i < hw; i++) {
                float s = b[o];
                for (int c = 0; c < F; c++) s += w[o * F + c] * x[(size_t)c * hw + i];
                feat[o * hw + i] = s;
            }
        bn(feat, 2, hw, g, be, m, v, 1);
        relu(feat, 2 * hw);
        const float *fw = take(&p, (size_t)2 * hw), *fb = take(&p, 1);
        float s = fb[0];
        for (int k = 0; k < 2 * hw; k++) s += fw[k] * feat[k];
        if (vlogit) vlogit[0] = s;
        value[0] = tanhf(s);
    }
    free(x); free(t); free(y); free(pad); free(feat); free(lg);
    return 0;
}
