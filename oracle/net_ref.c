/*
 * ORACLE (test infrastructure, NOT product code): plain-C float32 restatement of the residual
 * policy-value network forward, one position at a time -- the shape of the reference's
 * evaluator call (batch 1 per playout, policy_value_net_mxnet.py:261-280).
 *
 * PARITY UNPINNED for the arithmetic (MXNet 1.6.0 is not in /root/reference, no weights or
 * recorded outputs exist); it is cross-checked against oracle/net_ref.py (float64) in
 * tests/test_oracle_net.py and used as the `cpu_baseline` ("port", 1 core) of bench.py.
 *
 * Two implementations of the same graph:
 *   ref_net_forward        the plain loops below -- the restatement to read;
 *   ref_fast_*             the same arithmetic laid out for a CPU's vector units (activations [y][x][c],
 *                          weights [tap][c_in][c_out], a 5-pixel x 2-vector register block, FMA over the output
 *                          channels; AVX-512 or AVX2+FMA picked at run time): what a tuned CPU path (the
 *                          reference's MXNet/MKL-DNN build) would spend per leaf, so that bench.py's CPU baseline
 *                          is not a strawman.  tests/test_oracle_net.py holds it to the plain loops.
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


/* ------------------------------------------------------------------------------------------------------------
 * Vectorised forward (15x15-style boards with W % 5 == 0 and F % 32 == 0; anything else: use ref_net_forward).
 * ------------------------------------------------------------------------------------------------------------ */
typedef float v16f __attribute__((vector_size(64), aligned(4)));
typedef float v8f __attribute__((vector_size(32), aligned(4)));

#define DEFINE_CONV_KERNEL(NAME, VT, VW, TARGET)                                                                   \
    __attribute__((target(TARGET))) static void NAME(const float *in, int cin, int F, int H, int W, const float *wp, \
                                                     const float *bias, float *out, int out_stride_px) {           \
        const int PW = W + 2;                                                                                      \
        for (int cb = 0; cb < F; cb += 2 * VW)                                                                     \
            for (int y = 0; y < H; y++)                                                                            \
                for (int x0 = 0; x0 < W; x0 += 5) {                                                                \
                    VT a0[5], a1[5];                                                                               \
                    const VT b0 = *(const VT *)(bias + cb), b1 = *(const VT *)(bias + cb + VW);                    \
                    for (int p = 0; p < 5; p++) { a0[p] = b0; a1[p] = b1; }                                        \
                    for (int kk = 0; kk < 9; kk++) {                                                               \
                        const float *ip = in + ((size_t)(y + kk / 3) * PW + x0 + kk % 3) * cin;                    \
                        const float *w = wp + (size_t)kk * cin * F + cb;                                           \
                        for (int ci = 0; ci < cin; ci++) {                                                         \
                            const VT w0 = *(const VT *)(w + (size_t)ci * F), w1 = *(const VT *)(w + (size_t)ci * F + VW); \
                            for (int p = 0; p < 5; p++) {                                                          \
                                const float s = ip[(size_t)p * cin + ci];                                          \
                                a0[p] += s * w0;                                                                   \
                                a1[p] += s * w1;                                                                   \
                            }                                                                                      \
                        }                                                                                          \
                    }                                                                                              \
                    for (int p = 0; p < 5; p++) {                                                                  \
                        float *o = out + ((size_t)y * W + x0 + p) * out_stride_px + cb;                            \
                        *(VT *)o = a0[p];                                                                          \
                        *(VT *)(o + VW) = a1[p];                                                                   \
                    }                                                                                              \
                }                                                                                                  \
    }

DEFINE_CONV_KERNEL(conv_hwc_avx512, v16f, 16, "avx512f")
DEFINE_CONV_KERNEL(conv_hwc_avx2, v8f, 8, "avx2,fma")

typedef struct {
    int C, F, nb, H, W, isa;        /* isa: 2 = AVX-512, 1 = AVX2+FMA */
    float *blob;                    /* copy of the parameter blob (ref_net_forward's layout) */
    float **wp;                     /* 1 + 2*nb packed 3x3 weights [9][cin][F] */
    float *pad[3];                  /* zero-bordered activations [H+2][W+2][F] */
    float *tmp, *chw;               /* conv output [H][W][F]; final trunk output as [F][H][W] for the heads */
} ref_fast;

void ref_fast_destroy(ref_fast *h) {
    if (!h) return;
    if (h->wp) for (int i = 0; i < 1 + 2 * h->nb; i++) free(h->wp[i]);
    free(h->wp); free(h->blob); free(h->pad[0]); free(h->pad[1]); free(h->pad[2]); free(h->tmp); free(h->chw);
    free(h);
}

static size_t blob_floats(int C, int F, int nb, int hw) {
    return (size_t)F * C * 9 + 5 * F + (size_t)2 * nb * ((size_t)F * F * 9 + 5 * F) + 4 * F + 5 * 4 + (size_t)hw * 4 * hw + hw +
           2 * F + 5 * 2 + 2 * hw + 1;
}

/* returns NULL when the shape or the CPU is not supported (caller falls back to ref_net_forward) */
ref_fast *ref_fast_create(const float *params, int C, int F, int n_blocks, int H, int W) {
    __builtin_cpu_init();
    const int isa = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0;
    if (!isa || W % 5 || F % 32) return NULL;
    ref_fast *h = (ref_fast *)calloc(1, sizeof(ref_fast));
    if (!h) return NULL;
    h->C = C; h->F = F; h->nb = n_blocks; h->H = H; h->W = W; h->isa = isa;
    const size_t nf = blob_floats(C, F, n_blocks, H * W);
    h->blob = (float *)malloc(nf * sizeof(float));
    h->wp = (float **)calloc(1 + 2 * n_blocks, sizeof(float *));
    const size_t padn = (size_t)(H + 2) * (W + 2) * F;
    for (int i = 0; i < 3; i++) h->pad[i] = (float *)calloc(padn, sizeof(float));
    h->tmp = (float *)malloc((size_t)H * W * F * sizeof(float));
    h->chw = (float *)malloc((size_t)H * W * F * sizeof(float));
    if (!h->blob || !h->wp || !h->pad[0] || !h->pad[1] || !h->pad[2] || !h->tmp || !h->chw) { ref_fast_destroy(h); return NULL; }
    memcpy(h->blob, params, nf * sizeof(float));
    const float *p = h->blob;
    for (int l = 0; l < 1 + 2 * n_blocks; l++) {
        const int cin = l == 0 ? C : F;
        const float *w = take(&p, (size_t)F * cin * 9);
        take(&p, 5 * (size_t)F);
        float *q = (float *)malloc((size_t)9 * cin * F * sizeof(float));
        if (!q) { ref_fast_destroy(h); return NULL; }
        for (int o = 0; o < F; o++)
            for (int c = 0; c < cin; c++)
                for (int kk = 0; kk < 9; kk++) q[((size_t)kk * cin + c) * F + o] = w[((size_t)o * cin + c) * 9 + kk];
        h->wp[l] = q;
    }
    return h;
}

static void fast_conv(const ref_fast *h, const float *in, int cin, const float *wp, const float *bias, float *out) {
    if (h->isa == 2) conv_hwc_avx512(in, cin, h->F, h->H, h->W, wp, bias, out, h->F);
    else conv_hwc_avx2(in, cin, h->F, h->H, h->W, wp, bias, out, h->F);
}

/* BatchNorm (unfolded, as ref_net_forward) (+ residual) + ReLU from tmp [H][W][F] into the interior of a padded buffer */
static void fast_bn_relu(const ref_fast *h, const float *g, const float *be, const float *m, const float *v, int fix_gamma,
                         const float *resid_pad, float *dst_pad) {
    const int F = h->F, W = h->W, PW = W + 2;
    float inv[1024], gg[1024];
    for (int c = 0; c < F; c++) { inv[c] = 1.0f / sqrtf(v[c] + BN_EPS); gg[c] = fix_gamma ? 1.0f : g[c]; }
    for (int y = 0; y < h->H; y++)
        for (int x = 0; x < W; x++) {
            const float *t = h->tmp + ((size_t)y * W + x) * F;
            float *d = dst_pad + ((size_t)(y + 1) * PW + x + 1) * F;
            const float *r = resid_pad ? resid_pad + ((size_t)(y + 1) * PW + x + 1) * F : NULL;
            for (int c = 0; c < F; c++) {
                float z = (t[c] - m[c]) * inv[c] * gg[c] + be[c];
                if (r) z += r[c];
                d[c] = z > 0.f ? z : 0.f;
            }
        }
}

int ref_fast_forward(ref_fast *h, const float *planes, float *probs, float *value, float *logits, float *vlogit) {
    const int C = h->C, F = h->F, H = h->H, W = h->W, hw = H * W, PW = W + 2;
    if (F > 1024) return -1;
    const float *p = h->blob;
    /* stem input: CHW planes -> zero-bordered HWC with C channels (pad[2] is reused with stride C) */
    float *in0 = h->pad[2];
    memset(in0, 0, (size_t)(H + 2) * PW * C * sizeof(float));
    for (int c = 0; c < C; c++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) in0[((size_t)(y + 1) * PW + x + 1) * C + c] = planes[((size_t)c * H + y) * W + x];
    float *x = h->pad[0], *t = h->pad[1];
    {
        take(&p, (size_t)F * C * 9);
        const float *b = take(&p, F), *g = take(&p, F), *be = take(&p, F), *m = take(&p, F), *v = take(&p, F);
        fast_conv(h, in0, C, h->wp[0], b, h->tmp);
        fast_bn_relu(h, g, be, m, v, 1, NULL, x);
    }
    float *y2 = h->pad[2];
    memset(y2, 0, (size_t)(H + 2) * PW * F * sizeof(float));      /* borders (it held the stem input) */
    for (int i = 0; i < h->nb; i++) {
        take(&p, (size_t)F * F * 9);
        const float *ba = take(&p, F), *ga = take(&p, F), *bea = take(&p, F), *ma = take(&p, F), *va = take(&p, F);
        fast_conv(h, x, F, h->wp[1 + 2 * i], ba, h->tmp);
        fast_bn_relu(h, ga, bea, ma, va, 0, NULL, t);
        take(&p, (size_t)F * F * 9);
        const float *bb = take(&p, F), *gb = take(&p, F), *beb = take(&p, F), *mb = take(&p, F), *vb = take(&p, F);
        fast_conv(h, t, F, h->wp[2 + 2 * i], bb, h->tmp);
        fast_bn_relu(h, gb, beb, mb, vb, 0, x, y2);
        float *sw = x; x = y2; y2 = sw;
    }
    for (int c = 0; c < F; c++)
        for (int yy = 0; yy < H; yy++)
            for (int xx = 0; xx < W; xx++) h->chw[((size_t)c * H + yy) * W + xx] = x[((size_t)(yy + 1) * PW + xx + 1) * F + c];
    /* heads: the plain loops of ref_net_forward on the [F][H][W] copy */
    float feat[4 * 1024], lg[1024];
    if (hw > 1024) return -1;
    const float *xs = h->chw;
    {
        const float *w = take(&p, (size_t)4 * F), *b = take(&p, 4), *g = take(&p, 4), *be = take(&p, 4);
        const float *m = take(&p, 4), *v = take(&p, 4);
        for (int o = 0; o < 4; o++)
            for (int i = 0; i < hw; i++) {
                float s = b[o];
                for (int c = 0; c < F; c++) s += w[o * F + c] * xs[(size_t)c * hw + i];
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
    {
        const float *w = take(&p, (size_t)2 * F), *b = take(&p, 2), *g = take(&p, 2), *be = take(&p, 2);
        const float *m = take(&p, 2), *v = take(&p, 2);
        for (int o = 0; o < 2; o++)
            for (int i = 0; i < hw; i++) {
                float s = b[o];
                for (int c = 0; c < F; c++) s += w[o * F + c] * xs[(size_t)c * hw + i];
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
    return 0;
}

int ref_fast_isa(const ref_fast *h) { return h ? h->isa : 0; }
