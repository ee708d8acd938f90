// Training-side kernels for everything in the reference's loss graph that is not a 3x3 convolution or a
// BatchNorm (policy_value_net_mxnet.py:85-102, :173-193): the two 1x1 head convolutions, the two FullyConnected
// layers (as one fp32-MFMA GEMM kernel), Dropout(0.5), and the loss
//   mean((z - tanh(u))^2) + mean(-sum(pi * log softmax(logits)))   with the entropy monitor mean(sum(-p log p)),
// forward and backward.  gfx950 only.  (3x3 convolutions: conv3x3_mfma.h / trunk15_wino3.h / conv_train.h /
// wgrad_wino3.h; BatchNorm and Adam: conv_train.h.)
#pragma once
#include <hip/hip_runtime.h>

#include "conv3x3_mfma.h"
#include "conv_train.h"
#include "sampler.h"

namespace apz {

// ---- 1x1 convolution, C_in -> CO (CO <= 8), over n boards of P = H*W pixels.
// x / dx: planes with plane stride ps and row stride rs (dense: ps = P, rs = W; padded rows: 240 / 16); y / dy dense
// [n][CO][P].
// forward: grid (n, ceil(P / 64)); lane = pixel of the workgroup's 64, wave w = the channels w, w + 4, ... (a quarter of
// the dependent load chain each; round 3 had one workgroup per board and one thread per pixel walking all C channels:
// 41 us for 128 boards, all of it load latency); the four partial sums meet in LDS in wave order.
__global__ __launch_bounds__(256) void conv1x1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int C,
                                                          int CO, int H, int W, int ps, int rs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [CO][C] weights, [4][8][64] partial sums
    float* ex = lds + CO * C;
    const int P = H * W, n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.y * 64 + lane;
    for (int i = threadIdx.x; i < CO * C; i += 256) lds[i] = w[i];
    __syncthreads();
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p < P) {
        const float* xb = x + (size_t)n * C * ps + (p / W) * rs + (p % W);
        for (int c = wave; c < C; c += 4) {
            const float v = xb[(size_t)c * ps];
#pragma unroll
            for (int o = 0; o < 8; o++)
                if (o < CO) acc[o] = __builtin_fmaf(lds[o * C + c], v, acc[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < 8; o++) ex[(wave * 8 + o) * 64 + lane] = acc[o];
    __syncthreads();
    if (wave == 0 && p < P)
#pragma unroll
        for (int o = 0; o < 8; o++)
            if (o < CO)
                y[((size_t)n * CO + o) * P + p] =
                    (((ex[o * 64 + lane] + ex[(8 + o) * 64 + lane]) + ex[(16 + o) * 64 + lane]) + ex[(24 + o) * 64 + lane]) +
                    (bias ? bias[o] : 0.f);
}

// backward: dx[n][c][p] = sum_o w[o][c] dy[n][o][p]  (written in x's layout, pad cells zero; `accumulate`: added to
//           what dx holds -- the two heads share one input gradient),
//           part[n][o][c] = sum_p dy[n][o][p] x[n][c][p]  (per-board partial of dw; summed over n in a fixed order by
//           colsum_kernel below: no float atomics, results do not depend on arrival order).
// grid (n, ceil(C / 32)): a workgroup owns 32 channels of a board (one workgroup per board took 92 us for 128 boards:
// 128 workgroups of dependent loads on 256 CUs).  dw partials: wave w owns the channels 8 w .. 8 w + 7 of the group,
// lanes walk the pixels (coalesced), the 64 lane sums meet by shuffles.
// Two heads at once (w2 / dy2 / CO2, CO + CO2 <= 8; CO2 = 0: one): dx = W^T dy + W2^T dy2 in one pass over x and dx (the two
// heads of the reference share their input: separately the second call re-read x and read-modify-wrote dx, 315 MB instead of
// 126 MB per 512 boards); part rows: [n][CO + CO2][C], the second head's below the first's.
__global__ __launch_bounds__(256) void conv1x1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ dy, const float* __restrict__ w2,
                                                          const float* __restrict__ dy2, float* __restrict__ dx,
                                                          float* __restrict__ part, int C, int CO1, int CO2, int H, int W, int ps,
                                                          int rs, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [CO][32] weights of the group, [CO][P] dy
    const int CO = CO1 + CO2;
    const int P = H * W, n = blockIdx.x, c0 = blockIdx.y * 32, nc = min(32, C - c0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* wl = lds;
    float* dl = lds + CO * 32;
    for (int i = threadIdx.x; i < CO * 32; i += 256) {
        const int o = i >> 5, c = i & 31;
        wl[i] = c < nc ? (o < CO1 ? w[o * C + c0 + c] : w2[(o - CO1) * C + c0 + c]) : 0.f;
    }
    for (int i = threadIdx.x; i < CO1 * P; i += 256) dl[i] = dy[(size_t)n * CO1 * P + i];
    for (int i = threadIdx.x; i < CO2 * P; i += 256) dl[CO1 * P + i] = dy2[(size_t)n * CO2 * P + i];
    __syncthreads();
    const float* xb = x + ((size_t)n * C + c0) * ps;
    if (dx) {
        float* dxb = dx + ((size_t)n * C + c0) * ps;
        const int cells = H * rs;                 // cells of a plane, pad cells included (cells <= ps)
        for (int i = threadIdx.x; i < nc * cells; i += 256) {
            const int c = i / cells, q = i - c * cells, row = q / rs, col = q - row * rs;
            float s = 0.f;
            if (col < W) {
                const int p = row * W + col;
#pragma unroll
                for (int o = 0; o < 8; o++)
                    if (o < CO) s = __builtin_fmaf(wl[o * 32 + c], dl[o * P + p], s);
                if (accumulate) s += dxb[(size_t)c * ps + q];
            }
            dxb[(size_t)c * ps + q] = s;
        }
    }
    // all of the wave's loads first (8 channels x up to 4 pixels per lane in flight: one channel after the other, each
    // behind its own shuffle tree, was 8 dependent memory round trips), then the products and the lane sums
    const int ppl = (P + 63) / 64;                // pixels per lane (15x15: 4; 8x8: 1)
    for (int p0 = 0; p0 < ppl; p0 += 4) {         // (boards up to 16x16 in one pass)
        float v[8][4];
        int off[4];
        bool ok[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int p = (p0 + j) * 64 + lane;
            ok[j] = p0 + j < ppl && p < P;
            off[j] = ok[j] ? (p / W) * rs + (p % W) : 0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = min(wave * 8 + k, nc - 1);
#pragma unroll
            for (int j = 0; j < 4; j++) v[k][j] = xb[(size_t)c * ps + off[j]];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = wave * 8 + k;
#pragma unroll
            for (int o = 0; o < 8; o++)
                if (o < CO) {
                    float a = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (ok[j]) a = __builtin_fmaf(dl[o * P + (p0 + j) * 64 + lane], v[k][j], a);
#pragma unroll
                    for (int off2 = 32; off2 > 0; off2 >>= 1) a += __shfl_down(a, off2, 64);
                    if (lane == 0 && c < nc) {
                        float* dst = part + ((size_t)n * CO + o) * C + c0 + c;
                        *dst = p0 ? *dst + a : a;
                    }
                }
        }
    }
}

// out[j] = scale * sum_i in[i][j] over `rows` rows of `cols` floats, in a fixed order (deterministic): a workgroup owns
// 64 columns, wave w the rows w, w + 4, ...; eight loads are in flight per lane and the four partial sums meet in
// LDS in wave order.  (One thread per column walking the rows one dependent load at a time took 20 us for a few KB.)
// Used for dw of the 1x1 convolutions (rows = boards), bias gradients (rows = batch or batch slices) and the loss terms.
__device__ __forceinline__ void colsum_block(const float* in, float* out, int rows, int cols, float scale, int g) {
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = g * 64 + lane;
    double s = 0.0;
    if (j < cols) {
        int i = wave;
        for (; i + 28 < rows; i += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = in[(size_t)(i + 4 * u) * cols + j];
#pragma unroll
            for (int u = 0; u < 8; u++) s += (double)v[u];
        }
        for (; i < rows; i += 4) s += (double)in[(size_t)i * cols + j];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && j < cols) out[j] = (float)((((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) * (double)scale);
}
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols,
                                                     float scale) {
    colsum_block(in, out, rows, cols, scale, blockIdx.x);
}
// part[s][c] = sum over the boards of slice s and the cells of plane c of dy[n][c][.] (planes of `ps` floats: dense, or
// padded rows whose pad cells are zero); grid (C, slices); colsum_kernel then adds the slices in index order.
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ part, int n, int C,
                                                        int ps) {
    __shared__ double sh[4];
    const int c = blockIdx.x, slices = gridDim.y, sl = blockIdx.y;
    const int b0 = (int)((long)n * sl / slices), b1 = (int)((long)n * (sl + 1) / slices);
    double s = 0.0;
    for (int b = b0; b < b1; b++) {
        const float* pl = dy + ((size_t)b * C + c) * ps;
        float t = 0.f;
        for (int i = threadIdx.x; i < ps; i += 256) t += pl[i];
        s += (double)t;
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[(size_t)sl * C + c] = (float)s;
}

// y += x
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, long n4, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        float4 a = ((const float4*)y)[i];
        const float4 b = ((const float4*)x)[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        ((float4*)y)[i] = a;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] += x[n4 * 4 + threadIdx.x];
}

// ---- C[M][N] = sum_k A(m, k) B(k, n) (+ bias[n]) on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 FMA
// chains).  A(m, k) = a[m * a_rs + k * a_cs], B(k, n) = b[k * b_rs + n * b_cs]: the strides express the three products
// of a FullyConnected layer (y = x W^T + b;  dx = dy W;  dW = dy^T x) without transposed copies.
// A workgroup (4 waves) owns a 64 x 64 tile of C; wave w its rows 16w..16w+15, four 16 x 16 MFMA tiles wide.
// The matrices here are small (<= 512 x 900 x 900: 0.2-0.8 GFLOP per product), operands are read straight from
// global memory / L2 per k-step.
// NT = 16-column tiles per wave (workgroup tile 64 x 16 NT): 4 for big outputs, 1 when 64 x 64 tiles would leave most
// CUs idle (the layers here are 512 x 225 x 900: 32 workgroups of 64 x 64).
template <int NT>
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ bias, float* __restrict__ c, int M, int N,
                                                         int K, long a_rs, long a_cs, long b_rs, long b_cs, int ldc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 64 + wave * 16, n0 = blockIdx.x * (16 * NT);
    const int r = lane & 15, kq = lane >> 4;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool mrow = m0 + r < M;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + kq;
        const float av = (mrow && k < K) ? a[(long)(m0 + r) * a_rs + (long)k * a_cs] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int nn = n0 + t * 16 + r;
            const float bv = (nn < N && k < K) ? b[(long)k * b_rs + (long)nn * b_cs] : 0.f;
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
        }
    }
    // D layout: column = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int nn = n0 + t * 16 + r;
        if (nn >= N) continue;
        const float bb = bias ? bias[nn] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int mm = m0 + 4 * kq + g;
            if (mm < M) c[(size_t)mm * ldc + nn] = acc[t][g] + bb;
        }
    }
}

// ... for outputs of few tiles (the policy head's y = x W^T is 128 x 225 x 900 at the reference's batch size: 30
// workgroups of sgemm_mfma_kernel<1>, each wave 225 dependent k-steps of one load pair = 105 us): a workgroup owns ONE
// 16 x 16 tile of C, its four waves split the k-steps (wave w: steps w, w + 4, ...), eight steps' loads are issued
// before their MFMAs, and the four partial tiles meet in LDS in wave order.
__global__ __launch_bounds__(256) void sgemm_ksplit_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ bias, float* __restrict__ c, int M, int N,
                                                           int K, long a_rs, long a_cs, long b_rs, long b_cs, int ldc) {
    __shared__ f32x4 red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int r = lane & 15, kq = lane >> 4;
    const bool mrow = m0 + r < M, ncol = n0 + r < N;
    const float* ap = a + (long)(mrow ? m0 + r : 0) * a_rs;
    const float* bp = b + (long)(ncol ? n0 + r : 0) * b_cs;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8;
    for (int k0 = 4 * wave; k0 < K; k0 += 16 * U) {
        float av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = k0 + 16 * u + kq;
            const bool in = k < K;
            av[u] = ap[(long)(in ? k : 0) * a_cs];
            bv[u] = bp[(long)(in ? k : 0) * b_rs];
            if (!in || !mrow) av[u] = 0.f;
            if (!in || !ncol) bv[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
    if (wave) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave) return;
    acc = ((acc + red[0][lane]) + red[1][lane]) + red[2][lane];
    // D layout: column = lane & 15, row = 4 * (lane >> 4) + reg
    if (!ncol) return;
    const float bb = bias ? bias[n0 + r] : 0.f;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int mm = m0 + 4 * kq + g;
        if (mm < M) c[(size_t)mm * ldc + n0 + r] = acc[g] + bb;
    }
}

// ---- Dropout (policy_value_net_mxnet.py:88,95: p = 0.5 on both flattened head inputs in training mode).  The keep mask
// is a stateless hash of (seed, step, element index): the backward pass regenerates it instead of storing it.
// y = x * keep_mask / keep   (forward and backward are the same map)
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float keep,
                                                      unsigned long long seed, unsigned long long step) {
    const float inv = 1.0f / keep;
    const unsigned long long key = splitmix64(seed ^ splitmix64(step + 0x2545F4914F6CDD1Dull));
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const unsigned long long h = splitmix64(key + (unsigned long long)i * 0x9E3779B97F4A7C15ull);
        const float u = ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
        y[i] = u < keep ? x[i] * inv : 0.f;
    }
}

// ---- loss (policy_value_net_mxnet.py:180-193), one wavefront per sample:
//   p = softmax(logits), v = tanh(u);   terms[i] = ((z - v)^2, -sum_j pi_j log p_j, -sum_j p_j log p_j)
//   dlogits = (p * sum_j pi_j - pi) * gscale,   du = 2 (v - z) (1 - v^2) * gscale        (gscale = 1 / batch: the means)
// probs / values (may be NULL) get p and v (the inference outputs of the same forward).
__global__ __launch_bounds__(256) void pv_loss_kernel(const float* __restrict__ logits, const float* __restrict__ u,
                                                      const float* __restrict__ pi, const float* __restrict__ z, int n, int HW,
                                                      float gscale, float* __restrict__ terms, float* __restrict__ dlogits,
                                                      float* __restrict__ du, float* __restrict__ probs, float* __restrict__ values) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* lr = logits + (size_t)i * HW;
    float m = -INFINITY;
    for (int j = lane; j < HW; j += 64) m = fmaxf(m, lr[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < HW; j += 64) s += expf(lr[j] - m);
    s = wave_sum(s);
    const float logs = logf(s);
    float ce = 0.f, ent = 0.f, spi = 0.f;
    for (int j = lane; j < HW; j += 64) {
        const float lp = lr[j] - m - logs, p = expf(lp);
        const float t = pi ? pi[(size_t)i * HW + j] : 0.f;
        ce -= t * lp;
        ent -= p * lp;
        spi += t;
        if (probs) probs[(size_t)i * HW + j] = p;
    }
    ce = wave_sum(ce);
    ent = wave_sum(ent);
    spi = wave_sum(spi);
    if (dlogits)
        for (int j = lane; j < HW; j += 64) {
            const float p = expf(lr[j] - m - logs);
            dlogits[(size_t)i * HW + j] = (p * spi - pi[(size_t)i * HW + j]) * gscale;
        }
    if (lane == 0) {
        const float v = tanhf(u[i]);
        if (values) values[i] = v;
        if (terms) {
            const float d = (z ? z[i] : 0.f) - v;
            terms[(size_t)i * 3 + 0] = d * d;
            terms[(size_t)i * 3 + 1] = ce;
            terms[(size_t)i * 3 + 2] = ent;
        }
        if (du) du[i] = 2.f * (v - z[i]) * (1.f - v * v) * gscale;
    }
}

}  // namespace apz
