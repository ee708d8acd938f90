// Training-side convolution primitives for gfx950 (first hand-written part):
//
//   pack_conv3x3_kernel   raw [Cout][Cin][3][3] weights (device) -> the MFMA fragment layout of
//                         conv3x3_mfma_kernel; with `transpose_flip` it packs the weights of the
//                         data-gradient convolution  dX = conv(dY, W'),  W'[ci][co][ky][kx] =
//                         W[co][ci][2-ky][2-kx]  (so forward and dgrad share ONE kernel).
//   conv3x3_wgrad_kernel  dW[co][ci][ky][kx] += sum over boards and pixels of
//                         dY[b][co][y][x] * X[b][ci][y+ky-1][x+kx-1]
//                         as fp32 MFMA: D[16 co][16 ci] += A[16 co][4 px] * B[4 px][16 ci] per tap.
//
// wgrad mapping: a workgroup owns TWO 16-channel tiles of C_out and up to 64 input channels (one
// 16-channel tile per wave) for a slice of the batch; a wave keeps 2 x 9 accumulator tiles (one
// per C_out tile and tap, 72 registers).  Per board it stages dY[32][H*W] and X[64][(H+2) x (W+1)
// padded] in LDS; a k-step is 4 consecutive pixels: the A fragments (dY) are read once per k-step
// and reused by the nine taps, the B fragment is the input at the tap's shift (zero halo).
// Plane strides are odd multiples chosen so the 16 channel lanes of a fragment hit 16 distinct
// banks.  Partial sums of the batch slices are combined with float atomics (dW is zeroed by the
// launcher), so the summation order -- and the last bits -- vary from run to run.
#pragma once
#include <hip/hip_runtime.h>

#include "conv3x3_mfma.h"

namespace apz {

// w [cout][cin][3][3] -> wpk [cout'/16][cin'_pad/4][9][64] with (cout', cin') = (cout, cin) or,
// transposed, (cin, cout).  One thread per packed element.
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ wpk, int cin, int cout,
                                    int transpose_flip) {
    const int co_n = transpose_flip ? cin : cout;       // output channels of the packed conv
    const int ci_n = transpose_flip ? cout : cin;       // its input channels
    const int n4 = (ci_n + 3) / 4, ncot = co_n / 16;
    const long total = (long)ncot * n4 * 9 * 64;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        long t = i >> 6;
        const int tap = (int)(t % 9);
        t /= 9;
        const int c4 = (int)(t % n4);
        const int cot = (int)(t / n4);
        const int o = cot * 16 + (lane & 15), c = c4 * 4 + (lane >> 4);
        float v = 0.f;
        if (c < ci_n) {
            if (!transpose_flip) {
                v = w[((size_t)o * cin + c) * 9 + tap];
            } else {
                const int ky = tap / 3, kx = tap - ky * 3;
                v = w[((size_t)c * cin + o) * 9 + (2 - ky) * 3 + (2 - kx)];   // W[co=c][ci=o][2-ky][2-kx]
            }
        }
        wpk[i] = v;
    }
}

// w [128][128][3][3] -> the transformed weights of trunk15_wino3_kernel, U = G g G^T in fp32:
// [cot 8][row half 2][c4 32][lane 64][20] (wino_common.h).  One thread per (cot, pass, c4, lane): 18 values,
// 80 contiguous bytes.  transpose_flip: the weights of the data-gradient convolution (see above).
__global__ void pack_wino2_kernel(const float* __restrict__ w, float* __restrict__ upk, int transpose_flip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 8 * 2 * 32 * 64) return;
    const int lane = i & 63, c4 = (i >> 6) & 31, pass = (i >> 11) & 1, cot = i >> 12;
    const int co = cot * 16 + (lane & 15), ci = c4 * 4 + (lane >> 4);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++)
            g[a][b] = transpose_flip ? w[((size_t)ci * 128 + co) * 9 + (2 - a) * 3 + (2 - b)] : w[((size_t)co * 128 + ci) * 9 + a * 3 + b];
    auto G6 = [](float g0, float g1, float g2, float* u) {
        const float e = g0 + g2, f = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), h = g1 * (1.f / 12.f);
        u[0] = g0 * 0.25f;
        u[1] = (e + g1) * (-1.f / 6.f);
        u[2] = (e - g1) * (-1.f / 6.f);
        u[3] = f + h;
        u[4] = f - h;
        u[5] = g2;
    };
    float t[3][6];                               // t[b][i] = sum_a G[i][a] g[a][b]
#pragma unroll
    for (int b = 0; b < 3; b++) G6(g[0][b], g[1][b], g[2][b], t[b]);
    float out[20];
#pragma unroll
    for (int ii = 0; ii < 3; ii++) {
        const int r = 3 * pass + ii;
        float u[6];
        // (pass is a runtime value: pick the row without a dynamic register index)
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int rr = 0; rr < 6; rr++)
            if (rr == r) t0 = t[0][rr], t1 = t[1][rr], t2 = t[2][rr];
        G6(t0, t1, t2, u);
#pragma unroll
        for (int k = 0; k < 6; k++) out[ii * 6 + k] = u[k];
    }
    out[18] = out[19] = 0.f;
    f32x4* dst = reinterpret_cast<f32x4*>(upk + (size_t)i * 20);
#pragma unroll
    for (int v = 0; v < 5; v++) dst[v] = f32x4{out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]};
}

// dense [planes][15][15] <-> rows16 [planes][15][16] (pad column written as zero)
__global__ void rows16_from_dense_kernel(const float* __restrict__ x, float* __restrict__ y, long planes) {
    const long total = planes * 240;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pl = i / 240;
        const int rem = (int)(i - pl * 240), row = rem >> 4, col = rem & 15;
        y[i] = col < 15 ? x[pl * 225 + row * 15 + col] : 0.f;
    }
}
__global__ void rows16_to_dense_kernel(const float* __restrict__ x, float* __restrict__ y, long planes) {
    const long total = planes * 225;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pl = i / 225;
        const int rem = (int)(i - pl * 225), row = rem / 15, col = rem - row * 15;
        y[i] = x[pl * 240 + row * 16 + col];
    }
}

// ---- training-mode BatchNorm (+ residual, + ReLU) over [n][C][H][W] planes with plane stride PS and row
// stride RS (dense NCHW: PS = H*W, RS = W; the trunk's padded-row layout: PS = 240, RS = 16).  Statistics are
// over the n*H*W valid elements of a channel; pad columns (x >= W) are written as zero and never read.
//   bn_stats_kernel      sums[c] += (sum x, sum x^2) in double (atomics; zeroed by the launcher); grid (C, splits)
//   bn_finalize_kernel   mean, 1/sqrt(var + eps) per channel; moving statistics (torch semantics: unbiased
//                        variance in the moving average, momentum = weight of the NEW value)
//   bn_apply_kernel      y = act((x - mean) * invstd * gamma + beta (+ resid))
//   bn_bwd_reduce_kernel sums[c] += (sum dz, sum dz * xhat), dz = dy (* [out > 0] with ReLU)
//   bn_bwd_apply_kernel  dx = gamma * invstd * (dz - s1/M - xhat * s2/M); dres = dz
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, double* __restrict__ sums, int n, int C,
                                                       int PS, int RS, int H, int W) {
    __shared__ double sh[4];
    const int c = blockIdx.x, HW = H * W;
    double s1 = 0.0, s2 = 0.0;
    for (int b = blockIdx.y; b < n; b += gridDim.y) {
        const float* pl = x + ((size_t)b * C + c) * PS;
        for (int p = threadIdx.x; p < HW; p += 256) {
            const int y = p / W;
            const float v = pl[y * RS + (p - y * W)];
            s1 += v;
            s2 += (double)v * v;
        }
    }
    s1 = block_sum_256(s1, sh);
    s2 = block_sum_256(s2, sh);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * c], s1);
        atomicAdd(&sums[2 * c + 1], s2);
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ run_mean, float* __restrict__ run_var, int C, double M, float eps,
                                   float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sums[2 * c] / M;
    double var = sums[2 * c + 1] / M - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) run_mean[c] = run_mean[c] * (1.f - momentum) + momentum * (float)m;
    if (run_var) run_var[c] = run_var[c] * (1.f - momentum) + momentum * (float)(M > 1.0 ? var * M / (M - 1.0) : var);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ resid,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       float* __restrict__ y, long planes, int C, int PS, int RS, int W,
                                                       int relu) {
    const long total = planes * PS;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pl = i / PS;
        const int rem = (int)(i - pl * PS), col = rem % RS, c = (int)(pl % C);
        float v = 0.f;
        if (col < W) {
            const float g = gamma ? gamma[c] : 1.f;
            v = (x[i] - mean[c]) * invstd[c] * g + beta[c];
            if (resid) v += resid[i];
            if (relu) v = fmaxf(v, 0.f);
        }
        y[i] = v;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ out, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, double* __restrict__ sums,
                                                            int n, int C, int PS, int RS, int H, int W, int relu) {
    __shared__ double sh[4];
    const int c = blockIdx.x, HW = H * W;
    const float m = mean[c], is = invstd[c];
    double s1 = 0.0, s2 = 0.0;
    for (int b = blockIdx.y; b < n; b += gridDim.y) {
        const size_t base = ((size_t)b * C + c) * PS;
        for (int p = threadIdx.x; p < HW; p += 256) {
            const int y = p / W;
            const size_t i = base + y * RS + (p - y * W);
            float dz = dy[i];
            if (relu && !(out[i] > 0.f)) dz = 0.f;
            s1 += dz;
            s2 += (double)dz * ((x[i] - m) * is);
        }
    }
    s1 = block_sum_256(s1, sh);
    s2 = block_sum_256(s2, sh);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * c], s1);
        atomicAdd(&sums[2 * c + 1], s2);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ out, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const double* __restrict__ sums, float* __restrict__ dx,
                                                           float* __restrict__ dres, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, long planes, int C, int PS, int RS,
                                                           int W, int relu, double M) {
    const long total = planes * PS;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pl = i / PS;
        const int rem = (int)(i - pl * PS), col = rem % RS, c = (int)(pl % C);
        float gx = 0.f, gr = 0.f;
        if (col < W) {
            float dz = dy[i];
            if (relu && !(out[i] > 0.f)) dz = 0.f;
            const float xhat = (x[i] - mean[c]) * invstd[c];
            const float g = gamma ? gamma[c] : 1.f;
            gx = g * invstd[c] * (dz - (float)(sums[2 * c] / M) - xhat * (float)(sums[2 * c + 1] / M));
            gr = dz;
        }
        dx[i] = gx;
        if (dres) dres[i] = gr;
    }
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (dbeta) dbeta[c] = (float)sums[2 * c];
            if (dgamma) dgamma[c] = (float)sums[2 * c + 1];
        }
}

// ---- Adam as Module.init_optimizer configures it in the reference (policy_value_net_mxnet.py:198-205): ONE launch over
// every trainable tensor (blockIdx.y = tensor).  g = grad * rescale + wd * w;  m = b1 m + (1-b1) g;
// v = b2 v + (1-b2) g^2;  w -= lr_t * m / (sqrt(v) + eps), lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) from the host.
struct AdamTensor {
    float* w;
    const float* g;
    float* m;
    float* v;
    long long n;
    float wd;
    int pad_;
};
__global__ __launch_bounds__(256) void adam_step_kernel(const AdamTensor* __restrict__ tab, float lr_t, float b1, float b2,
                                                        float eps, float rescale) {
    const AdamTensor T = tab[blockIdx.y];
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < T.n; i += (long long)gridDim.x * blockDim.x) {
        const float w = T.w[i];
        const float g = T.g[i] * rescale + T.wd * w;
        const float m = T.m[i] * b1 + (1.f - b1) * g;
        const float v = T.v[i] * b2 + (1.f - b2) * (g * g);
        T.m[i] = m;
        T.v[i] = v;
        T.w[i] = w - lr_t * m / (sqrtf(v) + eps);
    }
}

// ---- the same four kernels for the padded-row layout (plane = 60 float4; pad elements are zero on input, so
// they add nothing to any sum, and are written back as zero): 16-byte accesses, no per-element index division
__global__ __launch_bounds__(256) void bn_stats_r16_kernel(const float* __restrict__ x, double* __restrict__ sums, int n, int C) {
    __shared__ double sh[4];
    const int c = blockIdx.x, t = threadIdx.x, sub = t / 60, k = t - sub * 60;   // 4 boards per trip, 60 float4 each
    double s1 = 0.0, s2 = 0.0;
    if (sub < 4)
        for (int b = blockIdx.y * 4 + sub; b < n; b += gridDim.y * 4) {
            const f32x4 v = reinterpret_cast<const f32x4*>(x + ((size_t)b * C + c) * 240)[k];
            s1 += (double)((v[0] + v[1]) + (v[2] + v[3]));
            s2 += (double)((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
        }
    s1 = block_sum_256(s1, sh);
    s2 = block_sum_256(s2, sh);
    if (t == 0) {
        atomicAdd(&sums[2 * c], s1);
        atomicAdd(&sums[2 * c + 1], s2);
    }
}

__global__ __launch_bounds__(256) void bn_apply_r16_kernel(const float* __restrict__ x, const float* __restrict__ resid,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           float* __restrict__ y, long planes, int C, int relu) {
    const long total = planes * 60;
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < total; v += (long)gridDim.x * blockDim.x) {
        const long pl = v / 60;
        const int c = (int)(pl % C);
        const float sc = invstd[c] * (gamma ? gamma[c] : 1.f), sh = beta[c] - mean[c] * sc;
        f32x4 a = reinterpret_cast<const f32x4*>(x)[v];
        a = a * sc + sh;
        if (resid) a += reinterpret_cast<const f32x4*>(resid)[v];
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; e++) a[e] = fmaxf(a[e], 0.f);
        }
        if ((v & 3) == 3) a[3] = 0.f;            // the pad column (60 float4 per plane: 4 per row)
        reinterpret_cast<f32x4*>(y)[v] = a;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_r16_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ out, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, double* __restrict__ sums,
                                                                int n, int C, int relu) {
    __shared__ double sh[4];
    const int c = blockIdx.x, t = threadIdx.x, sub = t / 60, k = t - sub * 60;
    const float m = mean[c], is = invstd[c];
    double s1 = 0.0, s2 = 0.0;
    if (sub < 4)
        for (int b = blockIdx.y * 4 + sub; b < n; b += gridDim.y * 4) {
            const size_t o = ((size_t)b * C + c) * 60 + k;
            f32x4 g = reinterpret_cast<const f32x4*>(dy)[o];
            const f32x4 xv = reinterpret_cast<const f32x4*>(x)[o];
            if (relu) {
                const f32x4 ov = reinterpret_cast<const f32x4*>(out)[o];
#pragma unroll
                for (int e = 0; e < 4; e++) g[e] = ov[e] > 0.f ? g[e] : 0.f;
            }
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                a1 += g[e];
                a2 += g[e] * ((xv[e] - m) * is);
            }
            s1 += a1;
            s2 += a2;
        }
    s1 = block_sum_256(s1, sh);
    s2 = block_sum_256(s2, sh);
    if (t == 0) {
        atomicAdd(&sums[2 * c], s1);
        atomicAdd(&sums[2 * c + 1], s2);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_r16_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ out, const float* __restrict__ gamma,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const double* __restrict__ sums, float* __restrict__ dx,
                                                               float* __restrict__ dres, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, long planes, int C, int relu, double M) {
    const long total = planes * 60;
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < total; v += (long)gridDim.x * blockDim.x) {
        const long pl = v / 60;
        const int c = (int)(pl % C);
        const float m = mean[c], is = invstd[c], k0 = (float)(sums[2 * c] / M), k1 = (float)(sums[2 * c + 1] / M);
        const float gi = (gamma ? gamma[c] : 1.f) * is;
        f32x4 g = reinterpret_cast<const f32x4*>(dy)[v];
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[v];
        if (relu) {
            const f32x4 ov = reinterpret_cast<const f32x4*>(out)[v];
#pragma unroll
            for (int e = 0; e < 4; e++) g[e] = ov[e] > 0.f ? g[e] : 0.f;
        }
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; e++) r[e] = gi * (g[e] - k0 - (xv[e] - m) * is * k1);
        if ((v & 3) == 3) r[3] = 0.f, g[3] = 0.f;
        reinterpret_cast<f32x4*>(dx)[v] = r;
        if (dres) reinterpret_cast<f32x4*>(dres)[v] = g;
    }
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (dbeta) dbeta[c] = (float)sums[2 * c];
            if (dgamma) dgamma[c] = (float)sums[2 * c + 1];
        }
}

template <int H, int W, bool R16 = false>
struct WgradGeo {
    static constexpr int WP = R16 ? 16 : W;                         // stored row length (padded-row layout: 16)
    static constexpr int HW = H * WP;                               // stored plane size = pixels walked by the k-steps
    static constexpr int RS = W + 1;
    static constexpr int XPLANE = (H + 2) * RS + 2;
    static constexpr int XPS = XPLANE | 1;                          // odd: 16 channel lanes -> 16 banks
    static constexpr int KSTEPS = (HW + 3) / 4;
    static constexpr int YPS = (KSTEPS * 4) | 1;                    // odd, >= padded pixel count
    static constexpr int COT = 2;                                   // 16-channel C_out tiles per workgroup
    static constexpr int TILE_FLOATS = 64 * XPS + COT * 16 * YPS + 64;
    static constexpr int EPI_FLOATS = COT * 16 * (64 * 9 + 1);      // epilogue staging [co][ci*9 (+1)]
    static constexpr int LDS_FLOATS = TILE_FLOATS > EPI_FLOATS ? TILE_FLOATS : EPI_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

// x [n][cin][H][W], dy [n][cout][H][W] dense; dw [cout][cin][3][3] (pre-zeroed, atomically added).
// grid: (cout/32, ceil(cin/64), batch slices).  The staged input tile (64 channels) is shared by
// the workgroup's two C_out tiles; a wave holds 2 x 9 accumulator tiles (72 registers).
// Staging is software-pipelined through registers: the NEXT board's 16-byte pieces are loaded
// before this board's MFMA loop and scattered into LDS after it, so HBM/L2 latency hides behind
// ~1000 MFMAs per wave.  The epilogue transposes the accumulators through LDS so that every
// atomic wave-instruction adds 64 consecutive floats of one dW row.
// R16: x and dy in the trunk's padded-row layout [n][C][15][16] (pad column zero: it adds nothing to the sums)
template <int H, int W, bool R16 = false>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dw, int n, int cin, int cout) {
    using G = WgradGeo<H, W, R16>;
    constexpr int WP = G::WP;
    constexpr int HW = G::HW, COT = G::COT;
    constexpr int XV = (64 * HW / 4 + 255) / 256;        // float4 pieces per thread: input tile
    constexpr int YV = (COT * 16 * HW / 4 + 255) / 256;  // dY tile
    static_assert((64 * HW) % 4 == 0 && (COT * 16 * HW) % 4 == 0, "tiles are whole float4s");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xt = sm + 32;                    // [64][XPS]
    float* yt = sm + 32 + 64 * G::XPS;      // [COT*16][YPS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int co0 = blockIdx.x * 16 * COT;
    const int ci0 = blockIdx.y * 64;
    const int nci = min(64, cin - ci0);     // input channels of this workgroup (may be < 64)
    const int nx4 = nci * HW / 4 + ((nci * HW) % 4 ? 1 : 0);

    for (int i = tid; i < G::TILE_FLOATS; i += 256) sm[i] = 0.f;  // halos / unused channels / tail pixels stay 0

    f32x4 acc[COT][9];
#pragma unroll
    for (int c = 0; c < COT; c++)
#pragma unroll
        for (int t = 0; t < 9; t++) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* arow = yt + j * G::YPS + q;                 // A: co = j (+16 per tile), pixel = 4s + q
    const float* brow = xt + (wave * 16 + j) * G::XPS;       // B: ci = wave*16 + j

    f32x4 px[XV], py[YV];
    auto fetch = [&](int b) {               // global -> registers (16-byte pieces; tiles are 16-B aligned)
        const f32x4* xb = reinterpret_cast<const f32x4*>(x + ((size_t)b * cin + ci0) * HW);
        const f32x4* yb = reinterpret_cast<const f32x4*>(dy + ((size_t)b * cout + co0) * HW);
        const bool xal = ((((size_t)b * cin + ci0) * HW) & 3) == 0 && (nci * HW) % 4 == 0;
#pragma unroll
        for (int u = 0; u < XV; u++) {
            const int v = tid + u * 256;
            px[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (b < n && v < nx4) {
                if (xal) {
                    px[u] = xb[v];
                } else {                    // C_in = 9 stem: planes are not 16-byte aligned
                    const float* xs = x + ((size_t)b * cin + ci0) * HW;
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (v * 4 + e < nci * HW) px[u][e] = xs[v * 4 + e];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < YV; u++) {
            const int v = tid + u * 256;
            py[u] = (b < n && v < COT * 16 * HW / 4) ? yb[v] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto scatter = [&]() {                  // registers -> padded LDS tiles
#pragma unroll
        for (int u = 0; u < XV; u++) {
            const int v = tid + u * 256;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int idx = v * 4 + e;
                if (idx < nci * HW) {
                    const int c = idx / HW, rem = idx - c * HW;
                    const int yy = rem / WP, xx = rem - yy * WP;
                    xt[c * G::XPS + (yy + 1) * G::RS + xx + 1] = px[u][e];   // (a pad element lands on the next row's zero halo)
                }
            }
        }
#pragma unroll
        for (int u = 0; u < YV; u++) {
            const int v = tid + u * 256;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int idx = v * 4 + e;
                if (idx < COT * 16 * HW) {
                    const int c = idx / HW;
                    yt[c * G::YPS + idx - c * HW] = py[u][e];
                }
            }
        }
    };

    fetch(blockIdx.z);
    for (int b = blockIdx.z; b < n; b += gridDim.z) {
        __syncthreads();                     // previous board's tiles fully consumed (and the zero fill done)
        scatter();
        __syncthreads();
        fetch(b + gridDim.z);                // in flight during the MFMA loop below
        if (wave * 16 < nci) {               // wave-uniform: this wave's 16 input channels exist
            int yy = 0, xx = q;              // pixel 4s + q as (row, col), advanced incrementally (W >= 4)
#pragma unroll 3
            for (int s = 0; s < G::KSTEPS; s++) {
                float a[COT];
#pragma unroll
                for (int c = 0; c < COT; c++) a[c] = arow[c * 16 * G::YPS + 4 * s];
                const int yc = yy < H ? yy : H - 1;          // tail lanes: dY there is 0, stay in bounds
                const float* bp = brow + yc * G::RS + xx;
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        const float bv = bp[ky * G::RS + kx];
#pragma unroll
                        for (int c = 0; c < COT; c++)
                            acc[c][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], bv, acc[c][ky * 3 + kx], 0, 0, 0);
                    }
                xx += 4;
                if (xx >= WP) { xx -= WP; yy++; }
            }
        }
    }
    // ---- epilogue: accumulators -> LDS as [co 32][ci 64][9] (row = 576 floats, +1 pad), then every
    // wave adds whole rows: 64 consecutive floats per atomic instruction.
    // D lane l, reg r: co = 16c + 4q + r, ci = wave*16 + j
    __syncthreads();
    constexpr int ROW = 64 * 9 + 1;
    static_assert(32 * ROW <= G::LDS_FLOATS, "epilogue staging fits the tile memory");
    float* st = sm;
    if (wave * 16 < nci) {
#pragma unroll
        for (int c = 0; c < COT; c++)
#pragma unroll
            for (int t = 0; t < 9; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) st[(c * 16 + q * 4 + r) * ROW + (wave * 16 + j) * 9 + t] = acc[c][t][r];
    }
    __syncthreads();
    const int rowlen = nci * 9;              // valid floats of a staged row
    for (int row = wave; row < COT * 16; row += 4) {
        float* drow = dw + ((size_t)(co0 + row) * cin + ci0) * 9;
        for (int i = lane; i < rowlen; i += 64) atomicAdd(drow + i, st[row * ROW + i]);
    }
}

}  // namespace apz
