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
// banks.  Every batch slice writes its own partial dW; the launcher adds the slices in index order
// (colsum_kernel): no atomics, the same bits on every run (rounds 1 - 3 used float atomics).
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
__device__ __forceinline__ void pack_wino2_thread(const float* __restrict__ w, float* __restrict__ upk, int transpose_flip, int i) {
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
__global__ void pack_wino2_kernel(const float* __restrict__ w, float* __restrict__ upk, int transpose_flip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 8 * 2 * 32 * 64) pack_wino2_thread(w, upk, transpose_flip, i);
}
// `count` layers whose weights sit back to back ([count][128][128][3][3]), both orientations, in one launch (grid.y = 2 count):
// upk [count][2][UPK_FLOATS] (forward, then data gradient) -- a training step packs its 20 trunk layers twice, and 40
// launches of 5 us were 4 % of the step at the reference's batch size
__global__ void pack_wino2_many_kernel(const float* __restrict__ w, float* __restrict__ upk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int layer = blockIdx.y >> 1, flip = blockIdx.y & 1;
    if (i < 8 * 2 * 32 * 64)
        pack_wino2_thread(w + (size_t)layer * (128 * 128 * 9), upk + (size_t)blockIdx.y * (8 * 2 * 32 * 64 * 20), flip, i);
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
// Two launches forward, two backward, every one a grid (channel, batch split); nothing is zeroed beforehand and there
// are no atomics (round 3: memset + statistics with double atomics + finalize + apply -- each small launch ~5 us of
// queue time; and the sums depended on the arrival order in their last bits).  Finishing the reduction INSIDE the
// statistics launch (the workgroup that draws the last ticket adds the partials) was built and measured three ways --
// a device-scope fence per workgroup writes back the XCD's whole L2 (step 5.0 -> 15.4 ms), one ticket counter per launch
// serialises 2 048 atomics on one address (+25 us per launch), per-channel tickets with write-through stores cost ~5 us of
// dependent device-scope round trips, as much as the launch they replace (profiles/r04_train_fusion.md) -- and dropped:
// the CONSUMER workgroups add the partials.
//   bn_stats_kernel      part[c][split] = (sum x, sum x^2) in double, plain stores
//   bn_apply_kernel      every workgroup adds ITS channel's partials (<= 64 pairs of doubles, a fixed order) and derives
//                        mean, 1/sqrt(var + eps); y = act((x - mean) * invstd * gamma + beta (+ resid)); workgroup
//                        (c, 0) writes mean / invstd for the backward pass and the moving statistics (torch semantics:
//                        unbiased variance in the moving average, momentum = weight of the NEW value)
//   bn_bwd_reduce_kernel part[c][split] = (sum dz, sum dz * xhat), dz = dy (* [out > 0] with ReLU)
//   bn_bwd_apply_kernel  adds its channel's partials: s1, s2; dx = gamma * invstd * (dz - s1/M - xhat * s2/M); dres = dz;
//                        workgroup (c, 0) writes dbeta = s1, dgamma = s2; dxsum (may be NULL): row `split` of
//                        [splits][ld] floats gets the sum of the dx this workgroup wrote -- the bias gradient of the
//                        convolution in front of the BatchNorm is the column sum of that matrix (colsum_kernel,
//                        heads_train.h; the trainer adds all layers' rows in ONE launch)
struct BnFinal {            // what workgroup (c, 0) of bn_apply writes (run_* may be NULL)
    float* mean;
    float* invstd;
    float* run_mean;
    float* run_var;
    double M;               // elements per channel
    float eps, momentum;
};
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
// the two sums of channel c over its `splits` partials, to every thread: thread k loads split k (k + 256, ...), then
// the shuffle tree -- a fixed order
__device__ __forceinline__ void channel_sums(const double* __restrict__ part, int c, int splits, double* sh, double& s1,
                                             double& s2) {
    s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < splits; k += 256) {
        s1 += part[2 * ((size_t)c * splits + k)];
        s2 += part[2 * ((size_t)c * splits + k) + 1];
    }
    s1 = block_sum_256(s1, sh);
    s2 = block_sum_256(s2, sh);
}
// ... turned into the channel's mean and 1 / sqrt(var + eps); workgroup (c, 0) records them
__device__ __forceinline__ void channel_stats(const double* __restrict__ part, int c, int splits, double* sh, const BnFinal& f,
                                              float& mean, float& invstd) {
    double s1, s2;
    channel_sums(part, c, splits, sh, s1, s2);
    const double m = s1 / f.M;
    double var = s2 / f.M - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    invstd = (float)(1.0 / sqrt(var + (double)f.eps));
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        f.mean[c] = mean;
        f.invstd[c] = invstd;
        if (f.run_mean) f.run_mean[c] = f.run_mean[c] * (1.f - f.momentum) + f.momentum * (float)m;
        if (f.run_var)
            f.run_var[c] = f.run_var[c] * (1.f - f.momentum) + f.momentum * (float)(f.M > 1.0 ? var * f.M / (f.M - 1.0) : var);
    }
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, double* __restrict__ part, int n, int C,
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
        part[2 * ((size_t)c * gridDim.y + blockIdx.y)] = s1;
        part[2 * ((size_t)c * gridDim.y + blockIdx.y) + 1] = s2;
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ resid,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const double* __restrict__ part, int psplits, BnFinal fin,
                                                       float* __restrict__ y, int n, int C, int PS, int RS, int H, int W,
                                                       int relu) {
    __shared__ double sh[4];
    const int c = blockIdx.x, cells = H * RS;
    float m, is;
    channel_stats(part, c, psplits, sh, fin, m, is);
    const float g = gamma ? gamma[c] : 1.f, bt = beta[c];
    for (int b = blockIdx.y; b < n; b += gridDim.y) {
        const size_t base = ((size_t)b * C + c) * PS;
        for (int q = threadIdx.x; q < cells; q += 256) {
            float v = 0.f;
            if (q % RS < W) {
                v = (x[base + q] - m) * is * g + bt;
                if (resid) v += resid[base + q];
                if (relu) v = fmaxf(v, 0.f);
            }
            y[base + q] = v;
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ out, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, double* __restrict__ part, int n,
                                                            int C, int PS, int RS, int H, int W, int relu) {
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
        part[2 * ((size_t)c * gridDim.y + blockIdx.y)] = s1;
        part[2 * ((size_t)c * gridDim.y + blockIdx.y) + 1] = s2;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ out, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const double* __restrict__ part, int psplits, float* __restrict__ dx,
                                                           float* __restrict__ dres, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ dxsum, int dxsum_ld,
                                                           int n, int C, int PS, int RS, int H, int W, int relu, double M) {
    __shared__ double sh[4];
    const int c = blockIdx.x, cells = H * RS;
    double s1, s2;
    channel_sums(part, c, psplits, sh, s1, s2);
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        if (dbeta) dbeta[c] = (float)s1;
        if (dgamma) dgamma[c] = (float)s2;
    }
    const float k0 = (float)(s1 / M), k1 = (float)(s2 / M), m = mean[c], is = invstd[c], g = gamma ? gamma[c] : 1.f;
    double ds = 0.0;
    for (int b = blockIdx.y; b < n; b += gridDim.y) {
        const size_t base = ((size_t)b * C + c) * PS;
        for (int q = threadIdx.x; q < cells; q += 256) {
            float gx = 0.f, gr = 0.f;
            if (q % RS < W) {
                float dz = dy[base + q];
                if (relu && !(out[base + q] > 0.f)) dz = 0.f;
                const float xhat = (x[base + q] - m) * is;
                gx = g * is * (dz - k0 - xhat * k1);
                gr = dz;
            }
            dx[base + q] = gx;
            if (dres) dres[base + q] = gr;
            ds += gx;
        }
    }
    if (dxsum) {
        ds = block_sum_256(ds, sh);
        if (threadIdx.x == 0) dxsum[(size_t)blockIdx.y * dxsum_ld + c] = (float)ds;
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
// they add nothing to any sum, and are written back as zero): 16-byte accesses, no per-element index division;
// a workgroup walks four boards per trip (4 x 60 of its 256 threads)
__global__ __launch_bounds__(256) void bn_stats_r16_kernel(const float* __restrict__ x, double* __restrict__ part, int n, int C) {
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
        part[2 * ((size_t)c * gridDim.y + blockIdx.y)] = s1;
        part[2 * ((size_t)c * gridDim.y + blockIdx.y) + 1] = s2;
    }
}

__global__ __launch_bounds__(256) void bn_apply_r16_kernel(const float* __restrict__ x, const float* __restrict__ resid,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const double* __restrict__ part, int psplits, BnFinal fin,
                                                           float* __restrict__ y, unsigned char* __restrict__ mask, int n, int C,
                                                           int relu) {
    __shared__ double sh[4];
    const int c = blockIdx.x, t = threadIdx.x, sub = t / 60, k = t - sub * 60;
    float m, is;
    channel_stats(part, c, psplits, sh, fin, m, is);
    const float sc = is * (gamma ? gamma[c] : 1.f), shf = beta[c] - m * sc;
    if (sub >= 4) return;
    for (int b = blockIdx.y * 4 + sub; b < n; b += gridDim.y * 4) {
        const size_t o = ((size_t)b * C + c) * 60 + k;
        f32x4 a = reinterpret_cast<const f32x4*>(x)[o];
        a = a * sc + shf;
        if (resid) a += reinterpret_cast<const f32x4*>(resid)[o];
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; e++) a[e] = fmaxf(a[e], 0.f);
        }
        if ((k & 3) == 3) a[3] = 0.f;            // the pad column (60 float4 per plane: 4 per row)
        reinterpret_cast<f32x4*>(y)[o] = a;
        // the ReLU decisions of these four elements for the backward pass: one byte per 16 bytes of y (bn_bwd_* then read
        // 3.9 MB instead of the 63 MB output tensor per 512-board layer, twice)
        if (mask) mask[o] = (unsigned char)((a[0] > 0.f) | ((a[1] > 0.f) << 1) | ((a[2] > 0.f) << 2) | ((a[3] > 0.f) << 3));
    }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_r16_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ out, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, double* __restrict__ part,
                                                                const unsigned char* __restrict__ mask, int n, int C, int relu) {
    __shared__ double sh[4];
    const int c = blockIdx.x, t = threadIdx.x, sub = t / 60, k = t - sub * 60;
    const float m = mean[c], is = invstd[c];
    double s1 = 0.0, s2 = 0.0;
    if (sub < 4)
        for (int b = blockIdx.y * 4 + sub; b < n; b += gridDim.y * 4) {
            const size_t o = ((size_t)b * C + c) * 60 + k;
            f32x4 g = reinterpret_cast<const f32x4*>(dy)[o];
            const f32x4 xv = reinterpret_cast<const f32x4*>(x)[o];
            if (relu && mask) {
                const unsigned bits = mask[o];
#pragma unroll
                for (int e = 0; e < 4; e++) g[e] = (bits >> e) & 1u ? g[e] : 0.f;
            } else if (relu) {
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
        part[2 * ((size_t)c * gridDim.y + blockIdx.y)] = s1;
        part[2 * ((size_t)c * gridDim.y + blockIdx.y) + 1] = s2;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_r16_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ out, const float* __restrict__ gamma,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const double* __restrict__ part, int psplits,
                                                               float* __restrict__ dx, float* __restrict__ dres,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ dxsum, int dxsum_ld,
                                                               const unsigned char* __restrict__ mask, int n, int C, int relu,
                                                               double M) {
    __shared__ double sh[4];
    const int c = blockIdx.x, t = threadIdx.x, sub = t / 60, k = t - sub * 60;
    double s1, s2;
    channel_sums(part, c, psplits, sh, s1, s2);
    if (blockIdx.y == 0 && t == 0) {
        if (dbeta) dbeta[c] = (float)s1;
        if (dgamma) dgamma[c] = (float)s2;
    }
    const float m = mean[c], is = invstd[c], k0 = (float)(s1 / M), k1 = (float)(s2 / M);
    const float gi = (gamma ? gamma[c] : 1.f) * is;
    double ds = 0.0;
    if (sub < 4)
        for (int b = blockIdx.y * 4 + sub; b < n; b += gridDim.y * 4) {
            const size_t o = ((size_t)b * C + c) * 60 + k;
            f32x4 g = reinterpret_cast<const f32x4*>(dy)[o];
            const f32x4 xv = reinterpret_cast<const f32x4*>(x)[o];
            if (relu && mask) {
                const unsigned bits = mask[o];
#pragma unroll
                for (int e = 0; e < 4; e++) g[e] = (bits >> e) & 1u ? g[e] : 0.f;
            } else if (relu) {
                const f32x4 ov = reinterpret_cast<const f32x4*>(out)[o];
#pragma unroll
                for (int e = 0; e < 4; e++) g[e] = ov[e] > 0.f ? g[e] : 0.f;
            }
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; e++) r[e] = gi * (g[e] - k0 - (xv[e] - m) * is * k1);
            if ((k & 3) == 3) r[3] = 0.f, g[3] = 0.f;
            reinterpret_cast<f32x4*>(dx)[o] = r;
            if (dres) reinterpret_cast<f32x4*>(dres)[o] = g;
            ds += (double)((r[0] + r[1]) + (r[2] + r[3]));
        }
    if (dxsum) {
        ds = block_sum_256(ds, sh);
        if (t == 0) dxsum[(size_t)blockIdx.y * dxsum_ld + c] = (float)ds;
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

// x [n][cin][H][W], dy [n][cout][H][W] dense; dw: [slices = gridDim.z][cout][cin][3][3] partial sums.
// grid: (cout/32, ceil(cin/64), batch slices).  The staged input tile (64 channels) is shared by
// the workgroup's two C_out tiles; a wave holds 2 x 9 accumulator tiles (72 registers).
// Staging is software-pipelined through registers: the NEXT board's 16-byte pieces are loaded
// before this board's MFMA loop and scattered into LDS after it, so HBM/L2 latency hides behind
// ~1000 MFMAs per wave.  The epilogue transposes the accumulators through LDS so that every
// store wave-instruction writes 64 consecutive floats of one dW row.
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
    const float* brow = xt + (wave * 16 + j) * G::XPS;       // B: ci = wave*16 + j (by_pixels: ci = j, below)

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

    // <= 16 input channels (the stem: 9 planes): one wave would do all of the workgroup's MFMAs (188 us per 512 boards for
    // a layer of 2.4 GFLOP) -- the four waves then split the k-steps (pixels) of channel tile 0 instead of the channel
    // tiles, and their partial sums meet in the epilogue's staging area
    const bool by_pixels = nci <= 16;
    if (by_pixels) brow = xt + j * G::XPS;
    fetch(blockIdx.z);
    for (int b = blockIdx.z; b < n; b += gridDim.z) {
        __syncthreads();                     // previous board's tiles fully consumed (and the zero fill done)
        scatter();
        __syncthreads();
        fetch(b + gridDim.z);                // in flight during the MFMA loop below
        if (by_pixels) {
            for (int s = wave; s < G::KSTEPS; s += 4) {
                float a[COT];
#pragma unroll
                for (int c = 0; c < COT; c++) a[c] = arow[c * 16 * G::YPS + 4 * s];
                const int p = 4 * s + q, yy = p / WP, xx = p - yy * WP;
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
            }
        } else if (wave * 16 < nci) {        // wave-uniform: this wave's 16 input channels exist
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
    if (by_pixels) {                         // the four waves' partial sums of channel tile 0, added in wave order
        for (int w = 0; w < 4; w++) {
            if (wave == w) {
#pragma unroll
                for (int c = 0; c < COT; c++)
#pragma unroll
                    for (int t = 0; t < 9; t++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            float* d = st + (c * 16 + q * 4 + r) * ROW + j * 9 + t;
                            *d = w ? *d + acc[c][t][r] : acc[c][t][r];
                        }
            }
            __syncthreads();
        }
    } else if (wave * 16 < nci) {
#pragma unroll
        for (int c = 0; c < COT; c++)
#pragma unroll
            for (int t = 0; t < 9; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) st[(c * 16 + q * 4 + r) * ROW + (wave * 16 + j) * 9 + t] = acc[c][t][r];
    }
    __syncthreads();
    // this batch slice's partial dW (every (slice, C_out tile, C_in tile) has exactly one workgroup: plain stores);
    // colsum_kernel (heads_train.h) adds the slices in index order -- no float atomics, the same bits on every run
    const int rowlen = nci * 9;              // valid floats of a staged row
    float* part = dw + (size_t)blockIdx.z * cout * cin * 9;
    for (int row = wave; row < COT * 16; row += 4) {
        float* drow = part + ((size_t)(co0 + row) * cin + ci0) * 9;
        for (int i = lane; i < rowlen; i += 64) drow[i] = st[row * ROW + i];
    }
}

}  // namespace apz
