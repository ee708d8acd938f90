// BatchNorm folding and weight packing ON THE DEVICE: the same maps apz_load_weights applies on the host (double
// arithmetic, rounded once), for weights that already live in device memory -- the trainer's tensors after an
// optimiser step (policy_value_net_mxnet.py:295-297: the reference copies the new parameters into its predict
// modules after every step).  The host path costs a device -> host copy of every tensor, 16 ms of folding and
// Winograd packing on one core and the upload; these kernels take a few tens of microseconds.  gfx950.
#pragma once
#include "trunk15_wino3b.h"
#include "trunk15_wino3h.h"
#include "wino_common.h"
#include <hip/hip_runtime.h>

namespace apz {

// scale[o] = gamma[o] / sqrt(var[o] + eps) (gamma == nullptr: 1), shift[o] = (bias[o] - mean[o]) * scale[o] + beta[o]
__global__ void fold_bn_kernel(const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, double* __restrict__ scale,
                               double* __restrict__ shift, float* __restrict__ bias_out, int cout, double eps) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= cout) return;
    const double g = gamma ? (double)gamma[o] : 1.0;
    const double s = g / sqrt((double)var[o] + eps);
    const double sh = ((double)bias[o] - (double)mean[o]) * s + (double)beta[o];
    scale[o] = s;
    shift[o] = sh;
    if (bias_out) bias_out[o] = (float)sh;
}

// direct-convolution fragments: X4 ? [cot][c4][lane][12] : [cot][c4][tap][lane]; element = w[co][ci][tap] * scale[co]
__global__ void pack_direct_kernel(const float* __restrict__ w, const double* __restrict__ scale, float* __restrict__ pk, int cin,
                                   int n4, int ncot, int x4) {
    const int total = ncot * n4 * 9 * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, tap = (i >> 6) % 9, c4 = (i / (64 * 9)) % n4, cot = i / (64 * 9 * n4);
        const int co = cot * 16 + (lane & 15), ci = c4 * 4 + (lane >> 4);
        float v = 0.f;
        if (ci < cin) v = (float)((double)w[((size_t)co * cin + ci) * 9 + tap] * scale[co]);
        if (x4)
            pk[(((size_t)cot * n4 + c4) * 64 + lane) * 12 + tap] = v;
        else
            pk[(((size_t)cot * n4 + c4) * 9 + tap) * 64 + lane] = v;
    }
}

// F(4x4,3x3) Winograd weights of the 128 -> 128 trunk shape, U[6i+k][co][ci] = (G g G^T)[i][k] of the folded kernel, in
// double, rounded once, in trunk15_wino3.h's layout (wino_common.h: [cot 8][row half 2][c4 32][lane 64][20]).  One thread
// per (co, ci).
__global__ void pack_wino_folded_kernel(const float* __restrict__ w, const double* __restrict__ scale, float* __restrict__ up2,
                                        float* __restrict__ up3s) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 128 * 128) return;
    const int co = idx >> 7, ci = idx & 127;
    const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    double g[3][3], t[6][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) g[a][b] = (double)w[((size_t)co * 128 + ci) * 9 + a * 3 + b] * scale[co];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int b = 0; b < 3; b++) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
    const int cot = co >> 4, jj = co & 15;
    const int qq = ci & 3, c4 = ci >> 2;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double u = t[i][0] * G[k][0] + t[i][1] * G[k][1] + t[i][2] * G[k][2];
            const int pass = i / 3;
            up2[((((size_t)cot * 2 + pass) * 32 + c4) * 64 + (qq * 16 + jj)) * 20 + (i - 3 * pass) * 6 + k] = (float)u;
            if (up3s) up3s[WinoPackSmall::index(co, ci, 6 * i + k)] = (float)u;
        }
}

// The same U as three bf16 terms for trunk15_wino3b.h (Wino3B::upk_offset): round to nearest even, remainders in double.
__global__ void pack_wino3b_folded_kernel(const float* __restrict__ w, const double* __restrict__ scale, unsigned short* __restrict__ up) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 128 * 128) return;
    const int co = idx >> 7, ci = idx & 127;
    const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    double g[3][3], t[6][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) g[a][b] = (double)w[((size_t)co * 128 + ci) * 9 + a * 3 + b] * scale[co];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int b = 0; b < 3; b++) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int k = 0; k < 6; k++) {
            double rem = t[i][0] * G[k][0] + t[i][1] * G[k][1] + t[i][2] * G[k][2];
#pragma unroll
            for (int term = 0; term < 3; term++) {
                unsigned u = __builtin_bit_cast(unsigned, (float)rem);
                if ((u & 0x7f800000u) != 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u);
                u &= 0xffff0000u;
                up[Wino3B::upk_offset(co, ci, 6 * i + k, term) / 2] = (unsigned short)(u >> 16);
                rem -= (double)__builtin_bit_cast(float, u);
            }
        }
}

// The same U for trunk15_wino3h.h: one workgroup per output channel (thread = input channel) finds max |U| of the channel,
// S = Wino3H::scale_for(max), and writes U S as two fp16 terms (both round to nearest even, the remainder in double) at
// Wino3H::upk_offset; bias3h = [128 folded biases][128 x 1 / S].  Launch: grid 128, block 128.
__global__ void pack_wino3h_folded_kernel(const float* __restrict__ w, const double* __restrict__ scale, const double* __restrict__ shift,
                                          unsigned short* __restrict__ up, float* __restrict__ bias3h) {
    const int co = blockIdx.x, ci = threadIdx.x;
    const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    double g[3][3], t[6][3], u[36];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) g[a][b] = (double)w[((size_t)co * 128 + ci) * 9 + a * 3 + b] * scale[co];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int b = 0; b < 3; b++) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
    double m = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int k = 0; k < 6; k++) {
            u[6 * i + k] = t[i][0] * G[k][0] + t[i][1] * G[k][1] + t[i][2] * G[k][2];
            m = fmax(m, fabs(u[6 * i + k]));
        }
    __shared__ double red[128];
    red[ci] = m;
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) {
        if (ci < s) red[ci] = fmax(red[ci], red[ci + s]);
        __syncthreads();
    }
    const float S = Wino3H::scale_for(red[0]);
    if (ci == 0) {
        bias3h[co] = (float)shift[co];
        bias3h[128 + co] = 1.f / S;
    }
#pragma unroll
    for (int pos = 0; pos < 36; pos++) {
        const double x = u[pos] * (double)S;
        const _Float16 hi = (_Float16)(float)x;
        const _Float16 lo = (_Float16)(float)(x - (double)(float)hi);
        up[Wino3H::upk_offset(co, ci, pos, 0) / 2] = __builtin_bit_cast(unsigned short, hi);
        up[Wino3H::upk_offset(co, ci, pos, 1) / 2] = __builtin_bit_cast(unsigned short, lo);
    }
}

// heads: rows [row0, row0 + rows) of the [6][C] matrix of both 1x1 convolutions, and their folded biases
__global__ void pack_head_conv_kernel(const float* __restrict__ w, const double* __restrict__ scale, const double* __restrict__ shift,
                                      float* __restrict__ w6, float* __restrict__ b6, int rows, int row0, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * C) {
        const int o = i / C;
        w6[(size_t)row0 * C + i] = (float)((double)w[i] * scale[o]);
    }
    if (i < rows) b6[row0 + i] = (float)shift[i];
}

// policy FullyConnected [hw][4 hw] -> head_fc_kernel's [tile][trip of 8 k-steps][lane][8] (zero past hw outputs / KS steps)
__global__ void pack_fc_kernel(const float* __restrict__ wfc, float* __restrict__ pk, int hw, int ntile, int KG) {
    const int K = 4 * hw, KS = hw;
    const long total = (long)ntile * KG * 64 * 8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int u = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const long r = i >> 9;
        const int trip = (int)(r % KG), nt = (int)(r / KG);
        const int s = trip * 8 + u, o = nt * 16 + (lane & 15), k = 4 * s + (lane >> 4);
        pk[i] = (s < KS && o < hw) ? wfc[(size_t)o * K + k] : 0.f;
    }
}

}  // namespace apz
