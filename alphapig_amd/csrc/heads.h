// Policy / value heads for gfx950.
//
//  head_conv1x1_kernel   both 1x1 convs (C -> 4 policy + 2 value channels) + folded BN + ReLU in
//                        one pass over the trunk output (HBM-bound: reads C*H*W floats per board
//                        once, coalesced), writing the flattened FC inputs [n][4*HW] and [n][2*HW].
//  head_fc_kernel        policy FullyConnected as a dense fp32 MFMA GEMM (16 boards x HW outputs
//                        x 4*HW deep per workgroup, v_mfma_f32_16x16x4_f32) + bias + row softmax
//                        as a wavefront reduction; value FullyConnected (2*HW -> 1) as a wavefront
//                        dot-reduce + tanh.
// Reference graph: policy_value_net_mxnet.py:85-97 (conv3_1_1 / fc_3_1_1 / SoftmaxActivation,
// conv3_2_1 / fc_3_2_1 / tanh); Dropout is the identity at inference.
#pragma once
#include <hip/hip_runtime.h>

#include "conv3x3_mfma.h"

namespace apz {

// w6 [6][C] (rows 0-3 policy, 4-5 value, BN folded), b6 [6]
// Thread = (pixel p, channel slice qq of Q): boards of at most 128 pixels (8x8) would leave most of the 256 threads idle
// behind one serial chain of C dependent loads per pixel (64 us per 32 boards at C = 256), so the channels are dealt
// round-robin over Q = 256 / HW (1, 2 or 4) slices and the slices' partial sums meet in LDS, added in slice order:
// a board's bits do not depend on the batch or on the grid.
__global__ __launch_bounds__(256) void head_conv1x1_kernel(const float* __restrict__ x, const float* __restrict__ w6,
                                                           const float* __restrict__ b6, float* __restrict__ featp,
                                                           float* __restrict__ featv, int n, int C, int HW, int W, int in_ps,
                                                           int in_rs) {
    // input plane stride in_ps / row stride in_rs: dense NCHW (HW / W) or rows16 (240 / 16)
    __shared__ float part[3 * 6 * 128];                   // slices 1..3: [slice - 1][output 6][pixel]
    const int Q = HW <= 64 ? 4 : HW <= 128 ? 2 : 1;
    const int tid = threadIdx.x;
    const int qq = Q > 1 ? tid / HW : 0, pq = Q > 1 ? tid - qq * HW : tid;
    for (int b = blockIdx.x; b < n; b += gridDim.x) {
        const float* xb = x + (size_t)b * C * in_ps;
        for (int p0 = 0; p0 < HW; p0 += (Q > 1 ? HW : (int)blockDim.x)) {
            const int p = p0 + pq;
            const bool live = p < HW && qq < Q;
            float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (live) {
                const int pin = (p / W) * in_rs + (p % W);
#pragma unroll 4
                for (int c = qq; c < C; c += Q) {
                    const float v = xb[c * in_ps + pin];
#pragma unroll
                    for (int o = 0; o < 6; o++) a[o] = fmaf(w6[o * C + c], v, a[o]);
                }
                if (qq > 0)
#pragma unroll
                    for (int o = 0; o < 6; o++) part[((qq - 1) * 6 + o) * 128 + p] = a[o];
            }
            if (Q > 1) __syncthreads();
            if (live && qq == 0) {
                for (int s = 1; s < Q; s++)
#pragma unroll
                    for (int o = 0; o < 6; o++) a[o] += part[((s - 1) * 6 + o) * 128 + p];
                float* fp = featp + (size_t)b * 4 * HW;
                float* fv = featv + (size_t)b * 2 * HW;
#pragma unroll
                for (int o = 0; o < 4; o++) fp[o * HW + p] = fmaxf(a[o] + b6[o], 0.f);
                fv[0 * HW + p] = fmaxf(a[4] + b6[4], 0.f);
                fv[1 * HW + p] = fmaxf(a[5] + b6[5], 0.f);
            }
            if (Q > 1) __syncthreads();                   // `part` is reused by the next board
        }
    }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup = 16 boards.  K = 4*HW (policy FC depth), NOUT = HW.
//   A (features)  lane l: board = l&15, k = 4s + (l>>4)   from LDS, row stride K+1
//   B (weights)   lane l: out = nt*16 + (l&15), k = 4s + (l>>4), pre-packed [NTILE][K/4][64]
//   D  lane l, reg r: board = (l>>4)*4 + r, out = nt*16 + (l&15)
// TPW = n-tiles per wave (ceil(NTILE / 4)); each wave keeps TPW independent accumulators so
// the 40-cycle dependent-MFMA latency is covered.
// The same two 1x1 head convolutions for the trunk's padded-row activations [n][C][15][16].  One workgroup per board;
// wave w walks the planes of channel quarter w (C / 4 channels), lane k < 60 owning one 16-byte piece (4 pixels of a
// row) with eight 16-byte loads in flight; the four partial sums meet in LDS and are added in quarter order.
// (One thread per piece walking all C planes -- 4 boards per workgroup -- was 16 dependent load rounds on 128 of the
// 256 CUs: 22 us per 512 boards for 63 MB.)  Reads C*960 bytes per board once, coalesced.
__global__ __launch_bounds__(256) void head_conv1x1_r16_kernel(const float* __restrict__ x, const float* __restrict__ w6,
                                                               const float* __restrict__ b6, float* __restrict__ featp,
                                                               float* __restrict__ featv, int n, int C) {
    __shared__ f32x4 part[4][6][60];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x, cq = C / 4;
    if (lane < 60) {
        const f32x4* xb = reinterpret_cast<const f32x4*>(x + ((size_t)b * C + wave * cq) * 240) + lane;
        const float* wq = w6 + wave * cq;
        f32x4 acc[6];
#pragma unroll
        for (int o = 0; o < 6; o++) acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < cq; c0 += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = xb[(size_t)(c0 + u) * 60];
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int o = 0; o < 6; o++) acc[o] += wq[o * C + c0 + u] * v[u];
        }
#pragma unroll
        for (int o = 0; o < 6; o++) part[wave][o][lane] = acc[o];
    }
    __syncthreads();
    if (lane >= 60) return;
    const int row = lane >> 2, col0 = (lane & 3) * 4;
    // wave w finishes output channels w and w + 4 (policy 0-3, value 4-5)
    for (int o = wave; o < 6; o += 4) {
        const f32x4 s = ((part[0][o][lane] + part[1][o][lane]) + part[2][o][lane]) + part[3][o][lane];
        float* dst = (o < 4 ? featp + ((size_t)b * 4 + o) * 225 : featv + ((size_t)b * 2 + (o - 4)) * 225) + row * 15 + col0;
        const float bo = b6[o];
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (col0 + e < 15) dst[e] = fmaxf(s[e] + bo, 0.f);
    }
}

// SPLIT: the n-tiles are dealt over gridDim.y workgroups per 16 boards (wave w of workgroup y: tile 4 y + w, TPW = 1)
// and the kernel ends with the biased logits in `logits_ws` [n][ntile * 16]; head_softmax_value_kernel finishes the
// heads.  With one workgroup per 16 boards a 512-board batch ran on 32 of the 256 CUs, each streaming the whole
// 810 KB weight matrix through 900 dependent MFMAs per wave (53 us); split four ways it is 128 workgroups x 225.
template <int TPW, bool SPLIT = false>
__global__ __launch_bounds__(256) void head_fc_kernel(const float* __restrict__ featp, const float* __restrict__ featv,
                                                      const float* __restrict__ wfc_pk, const float* __restrict__ bfc,
                                                      const float* __restrict__ wv, const float* __restrict__ bv,
                                                      float* __restrict__ probs, float* __restrict__ values,
                                                      float* __restrict__ logits_out, float* __restrict__ vlogits_out,
                                                      int n, int HW, float* __restrict__ logits_ws = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int K = 4 * HW, KS = HW;              // KS = K/4 k-steps
    const int ntile = (HW + 15) / 16;
    const int ldf = K + 1;                      // feature row stride
    const int ldl = ntile * 16;                 // logits row stride
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int b0 = blockIdx.x * 16;
    const int nb = min(16, n - b0);

    // stage 16 feature rows (zero rows beyond n): the rows are one contiguous block of 16*K floats, K a multiple
    // of 4 -- all 16-byte loads of a thread are issued before the first LDS store (one memory latency, not one per
    // loop trip: the staging was a third of the kernel's time)
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(featp + (size_t)b0 * K);
        const int n4 = 16 * K / 4;
        constexpr int NV = 16;                  // >= 16*K/4/256 for K <= 1024
        f32x4 v[NV];
#pragma unroll
        for (int u = 0; u < NV; u++) {
            const int i4 = tid + 256 * u;
            const int r = (4 * i4) / K;
            v[u] = (i4 < n4 && r < nb) ? src[i4] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < NV; u++) {
            const int i4 = tid + 256 * u;
            if (i4 < n4) {
                const int r = (4 * i4) / K, c = 4 * i4 - r * K;
                float* d = sm + r * ldf + c;
                d[0] = v[u][0];
                d[1] = v[u][1];
                d[2] = v[u][2];
                d[3] = v[u][3];
            }
        }
    }
    __syncthreads();

    f32x4 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* arow = sm + j * ldf + q;
    // K loop in trips of 8 k-steps.  wfc_pk is [tile][trip][lane][8]: a lane's eight weights of a trip are 32
    // contiguous bytes (two 16-byte loads; a wave reads 2 KB contiguously), zero-padded past KS.  The NEXT
    // trip's fragments are loaded before this trip's MFMAs issue (register double buffer), so the L2 latency
    // of a trip hides behind the 8 x TPW MFMAs of the previous one.
    const int KG = (KS + 7) / 8;
    f32x4 bw[2][TPW][2];
    auto load_trip = [&](int g, int buf) {
#pragma unroll
        for (int i = 0; i < TPW; i++) {
            const int nt = min((SPLIT ? 4 * (int)blockIdx.y : 0) + wave + 4 * i, ntile - 1);   // clamp: surplus tiles recompute the last one
            const f32x4* wp = reinterpret_cast<const f32x4*>(wfc_pk) + (((size_t)nt * KG + g) * 64 + lane) * 2;
            bw[buf][i][0] = wp[0];
            bw[buf][i][1] = wp[1];
        }
    };
    auto mma_trip = [&](int g, int buf) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = arow[4 * min(8 * g + u, KS - 1)];   // past KS: any valid step (weights are 0)
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < TPW; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bw[buf][i][u >> 2][u & 3], acc[i], 0, 0, 0);
    };
    load_trip(0, 0);
    int g = 0;
    for (; g + 2 <= KG; g += 2) {               // two trips per iteration: static buffer indices
        load_trip(g + 1, 1);
        mma_trip(g, 0);
        if (g + 2 < KG) load_trip(g + 2, 0);
        mma_trip(g + 1, 1);
    }
    if (g < KG) mma_trip(g, 0);                 // odd trip count: its fragments are in buffer 0
    if (SPLIT) {
        const int nt = 4 * (int)blockIdx.y + wave;
        if (nt < ntile) {
            const int o = nt * 16 + j;
            const float bb = (o < HW) ? bfc[o] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (q * 4 + r < nb) logits_ws[(size_t)(b0 + q * 4 + r) * ldl + o] = acc[0][r] + bb;
        }
        return;
    }
    __syncthreads();            // features consumed; reuse LDS for the logits
    float* lg = sm;
#pragma unroll
    for (int i = 0; i < TPW; i++) {
        const int nt = wave + 4 * i;
        if (nt < ntile) {
            const int o = nt * 16 + j;
            const float bb = (o < HW) ? bfc[o] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) lg[(q * 4 + r) * ldl + o] = acc[i][r] + bb;
        }
    }
    __syncthreads();
    // softmax: wave w owns boards 4w..4w+3 (SoftmaxActivation, instance mode)
    for (int r = wave * 4; r < wave * 4 + 4; r++) {
        if (r >= nb) break;
        const float* row = lg + r * ldl;
        float m = -INFINITY;
        for (int o = lane; o < HW; o += 64) m = fmaxf(m, row[o]);
        m = wave_max(m);
        float s = 0.f;
        for (int o = lane; o < HW; o += 64) s += expf(row[o] - m);
        s = wave_sum(s);
        const float inv = 1.0f / s;
        for (int o = lane; o < HW; o += 64) {
            const float l = row[o];
            probs[(size_t)(b0 + r) * HW + o] = expf(l - m) * inv;
            if (logits_out) logits_out[(size_t)(b0 + r) * HW + o] = l;
        }
        // value head: dot(featv[2*HW], wv) + bv -> tanh
        const float* fv = featv + (size_t)(b0 + r) * 2 * HW;
        float d = 0.f;
        for (int o = lane; o < 2 * HW; o += 64) d = fmaf(fv[o], wv[o], d);
        d = wave_sum(d) + bv[0];
        if (lane == 0) {
            values[b0 + r] = tanhf(d);
            if (vlogits_out) vlogits_out[b0 + r] = d;
        }
    }
}

// Second half of the split policy head: one wavefront per board -- row softmax of the logits head_fc_kernel<1, true>
// left in logits_ws [n][ldl] (SoftmaxActivation, instance mode), and the value head (2*HW -> 1 dot product + tanh).
__global__ __launch_bounds__(256) void head_softmax_value_kernel(const float* __restrict__ logits_ws, const float* __restrict__ featv,
                                                                 const float* __restrict__ wv, const float* __restrict__ bv,
                                                                 float* __restrict__ probs, float* __restrict__ values,
                                                                 float* __restrict__ logits_out, float* __restrict__ vlogits_out,
                                                                 int n, int HW, int ldl) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    // every load of the board is issued before the first reduction: one memory round trip instead of one per pass
    // (HW <= 256: four logits per lane; 2 HW <= 512: eight value-head terms per lane)
    const float* row = logits_ws + (size_t)r * ldl;
    const float* fv = featv + (size_t)r * 2 * HW;
    float l[4], f[8], w[8];
#pragma unroll
    for (int i = 0; i < 4; i++) l[i] = (lane + 64 * i < HW) ? row[lane + 64 * i] : -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int o = lane + 64 * i;
        f[i] = o < 2 * HW ? fv[o] : 0.f;
        w[i] = o < 2 * HW ? wv[o] : 0.f;
    }
    const float m = wave_max(fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3])));
    float ex[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ex[i] = (lane + 64 * i < HW) ? expf(l[i] - m) : 0.f;
        s += ex[i];
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int o = lane + 64 * i;
        if (o < HW) {
            probs[(size_t)r * HW + o] = ex[i] * inv;
            if (logits_out) logits_out[(size_t)r * HW + o] = l[i];
        }
    }
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) d = fmaf(f[i], w[i], d);
    d = wave_sum(d) + bv[0];
    if (lane == 0) {
        values[r] = tanhf(d);
        if (vlogits_out) vlogits_out[r] = d;
    }
}

// ---- position codes -> input planes (Board.current_state, game.py:68-94 / :96-115) ---------
// codes [n][stride] u8: byte m = h*W+w (un-flipped): 0 empty, 1+min(age,3) own, 5+min(age,3) opp;
// byte HW = colour plane value.  planes [n][NP][H][W] with the vertical flip of game.py:94.
__global__ void encode_planes_kernel(const unsigned char* __restrict__ codes, float* __restrict__ planes, int n, int H,
                                     int W, int stride, int NP) {
    const int HW = H * W;
    const int total = n * HW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / HW, m = i - b * HW;
        const int h = m / W, w = m - h * W;
        const int o = (H - 1 - h) * W + w;
        const unsigned char* cb = codes + (size_t)b * stride;
        const int code = cb[m];
        const float colour = cb[HW] ? 1.f : 0.f;
        float* pb = planes + (size_t)b * NP * HW;
        const int opp = code >= 5, age = (code - 1) & 3;
        if (NP == 9) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float on = (code && k <= age) ? 1.f : 0.f;
                pb[(6 - 2 * k) * HW + o] = opp ? 0.f : on;
                pb[(7 - 2 * k) * HW + o] = opp ? on : 0.f;
            }
            pb[8 * HW + o] = colour;
        } else {
            pb[0 * HW + o] = (code && !opp) ? 1.f : 0.f;
            pb[1 * HW + o] = (code && opp) ? 1.f : 0.f;
            pb[2 * HW + o] = (code && age == 0) ? 1.f : 0.f;
            pb[3 * HW + o] = colour;
        }
    }
}

// ---- 8-fold dihedral augmentation as a table-driven gather (train_mxnet.py:115-135) ---------
// perm_s [8][HW], perm_p [8][HW]: out[k][..][i] = in[..][perm[k][i]]
__global__ void augment8_kernel(const float* __restrict__ planes, const float* __restrict__ pi,
                                const int* __restrict__ perm_s, const int* __restrict__ perm_p,
                                float* __restrict__ planes_out, float* __restrict__ pi_out, int n, int C, int HW) {
    const long total = (long)n * 8 * (C + 1) * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        long t = i / HW;
        const int c = (int)(t % (C + 1));
        t /= (C + 1);
        const int k = (int)(t % 8);
        const int b = (int)(t / 8);
        if (c < C) {
            planes_out[(((size_t)b * 8 + k) * C + c) * HW + p] = planes[((size_t)b * C + c) * HW + perm_s[k * HW + p]];
        } else {
            pi_out[((size_t)b * 8 + k) * HW + p] = pi[(size_t)b * HW + perm_p[k * HW + p]];
        }
    }
}

}  // namespace apz
