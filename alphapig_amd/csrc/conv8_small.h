// 8x8 boards (BASELINE config 2: the 6-conv "simple" net, 64 concurrent games = 32-board launches; also the 8x8 residual
// nets): 3x3 convolution + folded BatchNorm (+ residual) + ReLU and the fused policy / value heads for gfx950.
// Reference graph: policy_value_net_mxnet_simple.py:68-92 (conv_act x 6, then the heads of :85-97).
//
// Why not conv3x3_mfma_kernel (one workgroup per board x 64 output channels): at 32 boards that is 32..128 workgroups of
// one wave per SIMD, each walking its whole K = C_in x 9 contraction alone -- 70 us for the 256 -> 256 layer where the
// chip-wide MFMA time is 15 us, 213 us of kernels per 32-board forward in 9 launches (profiles/r03_config2.md).
//
// conv8_kernel.  Work item = (board, 16 output channels): n x C_out / 16 items, every one the same size whatever the
// batch, so a board's bits never depend on the launch shape.  The four waves of a workgroup SPLIT THE CONTRACTION: wave w
// takes the quarter [w n4 / 4, (w + 1) n4 / 4) of the C_in / 4 k-steps for all 64 pixels (4 MFMA pixel tiles of two board
// rows each, 16 accumulator registers) -- every weight fragment is fetched by exactly one wave of one workgroup and feeds
// four MFMAs, and the 256 -> 256 layer at 32 boards is 512 workgroups / 2 048 waves = two waves on every SIMD of the
// chip.  Each wave stages ITS input channels itself, 16 at a time (global -> registers one sub-chunk ahead -> a
// wave-private LDS tile), so the main loop has no workgroup barrier at all; the four partial sums meet once in LDS and
// are added in wave order (fixed order: deterministic), + bias (+ residual), ReLU, and leave as one coalesced 4 KB block
// (16 channels x 64 pixels are contiguous in dense NCHW).
// CODES: the first layer decodes the 65-byte position codes (Board.current_state, game.py:68-94, vertical flip
// included) straight into its LDS tile -- no encode kernel, no planes buffer on the self-play path.
//
// LDS tile of a wave: [16 ch][10 rows][16] floats, plane stride 168 (== 8 mod 32: the four k-lanes of a B fragment sit
// on disjoint banks): board row y at tile row y + 1, column x at 4 + x -- every global 16-byte piece (four pixels of a
// row) is one aligned ds_write_b128, the zero border (rows 0 / 9, columns 3 / 12) is written once per launch.
//   B fragment (kx, first row f): lane (q = lane >> 4, j = lane & 15) reads tile[q][f + (j >> 3)][3 + kx + (j & 7)].
//   A fragments: wpk12 [C_out / 16][n4][lane 64][12]: lane (q, j) holds the nine taps of co = 16 cot + j, ci = 4 c4 + q
//   as three 16-byte loads.
//
// head8_kernel: both 1x1 head convolutions + BN + ReLU, the policy FullyConnected + softmax and the value FullyConnected
// + tanh for ONE board per workgroup, in one launch (was two launches, 11 + 12 us at 32 boards, both latency chains).
#pragma once
#include <hip/hip_runtime.h>

#include "wino_common.h"

namespace apz {

struct Conv8 {
    static constexpr int HW = 64;
    static constexpr int RS = 16, PS = 168;              // tile row / plane stride (floats)
    static constexpr int SUB = 16;                       // input channels per sub-chunk (4 k-steps)
    static constexpr int WAVE_FLOATS = SUB * PS + 32;    // + slack for the over-read of dead lanes
    static constexpr int RED_CS = 68;                    // channel stride of the reduction area (64 pixels + 4)
    static constexpr int RED_FLOATS = 4 * 16 * RED_CS;
    static constexpr int LDS_FLOATS = 4 * WAVE_FLOATS + RED_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;     // 60.9 KB: two workgroups per CU
    static_assert(PS % 32 == 8 && PS >= 10 * RS, "plane stride");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

// in: dense [n][cin][64] floats, or (CODES) position codes [n][code_stride] bytes; out / resid: dense [n][cout][64].
template <bool RESID, bool CODES>
__global__ __launch_bounds__(256) void conv8_kernel(const float* __restrict__ in, const float* __restrict__ wpk12,
                                                    const float* __restrict__ bias, const float* __restrict__ resid,
                                                    float* __restrict__ out, int n, int cin, int n4, int cout, int relu,
                                                    int code_stride) {
    using T = Conv8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, j = lane & 15;
    float* tile = lds + wave * T::WAVE_FLOATS;
    float* red = lds + 4 * T::WAVE_FLOATS;

    // zero the wave's tile once: border cells and padded channels are never written with anything else
    for (int i = lane * 4; i < T::WAVE_FLOATS; i += 256) *reinterpret_cast<f32x4*>(tile + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    wave_lds_fence();

    const int ncot = cout >> 4;
    const int nitems = n * ncot;
    // this wave's share of the contraction
    const int c4_lo = (wave * n4) >> 2, c4_hi = ((wave + 1) * n4) >> 2;
    const int nsub = (c4_hi - c4_lo + 3) >> 2;
    // staging role of a lane: sub-chunk piece i (0..3) = channel 4 i + (lane >> 4), row (lane & 15) >> 1, columns 4 (lane & 1)..
    const int st_ch = lane >> 4, st_off = ((lane & 15) >> 1) * T::RS + T::RS + 4 + (lane & 1) * 4;
    const float* brd = tile + q * T::PS + (j >> 3) * T::RS + 3 + (j & 7);

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int b = item / ncot, cot = item - b * ncot;
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 pre[4];
        auto fetch = [&](int s) {                   // sub-chunk s of this wave: global -> registers
            if (CODES) return;
            const int ch0 = 4 * (c4_lo + 4 * s);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int ch = ch0 + 4 * i + st_ch;
                pre[i] = (ch < cin && ch < 4 * c4_hi)
                             ? *reinterpret_cast<const f32x4*>(in + ((size_t)b * cin + ch) * T::HW + (lane & 15) * 4)
                             : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto stash = [&]() {                        // registers -> the wave's LDS tile
#pragma unroll
            for (int i = 0; i < 4; i++) *reinterpret_cast<f32x4*>(tile + (4 * i + st_ch) * T::PS + st_off) = pre[i];
        };
        f32x4 wA[4][3], wB[4][3];
        auto wload = [&](f32x4 (&dst)[4][3], int s) {      // the weights of sub-chunk s (k-steps past the wave's range: clamped)
            const int c4_0 = c4_lo + 4 * s, cn = min(4, c4_hi - c4_0);
            const float* wb = wpk12 + (((size_t)cot * n4 + c4_0) * 64 + lane) * 12;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int uu = u < cn ? u : cn - 1;
#pragma unroll
                for (int v = 0; v < 3; v++) dst[u][v] = *reinterpret_cast<const f32x4*>(wb + (size_t)uu * 768 + 4 * v);
            }
        };
        if (nsub > 0) wload(wA, 0);
        if (CODES) {
            // <= 3 k-steps in the whole layer: this wave's four planes straight from the position codes.
            // cell m = lane = h * 8 + w lands at board row 7 - h (the vertical flip of game.py:94)
            if (c4_lo < c4_hi) {
                const unsigned char* cb = reinterpret_cast<const unsigned char*>(in) + (size_t)b * code_stride;
                const int code = cb[lane];
                const float colour = cb[T::HW] ? 1.f : 0.f;
                const int opp = code >= 5, age = (code - 1) & 3;
                const int h = lane >> 3, w = lane & 7;
                float* dst = tile + (8 - h) * T::RS + 4 + w;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int p = 4 * c4_lo + u;    // plane index
                    float v = 0.f;
                    if (cin == 9) {
                        if (p == 8) v = colour;
                        else if (p < 8) {
                            const int k = (7 - p) >> 1, is_opp = p & 1;      // planes 6 - 2k (own), 7 - 2k (opponent)
                            v = (code && k <= age && opp == is_opp) ? 1.f : 0.f;
                        }
                    } else {                        // the 4-plane encoder (game.py:96-115)
                        v = p == 0 ? ((code && !opp) ? 1.f : 0.f)
                            : p == 1 ? ((code && opp) ? 1.f : 0.f)
                            : p == 2 ? ((code && age == 0) ? 1.f : 0.f)
                                     : colour;
                    }
                    dst[u * T::PS] = v;
                }
            }
        } else if (nsub > 0) {
            fetch(0);
            stash();
        }
        wave_lds_fence();
        // One sub-chunk = up to four k-steps x (9 taps x 4 pixel tiles) = 144 MFMAs (~2 us with the SIMD's other wave in
        // between).  Its 4 x 3 weight pieces are requested a whole sub-chunk ahead -- one k-step ahead (0.5 us) is less than
        // an L2 round trip under load -- into the buffer the sub-chunk before last has finished with (two named buffers,
        // the loop unrolled by two: no register copies).
        auto compute = [&](const f32x4 (&w)[4][3], int s) {
            const int c4_n = min(4, c4_hi - (c4_lo + 4 * s));
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (u < c4_n) {
                    const float* bp = brd + u * 4 * T::PS;
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        float r[9];
#pragma unroll
                        for (int f = 0; f < 9; f++) r[f] = bp[f * T::RS + kx];
#pragma unroll
                        for (int ky = 0; ky < 3; ky++) {
                            const int tap = ky * 3 + kx;
                            const float av = w[u][tap >> 2][tap & 3];
#pragma unroll
                            for (int t = 0; t < 4; t++)
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, r[2 * t + ky], acc[t], 0, 0, 0);
                        }
                    }
                }
            }
        };
        auto turn = [&](const f32x4 (&w)[4][3], f32x4 (&wnext)[4][3], int s) {     // sub-chunk s from `w`; s + 1 prepared
            if (s + 1 < nsub) {
                fetch(s + 1);
                wload(wnext, s + 1);
            }
            compute(w, s);
            if (s + 1 < nsub) {
                wave_lds_fence();                   // this wave's reads of the tile are done
                stash();
                wave_lds_fence();
            }
        };
        for (int s = 0; s < nsub; s += 2) {
            turn(wA, wB, s);
            if (s + 1 < nsub) turn(wB, wA, s + 1);
        }
        // ---- the four partial sums meet: red[wave][co 16][64 px (+4)]; lane (q, j) reg r = co 4 q + r, pixel 16 t + j
        __syncthreads();                            // the previous item's reduction reads are done
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[(wave * 16 + 4 * q + r) * T::RED_CS + 16 * t + j] = acc[t][r];
        __syncthreads();
        {
            const int co = tid >> 4, p4 = (tid & 15) * 4;
            const float* rp = red + co * T::RED_CS + p4;
            f32x4 sum = *reinterpret_cast<const f32x4*>(rp) + *reinterpret_cast<const f32x4*>(rp + 16 * T::RED_CS);
            sum = sum + *reinterpret_cast<const f32x4*>(rp + 32 * T::RED_CS);
            sum = sum + *reinterpret_cast<const f32x4*>(rp + 48 * T::RED_CS);
            sum = sum + bias[cot * 16 + co];
            const size_t o = ((size_t)b * cout + cot * 16 + co) * T::HW + p4;
            if (RESID) sum = sum + *reinterpret_cast<const f32x4*>(resid + o);
            if (relu)
#pragma unroll
                for (int e = 0; e < 4; e++) sum[e] = fmaxf(sum[e], 0.f);
            *reinterpret_cast<f32x4*>(out + o) = sum;
        }
    }
}

__device__ __forceinline__ float head8_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float head8_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// x: dense [n][C][64] (the last convolution's output).  w6 [6][C] / b6 [6]: both 1x1 convolutions, BatchNorm folded (rows
// 0-3 policy, 4-5 value); wfc [64][256], bfc [64]: policy FullyConnected on the row-major flatten of [4][8][8]
// (fc_3_1_1); wv [128], bv [1]: value FullyConnected (fc_3_2_1).  probs [n][64], values [n]; logits / vlogits optional.
// All sums in a fixed order: a board's bits do not depend on the batch.
__global__ __launch_bounds__(256) void head8_kernel(const float* __restrict__ x, const float* __restrict__ w6,
                                                    const float* __restrict__ b6, const float* __restrict__ wfc,
                                                    const float* __restrict__ bfc, const float* __restrict__ wv,
                                                    const float* __restrict__ bv, float* __restrict__ probs,
                                                    float* __restrict__ values, float* __restrict__ logits_out,
                                                    float* __restrict__ vlogits_out, int n, int C) {
    __shared__ float part[4][6][64];
    __shared__ __attribute__((aligned(16))) float feat[6 * 64];       // [4][64] policy features, then [2][64] value features
    __shared__ float fcp[4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cq = C >> 2;
    for (int b = blockIdx.x; b < n; b += gridDim.x) {
        // ---- 1x1 convolutions: wave = channel quarter, lane = pixel
        {
            const float* xb = x + ((size_t)b * C + wave * cq) * 64 + lane;
            const float* wq = w6 + wave * cq;
            float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int c0 = 0; c0 < cq; c0 += 16) {     // sixteen loads in flight: the phase is a chain of load round trips
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; u++) v[u] = xb[(size_t)(c0 + u) * 64];
#pragma unroll
                for (int u = 0; u < 16; u++)
#pragma unroll
                    for (int o = 0; o < 6; o++) a[o] = fmaf(wq[o * C + c0 + u], v[u], a[o]);
            }
#pragma unroll
            for (int o = 0; o < 6; o++) part[wave][o][lane] = a[o];
        }
        __syncthreads();
        for (int i = tid; i < 6 * 64; i += 256) {
            const int o = i >> 6, p = i & 63;
            const float s = ((part[0][o][p] + part[1][o][p]) + part[2][o][p]) + part[3][o][p];
            feat[i] = fmaxf(s + b6[o], 0.f);
        }
        __syncthreads();
        // ---- policy FullyConnected: wave = quarter of the 256 inputs, lane = output
        {
            const f32x4* wr = reinterpret_cast<const f32x4*>(wfc + (size_t)lane * 256 + wave * 64);
            const f32x4* fr = reinterpret_cast<const f32x4*>(feat + wave * 64);
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const f32x4 w4 = wr[k], f4 = fr[k];
#pragma unroll
                for (int e = 0; e < 4; e++) s = fmaf(w4[e], f4[e], s);
            }
            fcp[wave][lane] = s;
        }
        __syncthreads();
        if (wave == 0) {
            const float lg = (((fcp[0][lane] + fcp[1][lane]) + fcp[2][lane]) + fcp[3][lane]) + bfc[lane];
            const float mx = head8_wave_max(lg);
            const float ex = expf(lg - mx);
            const float sm = head8_wave_sum(ex);
            probs[(size_t)b * 64 + lane] = ex / sm;
            if (logits_out) logits_out[(size_t)b * 64 + lane] = lg;
        } else if (wave == 1) {
            float d = fmaf(feat[256 + lane], wv[lane], 0.f);
            d = fmaf(feat[320 + lane], wv[64 + lane], d);
            d = head8_wave_sum(d) + bv[0];
            if (lane == 0) {
                values[b] = tanhf(d);
                if (vlogits_out) vlogits_out[b] = d;
            }
        }
        __syncthreads();                            // part / feat / fcp are reused by the next board
    }
}

}  // namespace apz
