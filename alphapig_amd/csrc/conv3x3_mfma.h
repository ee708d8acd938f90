// 3x3 / pad 1 / stride 1 convolution + folded BatchNorm + (residual) + ReLU for gfx950.
//
// im2col-free direct convolution: one workgroup owns one board.  The board's whole input
// ([Cin][H][W], <= 136 KB at 128x15x15) is staged ONCE into LDS in a zero-padded layout
//     element (c, y, x)  ->  c*PS + (y+1)*RS + (x+1),     RS = W+1, PS % 32 == 16
// where the single pad column of every row doubles as the left halo of that row and the
// right halo of the previous one.  The nine taps are then just nine shifted reads of that
// tile (immediate offsets), and the contraction over (ci, ky, kx) is issued on the fp32
// matrix pipe: v_mfma_f32_16x16x4_f32 computes D[16 co][16 px] += A[16 co][4 ci] * B[4 ci][16 px],
// bit-for-bit an fmaf chain (exact fp32, no reduced precision), at the chip's full fp32
// rate (157 TF) which the VALU only reaches on paper.
//
//   A  (weights)      lane l: co = l&15, ci = l>>4   -- streamed from L2, pre-packed so that
//                                                       every fragment is one coalesced
//                                                       256-B wave load
//   B  (activations)  lane l: px = l&15, ci = l>>4   -- one ds_read_b32 per fragment; a
//                                                       fragment = one board row (W=15: 15
//                                                       pixels + the pad) or two rows (W<=8)
//   D  lane l, reg r: co = (l>>4)*4 + r, px = l&15
//
// Wave w of the 4 waves owns output channels [w*16*CT, (w+1)*16*CT) for all pixel tiles,
// i.e. CT x NT accumulator tiles (2 x 15 x 4 = 120 VGPRs at 128 channels, 15x15).  For a
// fixed kx the H+2 padded rows are read once and reused by the three ky taps.
// Bank check (ds_read_b32, 32-lane groups): lanes 0-15 hit banks a..a+15, lanes 16-31 are
// one channel plane further, PS % 32 == 16 -> banks a+16..a+31: conflict-free.
#pragma once
#include <hip/hip_runtime.h>

namespace apz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int H_, int W_>
struct ConvGeo {
    static constexpr int H = H_, W = W_;
    static constexpr int RS = W + 1;                        // row stride (floats)
    static constexpr int RPT = (W <= 8) ? 2 : 1;            // board rows per 16-lane tile
    static constexpr int WT = 16 / RPT;                     // lanes per row inside a tile
    static constexpr int NT = (H + RPT - 1) / RPT;          // pixel tiles
    static constexpr int PLANE = (H + 2) * RS;
    static constexpr int PS = ((PLANE + 15) / 32) * 32 + 16;  // >= PLANE, == 16 (mod 32)
    static constexpr int NFRAG = (NT - 1) * RPT + 3;        // distinct first rows per kx
    static constexpr int SLACK = 64;                        // floats; over-reads of dead lanes
    static_assert(W <= 15 && H <= 16, "tile mapping assumes W <= 15");
    static_assert(PS >= PLANE && PS % 32 == 16, "plane stride");
    static int lds_bytes(int cchunk) { return (cchunk * PS + SLACK) * (int)sizeof(float); }
    // largest multiple-of-4 channel count whose tile fits the 160 KiB LDS
    static int max_chunk() { return ((160 * 1024 / (int)sizeof(float) - SLACK) / PS) & ~3; }
};

// wpk layout: [Cout/16][Cin_pad/4][9][64]  (see pack_conv3x3 in apz_engine.hip)
template <int H, int W, int CT, bool RESID>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(const float* __restrict__ in,
                                                           const float* __restrict__ wpk,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ resid,
                                                           float* __restrict__ out, int n, int cin,
                                                           int cin_pad, int cchunk, int relu,
                                                           int out_ps, int out_rs, int cout_total) {
    using G = ConvGeo<H, W>;
    constexpr int HW = H * W;
    extern __shared__ __attribute__((aligned(16))) float tile[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int q = lane >> 4;          // k slot (input channel within the group of 4)
    const int j = lane & 15;          // pixel lane
    // gridDim.y > 1: this workgroup computes output channels [blockIdx.y*64*CT, +64*CT) of cout_total
    // (small batches: two or four workgroups per board keep all CUs busy)
    const int cout = cout_total;
    const int wtile0 = blockIdx.y * 4 * CT + wave * CT;   // first 16-channel tile of this wave
    const int n4 = cin_pad >> 2;
    const int lds_floats = cchunk * G::PS + G::SLACK;   // cchunk: channels resident at once

    // zero the whole tile once: pads / halo rows / padded channels are never written again
    for (int i = tid * 4; i < lds_floats; i += 256 * 4) {
        *reinterpret_cast<f32x4*>(&tile[i]) = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int lane_off = q * G::PS + (j / G::WT) * G::RS + (j % G::WT);
    const int px = j % G::WT, prow = j / G::WT;

    for (int b = blockIdx.x; b < n; b += gridDim.x) {
        f32x4 acc[CT][G::NT];
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int t = 0; t < G::NT; t++) acc[ct][t] = f32x4{0.f, 0.f, 0.f, 0.f};

        // channel chunks: all of Cin when it fits LDS (<=128 ch at 15x15), else 2+ passes
        for (int c0 = 0; c0 < cin_pad; c0 += cchunk) {
            __syncthreads();   // previous chunk / board fully consumed (and the zero fill done)
            // ---- stage channels [c0, c0+cchunk) of the board: dense [cin][H][W] -> padded LDS tile
            {
                const int cn = min(cchunk, cin - c0);
                const float* src = in + ((size_t)b * cin + c0) * HW;
                const int total = cn * HW;
                const int head = (int)((4 - ((((size_t)b * cin + c0) * HW) & 3)) & 3);   // to 16-B alignment
                for (int idx = tid; idx < min(head, total); idx += 256) {
                    const int c = idx / HW, rem = idx - c * HW;
                    const int y = rem / W, x = rem - y * W;
                    tile[c * G::PS + (y + 1) * G::RS + x + 1] = src[idx];
                }
                const int body4 = (total > head) ? ((total - head) & ~3) : 0;
                for (int e = tid * 4; e < body4; e += 256 * 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src + head + e);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int idx = head + e + u;
                        const int c = idx / HW, rem = idx - c * HW;
                        const int y = rem / W, x = rem - y * W;
                        tile[c * G::PS + (y + 1) * G::RS + x + 1] = v[u];
                    }
                }
                for (int idx = head + body4 + tid; idx < total; idx += 256) {
                    const int c = idx / HW, rem = idx - c * HW;
                    const int y = rem / W, x = rem - y * W;
                    tile[c * G::PS + (y + 1) * G::RS + x + 1] = src[idx];
                }
            }
            __syncthreads();

            // A fragments of this wave's CT channel tiles, double-buffered over ci4
            const int c4_lo = c0 >> 2, c4_hi = min(cin_pad, c0 + cchunk) >> 2;
            const float* wbase = wpk + ((size_t)wtile0 * n4 * 9) * 64 + lane;
            float a_cur[CT][9], a_nxt[CT][9];
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int tap = 0; tap < 9; tap++)
                    a_cur[ct][tap] = wbase[(((size_t)ct * n4 + c4_lo) * 9 + tap) * 64];

            const float* bptr = tile + lane_off;
            for (int c4 = c4_lo; c4 < c4_hi; c4++) {
                const int c4n = (c4 + 1 < c4_hi) ? c4 + 1 : c4;
#pragma unroll
                for (int ct = 0; ct < CT; ct++)
#pragma unroll
                    for (int tap = 0; tap < 9; tap++)
                        a_nxt[ct][tap] = wbase[(((size_t)ct * n4 + c4n) * 9 + tap) * 64];
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    float r[G::NFRAG];
#pragma unroll
                    for (int f = 0; f < G::NFRAG; f++) r[f] = bptr[f * G::RS + kx];
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int t = 0; t < G::NT; t++)
#pragma unroll
                            for (int ct = 0; ct < CT; ct++)
                                acc[ct][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    a_cur[ct][ky * 3 + kx], r[t * G::RPT + ky], acc[ct][t], 0, 0, 0);
                }
#pragma unroll
                for (int ct = 0; ct < CT; ct++)
#pragma unroll
                    for (int tap = 0; tap < 9; tap++) a_cur[ct][tap] = a_nxt[ct][tap];
                bptr += 4 * G::PS;
            }
        }

        // ---- epilogue: + folded bias (+ residual), ReLU; output plane stride out_ps, row stride
        //      out_rs (dense NCHW: H*W / W; rows16 hand-off to trunk15_ring.h: 240 / 16)
        float* dst = out + (size_t)b * cout * out_ps;
        const float* rsd = RESID ? resid + (size_t)b * cout * out_ps : nullptr;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const int co0 = (wtile0 + ct) * 16 + q * 4;
            float bv[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) bv[rr] = bias[co0 + rr];
#pragma unroll
            for (int t = 0; t < G::NT; t++) {
                const int y = t * G::RPT + prow;
                if (px < W && y < H) {
                    const int p = y * out_rs + px;
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        float v = acc[ct][t][rr] + bv[rr];
                        if (RESID) v += rsd[(co0 + rr) * out_ps + p];
                        if (relu) v = fmaxf(v, 0.f);
                        dst[(co0 + rr) * out_ps + p] = v;
                    }
                }
            }
        }
    }
}

}  // namespace apz
