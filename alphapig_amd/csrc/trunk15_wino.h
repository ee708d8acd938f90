// Trunk 3x3 convolution (128 -> 128 channels, 15x15 board) + folded BN + (residual) + ReLU as a
// fused F(4x4,3x3) Winograd convolution on the fp32 matrix cores.  gfx950 only.
//
// Why: the direct kernel (trunk15_ring.h) sits at ~85 % of the fp32 MFMA peak, so the only way
// left to make a leaf evaluation cheaper in fp32 is to issue fewer MFMAs.  A 15x15 board is
// covered by 4x4 output tiles of 4x4 pixels = 16 tiles: EXACTLY one 16-wide MFMA operand.  Per
// board the convolution becomes 36 independent GEMMs (one per position of the 6x6 transformed
// tile)   M[pos][co][tile] = sum_ci U[pos][co][ci] * V[pos][ci][tile]
// = 36 x 8 co-tiles x 32 k-steps = 9216 v_mfma_f32_16x16x4_f32 per board instead of 34560
// (3.75x fewer).  fp32 error of the whole 20-layer trunk stays ~1e-5 on the logits (measured:
// tests/winograd_numerics.py; tolerance of the path is 1e-4).
//
// One workgroup (8 waves, two per SIMD) owns a board at a time and keeps ALL of its
// 36 x 128 x 16 accumulators in registers (wave w: output channels 16w..16w+15, 144 accumulator
// registers per lane); it is persistent over boards.  Input channels stream through in chunks of 16:
//   global (rows16 planes) --regs--> raw LDS tile (zero halo) --B^T d B--> V LDS --MFMA--> acc
// each stage one chunk ahead of the next, ONE barrier per chunk:
//   iteration g:  [barrier]  raw(g+2) regs -> LDS;  issue loads raw(g+3);
//                 transform raw(g+1) -> V[(g+1)&1]   and   MFMA over V[g&1]
// The two waves of a SIMD run those last two steps in OPPOSITE order, so one wave's transform
// (VALU + LDS) executes under the other wave's MFMAs.
// The transformed weights U (2.36 MB per layer, G g G^T of the BN-folded weights computed in
// double on the host) stream from L2 straight into MFMA A operands: 16 B per lane feed 4 MFMAs;
// a 9-deep register ring hides the L2 latency.  Epilogue: A^T M A in registers (a lane holds all
// 36 positions of its (channel, tile)), + bias (+ residual), ReLU, 16-byte stores.
//
// Layouts.  in / resid / out: rows16 [n][128][15][16] (col 15 == 0), as trunk15_ring.h.
// upk: [cot 8][chunk 8][pos 36][lane 64][4]: element s of lane (q = lane>>4, j = lane&15) is
//      U[pos][co = cot*16 + j][ci = chunk*16 + 4*s + q].
#pragma once
#include <hip/hip_runtime.h>

#include "conv3x3_mfma.h"

namespace apz {

struct Wino15 {
    static constexpr int C = 128, CK = 16, NCHUNK = C / CK, NPOS = 36;
    static constexpr int GPLANE = 240;                 // floats per plane in HBM (15 rows x 16)
    static constexpr int LPS = 272;                    // LDS plane stride: 240 data + 32 zeros (rows 15, 16 == row -1 of the next)
    static constexpr int RAW_FRONT = 32;               // zeros in front of plane 0 (its row -1 / col -1)
    static constexpr int RAW_FLOATS = RAW_FRONT + CK * LPS;          // 4384
    static constexpr int V_FLOATS = NPOS * CK * 16;                   // 9216
    static constexpr int LDS_FLOATS = 2 * RAW_FLOATS + 2 * V_FLOATS;  // 27200 floats = 106.25 KiB
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr size_t UPK_FLOATS = (size_t)NPOS * C * C;        // per layer
    static constexpr int PF = 9;                       // weight prefetch depth (positions)
};

// 1-D input transform B^T x of F(4,3) (interpolation points 0, +-1, +-2, inf)
__device__ __forceinline__ void wino_bt6(const float x0, const float x1, const float x2, const float x3, const float x4,
                                         const float x5, float* y) {
    const float a = __builtin_fmaf(-4.f, x2, x4);      // x4 - 4 x2
    const float b = __builtin_fmaf(-4.f, x1, x3);      // x3 - 4 x1
    const float c = x4 - x2;
    const float d = x3 - x1;
    y[0] = __builtin_fmaf(4.f, x0, __builtin_fmaf(-5.f, x2, x4));
    y[1] = a + b;
    y[2] = a - b;
    y[3] = __builtin_fmaf(2.f, d, c);
    y[4] = __builtin_fmaf(-2.f, d, c);
    y[5] = __builtin_fmaf(4.f, x1, __builtin_fmaf(-5.f, x3, x5));
}

// 1-D output transform A^T m of F(4,3)
__device__ __forceinline__ void wino_at6(const float m0, const float m1, const float m2, const float m3, const float m4,
                                         const float m5, float* o) {
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    o[0] = (m0 + s12) + s34;
    o[1] = __builtin_fmaf(2.f, d34, d12);
    o[2] = __builtin_fmaf(4.f, s34, s12);
    o[3] = __builtin_fmaf(8.f, d34, d12) + m5;
}

template <bool RESID>
__global__ __launch_bounds__(512) void trunk15_wino_kernel(const float* __restrict__ in, const float* __restrict__ upk,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ resid, float* __restrict__ out,
                                                           int n) {
    using T = Wino15;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* rawb = lds;                          // [2][RAW_FLOATS]
    float* vb = lds + 2 * T::RAW_FLOATS;        // [2][V_FLOATS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, j = lane & 15;

    for (int i = tid * 4; i < 2 * T::RAW_FLOATS; i += 2048) *reinterpret_cast<f32x4*>(&lds[i]) = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nb = (n - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // boards of this workgroup
    const int total_chunks = nb * T::NCHUNK;

    // ---- staging roles: thread -> (plane = tid>>5, 16-byte pieces (tid&31) + 32u, u < 2; 60 pieces per plane)
    const int st_plane = tid >> 5, st_piece = tid & 31;
    f32x4 rg[2];
    auto raw_fetch = [&](int g) {               // global -> registers (chunk g of this workgroup's stream)
        if (g < total_chunks) {
            const int board = (int)blockIdx.x + (g >> 3) * (int)gridDim.x;
            const float* src = in + ((size_t)board * T::C + (g & 7) * T::CK + st_plane) * T::GPLANE + st_piece * 4;
            rg[0] = *reinterpret_cast<const f32x4*>(src);
            if (st_piece < 28) rg[1] = *reinterpret_cast<const f32x4*>(src + 128);
        }
    };
    auto raw_store = [&](int g) {               // registers -> raw LDS buffer g&1
        float* dst = rawb + (g & 1) * T::RAW_FLOATS + T::RAW_FRONT + st_plane * T::LPS + st_piece * 4;
        *reinterpret_cast<f32x4*>(dst) = rg[0];
        if (st_piece < 28) *reinterpret_cast<f32x4*>(dst + 128) = rg[1];
    };
    // ---- transform roles: thread -> (channel = tid>>5, tile = (tid>>1)&15 = 4*ty + tx, half = tid&1):
    // both threads of a pair run the row pass of the whole 6x6 patch, each finishes three of the
    // six rows of B^T d B.
    const int ty = (tid >> 3) & 3, tx = (tid >> 1) & 3;
    const bool hi = tid & 1;
    const int tr_off = T::RAW_FRONT + (tid >> 5) * T::LPS + (4 * ty - 1) * 16 + 4 * tx - 1;
    const int tv_off = (hi ? 18 * 256 : 0) + (tid >> 1);
    auto transform = [&](int g) {               // raw[g&1] -> V[g&1]
        const float* rp = rawb + (g & 1) * T::RAW_FLOATS + tr_off;
        float* vp = vb + (g & 1) * T::V_FLOATS + tv_off;
        float t[6][3];                          // t[k][ii]: rows 3*half + ii of column k after the row pass
#pragma unroll
        for (int k = 0; k < 6; k++) {
            float x[6], y[6];
#pragma unroll
            for (int i = 0; i < 6; i++) x[i] = rp[i * 16 + k];
            if (k == 5) {                       // column 16 does not exist (the LDS word is the next row's column 0)
#pragma unroll
                for (int i = 0; i < 6; i++) x[i] = (tx == 3) ? 0.f : x[i];
            }
            wino_bt6(x[0], x[1], x[2], x[3], x[4], x[5], y);
#pragma unroll
            for (int ii = 0; ii < 3; ii++) t[k][ii] = hi ? y[3 + ii] : y[ii];
        }
#pragma unroll
        for (int ii = 0; ii < 3; ii++) {
            float y[6];
            wino_bt6(t[0][ii], t[1][ii], t[2][ii], t[3][ii], t[4][ii], t[5][ii], y);
#pragma unroll
            for (int k = 0; k < 6; k++) vp[(ii * 6 + k) * 256] = y[k];
        }
    };

    // ---- prologue: raw(0), raw(1) in LDS, V(0) transformed, raw(2) in registers
    __syncthreads();                            // zero fill done
    raw_fetch(0);
    raw_store(0);
    raw_fetch(1);
    raw_store(1);
    __syncthreads();
    transform(0);
    raw_fetch(2);

    // weight stream of this wave: step (chunk, pos) -> one f32x4 per lane
    const f32x4* ubase = reinterpret_cast<const f32x4*>(upk) + (size_t)wave * (T::NCHUNK * T::NPOS * 64) + lane;
    f32x4 wq[T::PF];
#pragma unroll
    for (int p = 0; p < T::PF; p++) wq[p] = ubase[p * 64];

    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + wave * 16 + q * 4);
    const bool mfma_first = wave >= 4;          // the second wave of each SIMD

    for (int bi = 0; bi < nb; bi++) {
        const int board = (int)blockIdx.x + bi * (int)gridDim.x;
        f32x4 acc[T::NPOS];
#pragma unroll
        for (int p = 0; p < T::NPOS; p++) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int c = 0; c < T::NCHUNK; c++) {
            const int g = bi * T::NCHUNK + c;
            __syncthreads();                    // V[g&1] complete, V[(g+1)&1] and raw[g&1] free, raw[(g+1)&1] visible
            if (g + 2 < total_chunks) raw_store(g + 2);
            raw_fetch(g + 3);
            if (!mfma_first && g + 1 < total_chunks) transform(g + 1);

            const float* vp = vb + (g & 1) * T::V_FLOATS + lane;
            const f32x4* unext = ubase + (size_t)((c + 1) & 7) * (T::NPOS * 64);
            const f32x4* ucur = ubase + (size_t)c * (T::NPOS * 64);
#pragma unroll
            for (int p = 0; p < T::NPOS; p++) {
                float b[4];
#pragma unroll
                for (int s = 0; s < 4; s++) b[s] = vp[p * 256 + s * 64];
                const f32x4 w0 = wq[p % T::PF];
#pragma unroll
                for (int s = 0; s < 4; s++) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[s], b[s], acc[p], 0, 0, 0);
                // refill the ring slot: position p + PF of this chunk or of the next one (the stream
                // wraps to the next board's chunk 0: same weights)
                wq[p % T::PF] = (p + T::PF < T::NPOS) ? ucur[(p + T::PF) * 64] : unext[(p + T::PF - T::NPOS) * 64];
            }
            if (mfma_first && g + 1 < total_chunks) transform(g + 1);
        }

        // ---- epilogue: output transform in registers.  Lane (q, j): tile j = 4*ty + tx, channels co0 + r.
        const int ety = j >> 2, etx = j & 3;
        const int co0 = wave * 16 + q * 4;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const size_t pbase = ((size_t)board * T::C + co0 + r) * T::GPLANE + (4 * ety) * 16 + 4 * etx;
            f32x4 rs[4];
            if (RESID) {
#pragma unroll
                for (int a = 0; a < 4; a++)
                    rs[a] = (4 * ety + a < 15) ? *reinterpret_cast<const f32x4*>(resid + pbase + a * 16)
                                               : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            float h[6][4];                      // h[k][a]: column k after the row-direction transform
#pragma unroll
            for (int k = 0; k < 6; k++)
                wino_at6(acc[0 * 6 + k][r], acc[1 * 6 + k][r], acc[2 * 6 + k][r], acc[3 * 6 + k][r], acc[4 * 6 + k][r],
                         acc[5 * 6 + k][r], h[k]);
            const float bvr = bv[r];
#pragma unroll
            for (int a = 0; a < 4; a++) {
                float o[4];
                wino_at6(h[0][a], h[1][a], h[2][a], h[3][a], h[4][a], h[5][a], o);
                f32x4 v;
#pragma unroll
                for (int b2 = 0; b2 < 4; b2++) {
                    float y = o[b2] + bvr;
                    if (RESID) y += rs[a][b2];
                    v[b2] = fmaxf(y, 0.f);
                }
                if (etx == 3) v[3] = 0.f;       // column 15 is the halo column of the rows16 layout
                if (4 * ety + a < 15) *reinterpret_cast<f32x4*>(out + pbase + a * 16) = v;
            }
        }
    }
}

}  // namespace apz
