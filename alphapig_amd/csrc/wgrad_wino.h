// Weight gradient of the trunk convolution (128 -> 128, 15x15) through the Winograd F(4x4,3x3) domain.  gfx950.
//
// Forward (trunk15_wino2.h): Y = A^T [ sum_ci U ⊙ V ] A with U = G g G^T, V = B^T d B.  Hence
//   dU[pos][co][ci] = sum over boards and tiles of  dM[pos][co][tile] * V[pos][ci][tile],   dM = A dY A^T (6x6 from 4x4)
//   dg[co][ci]      = G^T dU G                                                              (3x3 from 6x6)
// i.e. 36 independent [128 x K] x [K x 128] products with K = 16 tiles per board: 9 216 fp32 MFMAs per board
// instead of the 32 832 of the direct form (conv3x3_wgrad_kernel).
//
// A workgroup owns SIX of the 36 positions (row i of the 6x6: 6 groups) for ALL 128 x 128 channel pairs
// (wave w = output channels 16w.., all eight input-channel tiles: 192 accumulator registers) and a slice of
// the batch.  Per board it streams the 128 input planes and the 128 output-gradient
// planes through LDS in chunks of 16 + 16 (padded-row layout of the trunk), transforms each (channel, tile)
// to its six positions (waves 0-3: V from 6x6 input patches, waves 4-7: dM from 4x4 gradient tiles) into
// two LDS operand arrays [pos 6][tile 16][channel 128 (+16)], then issues 6 x 8 x 4 MFMAs per wave.  Partial dU
// of the slices go to a scratch tensor; wgrad_wino_reduce_kernel sums them and applies G^T . G.
// VALU work does not hide under fp32 MFMAs on this chip, so the phases are simply sequential.
#pragma once
#include <hip/hip_runtime.h>

#include "conv3x3_mfma.h"

namespace apz {

struct WgradWino {
    static constexpr int C = 128, CK = 16, NCHUNK = C / CK;
    static constexpr int GPLANE = 240;
    static constexpr int RROW = 20, RPS = 17 * RROW, RFRONT = 24;      // raw input tile: as trunk15_wino2.h
    static constexpr int RAWX_FLOATS = RFRONT + CK * RPS;               // 5464
    static constexpr int RAWY_FLOATS = CK * GPLANE + 16;                // 3856 (gradient planes as stored, + slack)
    static constexpr int OPS = 144;                                     // operand row: 128 channels + 16 (bank spread)
    static constexpr int NPOS = 6;                                      // positions per workgroup: one row of the 6x6
    static constexpr int OP_FLOATS = NPOS * 16 * OPS;                   // 13824 per operand array
    static constexpr int LDS_FLOATS = RAWX_FLOATS + RAWY_FLOATS + 2 * OP_FLOATS;   // 36968 floats = 144.4 KiB
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int GROUPS = 6;
    static constexpr size_t SCRATCH_FLOATS_PER_SLICE = (size_t)36 * C * C;
};

// rows of B^T (6x6) and of A (6x4): the workgroup's position row / columns are runtime values, so the transforms
// are plain coefficient dot products (a `switch` on the row made hipcc evaluate every case and select)
__device__ const float WGW_BT[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                       {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
__device__ const float WGW_A[6][4] = {{1, 0, 0, 0}, {1, 1, 1, 1}, {1, -1, 1, -1}, {1, 2, 4, 8}, {1, -2, 4, -8}, {0, 0, 0, 1}};

// x, dy: padded-row layout [n][128][15][16].  scratch: [slices][36][128 co][128 ci].  grid (12 groups, slices).
__global__ __launch_bounds__(512) void wgrad_wino_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ scratch, int n) {
    using T = WgradWino;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* rawx = lds;                               // [16 planes][17 rows x 20] + front
    float* rawy = lds + T::RAWX_FLOATS;              // [16 planes][240]
    float* opv = rawy + T::RAWY_FLOATS;              // V  [6][16 tiles][144]
    float* opm = opv + T::OP_FLOATS;                 // dM [6][16 tiles][144]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, j = lane & 15;
    const int gi = blockIdx.x;                       // transformed row i: the workgroup's positions are (i, 0..5)
    const int wa = wave >> 2, wb = wave & 3;         // MFMA phase: output-channel half, input-channel quarter

    for (int i = tid; i < T::RAWX_FLOATS; i += 512) rawx[i] = 0.f;   // halo cells stay zero

    f32x4 acc[T::NPOS][8];
#pragma unroll
    for (int p = 0; p < T::NPOS; p++)
#pragma unroll
        for (int t = 0; t < 8; t++) acc[p][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transform roles: waves 0-3 -> input units, waves 4-7 -> gradient units; unit = (channel of the chunk, tile)
    const int unit = tid & 255, uch = unit >> 4, utile = unit & 15, uty = utile >> 2, utx = utile & 3;
    const bool is_x = tid < 256;
    float cbr[6], car[4];                            // runtime coefficient rows: B^T row i, A row i
#pragma unroll
    for (int a = 0; a < 6; a++) cbr[a] = WGW_BT[gi][a];
#pragma unroll
    for (int a = 0; a < 4; a++) car[a] = WGW_A[gi][a];
    constexpr float cbc[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                 {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};      // B^T, all rows
    constexpr float cac[6][4] = {{1, 0, 0, 0}, {1, 1, 1, 1}, {1, -1, 1, -1}, {1, 2, 4, 8}, {1, -2, 4, -8}, {0, 0, 0, 1}};

    // staging pipeline: the 16 + 16 planes of a chunk are 2 x 960 pieces of 16 B = 4 per thread; the pieces of the
    // TWO following chunks are in flight in registers (192 accumulators leave room for no more)
    f32x4 pq[2][4];
    // (every thread issues exactly four loads and four LDS stores per chunk, unconditionally -- past the end the last
    // board is re-read, threads 448..511 repeat their first pieces -- so that hipcc can count vmcnt exactly: with a
    // conditional load it waits for ALL outstanding loads at every use and the four-deep pipeline collapses)
    const int v1 = tid < 448 ? tid + 512 : tid;
    const int pl0 = tid / 60, k0 = tid - pl0 * 60, pl1 = v1 / 60, k1 = v1 - pl1 * 60;
    const int dx0 = T::RFRONT + pl0 * T::RPS + (k0 >> 2) * T::RROW + (k0 & 3) * 4;
    const int dx1 = T::RFRONT + pl1 * T::RPS + (k1 >> 2) * T::RROW + (k1 & 3) * 4;
    auto fetch = [&](int bb, int cc, int slot) {
        bb = bb < n ? bb : n - 1;
        const f32x4* xs = reinterpret_cast<const f32x4*>(x + ((size_t)bb * T::C + cc * T::CK) * T::GPLANE);
        const f32x4* ys = reinterpret_cast<const f32x4*>(dy + ((size_t)bb * T::C + cc * T::CK) * T::GPLANE);
        pq[slot][0] = xs[tid];
        pq[slot][1] = xs[v1];
        pq[slot][2] = ys[tid];
        pq[slot][3] = ys[v1];
    };
    auto put = [&](int slot) {
        *reinterpret_cast<f32x4*>(rawx + dx0) = pq[slot][0];
        *reinterpret_cast<f32x4*>(rawx + dx1) = pq[slot][1];
        reinterpret_cast<f32x4*>(rawy)[tid] = pq[slot][2];
        reinterpret_cast<f32x4*>(rawy)[v1] = pq[slot][3];
    };
#pragma unroll
    for (int d = 0; d < 2; d++) fetch((int)blockIdx.y, d, d);

    for (int b = blockIdx.y; b < n; b += gridDim.y) {
        for (int half = 0; half < 4; half++) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int c = half * 2 + u;
            __syncthreads();                         // previous chunk's raw tiles consumed (and the MFMA phase done)
            put(u);
            if (half < 3)
                fetch(b, c + 2, u);                  // two chunks ahead: same board ...
            else
                fetch(b + (int)gridDim.y, u, u);     // ... or the first chunks of this workgroup's next board
            __syncthreads();
            // ---- transform this chunk's units to the workgroup's three positions
            if (is_x) {
                // the 6x6 patch row by row: column -1, columns 0..3 (one aligned 16-byte read), column 4
                const float* rp = rawx + T::RFRONT + uch * T::RPS + (4 * uty - 1) * T::RROW + 4 * utx;
                float r[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // row i of B^T d, six columns
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    const float cm1 = rp[a * T::RROW - 1];
                    const f32x4 c03 = *reinterpret_cast<const f32x4*>(rp + a * T::RROW);
                    const float c4 = rp[a * T::RROW + 4];
                    r[0] = __builtin_fmaf(cbr[a], cm1, r[0]);
                    r[1] = __builtin_fmaf(cbr[a], c03[0], r[1]);
                    r[2] = __builtin_fmaf(cbr[a], c03[1], r[2]);
                    r[3] = __builtin_fmaf(cbr[a], c03[2], r[3]);
                    r[4] = __builtin_fmaf(cbr[a], c03[3], r[4]);
                    r[5] = __builtin_fmaf(cbr[a], c4, r[5]);
                }
#pragma unroll
                for (int kk = 0; kk < 6; kk++) {
                    float o = 0.f;
#pragma unroll
                    for (int k = 0; k < 6; k++)
                        if (cbc[kk][k] != 0.f) o = __builtin_fmaf(cbc[kk][k], r[k], o);
                    opv[(kk * 16 + utile) * T::OPS + c * T::CK + uch] = o;
                }
            } else {
                const float* yp = rawy + uch * T::GPLANE + (4 * uty) * 16 + 4 * utx;
                f32x4 r4 = f32x4{0.f, 0.f, 0.f, 0.f};   // row i of A dY, four columns (one 16-byte read per tile row)
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(yp + (4 * uty + a < 15 ? a : 0) * 16);
                    r4 += (4 * uty + a < 15 ? car[a] : 0.f) * v;
                }
                const float r[4] = {r4[0], r4[1], r4[2], r4[3]};
#pragma unroll
                for (int kk = 0; kk < 6; kk++) {
                    float o = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (cac[kk][k] != 0.f) o = __builtin_fmaf(cac[kk][k], r[k], o);
                    opm[(kk * 16 + utile) * T::OPS + c * T::CK + uch] = o;
                }
            }
        }
        }
        __syncthreads();                             // both operand arrays complete for board b
        // ---- dU[pos][co][ci] += dM[pos][co][tile] * V[pos][ci][tile]:  A = dM (m = co), B = V (n = ci), k = tile.
        // wave = (co half wa: four 16-channel tiles, ci quarter wb: two tiles): 4 + 2 operand reads per 8 MFMAs
#pragma unroll
        for (int p = 0; p < T::NPOS; p++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const float* om = opm + (p * 16 + 4 * s + q) * T::OPS + wa * 64 + j;
                const float* ov = opv + (p * 16 + 4 * s + q) * T::OPS + wb * 32 + j;
                const float b0 = ov[0], b1 = ov[16];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const float a = om[t * 16];
                    acc[p][2 * t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[p][2 * t], 0, 0, 0);
                    acc[p][2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[p][2 * t + 1], 0, 0, 0);
                }
            }
    }
    // ---- partial dU of this slice: accumulator (t, u), lane (q, j), register r -> co = 64 wa + 16 t + 4q + r,
    // ci = 32 wb + 16 u + j
    float* out = scratch + (size_t)blockIdx.y * T::SCRATCH_FLOATS_PER_SLICE;
#pragma unroll
    for (int p = 0; p < T::NPOS; p++) {
        const int pos = gi * 6 + p;
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                out[((size_t)pos * T::C + wa * 64 + (t >> 1) * 16 + 4 * q + r) * T::C + wb * 32 + (t & 1) * 16 + j] = acc[p][t][r];
    }
}

// stage 1: dU[pos][co][ci] = sum over slices (in place into slice 0); one thread per element, 16-byte accesses
__global__ __launch_bounds__(256) void wgrad_wino_sum_kernel(float* __restrict__ scratch, int slices) {
    const long i = blockIdx.x * 256L + threadIdx.x;          // float4 index into [36][128][128]
    if (i >= 36L * 128 * 128 / 4) return;
    f32x4 a = reinterpret_cast<const f32x4*>(scratch)[i];
    for (int s = 1; s < slices; s++) a += reinterpret_cast<const f32x4*>(scratch + (size_t)s * 36 * 128 * 128)[i];
    reinterpret_cast<f32x4*>(scratch)[i] = a;
}

// stage 2: dw[co][ci][a][b] = sum_{i,k} G[i][a] G[k][b] dU[6i+k][co][ci]   (one thread per (co, ci))
__global__ __launch_bounds__(256) void wgrad_wino_reduce_kernel(const float* __restrict__ du, float* __restrict__ dw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // co * 128 + ci
    if (idx >= 128 * 128) return;
    const float G[6][3] = {{0.25f, 0.f, 0.f},           {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    float u[36];
#pragma unroll
    for (int p = 0; p < 36; p++) u[p] = du[(size_t)p * (128 * 128) + idx];
    float t[3][6];                                   // t[a][k] = sum_i G[i][a] dU[i][k]
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int k = 0; k < 6; k++) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 6; i++) v += G[i][a] * u[i * 6 + k];
            t[a][k] = v;
        }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b2 = 0; b2 < 3; b2++) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 6; k++) v += t[a][k] * G[k][b2];
            dw[(size_t)idx * 9 + a * 3 + b2] = v;
        }
}

}  // namespace apz
