// Weight gradient of the trunk convolution (128 -> 128, 15x15) through the Winograd F(4x4,3x3) domain.  gfx950.
//
// Forward (trunk15_wino2.h): Y = A^T [ sum_ci U ⊙ V ] A with U = G g G^T, V = B^T d B.  Hence
//   dU[pos][co][ci] = sum over boards and tiles of  dM[pos][co][tile] * V[pos][ci][tile],   dM = A dY A^T (6x6 from 4x4)
//   dg[co][ci]      = G^T dU G                                                              (3x3 from 6x6)
// i.e. 36 independent [128 x K] x [K x 128] products with K = 16 tiles per board: 9 216 fp32 MFMAs per board
// instead of the 32 832 of the direct form (conv3x3_wgrad_kernel).
//
// A workgroup (4 waves) owns THREE of the 36 positions (half a row of the 6x6: 12 groups) for ALL 128 x 128 channel
// pairs (wave = output-channel half x input-channel half: 3 x 4 x 4 accumulator tiles = 192 registers) and a slice
// of the batch; TWO workgroups share a CU (78 KB of LDS and 256 registers per wave each), so that one is in its
// transform phase (LDS, barriers, global latency) while the other feeds the matrix pipe.
// Per board a workgroup streams the 128 input planes and the 128 output-gradient planes through LDS in chunks of
// 8 + 8 planes, double-buffered and filled by LDS-DMA (global_load_lds_dwordx4: no staging registers -- with 192
// accumulators every register spent on staging came back as a scratch spill, and a scratch reload waits for ALL
// outstanding loads), transforms each (channel, tile) to its three positions (waves 0-1: V from 6x6 input patches,
// waves 2-3: dM from 4x4 gradient tiles) into two LDS operand arrays [pos 3][channel group 8][tile 16][16 channels],
// then issues 3 x 4 x 16 MFMAs per wave.  One barrier per chunk.  Partial dU of the slices go to a scratch tensor;
// wgrad_wino_sum_kernel / wgrad_wino_reduce_kernel add them and apply G^T . G.
//
// Measured (MI355X, 512 boards): 219 us for the three kernels (the previous 8-wave / 6-position / register-staged
// version: 292 us; the direct conv3x3_wgrad_kernel: 397 us).  Counters of the main kernel: matrix pipe busy 41 % of
// the time, LDS array 26 % (bank conflicts 38 % of that), waves parked 44 %.  What bounds it now is the VALU
// transform, which fp32 MFMAs do not overlap on this chip and which this decomposition repeats (every position row's
// first stage is computed by two workgroups; the two input-transform waves of a workgroup carry 84 instructions per
// unit against 30 of the gradient-transform waves).  An L2 touch of the chunks three ahead (4-byte LDS-DMA per line)
// was tried and made it slower (235 us): the parked time is barrier skew between those roles, not DMA latency.
#pragma once
#include <hip/hip_runtime.h>

#include "conv3x3_mfma.h"

namespace apz {

struct WgradWino {
    static constexpr int C = 128, CK = 8, NCHUNK = C / CK;
    static constexpr int GPLANE = 240;
    static constexpr int RAW_FLOATS = CK * GPLANE;                      // 1920: the chunk's planes of one tensor, as stored
    static constexpr int RAWBUF_FLOATS = 2 * RAW_FLOATS;                // input planes, then gradient planes
    static constexpr int NPOS = 3;                                      // positions per workgroup: half a row of the 6x6
    static constexpr int OP_FLOATS = NPOS * 8 * 16 * 16;                // 6144 per operand array
    static constexpr int LDS_FLOATS = 2 * RAWBUF_FLOATS + 2 * OP_FLOATS;   // 19968 floats = 78 KiB
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int GROUPS = 12;
    static constexpr int THREADS = 256;
    static constexpr size_t SCRATCH_FLOATS_PER_SLICE = (size_t)36 * C * C;
};
static_assert(2 * WgradWino::LDS_BYTES <= 160 * 1024, "two workgroups per CU");

// rows of B^T (6x6) and of A (6x4): the workgroup's position row is a runtime value, so the first transform stage is
// a plain coefficient dot product (a `switch` on the row made hipcc evaluate every case and select)
__device__ const float WGW_BT[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                       {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
__device__ const float WGW_A[6][4] = {{1, 0, 0, 0}, {1, 1, 1, 1}, {1, -1, 1, -1}, {1, 2, 4, 8}, {1, -2, 4, -8}, {0, 0, 0, 1}};

// One wave-instruction of LDS-DMA: lane l copies 16 bytes from its global address to LDS byte lds_base + 16 l.
// Inline assembly, so that hipcc's wait-count insertion does not see it (it would put vmcnt(0) in front of every LDS
// read that might alias a DMA destination, i.e. wait for the chunk just requested); the kernel counts by hand.
__device__ __forceinline__ void wgw_dma16(const float* gsrc_lane, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc_lane), "s"(lds_base) : "memory");
}

// lane k of every quad takes `v` of lane k-1 (DOWN) / k+1 (UP); the quad's ends get 0
template <bool UP>
__device__ __forceinline__ float wgw_quad_neighbour(float v, int k) {
    const int moved = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), UP ? 0xf9 : 0x90, 0xf, 0xf, false);   // quad_perm [1,2,3,3] / [0,0,1,2]
    return k == (UP ? 3 : 0) ? 0.f : __builtin_bit_cast(float, moved);
}

// GH = which half of the row (columns 3 GH .. 3 GH + 2): compile-time, so that zero coefficients of the second
// transform stage cost nothing
template <int GH>
__device__ __forceinline__ void wgrad_wino_body(const float* __restrict__ x, const float* __restrict__ dy,
                                                float* __restrict__ scratch, int n, int gi, int slice, int slices, float* lds) {
    using T = WgradWino;
    float* raw = lds;                                // [2 buffers][8 input planes | 8 gradient planes][240]
    float* opv = lds + 2 * T::RAWBUF_FLOATS;         // V  [3][8][16 tiles][16]
    float* opm = opv + T::OP_FLOATS;                 // dM [3][8][16 tiles][16]
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, j = lane & 15;
    const int wa = wave >> 1, wb = wave & 1;         // MFMA phase: output-channel half, input-channel half

    f32x4 acc[T::NPOS][4][4];
#pragma unroll
    for (int p = 0; p < T::NPOS; p++)
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int u = 0; u < 4; u++) acc[p][t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transform roles: waves 0-1 -> input units, waves 2-3 -> gradient units; unit = (channel of the chunk, tile);
    // the four tiles of a tile row are the four lanes of a quad
    const int unit = tid & 127, uch = unit >> 4, utile = unit & 15, uty = utile >> 2, utx = utile & 3;
    const bool is_x = wave < 2;
    // first-stage rows with the board's edge folded in: the planes sit in LDS as stored (15 rows of 16, pad column
    // zero), a row outside the board gets coefficient 0 and a clamped address
    float cf[6];
    int ro[6];
#pragma unroll
    for (int a = 0; a < 6; a++) {
        const int r = is_x ? 4 * uty - 1 + a : 4 * uty + a;
        const bool ok = r >= 0 && r < 15 && (is_x || a < 4);
        cf[a] = ok ? (is_x ? WGW_BT[gi][a] : WGW_A[gi][a & 3]) : 0.f;
        ro[a] = (is_x ? 0 : T::RAW_FLOATS) + uch * T::GPLANE + (ok ? r : 0) * 16 + 4 * utx;
    }
    constexpr float cbc[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                 {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};      // B^T, all rows
    constexpr float cac[6][4] = {{1, 0, 0, 0}, {1, 1, 1, 1}, {1, -1, 1, -1}, {1, 2, 4, 8}, {1, -2, 4, -8}, {0, 0, 0, 1}};

    // a chunk = 2 x 7680 contiguous bytes in global memory = 2 x 7.5 DMA instructions of 1 KiB; wave w moves pieces
    // w and w + 4 of either tensor (piece 7 is half a KiB: lanes 0-31)
    auto issue = [&](int bb, int cc, int buf) {
        bb = bb < n ? bb : n - 1;
        const size_t off = ((size_t)bb * T::C + cc * T::CK) * T::GPLANE + wave * 256 + lane * 4;
        const unsigned l0 = lds_base + buf * (T::RAWBUF_FLOATS * 4) + wave * 1024;
        wgw_dma16(x + off, l0);
        wgw_dma16(dy + off, l0 + T::RAW_FLOATS * 4);
        if (wave < 3 || lane < 32) {
            wgw_dma16(x + off + 1024, l0 + 4096);
            wgw_dma16(dy + off + 1024, l0 + T::RAW_FLOATS * 4 + 4096);
        }
    };
    issue(slice, 0, 0);

    for (int b = slice; b < n; b += slices) {
#pragma unroll 2
        for (int c = 0; c < T::NCHUNK; c++) {
            const int buf = c & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk c have landed ...
            __syncthreads();                                    // ... everybody's have, and chunk c-1 (and the MFMA phase) is consumed
            if (c + 1 < T::NCHUNK)
                issue(b, c + 1, buf ^ 1);            // the next chunk of this board ...
            else
                issue(b + slices, 0, buf ^ 1);       // ... or the first chunk of this workgroup's next board
            // ---- transform this chunk's units to the workgroup's three positions.  Operand rows are swizzled: channel
            // ch of tile t sits at slot (ch & 15) ^ 2 (t >> 1) of its 16-channel group, which spreads the 32 lanes of a
            // write (2 channels x 16 tiles; tiles are 16 banks apart) over 32 banks and permutes only within the 16
            // channels that one MFMA operand read takes
            const float* rb = raw + buf * T::RAWBUF_FLOATS;
            const int wpos = ((c >> 1) * 16 + utile) * 16 + ((((c & 1) * 8) + uch) ^ (2 * (utile >> 1)));
            if (is_x) {
                float r[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // row i of B^T d, six columns
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    // columns 0..3 of the 6x6 patch are one 16-byte read; columns -1 and 4 are the neighbouring tiles'
                    // columns 3 and 0, taken from the neighbouring lanes (as LDS reads they were 4-way bank-conflicted)
                    const f32x4 c03 = *reinterpret_cast<const f32x4*>(rb + ro[a]);
                    const float cm1 = wgw_quad_neighbour<false>(c03[3], utx);
                    const float c4 = wgw_quad_neighbour<true>(c03[0], utx);
                    r[0] = __builtin_fmaf(cf[a], cm1, r[0]);
                    r[1] = __builtin_fmaf(cf[a], c03[0], r[1]);
                    r[2] = __builtin_fmaf(cf[a], c03[1], r[2]);
                    r[3] = __builtin_fmaf(cf[a], c03[2], r[3]);
                    r[4] = __builtin_fmaf(cf[a], c03[3], r[4]);
                    r[5] = __builtin_fmaf(cf[a], c4, r[5]);
                }
#pragma unroll
                for (int kk = 0; kk < T::NPOS; kk++) {
                    float o = 0.f;
#pragma unroll
                    for (int k = 0; k < 6; k++)
                        if (cbc[3 * GH + kk][k] != 0.f) o = __builtin_fmaf(cbc[3 * GH + kk][k], r[k], o);
                    opv[kk * 2048 + wpos] = o;
                }
            } else {
                f32x4 r4 = f32x4{0.f, 0.f, 0.f, 0.f};   // row i of A dY, four columns (one 16-byte read per tile row)
#pragma unroll
                for (int a = 0; a < 4; a++) r4 += cf[a] * *reinterpret_cast<const f32x4*>(rb + ro[a]);
                const float r[4] = {r4[0], r4[1], r4[2], r4[3]};
#pragma unroll
                for (int kk = 0; kk < T::NPOS; kk++) {
                    float o = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (cac[3 * GH + kk][k] != 0.f) o = __builtin_fmaf(cac[3 * GH + kk][k], r[k], o);
                    opm[kk * 2048 + wpos] = o;
                }
            }
        }
        __syncthreads();                             // both operand arrays complete for board b
        // ---- dU[pos][co][ci] += dM[pos][co][tile] * V[pos][ci][tile]:  A = dM (m = co), B = V (n = ci), k = tile.
        // wave = (co half wa, ci half wb): four 16-channel groups each way, 4 + 4 operand reads per 16 MFMAs
#pragma unroll
        for (int p = 0; p < T::NPOS; p++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int slot = (4 * s + q) * 16 + (j ^ (4 * s + 2 * (q >> 1)));   // tile 4 s + q, swizzled channel slot
                const float* om = opm + (p * 8 + wa * 4) * 256 + slot;
                const float* ov = opv + (p * 8 + wb * 4) * 256 + slot;
                float bv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) bv[u] = ov[u * 256];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const float a = om[t * 256];
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        acc[p][t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[u], acc[p][t][u], 0, 0, 0);
                }
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last (unused) chunk request: a wave must not end with it in flight
    // ---- partial dU of this slice: accumulator (t, u), lane (q, j), register r -> co = 64 wa + 16 t + 4q + r,
    // ci = 64 wb + 16 u + j
    float* out = scratch + (size_t)slice * T::SCRATCH_FLOATS_PER_SLICE;
#pragma unroll
    for (int p = 0; p < T::NPOS; p++) {
        const int pos = gi * 6 + 3 * GH + p;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    out[((size_t)pos * T::C + wa * 64 + t * 16 + 4 * q + r) * T::C + wb * 64 + u * 16 + j] = acc[p][t][u][r];
    }
}

// x, dy: padded-row layout [n][128][15][16].  scratch: [slices][36][128 co][128 ci].
// Grid: 8 * 12 * spx workgroups (spx = batch slices per XCD, slices = 8 * spx).  The twelve position groups of a slice
// read the SAME boards: workgroup L (dispatched round-robin, L mod 8 = its XCD) takes slice (L mod 8) * spx + (L / 8) / 12
// and group (L / 8) mod 12, so the sharers sit on one XCD, run in step and all but one read of a board hit that
// XCD's L2 -- spread over the XCDs every board crosses the fabric once per group.
__global__ __launch_bounds__(256, 2) void wgrad_wino_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ scratch, int n, int spx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wg_k = blockIdx.x >> 3;
    const int g = wg_k % WgradWino::GROUPS;          // position group: row g / 2 of the 6x6, columns 3 (g % 2) ..
    const int slice = (blockIdx.x & 7) * spx + wg_k / WgradWino::GROUPS, slices = 8 * spx;
    if (g & 1)
        wgrad_wino_body<1>(x, dy, scratch, n, g >> 1, slice, slices, lds);
    else
        wgrad_wino_body<0>(x, dy, scratch, n, g >> 1, slice, slices, lds);
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
