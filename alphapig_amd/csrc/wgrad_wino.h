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
// 16 planes (input and gradient chunks alternate), double-buffered and filled by LDS-DMA (global_load_lds_dwordx4: no staging registers -- with 192
// accumulators every register spent on staging came back as a scratch spill, and a scratch reload waits for ALL
// outstanding loads), transforms each (channel, tile) to its three positions (V from 6x6 input patches in the input
// chunks, dM from 4x4 gradient tiles in the gradient chunks; all four waves alike) into two LDS operand arrays [pos 3][channel group 8][tile 16][16 channels],
// then issues 3 x 4 x 16 MFMAs per wave.  One barrier per chunk.  Partial dU of the slices go to a scratch tensor;
// wgrad_wino_sum_kernel / wgrad_wino_reduce_kernel add them and apply G^T . G.
//
// Measured (MI355X, 512 boards, operands cache-resident): 206 us for the three kernels, 241 us with operands from HBM
// (the previous 8-wave / 6-position / register-staged version: 292 us; the direct conv3x3_wgrad_kernel: 397 us).  Counters of the main kernel: matrix pipe busy 41 % of
// the time, LDS array 26 % (bank conflicts 38 % of that), waves parked 44 %.  What bounds it now is the VALU
// transform, which fp32 MFMAs do not overlap on this chip and which this decomposition repeats (every position row's
// first stage is computed by two workgroups).  An L2 touch of the chunks a few ahead (4-byte LDS-DMA per line,
// APZ_WGW_TOUCH) was tried twice and is slower with cache-resident and with HBM-resident operands (+20 us).
#pragma once
#include <hip/hip_runtime.h>

#include "conv3x3_mfma.h"

#ifndef APZ_WGW_TOUCH
#define APZ_WGW_TOUCH 0
#endif

namespace apz {

struct WgradWino {
    static constexpr int C = 128, CK = 16, NCHUNK = 2 * C / CK;       // 16 chunks per board: 8 input, 8 gradient
    static constexpr int GPLANE = 240;
    static constexpr int RAWBUF_FLOATS = CK * GPLANE;                   // 3840: the chunk's 16 planes, as stored
    static constexpr int NPOS = 3;                                      // positions per workgroup: half a row of the 6x6
    static constexpr int OP_FLOATS = NPOS * 8 * 16 * 16;                // 6144 per operand array
    static constexpr int JUNK_FLOATS = 256;                             // landing zone of the L2 touches (never read)
    static constexpr int LDS_FLOATS = 2 * RAWBUF_FLOATS + 2 * OP_FLOATS + JUNK_FLOATS;   // 20224 floats = 79 KiB
    static constexpr int TOUCH_AHEAD = 4;                               // chunks between a touch and its use
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
// ... and of 4 bytes per lane: pulls the 128-byte lines of a later chunk into this XCD's L2 (one lane per line); the
// bytes land in a junk area of LDS, so no register waits for them
__device__ __forceinline__ void wgw_touch(const float* gsrc_lane, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc_lane), "s"(lds_base) : "memory");
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
    float* raw = lds;                                // [2 buffers][16 planes][240]
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

    // transform roles: the chunks alternate between 16 input planes and 16 gradient planes, and in every chunk all
    // four waves do the same thing (256 units = (channel of the chunk, tile); the four tiles of a tile row are the
    // four lanes of a quad).  With input and gradient units on different waves of one chunk the gradient waves (30
    // instructions per unit against 84) spent most of every chunk at the barrier.
    const int uch = tid >> 4, utile = tid & 15, uty = utile >> 2, utx = utile & 3;
    // First-stage rows with the board's edge folded in: the planes sit in LDS as stored (15 rows of 16, pad column
    // zero); a row outside the board gets coefficient 0 and the address of row 14 / row 0.  Only rows 0, 4, 5 of an
    // input patch and row 3 of a gradient tile can fall outside: the other coefficients are wave-uniform.
    const int xbase = uch * T::GPLANE + (4 * uty - 1) * 16 + 4 * utx, ybase = uch * T::GPLANE + 4 * uty * 16 + 4 * utx;
    const int xo0 = xbase + (uty == 0 ? 16 : 0), xo4 = xbase + 64 - (uty == 3 ? 16 : 0), xo5 = xbase + 80 - (uty == 3 ? 32 : 0);
    const int yo3 = ybase + 48 - (uty == 3 ? 16 : 0);
    const float cx0 = uty == 0 ? 0.f : WGW_BT[gi][0], cx4 = uty == 3 ? 0.f : WGW_BT[gi][4], cx5 = uty == 3 ? 0.f : WGW_BT[gi][5];
    const float cx1 = WGW_BT[gi][1], cx2 = WGW_BT[gi][2], cx3 = WGW_BT[gi][3];
    const float cy0 = WGW_A[gi][0], cy1 = WGW_A[gi][1], cy2 = WGW_A[gi][2], cy3 = uty == 3 ? 0.f : WGW_A[gi][3];
    constexpr float cbc[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                 {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};      // B^T, all rows
    constexpr float cac[6][4] = {{1, 0, 0, 0}, {1, 1, 1, 1}, {1, -1, 1, -1}, {1, 2, 4, 8}, {1, -2, 4, -8}, {0, 0, 0, 1}};

    // chunk cc of a board = 16 planes of ONE tensor (even: input channels 8 cc .., odd: gradient channels 8 (cc - 1) ..):
    // 15 360 contiguous bytes = 15 DMA instructions of 1 KiB; wave w moves pieces w, w + 4, w + 8 and (w < 3) w + 12
    auto issue = [&](int bb, int cc, int buf) {
        bb = bb < n ? bb : n - 1;
        const float* src = ((cc & 1) ? dy : x) + ((size_t)bb * T::C + (cc >> 1) * 16) * T::GPLANE + wave * 256 + lane * 4;
        const unsigned l0 = lds_base + buf * (T::RAWBUF_FLOATS * 4) + wave * 1024;
        wgw_dma16(src, l0);
        wgw_dma16(src + 1024, l0 + 4096);
        wgw_dma16(src + 2048, l0 + 8192);
        if (wave < 3) wgw_dma16(src + 3072, l0 + 12288);
    };
    // The DMA of chunk g+1 is requested one chunk ahead (two LDS buffers), which covers an L2 hit but not an HBM miss,
    // and in a training step every chunk is a first touch for the XCD (its twelve sharers ask at the same moment).  So
    // the 120 lines of chunk g + TOUCH_AHEAD are touched ahead of time: one 4-byte LDS-DMA per line, 30 lines per wave.
    // Loads retire in order: the `vmcnt(1)` at the top of a chunk leaves the youngest request -- the touch -- in flight.
    auto touch = [&](int bb, int cc) {
        bb += (cc >> 4) * slices;
        cc &= 15;
        bb = bb < n ? bb : n - 1;
        const float* src = ((cc & 1) ? dy : x) + ((size_t)bb * T::C + (cc >> 1) * 16) * T::GPLANE + (wave * 30 + lane) * 32;
        if (lane < 30 && APZ_WGW_TOUCH) wgw_touch(src, lds_base + (2 * T::RAWBUF_FLOATS + 2 * T::OP_FLOATS) * 4);
    };
    issue(slice, 0, 0);
#pragma unroll
    for (int d = 1; d < T::TOUCH_AHEAD; d++) touch(slice, d);

    for (int b = slice; b < n; b += slices) {
#pragma unroll 2
        for (int c = 0; c < T::NCHUNK; c++) {
            const int buf = c & 1;
            if (APZ_WGW_TOUCH)
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   // this wave's pieces of chunk c have landed ...
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                    // ... everybody's have, and chunk c-1 (and the MFMA phase) is consumed
            if (c + 1 < T::NCHUNK)
                issue(b, c + 1, buf ^ 1);            // the next chunk of this board ...
            else
                issue(b + slices, 0, buf ^ 1);       // ... or the first chunk of this workgroup's next board
            touch(b, c + T::TOUCH_AHEAD);
            // ---- transform this chunk's units to the workgroup's three positions.  Operand rows are swizzled: channel
            // ch of tile t sits at slot (ch & 15) ^ 2 (t >> 1) of its 16-channel group, which spreads the 32 lanes of a
            // write (2 channels x 16 tiles; tiles are 16 banks apart) over 32 banks and permutes only within the 16
            // channels that one MFMA operand read takes
            const float* rb = raw + buf * T::RAWBUF_FLOATS;
            const int wpos = ((c >> 1) * 16 + utile) * 16 + (uch ^ (2 * (utile >> 1)));
            if ((c & 1) == 0) {
                float r[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // row i of B^T d, six columns
                // columns 0..3 of the 6x6 patch are one 16-byte read; columns -1 and 4 are the neighbouring tiles'
                // columns 3 and 0, taken from the neighbouring lanes (as LDS reads they were 4-way bank-conflicted)
                auto row = [&](const float cf, const int off) {
                    const f32x4 c03 = *reinterpret_cast<const f32x4*>(rb + off);
                    const float cm1 = wgw_quad_neighbour<false>(c03[3], utx);
                    const float c4 = wgw_quad_neighbour<true>(c03[0], utx);
                    r[0] = __builtin_fmaf(cf, cm1, r[0]);
                    r[1] = __builtin_fmaf(cf, c03[0], r[1]);
                    r[2] = __builtin_fmaf(cf, c03[1], r[2]);
                    r[3] = __builtin_fmaf(cf, c03[2], r[3]);
                    r[4] = __builtin_fmaf(cf, c03[3], r[4]);
                    r[5] = __builtin_fmaf(cf, c4, r[5]);
                };
                row(cx0, xo0);
                row(cx1, xbase + 16);
                row(cx2, xbase + 32);
                row(cx3, xbase + 48);
                row(cx4, xo4);
                row(cx5, xo5);
#pragma unroll
                for (int kk = 0; kk < T::NPOS; kk++) {
                    float o = 0.f;
#pragma unroll
                    for (int k = 0; k < 6; k++)
                        if (cbc[3 * GH + kk][k] != 0.f) o = __builtin_fmaf(cbc[3 * GH + kk][k], r[k], o);
                    opv[kk * 2048 + wpos] = o;
                }
            } else {
                // row i of A dY, four columns (one 16-byte read per tile row)
                f32x4 r4 = cy0 * *reinterpret_cast<const f32x4*>(rb + ybase);
                r4 += cy1 * *reinterpret_cast<const f32x4*>(rb + ybase + 16);
                r4 += cy2 * *reinterpret_cast<const f32x4*>(rb + ybase + 32);
                r4 += cy3 * *reinterpret_cast<const f32x4*>(rb + yo3);
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
