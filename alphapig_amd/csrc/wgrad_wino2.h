// Weight gradient of the trunk convolution (128 -> 128, 15x15) through the Winograd F(4x4,3x3) domain, decomposed by
// CHANNEL BLOCKS.  gfx950.
//
// Forward (trunk15_wino3.h): Y = A^T [ sum_ci U (.) V ] A with U = G g G^T, V = B^T d B.  Hence
//   dU[pos][co][ci] = sum over boards and tiles of dM[pos][co][tile] * V[pos][ci][tile],  dM = A dY A^T (6x6 from 4x4),
//   dg[co][ci]      = G^T dU G                                                             (3x3 from 6x6)
// i.e. 36 independent [128 x K] x [K x 128] products with K = 16 tiles per board: 9 216 fp32 MFMAs per board instead of the
// 32 832 of the direct form (conv3x3_wgrad_kernel).  Every workgroup applies G^T . G to its own partial dU (epilogue) and
// writes a partial dg of its batch slice to a scratch tensor; wgrad_wino_finish_kernel (below) adds the slices.
//
// Round 2's first version (by position groups, in the git history) gave a workgroup three of the 36 positions and all 128 x 128 channel pairs: every board's 256
// planes are read by twelve workgroups (2.9 MB of L2 -> CU traffic per board, 1.5 GB per 512-board launch), and each of
// them repeats the first transform stage of every (plane, tile) for its own position row (3 220 VALU instructions per
// workgroup and board against 768 MFMAs).  Here a workgroup owns a block of 64 output x 32 input channels with ALL 36
// positions (8 waves x 144 accumulator registers, as in the forward kernel): a board's planes are read by two (input)
// or four (gradient) workgroups, and every (plane, tile) is transformed once per reader, completely -- 66 / 80 VALU
// lane-operations for all 36 positions instead of ~40 for three.
//
// One workgroup (512 threads) per CU.  The operands of a whole board for 36 positions do not fit LDS (221 KB), so a
// board is processed in two halves of two tile rows (8 tiles = two MFMA k-steps): per half six chunks of 16 planes
// (two input chunks, four gradient chunks) stream through a raw area of three buffers by LDS-DMA (two chunks in
// flight), six waves transform a chunk (wave = tile row x one of three pairs of transform rows; lane = channel x tile
// of the row, the four tiles of a row in one quad so that the patch's halo columns come from neighbouring lanes), two
// waves are the loaders; then all eight waves run the half's MFMAs: wave = (nine positions) x (two of the four
// output-channel groups) x both input-channel groups, four operand reads per four MFMAs, requested one step ahead.
// Each half requests only the rows of a plane its tiles look at (9 + 8 of 15 input rows, 8 + 7 gradient rows).
//
// LDS: raw [5][16 planes][148] (the nine rows of a plane a half looks at + 16 bytes: sixteen lanes that read the same
// tile of sixteen planes would otherwise sit on two banks; round 3 staged whole-plane slots, three buffers = two chunks
// = 17 KB in flight per CU, and the launch's skeleton -- DMA stream + barriers, no transform, no MFMA -- took 82 of its
// 192 us at 512 boards: latency-bound at 19 GB/s per CU), V [36][2 groups][8 tiles][16], dM [36][4 groups][8 tiles][16]; channel c of tile t
// sits in slot c ^ 8 (t >> 1 & 1) of its group row, which makes the transform's writes and the MFMA's reads both
// conflict-free.  157.4 KB.
#pragma once
#include <hip/hip_runtime.h>

#include "wino_common.h"

#ifndef APZ_WGW2_NO_TRANSFORM
#define APZ_WGW2_NO_TRANSFORM 0   /* measurement builds: skip the transforms / the MFMA phase */
#endif
#ifndef APZ_WGW2_NO_MFMA
#define APZ_WGW2_NO_MFMA 0
#endif

namespace apz {

struct WgradWino2 {
    static constexpr int C = 128, CO_B = 64, CI_B = 32, BLOCKS = (C / CO_B) * (C / CI_B);   // 8 channel blocks
    static constexpr int GPLANE = 240, RSTRIDE = 148;                   // plane as stored / the <= 9 rows a half looks at as staged in LDS (floats)
    static constexpr int CK = 16, CHUNKS = (CI_B + CO_B) / CK;          // 6 chunks per half board: 2 input, 4 gradient
    static constexpr int RAW_FLOATS = CK * RSTRIDE;                     // 3904 per buffer
    static constexpr int OPV_FLOATS = 36 * (CI_B / 16) * 128;           // 9216
    static constexpr int OPM_FLOATS = 36 * (CO_B / 16) * 128;           // 18432
    static constexpr int NBUF = 5;                                      // raw buffers: chunks u + 1 .. u + 4 are in flight while u is transformed
    static constexpr int LDS_FLOATS = NBUF * RAW_FLOATS + OPV_FLOATS + OPM_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;                    // 157 440
    static constexpr int THREADS = 512;
};
static_assert(WgradWino2::LDS_BYTES <= 160 * 1024, "LDS");

// partial weight gradient of one batch slice: [128 co][128 ci][3][3]
struct WgradWino {
    static constexpr size_t SCRATCH_FLOATS_PER_SLICE = (size_t)128 * 128 * 9;
};

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


// x, dy: padded-row layout [n][128][15][16].  scratch: [slices][128 co][128 ci][3][3] (partial dg per batch slice).
// Grid: 8 * BLOCKS * spx workgroups, slices = 8 * spx.  Workgroup L (dispatched round-robin over the XCDs, L mod 8 = its
// XCD) takes slice (L mod 8) * spx + (L / 8) / BLOCKS and channel block (L / 8) mod BLOCKS: the eight blocks of a slice
// read the same boards and sit on one XCD, so all but the first read of a plane is an L2 hit.
__global__ __launch_bounds__(512) void wgrad_wino2_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ scratch, int n, int spx) {
    using T = WgradWino2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* raw = lds;                                 // [NBUF][16][RSTRIDE]
    float* opv = lds + T::NBUF * T::RAW_FLOATS;       // V  [36][2][8 tiles][16]
    float* opm = opv + T::OPV_FLOATS;                 // dM [36][4][8 tiles][16]
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;

    const int wg_k = blockIdx.x >> 3;
    const int blk = wg_k % T::BLOCKS, cob = blk >> 2, cib = blk & 3;
    const int slice = (blockIdx.x & 7) * spx + wg_k / T::BLOCKS, slices = 8 * spx;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, j = lane & 15;

    // ---- MFMA roles: wave = (position group pg: positions 9 pg .. 9 pg + 8) x (output-channel groups 2 cc, 2 cc + 1)
    const int pg = wave & 3, cc = wave >> 2;
    f32x4 acc[9][2][2];
#pragma unroll
    for (int p = 0; p < 9; p++)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[p][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- transform roles (waves 0..5): tile row tr of the half, row pair `role` of the first transform stage;
    // lane = (channel of the chunk uch, tile column ttx): the four tiles of a tile row are the four lanes of a quad.
    // (Eight transform waves -- four row sets per unit: {0}, {5}, {1, 2}, {3, 4} / {0, 1}, {2, 5}, {3}, {4} -- were measured
    // SLOWER, with the loads spread over all waves or kept on two: 238 / 250 us against 202 us per 512-board launch.)
    const int tr = wave & 1, role = wave >> 1;
    const int ttx = lane & 3, uch = lane >> 2;
    const int wslot = (tr * 4 + ttx) * 16 + (uch ^ (8 * (ttx >> 1)));   // (tile of the half, swizzled channel slot)
    // ---- loader roles (waves 6, 7): plane 8 (wave - 6) + i of the chunk, i = 0..7; lanes 0..59 move its 960 bytes
    const int lw = wave - 6;

    // stream of this workgroup: unit u = (board of the slice, half, chunk); chunk c < 2: input planes
    // 32 cib + 16 c .., else gradient planes 64 cob + 16 (c - 2) ..
    const int nboards = slice < n ? (n - slice + slices - 1) / slices : 0;
    const int total = nboards * 2 * T::CHUNKS;
    auto issue = [&](int u, int buf) {                // (loader waves) request chunk u into raw buffer `buf`
        if (u >= total) return;
        const int bi = u / (2 * T::CHUNKS), c = u % T::CHUNKS;
        const int b = slice + bi * slices;
        const float* src = (c < 2 ? x + ((size_t)b * T::C + cib * T::CI_B + c * 16) * T::GPLANE
                                  : dy + ((size_t)b * T::C + cob * T::CO_B + (c - 2) * 16) * T::GPLANE);
        // only the rows this half's tiles look at (input: patch rows -1 .. 8 / 7 .. 16 of the board, gradient: rows
        // 0 .. 7 / 8 .. 14): 1 088 / 960 bytes per plane for both halves together instead of 2 x 960
        const int hh = (u / T::CHUNKS) & 1;
        const int r0 = c < 2 ? (hh ? 7 : 0) : (hh ? 8 : 0), r1 = c < 2 ? (hh ? 15 : 9) : (hh ? 15 : 8);   // rows [r0, r1)
        if (lane < 4 * (r1 - r0)) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int pl = lw * 8 + i;
                wgw_dma16(src + pl * T::GPLANE + r0 * 16 + lane * 4, lds_base + (buf * T::RAW_FLOATS + pl * T::RSTRIDE) * 4);   // row r0 first
            }
        }
    };
    if (wave >= 6)
        for (int u = 0; u < T::NBUF - 1; u++) issue(u, u);

    for (int u = 0; u < total; u++) {
        const int buf = u % T::NBUF;
        const int c = u % T::CHUNKS, hh = (u / T::CHUNKS) & 1;
        // this loader's eight planes of chunk u have landed: loads retire in order, the youngest 24 (chunks u + 1 .. u + 3)
        // may still be in flight (past the end of the stream nothing was issued)
        if (wave >= 6) {
            const int younger = min(T::NBUF - 2, total - 1 - u);
            if (younger >= 3)
                asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (younger == 2)
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (younger == 1)
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                              // ... everybody's; chunk u - 1 (and a finished half's MFMAs) consumed
        if (wave >= 6) {
            issue(u + T::NBUF - 1, (u + T::NBUF - 1) % T::NBUF);   // into the buffer chunk u - 1 has just left
        } else if (!APZ_WGW2_NO_TRANSFORM) {
            const float* rb = raw + buf * T::RAW_FLOATS + uch * T::RSTRIDE;
            const int trow = 2 * hh + tr;             // tile row of the board (wave-uniform)
            if (c < 2) {
                // ---- V = B^T d B of (input channel 16 c + uch, tile (trow, ttx)): rows y[2] of B^T d for this role
                float y[2][6];
                auto row6 = [&](int i, float* v) {    // patch row i (board row 4 trow - 1 + i), columns -1 .. 4
                    const int R = 4 * trow - 1 + i;
                    const bool in = R >= 0 && R <= 14;                  // (wave-uniform)
                    const f32x4 c03 = in ? *reinterpret_cast<const f32x4*>(rb + (in ? R - (hh ? 7 : 0) : 0) * 16 + 4 * ttx) : f32x4{0.f, 0.f, 0.f, 0.f};
                    v[0] = wgw_quad_neighbour<false>(c03[3], ttx);
                    v[1] = c03[0];
                    v[2] = c03[1];
                    v[3] = c03[2];
                    v[4] = c03[3];
                    v[5] = wgw_quad_neighbour<true>(c03[0], ttx);
                };
                int i0, i1;                           // the two output rows of this role
                if (role == 0) {
                    float x0[6], x2[6], x4[6], x1[6], x3[6], x5[6];
                    row6(0, x0); row6(2, x2); row6(4, x4); row6(1, x1); row6(3, x3); row6(5, x5);
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        y[0][k] = __builtin_fmaf(4.f, x0[k], __builtin_fmaf(-5.f, x2[k], x4[k]));
                        y[1][k] = __builtin_fmaf(4.f, x1[k], __builtin_fmaf(-5.f, x3[k], x5[k]));
                    }
                    i0 = 0; i1 = 5;
                } else {
                    float x1[6], x2[6], x3[6], x4[6];
                    row6(1, x1); row6(2, x2); row6(3, x3); row6(4, x4);
                    if (role == 1) {
#pragma unroll
                        for (int k = 0; k < 6; k++) {
                            const float a = __builtin_fmaf(-4.f, x2[k], x4[k]), b = __builtin_fmaf(-4.f, x1[k], x3[k]);
                            y[0][k] = a + b;
                            y[1][k] = a - b;
                        }
                        i0 = 1; i1 = 2;
                    } else {
#pragma unroll
                        for (int k = 0; k < 6; k++) {
                            const float cdiff = x4[k] - x2[k], d = x3[k] - x1[k];
                            y[0][k] = __builtin_fmaf(2.f, d, cdiff);
                            y[1][k] = __builtin_fmaf(-2.f, d, cdiff);
                        }
                        i0 = 3; i1 = 4;
                    }
                }
                float* dst = opv + c * 128 + wslot;   // group c of the block's two input-channel groups
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const float* v = y[rr];
                    const float a = __builtin_fmaf(-4.f, v[2], v[4]), b = __builtin_fmaf(-4.f, v[1], v[3]);
                    const float cdiff = v[4] - v[2], d = v[3] - v[1];
                    float o[6];
                    o[0] = __builtin_fmaf(4.f, v[0], __builtin_fmaf(-5.f, v[2], v[4]));
                    o[1] = a + b;
                    o[2] = a - b;
                    o[3] = __builtin_fmaf(2.f, d, cdiff);
                    o[4] = __builtin_fmaf(-2.f, d, cdiff);
                    o[5] = __builtin_fmaf(4.f, v[1], __builtin_fmaf(-5.f, v[3], v[5]));
                    const int ir = rr == 0 ? i0 : i1;
#pragma unroll
                    for (int k = 0; k < 6; k++) dst[(ir * 6 + k) * 256] = o[k];      // V: 2 groups x 128 floats per position
                }
            } else {
                // ---- dM = A dY A^T of (output channel 16 (c - 2) + uch, tile (trow, ttx)): rows m[2] of A dY for this role
                auto row4 = [&](int i) {              // tile row i (board row 4 trow + i), columns 0 .. 3
                    const int R = 4 * trow + i;
                    const bool in = R <= 14;
                    return in ? *reinterpret_cast<const f32x4*>(rb + (in ? R - (hh ? 8 : 0) : 0) * 16 + 4 * ttx) : f32x4{0.f, 0.f, 0.f, 0.f};
                };
                f32x4 m0, m1;
                int i0, i1;
                if (role == 0) {
                    m0 = row4(0);
                    m1 = row4(3);
                    i0 = 0; i1 = 5;
                } else {
                    const f32x4 d0 = row4(0), d1 = row4(1), d2 = row4(2), d3 = row4(3);
                    if (role == 1) {
                        const f32x4 s02 = d0 + d2, s13 = d1 + d3;
                        m0 = s02 + s13;
                        m1 = s02 - s13;
                        i0 = 1; i1 = 2;
                    } else {
                        const f32x4 sv = d0 + 4.f * d2, tv = 2.f * d1 + 8.f * d3;
                        m0 = sv + tv;
                        m1 = sv - tv;
                        i0 = 3; i1 = 4;
                    }
                }
                float* dst = opm + (c - 2) * 128 + wslot;   // group c - 2 of the block's four output-channel groups
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const f32x4 w = rr == 0 ? m0 : m1;
                    const float s02 = w[0] + w[2], s13 = w[1] + w[3];
                    const float sv = __builtin_fmaf(4.f, w[2], w[0]), tv = __builtin_fmaf(8.f, w[3], 2.f * w[1]);
                    float o[6];
                    o[0] = w[0];
                    o[1] = s02 + s13;
                    o[2] = s02 - s13;
                    o[3] = sv + tv;
                    o[4] = sv - tv;
                    o[5] = w[3];
                    const int ir = rr == 0 ? i0 : i1;
#pragma unroll
                    for (int k = 0; k < 6; k++) dst[(ir * 6 + k) * 512] = o[k];      // dM: 4 groups x 128 floats per position
                }
            }
        }
        if (c == T::CHUNKS - 1 && !APZ_WGW2_NO_MFMA) {
            __syncthreads();                          // the half's operand arrays are complete
            // ---- dU[pos][co][ci] += dM[pos][co][tile] * V[pos][ci][tile] over the half's 8 tiles (two k-steps):
            // A = dM (m = co), B = V (n = ci), k = tile; lane (q, j): tile 4 s + q, channel slot j ^ 8 (q >> 1)
            // Operands are requested one step (position, k-step) ahead of their four MFMAs and the order is pinned:
            // left alone hipcc puts every step's four LDS reads right in front of its MFMAs and waits out the LDS
            // latency eighteen times per half (the phase then runs at 45 % of the matrix pipe).
            auto fetch = [&](int st, float* o) {      // step st = 2 p + s
                const int pos = pg * 9 + (st >> 1), sk = st & 1;
                const int slot = (4 * sk + q) * 16 + (j ^ (8 * (q >> 1)));
                o[0] = opm[(pos * 4 + 2 * cc) * 128 + slot];
                o[1] = opm[(pos * 4 + 2 * cc + 1) * 128 + slot];
                o[2] = opv[(pos * 2) * 128 + slot];
                o[3] = opv[(pos * 2 + 1) * 128 + slot];
            };
            float cur[4], nxt[4];
            fetch(0, cur);
#pragma unroll
            for (int st = 0; st < 18; st++) {
                if (st + 1 < 18) fetch(st + 1, nxt);
                const int p = st >> 1;
                acc[p][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[0], cur[2], acc[p][0][0], 0, 0, 0);
                acc[p][0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[0], cur[3], acc[p][0][1], 0, 0, 0);
                acc[p][1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[1], cur[2], acc[p][1][0], 0, 0, 0);
                acc[p][1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[1], cur[3], acc[p][1][1], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) cur[i] = nxt[i];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- partial dg of this slice and channel block.  accumulator (p, a, b), lane (q, j), register r holds dU at
    // pos = 9 pg + p, co = 64 cob + 16 (2 cc + a) + 4 q + r, ci = 32 cib + 16 b + j.  The 36 positions of a (co, ci) pair
    // sit in four waves: they meet in LDS (free now), half of the block's output channels per pass
    // ([36][32 co][32 ci + 1 pad] = 152 KB), and every thread turns two pairs' 6x6 into the 3x3 weight gradient
    // G^T dU G -- the slices' partials are 9 instead of 36 floats per pair: 18.9 MB written and read back per launch
    // instead of 75.5 MB (round 3: 64-byte pieces written at the very end of every workgroup at once, then a 12 us
    // sum; together the "39 us fixed per launch" of profiles/r03_train_bench.md).
    constexpr int CS = 33, PST = 32 * CS;
    static_assert(36 * PST <= T::LDS_FLOATS, "epilogue staging fits the operand memory");
    float* du = lds;
    float* outw = scratch + (size_t)slice * WgradWino::SCRATCH_FLOATS_PER_SLICE;
    const float G[6][3] = {{0.25f, 0.f, 0.f},           {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
#pragma unroll
    for (int a = 0; a < 2; a++) {
        __syncthreads();                              // the last half's operand reads / the previous pass's readers are done
#pragma unroll
        for (int p = 0; p < 9; p++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) du[(pg * 9 + p) * PST + (cc * 16 + 4 * q + r) * CS + b * 16 + j] = acc[p][a][b][r];
        __syncthreads();
#pragma unroll
        for (int e0 = 0; e0 < 1024; e0 += T::THREADS) {
            const int e = e0 + tid, col = e >> 5, cil = e & 31;
            float u[36];
#pragma unroll
            for (int p = 0; p < 36; p++) u[p] = du[p * PST + col * CS + cil];
            float tt[3][6];                          // tt[x][k] = sum_i G[i][x] dU[i][k]
#pragma unroll
            for (int x3 = 0; x3 < 3; x3++)
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    float v = 0.f;
#pragma unroll
                    for (int i6 = 0; i6 < 6; i6++) v += G[i6][x3] * u[i6 * 6 + k];
                    tt[x3][k] = v;
                }
            const int co = cob * T::CO_B + (2 * (col >> 4) + a) * 16 + (col & 15), ci = cib * T::CI_B + cil;
            float* d = outw + ((size_t)co * T::C + ci) * 9;
#pragma unroll
            for (int x3 = 0; x3 < 3; x3++)
#pragma unroll
                for (int y3 = 0; y3 < 3; y3++) {
                    float v = 0.f;
#pragma unroll
                    for (int k = 0; k < 6; k++) v += tt[x3][k] * G[k][y3];
                    d[x3 * 3 + y3] = v;
                }
        }
    }
}

// dw[co][ci][3][3] = sum over the slices' partials, in slice order (16-byte accesses; 0.59 MB per slice)
__global__ __launch_bounds__(256) void wgrad_wino_finish_kernel(const float* __restrict__ scratch, int slices,
                                                                float* __restrict__ dw) {
    constexpr size_t N4 = WgradWino::SCRATCH_FLOATS_PER_SLICE / 4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N4) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(scratch) + i;
    f32x4 a = src[0];
    int sl = 1;
    for (; sl + 7 < slices; sl += 8) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = src[(size_t)(sl + k) * N4];
#pragma unroll
        for (int k = 0; k < 8; k++) a += v[k];
    }
    for (; sl < slices; sl++) a += src[(size_t)sl * N4];
    reinterpret_cast<f32x4*>(dw)[i] = a;
}

}  // namespace apz
