// Weight gradient of the trunk convolution (128 -> 128, 15x15) through the Winograd F(4x4,3x3) domain.  gfx950.
//
// Forward (trunk15_wino3.h): Y = A^T [ sum_ci U (.) V ] A with U = G g G^T, V = B^T d B.  Hence
//   dU[pos][co][ci] = sum over boards and tiles of dM[pos][co][tile] * V[pos][ci][tile],  dM = A dY A^T (6x6 from 4x4),
//   dg[co][ci]      = G^T dU G                                                             (3x3 from 6x6)
// i.e. 36 independent [128 x K] x [K x 128] products with K = 16 tiles per board: 9 216 fp32 MFMAs per board instead of the
// 32 832 of the direct form (conv3x3_wgrad_kernel).  Every workgroup applies G^T . G to its own partial dU (epilogue) and
// writes a partial dg of its batch slice to a scratch tensor; wgrad_wino_finish_kernel adds the slices.
//
// History (profiles/r04_wgrad_wino3.md).  Round 2: a workgroup = three positions x all channels (every plane read twelve
// times).  Rounds 2 - 4, `wgrad_wino2_kernel` (in the git history): 64 x 32 channel blocks, 144 accumulator registers per
// wave, which left a wave room for a THIRD of a (plane, tile) transform at a time (two of the six rows: three "roles"), so
// every patch was read and its first stage recomputed three times; planes staged in LDS by LDS-DMA (16 bytes per clock and
// CU: 64 cycles per plane request); six of eight waves transformed one 16-plane chunk per barrier (14 barriers per board)
// and the phases added up: 187 us per 512 boards, matrix pipe busy 0.35.
//
// Here a workgroup (512 threads, one per CU) owns 32 output x 32 input channels with all 36 positions: 72 accumulator
// registers per wave (wave = nine positions x one of the two output-channel groups x both input-channel groups).  A half
// board (two tile rows = 8 tiles = two MFMA k-steps) is 64 planes x 8 tiles = 512 (plane, tile) pairs: ONE PAIR PER
// THREAD, transformed completely (all 36 positions) in registers from patch rows that come straight from global memory
// (requested a half board ahead; no staging buffers, no LDS-DMA).  Waves 0..3 transform the block's 32 input planes
// (V = B^T d B), waves 4..7 its 32 gradient planes (dM = A dY A^T).  The operand arrays are double-buffered, so there is
// ONE barrier per half board, and a wave that is done with a half's MFMAs goes straight on to the next half's transform
// while the SIMD's other wave still feeds the matrix pipe.  Input planes are read by four workgroups instead of two (all
// of a slice's sixteen blocks sit on one XCD: L2 hits).  134 - 145 us per 512 boards (0.42 - 0.46 of the fp32 matrix peak), 38 - 40 us at 128.
// What bounds it now: the fp32 matrix instructions run on the vector unit's FMA lanes, so a half board costs its 2 304
// cycles of MFMAs PLUS ~1 800 cycles of transform instructions (373 per SIMD and half at ~4.8 cycles each), + ~700 of
// barrier and loop ends; packed two-wide transform arithmetic was measured twice (rows-then-columns with repacking moves; columns-first
// with none: 72 + 36 instead of 144 instructions) and changed nothing in same-box A/B runs (133.3 against 132.7 us); raw buffer loads with hardware zero fill instead of 37 selects: 133.6 against 144.4 us at 512 boards
// (0.46 of the peak), slightly slower at 128 -- chosen by batch size.
//
// LDS: two operand sets of V [36][2 groups][8 tiles][16] + dM [36][2 groups][8 tiles][16]; channel c of tile t sits in slot
// (c + 4 (t >> 1)) & 15 of its (position, group, tile) row: the transform's stores (lane = plane x tile) and the MFMA's
// loads (lane = tile x channel) are both conflict-free.  147.5 KB; the epilogue (G^T dU G through LDS) takes 152 KB.
#pragma once
#include <hip/hip_runtime.h>

#include "wino_common.h"

#ifndef APZ_WGW3_NO_TRANSFORM
#define APZ_WGW3_NO_TRANSFORM 0   /* measurement builds: skip the transforms / the MFMA phase */
#endif
#ifndef APZ_WGW3_NO_MFMA
#define APZ_WGW3_NO_MFMA 0
#endif

namespace apz {

// partial weight gradient of one batch slice: [128 co][128 ci][3][3]
struct WgradWino {
    static constexpr size_t SCRATCH_FLOATS_PER_SLICE = (size_t)128 * 128 * 9;
};

// lane k of every quad takes `v` of lane k-1 (DOWN) / k+1 (UP); the quad's ends get 0
template <bool UP>
__device__ __forceinline__ float wgw_quad_neighbour(float v, int k) {
    const int moved = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), UP ? 0xf9 : 0x90, 0xf, 0xf, false);   // quad_perm [1,2,3,3] / [0,0,1,2]
    return k == (UP ? 3 : 0) ? 0.f : __builtin_bit_cast(float, moved);
}

#ifdef APZ_WGW3_STAMPS
// measurement builds: cycles per wave of workgroup 0 in (0) loop overhead, (2) transform, (3) barrier, (4) MFMA phase,
// (5) epilogue
__device__ unsigned long long apz_wgw3_stamps[8][6];
#define WGW3_STAMP(k_)                                                  \
    {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();   \
        st_acc[k_] += now_ - st_t;                                      \
        st_t = now_;                                                    \
    }
#else
#define WGW3_STAMP(k_)
#endif

struct WgradWino3 {
    static constexpr int C = 128, CO_B = 32, CI_B = 32, BLOCKS = (C / CO_B) * (C / CI_B);   // 16 channel blocks
    static constexpr int GPLANE = 240;
    static constexpr int OP_FLOATS = 36 * 2 * 128;                      // V or dM of a half board
    static constexpr int SET_FLOATS = 2 * OP_FLOATS;                    // one operand set (V, dM); two sets
    static constexpr int EPI_CS = 33, EPI_PST = 32 * EPI_CS;            // epilogue staging [36][32 co][32 ci + 1]
    static constexpr int MAIN_FLOATS = 2 * SET_FLOATS, EPI_FLOATS = 36 * EPI_PST;
    static constexpr int LDS_FLOATS = MAIN_FLOATS > EPI_FLOATS ? MAIN_FLOATS : EPI_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;                    // 152 064
    static constexpr int THREADS = 512;
};
static_assert(WgradWino3::LDS_BYTES <= 160 * 1024, "LDS");

// x, dy: padded-row layout [n][128][15][16].  scratch: [slices][128 co][128 ci][3][3] (partial dg per batch slice; added by
// wgrad_wino_finish_kernel).  Grid: 8 * BLOCKS * spx workgroups, slices = 8 * spx.  Workgroup L (dispatched round-robin
// over the XCDs, L mod 8 = its XCD) takes slice (L mod 8) * spx + (L / 8) / BLOCKS and channel block (L / 8) mod BLOCKS.
// BUF: the patch rows through raw buffer loads (a row off the board gets an offset beyond the buffer and reads as zero: no
// selects; the board offset is a scalar) -- 7 % faster at 512 boards, 2 % slower at 128: the launcher picks by batch size.
template <bool BUF = false>
__global__ __launch_bounds__(512) void wgrad_wino3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ scratch, int n, int spx) {
    using T = WgradWino3;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // two operand sets: V [36][2][8 tiles][16], dM [36][2][8 tiles][16]

    const int wg_k = blockIdx.x >> 3;
    const int blk = wg_k % T::BLOCKS, cob = blk >> 2, cib = blk & 3;
    const int slice = (blockIdx.x & 7) * spx + wg_k / T::BLOCKS, slices = 8 * spx;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, j = lane & 15;

    // ---- MFMA roles: wave = (position group pg: positions 9 pg .. 9 pg + 8) x (output-channel group cc)
    const int pg = wave & 3, cc = wave >> 2;
    f32x4 acc[9][2];
#pragma unroll
    for (int p = 0; p < 9; p++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[p][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- transform roles: wave w < 4: input planes 8 w .. 8 w + 7 of the block's 32; wave w >= 4: gradient planes
    // 8 (w - 4) .. + 7.  Lane = (plane pl of the wave's eight, tile row tr of the half, tile column ttx): the four tiles of a
    // tile row are the four lanes of a quad (halo columns by DPP).
    const bool grad = wave >= 4;
    const int ttx = lane & 3, tr = (lane >> 2) & 1, pl = lane >> 3;
    const int ch = (wave & 3) * 8 + pl;                             // channel of the block's 32
    const int tile = tr * 4 + ttx, c16 = ch & 15, grp = ch >> 4;    // tile of the half, channel of the 16-channel group, group
    const int wslot = grp * 128 + tile * 16 + ((c16 + 4 * (tile >> 1)) & 15);
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(grad ? dy : x), 0, (unsigned)n * T::C * T::GPLANE * 4u, 0x00020000);
    const unsigned plane_off = (unsigned)(((grad ? cob * T::CO_B : cib * T::CI_B) + ch) * T::GPLANE + 4 * ttx) * 4u;
    const float* plane0 = grad ? dy + ((size_t)cob * T::CO_B + ch) * T::GPLANE + 4 * ttx : x + ((size_t)cib * T::CI_B + ch) * T::GPLANE + 4 * ttx;

    const int nboards = slice < n ? (n - slice + slices - 1) / slices : 0;
    const int total = nboards * 2;                    // half boards of this workgroup's stream
    // The rows of this thread's patch, straight from global memory into registers, a half board ahead (round 4's first form
    // staged them in LDS by LDS-DMA: that path moves 16 bytes per clock and CU -- 64 cycles per plane request whatever its
    // size --, 4 100 of a half board's 5 700 cycles, and every wave that requests blocks on it; profiles/r04_wgrad_wino3.md).
    // Input: patch rows 4 trow - 1 .. 4 trow + 4; gradient: tile rows 4 trow .. 4 trow + 3; rows off the board: zero.
    // rows off the board: a per-lane offset past ANY legal num_records (the range check sees the per-lane offset only, not
    // the scalar board offset; apz_wgrad_wino admits n <= 32768 boards = 0xF0000000 bytes < WGW3_OOB, and WGW3_OOB + 16 does not wrap)
    constexpr unsigned WGW3_OOB = 0xF8000000u;
    f32x4 nx[6];
    auto prefetch = [&](int u) {
        const int uu = u < total ? u : total - 1;     // (past the end: a harmless repeat)
        const int b = slice + (uu >> 1) * slices, trow = 2 * (uu & 1) + tr;
        const float* pb = plane0 + (size_t)b * T::C * T::GPLANE;
        const unsigned soff = (unsigned)b * (unsigned)(T::C * T::GPLANE * 4);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (grad && i >= 4) break;
            const int R = grad ? 4 * trow + i : 4 * trow - 1 + i;
            const bool in = R >= 0 && R <= 14;
            if constexpr (BUF) {
                nx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, in ? plane_off + (unsigned)R * 64u : WGW3_OOB, soff, 0));
            } else {
                nx[i] = *reinterpret_cast<const f32x4*>(pb + (in ? R : 0) * 16);
                if (!in) nx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    if (total > 0) prefetch(0);
#ifdef APZ_WGW3_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_readcyclecounter();
#endif

    for (int u = 0; u < total; u++) {
        float* opv = lds + (u & 1) * T::SET_FLOATS;   // this half's operand set (the other one: the previous half's MFMAs)
        float* opm = opv + T::OP_FLOATS;
        WGW3_STAMP(0)
        if (!APZ_WGW3_NO_TRANSFORM) {
            if (!grad) {
                // ---- V = B^T d B of (input channel ch, tile (trow, ttx)): the 6 x 6 patch, rows first
                float xr[6][6];
#pragma unroll
                for (int i = 0; i < 6; i++) {         // patch row i, columns -1 .. 4
                    const f32x4 c03 = nx[i];
                    xr[i][0] = wgw_quad_neighbour<false>(c03[3], ttx);
                    xr[i][1] = c03[0];
                    xr[i][2] = c03[1];
                    xr[i][3] = c03[2];
                    xr[i][4] = c03[3];
                    xr[i][5] = wgw_quad_neighbour<true>(c03[0], ttx);
                }
                prefetch(u + 1);                      // in flight during the barrier and this half's MFMAs
                float y[6][6];                        // y = B^T d (rows 0 / 5, 1 / 2, 3 / 4 share their partial sums)
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    y[0][k] = __builtin_fmaf(4.f, xr[0][k], __builtin_fmaf(-5.f, xr[2][k], xr[4][k]));
                    y[5][k] = __builtin_fmaf(4.f, xr[1][k], __builtin_fmaf(-5.f, xr[3][k], xr[5][k]));
                    const float a = __builtin_fmaf(-4.f, xr[2][k], xr[4][k]), b = __builtin_fmaf(-4.f, xr[1][k], xr[3][k]);
                    y[1][k] = a + b;
                    y[2][k] = a - b;
                    const float cdiff = xr[4][k] - xr[2][k], d = xr[3][k] - xr[1][k];
                    y[3][k] = __builtin_fmaf(2.f, d, cdiff);
                    y[4][k] = __builtin_fmaf(-2.f, d, cdiff);
                }
                float* dst = opv + wslot;
#pragma unroll
                for (int ir = 0; ir < 6; ir++) {
                    const float* v = y[ir];
                    const float a = __builtin_fmaf(-4.f, v[2], v[4]), b = __builtin_fmaf(-4.f, v[1], v[3]);
                    const float cdiff = v[4] - v[2], d = v[3] - v[1];
                    float o[6];
                    o[0] = __builtin_fmaf(4.f, v[0], __builtin_fmaf(-5.f, v[2], v[4]));
                    o[1] = a + b;
                    o[2] = a - b;
                    o[3] = __builtin_fmaf(2.f, d, cdiff);
                    o[4] = __builtin_fmaf(-2.f, d, cdiff);
                    o[5] = __builtin_fmaf(4.f, v[1], __builtin_fmaf(-5.f, v[3], v[5]));
#pragma unroll
                    for (int k = 0; k < 6; k++) dst[(ir * 6 + k) * 256] = o[k];      // 2 groups x 128 floats per position
                }
            } else {
                // ---- dM = A dY A^T of (output channel ch, tile (trow, ttx)): 4 x 4 -> 6 x 6
                const f32x4 d0 = nx[0], d1 = nx[1], d2 = nx[2], d3 = nx[3];
                prefetch(u + 1);
                f32x4 m[6];
                m[0] = d0;
                m[5] = d3;
                {
                    const f32x4 s02 = d0 + d2, s13 = d1 + d3;
                    m[1] = s02 + s13;
                    m[2] = s02 - s13;
                    const f32x4 sv = d0 + 4.f * d2, tv = 2.f * d1 + 8.f * d3;
                    m[3] = sv + tv;
                    m[4] = sv - tv;
                }
                float* dst = opm + wslot;
#pragma unroll
                for (int ir = 0; ir < 6; ir++) {
                    const f32x4 w = m[ir];
                    const float s02 = w[0] + w[2], s13 = w[1] + w[3];
                    const float sv = __builtin_fmaf(4.f, w[2], w[0]), tv = __builtin_fmaf(8.f, w[3], 2.f * w[1]);
                    float o[6];
                    o[0] = w[0];
                    o[1] = s02 + s13;
                    o[2] = s02 - s13;
                    o[3] = sv + tv;
                    o[4] = sv - tv;
                    o[5] = w[3];
#pragma unroll
                    for (int k = 0; k < 6; k++) dst[(ir * 6 + k) * 256] = o[k];
                }
            }
        }
#ifdef APZ_WGW3_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        WGW3_STAMP(2)
        // the half's operand arrays are complete -- and everybody is past the previous half's MFMAs, whose operand set the
        // NEXT transform overwrites: one barrier per half board
        __syncthreads();
        WGW3_STAMP(3)
        // ---- dU[pos][co][ci] += dM[pos][co][tile] * V[pos][ci][tile] over the half's 8 tiles (two k-steps):
        // A = dM (m = co), B = V (n = ci), k = tile; lane (q, j): tile t = 4 s + q, channel slot (j + 4 (t >> 1)) & 15.
        // Operands are requested one step (position, k-step) ahead of their two MFMAs and the order is pinned.
        if (!APZ_WGW3_NO_MFMA) {
            // (every accumulator's two MFMAs of the half are nine steps apart: k-step 0 of all positions, then k-step 1)
            auto fetch = [&](int st, float* o) {      // step st = 9 s + p
                const int pos = pg * 9 + (st % 9), t = 4 * (st / 9) + q;
                const int slot = t * 16 + ((j + 4 * (t >> 1)) & 15);
                o[0] = opm[(pos * 2 + cc) * 128 + slot];
                o[1] = opv[(pos * 2) * 128 + slot];
                o[2] = opv[(pos * 2 + 1) * 128 + slot];
            };
            float cur[3], nxt[3];
            fetch(0, cur);
#pragma unroll
            for (int st = 0; st < 18; st++) {
                if (st + 1 < 18) fetch(st + 1, nxt);
                const int p = st % 9;
                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[0], cur[1], acc[p][0], 0, 0, 0);
                acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[0], cur[2], acc[p][1], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 3; i++) cur[i] = nxt[i];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        WGW3_STAMP(4)
    }
    // ---- partial dg of this slice and channel block: accumulator (p, b), lane (q, j), register r holds dU at
    // pos = 9 pg + p, co = 32 cob + 16 cc + 4 q + r, ci = 32 cib + 16 b + j.  The 36 positions of a (co, ci) pair sit in four
    // waves: they meet in LDS ([36][32 co][32 ci + 1 pad]) and every thread turns two pairs' 6x6 into G^T dU G.
    constexpr int CS = T::EPI_CS, PST = T::EPI_PST;
    float* du = lds;
    float* outw = scratch + (size_t)slice * WgradWino::SCRATCH_FLOATS_PER_SLICE;
    const float G[6][3] = {{0.25f, 0.f, 0.f},           {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    __syncthreads();                                  // the last half's operand reads are done
#pragma unroll
    for (int p = 0; p < 9; p++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) du[(pg * 9 + p) * PST + (cc * 16 + 4 * q + r) * CS + b * 16 + j] = acc[p][b][r];
    __syncthreads();
#pragma unroll
    for (int e0 = 0; e0 < 1024; e0 += T::THREADS) {
        const int e = e0 + tid, col = e >> 5, cil = e & 31;
        float u[36];
#pragma unroll
        for (int p = 0; p < 36; p++) u[p] = du[p * PST + col * CS + cil];
        float tt[3][6];                              // tt[x][k] = sum_i G[i][x] dU[i][k]
#pragma unroll
        for (int x3 = 0; x3 < 3; x3++)
#pragma unroll
            for (int k = 0; k < 6; k++) {
                float v = 0.f;
#pragma unroll
                for (int i6 = 0; i6 < 6; i6++) v += G[i6][x3] * u[i6 * 6 + k];
                tt[x3][k] = v;
            }
        float* d = outw + ((size_t)(cob * T::CO_B + col) * T::C + cib * T::CI_B + cil) * 9;
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
#ifdef APZ_WGW3_STAMPS
    WGW3_STAMP(5)
    if (blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 6; k++) apz_wgw3_stamps[wave][k] = st_acc[k];
#endif
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
