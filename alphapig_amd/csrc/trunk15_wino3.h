// Trunk 3x3 convolution (128 -> 128 channels, 15x15 board) + folded BN + (residual) + ReLU as a
// fused F(4x4,3x3) Winograd convolution on the fp32 matrix cores -- SINGLE PASS: a work item is
// two boards x 64 output channels with all 36 transformed positions.  gfx950 only.
// Successor of round 1's two-pass kernel (same math, same HBM layouts, same packed weights; git history / HISTORY.md section 4).
//
// Why (measurements of that kernel, profiles/r01_trunk_winograd.md): two boards x 128
// channels x 36 positions of accumulators (590 KB) do not fit the 512 KB register file, so round 1 made
// two passes over the input (18 positions each), parked pass 0's partial outputs in `out` and re-read
// them: 2.5x the direct-convolution HBM/fabric traffic, two epilogues (25 % of the launch) and an
// input transform run by one wave per SIMD against the other's MFMA stream (16 %).
// Here the accumulator budget is spent the other way round: 2 boards x 64 channels x 36 positions
// (295 KB) -- a weight fragment still feeds two MFMAs, nothing is parked, every output is written
// once, the input is read once per channel half (2x, as before), and all eight waves do the same
// thing in every chunk.  Price: the input transform runs once per channel half (2x the VALU work of
// the two-pass kernel); it is spread over all 512 threads (half a 6x6 tile each) and sits inside each wave's own
// MFMA stream, where a VALU instruction costs ~4 cycles instead of one MFMA slot (tools/mfma_valu_probe).
//
// Work item of a workgroup = (board pair, channel half h) (mapping: see the kernel).  Wave w: ct = w&3 (16 output
// channels cot*16.., cot = 4h + ct), ph = w>>2 (transformed rows 3ph..3ph+2 = 18 positions), both
// boards: 2 x 18 x 4 = 144 accumulator registers.  Per item the 128 input channels stream through in
// 16 chunks of 8, one barrier per chunk:
//   iteration g: [barrier] raw(g+2) regs -> LDS; issue loads raw(g+3);
//                transform raw(g+1) -> V[(g+1)&1] (thread = (board, channel, tile) x row half);
//                72 MFMAs over V[g&1]; weight ring (5 x 16 B per lane) refilled one k-step ahead.
// Epilogue of an item: wave (ct, ph) holds rows 3ph..3ph+2 of M; Y = A^T M A needs all six, so the
// two waves of a channel tile swap 12 values per (channel, tile) through LDS (X) -- wave ph sends its
// row-partial of board 1-ph and finishes board ph: + bias (+ residual), ReLU, whole-plane stores
// through the wave-private staging area.
//
// Layouts.  in / resid / out: rows16 [n][128][15][16] (col 15 == 0), as trunk15_ring.h.
// upk: [cot 8][ph 2][c4 32][lane 64][20] (wino_common.h): lane (q = lane>>4, j = lane&15) holds
//      U[row 3*ph + ii][k] at index 6*ii + k of co = cot*16 + j, ci = c4*4 + q.
// raw (LDS): [2 boards x 8 channels] planes, row stride 20, plane stride 340, zero halo.
// V (LDS): [pp 18][board 2][ch 8][tile 16][2]: pp = position pair (row i, columns 2kp, 2kp+1) = 3i + kp;
//      the B operands of positions 2pp, 2pp+1 are one conflict-free ds_read_b64, no padding.
// X (LDS, inside V[1], which is idle during an epilogue): [wave 8][12 values][lane 64].
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wino_common.h"

namespace apz {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Wino3 {
    static constexpr int C = 128, CK = 8, NCHUNK = C / CK;            // 16 iterations per item
    static constexpr int GPLANE = 240;                 // floats per plane in HBM (15 rows x 16)
    static constexpr int RROW = 20, RPS = 17 * RROW;   // LDS row / plane stride
    static constexpr int RFRONT = 24;
    static constexpr int RAW_FLOATS = RFRONT + 2 * CK * RPS;          // 5464
    static constexpr int VPP = 2 * CK * 16 * 2;        // floats per position pair: 256 units x 2
    static constexpr int V_FLOATS = 18 * VPP;          // 9216 (36 KiB)
    static constexpr int XW = 3 * 64 * 4;              // exchange floats per wave
    static constexpr int SROW = 20, SPLANE = 16 * SROW;               // epilogue staging: 16 rows x 20 floats per plane
    static constexpr int STAGE_FLOATS = 8 * 4 * SPLANE;               // 8 waves x 4 planes (40 KiB)
    static constexpr int LDS_FLOATS = 2 * RAW_FLOATS + 2 * V_FLOATS + STAGE_FLOATS;   // 39600 floats = 154.7 KiB
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int UROW = 20;                    // floats per lane and k-step in upk
    static constexpr int USTEP = 64 * UROW;            // floats per k-step of one (cot, ph)
    static_assert(8 * XW <= V_FLOATS, "X lives inside V[1]");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

#ifndef APZ3_EARLY_BARRIER
#define APZ3_EARLY_BARRIER 1
#endif
#ifndef APZ3_EARLY_RESID
#define APZ3_EARLY_RESID 1
#endif


#ifdef APZ3_EPI_WAIT
#define APZ3_FENCE()                                  \
    {                                                 \
        __builtin_amdgcn_s_waitcnt(0xc07f);           \
        wave_lds_fence();                             \
    }
#else
#define APZ3_FENCE() wave_lds_fence()
#endif

#ifdef APZ3_DEBUG_X
__device__ float apz_wino3_dbg[8 * 4 * 2 * 64 * 4];   // [wave][r][sent/received][lane][4]: P2 of the exchange
#endif
#ifdef APZ_WINO3_STAMPS
// cycle accounting (-DAPZ_WINO3_STAMPS build of tools/wino3_bench.hip): [workgroup 4][wave 8][phase 8], read with hipMemcpyFromSymbol
__device__ unsigned long long apz_wino3_stamps[4 * 8 * 8];
#endif

// RELU = false: the plain convolution + bias (training graph: forward before BatchNorm, data gradient)
// QUARTER (round 4): work item = (board pair, 32 output channels) -- four workgroups per pair, for batches with fewer pairs
// than a quarter of the CUs (the reference's training batch of 128, a 64-match arena): wave = (16-channel tile w & 1, board
// (w >> 1) & 1, row half w >> 2) with 18 x 4 = 72 accumulators; a weight fragment feeds ONE MFMA per wave (the two board
// waves of a tile fetch the same fragment: L1), the input transform is unchanged (all eight waves, both boards).  The two row
// halves of a (tile, board) meet through X as before: wave ph = 0 finishes channel sub-steps 0, 1, wave ph = 1 sub-steps 2, 3.
// Same MFMA order per output, same transform and epilogue formulas: the same bits as the 64-channel items.
// STATS (the training forward: no residual, no ReLU; `resid` then carries a double [128][n][2] buffer instead): the epilogue
// also leaves, per (output channel, board), the sum and the sum of squares of the board's 225 outputs -- the per-split
// partials of the BatchNorm that follows (conv_train.h: bn_apply adds a channel's partials itself), so that the training
// step has no statistics pass over the tensor this kernel has just written.  A board's 225 values: a fixed fp32 tree (the
// lane's 4x4 patch row by row, its four rows, then the sixteen lanes of the tile group by DPP adds -- in double the same
// chain cost 7.7 us per 512-board launch, more than half of the pass it replaces); boards are added in double by the consumer.
template <int CTRL>
__device__ __forceinline__ float wino3_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wino3_row16_sum(float v) {
    v = wino3_dpp_add<0xB1>(v);       // quad_perm [1, 0, 3, 2]
    v = wino3_dpp_add<0x4E>(v);       // quad_perm [2, 3, 0, 1]
    v = wino3_dpp_add<0x141>(v);      // row_half_mirror: the other quad of the 8
    return wino3_dpp_add<0x140>(v);   // row_mirror: the other 8 of the 16
}

template <bool RESID, bool RELU = true, bool QUARTER = false, bool STATS = false>
__global__ __launch_bounds__(512) void trunk15_wino3_kernel(const float* __restrict__ in, const float* __restrict__ upk,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ resid, float* __restrict__ out,
                                                            int n) {
    using T = Wino3;
#ifdef APZ_WINO3_STAMPS
    // phases: 0 prologue, 1 barrier wait, 2 chunk body (staging + transform + MFMA); epilogue: 3 compute, 4 barrier waits,
    // 5 residual wait + staging, 6 stores; 7 total
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_readcyclecounter();
    const unsigned long long st_t0 = st_t;
#define APZ3_STAMP(ph_)                                               \
    {                                                                 \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        st_acc[ph_] += now_ - st_t;                                   \
        st_t = now_;                                                  \
    }
#else
#define APZ3_STAMP(ph_)
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* rawb = lds;                                        // [2][RAW_FLOATS]
    float* vb = lds + 2 * T::RAW_FLOATS;                      // [2][V_FLOATS]
    float* stg = lds + 2 * T::RAW_FLOATS + 2 * T::V_FLOATS;   // [8 waves][4 planes][16 rows x 20]
    float* xb = vb + T::V_FLOATS;                             // X: inside V[1] (free between an item's last MFMA and the next item's first transform)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, j = lane & 15;
    const int ct = QUARTER ? (wave & 1) : (wave & 3);          // 16-channel tile of the item
    const int bsel = (wave >> 1) & 1;                          // QUARTER: this wave's board


    // ---- work items of this workgroup.  Item = (board pair, channel half h).
    // Grids that are a multiple of 16 run in "duo" mode: two workgroups that the dispatcher (observed: round-robin
    // over the 8 XCDs) places on the same XCD take the two channel halves of the SAME pairs at the same time, so
    // the second read of a pair's input planes is an L2 hit instead of a second trip over the fabric.  Placement
    // changes only speed: any other grid (and any other placement) computes the same thing.
    // duo:   block b -> XCD group c = b % 8, i = b / 8; duo d = (i / 2) * 8 + c takes pairs d, d + G/2, ...; h = i & 1.
    // plain: block b takes pairs b, b + G, ... and both halves of each.
    const int npairs = (n + 1) >> 1, G_ = (int)gridDim.x, b_ = (int)blockIdx.x;
    // QUARTER: the same with four channel quarters: grids that are a multiple of 32 put the quarters of a pair on blocks
    // b, b + 8, b + 16, b + 24 (one XCD); any other grid gives a workgroup whole pairs, quarter after quarter.
    const bool duo = QUARTER ? (G_ & 31) == 0 : (G_ & 15) == 0;
    const int pair0 = duo ? (QUARTER ? ((b_ >> 5) * 8 + (b_ & 7)) : ((b_ >> 4) * 8 + (b_ & 7))) : b_;
    const int pstride = duo ? (QUARTER ? (G_ >> 2) : (G_ >> 1)) : G_;
    const int h_fix = QUARTER ? ((b_ >> 3) & 3) : ((b_ >> 3) & 1);   // duo mode: this workgroup's channel half (quarter)
    const int np = pair0 < npairs ? (npairs - pair0 + pstride - 1) / pstride : 0;   // board pairs of this workgroup
    const int nitems = duo ? np : (QUARTER ? 4 : 2) * np;
    const int total_iters = nitems * T::NCHUNK;
    if (np == 0) return;                        // (uniform, before any barrier)
    auto item_pair = [&](int t) { return pair0 + (duo ? t : (QUARTER ? (t >> 2) : (t >> 1))) * pstride; };
    auto item_half = [&](int t) { return duo ? h_fix : (QUARTER ? (t & 3) : (t & 1)); };   // channel half (QUARTER: quarter)

    // All global memory traffic goes through raw buffer instructions: descriptor (SGPRs) + per-lane 32-bit offset
    // + wave-uniform SGPR offset.  No 64-bit per-lane addresses (registers, VALU), and lanes that must not take
    // part (60..63 of a 960-byte plane) get an offset beyond the buffer: their loads return 0, their stores are dropped.
    const unsigned plane_b = T::GPLANE * 4;                    // 960
    const unsigned act_bytes = (unsigned)n * T::C * plane_b;   // launchers keep this below 2^31
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, act_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_res =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RESID ? resid : in), 0, act_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, act_bytes, 0x00020000);
    static_assert(!STATS || (!RESID && !RELU), "STATS: the training forward (bias only)");
    const __amdgpu_buffer_rsrc_t r_st =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(STATS ? resid : in), 0, STATS ? (unsigned)n * T::C * 16u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, T::C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_u =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(upk), 0, (unsigned)(WinoPack::UPK_FLOATS * 4), 0x00020000);
    auto bload = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    // Stores keep soffset = 0 and add the uniform part into the per-lane offset.  With an SGPR soffset hipcc (ROCm 7.2)
    // emits no wait state between a 128-bit buffer store and a VALU write of its data registers (LLVM's hazard
    // recognizer exempts that form), and on gfx950 the store then picks up the NEW value of the last dword in some
    // lanes (found the hard way: one element per 4x4 tile of every 16th channel wrong, and only in some builds).
    auto bstore = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff, const f32x4 v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff + soff, 0, 0);
    };

    // ---- staging roles: 2 boards x 8 planes x 60 pieces of 16 B; 32 threads per plane, two pieces each (four of them
    // twice).  thread -> plane tid>>5: board = tid>>8 (wave-uniform: waves 0..3 stage board 0), channel (tid>>5)&7.
    // Every thread issues exactly two loads and two LDS stores per iteration, unconditionally (exact vmcnt counting).
    const int st_k = tid & 31, st_k2 = (st_k + 32 < 60) ? st_k + 32 : st_k;
    const unsigned st_vo = ((tid >> 5) & 7) * plane_b + st_k * 16, st_vo2 = ((tid >> 5) & 7) * plane_b + st_k2 * 16;
    f32x4 rg[2];
    auto raw_fetch = [&](int g) {               // global -> registers (iteration g of this workgroup's stream, clamped)
        g = g < total_iters ? g : total_iters - 1;
        const int bdp = 2 * item_pair(g / T::NCHUNK) + (wave >> 2), c = g & (T::NCHUNK - 1);
        const int bd = bdp < n ? bdp : n - 1;
        const unsigned so = (unsigned)(bd * T::C + c * T::CK) * plane_b;
        rg[0] = bload(r_in, st_vo, so);
        rg[1] = bload(r_in, st_vo2, so);
    };
    auto raw_store = [&](int par) {             // registers -> raw LDS buffer `par` (= iteration & 1)
        float* dst = rawb + par * T::RAW_FLOATS + T::RFRONT + (tid >> 5) * T::RPS;
        *reinterpret_cast<f32x4*>(dst + (st_k >> 2) * T::RROW + (st_k & 3) * 4) = rg[0];
        *reinterpret_cast<f32x4*>(dst + (st_k2 >> 2) * T::RROW + (st_k2 & 3) * 4) = rg[1];
    };
    // Everything below is instantiated twice, for ph = 0 and ph = 1 (wave-uniform branch at the bottom): the row
    // half decides transform formulas and epilogue register indices, and a branch inside the chunk body would cut
    // the basic block the scheduler interleaves the transform's VALU work with the MFMAs in.
    auto run = [&](auto PH) {
    constexpr int ph = decltype(PH)::value;
    // ---- transform roles: thread = unit (board, channel, tile) x row half (tid>>8 == ph: rows 3ph..3ph+2)
    const int unit = tid & 255;
    const int tty = (unit >> 2) & 3, ttx = unit & 3;
    const int tr_off = T::RFRONT + (unit >> 4) * T::RPS + (4 * tty - 1 + ph) * T::RROW + 4 * ttx;   // HI half skips patch row 0
    const int tv_off = (9 * ph) * T::VPP + unit * 2;
    // The transform of one chunk is cut into 18 slices (one per MFMA slot of the chunk body, see below); all of its
    // temporaries are named here so that a slice can pick up where the previous one stopped.
    //   slices 0..2: LDS reads of the five patch rows, as column pairs that need no register moves:
    //                xr[i][0] = (col -1, col 4) (one ds_read2_b32), xr[i][1] = (col 0, col 1), xr[i][2] = (col 2, col 3);
    //   slices 3..7: B^T over the rows (elementwise in the columns), 18 packed operations:
    //                ph 0: y0 = 4x0 - 5x2 + x4, y1 = a + b, y2 = a - b with a = x4 - 4x2, b = x3 - 4x1   (x = patch rows 0..4)
    //                ph 1: y3 = c + 2d, y4 = c - 2d with c = z3 - z1, d = z2 - z0, y5 = 4z0 - 5z2 + z4  (z = patch rows 1..5)
    //   slices 8..16: B^T over the columns of each of the three rows (y0 = 4x0 - 5x2 + x4, y1/y2 = (x4 - 4x2) +- (x3 - 4x1), y3/y4 = (x4 - x2) +- 2(x3 - x1), y5 = 4x1 - 5x3 + x5), three slices per row,
    //                the row's three ds_write_b64 in its last slice.
    f32x2 xr[5][3], u0[3], u1[3], u2[3], tt[3][3];
    float o14[4];
    auto tslice = [&](int par, auto KK) {       // raw[par] -> V[par], rows 3ph .. 3ph+2; slice KK of 18
        constexpr int K = decltype(KK)::value;
        const float* rp = rawb + par * T::RAW_FLOATS + tr_off;
        float* vp = vb + par * T::V_FLOATS + tv_off;
        auto load_row = [&](int i) {
#ifdef APZ3_ABLATE_TREADS             /* measurement build: no LDS reads in the transform */
            xr[i][0] = f32x2{(float)i, 1.f};
            xr[i][1] = f32x2{2.f, (float)lane};
            xr[i][2] = f32x2{3.f, 4.f};
#else
            const f32x4 c03 = *reinterpret_cast<const f32x4*>(rp + i * T::RROW);
#ifdef APZ3_HALO_DPP                  /* experiment (round 6): the halo columns from the neighbouring tile lanes (a quad = the four tile columns of a tile row) instead of two 4-way conflicted LDS reads */
            {
                const float c3 = c03[3], c0 = c03[0];
                const int l = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, c3), 0x90, 0xF, 0xF, true);     // quad_perm [0,0,1,2]
                const int r = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, c0), 0xF9, 0xF, 0xF, true);     // quad_perm [1,2,3,3]
                xr[i][0] = f32x2{__builtin_bit_cast(float, (unsigned)l & (ttx == 0 ? 0u : 0xffffffffu)),
                                 __builtin_bit_cast(float, (unsigned)r & (ttx == 3 ? 0u : 0xffffffffu))};
            }
#else
            xr[i][0] = f32x2{rp[i * T::RROW - 1], rp[i * T::RROW + 4]};
#endif
            xr[i][1] = f32x2{c03[0], c03[1]};
            xr[i][2] = f32x2{c03[2], c03[3]};
#endif
        };
        if constexpr (K == 0) {
            load_row(0);
            load_row(1);
        } else if constexpr (K == 1) {
            load_row(2);
            load_row(3);
        } else if constexpr (K == 2) {
            load_row(4);
        } else if constexpr (K == 3) {          // ph 0: b = x3 - 4 x1;  ph 1: c = z3 - z1
#pragma unroll
            for (int cp = 0; cp < 3; cp++) u2[cp] = ph == 0 ? fma2(-4.f, xr[1][cp], xr[3][cp]) : xr[3][cp] - xr[1][cp];
        } else if constexpr (K == 4) {          // ph 0: a = x4 - 4 x2;  ph 1: d = z2 - z0
#pragma unroll
            for (int cp = 0; cp < 3; cp++) u1[cp] = ph == 0 ? fma2(-4.f, xr[2][cp], xr[4][cp]) : xr[2][cp] - xr[0][cp];
        } else if constexpr (K == 5) {          // both: x4 - 5 x2 (z4 - 5 z2)
#pragma unroll
            for (int cp = 0; cp < 3; cp++) u0[cp] = fma2(-5.f, xr[2][cp], xr[4][cp]);
        } else if constexpr (K == 6) {          // y0 = 4 x0 + (x4 - 5 x2)   (y5 = 4 z0 + (z4 - 5 z2))
#pragma unroll
            for (int cp = 0; cp < 3; cp++) u0[cp] = fma2(4.f, xr[0][cp], u0[cp]);
        } else if constexpr (K == 7) {
#pragma unroll
            for (int cp = 0; cp < 3; cp++) {
                if (ph == 0) {
                    tt[0][cp] = u0[cp];
                    tt[1][cp] = u1[cp] + u2[cp];
                    tt[2][cp] = u1[cp] - u2[cp];
                } else {
                    tt[0][cp] = fma2(2.f, u1[cp], u2[cp]);
                    tt[1][cp] = fma2(-2.f, u1[cp], u2[cp]);
                    tt[2][cp] = u0[cp];
                }
            }
        } else if constexpr (K >= 8 && K <= 16) {
            constexpr int ii = (K - 8) / 3, part = (K - 8) % 3;
            const float v0 = tt[ii][0][0], v5 = tt[ii][0][1], v1 = tt[ii][1][0], v2 = tt[ii][1][1], v3 = tt[ii][2][0],
                        v4 = tt[ii][2][1];
            if constexpr (part == 0) {
                const float a = __builtin_fmaf(-4.f, v2, v4), b = __builtin_fmaf(-4.f, v1, v3);
                o14[0] = a + b;
                o14[1] = a - b;
            } else if constexpr (part == 1) {
                const float c = v4 - v2, d = v3 - v1;
                o14[2] = __builtin_fmaf(2.f, d, c);
                o14[3] = __builtin_fmaf(-2.f, d, c);
            } else {
                const float o0 = __builtin_fmaf(4.f, v0, __builtin_fmaf(-5.f, v2, v4));
                const float o5 = __builtin_fmaf(4.f, v1, __builtin_fmaf(-5.f, v3, v5));
                // two scalar stores per position pair: ds_write2_b32 takes any two registers (a b64 store wants an
                // aligned pair and costs a v_mov per value)
#ifdef APZ3_ABLATE_TWRITES            /* measurement build: one LDS write instead of six per row */
                vp[(ii * 3 + 0) * T::VPP] = ((o0 + o14[0]) + (o14[1] + o14[2])) + (o14[3] + o5);
#else
                vp[(ii * 3 + 0) * T::VPP] = o0;
                vp[(ii * 3 + 0) * T::VPP + 1] = o14[0];
                vp[(ii * 3 + 1) * T::VPP] = o14[1];
                vp[(ii * 3 + 1) * T::VPP + 1] = o14[2];
                vp[(ii * 3 + 2) * T::VPP] = o14[3];
                vp[(ii * 3 + 2) * T::VPP + 1] = o5;
#endif
            }
        }
    };
#define APZ3_ALL18(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15) F(16) F(17)
    auto transform = [&](int par) {             // the whole transform at once (prologue)
#define APZ3_TS(k) tslice(par, std::integral_constant<int, k>{});
        APZ3_ALL18(APZ3_TS)
#undef APZ3_TS
    };

    // ---- weight stream of this wave: k-step (item half h, c4) -> five f32x4 per lane; ring = one k-step,
    // piece v refilled right after its last MFMA with the same piece of the next k-step (28 MFMAs ahead).
    const unsigned ulane = lane * (T::UROW * 4);
    auto uload = [&](int ks, int v) {           // ks = global k-step of this workgroup's stream (32 per item), v = piece 0..4
        const int hh = item_half(ks >> 5), kk = ks & 31;
        const unsigned so = (unsigned)((((QUARTER ? 2 : 4) * hh + ct) * 2 + ph) * 32 + kk) * (T::USTEP * 4);
        if (v == 4) {                           // values 16, 17 (+ 2 pad floats that are never loaded: dead registers under an
            const auto w = __builtin_amdgcn_raw_buffer_load_b64(r_u, ulane + 64, so, 0);   // in-flight load get reused -> WAW waits)
            const f32x2 f = __builtin_bit_cast(f32x2, w);
            return f32x4{f[0], f[1], 0.f, 0.f};
        }
        return bload(r_u, ulane + v * 16, so);
    };
    f32x4 ur[5];
    {
        raw_fetch(0);
        const f32x4 r0 = rg[0], r1 = rg[1];
        raw_fetch(1);
        // zero halo of both raw buffers, while the first planes are on their way
        for (int i = tid * 4; i < 2 * T::RAW_FLOATS; i += 2048) *reinterpret_cast<f32x4*>(&lds[i]) = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                        // zero fill done
        raw_store(1);
        rg[0] = r0;
        rg[1] = r1;
        raw_store(0);
    }
    __syncthreads();
    transform(0);
    raw_fetch(2);
    // the first weights AFTER the staging loads: at the top of the chunk loop the staging registers are waited for
    // with vmcnt(N), N = the loads issued after them -- N is the minimum over the paths into the loop, and this
    // path must not make it 0 (a full drain of the weight ring at every chunk)
#pragma unroll
    for (int v = 0; v < 5; v++) ur[v] = uload(0, v);
    APZ3_STAMP(0)

#if defined(APZ3_PRIO) && APZ3_PRIO == 1
    if (ph == 1) __builtin_amdgcn_s_setprio(1);     // the later-dispatched half loses issue arbitration by age
#elif defined(APZ3_PRIO) && APZ3_PRIO == 2
    if (ph == 0) __builtin_amdgcn_s_setprio(1);
#endif
    f32x2 bc0, bc1;                                 // B operands of the current slot (both boards), carried across chunks
#if APZ3_EARLY_BARRIER
    __syncthreads();                                // V[0] (chunk 0) complete
    {
        const float* vp0 = vb + (9 * ph) * T::VPP + (q * 16 + j) * 2 + (QUARTER ? bsel * 256 : 0);
        bc0 = *reinterpret_cast<const f32x2*>(vp0);
        if (!QUARTER) bc1 = *reinterpret_cast<const f32x2*>(vp0 + 256);
    }
#endif
    for (int t = 0; t < nitems; t++) {
        const int h = item_half(t);
        const int bd0 = 2 * item_pair(t);
        const bool two = bd0 + 1 < n;           // the last pair of an odd batch has one board (computed twice, stored once)
        f32x4 acc[QUARTER ? 1 : 2][18];
#pragma unroll
        for (int b = 0; b < (QUARTER ? 1 : 2); b++)
#pragma unroll
            for (int p = 0; p < 18; p++) acc[b][p] = f32x4{0.f, 0.f, 0.f, 0.f};

        // One chunk = 18 slots of 4 MFMAs (k-step s = slot / 9, position pair m = slot % 9, both boards).  A slot reads
        // the B operands of the NEXT slot, issues its MFMAs, runs one slice of the transform of chunk g+1 and, when a
        // weight piece has seen its last MFMA, refills it for the next k-step.  sched_barrier(0) pins the slots: left
        // alone, hipcc clusters the transform in front of the MFMAs and sinks the weight loads to their use.
        // The chunk loop is unrolled by two so that the LDS buffer parity (g & 1 == c & 1) is a compile-time
        // constant: every LDS address is then a loop-invariant register + immediate, no per-chunk address VALU.
#if defined(APZ3_PRIO) && APZ3_PRIO == 3
#define APZ3_SLOT_PRIO(k) if ((k) % 3 == 0) __builtin_amdgcn_s_setprio((((k) / 3) + ph) & 1);
#else
#define APZ3_SLOT_PRIO(k)
#endif
        // (APZ3_ABLATE_TRANSFORM: measurement build of tools/wino3_bench.hip -- staging and input transform of the chunk
        // body removed, the MFMAs run over whatever the prologue left in V: what the body would cost if V came ready-made)
#ifdef APZ3_ABLATE_TRANSFORM
#define APZ3_BODY_STAGING(k)
#else
#if defined(APZ3_ABLATE_STAGING)      /* no global -> LDS staging of the raw planes (the transform reads stale tiles) */
#define APZ3_BODY_STAGING(k) tslice(1 - par, std::integral_constant<int, (k)>{});
#elif defined(APZ3_ABLATE_TSLICE)     /* staging only, no transform */
#define APZ3_BODY_STAGING(k)                                                                     \
    if ((k) == 0) raw_store(par);                                                                \
    if ((k) == 1) raw_fetch(g + 3);
#else
#define APZ3_BODY_STAGING(k)                                                                     \
    if ((k) == 0) raw_store(par);                        /* raw(g+2) */                          \
    if ((k) == 1) raw_fetch(g + 3);                                                              \
    tslice(1 - par, std::integral_constant<int, (k)>{}); /* chunk g+1 */
#endif
#endif
        auto chunk = [&](int g, auto PAR) {
            constexpr int par = decltype(PAR)::value;
            const float* vp = vb + par * T::V_FLOATS + (9 * ph) * T::VPP + (q * 16 + j) * 2 + (QUARTER ? bsel * 256 : 0);   // QUARTER: own board only
            // APZ3_EARLY_BARRIER: the chunk's barrier sits between slots 16 and 17 of the PREVIOUS chunk (see there) and
            // bc0 / bc1 already hold this chunk's first operands.
            const float* vpn = vb + (1 - par) * T::V_FLOATS + (9 * ph) * T::VPP + (q * 16 + j) * 2 + (QUARTER ? bsel * 256 : 0);
#if !APZ3_EARLY_BARRIER
            __syncthreads();                    // V[par] complete, V[1-par] and raw[par] free, raw[1-par] visible
            APZ3_STAMP(1)
            bc0 = *reinterpret_cast<const f32x2*>(vp);
            if (!QUARTER) bc1 = *reinterpret_cast<const f32x2*>(vp + 256);
#endif
#define APZ3_SLOT(k)                                                                                               \
            {                                                                                                      \
                constexpr int s = (k) / 9, m = (k) % 9, sn = ((k) + 1) / 9, mn = ((k) + 1) % 9;                        \
                APZ3_SLOT_PRIO(k)                                                                                  \
                f32x2 bn0 = bc0, bn1 = bc1;                                                                        \
                if ((k) + 1 < 18) {                                                                                \
                    bn0 = *reinterpret_cast<const f32x2*>(vp + mn * T::VPP + sn * 128);                            \
                    if (!QUARTER) bn1 = *reinterpret_cast<const f32x2*>(vp + mn * T::VPP + 256 + sn * 128);        \
                } else if (APZ3_EARLY_BARRIER) {          /* the next chunk's first operands (V[1-par], complete) */ \
                    bn0 = *reinterpret_cast<const f32x2*>(vpn);                                                    \
                    if (!QUARTER) bn1 = *reinterpret_cast<const f32x2*>(vpn + 256);                                \
                }                                                                                                  \
                const float a0 = ur[m >> 1][(2 * m) & 3], a1 = ur[m >> 1][(2 * m + 1) & 3];                        \
                acc[0][2 * m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bc0[0], acc[0][2 * m], 0, 0, 0);          \
                if constexpr (!QUARTER) acc[1][2 * m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bc1[0], acc[1][2 * m], 0, 0, 0); \
                acc[0][2 * m + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bc0[1], acc[0][2 * m + 1], 0, 0, 0);  \
                if constexpr (!QUARTER) acc[1][2 * m + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bc1[1], acc[1][2 * m + 1], 0, 0, 0); \
                APZ3_BODY_STAGING(k)                                                                               \
                if ((m & 1) || m == 8) ur[m >> 1] = uload(2 * g + s + 1, m >> 1);                                  \
                bc0 = bn0;                                                                                         \
                bc1 = bn1;                                                                                         \
                if (APZ3_EARLY_BARRIER && (k) == 16) {                                                             \
                    /* The chunk barrier, one slot early: the transform's last V writes were in slice 16, the last    \
                       reads of V[par] were this slot's operand prefetch, raw[1-par] was last read in slices 0-2.     \
                       Slot 17 then fetches the next chunk's first operands while its own MFMAs run, instead of all  \
                       eight waves waiting out an LDS round trip with an empty matrix pipe after every barrier. */   \
                    __syncthreads();                                                                               \
                    APZ3_STAMP(1)                                                                                  \
                }                                                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
            }
            APZ3_ALL18(APZ3_SLOT)
#undef APZ3_SLOT
            APZ3_STAMP(2)
        };
        for (int c = 0; c < T::NCHUNK; c += 2) {
            chunk(t * T::NCHUNK + c, std::integral_constant<int, 0>{});
            chunk(t * T::NCHUNK + c + 1, std::integral_constant<int, 1>{});
        }

        // ---- epilogue of the item.  Lane (q, j): tile j = 4*ety + etx, channels cot*16 + 4q + r.
        // h_i = the k-direction transform A^T of row i (o0 = m0+m1+m2+m3+m4, o1 = (m1-m2) + 2(m3-m4), o2 = (m1+m2) + 4(m3+m4), o3 = (m1-m2) + 8(m3-m4) + m5).  With lo = (h0+h1+h2, h1-h2, h1+h2) from the ph = 0
        // wave and hi = (h3+h4, h3-h4, h5) from the ph = 1 wave:
        //   y0 = lo0 + hi0;  y1 = lo1 + 2 hi1;  y2 = lo2 + 4 hi0;  y3 = lo1 + 8 hi1 + hi2.
        // Wave ph finishes board ph and sends its partial of board 1-ph.  No divergence: every wave runs the same
        // barriers; the missing second board of an odd batch's last pair is computed from a copy of the first and
        // never stored.
        const int cot = (QUARTER ? 2 : 4) * h + ct;
        const int bd_own = ((QUARTER ? bsel == 0 : ph == 0) || !two) ? bd0 : bd0 + 1;
        // The epilogue's per-lane addresses are derived from an opaque copy of the lane id: computed from `lane` they
        // are loop invariants, hipcc keeps them in registers across the chunk loop (which has none to spare) and
        // spills them -- and every scratch reload is followed by vmcnt(0), a full drain of the loads and stores in flight.
        int le = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // lane id, rematerialised (2 VALU)
        asm volatile("" : "+v"(le));
        const int eq = le >> 4, ety = (le >> 2) & 3, etx = le & 3;
        float* sw = stg + wave * (4 * T::SPLANE);
        const int s_own = eq * T::SPLANE + (4 * ety) * T::SROW + 4 * etx;       // this lane's 4x4 patch (row a: + a*SROW)
        const int s_lin = (le >> 2) * T::SROW + (le & 3) * 4;                   // plane piece `lane` (row lane>>2, quarter lane&3)
        const f32x4 bv = bload(r_bias, (unsigned)(eq * 16), (unsigned)(cot * 64));
        const unsigned ep_vo = le < 60 ? le * 16 : 0x80000000u;                 // piece `lane` of a plane; lanes 60..63 out of range
        const unsigned st_out_vo = ((QUARTER ? bsel == 0 : ph == 0) || two) ? ep_vo : 0x80000000u;   // the missing second board of an odd batch: stores dropped
        auto plane_so = [&](int r, int qp) {                                    // plane q' of step r
            return (unsigned)__builtin_amdgcn_readfirstlane(bd_own * T::C + cot * 16 + qp * 4 + r) * plane_b;
        };
        // rows of this wave -> (P0, P1, P2) for TWO channels at once: components r0, r0 + 1 of an accumulator are
        // neighbouring registers, so the whole k-direction transform (the formulas above) and the row sums run as
        // packed two-wide operations -- half the VALU instructions of the epilogue's biggest part
        auto partial2 = [&](const f32x4* a, auto R0, f32x2 (*p)[4]) {
            constexpr int r0 = decltype(R0)::value;
            f32x2 hh[3][4];
#pragma unroll
            for (int ii = 0; ii < 3; ii++) {
                f32x2 m[6];
#pragma unroll
                for (int k = 0; k < 6; k++) m[k] = __builtin_shufflevector(a[ii * 6 + k], a[ii * 6 + k], r0, r0 + 1);
                const f32x2 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
                hh[ii][0] = (m[0] + s12) + s34;
                hh[ii][1] = fma2(2.f, d34, d12);
                hh[ii][2] = fma2(4.f, s34, s12);
                hh[ii][3] = fma2(8.f, d34, d12) + m[5];
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (ph == 0) {
                    const f32x2 s12 = hh[1][e] + hh[2][e];
                    p[0][e] = hh[0][e] + s12;
                    p[1][e] = hh[1][e] - hh[2][e];
                    p[2][e] = s12;
                } else {
                    p[0][e] = hh[0][e] + hh[1][e];
                    p[1][e] = hh[0][e] - hh[1][e];
                    p[2][e] = hh[2][e];
                }
            }
        };
        {
            constexpr int own = ph;             // static register indices
            f32x4 winb[2][4];                   // residual planes in flight: step r in winb[r & 1]
            // QUARTER: sub-step r of a (tile, board) is finished by the row-half wave ph = (r >= 2); the other one only sends
            auto finisher = [&](int r) { return !QUARTER || ((r >= 2) == (ph == 1)); };
            auto resid_load = [&](int r) {
#pragma unroll
                for (int qp = 0; qp < 4; qp++) winb[r & 1][qp] = bload(r_res, finisher(r) ? ep_vo : 0x80000000u, plane_so(r, qp));
            };
            // X (one float per lane and value: [12][64]; any register can be stored, no packing moves)
            float* xs = xb + wave * T::XW + le;
            const float* xr = xb + (wave ^ 4) * T::XW + le;
            auto step = [&](int r, int sub, f32x2 (*qs)[4], f32x2 (*qo)[4]) {
                APZ3_STAMP(3)
                __syncthreads();                // r = 0: every wave's MFMAs over V[1] are done; r > 0: X of step r-1 consumed
                APZ3_STAMP(4)
#pragma unroll
                for (int v = 0; v < 3; v++)
#pragma unroll
                    for (int e = 0; e < 4; e++) xs[(v * 4 + e) * 64] = qs[v][e][sub];
                // Residual planes: steps 0 and 1 only now -- at the top of the epilogue all 144 accumulators are live
                // and registers in flight would spill -- steps 2, 3 one step ahead.  Always BEFORE the step's stores:
                // vmcnt counts in issue order, so the wait for these loads leaves the stores in flight.
#if !APZ3_EARLY_RESID
                if (RESID && r == 0) {
                    resid_load(0);
                    resid_load(1);
                }
#endif
                if (RESID && r >= 1 && r + 1 < 4) resid_load(r + 1);
                f32x4 (&win)[4] = winb[r & 1];
                APZ3_STAMP(3)
                __syncthreads();                // X of step r complete
                APZ3_STAMP(4)
                float px[3][4];
#pragma unroll
                for (int v = 0; v < 3; v++)
#pragma unroll
                    for (int e = 0; e < 4; e++) px[v][e] = xr[(v * 4 + e) * 64];
                f32x4 w4[4];
                if (RESID) {                    // plane pieces -> staging -> this lane's 4x4 patch
#pragma unroll
                    for (int qp = 0; qp < 4; qp++) *reinterpret_cast<f32x4*>(sw + qp * T::SPLANE + s_lin) = win[qp];   // (lanes 60..63: row 15, unused)
                    APZ3_FENCE();
#pragma unroll
                    for (int a = 0; a < 4; a++) w4[a] = *reinterpret_cast<const f32x4*>(sw + s_own + a * T::SROW);
                    APZ3_FENCE();
#ifdef APZ_WINO3_STAMPS
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                    APZ3_STAMP(5)
                }
                const float bvr = bv[r];
                f32x4 y[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float lo0 = own == 0 ? qo[0][e][sub] : px[0][e], lo1 = own == 0 ? qo[1][e][sub] : px[1][e],
                                lo2 = own == 0 ? qo[2][e][sub] : px[2][e];
                    const float hi0 = own == 0 ? px[0][e] : qo[0][e][sub], hi1 = own == 0 ? px[1][e] : qo[1][e][sub],
                                hi2 = own == 0 ? px[2][e] : qo[2][e][sub];
                    y[0][e] = lo0 + hi0;
                    y[1][e] = __builtin_fmaf(2.f, hi1, lo1);
                    y[2][e] = __builtin_fmaf(4.f, hi0, lo2);
                    y[3][e] = lo1 + __builtin_fmaf(8.f, hi1, hi2);
                }
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    f32x4 v = y[a] + bvr;
                    if (RESID) v += w4[a];
#pragma unroll
                    for (int e = 0; e < 4; e++) y[a][e] = RELU ? fmaxf(v[e], 0.f) : v[e];
                    if (etx == 3) y[a][3] = 0.f;   // column 15 is the halo column of the rows16 layout
                    *reinterpret_cast<f32x4*>(sw + s_own + a * T::SROW) = y[a];   // (row 15 of tile row 3 lands in the pad row)
                }
                APZ3_FENCE();
                f32x4 pv[4];
#pragma unroll
                for (int qp = 0; qp < 4; qp++) pv[qp] = *reinterpret_cast<const f32x4*>(sw + qp * T::SPLANE + s_lin);
                APZ3_FENCE();
                APZ3_STAMP(3)
#pragma unroll
                for (int qp = 0; qp < 4; qp++)
                    bstore(r_out, finisher(r) ? st_out_vo : 0x80000000u, plane_so(r, qp), pv[qp]);
                if constexpr (STATS) {
                    float r1[4], r2[4];
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        r1[a] = (y[a][0] + y[a][1]) + (y[a][2] + y[a][3]);
                        r2[a] = (y[a][0] * y[a][0] + y[a][1] * y[a][1]) + (y[a][2] * y[a][2] + y[a][3] * y[a][3]);
                    }
                    if (ety == 3) r1[3] = 0.f, r2[3] = 0.f;                      // board row 15 does not exist
                    const double d1 = (double)wino3_row16_sum((r1[0] + r1[1]) + (r1[2] + r1[3]));
                    const double d2 = (double)wino3_row16_sum((r2[0] + r2[1]) + (r2[2] + r2[3]));
                    const bool mine = finisher(r) && ((QUARTER ? bsel == 0 : ph == 0) || two) && (le & 15) == 0;
                    const unsigned so = (unsigned)((cot * 16 + eq * 4 + r) * n + bd_own) * 16u;
                    typedef double f64x2 __attribute__((ext_vector_type(2)));
                    const f32x4 pk = __builtin_bit_cast(f32x4, f64x2{d1, d2});
                    bstore(r_st, mine ? so : 0x80000000u, 0u, pk);
                }
            };
#if APZ3_EARLY_RESID
            if constexpr (QUARTER) {
                // one board per wave: its row partial is what it sends AND what it combines with the partner's
                f32x2 q01[3][4], q23[3][4];
                partial2(acc[0], std::integral_constant<int, 0>{}, q01);
                if (RESID) {
                    resid_load(0);
                    resid_load(1);
                }
                step(0, 0, q01, q01);
                step(1, 1, q01, q01);
                partial2(acc[0], std::integral_constant<int, 2>{}, q23);
                step(2, 0, q23, q23);
                step(3, 1, q23, q23);
            } else {
                // Both partials of the OTHER board first: its 72 accumulators die, and the registers they free take the
                // residual planes of steps 0 and 1 -- requested here, a transform pass ahead of their use, instead of
                // inside step 0 where the epilogue then waited out their latency.
                f32x2 qsa[3][4], qsb[3][4], qo[3][4];
                partial2(acc[1 - own], std::integral_constant<int, 0>{}, qsa);
                partial2(acc[1 - own], std::integral_constant<int, 2>{}, qsb);
                if (RESID) {
                    resid_load(0);
                    resid_load(1);
                }
                partial2(acc[own], std::integral_constant<int, 0>{}, qo);
                step(0, 0, qsa, qo);
                step(1, 1, qsa, qo);
                partial2(acc[own], std::integral_constant<int, 2>{}, qo);
                step(2, 0, qsb, qo);
                step(3, 1, qsb, qo);
            }
#else
            {
                f32x2 qs[3][4], qo[3][4];
                partial2(acc[1 - own], std::integral_constant<int, 0>{}, qs);
                partial2(acc[own], std::integral_constant<int, 0>{}, qo);
                step(0, 0, qs, qo);
                step(1, 1, qs, qo);
            }
            {
                f32x2 qs[3][4], qo[3][4];
                partial2(acc[1 - own], std::integral_constant<int, 2>{}, qs);
                partial2(acc[own], std::integral_constant<int, 2>{}, qo);
                step(2, 0, qs, qo);
                step(3, 1, qs, qo);
            }
#endif
        }
        APZ3_STAMP(6)
#if APZ3_EARLY_BARRIER
        // the next item's first barrier comes only after its slot 16, and its transform slices write V[1] (= X) from
        // slot 10 on: every wave must have read its last X values before anybody goes on
        __syncthreads();
        APZ3_STAMP(4)
#endif
    }
    };
    if ((wave >> 2) == 0)
        run(std::integral_constant<int, 0>{});
    else
        run(std::integral_constant<int, 1>{});
#ifdef APZ_WINO3_STAMPS
    st_acc[7] = __builtin_readcyclecounter() - st_t0;
    if (lane == 0 && blockIdx.x < 4)
        for (int i = 0; i < 8; i++) apz_wino3_stamps[(blockIdx.x * 8 + wave) * 8 + i] = st_acc[i];
#endif
}

// Launch grid for n boards on `num_cu` CUs (and, where the caller can launch the QUARTER variant, whether to).  At least as many board pairs as CUs: one workgroup per CU (duo mode when
// num_cu is a multiple of 16, every workgroup takes >= 1 pair x one channel half).  Fewer pairs (training at the
// reference's batch_size 128, small inference batches): TWO workgroups per pair -- the channel halves of a pair run on
// two CUs at once instead of one after the other on one -- rounded up to whole groups of 16 blocks, which is what makes
// the kernel pair blocks b and b + 8 up (duos beyond the last pair have no work and return).
inline int wino3_grid(int n, int num_cu, bool* quarter = nullptr) {
    const int npairs = (n + 1) >> 1;
    if (quarter) *quarter = false;
    if (npairs >= num_cu) return num_cu;
    if (quarter && 4 * npairs <= num_cu) {      // at most a quarter of the CUs would get a pair: FOUR workgroups per pair
        *quarter = true;                        // (QUARTER items of 32 output channels), whole groups of 32 blocks
        const int g4 = (4 * npairs + 31) & ~31;
        if (g4 <= num_cu) return g4;
        *quarter = false;
    }
    const int g = (2 * npairs + 15) & ~15;
    return g <= num_cu ? g : num_cu;
}

}  // namespace apz
