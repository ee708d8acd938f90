// Trunk 3x3 convolution (128 -> 128 channels, 15x15 board) + folded BN + (residual) + ReLU:
// the kernel ~97 % of a leaf evaluation is spent in.  gfx950 only.
//
// Same im2col-free fp32-MFMA direct convolution as conv3x3_mfma.h, restructured so that the
// matrix pipe never waits for staging:
//
//  * activations between trunk layers live in HBM as [n][128][15][16] floats ("rows16": every
//    board row padded to 16 floats, the 16th always 0).  A plane is 960 B, rows are 64-B
//    aligned, so the epilogue moves 16 B per lane and staging is a linear copy.
//  * LDS holds a RING of four 32-channel chunks (4 x 34 KB).  A chunk is filled by LDS-DMA
//    (global_load_lds_dwordx4, no VGPRs): one 960-B plane per wave-instruction, each wave
//    issues ONE plane per ci4 iteration (8 iterations x 4 waves = the 32 planes of the chunk
//    three chunks ahead), i.e. ~4 us before the next counted wait -- the DMA is invisible.
//    In LDS the planes are 272 floats apart (== 16 mod 32: the four ci lanes-groups of a
//    ds_read_b32 fall on disjoint bank halves) and the 32 floats between two planes stay zero:
//    they are the bottom halo row of one plane and the top halo row of the next; column 15
//    of every row is the right halo of that row and the left halo of the next.
//  * MFMA operands swapped w.r.t. conv3x3_mfma.h: D[16 px][16 co] += A[16 px][4 ci] * B[4 ci][16 co],
//    so lane l holds 4 consecutive pixels (x = 4*(l>>4) .. +3) of channel co = l&15:
//    one aligned global_store_dwordx4 (and one dwordx4 residual load) per accumulator tile.
//  * persistent workgroups: the chunk stream runs across board boundaries, so the next board's
//    first chunks land while the current board's epilogue stores drain.
//
// One barrier per chunk (2160 MFMAs per wave between barriers).
#pragma once
#include <hip/hip_runtime.h>

#include "conv3x3_mfma.h"

namespace apz {

struct Trunk15 {
    static constexpr int C = 128, H = 15, W = 15;
    static constexpr int GROW = 16;              // global / LDS row stride (floats)
    static constexpr int GPLANE = H * GROW;      // 240 floats = 960 B per plane in HBM
    static constexpr int LPS = 272;              // LDS plane stride, == 16 (mod 32)
    static constexpr int CH = 32;                // channels per ring slot
    static constexpr int NSLOT = 4;
    static constexpr int SLOT = CH * LPS;        // floats per slot
    static constexpr int FRONT = 32;             // zero floats in front of slot 0 (row -1 of plane 0)
    static constexpr int LDS_FLOATS = FRONT + NSLOT * SLOT + 32;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "ring must fit the 160 KiB LDS");
};

__device__ __forceinline__ void glds16(const float* gsrc, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// in / resid / out: [n][128][15][16] rows16 layout; wpk: [8][32][64][12] (cot, ci4, lane, tap: 9 taps + 3 pad)
// so that a lane fetches its nine weights of a ci4 step with three 16-byte loads
// NW = waves per workgroup: 4 (one per SIMD, 2 channel tiles each) or 8 (two per SIMD, one channel
// tile each: the second wave's MFMAs fill the first one's wait slots at iteration boundaries).
template <bool RESID, int NW>
__global__ __launch_bounds__(64 * NW) void trunk15_ring_kernel(const float* __restrict__ in,
                                                               const float* __restrict__ wpk,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ resid,
                                                               float* __restrict__ out, int n) {
    using T = Trunk15;
    constexpr int NT = 64 * NW;
    constexpr int CT = 8 / NW;            // 16-channel tiles per wave
    constexpr int PPW = T::CH / NW;       // DMA planes per wave per chunk (8 or 4)
    constexpr int RES_CHUNK = 1;          // chunk during which the residual burst is issued
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ring = lds + T::FRONT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, j = lane & 15;

    for (int i = tid * 4; i < T::LDS_FLOATS; i += NT * 4) *reinterpret_cast<f32x4*>(&lds[i]) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int nb = (n - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // boards of this workgroup
    const int total_chunks = nb * 4;

    // DMA plane number `pl` (0..PPW-1) of this wave for global chunk g: plane p = NW*pl + wave
    auto issue_plane = [&](int g, int pl) {
        if (g < total_chunks) {
            const int board = (int)blockIdx.x + (g >> 2) * (int)gridDim.x;
            const int p = pl * NW + wave;
            const int c = (g & 3) * T::CH + p;
            const float* src = in + ((size_t)board * T::C + c) * T::GPLANE + lane * 4;
            float* dst = ring + (g & 3) * T::SLOT + p * T::LPS;
            if (lane < 60) glds16(src, dst);
        }
    };

    for (int g = 0; g < 3; g++)
#pragma unroll
        for (int pl = 0; pl < PPW; pl++) issue_plane(g, pl);
    __syncthreads();   // emits vmcnt(0): chunks 0..2 of the first board have landed for every wave

    const f32x4* wbase = reinterpret_cast<const f32x4*>(wpk) + ((size_t)(wave * CT) * 32 * 64 + lane) * 3;
    float a_cur[CT][12], a_nxt[CT][12];
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int v = 0; v < 3; v++) {
            const f32x4 w4 = wbase[((size_t)ct * 32 * 64) * 3 + v];
#pragma unroll
            for (int u = 0; u < 4; u++) a_cur[ct][v * 4 + u] = w4[u];
        }

    const int lane_off = q * T::LPS + j - 17;   // (row f-1, col j+kx-1) = f*16 + kx + (j - 17)

    for (int bi = 0; bi < nb; bi++) {
        const int board = (int)blockIdx.x + bi * (int)gridDim.x;
        f32x4 acc[CT][15];
        f32x4 res[CT][15];               // residual, prefetched during the last chunk
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int t = 0; t < 15; t++) acc[ct][t] = f32x4{0.f, 0.f, 0.f, 0.f};

        // Software pipeline over the 3 kx phases of every ci4 step: the 17 row fragments of phase
        // p+1 are read from LDS while the 90 MFMAs of phase p issue, so the matrix pipe never sits
        // behind an exposed ds_read latency (one wave per SIMD: nobody else would cover it).
        // The c4l loop is unrolled by two so the ping-pong fragment arrays keep static indices.
#define APZ_LOAD_ROWS(dst, ptr, kx_)                                  \
    _Pragma("unroll") for (int f = 0; f < 17; f++) dst[f] = (ptr)[f * 16 + (kx_)];
#define APZ_MFMA_PHASE(rows, kx_)                                                                      \
    _Pragma("unroll") for (int ky = 0; ky < 3; ky++) _Pragma("unroll") for (int t = 0; t < 15; t++)    \
        _Pragma("unroll") for (int ct = 0; ct < CT; ct++) acc[ct][t] = __builtin_amdgcn_mfma_f32_16x16x4f32( \
            rows[t + ky], a_cur[ct][ky * 3 + (kx_)], acc[ct][t], 0, 0, 0);
        for (int chunk = 0; chunk < 4; chunk++) {
            const int g = bi * 4 + chunk;
            const float* sptr = ring + (g & 3) * T::SLOT + lane_off;
            // first fragments of the NEXT chunk (already landed: its slot was drained two barriers ago)
            const float* snext = ring + ((g + 1) & 3) * T::SLOT + lane_off;
            float rA[17], rB[17];
            APZ_LOAD_ROWS(rA, sptr, 0)
            for (int c4l = 0; c4l < 8; c4l++) {
                const float* bptr = sptr + c4l * 4 * T::LPS;
                const float* bnext = (c4l < 7) ? bptr + 4 * T::LPS : snext;
                // ---- phase kx = 0 (rows in rA), prefetch kx = 1 into rB
                APZ_LOAD_ROWS(rB, bptr, 1)
                APZ_MFMA_PHASE(rA, 0)
                {
                    // Issue this iteration's DMA plane and the next weights AFTER the first third of
                    // the MFMAs: hipcc waits vmcnt(0) at the first use of an ordinary load while an
                    // LDS-DMA is in flight, so that wait must find loads that are two thirds of an
                    // iteration (~5.7k cycles) old, not fresh ones.
                    __builtin_amdgcn_sched_barrier(0);
                    if (NW == 4 || c4l < PPW) issue_plane(g + 3, c4l);   // lands ~3 chunks before it is read
                    const int c4n = (chunk * 8 + c4l + 1) & 31;          // wraps to the next board's first step
#pragma unroll
                    for (int ct = 0; ct < CT; ct++)
#pragma unroll
                        for (int v = 0; v < 3; v++) {
                            const f32x4 w4 = wbase[(((size_t)ct * 32 + c4n) * 64) * 3 + v];
#pragma unroll
                            for (int u = 0; u < 4; u++) a_nxt[ct][v * 4 + u] = w4[u];
                        }
                    if (RESID && chunk == RES_CHUNK && c4l == 0) {
                        // all residual tiles of this board in ONE burst, chunks ahead of the epilogue:
                        // the compiler's per-iteration vmcnt(0) (LDS-DMA in flight) then stalls at most
                        // once per board on them instead of a little at every iteration top
#pragma unroll
                        for (int ct = 0; ct < CT; ct++) {
                            const size_t pb = ((size_t)board * T::C + (wave * CT + ct) * 16 + j) * T::GPLANE + q * 4;
#pragma unroll
                            for (int t = 0; t < 15; t++) res[ct][t] = *reinterpret_cast<const f32x4*>(resid + pb + t * 16);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- phase kx = 1 (rB), prefetch kx = 2 into rA
                APZ_LOAD_ROWS(rA, bptr, 2)
                APZ_MFMA_PHASE(rB, 1)
                // ---- phase kx = 2 (rA), prefetch the next step's kx = 0 into rB, then rotate
                APZ_LOAD_ROWS(rB, bnext, 0)
                APZ_MFMA_PHASE(rA, 2)
#pragma unroll
                for (int f = 0; f < 17; f++) rA[f] = rB[f];
#pragma unroll
                for (int ct = 0; ct < CT; ct++)
#pragma unroll
                    for (int tap = 0; tap < 9; tap++) a_cur[ct][tap] = a_nxt[ct][tap];
            }
            __syncthreads();   // slot g&3 fully consumed by all waves; pending DMA drained (vmcnt(0))
        }
#undef APZ_LOAD_ROWS
#undef APZ_MFMA_PHASE

        // ---- epilogue: lane holds pixels x = 4q..4q+3 of row t for channel co
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const int co = (wave * CT + ct) * 16 + j;
            const float bv = bias[co];
            const size_t pbase = ((size_t)board * T::C + co) * T::GPLANE + q * 4;
#pragma unroll
            for (int t = 0; t < 15; t++) {
                f32x4 v = acc[ct][t];
                if (RESID) v += res[ct][t];
                v[0] = fmaxf(v[0] + bv, 0.f);
                v[1] = fmaxf(v[1] + bv, 0.f);
                v[2] = fmaxf(v[2] + bv, 0.f);
                v[3] = (q == 3) ? 0.f : fmaxf(v[3] + bv, 0.f);   // column 15 is the halo: keep it zero
                *reinterpret_cast<f32x4*>(out + pbase + t * 16) = v;
            }
        }
    }
}

}  // namespace apz

namespace apz {

// Stem 3x3 convolution (C_in = 4 or 9 planes -> 128 channels, 15x15) + folded BN + ReLU.
// Output-write bound at C_in = 4 (17.5 FLOP/B): the whole job is to keep ~1 GB of stores per
// 8192 boards streaming while the (small) contraction runs.  Same MFMA tile mapping and rows16
// output as trunk15_ring_kernel, but:
//   * the 12 (or 4) input planes of a board are 13 KB in LDS, so TWO workgroups are resident per
//     CU (launch bound 2 waves/SIMD): one computes while the other drains its 123 KB of stores;
//   * the packed weights (55 KB for the whole layer) are loaded into registers ONCE per
//     persistent workgroup: the board loop issues no weight loads at all;
//   * input is the dense NCHW [n][C_in][15][15] planes buffer of the C ABI (the external
//     contract), re-laid-out while staging.
// CODES: `in` is not planes but the self-play path's position codes ([n][code_stride] bytes, heads.h
//     encode_planes_kernel's input): the board's planes (Board.current_state, game.py:68-115, incl. the vertical flip)
//     are built straight into the LDS tile -- the separate encode kernel, its 4 MB of planes and a launch disappear
//     from the forward.
// wpk: [8][C4][9][64]; out: rows16 [n][128][15][16].
// C4 == 1 (the HBM-bound shape): a wave accumulates ONE of its two 16-channel tiles at a time, so the first tile's
// stores are in flight while the second tile's MFMAs run (within the wave, on top of the overlap between the two
// resident workgroups).  Measured with tools/stem_bench.hip at 8192 boards (mean of 10 launches): both tiles at once
// 214-218 us, one at a time 198-205 us; staging 8 instead of 16 channels per store burst (which would fit a third
// workgroup per CU) 227-254 us -- shorter bursts cost more than the extra workgroup gives; grids that do not divide
// the batch (768) lose 20 % to the tail.
#ifndef APZ_STEM_OCC        // experiment switches of tools/stem_bench.hip (workgroups per CU, channels staged at a time,
#define APZ_STEM_OCC 2     // channel tiles accumulated at a time) for the C_in = 4 shape
#endif
#ifndef APZ_STEM_SCH
#define APZ_STEM_SCH 16
#endif
#ifndef APZ_STEM_CTB
#define APZ_STEM_CTB 1
#endif
#ifndef APZ_STEM_CTB9      // the same for the C_in = 9 shape
#define APZ_STEM_CTB9 1
#endif
template <int C4, int CIN, bool CODES = false>
__global__ __launch_bounds__(256, (C4 == 1 ? APZ_STEM_OCC : 2)) void stem15_kernel(const float* __restrict__ in, const float* __restrict__ wpk,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int n, int cin, int code_stride = 0) {
    using T = Trunk15;
    constexpr int NPL = 4 * C4;
    constexpr int LDSF = T::FRONT + NPL * T::LPS + 32;
    constexpr int OST = 244;                       // staged output row stride: 16 lanes x 16 B hit 64 distinct banks
    constexpr int SCH = (C4 == 1) ? APZ_STEM_SCH : 16;   // channels staged at a time
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [LDSF tile][4 waves x 16 x OST staging]
    float* tile = lds + T::FRONT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, j = lane & 15;

    for (int i = tid; i < LDSF; i += 256) lds[i] = 0.f;

    float a[2][C4][9];
    const float* wbase = wpk + ((size_t)(wave * 2) * C4 * 9) * 64 + lane;
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int c4 = 0; c4 < C4; c4++)
#pragma unroll
            for (int tap = 0; tap < 9; tap++) a[ct][c4][tap] = wbase[(((size_t)ct * C4 + c4) * 9 + tap) * 64];
    float bv[2];
#pragma unroll
    for (int ct = 0; ct < 2; ct++) bv[ct] = bias[(wave * 2 + ct) * 16 + j];
    // C_in = 9: channels 0..7 are two full k-groups of four; the ninth channel alone would fill a third group to a
    // quarter (9 of 27 k-steps multiplying zeros).  Its nine taps are contracted as k instead: three k-steps in which
    // lane group q holds tap 4 s + q (taps 9..11: zero weight) -- 21 k-steps per output row instead of 27.
    constexpr bool NINTH = (CIN == 9);
    constexpr int NC4 = NINTH ? 2 : C4;            // k-groups walked tap by tap with the sliding row window
    float a9[2][3];
    int b9[3];
#pragma unroll
    for (int s9 = 0; s9 < 3; s9++) {
        const int tap = 4 * s9 + q, tp = tap < 9 ? tap : 0;
        b9[s9] = 8 * T::LPS + (tp / 3) * 16 + (tp % 3) + j - 17;
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
            a9[ct][s9] = (NINTH && tap < 9) ? wpk[((((size_t)(wave * 2 + ct)) * C4 + 2) * 9 + tp) * 64 + j] : 0.f;
    }

    const int lane_off = q * T::LPS + j - 17;
    const int total = cin * 225;
    constexpr int NPF = (CIN * 225 + 255) / 256;   // input floats per thread per board (4 or 8)
    constexpr bool PREFETCH = true;                // (one tile at a time leaves the registers for it at C_in = 9 too)
    // The stores of board b must drain WHILE board b+1 computes.  vmcnt retires in order, so
    // the next board's planes are loaded into registers BEFORE this board's stores are issued
    // (their wait then never covers a store), and the barriers protecting the LDS tile are raw
    // s_barrier + lgkmcnt(0): __syncthreads() would add vmcnt(0) and stall on the store acks.
    float pf[CODES ? 1 : NPF];
    unsigned code = 0, colour = 0;                 // CODES: this thread's cell (tid < 225) and the board's colour byte
    auto prefetch = [&](int b) {
        if (CODES) {
            const unsigned char* cb = reinterpret_cast<const unsigned char*>(in) + (size_t)b * code_stride;
            code = (b < n && tid < 225) ? cb[tid] : 0;
            colour = b < n ? cb[225] : 0;
            return;
        }
        const float* src = in + (size_t)b * total;
#pragma unroll
        for (int u = 0; u < NPF; u++) {
            const int idx = tid + u * 256;
            pf[u] = (b < n && idx < total) ? src[idx] : 0.f;
        }
    };
    auto lds_barrier = [&]() {
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0) only
        __builtin_amdgcn_s_barrier();
    };
    __syncthreads();                          // zero fill done
    prefetch(blockIdx.x);
    for (int b = blockIdx.x; b < n; b += gridDim.x) {
        if (!PREFETCH && b != (int)blockIdx.x) prefetch(b);
        if (CODES) {
            if (tid < 225) {                      // cell m = tid of the un-flipped board -> row 14 - h of the planes
                const int h = tid / 15, w = tid - h * 15;
                float* cell = tile + (14 - h) * 16 + w;
                const bool opp = code >= 5;
                const int age = (code - 1) & 3;
                const float col = colour ? 1.f : 0.f;
                if (CIN == 9) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float on = (code && k <= age) ? 1.f : 0.f;
                        cell[(6 - 2 * k) * T::LPS] = opp ? 0.f : on;
                        cell[(7 - 2 * k) * T::LPS] = opp ? on : 0.f;
                    }
                    cell[8 * T::LPS] = col;
                } else {
                    cell[0 * T::LPS] = (code && !opp) ? 1.f : 0.f;
                    cell[1 * T::LPS] = (code && opp) ? 1.f : 0.f;
                    cell[2 * T::LPS] = (code && age == 0) ? 1.f : 0.f;
                    cell[3 * T::LPS] = col;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < NPF; u++) {
                const int idx = tid + u * 256;
                if (idx < total) {
                    const int c = idx / 225, rem = idx - c * 225;
                    const int y = rem / 15, x = rem - y * 15;
                    tile[c * T::LPS + y * 16 + x] = pf[u];
                }
            }
        }
        lds_barrier();
        if (PREFETCH) prefetch(b + gridDim.x);  // lands during the MFMAs below

        constexpr int CTB = (C4 == 1) ? APZ_STEM_CTB : APZ_STEM_CTB9;   // channel tiles accumulated at a time
        float* st = lds + LDSF + wave * (SCH * OST);
#pragma unroll
        for (int cb = 0; cb < 2; cb += CTB) {
            f32x4 acc[CTB][15];
#pragma unroll
            for (int ct = 0; ct < CTB; ct++)
#pragma unroll
                for (int t = 0; t < 15; t++) acc[ct][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c4 = 0; c4 < NC4; c4++) {
                const float* bptr = tile + lane_off + c4 * 4 * T::LPS;
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    float r[17];
#pragma unroll
                    for (int f = 0; f < 17; f++) r[f] = bptr[f * 16 + kx];
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int t = 0; t < 15; t++)
#pragma unroll
                            for (int ct = 0; ct < CTB; ct++)
                                acc[ct][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(r[t + ky], a[cb + ct][c4][ky * 3 + kx],
                                                                                  acc[ct][t], 0, 0, 0);
                }
            }
            if (NINTH) {
#pragma unroll
                for (int s9 = 0; s9 < 3; s9++) {
                    float r9[15];
#pragma unroll
                    for (int t = 0; t < 15; t++) r9[t] = tile[b9[s9] + t * 16];
#pragma unroll
                    for (int t = 0; t < 15; t++)
#pragma unroll
                        for (int ct = 0; ct < CTB; ct++)
                            acc[ct][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(r9[t], a9[cb + ct][s9], acc[ct][t], 0, 0, 0);
                }
            }
            if (cb + CTB == 2) lds_barrier();     // every wave is done reading the tile
            // ---- epilogue through LDS: the accumulator layout gives 64-B pieces per (channel, row);
            // written straight to HBM that pattern tops out at ~4.2 TB/s.  Each wave transposes its
            // 16-channel tile in a private LDS staging area and streams whole 960-B planes instead.
#pragma unroll
            for (int ctl = 0; ctl < CTB; ctl++) {
                const int ct = cb + ctl;
#pragma unroll
                for (int half = 0; half < 16 / SCH; half++) {
                    if (SCH == 16 || (j >> 3) == half) {
#pragma unroll
                        for (int t = 0; t < 15; t++) {
                            f32x4 v = acc[ctl][t];
                            v[0] = fmaxf(v[0] + bv[ct], 0.f);
                            v[1] = fmaxf(v[1] + bv[ct], 0.f);
                            v[2] = fmaxf(v[2] + bv[ct], 0.f);
                            v[3] = (q == 3) ? 0.f : fmaxf(v[3] + bv[ct], 0.f);
                            *reinterpret_cast<f32x4*>(st + (j & (SCH - 1)) * OST + t * 16 + q * 4) = v;
                        }
                    }
                    // lanes exchange data through the wave-private staging area: keep the compiler from moving a
                    // lane's reads above its own (different-address) writes, and the next tile's writes above these reads
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    float* dst = out + ((size_t)b * T::C + (wave * 2 + ct) * 16 + half * SCH) * T::GPLANE + lane * 4;
                    if (lane < 60) {
#pragma unroll
                        for (int c = 0; c < SCH; c++)
                            __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(st + c * OST + lane * 4),
                                                        reinterpret_cast<f32x4*>(dst + c * T::GPLANE));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
        }
    }
}

template <int C4>
constexpr int stem15_lds_bytes() {
    return (Trunk15::FRONT + 4 * C4 * Trunk15::LPS + 32 + 4 * (C4 == 1 ? APZ_STEM_SCH : 16) * 244) * (int)sizeof(float);
}

}  // namespace apz
