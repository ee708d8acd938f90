// Trunk 3x3 convolution (128 -> 128 channels, 15x15 board) + folded BN + (residual) + ReLU as a
// fused F(4x4,3x3) Winograd convolution on the fp32 matrix cores -- two boards per workgroup.
// gfx950 only.  Successor of trunk15_wino.h (same math, same layouts in HBM).
//
// What bounded the one-board kernel (measured, tools/wino_ablate.hip + tools/mfma_valu_probe.hip):
//   * a CU pulls ~70 GB/s from L2; the transformed weights U = G g G^T are 2.36 MB per layer and
//     every board needs all of them once -> 34 us of weight stream per board against 33 us of
//     MFMA time, and the two did not overlap well (17 of 66 us);
//   * VALU instructions do NOT hide under v_mfma_f32_16x16x4_f32 on this chip: each one costs
//     ~4 SIMD cycles on top of the MFMA's 32 (expanding U from the raw taps in registers,
//     ~3 VALU per MFMA, was slower than streaming it).
// So: share every weight fragment between TWO boards.  A board's 36 x 128 x 16 accumulators are
// 288 KB -- two boards do not fit the 512 KB register file -- hence two passes over the input
// channels: pass 0 accumulates the transformed rows 0..2 (18 of the 36 positions) of both
// boards, pass 1 the rows 3..5.  A wave (8 per workgroup, two per SIMD; wave w = output
// channels 16w..16w+15) holds 2 boards x 18 positions x 4 = 144 accumulator registers and every
// A fragment (weights) feeds two MFMAs.  The output transform is linear in the rows, so pass 0
// writes its partial 4x4 outputs to `out` and pass 1 adds them back (same lane, same address).
//
// Per pass the 128 input channels stream through in 16 chunks of 8:
//   global (rows16 planes) --regs--> raw LDS tile (zero halo) --B^T d B (3 rows)--> V LDS --MFMA--> acc
// one barrier per chunk:  iteration g: [barrier] raw(g+2) regs -> LDS; issue loads raw(g+3);
//                         transform raw(g+1) -> V[(g+1)&1] (one wave of each SIMD, alternating);
//                         MFMA over V[g&1] (all waves).
//
// Layouts.  in / resid / out: rows16 [n][128][15][16] (col 15 == 0), as trunk15_ring.h.
// upk: [cot 8][pass 2][c4 32][lane 64][20]: lane (q = lane>>4, j = lane&15) holds
//      U[row 3*pass + ii][k] at index 6*ii + k (18 values + 2 pad) of co = cot*16 + j, ci = c4*4 + q.
// raw (LDS): [2 boards x 8 channels] planes, row stride 20 floats (cols 16..19 zero: right halo,
//      and col -1 of the next row), plane stride 340 (rows 15, 16 zero; row 16 == row -1 of the next).
// V (LDS): [board 2][ch 8][tile 16][row 3][8] floats, (ch, tile) stride 28: a row's 6 values are one
//      ds_read_b128 + one ds_read_b64, conflict-free over the 64 lanes.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "conv3x3_mfma.h"

namespace apz {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Wino2 {
    static constexpr int C = 128, CK = 8, NCHUNK = C / CK, ITERS = 2 * NCHUNK;   // iterations per board pair
    static constexpr int GPLANE = 240;                 // floats per plane in HBM (15 rows x 16)
    static constexpr int RROW = 20, RPS = 17 * RROW;   // LDS row / plane stride
    static constexpr int RFRONT = 24;
    static constexpr int RAW_FLOATS = RFRONT + 2 * CK * RPS;          // 5464
    static constexpr int VTS = 28;                     // floats per (board, channel, tile): 3 rows x 8, + 4 (bank spread)
    static constexpr int V_FLOATS = 2 * CK * 16 * VTS;                // 7168
    static constexpr int SROW = 20, SPLANE = 16 * SROW;               // epilogue staging: 16 rows x 20 floats per plane
    static constexpr int STAGE_FLOATS = 8 * 4 * SPLANE;               // 8 waves x 4 planes (10240 floats = 40 KiB)
    static constexpr int LDS_FLOATS = 2 * RAW_FLOATS + 2 * V_FLOATS + STAGE_FLOATS;   // 35504 floats = 138.7 KiB
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int UROW = 20;                    // floats per lane and k-step in upk
    static constexpr size_t UPK_FLOATS = (size_t)8 * 2 * 32 * 64 * UROW;   // per layer (2.6 MB)
};

__device__ __forceinline__ f32x2 fma2(const float a, const f32x2 b, const f32x2 c) {   // a*b + c (v_pk_fma_f32)
    return __builtin_elementwise_fma(f32x2{a, a}, b, c);
}

// B^T x, packed over two columns at a time.  LO: outputs 0..2 (inputs x0..x4), else 3..5 (x1..x5).
template <bool LO>
__device__ __forceinline__ void wino2_bt3(const f32x2 x0, const f32x2 x1, const f32x2 x2, const f32x2 x3, const f32x2 x4,
                                          const f32x2 x5, f32x2* y) {
    if (LO) {
        const f32x2 a = fma2(-4.f, x2, x4), b = fma2(-4.f, x1, x3);
        y[0] = fma2(4.f, x0, fma2(-5.f, x2, x4));
        y[1] = a + b;
        y[2] = a - b;
    } else {
        const f32x2 c = x4 - x2, d = x3 - x1;
        y[0] = fma2(2.f, d, c);
        y[1] = fma2(-2.f, d, c);
        y[2] = fma2(4.f, x1, fma2(-5.f, x3, x5));
    }
}

// full 1-D B^T over one row of six values held as three column pairs (x01, x23, x45)
__device__ __forceinline__ void wino2_bt6(const f32x2 x01, const f32x2 x23, const f32x2 x45, float* y) {
    const float x0 = x01[0], x1 = x01[1], x2 = x23[0], x3 = x23[1], x4 = x45[0], x5 = x45[1];
    const float a = __builtin_fmaf(-4.f, x2, x4), b = __builtin_fmaf(-4.f, x1, x3), c = x4 - x2, d = x3 - x1;
    y[0] = __builtin_fmaf(4.f, x0, __builtin_fmaf(-5.f, x2, x4));
    y[1] = a + b;
    y[2] = a - b;
    y[3] = __builtin_fmaf(2.f, d, c);
    y[4] = __builtin_fmaf(-2.f, d, c);
    y[5] = __builtin_fmaf(4.f, x1, __builtin_fmaf(-5.f, x3, x5));
}

// 1-D output transform A^T m of F(4,3) along k (all six columns present)
__device__ __forceinline__ void wino2_at6(const float m0, const float m1, const float m2, const float m3, const float m4,
                                          const float m5, float* o) {
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    o[0] = (m0 + s12) + s34;
    o[1] = __builtin_fmaf(2.f, d34, d12);
    o[2] = __builtin_fmaf(4.f, s34, s12);
    o[3] = __builtin_fmaf(8.f, d34, d12) + m5;
}

#ifndef APZ_WINO2_GSH
#define APZ_WINO2_GSH 2
#endif

// Lanes of ONE wave exchange data through LDS (write in one layout, read in another).  The LDS executes a
// wave's instructions in order, so no s_barrier is needed -- but the compiler reasons per thread and may move
// a lane's read above its own (provably different-address) write.  This pins the order for the compiler.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// RELU = false: the plain convolution + bias (training graph: forward before BatchNorm, data gradient)
template <bool RESID, bool RELU = true>
__global__ __launch_bounds__(512) void trunk15_wino2_kernel(const float* __restrict__ in, const float* __restrict__ upk,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ resid, float* __restrict__ out,
                                                            int n) {
    using T = Wino2;
#ifdef APZ_WINO_STAMPS
    // cycle accounting per wave (tools/wino_ablate.hip): phases 0 prologue, 1 MFMA block + barrier wait,
    // 2 staging, 3 transform, 4 (unused), 5 epilogue pass 0, 6 epilogue pass 1, 7 total
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_readcyclecounter();
    const unsigned long long st_t0 = st_t;
#define APZ_STAMP(ph)                                            \
    {                                                            \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        st_acc[ph] += now_ - st_t;                               \
        st_t = now_;                                             \
    }
#else
#define APZ_STAMP(ph)
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* rawb = lds;                          // [2][RAW_FLOATS]
    float* vb = lds + 2 * T::RAW_FLOATS;        // [2][V_FLOATS]
    float* stg = lds + 2 * T::RAW_FLOATS + 2 * T::V_FLOATS;   // [8 waves][4 planes][16 rows x 20]: wave-private epilogue staging

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, j = lane & 15;
    const int group = (wave >> APZ_WINO2_GSH) & 1;   // which of the two waves of a SIMD this is

    for (int i = tid * 4; i < 2 * T::RAW_FLOATS; i += 2048) *reinterpret_cast<f32x4*>(&lds[i]) = f32x4{0.f, 0.f, 0.f, 0.f};

    const int npairs = (n + 1) >> 1;
    const int np = (npairs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // board pairs of this workgroup
    const int total_iters = np * T::ITERS;
    if (np == 0) return;                        // (launchers never oversubscribe; uniform, before any barrier)

    // ---- staging roles: 2 boards x 8 planes x 60 pieces of 16 B, 32 threads per plane, two pieces each (four of
    // them twice).  Every thread issues exactly two loads and two LDS
    // stores per iteration, unconditionally -- the compiler can then count vmcnt exactly and the
    // MFMA phase never waits for the (HBM-latency) staging loads that were issued after its weights.
    // thread -> plane tid>>5 (board = tid>>8, channel = (tid>>5)&7), pieces k = tid&31 and k+32 (k+32 >= 60: k again);
    // everything below is a shift or an add of tid, so nothing has to stay in registers across the chunk loop
    const int st_k = tid & 31, st_k2 = (st_k + 32 < 60) ? st_k + 32 : st_k;
    f32x4 rg[2];
    auto raw_fetch = [&](int g) {               // global -> registers (iteration g of this workgroup's stream, clamped)
        g = g < total_iters ? g : total_iters - 1;
        const int bdp = 2 * ((int)blockIdx.x + (g / T::ITERS) * (int)gridDim.x) + (tid >> 8), c = g & (T::NCHUNK - 1);
        const int bd = bdp < n ? bdp : n - 1;
        const float* pl = in + ((size_t)bd * T::C + c * T::CK + ((tid >> 5) & 7)) * T::GPLANE;
        rg[0] = *reinterpret_cast<const f32x4*>(pl + st_k * 4);
        rg[1] = *reinterpret_cast<const f32x4*>(pl + st_k2 * 4);
    };
    auto raw_store = [&](int g) {               // registers -> raw LDS buffer g&1
        float* dst = rawb + (g & 1) * T::RAW_FLOATS + T::RFRONT + (tid >> 5) * T::RPS;
        *reinterpret_cast<f32x4*>(dst + (st_k >> 2) * T::RROW + (st_k & 3) * 4) = rg[0];
        *reinterpret_cast<f32x4*>(dst + (st_k2 >> 2) * T::RROW + (st_k2 & 3) * 4) = rg[1];
    };
    // ---- transform roles (256 threads of one wave group): unit = (board, channel, tile)
    const int unit = (APZ_WINO2_GSH == 2 ? (wave & 3) : (wave >> 1)) * 64 + lane;
    const int tty = (unit >> 2) & 3, ttx = unit & 3;
    const int tr_off = T::RFRONT + (unit >> 4) * T::RPS + (4 * tty - 1) * T::RROW + 4 * ttx;
    auto transform = [&](int g, auto LO) {      // raw[g&1] -> V[g&1], rows 3*pass .. 3*pass+2 (LO: pass 0)
        constexpr bool lo = decltype(LO)::value;
        const float* rp = rawb + (g & 1) * T::RAW_FLOATS + tr_off + (lo ? 0 : T::RROW);   // pass 1 skips patch row 0
        float* vp = vb + (g & 1) * T::V_FLOATS + unit * T::VTS;
        f32x2 x[5][3];                          // patch rows (0..4 or 1..5) x column pairs
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const float cm1 = rp[i * T::RROW - 1];
            const f32x4 c03 = *reinterpret_cast<const f32x4*>(rp + i * T::RROW);
            const float c4 = rp[i * T::RROW + 4];
            x[i][0] = f32x2{cm1, c03[0]};
            x[i][1] = f32x2{c03[1], c03[2]};
            x[i][2] = f32x2{c03[3], c4};
        }
        f32x2 t[3][3];                          // t[cp][ii]: row 3*pass + ii of column pair cp after the row pass
#pragma unroll
        for (int cp = 0; cp < 3; cp++) {
            if (lo)
                wino2_bt3<true>(x[0][cp], x[1][cp], x[2][cp], x[3][cp], x[4][cp], x[4][cp], t[cp]);
            else
                wino2_bt3<false>(x[0][cp], x[0][cp], x[1][cp], x[2][cp], x[3][cp], x[4][cp], t[cp]);
        }
#pragma unroll
        for (int ii = 0; ii < 3; ii++) {
            float y[6];
            wino2_bt6(t[0][ii], t[1][ii], t[2][ii], y);
            *reinterpret_cast<f32x4*>(vp + ii * 8) = f32x4{y[0], y[1], y[2], y[3]};
            *reinterpret_cast<f32x2*>(vp + ii * 8 + 4) = f32x2{y[4], y[5]};
        }
    };

    // ---- prologue: weights of the first two k-steps, raw(0), raw(1) (all loads in flight together),
    // then V(0) transformed and raw(2) in registers.
    // weight stream of this wave: k-step (pass, c4) -> five f32x4 per lane; a ring of two k-steps
    // (= one iteration) ahead.  The stream wraps (pass 1 -> pass 0 of the next pair: same weights).
    // uniform (SGPR) base + one 32-bit lane offset: `global_load ... v_off, s[base]`, no 64-bit VALU address math
    const char* uwave = reinterpret_cast<const char*>(upk) + (size_t)wave * (2 * 32 * 64 * T::UROW * 4);
    const unsigned ulane = lane * (T::UROW * 4);
    auto uload = [&](int kstep, int v) {       // k-step = pass*32 + c4 (uniform), v = 16-byte piece 0..4
        return *reinterpret_cast<const f32x4*>(uwave + (size_t)kstep * (64 * T::UROW * 4) + (ulane + v * 16));
    };
    f32x4 ur[2][5];
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int v = 0; v < 5; v++) ur[s][v] = uload(s, v);
    {
        raw_fetch(0);
        const f32x4 r0 = rg[0], r1 = rg[1];
        raw_fetch(1);
        __syncthreads();                        // zero fill done
        raw_store(1);
        rg[0] = r0;
        rg[1] = r1;
        raw_store(0);
    }
    __syncthreads();
    if (group == 1) transform(0, std::true_type{});
    raw_fetch(2);
    APZ_STAMP(0)

    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + wave * 16 + q * 4);
    const int ety = j >> 2, etx = j & 3;
    const int co0 = wave * 16 + q * 4;

    for (int pi = 0; pi < np; pi++) {
        const int bd0 = 2 * ((int)blockIdx.x + pi * (int)gridDim.x);
        const bool two = bd0 + 1 < n;           // the last pair of an odd batch has one board (computed twice, stored once)
        auto run_pass = [&](auto PASS) {
            constexpr int pass = decltype(PASS)::value;
            f32x4 acc[2][18];
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int p = 0; p < 18; p++) acc[b][p] = f32x4{0.f, 0.f, 0.f, 0.f};

            for (int c = 0; c < T::NCHUNK; c++) {
                const int g = (pi * 2 + pass) * T::NCHUNK + c;
#ifndef APZ_WINO_ABL_NOBAR
                __syncthreads();                // V[g&1] complete, V[(g+1)&1] and raw[g&1] free, raw[(g+1)&1] visible
#endif
                APZ_STAMP(1)
#ifndef APZ_WINO_ABL_NORAW
                raw_store(g + 2);
                raw_fetch(g + 3);
#endif
                APZ_STAMP(2)
#ifndef APZ_WINO_ABL_NOT
                if (group == (g & 1) && g + 1 < total_iters) {
                    if (c + 1 < T::NCHUNK)      // the next chunk belongs to this pass, the last one to the other
                        transform(g + 1, std::integral_constant<bool, pass == 0>{});
                    else
                        transform(g + 1, std::integral_constant<bool, pass != 0>{});
                }
#endif

                APZ_STAMP(3)
                const float* vp = vb + (g & 1) * T::V_FLOATS + (q * 16 + j) * T::VTS;
                // k-steps of the NEXT iteration, for the ring refill
                const int gn = g + 1;
                const int kn = ((gn / T::NCHUNK) & 1) * 32 + (gn & (T::NCHUNK - 1)) * 2;   // (pass, c4) flattened: pass*32 + c4
#pragma unroll
                for (int s = 0; s < 2; s++) {
#pragma unroll
                    for (int ii = 0; ii < 3; ii++) {
#pragma unroll
                        for (int b = 0; b < 2; b++) {
                            const float* vr = vp + (b * 128 + s * 64) * T::VTS + ii * 8;
#ifdef APZ_WINO_ABL_NOB
                            const f32x4 b03 = f32x4{1.f, 2.f, 3.f, 4.f};
                            const f32x2 b45 = f32x2{5.f, 6.f};
#else
                            const f32x4 b03 = *reinterpret_cast<const f32x4*>(vr);
                            const f32x2 b45 = *reinterpret_cast<const f32x2*>(vr + 4);
#endif
                            const float bb[6] = {b03[0], b03[1], b03[2], b03[3], b45[0], b45[1]};
#pragma unroll
                            for (int k = 0; k < 6; k++)
                                acc[b][ii * 6 + k] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    ur[s][(ii * 6 + k) >> 2][(ii * 6 + k) & 3], bb[k], acc[b][ii * 6 + k], 0, 0, 0);
                        }
                    }
                    // refill: the same k-step of the next iteration (two k-steps = ~2 us ahead of its use)
#ifndef APZ_WINO_ABL_NOW
                    if (c + 1 < T::NCHUNK) {    // at the end of a pass the ring is reloaded after the epilogue (registers)
#pragma unroll
                        for (int v = 0; v < 5; v++) ur[s][v] = uload(kn + s, v);
                    }
#endif
                }
            }
#ifdef APZ_WINO_ABL_NOEPI
            {
                f32x4 sum = acc[0][0];
#pragma unroll
                for (int p = 1; p < 18; p++) sum += acc[0][p] + acc[1][p];
                if (sum[0] + sum[1] + sum[2] + sum[3] == 123.456f) out[tid] = sum[0];
                return;
            }
#endif

            // ---- epilogue of the pass.  Lane (q, j): tile j = 4*ety + etx, channels co0 + r.
            // Y = A^T M A is a sum over the transformed rows i; pass 0 holds rows 0..2, pass 1 rows 3..5:
            //   rows 0..2 -> (h0+h1+h2, h1-h2, h1+h2, h1-h2);  rows 3..5 -> (h3+h4, 2(h3-h4), 4(h3+h4), 8(h3-h4)+h5)
            // with h_i = the k-direction transform of row i.  Eight items (board, r) of 4 rows x 16 B per lane.
            // No branches: the missing second board of an odd batch's last pair is computed from a copy of
            // the first and its stores are masked; rows beyond the board are loaded from row 14, never stored.
            APZ_STAMP(1)
            __builtin_amdgcn_sched_barrier(0);   // keep the epilogue's loads out of the last chunk's MFMA block (register pressure)
            const int bd1 = two ? bd0 + 1 : bd0;
            // Memory access of the epilogue goes through a wave-private LDS staging area so that every global
            // load / store instruction moves one whole 960-byte plane (60 lanes x 16 B, contiguous): in the
            // accumulator layout a wave-instruction would touch 16 scattered 64-byte pieces, which the
            // address coalescer serialises (measured: 19 us of stores + 10..28 us of loads per board pair).
            // An item (board, r) covers the four planes co = 16*wave + 4*q' + r, q' = 0..3.
            float* sw = stg + wave * (4 * T::SPLANE);
            const int s_own = q * T::SPLANE + (4 * ety) * T::SROW + 4 * etx;        // this lane's 4x4 patch (row a: + a*SROW)
            const int s_lin = (lane >> 2) * T::SROW + (lane & 3) * 4;               // plane piece `lane` (row lane>>2, quarter lane&3)
            auto plane_ptr = [&](const float* basep, int it, int qp) {               // piece `lane` of plane q' of item it
                const int pl = __builtin_amdgcn_readfirstlane(((it >> 2) ? bd1 : bd0) * T::C + wave * 16 + qp * 4 + (it & 3));
                return reinterpret_cast<const float*>(reinterpret_cast<const char*>(basep) + (size_t)pl * (T::GPLANE * 4) +
                                                      (unsigned)(lane * 16));
            };
            auto item_y = [&](int it, f32x4* y) {
                const int b = it >> 2, r = it & 3;
                float h[3][4];
#pragma unroll
                for (int ii = 0; ii < 3; ii++)
                    wino2_at6(acc[b][ii * 6 + 0][r], acc[b][ii * 6 + 1][r], acc[b][ii * 6 + 2][r], acc[b][ii * 6 + 3][r],
                              acc[b][ii * 6 + 4][r], acc[b][ii * 6 + 5][r], h[ii]);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (pass == 0) {
                        const float s12 = h[1][e] + h[2][e], d12 = h[1][e] - h[2][e];
                        y[0][e] = h[0][e] + s12;
                        y[1][e] = d12;
                        y[2][e] = s12;
                        y[3][e] = d12;
                    } else {
                        const float s34 = h[0][e] + h[1][e], d34 = h[0][e] - h[1][e];
                        y[0][e] = s34;
                        y[1][e] = 2.f * d34;
                        y[2][e] = 4.f * s34;
                        y[3][e] = __builtin_fmaf(8.f, d34, h[2][e]);
                    }
                }
            };
            auto item_store = [&](int it, f32x4* y) {   // patch layout -> staging -> four contiguous plane stores
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    if (etx == 3) y[a][3] = 0.f;   // column 15 is the halo column of the rows16 layout
                    *reinterpret_cast<f32x4*>(sw + s_own + a * T::SROW) = y[a];   // (row 15 of tile row 3 lands in the pad row)
                }
                wave_lds_fence();
                f32x4 pv[4];
#pragma unroll
                for (int qp = 0; qp < 4; qp++) pv[qp] = *reinterpret_cast<const f32x4*>(sw + qp * T::SPLANE + s_lin);
                wave_lds_fence();
#pragma unroll
                for (int qp = 0; qp < 4; qp++) {
                    const f32x4 v = pv[qp];
#ifdef APZ_WINO_ABL_NOST
                    if (v[0] == 123.456f)
#else
                    if (lane < 60 && (it < 4 || two))
#endif
                        *reinterpret_cast<f32x4*>(const_cast<float*>(plane_ptr(out, it, qp))) = v;
                }
            };
            // Rolling window of two items: the plane loads of item k+2 are in flight while item k is finished.
            // Pass 0 adds bias (+ residual) to its partial outputs, pass 1 adds pass 0's result and applies ReLU,
            // so each epilogue streams ONE tensor.
            {
                const float* src = (pass == 0) ? resid : out;
                const bool has_src = (pass == 1) || RESID;
                constexpr int WIN = 2;
                f32x4 win[WIN][4];               // as loaded: plane pieces (lane = piece); rearranged through staging on use
                auto item_load = [&](int it) {
#pragma unroll
                    for (int qp = 0; qp < 4; qp++)
                        win[it % WIN][qp] = (lane < 60) ? *reinterpret_cast<const f32x4*>(plane_ptr(src, it, qp))
                                                        : f32x4{0.f, 0.f, 0.f, 0.f};
                };
                if (has_src) {
#pragma unroll
                    for (int it = 0; it < WIN; it++) item_load(it);
                }
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    f32x4 y[4];
                    item_y(it, y);
                    const float bvr = bv[it & 3];
                    f32x4 w4[4];
                    if (has_src) {               // plane pieces -> staging -> this lane's 4x4 patch
#pragma unroll
                        for (int qp = 0; qp < 4; qp++)
                            if (lane < 60) *reinterpret_cast<f32x4*>(sw + qp * T::SPLANE + s_lin) = win[it % WIN][qp];
                        wave_lds_fence();
#pragma unroll
                        for (int a = 0; a < 4; a++) w4[a] = *reinterpret_cast<const f32x4*>(sw + s_own + a * T::SROW);
                        wave_lds_fence();
                    }
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        if (pass == 0) {
                            y[a] = y[a] + bvr;
                            if (RESID) y[a] += w4[a];
                        } else {
                            const f32x4 v = y[a] + w4[a];
#pragma unroll
                            for (int e = 0; e < 4; e++) y[a][e] = RELU ? fmaxf(v[e], 0.f) : v[e];
                        }
                    }
                    if (has_src && it + WIN < 8) item_load(it + WIN);
                    item_store(it, y);
                }
            }
#ifndef APZ_WINO_ABL_NOW
            {                                   // reload the weight ring: first two k-steps of the next pass
                int kn = (1 - pass) * 32;
                asm volatile("" : "+s"(kn));    // loop-variant to the compiler: keeps these loads here (not hoisted + spilled)
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int v = 0; v < 5; v++) ur[s][v] = uload(kn + s, v);
            }
#endif
            APZ_STAMP(5 + pass)
        };
        run_pass(std::integral_constant<int, 0>{});
        run_pass(std::integral_constant<int, 1>{});
    }
#ifdef APZ_WINO_STAMPS
    st_acc[7] = __builtin_readcyclecounter() - st_t0;
    if (lane == 0 && blockIdx.x < 4) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(const_cast<float*>(bias) + 256) + (blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 8; i++) dst[i] = st_acc[i];
    }
#endif
}

}  // namespace apz
