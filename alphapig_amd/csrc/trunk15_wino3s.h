// Small-batch form of trunk15_wino3_kernel (trunk 3x3 convolution 128 -> 128 at 15x15 + folded BN (+ residual) + ReLU as a
// fused F(4x4,3x3) Winograd convolution on fp32 MFMA) for gfx950 -- the latency path of the drop-in single-game API:
// PolicyValueNet.policy_value_fn evaluates ONE board per call (policy_value_net_mxnet.py:261-280, mcts_alphaZero.py:124),
// MCTSPlayer.get_action 400 of them in sequence, and the arena / serving paths (evaluate/ChessClient.py:189-239,
// human_play_mxnet.py:66-69) a handful.
//
// Why: trunk15_wino3_kernel's work item is a board PAIR x 64 output channels -- one board runs on two of the 256 CUs,
// each streaming 1.18 MB of weights and issuing both boards' MFMAs (the missing second board is computed and dropped):
// 45 us per layer, 1.0 ms per leaf.  Here a board is SIXTEEN workgroups: 8 tiles of 16 output channels x 2 halves of the
// 36 transformed positions (rows 0..2 / 3..5 of the 6x6).  On this chip an fp32 MFMA and the VALU work of the input
// transform never overlap on a SIMD, so a workgroup's time is their SUM per SIMD; the position split halves both (a
// workgroup transforms only its three rows), and gives the MFMA waves and the transform waves SIMDs of their own.
// (One workgroup per (board, 16 channels) with all 36 positions: 17 us per layer at one board; this form: see
// profiles/r03_latency.md.)
//
// SAME BITS as trunk15_wino3_kernel for every board: the input transform (B^T d B), the accumulation order (one
// v_mfma_f32_16x16x4_f32 per position and k-step, k-steps in ascending channel order) and the output transform (A^T M A
// with wino3's row-partial formulas lo / hi, + bias, + residual, ReLU) are restated operation for operation -- every
// operation is an elementwise IEEE fp32 add / sub / fma, so the packed forms of wino3 and the scalar forms here round
// identically.  tests/test_gpu_net.py holds the two kernels to bit equality across batch shapes.
//
// Workgroup (board, cot, half), 256 threads, per chunk of 8 input channels (16 chunks, one barrier each):
//   waves 0-1  MFMA: wave w = positions 18 half + 9 w .. + 8, 2 k-steps x 9 MFMAs over V[chunk g]; weights in
//              WinoPackSmall order (two coalesced 16-byte loads + one 4-byte load per k-step); and ALL the staging:
//              thread -> plane, four 16-byte pieces, global -> registers -> raw LDS tile.  Planes and weights are
//              requested FOUR chunks ahead (register rings): a chunk is ~0.3 us of work, a round trip 1-2 us
//   waves 2-3  transform of chunk g + 1: thread = (channel, tile), the `half` rows of its 6x6 patch -> 18 values of
//              V[pos 18][ch 8][tile 16]; LDS only, no global load
// The two roles run SEPARATE loops with the same barriers.  What that is for (profiles/r03_latency.md): with global loads
// on one side of a wave-role branch hipcc sizes every vmcnt wait for the side without them, and with 16-byte loads that
// leave dead registers the allocator reuses those while the load is in flight -- both made every chunk wait out a round
// trip (15.9 us per layer, 11 of them in this loop; 13.6 us after).
// Epilogue: the 18 x 16 x 16 accumulators through LDS (M aliases the two V buffers); thread = (channel, tile) forms its row
// partial (wino3's lo = (h0+h1+h2, h1-h2, h1+h2) or hi = (h3+h4, h3-h4, h5) per output column: 12 floats) and writes it to
// a global slab.  The two halves of a (board, cot) meet through an in-launch reduction (cdna_hip_programming.md, "In-launch
// split-K reduction", counter form): every wave drains its stores, one lane issues an agent-scope release and draws a
// ticket from the pair's counter; the workgroup that draws the second ticket issues an agent-scope acquire, reads the
// other half's slab and finishes the 4x4 output patches.  Ticket = an atomic EXCHANGE of the launch's `epoch` (a number
// the engine never hands out twice, never 0) into the pair's word: the second arriver is the one that gets the epoch back.
// Nothing depends on what earlier launches left behind (round 3 counted arrivals and told the second by an odd value: one
// launch that died half way would have flipped every later result on that engine); the words start at zero, set by a
// stream-ordered memset at allocation and again whenever the epoch counter wraps.
// Blocks b and b + 8 (the two halves) are dealt to the same XCD by the dispatcher as observed -- speed only.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wino_common.h"

namespace apz {

struct Wino3S {
    static constexpr int C = 128, CK = 8, NCHUNK = C / CK;
    static constexpr int GPLANE = 240;
    static constexpr int RROW = 20, RPS = 17 * RROW, RFRONT = 24;
    static constexpr int RAW_FLOATS = RFRONT + CK * RPS;          // 2744
    static constexpr int VPOS = CK * 16;                           // floats per position
    static constexpr int V_FLOATS = 18 * VPOS;                     // 2304: the workgroup's 18 positions
    static constexpr int M_FLOATS = 18 * 16 * 16;                  // == 2 * V_FLOATS: M lives in the two V buffers
    static constexpr int LDS_FLOATS = 2 * RAW_FLOATS + 2 * V_FLOATS + 4;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;               // 40.4 KB
    static constexpr int SLAB_FLOATS = 12 * 256;                   // row partials of one (board, cot, half)
    static constexpr int UROW = WinoPack::UROW, USTEP = 64 * UROW;
    static constexpr int MAX_BOARDS = 32;                          // launchers: at most two workgroups per CU
    static size_t slab_floats() { return (size_t)MAX_BOARDS * 8 * 2 * SLAB_FLOATS; }
    static size_t counters() { return (size_t)MAX_BOARDS * 8; }
    static_assert(M_FLOATS == 2 * V_FLOATS, "M aliases V");
};

template <bool RESID, bool RELU = true>
__global__ __launch_bounds__(256) void trunk15_wino3s_kernel(const float* __restrict__ in, const float* __restrict__ upk /* WinoPackSmall */,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ resid, float* __restrict__ out,
                                                             int n, float* __restrict__ slabs, unsigned* __restrict__ tickets, unsigned epoch) {
    using T = Wino3S;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* rawb = lds;                          // [2][RAW_FLOATS]
    float* vb = lds + 2 * T::RAW_FLOATS;        // [2][V_FLOATS]; after the last chunk: M [18][16 ch][16 tiles]
    unsigned* lflag = reinterpret_cast<unsigned*>(lds + 2 * T::RAW_FLOATS + 2 * T::V_FLOATS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, j = lane & 15;
    const int bd = blockIdx.x >> 4, half = (blockIdx.x >> 3) & 1, cot = blockIdx.x & 7;
    if (bd >= n) return;                        // (uniform)

    // ---- staging role (the MFMA waves: threads 0..127; ALL global loads of the workgroup are theirs, see below): thread ->
    // plane tid >> 4, pieces (tid & 15) + 16 i, i = 0..3 (of 60; the last clamps)
    // A chunk's interval is ~0.3 us of work, an L2 / HBM round trip is 1-2 us: planes and weights are requested FOUR chunks
    // ahead into register rings (slot = chunk & 3; the loops are unrolled by four so that slots are static).
    const int st_p = (tid >> 4) & 7, st_k0 = tid & 15;
    f32x4 rg[4][4];
    auto raw_fetch = [&](auto SLOT, int g) {
        constexpr int sl = decltype(SLOT)::value;
        g = g < T::NCHUNK ? g : T::NCHUNK - 1;
        const float* src = in + ((size_t)bd * T::C + g * T::CK + st_p) * T::GPLANE;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int k = st_k0 + 16 * i < 60 ? st_k0 + 16 * i : st_k0 + 32;
            rg[sl][i] = *reinterpret_cast<const f32x4*>(src + k * 4);
        }
    };
    auto raw_store = [&](auto SLOT, int par) {
        constexpr int sl = decltype(SLOT)::value;
        float* dst = rawb + par * T::RAW_FLOATS + T::RFRONT + st_p * T::RPS;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int k = st_k0 + 16 * i < 60 ? st_k0 + 16 * i : st_k0 + 32;
            *reinterpret_cast<f32x4*>(dst + (k >> 2) * T::RROW + (k & 3) * 4) = rg[sl][i];
        }
    };
    // ---- transform role (waves 2, 3): unit = (channel, tile); the workgroup's row half ph = half
    const bool producer = wave >= 2;
    const int unit = tid & 127, ph = half;
    const int tty = (unit >> 2) & 3, ttx = unit & 3;
    const int tr_off = T::RFRONT + (unit >> 4) * T::RPS + (4 * tty - 1 + ph) * T::RROW + 4 * ttx;   // the upper half skips patch row 0
    // trunk15_wino3_kernel's tslice, all 18 slices in one go (same operations, same order of roundings)
    auto transform = [&](int par) {
        const float* rp = rawb + par * T::RAW_FLOATS + tr_off;
        float* vp = vb + par * T::V_FLOATS + unit;
        f32x2 xr[5][3], u0[3], u1[3], u2[3], tt[3][3];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const f32x4 c03 = *reinterpret_cast<const f32x4*>(rp + i * T::RROW);
            xr[i][0] = f32x2{rp[i * T::RROW - 1], rp[i * T::RROW + 4]};
            xr[i][1] = f32x2{c03[0], c03[1]};
            xr[i][2] = f32x2{c03[2], c03[3]};
        }
#pragma unroll
        for (int cp = 0; cp < 3; cp++) {
            u2[cp] = ph == 0 ? fma2(-4.f, xr[1][cp], xr[3][cp]) : xr[3][cp] - xr[1][cp];
            u1[cp] = ph == 0 ? fma2(-4.f, xr[2][cp], xr[4][cp]) : xr[2][cp] - xr[0][cp];
            u0[cp] = fma2(-5.f, xr[2][cp], xr[4][cp]);
            u0[cp] = fma2(4.f, xr[0][cp], u0[cp]);
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
#pragma unroll
        for (int ii = 0; ii < 3; ii++) {
            const float v0 = tt[ii][0][0], v5 = tt[ii][0][1], v1 = tt[ii][1][0], v2 = tt[ii][1][1], v3 = tt[ii][2][0],
                        v4 = tt[ii][2][1];
            const float a = __builtin_fmaf(-4.f, v2, v4), b = __builtin_fmaf(-4.f, v1, v3);
            const float c = v4 - v2, d = v3 - v1;
            float* row = vp + (6 * ii) * T::VPOS;
            row[0 * T::VPOS] = __builtin_fmaf(4.f, v0, __builtin_fmaf(-5.f, v2, v4));
            row[1 * T::VPOS] = a + b;
            row[2 * T::VPOS] = a - b;
            row[3 * T::VPOS] = __builtin_fmaf(2.f, d, c);
            row[4 * T::VPOS] = __builtin_fmaf(-2.f, d, c);
            row[5 * T::VPOS] = __builtin_fmaf(4.f, v1, __builtin_fmaf(-5.f, v3, v5));
        }
    };
    // ---- weights of MFMA wave w (positions 18 half + 9 w .. + 8): WinoPackSmall, three coalesced 16-byte loads per k-step
    const int mw = wave & 1;
    const float* ubase = upk + (size_t)((cot * 4 + 2 * half + mw) * 32) * WinoPackSmall::STEP + lane * 4;
    // (value 8 of a k-step as ONE float: a 16-byte load would leave three dead registers per k-step, the allocator reuses
    // them for temporaries, and every such write then waits for the load still in flight -- the wait of a recent load)
    f32x4 uq[4][2][2];
    float u8[4][2];
    auto uload = [&](auto SLOT, int g) {
        constexpr int sl = decltype(SLOT)::value;
        g = g < T::NCHUNK ? g : T::NCHUNK - 1;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const float* up = ubase + (size_t)(2 * g + s) * WinoPackSmall::STEP;
            uq[sl][s][0] = *reinterpret_cast<const f32x4*>(up);
            uq[sl][s][1] = *reinterpret_cast<const f32x4*>(up + 256);
            u8[sl][s] = up[512];
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    f32x4 acc[9];
#pragma unroll
    for (int m = 0; m < 9; m++) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Role-specific code paths from here to the epilogue, with the same barriers in both: the transform waves touch LDS
    // only; every global load sits in the MFMA waves' path, unconditionally, so that hipcc's vmcnt bookkeeping is exact
    // (with loads on one side of a wave-role branch it sized the staging wait for the side WITHOUT them -- vmcnt(7) -- and
    // the MFMA waves waited for all but their seven youngest loads at every chunk: 0.7 us per chunk for 0.3 us of work).
    const float* vrd = vb + (9 * mw) * T::VPOS + q * 16 + j;
#ifndef APZ_W3S_CHUNKS
#define APZ_W3S_CHUNKS T::NCHUNK        /* (measurement builds shorten the loop: what a layer costs without it) */
#endif
    if (producer) {
        for (int i = tid * 4; i < 2 * T::RAW_FLOATS; i += 1024) *reinterpret_cast<f32x4*>(&lds[i]) = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                        // zero halo in place
        __syncthreads();                        // raw[0] = chunk 0
        transform(0);
        __syncthreads();                        // V[0] = chunk 0, raw[1] = chunk 1
        for (int g = 0; g < APZ_W3S_CHUNKS; g++) {
            if (g + 1 < T::NCHUNK) transform((g & 1) ^ 1);       // raw[(g + 1) & 1] -> V[(g + 1) & 1]
            __syncthreads();
        }
    } else {
        raw_fetch(I0{}, 0);
        raw_fetch(I1{}, 1);
        raw_fetch(I2{}, 2);
        raw_fetch(I3{}, 3);
        uload(I0{}, 0);
        uload(I1{}, 1);
        uload(I2{}, 2);
        uload(I3{}, 3);
        for (int i = tid * 4; i < 2 * T::RAW_FLOATS; i += 1024) *reinterpret_cast<f32x4*>(&lds[i]) = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                        // zero halo in place
        raw_store(I0{}, 0);
        raw_fetch(I0{}, 4);
        __syncthreads();                        // raw[0] = chunk 0
        raw_store(I1{}, 1);
        raw_fetch(I1{}, 5);
        __syncthreads();                        // V[0] = chunk 0, raw[1] = chunk 1
        // interval g: stage chunk g + 2 into raw[g & 1] (its chunk g was transformed in the interval before), request chunk
        // g + 6, MFMAs over V[g & 1], then request the weights of chunk g + 4.  One barrier per chunk.
        auto chunk = [&](auto SLOT, int g) {    // SLOT = g & 3
            constexpr int sl = decltype(SLOT)::value;
            using RS = std::integral_constant<int, (sl + 2) & 3>;
            const int par = g & 1;
            raw_store(RS{}, par);               // chunk g + 2 (past the end: a harmless re-store of chunk 15)
            raw_fetch(RS{}, g + 6);
            const float* vp = vrd + par * T::V_FLOATS;
            float bo[2][9];
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int m = 0; m < 9; m++) bo[s][m] = vp[m * T::VPOS + s * 64];
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int m = 0; m < 9; m++) {
                    const float a = m < 8 ? uq[sl][s][m >> 2][m & 3] : u8[sl][s];
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bo[s][m], acc[m], 0, 0, 0);
                }
            uload(SLOT, g + 4);
            __syncthreads();
        };
        for (int g = 0; g < APZ_W3S_CHUNKS; g += 4) {
            chunk(I0{}, g);
            chunk(I1{}, g + 1);
            chunk(I2{}, g + 2);
            chunk(I3{}, g + 3);
        }
    }
    // ---- epilogue, part 1: M[pos 18][ch 16][tile 16] through LDS; thread (channel c, tile) forms its row partial
    // (the chunk loop's last barrier: every wave's MFMAs over V are done)
    if (!producer) {
#pragma unroll
        for (int m = 0; m < 9; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) vb[(9 * mw + m) * 256 + (4 * q + r) * 16 + j] = acc[m][r];
    }
    const int ec = tid >> 4, et = tid & 15, ety = et >> 2, etx = et & 3;
    const size_t plane = ((size_t)bd * T::C + cot * 16 + ec) * T::GPLANE;
    const int po = (4 * ety) * 16 + 4 * etx;    // this thread's patch inside the plane (row a: + 16 a)
    __syncthreads();
    float own[3][4];                            // half 0: lo = (h0+h1+h2, h1-h2, h1+h2); half 1: hi = (h3+h4, h3-h4, h5); per column e
    {
        const float* mp = vb + ec * 16 + et;
        float hh[3][4];
#pragma unroll
        for (int i = 0; i < 3; i++) {           // the k-direction transform of row 3 half + i (wino3's partial2)
            float m[6];
#pragma unroll
            for (int k = 0; k < 6; k++) m[k] = mp[(6 * i + k) * 256];
            const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
            hh[i][0] = (m[0] + s12) + s34;
            hh[i][1] = __builtin_fmaf(2.f, d34, d12);
            hh[i][2] = __builtin_fmaf(4.f, s34, s12);
            hh[i][3] = __builtin_fmaf(8.f, d34, d12) + m[5];
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (half == 0) {
                const float s12 = hh[1][e] + hh[2][e];
                own[0][e] = hh[0][e] + s12;
                own[1][e] = hh[1][e] - hh[2][e];
                own[2][e] = s12;
            } else {
                own[0][e] = hh[0][e] + hh[1][e];
                own[1][e] = hh[0][e] - hh[1][e];
                own[2][e] = hh[2][e];
            }
        }
    }
    // ---- part 2: publish the partial, draw a ticket; the pair's second arriver combines (in-launch reduction, counter form)
    const int pairi = bd * 8 + cot;
    {
        f32x4* slab = reinterpret_cast<f32x4*>(slabs + ((size_t)pairi * 2 + half) * T::SLAB_FLOATS) + tid * 3;
#pragma unroll
        for (int v = 0; v < 3; v++) slab[v] = f32x4{own[v][0], own[v][1], own[v][2], own[v][3]};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every wave: its slab stores have left
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (keep: the fence's own wait can be dropped by the compiler)
        const unsigned old = __hip_atomic_exchange(tickets + pairi, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned second = old == epoch ? 1u : 0u;     // the other half of this launch has been here
        if (second) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *lflag = second;
    }
    __syncthreads();
    if (*lflag == 0u) return;                   // first arriver of the pair: done (uniform)
    float other[3][4];
    {
        const f32x4* slab = reinterpret_cast<const f32x4*>(slabs + ((size_t)pairi * 2 + (half ^ 1)) * T::SLAB_FLOATS) + tid * 3;
#pragma unroll
        for (int v = 0; v < 3; v++) {
            const f32x4 t4 = slab[v];
            other[v][0] = t4[0]; other[v][1] = t4[1]; other[v][2] = t4[2]; other[v][3] = t4[3];
        }
    }
    f32x4 rs[4];
    if (RESID) {
#pragma unroll
        for (int a = 0; a < 4; a++)
            rs[a] = (4 * ety + a < 15) ? *reinterpret_cast<const f32x4*>(resid + plane + po + 16 * a) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float bvr = bias[cot * 16 + ec];
    f32x4 y[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float lo0 = half == 0 ? own[0][e] : other[0][e], lo1 = half == 0 ? own[1][e] : other[1][e],
                    lo2 = half == 0 ? own[2][e] : other[2][e];
        const float hi0 = half == 0 ? other[0][e] : own[0][e], hi1 = half == 0 ? other[1][e] : own[1][e],
                    hi2 = half == 0 ? other[2][e] : own[2][e];
        y[0][e] = lo0 + hi0;
        y[1][e] = __builtin_fmaf(2.f, hi1, lo1);
        y[2][e] = __builtin_fmaf(4.f, hi0, lo2);
        y[3][e] = lo1 + __builtin_fmaf(8.f, hi1, hi2);
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
        f32x4 v = y[a] + bvr;
        if (RESID) v += rs[a];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = RELU ? fmaxf(v[e], 0.f) : v[e];
        if (etx == 3) v[3] = 0.f;               // column 15 is the halo column of the rows16 layout
        if (4 * ety + a < 15) *reinterpret_cast<f32x4*>(out + plane + po + 16 * a) = v;
    }
}

}  // namespace apz
