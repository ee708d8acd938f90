// 8x8 boards: 3x3 convolution + folded BatchNorm (+ residual) + ReLU on the FP16 matrix pipe at fp32-level accuracy -- every
// fp32 operand as two fp16 terms (x = hi + lo, both rounded to nearest even), fp32 accumulation; the 8x8 sibling of
// trunk15_wino3h.h (read its header for the arithmetic, the error bound and the overflow word).  gfx950 only.  Round 6.
// Reference layers: policy_value_net_mxnet_simple.py:68-92 (the 6-conv net of BASELINE configs[1]), the 8x8 residual nets.
//
// Why: at 32 boards the layers with >= 64 input channels are bound by the fp32 matrix pipe (256 -> 256: 26 us at 0.59 of
// its peak, 15.4 us of chip-wide MFMA time; profiles/r03_config2.md).  v_mfma_f32_16x16x32_f16 does 16x the flops per cycle
// and the split needs 4 products per fp32 product, two instructions per (tap, 16 output channels, 16 input channels, 16
// pixels) against four v_mfma_f32_16x16x4_f32: a quarter of the matrix time.
//
// Same decomposition as conv8_kernel (conv8_small.h): work item = (board, 16 output channels), the four waves of a workgroup
// split the contraction (wave w: input channels [w C_in / 4, (w + 1) C_in / 4), in sub-chunks of 16), no workgroup barrier in
// the main loop, the four partial sums meet once in LDS and are added in wave order; a board's bits do not depend on the
// batch.  C_in must be a multiple of 64 (every wave whole sub-chunks): the net's first layer stays on conv8_kernel.
//   K = 32 = 16 input channels x the two WEIGHT terms: A = [Whi | Wlo] (lane (co j, k group kg): kg 0, 1 = hi of channels
//   0-7, 8-15; kg 2, 3 = lo), B = [X | X] for X = lo, then hi (lanes kg and kg + 2 read the same 16 bytes).
//   LDS tile of a wave, per term: [10 rows][16 cells][16 ch] fp16 -- a cell = one pixel's 16 channels = 32 bytes, row stride
//   512 bytes (ten cells used: board columns -1 .. 8): with that stride the 16-lane groups a ds_read_b128 is served in cover
//   all 64 banks exactly once for every tap.  Staging: lane = pixel; 16 coalesced dword loads (one per channel), 8
//   v_cvt_pk_f16_f32 + 16 v_fma_mixlo/hi_f16, four ds_write_b128.
//   Weights: [C_out / 16][C_in / 16][tap 9][lane 64][8] fp16 of w S[co] (per-channel power of two, undone in the bias FMA):
//   one coalesced 1 KB load per tap and sub-chunk, requested a sub-chunk ahead.
// Overflow: a non-finite pre-ReLU sum raises flag[0]; the engine repeats the forward on conv8_kernel (apz_engine.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "trunk15_wino3h.h"

namespace apz {

struct Conv8H {
    static constexpr int HW = 64;
    static constexpr int CELL = 32, RSB = 16 * CELL;            // bytes: a pixel's 16 channels, a tile row
    static constexpr int TERM_BYTES = 10 * RSB;                 // rows -1 .. 8
    static constexpr int WAVE_BYTES = 2 * TERM_BYTES;           // hi tile, lo tile
    static constexpr int RED_CS = 68;                           // channel stride of the reduction area (64 pixels + 4)
    static constexpr int RED_FLOATS = 4 * 16 * RED_CS;
    static constexpr int LDS_BYTES = 4 * WAVE_BYTES + RED_FLOATS * 4;   // 58 368: two workgroups per CU
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
    static constexpr int UNIT = 64 * 16;                        // bytes of one (cot, sub-chunk, tap): 1 KB
    __host__ __device__ static size_t pk_bytes(int cin, int cout) { return (size_t)(cout / 16) * (cin / 16) * 9 * UNIT; }
    // element (co, ci, tap, term) -> index in halfs
    __host__ __device__ static size_t pk_index(int co, int ci, int tap, int term, int cin) {
        const int cot = co >> 4, j = co & 15, s = ci >> 4, c16 = ci & 15;
        const int kg = 2 * term + (c16 >> 3), lane = kg * 16 + j;
        return ((((size_t)cot * (cin >> 4) + s) * 9 + tap) * 64 + lane) * 8 + (c16 & 7);
    }
    static bool supports(int cin, int cout) { return cin % 64 == 0 && cout % 16 == 0; }
};

// Host packing: w [cout][cin][9] fp32, scale [cout] (folded BatchNorm) -> the kernel's layout + bias8h = [cout bias][cout 1/S]
inline void conv8h_pack_host(const float* w, const double* scale, const double* shift, int cin, int cout, std::vector<uint16_t>& out,
                             std::vector<float>& bias8h) {
    out.assign(Conv8H::pk_bytes(cin, cout) / 2, 0);
    bias8h.assign(2 * (size_t)cout, 0.f);
    for (int co = 0; co < cout; co++) {
        double m = 0;
        for (int i = 0; i < cin * 9; i++) m = std::max(m, std::fabs((double)w[(size_t)co * cin * 9 + i] * scale[co]));
        const float S = Wino3H::scale_for(m);
        bias8h[co] = (float)shift[co];
        bias8h[cout + co] = 1.f / S;
        for (int ci = 0; ci < cin; ci++)
            for (int tap = 0; tap < 9; tap++) {
                const double x = (double)w[((size_t)co * cin + ci) * 9 + tap] * scale[co] * (double)S;
                const _Float16 hi = (_Float16)(float)x;
                const _Float16 lo = (_Float16)(float)(x - (double)(float)hi);
                uint16_t hb, lb;
                std::memcpy(&hb, &hi, 2);
                std::memcpy(&lb, &lo, 2);
                out[Conv8H::pk_index(co, ci, tap, 0, cin)] = hb;
                out[Conv8H::pk_index(co, ci, tap, 1, cin)] = lb;
            }
    }
}

// The same on the device (the trainer's refresh path): one workgroup of 256 threads per output channel, a thread walks the
// input channels tid, tid + 256, ... twice (the channel's maximum first, then the two terms).
__global__ void pack_conv8h_kernel(const float* __restrict__ w, const double* __restrict__ scale, const double* __restrict__ shift,
                                   unsigned short* __restrict__ pk, float* __restrict__ bias8h, int cin, int cout) {
    const int co = blockIdx.x, tid = threadIdx.x;
    const double sc = scale[co];
    double m = 0.0;
    for (int ci = tid; ci < cin; ci += blockDim.x)
#pragma unroll
        for (int tap = 0; tap < 9; tap++) m = fmax(m, fabs((double)w[((size_t)co * cin + ci) * 9 + tap] * sc));
    __shared__ double red[256];
    red[tid] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s && tid + s < (int)blockDim.x) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    const float S = Wino3H::scale_for(red[0]);
    if (tid == 0) {
        bias8h[co] = (float)shift[co];
        bias8h[cout + co] = 1.f / S;
    }
    for (int ci = tid; ci < cin; ci += blockDim.x)
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const double x = (double)w[((size_t)co * cin + ci) * 9 + tap] * sc * (double)S;
            const _Float16 hi = (_Float16)(float)x;
            const _Float16 lo = (_Float16)(float)(x - (double)(float)hi);
            pk[Conv8H::pk_index(co, ci, tap, 0, cin)] = __builtin_bit_cast(unsigned short, hi);
            pk[Conv8H::pk_index(co, ci, tap, 1, cin)] = __builtin_bit_cast(unsigned short, lo);
        }
}

// in: dense [n][cin][64] floats; out / resid: dense [n][cout][64]; pk: Conv8H layout; bias8h: [cout bias][cout 1/S].
template <bool RESID>
__global__ __launch_bounds__(256) void conv8h_kernel(const float* __restrict__ in, const void* __restrict__ pk,
                                                     const float* __restrict__ bias8h, const float* __restrict__ resid,
                                                     float* __restrict__ out, int n, int cin, int cout, int relu,
                                                     unsigned* __restrict__ flag) {
    using T = Conv8H;
    extern __shared__ __attribute__((aligned(16))) char lds8[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 4, j = lane & 15;
    char* tile = lds8 + wave * T::WAVE_BYTES;                   // [term 2][10 rows][16 cells][16 ch] fp16
    float* red = reinterpret_cast<float*>(lds8 + 4 * T::WAVE_BYTES);

    // zero the wave's tiles once: border cells are never written with anything else
    for (int i = lane * 16; i < T::WAVE_BYTES; i += 1024) *reinterpret_cast<f32x4*>(tile + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    wave_lds_fence();

    const int ncot = cout >> 4, nitems = n * ncot;
    const int nsub_all = cin >> 4, nsub = nsub_all >> 2;        // sub-chunks of 16 channels: in the layer, of this wave
    const int s_lo = wave * nsub;
    // staging role: lane = pixel (y = lane >> 3, x = lane & 7) -> cell (y + 1, x + 1)
    const int st_cell = ((lane >> 3) + 1) * T::RSB + ((lane & 7) + 1) * T::CELL;
    // B fragment of lane (pixel j of a 16-pixel tile = two board rows, k group kg): channels 8 (kg & 1) .. + 7
    const int brd = (j >> 3) * T::RSB + (j & 7) * T::CELL + (kg & 1) * 16;
    const int brd2 = brd + (kg >> 1) * T::TERM_BYTES;            // (APZ_CONV8H_SWAP) ... of the hi tile (kg 0, 1) or the lo tile (kg 2, 3)
    (void)brd2;
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pk), 0, (unsigned)T::pk_bytes(cin, cout), 0x00020000);
    unsigned nonfinite = 0;

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int b = item / ncot, cot = item - b * ncot;
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float pre[16];
        auto fetch = [&](int s) {                   // sub-chunk s of this wave: one dword per channel and lane (a pixel)
            const float* p = in + ((size_t)b * cin + 16 * (s_lo + s)) * T::HW + lane;
#pragma unroll
            for (int u = 0; u < 16; u++) pre[u] = p[(size_t)u * T::HW];
        };
        auto stash = [&]() {                        // split and store: the pixel's 16 channels, hi cell and lo cell
            typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
            u32x4 h[2], l[2];
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const float a = pre[2 * m], c = pre[2 * m + 1];
                const f16x2_ h2 = {(_Float16)a, (_Float16)c};
                const unsigned hu = __builtin_bit_cast(unsigned, h2);
                unsigned lu;
                asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lu) : "v"(hu), "v"(a));
                asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu) : "v"(hu), "v"(c));
                h[m >> 2][m & 3] = hu;
                l[m >> 2][m & 3] = lu;
            }
            *reinterpret_cast<u32x4*>(tile + st_cell) = h[0];
            *reinterpret_cast<u32x4*>(tile + st_cell + 16) = h[1];
            *reinterpret_cast<u32x4*>(tile + T::TERM_BYTES + st_cell) = l[0];
            *reinterpret_cast<u32x4*>(tile + T::TERM_BYTES + st_cell + 16) = l[1];
        };
#ifndef APZ_CONV8H_SWAP
#define APZ_CONV8H_SWAP 0     /* 1: one B fragment [Xhi | Xlo] per (tap, tile) against A = [Whi | Wlo] and A' = [Wlo | Whi] (the same 1 KB
                                 unit read with the lane halves exchanged: an L1 hit) instead of two fragments [X | X] against A: half the
                                 LDS reads for twice the weight loads.  Measured, not faster: 256 -> 256 at 32 boards 11.9 against 11.8 us,
                                 the smaller layers 3 - 8 % slower (profiles/r06_config2.md) */
#endif
#if APZ_CONV8H_SWAP
        typedef f16x8 WFrag[18];                    // [tap][A, A']
#else
        typedef f16x8 WFrag[9];
#endif
        WFrag wA, wB;
        auto wload = [&](WFrag& dst, int s) {       // the nine taps of sub-chunk s: one coalesced 1 KB load each
            const unsigned so = (unsigned)((cot * nsub_all + s_lo + s) * 9) * T::UNIT;
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
#if APZ_CONV8H_SWAP
                dst[2 * tap] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r_w, lane * 16 + (tap & 3) * T::UNIT,
                                                                                              so + (unsigned)(tap >> 2) * 4u * T::UNIT, 0));
                dst[2 * tap + 1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r_w, (lane ^ 32) * 16 + (tap & 3) * T::UNIT,
                                                                                                  so + (unsigned)(tap >> 2) * 4u * T::UNIT, 0));
#else
                dst[tap] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r_w, lane * 16 + (tap & 3) * T::UNIT,
                                                                                          so + (unsigned)(tap >> 2) * 4u * T::UNIT, 0));
#endif
            }
        };
        auto compute = [&](const WFrag& w) {
#pragma unroll
            for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
#pragma unroll
                    for (int t = 0; t < 4; t++) {
#if APZ_CONV8H_SWAP
                        // lanes kg 0, 1: the hi terms of channels 0-7, 8-15; kg 2, 3: the lo terms (Whi.Xhi + Wlo.Xlo, then Wlo.Xhi + Whi.Xlo)
                        const f16x8 b = *reinterpret_cast<const f16x8*>(tile + brd2 + (2 * t + ky) * T::RSB + kx * T::CELL);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2 * (ky * 3 + kx) + 1], b, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2 * (ky * 3 + kx)], b, acc[t], 0, 0, 0);
#else
                        const f16x8 a = w[ky * 3 + kx];
                        const char* bp = tile + brd + (2 * t + ky) * T::RSB + kx * T::CELL;
                        const f16x8 blo = *reinterpret_cast<const f16x8*>(bp + T::TERM_BYTES);
                        const f16x8 bhi = *reinterpret_cast<const f16x8*>(bp);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, blo, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bhi, acc[t], 0, 0, 0);
#endif
                    }
                }
        };
        auto turn = [&](const WFrag& w, WFrag& wnext, int s) {     // sub-chunk s from `w`; s + 1 prepared
            if (s + 1 < nsub) {
                fetch(s + 1);
                wload(wnext, s + 1);
            }
            compute(w);
            if (s + 1 < nsub) {
                wave_lds_fence();                   // this wave's reads of the tile are done
                stash();
                wave_lds_fence();
            }
        };
        wload(wA, 0);
        fetch(0);
        stash();
        wave_lds_fence();
        for (int s = 0; s < nsub; s += 2) {
            turn(wA, wB, s);
            if (s + 1 < nsub) turn(wB, wA, s + 1);
        }
        // ---- the four partial sums meet: red[wave][co 16][64 px (+4)]; lane (kg, j) reg r = co 4 kg + r, pixel 16 t + j
        __syncthreads();                            // the previous item's reduction reads are done
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[(wave * 16 + 4 * kg + r) * T::RED_CS + 16 * t + j] = acc[t][r];
        __syncthreads();
        {
            const int co = tid >> 4, p4 = (tid & 15) * 4;
            const float* rp = red + co * T::RED_CS + p4;
            f32x4 sum = *reinterpret_cast<const f32x4*>(rp) + *reinterpret_cast<const f32x4*>(rp + 16 * T::RED_CS);
            sum = sum + *reinterpret_cast<const f32x4*>(rp + 32 * T::RED_CS);
            sum = sum + *reinterpret_cast<const f32x4*>(rp + 48 * T::RED_CS);
            const float bv = bias8h[cot * 16 + co], is = bias8h[cout + cot * 16 + co];
#pragma unroll
            for (int e = 0; e < 4; e++) sum[e] = __builtin_fmaf(sum[e], is, bv);
            const size_t o = ((size_t)b * cout + cot * 16 + co) * T::HW + p4;
            if (RESID) sum = sum + *reinterpret_cast<const f32x4*>(resid + o);
            const float chk = (sum[0] + sum[1]) + (sum[2] + sum[3]);   // an overflow of the fp16 split shows as +-inf / NaN here
            nonfinite |= ((chk - chk) != 0.f) ? 1u : 0u;
            if (relu)
#pragma unroll
                for (int e = 0; e < 4; e++) sum[e] = fmaxf(sum[e], 0.f);
            *reinterpret_cast<f32x4*>(out + o) = sum;
        }
    }
    if (nonfinite && flag) *reinterpret_cast<volatile unsigned*>(flag) = 1u;
}

}  // namespace apz
