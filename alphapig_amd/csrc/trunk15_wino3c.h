// Trunk 3x3 convolution (128 -> 128 channels, 15x15 board) + folded BN + (residual) + ReLU as a fused F(4x4,3x3)
// Winograd convolution on the BF16 matrix pipe at fp32 accuracy (three bf16 terms per operand, six products per fp32
// product: the arithmetic of trunk15_wino3b.h, same packed weights) -- the TWO-PASS form (round 5).
//
// Why a second form.  trunk15_wino3b_kernel's work item is (board pair, 64 output channels, all 36 positions): every
// item transforms and splits ALL 128 input channels, i.e. the vector work of B^T d B + the hi / mid / lo split runs twice
// per board pair, and its counters (profiles/r04_wino3b.md) show a kernel bound by the SUM of four half-loaded resources
// (matrix pipe 0.27, vector unit 0.38, LDS 0.49, vector memory 0.45) of which the vector unit is the one the in-order
// issue cannot hide.  Here the accumulator budget (295 KB = the register file's share) is spent the other way round:
//   work item = (board pair, ALL 128 output channels, one ROW HALF of the transformed tile = 18 of the 36 positions)
// A pair is two items in a row in ONE workgroup (pass 0: rows 0..2, pass 1: rows 3..5).  The input transform of a pass
// produces only the three transformed rows the pass contracts, so a board pair is transformed ONCE (half the vector
// work, half the V bytes written to LDS); the weights a pair streams are unchanged (each pass reads the units of its
// own positions); the output transform Y = A^T M A is linear in the rows of M, so pass 0 parks its partial tile sums
// (+ bias + residual, per thread, 64 contiguous bytes) in a per-workgroup scratch and pass 1 adds its own and applies the ReLU.
// No cross-workgroup hand-off, no duo placement, one workgroup per pair.
//
// The split itself is cheaper too: thread = (tile, channel) emits its own channel's terms as three 16-bit stores of the
// UPPER halves of (value, value - hi, value - hi - mid) (ds_write_b16_d16_hi: no pack instruction, no lane exchange, no
// select) -- 4 vector instructions per value instead of 7.
//
// Eight waves:
//   MFMA role       wave w -> 32 output channels cog = w >> 1, position block ki = w & 1 (columns 3 ki .. 3 ki + 2 of the
//                   pass's three rows: 9 positions x 16 = 144 accumulator registers); unit = Wino3B's (cog, 2 pass + ki)
//   transform role  256 (board, channel, tile) tasks per chunk = four waves: the chunks alternate between waves 0-3 and
//                   waves 4-7 (one of the two waves of every SIMD), wave w -> board w & 1, channels 4 ((w >> 1) & 1) .. + 3
//   staging role    wave w -> planes w (board 0) and w + 8 (board 1) of the chunk; lane -> 16-byte piece (LDS-DMA)
// Epilogue per pass: four steps of 32 output channels; the two waves of the step's cog leave M[pos 18][co 32][col 32] in
// LDS (over V), every thread gathers two (channel, column) units.
//
// Layouts.  in / resid / out: rows16 [n][128][15][16] (col 15 == 0).  raw (LDS): as Wino3B.  V (LDS): [pos 18][term 3]
// [col 32][8 ch] bf16.  Weights: Wino3B::upk_offset (unchanged).  scratch: [workgroup][co 128][col 32][16] floats.
#pragma once
#include "trunk15_wino3b.h"

namespace apz {

struct Wino3C {
    static constexpr int C = 128, CK = 8, NCHUNK = C / CK;
    static constexpr int GPLANE = 240;
    static constexpr int RROW = Wino3B::RROW, RPS = Wino3B::RPS, RFRONT = Wino3B::RFRONT, RAW_FLOATS = Wino3B::RAW_FLOATS;
    static constexpr int VTERM = 32 * 16, VPOS = 3 * VTERM, NPOS = 18, V_BYTES = NPOS * VPOS;   // 512, 1536, 27648 bytes
    static constexpr int UNIT = Wino3B::UNIT;
    static constexpr int MQ_FLOATS = NPOS * 32 * 32;                   // M[pos 18][co 32][col 32]: 73728 bytes over both V buffers
    static constexpr int SROW = 20, SPLANE = 16 * SROW;                // staging plane: 16 rows x 20 floats
    static constexpr int STG_FLOATS = 8 * 4 * SPLANE;                  // 8 waves x 4 planes (40 KiB)
    static constexpr int LDS_BYTES = 2 * RAW_FLOATS * 4 + (MQ_FLOATS + STG_FLOATS) * 4;   // 150016
    static_assert(MQ_FLOATS * 4 >= 2 * V_BYTES, "the epilogue area covers both V buffers");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static constexpr size_t SCRATCH_FLOATS_PER_WG = (size_t)128 * 32 * 16;   // pass 0's partial tile sums of one pair (256 KiB)
    static constexpr int THREADS = 512;
};

// grid: one workgroup per board pair, at most one per CU (a workgroup walks pairs b, b + grid, ...)
inline int wino3c_grid(int n, int num_cu) {
    const int npairs = (n + 1) >> 1;
    return npairs < num_cu ? npairs : num_cu;
}

#ifdef APZ_WINO3C_STAMPS
__device__ unsigned long long apz_wino3c_stamps[4 * 8 * 8];   // [workgroup 4][wave 8][phase 8]
__device__ unsigned apz_wino3c_trace[8 * 64 * 2];              // workgroup 0: [wave 8][event 64][barrier wait, work] cycles
#endif

template <bool RESID, bool RELU = true>
__global__ __launch_bounds__(512) void trunk15_wino3c_kernel(const float* __restrict__ in, const void* __restrict__ upk,
                                                             const float* __restrict__ bias, const float* __restrict__ resid,
                                                             float* __restrict__ out, float* __restrict__ scratch, int n) {
    using T = Wino3C;
#ifdef APZ_WINO3C_STAMPS
    // phases: 0 item prologue, 1 barrier waits, 2 chunk bodies, 3 epilogue, 7 total
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_readcyclecounter();
    const unsigned long long st_t0 = st_t;
    int st_ev = 0;
#define APZC_STAMP(ph_)                                               \
    {                                                                 \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        st_acc[ph_] += now_ - st_t;                                   \
        if (blockIdx.x == 0 && lane == 0 && st_ev < 128) { apz_wino3c_trace[wave * 128 + st_ev] = ((unsigned)(ph_) << 28) | (unsigned)(now_ - st_t); st_ev++; } \
        st_t = now_;                                                  \
    }
#else
#define APZC_STAMP(ph_)
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* rawb = lds;                                                    // [2][RAW_FLOATS]
    char* vbase = reinterpret_cast<char*>(lds + 2 * T::RAW_FLOATS);       // [2][V_BYTES]
    float* mq = lds + 2 * T::RAW_FLOATS;                                  // epilogue: M[pos 18][co 32][col 32] (over V)
    float* stg = mq + T::MQ_FLOATS;                                       // epilogue: [wave 8][plane 4][16 x 20]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- work items: item t = (pair pair0 + (t >> 1) * grid, pass t & 1)
    const int npairs = (n + 1) >> 1, G_ = (int)gridDim.x, b_ = (int)blockIdx.x;
    const int np = b_ < npairs ? (npairs - b_ + G_ - 1) / G_ : 0;
    const int nitems = 2 * np;
    if (np == 0) return;
    auto item_pair = [&](int t) { return b_ + (t >> 1) * G_; };
    float* p0s = scratch + (size_t)b_ * T::SCRATCH_FLOATS_PER_WG;

    const unsigned plane_b = T::GPLANE * 4;
    const unsigned act_bytes = (unsigned)n * T::C * plane_b;
    const __amdgpu_buffer_rsrc_t r_res =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RESID ? resid : in), 0, act_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, act_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_u =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(upk), 0, (unsigned)Wino3B::UPK_BYTES, 0x00020000);
    // pass 0's partial sums come back to the thread that wrote them (same CU, same L1: a store updates the line it hits)
    const __amdgpu_buffer_rsrc_t r_p0 =
        __builtin_amdgcn_make_buffer_rsrc(p0s, 0, (unsigned)(T::SCRATCH_FLOATS_PER_WG * 4), 0x00020000);
    auto bload = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    auto bstore = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff, const f32x4 v) {   // soffset = 0: see trunk15_wino3.h
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff + soff, 0, 0);
    };

    // ---- staging role (as trunk15_wino3b_kernel): planes by LDS-DMA, one instruction per plane
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)rawb;
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    const unsigned long long in_a = (unsigned long long)in;
    const i32x4_ dma_rsrc = {(int)(unsigned)(in_a & 0xffffffffull), (int)(unsigned)((in_a >> 32) & 0xffffull), (int)act_bytes, 0x00020000};
    const unsigned dma_vo = lane < 60 ? lane * 16 : 0x80000000u;   // lanes 60..63: out of range (row 15 of the tile stays zero)
    auto raw_dma = [&](int t, int c, int par) {       // chunk c (clamped) of item t -> raw[par]
        c = c < T::NCHUNK ? c : T::NCHUNK - 1;
        const int bd0_ = 2 * item_pair(t);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int plane = 8 * j + wave;           // board j, channel `wave`
            const int bdp = bd0_ + j;
            const int bd = bdp < n ? bdp : n - 1;
            const unsigned so = (unsigned)(bd * T::C + c * T::CK + wave) * plane_b;
            const unsigned dst = lds0 + (unsigned)(par * T::RAW_FLOATS + T::RFRONT + plane * T::RPS) * 4;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(dma_vo), "s"(dma_rsrc), "s"(dst), "s"(so) : "memory");
        }
    };

    // ---- MFMA role: 32 output channels cog, columns 3 ki .. 3 ki + 2 of the pass's three rows
    const int cog = wave >> 1, ki = wave & 1;
    const int r31 = lane & 31, hh = lane >> 5;
    // weights: per-lane byte offsets of the two A fragments inside a unit: Fa = [hi | mid] (M1 and M3), Fb = [lo | hi] (M2)
    const unsigned a_vo0 = r31 * 16 + hh * T::VTERM, a_vo1 = r31 * 16 + (1 - hh) * 2 * T::VTERM;
    auto pos_off = [](int p9) { return (6 * (p9 / 3) + (p9 % 3)) * T::VPOS; };   // position p9 of the block, relative to its first

    // weight stream: unit index of this wave = (item t * 16 + chunk c) * 9 + p9
    static constexpr int RING = 6;                    // weight units in registers (5 in flight); 18 units per two chunks
    bf16x8 af[RING][2];
    auto unit_load = [&](int t, int c, int p9, int slot) {
        // (c, p9) may run past the end of the item: carry into the next item; past the last item: reload the last unit
        if (p9 >= 9) { p9 -= 9; c += 1; }
        if (c >= T::NCHUNK) { c -= T::NCHUNK; t += 1; }
        if (t >= nitems) { t = nitems - 1; c = T::NCHUNK - 1; p9 = 8; }
        const int blk = 2 * (t & 1) + ki;             // Wino3B's position block (ri = pass, ki)
        const unsigned so = (unsigned)(((cog * 4 + blk) * T::NCHUNK + c) * 9 + p9) * T::UNIT;
        af[slot][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_u, a_vo0, so, 0));
        af[slot][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_u, a_vo1, so, 0));
    };

    // ---- transform role (the chunk's duty waves): board tb, channels 4 ch4 .. 4 ch4 + 3; lane -> (tile row tty, tile
    // column ttx, channel) exactly as in trunk15_wino3b_kernel (conflict-free ds_read_b128 of the raw tile).  The per-lane
    // offsets are rebuilt from an opaque copy of the lane id wherever a transform starts (three registers for the length
    // of a chunk instead of three more live across the whole kernel: hipcc spills those, and a scratch reload in the
    // chunk loop is followed by vmcnt(0), a full drain of the weight ring)
    // (the lane id produced INSIDE a volatile asm: an mbcnt builtin is hoisted to the kernel's start, spilled, and reloaded here)
    auto lane_now = []() {
        int le;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(le));
        return le;
    };
    const int grp = wave >> 2;                        // duty group: transforms the chunks whose V buffer is `grp`
    struct TLane { int tr_off, tv_off; unsigned col16_mask; };
    auto tlane = [&](int ph) {
        const int le = lane_now();
        const int tb = wave & 1, ch4 = (wave >> 1) & 1;
        const int e_ = le & 1, run4 = (le >> 2) & 7;
        const int ttx = 2 * ((run4 >> 1) & 1) + ((le >> 1) & 1), cpl = run4 >> 2;
        const int tty = 2 * (le >> 5) + 1 - ((0x69 >> run4) & 1);
        const int tile = 4 * tty + ttx, chl = 4 * ch4 + 2 * cpl + e_;         // channel of the chunk (0..7)
        TLane L;
        L.tr_off = T::RFRONT + (tb * 8 + chl) * T::RPS + (4 * tty - 1 + ph) * T::RROW + 4 * ttx;
        L.tv_off = (tb * 16 + tile) * 16 + chl * 2;            // byte offset inside a (position, term) block
        L.col16_mask = ttx == 3 ? 0u : 0xffffffffu;            // column 16 does not exist: the word there is column 0 of the next row
        return L;
    };

    // zero halo rows of both raw buffers (the DMA never touches them), once
    for (int i = tid * 4; i < 2 * T::RAW_FLOATS; i += 2048) *reinterpret_cast<f32x4*>(&lds[i]) = f32x4{0.f, 0.f, 0.f, 0.f};
    // the first RING - 1 weight units
#pragma unroll
    for (int u = 0; u < RING - 1; u++) unit_load(0, 0, u, u);
    __syncthreads();
    raw_dma(0, 0, 0);                                  // the first item's first two chunks (later items: from the epilogue before)
    raw_dma(0, 1, 1);

    // Everything that depends on the pass (= the row half of the transformed tile) is instantiated twice
    auto item = [&](int t, auto PH) {
        constexpr int ph = decltype(PH)::value;
        const int bd0 = 2 * item_pair(t);
        const bool two = bd0 + 1 < n;
        TLane TL;
        float xr[5][4];                                // five patch rows of the channel: four columns at a time
        float tt[3][6];                                // row-pass results (rows 3 ph .. 3 ph + 2), columns -1 .. 4
        float oo[6];
        auto row_pass = [&](const float* rp, auto PART) {
            constexpr int part = decltype(PART)::value;
            constexpr int nc = part == 0 ? 4 : 2;
#pragma unroll
            for (int i = 0; i < 5; i++) {
                if (part == 0) {
                    const f32x4 c03 = *reinterpret_cast<const f32x4*>(rp + i * T::RROW);
                    xr[i][0] = c03[0]; xr[i][1] = c03[1]; xr[i][2] = c03[2]; xr[i][3] = c03[3];
                } else {
                    xr[i][0] = rp[i * T::RROW - 1];
                    xr[i][1] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, rp[i * T::RROW + 4]) & TL.col16_mask);   // (no branch)
                }
            }
#pragma unroll
            for (int k = 0; k < nc; k++) {
                const int kc = part == 0 ? k + 1 : 5 * k;      // column index in tt (0 = column -1, 5 = column 4)
                if (ph == 0) {                         // x = patch rows 0..4: y0 = 4x0 - 5x2 + x4, y1/y2 = (x4 - 4x2) +- (x3 - 4x1)
                    const float a = __builtin_fmaf(-4.f, xr[2][k], xr[4][k]), b = __builtin_fmaf(-4.f, xr[1][k], xr[3][k]);
                    tt[0][kc] = __builtin_fmaf(4.f, xr[0][k], __builtin_fmaf(-5.f, xr[2][k], xr[4][k]));
                    tt[1][kc] = a + b;
                    tt[2][kc] = a - b;
                } else {                               // z = patch rows 1..5: y3/y4 = (z3 - z1) +- 2(z2 - z0), y5 = 4z0 - 5z2 + z4
                    const float c = xr[3][k] - xr[1][k], d = xr[2][k] - xr[0][k];
                    tt[0][kc] = __builtin_fmaf(2.f, d, c);
                    tt[1][kc] = __builtin_fmaf(-2.f, d, c);
                    tt[2][kc] = __builtin_fmaf(4.f, xr[0][k], __builtin_fmaf(-5.f, xr[2][k], xr[4][k]));
                }
            }
        };
        auto col_pass = [&](const float* v, float* o) {   // B^T over the columns of one row
            const float a = __builtin_fmaf(-4.f, v[2], v[4]), b = __builtin_fmaf(-4.f, v[1], v[3]);
            const float c = v[4] - v[2], d = v[3] - v[1];
            o[0] = __builtin_fmaf(4.f, v[0], __builtin_fmaf(-5.f, v[2], v[4]));
            o[1] = a + b;
            o[2] = a - b;
            o[3] = __builtin_fmaf(2.f, d, c);
            o[4] = __builtin_fmaf(-2.f, d, c);
            o[5] = __builtin_fmaf(4.f, v[1], __builtin_fmaf(-5.f, v[3], v[5]));
        };
        // one value -> its three bf16 terms, each the UPPER half of a float: hi = trunc16(x), mid = trunc16(x - hi) (the
        // remainder is exact), lo = trunc16(x - hi - mid): hi + mid + lo == x, bit for bit.  Three 16-bit stores of upper
        // halves (ds_write_b16_d16_hi), two AND + two SUB.
        auto emit = [&](char* vp, float x) {
            const unsigned u = __builtin_bit_cast(unsigned, x);
            const float r = x - __builtin_bit_cast(float, u & 0xffff0000u);
            const unsigned ur = __builtin_bit_cast(unsigned, r);
            const float s = r - __builtin_bit_cast(float, ur & 0xffff0000u);
            const unsigned us = __builtin_bit_cast(unsigned, s);
#if defined(APZC_ABL_S32)      /* timing only: dword stores instead of 16-bit ones (wrong results) */
            char* vq = reinterpret_cast<char*>(reinterpret_cast<size_t>(vp) & ~(size_t)3);
            *reinterpret_cast<unsigned*>(vq) = u;
            *reinterpret_cast<unsigned*>(vq + T::VTERM) = ur;
            *reinterpret_cast<unsigned*>(vq + 2 * T::VTERM) = us;
#elif defined(APZC_ABL_S1)     /* timing only: one store per value */
            *reinterpret_cast<unsigned short*>(vp) = (unsigned short)((u ^ ur ^ us) >> 16);
#else
            *reinterpret_cast<unsigned short*>(vp) = (unsigned short)(u >> 16);
            *reinterpret_cast<unsigned short*>(vp + T::VTERM) = (unsigned short)(ur >> 16);
            *reinterpret_cast<unsigned short*>(vp + 2 * T::VTERM) = (unsigned short)(us >> 16);
#endif
        };
        // The transform of one chunk (raw[rpar] -> V[vpar], this thread's channel and tile, rows 3 ph .. 3 ph + 2) in 18
        // slices, two per MFMA slot of a chunk body
        auto tslice = [&](int rpar, int vpar, auto KK) {
            constexpr int K = decltype(KK)::value;
            const float* rp = rawb + rpar * T::RAW_FLOATS + TL.tr_off;
            char* vp = vbase + vpar * T::V_BYTES + TL.tv_off;
            if constexpr (K == 0) row_pass(rp, std::integral_constant<int, 0>{});
            else if constexpr (K == 1) row_pass(rp, std::integral_constant<int, 1>{});
            else if constexpr (K >= 3 && K < 18) {
                constexpr int ii = (K - 3) / 5, part = (K - 3) % 5;
                if constexpr (part == 0) col_pass(tt[ii], oo);
                else if constexpr (part == 1) {
                    emit(vp + (ii * 6 + 0) * T::VPOS, oo[0]);
                    emit(vp + (ii * 6 + 1) * T::VPOS, oo[1]);
                } else if constexpr (part == 2) {
                    emit(vp + (ii * 6 + 2) * T::VPOS, oo[2]);
                    emit(vp + (ii * 6 + 3) * T::VPOS, oo[3]);
                } else if constexpr (part == 3) emit(vp + (ii * 6 + 4) * T::VPOS, oo[4]);
                else emit(vp + (ii * 6 + 5) * T::VPOS, oo[5]);
            }
        };
#define APZC_ALL18(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15) F(16) F(17)

        // ---- item prologue: raw(0), raw(1) have been requested; V[0] = transform(raw(0)) by duty group 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (grp == 0) {
            TL = tlane(ph);
#define APZC_TS(k) tslice(0, 0, std::integral_constant<int, k>{});
            APZC_ALL18(APZC_TS)
#undef APZC_TS
        }
        f32x16 acc[9];
#pragma unroll
        for (int p = 0; p < 9; p++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[p][v] = 0.f;
        APZC_STAMP(0)

        // ---- chunk loop.  Iteration c: [barrier] DMA of raw(c+2) -> raw[c & 1] (read last by the transform of iteration
        // c - 1); duty group (c + 1) & 1: transform of raw[(c+1) & 1] -> V[(c+1) & 1]; all waves: MFMAs over V[c & 1]: 9 slots =
        // the wave's 9 positions, each slot 3 MFMAs (+ two slices of the transform) + the refill of the weight ring slot
        // freed by the previous slot.
#ifndef APZC_ABL_T
#define APZC_ABL_T 0      /* measurement builds of tools/wino3b_bench.hip: no transform / no weight loads in the chunk body */
#endif
#ifndef APZC_ABL_W
#define APZC_ABL_W 0
#endif
#if APZC_ABL_W
#define APZC_ULOAD(k)
#else
#define APZC_ULOAD(k) unit_load(t, c, (k) + RING - 1, (par * 9 + (k) + RING - 1) % RING);
#endif
        bf16x8 bfr[3];
        auto chunk = [&](int c, auto PAR, auto DUTY) {
            constexpr int par = decltype(PAR)::value;
            constexpr bool duty = decltype(DUTY)::value && !APZC_ABL_T;
            __syncthreads();                          // V[par] and raw[1 - par] complete; V[1 - par] and raw[par] free
            APZC_STAMP(1)
            const char* vp = vbase + par * T::V_BYTES;
            // per-lane fragment offsets rebuilt from an opaque copy of the lane id (kept live across the kernel they are
            // what hipcc spills, and every scratch reload is followed by vmcnt(0): a full drain of the weight ring)
            const int le = lane_now();
            const int b_lo0 = 3 * ki * T::VPOS + (le & 31) * 16;                  // M1: B = [hi | hi]
            const int b_lo1 = b_lo0 + (le >> 5) * T::VTERM;                       // M2: B = [hi | mid]
            const int b_lo2 = b_lo0 + (2 - (le >> 5)) * T::VTERM;                 // M3: B = [lo | mid]
            bfr[0] = *reinterpret_cast<const bf16x8*>(vp + b_lo0);
            bfr[1] = *reinterpret_cast<const bf16x8*>(vp + b_lo1);
            bfr[2] = *reinterpret_cast<const bf16x8*>(vp + b_lo2);
            if constexpr (duty) TL = tlane(ph);
#define APZC_SLOT(k)                                                                                                     \
            {                                                                                                            \
                constexpr int p9 = (k), slot = (par * 9 + (k)) % RING;                                                   \
                /* smallest products first: M3 = hi.lo + mid.mid, M2 = lo.hi + hi.mid, M1 = hi.hi + mid.hi */            \
                acc[p9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[slot][0], bfr[2], acc[p9], 0, 0, 0);                \
                if (p9 + 1 < 9) bfr[2] = *reinterpret_cast<const bf16x8*>(vp + b_lo2 + pos_off(p9 + 1));                 \
                acc[p9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[slot][1], bfr[1], acc[p9], 0, 0, 0);                \
                if (p9 + 1 < 9) bfr[1] = *reinterpret_cast<const bf16x8*>(vp + b_lo1 + pos_off(p9 + 1));                 \
                acc[p9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[slot][0], bfr[0], acc[p9], 0, 0, 0);                \
                if (p9 + 1 < 9) bfr[0] = *reinterpret_cast<const bf16x8*>(vp + b_lo0 + pos_off(p9 + 1));                 \
                if ((k) == 1) raw_dma(t, c + 2, par);                                                                    \
                if constexpr (duty) {                                                                                    \
                    tslice(1 - par, 1 - par, std::integral_constant<int, 2 * (k)>{});                                    \
                    tslice(1 - par, 1 - par, std::integral_constant<int, 2 * (k) + 1>{});                                \
                }                                                                                                        \
                /* unit k + RING - 1 goes into the ring slot of unit k - 1, whose MFMAs are done */                     \
                APZC_ULOAD(k)                                                                                            \
                __builtin_amdgcn_sched_barrier(0);                                                                       \
            }
            APZC_SLOT(0) APZC_SLOT(1) APZC_SLOT(2) APZC_SLOT(3) APZC_SLOT(4) APZC_SLOT(5) APZC_SLOT(6) APZC_SLOT(7) APZC_SLOT(8)
#undef APZC_SLOT
            APZC_STAMP(2)
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        if (grp == 0) {                                // waves 0-3: V[0] is theirs (written while V[1] is contracted: odd chunks)
            for (int c = 0; c < T::NCHUNK; c += 2) {
                chunk(c, I0{}, std::false_type{});
                chunk(c + 1, I1{}, std::true_type{});
            }
        } else {
            for (int c = 0; c < T::NCHUNK; c += 2) {
                chunk(c, I0{}, std::true_type{});
                chunk(c + 1, I1{}, std::false_type{});
            }
        }

        // ---- epilogue: four steps of 32 output channels (cog s).  Layout of the 32 x 32 tile: lane (col = lane & 31, hh =
        // lane >> 5), register v: channel (v & 3) + 8 (v >> 2) + 4 hh.
        const int cosel = lane >> 5;                   // gather role: channels 2 wave + cosel and + 16 of the step's 32, column lane & 31
        const int col = lane & 31, gbd = col >> 4, gtile = col & 15;
        const int gty = gtile >> 2, gtx = gtile & 3;
        float* sw = stg + wave * (4 * T::SPLANE);
        const int s_lin = (lane >> 2) * T::SROW + (lane & 3) * 4;
        const unsigned ep_vo = lane < 60 ? lane * 16 : 0x80000000u;
        auto ep_step = [&](auto S_) {
            constexpr int s = decltype(S_)::value;
            const int co_base = 32 * s;                // first output channel of the step
            __syncthreads();                           // MFMAs over V done (s = 0) / M and staging of the previous step consumed
            APZC_STAMP(1)
            if (s == 0 && t + 1 < nitems) {            // the raw tiles are free: the next item's first two chunks
                raw_dma(t + 1, 0, 0);
                raw_dma(t + 1, 1, 1);
            }
            // pass 0 finishes with + bias + residual and parks; pass 1 adds its share to the parked sums, ReLU, stores.  The
            // residual planes of this wave (pass 0: 2 units x 2 channels x 2 boards) / the parked sums of this thread's two
            // units (pass 1) are requested before the accumulators move
            f32x4 pre[2][4];
            if (ph == 1) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const unsigned po = (unsigned)(((co_base + 16 * u + 2 * wave + cosel) * 32 + col) * 64);
#pragma unroll
                    for (int a = 0; a < 4; a++)
                        pre[u][a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_p0, po + 16 * a, 0, 0));
                }
            } else if (RESID) {
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int pl = 0; pl < 4; pl++) {
                        const int bdp = bd0 + (pl & 1);
                        const int bd = bdp < n ? bdp : n - 1;
                        pre[u][pl] = bload(r_res, ep_vo, (unsigned)(bd * T::C + co_base + 16 * u + 2 * wave + (pl >> 1)) * plane_b);
                    }
            }
            if (cog == s) {
                // 144 single ds_write_b32 with immediate offsets from TWO base registers (rows 0-1 / row 2 of the pass: the
                // tile spans 72 KB, an immediate reaches 64 KB).  Written as assembly: hipcc pairs the stores into
                // ds_write2_b32, whose offsets reach 1 KB, builds ~70 base registers for them, spills those, and every
                // reload is a scratch round trip behind vmcnt(0) -- 30 000 cycles per step, measured
                const unsigned mb0 = lds0 + (unsigned)(2 * T::RAW_FLOATS + (3 * ki) * 1024 + (4 * hh) * 32 + r31) * 4;
                const unsigned mb1 = mb0 + 12 * 4096;
#pragma unroll
                for (int p9 = 0; p9 < 9; p9++)
#pragma unroll
                    for (int v = 0; v < 16; v++) {
                        const int off = ((p9 % 3) * 1024 + ((v & 3) + 8 * (v >> 2)) * 32) * 4 + (p9 / 3 == 1 ? 6 * 4096 : 0);
                        asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(p9 / 3 == 2 ? mb1 : mb0), "v"(acc[p9][v]), "n"(off) : "memory");
                    }
            }
            if (ph == 0 && RESID) {
#pragma unroll
                for (int pl = 0; pl < 4; pl++) *reinterpret_cast<f32x4*>(sw + pl * T::SPLANE + s_lin) = pre[0][pl];
            }
            __syncthreads();                           // M complete
            APZC_STAMP(1)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int co32 = 16 * u + 2 * wave + cosel;
                const float* mp = mq + co32 * 32 + col;
                float hrow[3][4];                       // the k-direction transform of the pass's three rows
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    float m[6];
#pragma unroll
                    for (int k = 0; k < 6; k++) m[k] = mp[(6 * i + k) * 1024];
                    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
                    hrow[i][0] = (m[0] + s12) + s34;
                    hrow[i][1] = __builtin_fmaf(2.f, d34, d12);
                    hrow[i][2] = __builtin_fmaf(4.f, s34, s12);
                    hrow[i][3] = __builtin_fmaf(8.f, d34, d12) + m[5];
                }
                // this pass's share of Y = A^T (.): rows 0..2 contribute with A^T's columns (1,0,0,0), (1,1,1,1), (1,-1,1,-1);
                // rows 3..5 with (1,2,4,8), (1,-2,4,-8), (0,0,0,1)
                f32x4 y[4];
#pragma unroll
                for (int ee = 0; ee < 4; ee++) {
                    if (ph == 0) {
                        const float s12 = hrow[1][ee] + hrow[2][ee], d12 = hrow[1][ee] - hrow[2][ee];
                        y[0][ee] = hrow[0][ee] + s12;
                        y[1][ee] = d12;
                        y[2][ee] = s12;
                        y[3][ee] = d12;
                    } else {
                        const float s34 = hrow[0][ee] + hrow[1][ee], d34 = hrow[0][ee] - hrow[1][ee];
                        y[0][ee] = s34;
                        y[1][ee] = 2.f * d34;
                        y[2][ee] = 4.f * s34;
                        y[3][ee] = __builtin_fmaf(8.f, d34, hrow[2][ee]);
                    }
                }
                float* sp = sw + (cosel * 2 + gbd) * T::SPLANE + (4 * gty) * T::SROW + 4 * gtx;
                if (ph == 0) {                          // (rows 0..2) + bias (+ residual) -> parked: 64 contiguous bytes per thread, read back by the same thread
                    const float bv = bias[co_base + co32];
                    float* pp = p0s + ((size_t)(co_base + co32) * 32 + col) * 16;
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        f32x4 v = y[a] + bv;
                        if (RESID) v += *reinterpret_cast<const f32x4*>(sp + a * T::SROW);   // (wave-private: written by this wave)
                        *reinterpret_cast<f32x4*>(pp + 4 * a) = v;
                    }
                    if (u == 0 && RESID) {              // the second unit's residual planes into the staging area
                        wave_lds_fence();
#pragma unroll
                        for (int pl = 0; pl < 4; pl++) *reinterpret_cast<f32x4*>(sw + pl * T::SPLANE + s_lin) = pre[1][pl];
                        wave_lds_fence();
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        f32x4 v = pre[u][a] + y[a];
#pragma unroll
                        for (int ee = 0; ee < 4; ee++) v[ee] = RELU ? fmaxf(v[ee], 0.f) : v[ee];
                        if (gtx == 3) v[3] = 0.f;      // column 15 is the halo column of the rows16 layout
                        *reinterpret_cast<f32x4*>(sp + a * T::SROW) = v;
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int pl = 0; pl < 4; pl++) {
                        const f32x4 pv = *reinterpret_cast<const f32x4*>(sw + pl * T::SPLANE + s_lin);
                        const unsigned vo = ((pl & 1) == 0 || two) ? ep_vo : 0x80000000u;   // the missing second board of an odd batch
                        bstore(r_out, vo, (unsigned)((bd0 + (pl & 1)) * T::C + co_base + 16 * u + 2 * wave + (pl >> 1)) * plane_b, pv);
                    }
                    if (u == 0) wave_lds_fence();       // (the second unit's tiles go into the staging area the stores just read)
                }
            }
            APZC_STAMP(3)
        };
        ep_step(std::integral_constant<int, 0>{});
        ep_step(std::integral_constant<int, 1>{});
        ep_step(std::integral_constant<int, 2>{});
        ep_step(std::integral_constant<int, 3>{});
        // (the next item's prologue starts with a barrier: M / staging are consumed before its transform writes V; its
        // vmcnt(0) also covers this pass's scratch stores before the same threads read them back)
    };
    for (int t = 0; t < nitems; t += 2) {
        item(t, std::integral_constant<int, 0>{});
        item(t + 1, std::integral_constant<int, 1>{});
    }
#ifdef APZ_WINO3C_STAMPS
    st_acc[7] = __builtin_readcyclecounter() - st_t0;
    if (lane == 0 && blockIdx.x < 4)
        for (int i = 0; i < 8; i++) apz_wino3c_stamps[(blockIdx.x * 8 + wave) * 8 + i] = st_acc[i];
#endif
}

}  // namespace apz
