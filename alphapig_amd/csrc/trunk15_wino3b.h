// Trunk 3x3 convolution (128 -> 128 channels, 15x15 board) + folded BN + (residual) + ReLU as a fused F(4x4,3x3)
// Winograd convolution whose 36 per-position GEMMs run on the BF16 matrix pipe at fp32 accuracy: every fp32 operand is
// split into three bf16 terms (x = hi + mid + lo, 8 significant bits each) and a product is the six term products of
// order <= 2 (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi), accumulated in fp32 by the MFMA.  The dropped products are
// below 2^-24 of the product; fp32 rounds every product to 2^-24.  gfx950 only.  Opt-in (apz_set_trunk_arith): the
// default trunk kernel stays trunk15_wino3.h's exact-fp32 one, whose bits the parity tests rest on.
//
// Why this shape (HISTORY.md section 9, round 4; round 5: profiles/r05_wino3b.md).  v_mfma_f32_32x32x16_bf16 does 16x the flops of v_mfma_f32_16x16x4_f32
// per cycle, so six products per fp32 product leave 2.7x -- but operands now have to arrive 8x faster per MFMA cycle:
//   * K packing.  One MFMA contracts k = 16.  A chunk is 8 input channels; the two k halves (lanes 0-31 / 32-63 hold
//     k = 0-7 / 8-15) carry two DIFFERENT term pairings of the same 8 channels:
//         M1: A = [hi | mid], B = [hi | hi]      M2: A = [lo | hi], B = [hi | mid]      M3: A = [hi | mid], B = [lo | mid]
//     three MFMAs = all six products, no wasted k slot; M1 and M3 share ONE weight fragment, so a unit (position x
//     32 output channels x 8 input channels) is two 16-byte loads per lane for 1.5 KB of distinct bytes (the term
//     expansion needs hi three times and mid twice: the duplication lives in registers, not in the loads); and a chunk's V tile (36 positions x 32 columns x 8 channels x 3
//     terms) is 54 KB, so V and the raw input tiles can be double buffered in LDS (a 16-channel chunk would need 108 KB).
//   * Weights straight from L2 into registers, once: [cog 4][wave 4][chunk 16][position 9][term 3][co 32][8 ch] bf16 --
//     lane (co r, half h) fetches 16 bytes of term hi and lo (h = 0) or mid and hi (h = 1); the second read of hi hits L1.  6 bytes per weight instead of 4: 1.77 MB per work item, which at the ~70 GB/s a CU pulls from L2 is what
//     bounds the kernel (25 us per item, two items per CU and 512-board launch).
//   * Two waves per SIMD.  A first version gave one wave a 3x3 position block for BOTH 32-channel halves (288
//     accumulators, one wave per SIMD, every V fragment feeding two MFMAs): correct, and no faster than the fp32 kernel
//     (97 us per 512 boards against 95) -- one wave per SIMD issues ~650 instructions per chunk at 4 - 8 cycles each with
//     every LDS / L2 latency exposed (profiles/r04_wino3b.md: ablations, the co-issue probe).  Now eight waves: wave =
//     (32-channel half cc, 3x3 position block), 9 x 16 = 144 accumulator registers, as in trunk15_wino3.h; every V fragment
//     is read by two waves.
//   * The input transform B^T d B (fp32 VALU, as in trunk15_wino3.h) now also splits its results: thread = (board,
//     channel, tile, row half); the two channels of a pair sit in neighbouring lanes, exchange their transformed values
//     by DPP and each packs half of the row's positions: hi = the upper halves of both values (one v_perm_b32, truncation:
//     the three terms then sum to the value EXACTLY), remainder = value - hi (exact), ... one dword per position and term.
// Work item = (board pair, channel half) as in trunk15_wino3.h (same grids).  Eight waves:
//   MFMA role       wave w -> 32-channel half cc = w >> 2, position block (ri, ki) = ((w >> 1) & 1, w & 1)
//   transform role  wave w -> board w & 1, row half (w >> 1) & 1, channels 4 (w >> 2) .. + 3; lane -> (tile, channel)
//   staging role    wave w -> planes w (board 0) and w + 8 (board 1) of the chunk; lane -> 16-byte piece (LDS-DMA)
// Epilogue (per quarter of 16 output channels): accumulators -> LDS as M[pos 36][co 16][col 32], thread = one (channel,
// column) gathers its 36 values, Y = A^T M A, + bias (+ residual) + ReLU, whole-plane stores through a staging area.
//
// Layouts.  in / resid / out: rows16 [n][128][15][16] (col 15 == 0).  raw (LDS): [2 boards x 8 channels] planes, row
// stride 16, plane stride 272 (see Wino3B).  V (LDS): [pos 36][term 3][col 32][8 ch] bf16, col = board *
// 16 + tile: the B fragment of lane (col, h) is ONE conflict-free ds_read_b128.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <vector>

#include "trunk15_wino3.h"

namespace apz {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct Wino3B {
    static constexpr int C = 128, CK = 8, NCHUNK = C / CK;             // 16 chunks of 8 input channels
    static constexpr int GPLANE = 240;
    // raw input tile: planes as they are in HBM (15 rows x 16 floats, column 15 == 0 is the left halo of the next row) +
    // two zero rows behind each (bottom halo; the second one is the top halo of the next plane) -- the 960 bytes of a plane
    // arrive by ONE LDS-DMA instruction, no registers.  Column 16 (right halo of the last tile column) is forced to zero in
    // the transform.
    // Plane stride 272 floats = 17 x 64 bytes: the four channels a transform wave reads start 4, 8, 12 bank quads apart,
    // and with the transform's lane mapping (see the kernel) every 16-lane group of its ds_read_b128 covers the 64 banks
    // exactly once.
    static constexpr int RROW = 16, RPS = 17 * RROW, RFRONT = 32;
    static constexpr int RAW_FLOATS = RFRONT + 2 * CK * RPS + 32;      // 4416 floats (17.3 KiB)
    static constexpr int VTERM = 32 * 16, VPOS = 3 * VTERM, V_BYTES = 36 * VPOS;   // 512, 1536, 55296 bytes
    static constexpr int MAIN_BYTES = 2 * RAW_FLOATS * 4 + 2 * V_BYTES; // 146688
    static constexpr int UNIT = 3 * 32 * 16;                           // bytes of one (cog, wave, chunk, position): 1536
    static constexpr size_t UPK_BYTES = (size_t)4 * 4 * NCHUNK * 9 * UNIT;         // 3.54 MB per layer
    // epilogue: M and the store staging alias the two V buffers (and 4 KB behind them); the raw tiles stay untouched, so
    // the next item's first planes arrive while this item's outputs leave
    static constexpr int MQ_FLOATS = 36 * 16 * 32;                     // 73728 bytes at the V base
    static constexpr int SROW = 20, SPLANE = 16 * SROW;                // staging plane: 16 rows x 20 floats
    static constexpr int STG_FLOATS = 8 * 4 * SPLANE;                  // 8 waves x 4 planes (40 KiB)
    static constexpr int LDS_BYTES = 2 * RAW_FLOATS * 4 + (MQ_FLOATS + STG_FLOATS) * 4;   // 150784
    static_assert(LDS_BYTES >= MAIN_BYTES, "epilogue area covers V");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    // position p9 (= 3 a + b) of block (ri, ki) -> position index 6 i + k
    __host__ __device__ static constexpr int pos_of(int ri, int ki, int p9) { return 6 * (3 * ri + p9 / 3) + 3 * ki + p9 % 3; }
    // byte offset of element (co, ci, pos, term) in the packed weights
    __host__ __device__ static size_t upk_offset(int co, int ci, int pos, int term) {
        const int i = pos / 6, k = pos % 6, ri = i / 3, ki = k / 3, p9 = 3 * (i % 3) + (k % 3);
        const int cog = co >> 5, r = co & 31, chunk = ci >> 3, ch = ci & 7, w = 2 * ri + ki;
        return ((((size_t)(cog * 4 + w) * NCHUNK + chunk) * 9 + p9) * 3 + term) * VTERM + r * 16 + ch * 2;
    }
};

// round to nearest even, as v_cvt_pk_bf16_f32 does (host side of the split)
inline uint16_t wino3b_bf16_rne(double x, double* back) {
    float f = (float)x;
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) != 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    std::memcpy(&f, &u, 4);
    *back = (double)f;
    return (uint16_t)(u >> 16);
}

// Host packing: U[pos][co][ci] (double) -> the kernel's bf16 x 3 layout.  `u_of(co, ci, pos)` returns the value.
template <class F>
inline void wino3b_pack_host(F u_of, std::vector<uint16_t>& out) {
    out.assign(Wino3B::UPK_BYTES / 2, 0);
    for (int co = 0; co < 128; co++)
        for (int ci = 0; ci < 128; ci++)
            for (int pos = 0; pos < 36; pos++) {
                double rem = u_of(co, ci, pos), back;
                for (int term = 0; term < 3; term++) {
                    out[Wino3B::upk_offset(co, ci, pos, term) / 2] = wino3b_bf16_rne(rem, &back);
                    rem -= back;
                }
            }
}

#ifdef APZ_WINO3B_STAMPS
__device__ unsigned long long apz_wino3b_stamps[4 * 8 * 8];   // [workgroup 4][wave 8][phase 8]
#endif

template <bool RESID, bool RELU = true>
__global__ __launch_bounds__(512) void trunk15_wino3b_kernel(const float* __restrict__ in, const void* __restrict__ upk,
                                                             const float* __restrict__ bias, const float* __restrict__ resid,
                                                             float* __restrict__ out, int n) {
    using T = Wino3B;
#ifdef APZ_WINO3B_STAMPS
    // phases: 0 item prologue, 1 barrier waits, 2 chunk bodies, 3 epilogue, 7 total
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_readcyclecounter();
    const unsigned long long st_t0 = st_t;
#define APZB_STAMP(ph_)                                               \
    {                                                                 \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        st_acc[ph_] += now_ - st_t;                                   \
        st_t = now_;                                                  \
    }
#else
#define APZB_STAMP(ph_)
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* rawb = lds;                                                    // [2][RAW_FLOATS]
    char* vbase = reinterpret_cast<char*>(lds + 2 * T::RAW_FLOATS);       // [2][V_BYTES]
    float* mq = lds + 2 * T::RAW_FLOATS;                                  // epilogue: M[pos 36][co 16][col 32] (over V)
    float* stg = mq + T::MQ_FLOATS;                                       // epilogue: [wave 8][plane 4][16 x 20]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- work items: as trunk15_wino3_kernel (duo mode: blocks b and b + 8 take the two channel halves of the same pairs)
    const int npairs = (n + 1) >> 1, G_ = (int)gridDim.x, b_ = (int)blockIdx.x;
    const bool duo = (G_ & 15) == 0;
    const int pair0 = duo ? ((b_ >> 4) * 8 + (b_ & 7)) : b_;
    const int pstride = duo ? (G_ >> 1) : G_;
    const int h_fix = (b_ >> 3) & 1;
    const int np = pair0 < npairs ? (npairs - pair0 + pstride - 1) / pstride : 0;
    const int nitems = duo ? np : 2 * np;
    if (np == 0) return;
    auto item_pair = [&](int t) { return pair0 + (duo ? t : (t >> 1)) * pstride; };
    auto item_half = [&](int t) { return duo ? h_fix : (t & 1); };

    const unsigned plane_b = T::GPLANE * 4;
    const unsigned act_bytes = (unsigned)n * T::C * plane_b;
    const __amdgpu_buffer_rsrc_t r_res =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RESID ? resid : in), 0, act_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, act_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_u =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(upk), 0, (unsigned)T::UPK_BYTES, 0x00020000);
    auto bload = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    auto bstore = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff, const f32x4 v) {   // soffset = 0: see trunk15_wino3.h
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff + soff, 0, 0);
    };

    // ---- staging role: wave w brings planes w (board 0) and w + 8 (board 1) of a chunk into LDS by LDS-DMA
    // (buffer_load_dwordx4 ... lds): one instruction per plane, lane l copies 16 bytes to LDS byte m0 + 16 l (lanes 60..63
    // are out of range: row 15 of the tile stays zero; no branch, the chunk body stays one scheduling region).  Inline assembly: hipcc's
    // wait-count insertion does not see these loads (it would drain the weight ring in front of every LDS read that might
    // alias a DMA destination).  They need no wait of their own: a plane requested in slot 1 of a chunk is older than the
    // weight loads issued behind it, whose data the MFMAs of a later slot wait for -- vector memory operations complete in
    // order -- and the tile is first read behind the next chunk's barrier.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)rawb;   // LDS byte address of rawb
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

    // ---- MFMA role: 32-channel half cc of the item's 64, position block (ri, ki)
    const int cc = wave >> 2, ri = (wave >> 1) & 1, ki = wave & 1, blk = wave & 3;
    const int r31 = lane & 31, hh = lane >> 5;
    // weights: per-lane byte offsets of the two A fragments inside a unit: Fa = [hi | mid] (M1 and M3), Fb = [lo | hi] (M2)
    const unsigned a_vo0 = r31 * 16 + hh * T::VTERM, a_vo1 = r31 * 16 + (1 - hh) * 2 * T::VTERM;
    const int wpos0 = 18 * ri + 3 * ki;               // first position of this wave's block
    auto pos_off = [](int p9) { return (6 * (p9 / 3) + (p9 % 3)) * T::VPOS; };   // position p9 of the block, relative to wpos0

    // weight stream: unit index of this wave = (item t * 16 + chunk c) * 9 + p9
    static constexpr int RING = 6;                    // weight units in registers (5 in flight); 18 units per two chunks
    bf16x8 af[RING][2];
    auto unit_load = [&](int t, int c, int p9, int slot) {
        // (c, p9) may run past the end of the item: carry into the next item; past the last item: reload the last unit
        if (p9 >= 9) { p9 -= 9; c += 1; }
        if (c >= T::NCHUNK) { c -= T::NCHUNK; t += 1; }
        if (t >= nitems) { t = nitems - 1; c = T::NCHUNK - 1; p9 = 8; }
        const int cog = 2 * item_half(t) + cc;
        const unsigned so = (unsigned)(((cog * 4 + blk) * T::NCHUNK + c) * 9 + p9) * T::UNIT;
        af[slot][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_u, a_vo0, so, 0));
        af[slot][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_u, a_vo1, so, 0));
    };

    // Everything that depends on the row half of the transform role is instantiated twice (wave-uniform branch below)
    auto run = [&](auto PH) {
        constexpr int ph = decltype(PH)::value;
        // ---- transform role: board tb, row half ph, channels 4 ch4 .. 4 ch4 + 3.  lane -> (tile row tty, tile column
        // ttx, channel).  Bit 0 = channel parity e: the two channels of a pair are DPP neighbours.  The four 16-lane
        // groups the LDS serves a ds_read_b128 in ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32) each hold ONE
        // tile row and all sixteen (channel, ttx): with the plane stride of 272 floats their reads fall on sixteen
        // different bank quads.
        const int tb = wave & 1, ch4 = wave >> 2;
        const int e = lane & 1, run4 = (lane >> 2) & 7;
        const int ttx = 2 * ((run4 >> 1) & 1) + ((lane >> 1) & 1), cpl = run4 >> 2;
        const int tty = 2 * (lane >> 5) + 1 - ((0x69 >> run4) & 1);
        const int tile = 4 * tty + ttx, chl = 4 * ch4 + 2 * cpl + e;          // channel of the chunk (0..7)
        const int tr_off = T::RFRONT + (tb * 8 + chl) * T::RPS + (4 * tty - 1 + ph) * T::RROW + 4 * ttx;
        const unsigned col16_mask = ttx == 3 ? 0u : 0xffffffffu;   // column 16 does not exist: the word there is column 0 of the next row
        // bytes: column tb * 16 + tile, dword = channel pair; even lanes pack positions k = 0..2 of a row, odd lanes 3..5
        const int tv_off = tb * 256 + tile * 16 + (2 * ch4 + cpl) * 4 + e * 3 * T::VPOS;
        // v_perm_b32 selector: dword = (pair's even channel: low half, odd channel: high half) of the upper 16 bits of
        // (mine, partner's): even lanes (mine = even channel) {partner[3], partner[2], mine[3], mine[2]}, odd lanes swapped
        const unsigned psel = e ? 0x03020706u : 0x07060302u;
        float xr[5][4];                                // five patch rows of the channel: four columns at a time
        float tt[3][6];                                // row-pass results (rows 3 ph .. 3 ph + 2), columns -1 .. 4
        float oo[6];
        // B^T over the rows, elementwise in the columns -- first the four centre columns (one 16-byte read per row), then
        // the two halo columns (-1 and 4)
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
                    xr[i][1] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, rp[i * T::RROW + 4]) & col16_mask);   // (no branch)
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
        // my value and my pair partner's value of one position -> three dwords of bf16 pairs: hi = the upper halves of both
        // (truncation), remainder = value - hi (exact), ...: hi + mid + lo == value, bit for bit
        auto emit = [&](char* vp, float mine, float theirs) {
            const unsigned um = __builtin_bit_cast(unsigned, mine), ut = __builtin_bit_cast(unsigned, theirs);
            const unsigned h = __builtin_amdgcn_perm(ut, um, psel);
            const float rm = mine - __builtin_bit_cast(float, um & 0xffff0000u), rt = theirs - __builtin_bit_cast(float, ut & 0xffff0000u);
            const unsigned urm = __builtin_bit_cast(unsigned, rm), urt = __builtin_bit_cast(unsigned, rt);
            const unsigned m = __builtin_amdgcn_perm(urt, urm, psel);
            const float sm = rm - __builtin_bit_cast(float, urm & 0xffff0000u), st = rt - __builtin_bit_cast(float, urt & 0xffff0000u);
            const unsigned l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, st), __builtin_bit_cast(unsigned, sm), psel);
            *reinterpret_cast<unsigned*>(vp) = h;
            *reinterpret_cast<unsigned*>(vp + T::VTERM) = m;
            *reinterpret_cast<unsigned*>(vp + 2 * T::VTERM) = l;
        };
        // The transform of one chunk (raw[rpar] -> V[vpar], this thread's channel, rows 3 ph .. 3 ph + 2) in 18 slices, two
        // per MFMA slot of a chunk body
        float mine3[3], theirs3[3];
        auto tslice = [&](int rpar, int vpar, auto KK) {
            constexpr int K = decltype(KK)::value;
            const float* rp = rawb + rpar * T::RAW_FLOATS + tr_off;
            char* vp = vbase + vpar * T::V_BYTES + tv_off;
            if constexpr (K == 0) row_pass(rp, std::integral_constant<int, 0>{});
            else if constexpr (K == 1) row_pass(rp, std::integral_constant<int, 1>{});
            else if constexpr (K >= 3 && K < 18) {
                constexpr int ii = (K - 3) / 5, part = (K - 3) % 5;
                if constexpr (part == 0) col_pass(tt[ii], oo);
                else if constexpr (part == 1) {
                    // the three values I pack myself and the three my partner packs; the partner's come over by DPP
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        mine3[k] = e ? oo[k + 3] : oo[k];
                        const float send = e ? oo[k] : oo[k + 3];
                        theirs3[k] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                    }
                } else {
                    constexpr int k = part - 2;
                    emit(vp + ((3 * ph + ii) * 6 + k) * T::VPOS, mine3[k], theirs3[k]);
                }
            }
        };
#define APZB_ALL18(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15) F(16) F(17)
        auto transform = [&](int rpar, int vpar) {
#define APZB_TS(k) tslice(rpar, vpar, std::integral_constant<int, k>{});
            APZB_ALL18(APZB_TS)
#undef APZB_TS
        };

        // zero halo rows of both raw buffers (the DMA never touches them), once
        for (int i = tid * 4; i < 2 * T::RAW_FLOATS; i += 2048) *reinterpret_cast<f32x4*>(&lds[i]) = f32x4{0.f, 0.f, 0.f, 0.f};
        // the first RING - 1 weight units
#pragma unroll
        for (int u = 0; u < RING - 1; u++) unit_load(0, 0, u, u);
        __syncthreads();
        raw_dma(0, 0, 0);                              // the first item's first two chunks (later items: from the epilogue before)
        raw_dma(0, 1, 1);

        for (int t = 0; t < nitems; t++) {
            const int h = item_half(t);
            const int bd0 = 2 * item_pair(t);
            const bool two = bd0 + 1 < n;
            // ---- item prologue: raw(0), raw(1) have been requested; V[0] = transform(raw(0))
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            transform(0, 0);
            f32x16 acc[9];
#pragma unroll
            for (int p = 0; p < 9; p++)
#pragma unroll
                for (int v = 0; v < 16; v++) acc[p][v] = 0.f;
            APZB_STAMP(0)

            // ---- chunk loop.  Iteration c: [barrier] DMA of raw(c+2) -> raw[c & 1] (read last by the transform of iteration
            // c - 1); transform of raw[(c+1) & 1] -> V[(c+1) & 1]; MFMAs over V[c & 1]: 9 slots = the wave's 9 positions,
            // each slot 3 MFMAs + two slices of the transform + the refill of the weight ring slot freed by the previous slot.
            // (measurement builds of tools/wino3b_bench.hip: the chunk body without its transform / weight loads / fragment reads)
#ifndef APZB_ABL_T
#define APZB_ABL_T 0
#endif
#ifndef APZB_ABL_W
#define APZB_ABL_W 0
#endif
#ifndef APZB_ABL_D
#define APZB_ABL_D 0      /* no LDS-DMA of the next chunks' planes inside the chunk loop (stale tiles: timing only) */
#endif
#ifndef APZB_ABL_B
#define APZB_ABL_B 0
#endif
#if APZB_ABL_T
#define APZB_TSLICE(k)
#else
#define APZB_TSLICE(k) tslice(1 - par, 1 - par, std::integral_constant<int, (k)>{});
#endif
#if APZB_ABL_W
#define APZB_ULOAD(k)
#else
#define APZB_ULOAD(k) unit_load(t, c, (k) + RING - 1, (par * 9 + (k) + RING - 1) % RING);
#endif
#ifndef APZB_PRIO
#define APZB_PRIO 1
#endif
#if APZB_PRIO
            // the later-dispatched half of the workgroup (waves 4..7, the SIMD partners of 0..3) loses every issue
            // arbitration by age (MI355X_MICROARCH.md, "Two waves per SIMD", item 4): one static priority raise evens it out
            if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
            bf16x8 bfr[3];
            auto chunk = [&](int c, auto PAR) {
                constexpr int par = decltype(PAR)::value;
                __syncthreads();                      // V[par] and raw[1 - par] complete; V[1 - par] and raw[par] free
                APZB_STAMP(1)
                const char* vp = vbase + par * T::V_BYTES;
                // per-lane fragment offsets rebuilt from an opaque copy of the lane id (kept live across the kernel they are
                // what hipcc spills, and every scratch reload is followed by vmcnt(0): a full drain of the weight ring)
                int le = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                asm volatile("" : "+v"(le));
                const int b_lo0 = wpos0 * T::VPOS + (le & 31) * 16;                  // M1: B = [hi | hi]
                const int b_lo1 = b_lo0 + (le >> 5) * T::VTERM;                       // M2: B = [hi | mid]
                const int b_lo2 = b_lo0 + (2 - (le >> 5)) * T::VTERM;                 // M3: B = [lo | mid]
                bfr[0] = *reinterpret_cast<const bf16x8*>(vp + b_lo0);
                bfr[1] = *reinterpret_cast<const bf16x8*>(vp + b_lo1);
                bfr[2] = *reinterpret_cast<const bf16x8*>(vp + b_lo2);
#define APZB_SLOT(k)                                                                                                     \
                {                                                                                                        \
                    constexpr int p9 = (k), slot = (par * 9 + (k)) % RING;                                               \
                    /* smallest products first: M3 = hi.lo + mid.mid, M2 = lo.hi + hi.mid, M1 = hi.hi + mid.hi */        \
                    /* every V fragment is re-read for the next position right behind the one MFMA that uses it: the      \
                       fragment the next slot needs first (M3's) is requested two MFMA times ahead */                      \
                    acc[p9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[slot][0], bfr[2], acc[p9], 0, 0, 0);            \
                    if (p9 + 1 < 9 && !APZB_ABL_B) bfr[2] = *reinterpret_cast<const bf16x8*>(vp + b_lo2 + pos_off(p9 + 1)); \
                    acc[p9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[slot][1], bfr[1], acc[p9], 0, 0, 0);            \
                    if (p9 + 1 < 9 && !APZB_ABL_B) bfr[1] = *reinterpret_cast<const bf16x8*>(vp + b_lo1 + pos_off(p9 + 1)); \
                    acc[p9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[slot][0], bfr[0], acc[p9], 0, 0, 0);            \
                    if (p9 + 1 < 9 && !APZB_ABL_B) bfr[0] = *reinterpret_cast<const bf16x8*>(vp + b_lo0 + pos_off(p9 + 1)); \
                    if ((k) == 1 && !APZB_ABL_D) raw_dma(t, c + 2, par);                                                 \
                    APZB_TSLICE(2 * (k))                                                                                 \
                    APZB_TSLICE(2 * (k) + 1)                                                                             \
                    /* unit k + RING - 1 goes into the ring slot of unit k - 1, whose MFMAs are done */                 \
                    APZB_ULOAD(k)                                                                                        \
                    __builtin_amdgcn_sched_barrier(0);                                                                   \
                }
                APZB_SLOT(0) APZB_SLOT(1) APZB_SLOT(2) APZB_SLOT(3) APZB_SLOT(4) APZB_SLOT(5) APZB_SLOT(6) APZB_SLOT(7) APZB_SLOT(8)
#undef APZB_SLOT
                APZB_STAMP(2)
            };
            for (int c = 0; c < T::NCHUNK; c += 2) {
                chunk(c, std::integral_constant<int, 0>{});
                chunk(c + 1, std::integral_constant<int, 1>{});
            }

            // ---- epilogue: four steps of 16 output channels (32-channel half cs, quarter q2); the waves of half cs hold
            // the step's accumulators.  Layout of the 32 x 32 tile: lane (col = lane & 31, hh = lane >> 5), register v:
            // channel (v & 3) + 8 (v >> 2) + 4 hh.
            const int cosel = lane >> 5;               // gather role: channel 2 wave + cosel of the step's 16, column lane & 31
            const int col = lane & 31, gbd = col >> 4, gtile = col & 15;
            const int gty = gtile >> 2, gtx = gtile & 3;
            float* sw = stg + wave * (4 * T::SPLANE);
            const int s_lin = (lane >> 2) * T::SROW + (lane & 3) * 4;
            const unsigned ep_vo = lane < 60 ? lane * 16 : 0x80000000u;
            auto ep_step = [&](auto S_) {
                constexpr int s = decltype(S_)::value;
                constexpr int cs = s >> 1, q2 = s & 1;
                const int co_base = (2 * h + cs) * 32 + 16 * q2;         // first output channel of the step
                __syncthreads();                       // MFMAs over V done (s = 0) / M and staging of the previous step consumed
                APZB_STAMP(1)
                if (s == 0 && t + 1 < nitems) {        // the raw tiles are free: the next item's first two chunks
                    raw_dma(t + 1, 0, 0);
                    raw_dma(t + 1, 1, 1);
                }
                // residual planes of this wave (2 channels x 2 boards), requested before the accumulators move
                f32x4 rs[4];
                if (RESID) {
#pragma unroll
                    for (int pl = 0; pl < 4; pl++) {
                        const int bdp = bd0 + (pl & 1);
                        const int bd = bdp < n ? bdp : n - 1;
                        rs[pl] = bload(r_res, ep_vo, (unsigned)(bd * T::C + co_base + 2 * wave + (pl >> 1)) * plane_b);
                    }
                }
                if (cc == cs) {
                    float* mw = mq + wpos0 * 512 + (4 * hh) * 32 + r31;
#pragma unroll
                    for (int p9 = 0; p9 < 9; p9++)
#pragma unroll
                        for (int ee = 0; ee < 8; ee++)
                            mw[(6 * (p9 / 3) + p9 % 3) * 512 + ((ee & 3) + 8 * (ee >> 2)) * 32] = acc[p9][8 * q2 + ee];
                }
                if (RESID) {
#pragma unroll
                    for (int pl = 0; pl < 4; pl++) *reinterpret_cast<f32x4*>(sw + pl * T::SPLANE + s_lin) = rs[pl];
                }
                __syncthreads();                       // M complete
                APZB_STAMP(1)
                {
                    const int co16 = 2 * wave + cosel;
                    const float* mp = mq + co16 * 32 + col;
                    float hrow[6][4];                   // the k-direction transform of every row
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        float m[6];
#pragma unroll
                        for (int k = 0; k < 6; k++) m[k] = mp[(6 * i + k) * 512];
                        const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
                        hrow[i][0] = (m[0] + s12) + s34;
                        hrow[i][1] = __builtin_fmaf(2.f, d34, d12);
                        hrow[i][2] = __builtin_fmaf(4.f, s34, s12);
                        hrow[i][3] = __builtin_fmaf(8.f, d34, d12) + m[5];
                    }
                    const float bv = bias[co_base + co16];
                    float* sp = sw + (cosel * 2 + gbd) * T::SPLANE + (4 * gty) * T::SROW + 4 * gtx;
                    f32x4 y[4];
#pragma unroll
                    for (int ee = 0; ee < 4; ee++) {
                        const float s12 = hrow[1][ee] + hrow[2][ee], d12 = hrow[1][ee] - hrow[2][ee];
                        const float s34 = hrow[3][ee] + hrow[4][ee], d34 = hrow[3][ee] - hrow[4][ee];
                        y[0][ee] = (hrow[0][ee] + s12) + s34;
                        y[1][ee] = __builtin_fmaf(2.f, d34, d12);
                        y[2][ee] = __builtin_fmaf(4.f, s34, s12);
                        y[3][ee] = __builtin_fmaf(8.f, d34, d12) + hrow[5][ee];
                    }
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        f32x4 v = y[a] + bv;
                        if (RESID) v += *reinterpret_cast<const f32x4*>(sp + a * T::SROW);   // (wave-private: written above by this wave)
#pragma unroll
                        for (int ee = 0; ee < 4; ee++) v[ee] = RELU ? fmaxf(v[ee], 0.f) : v[ee];
                        if (gtx == 3) v[3] = 0.f;      // column 15 is the halo column of the rows16 layout
                        *reinterpret_cast<f32x4*>(sp + a * T::SROW) = v;   // (same lane, same addresses as the residual it read)
                    }
                }
                wave_lds_fence();
#pragma unroll
                for (int pl = 0; pl < 4; pl++) {
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(sw + pl * T::SPLANE + s_lin);
                    const unsigned vo = ((pl & 1) == 0 || two) ? ep_vo : 0x80000000u;   // the missing second board of an odd batch
                    bstore(r_out, vo, (unsigned)((bd0 + (pl & 1)) * T::C + co_base + 2 * wave + (pl >> 1)) * plane_b, pv);
                }
                APZB_STAMP(3)
            };
            ep_step(std::integral_constant<int, 0>{});
            ep_step(std::integral_constant<int, 1>{});
            ep_step(std::integral_constant<int, 2>{});
            ep_step(std::integral_constant<int, 3>{});
            // (the next item's prologue starts with a barrier: M / staging are consumed before its transform writes V)
        }
    };
    if (((wave >> 1) & 1) == 0)
        run(std::integral_constant<int, 0>{});
    else
        run(std::integral_constant<int, 1>{});
#ifdef APZ_WINO3B_STAMPS
    st_acc[7] = __builtin_readcyclecounter() - st_t0;
    if (lane == 0 && blockIdx.x < 4)
        for (int i = 0; i < 8; i++) apz_wino3b_stamps[(blockIdx.x * 8 + wave) * 8 + i] = st_acc[i];
#endif
}

}  // namespace apz
