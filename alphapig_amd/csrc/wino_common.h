// Shared pieces of the Winograd F(4x4,3x3) trunk kernels (trunk15_wino3.h, wgrad_wino3.h) for gfx950: vector types,
// the packed FMA, the wavefront-scope LDS fence and the packed-weight geometry.
#pragma once
#include <hip/hip_runtime.h>

#include "conv3x3_mfma.h"

namespace apz {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Transformed weights U = G g G^T of one 128 -> 128 layer, packed per MFMA A fragment:
// [cot 8][row half 2][c4 32][lane 64][20] -- lane (q = lane>>4, j = lane&15) holds U[row 3*half + ii][k] at index
// 6*ii + k of co = cot*16 + j, ci = c4*4 + q (18 values + 2 pad: four 16-byte loads + one 8-byte load per k-step).
struct WinoPack {
    static constexpr int UROW = 20;                    // floats per lane and k-step
    static constexpr size_t UPK_FLOATS = (size_t)8 * 2 * 32 * 64 * UROW;   // per layer (2.6 MB)
};

// The same values for the small-batch kernel (trunk15_wino3s.h), whose MFMA wave w owns positions 9 w .. 9 w + 8:
// [cot 8][w 4][c4 32][piece 3][lane 64][4] -- value m = 4 piece + e of the wave's nine (pad: 0), so that one k-step of a
// wave is three fully coalesced 1 KB loads (from the layout above it was nine loads touching forty 128-byte lines each).
struct WinoPackSmall {
    static constexpr int STEP = 3 * 64 * 4;            // floats per (cot, w, c4)
    static constexpr size_t UPK_FLOATS = (size_t)8 * 4 * 32 * STEP;   // per layer (3.1 MB)
    __host__ __device__ static size_t index(int co, int ci, int pos) {
        const int cot = co >> 4, jj = co & 15, c4 = ci >> 2, qq = ci & 3, w = pos / 9, m = pos % 9;
        return ((((size_t)(cot * 4 + w) * 32 + c4) * 3 + (m >> 2)) * 64 + (qq * 16 + jj)) * 4 + (m & 3);
    }
};

__device__ __forceinline__ f32x2 fma2(const float a, const f32x2 b, const f32x2 c) {   // a*b + c (v_pk_fma_f32)
    return __builtin_elementwise_fma(f32x2{a, a}, b, c);
}

// Lanes of ONE wave exchange data through LDS (write in one layout, read in another).  The LDS executes a
// wave's instructions in order, so no s_barrier is needed -- but the compiler reasons per thread and may move
// a lane's read above its own (provably different-address) write.  This pins the order for the compiler.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace apz
