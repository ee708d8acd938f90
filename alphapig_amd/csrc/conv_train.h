// Training-side convolution primitives for gfx950 (first hand-written part):
//
//   pack_conv3x3_kernel   raw [Cout][Cin][3][3] weights (device) -> the MFMA fragment layout of
//                         conv3x3_mfma_kernel; with `transpose_flip` it packs the weights of the
//                         data-gradient convolution  dX = conv(dY, W'),  W'[ci][co][ky][kx] =
//                         W[co][ci][2-ky][2-kx]  (so forward and dgrad share ONE kernel).
//   conv3x3_wgrad_kernel  dW[co][ci][ky][kx] += sum over boards and pixels of
//                         dY[b][co][y][x] * X[b][ci][y+ky-1][x+kx-1]
//                         as fp32 MFMA: D[16 co][16 ci] += A[16 co][4 px] * B[4 px][16 ci] per tap.
//
// wgrad mapping: a workgroup owns ONE 16-channel tile of C_out and up to 64 input channels (one
// 16-channel tile per wave) for a slice of the batch; a wave keeps 9 accumulator tiles (one per
// tap, 36 registers).  Per board it stages dY[16][H*W] and X[64][(H+2) x (W+1) padded] in LDS; a
// k-step is 4 consecutive pixels: the A fragment (dY) is read once per k-step and reused by the
// nine taps, the B fragment is the input at the tap's shift (zero halo in the padded tile).
// Plane strides are odd multiples chosen so the 16 channel lanes of a fragment hit 16 distinct
// banks.  Partial sums of the batch slices are combined with float atomics (dW is zeroed by the
// launcher), so the summation order -- and the last bits -- vary from run to run.
#pragma once
#include <hip/hip_runtime.h>

#include "conv3x3_mfma.h"

namespace apz {

// w [cout][cin][3][3] -> wpk [cout'/16][cin'_pad/4][9][64] with (cout', cin') = (cout, cin) or,
// transposed, (cin, cout).  One thread per packed element.
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ wpk, int cin, int cout,
                                    int transpose_flip) {
    const int co_n = transpose_flip ? cin : cout;       // output channels of the packed conv
    const int ci_n = transpose_flip ? cout : cin;       // its input channels
    const int n4 = (ci_n + 3) / 4, ncot = co_n / 16;
    const long total = (long)ncot * n4 * 9 * 64;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        long t = i >> 6;
        const int tap = (int)(t % 9);
        t /= 9;
        const int c4 = (int)(t % n4);
        const int cot = (int)(t / n4);
        const int o = cot * 16 + (lane & 15), c = c4 * 4 + (lane >> 4);
        float v = 0.f;
        if (c < ci_n) {
            if (!transpose_flip) {
                v = w[((size_t)o * cin + c) * 9 + tap];
            } else {
                const int ky = tap / 3, kx = tap - ky * 3;
                v = w[((size_t)c * cin + o) * 9 + (2 - ky) * 3 + (2 - kx)];   // W[co=c][ci=o][2-ky][2-kx]
            }
        }
        wpk[i] = v;
    }
}

template <int H, int W>
struct WgradGeo {
    static constexpr int HW = H * W;
    static constexpr int RS = W + 1;
    static constexpr int XPLANE = (H + 2) * RS + 2;
    static constexpr int XPS = XPLANE | 1;                          // odd: 16 channel lanes -> 16 banks
    static constexpr int KSTEPS = (HW + 3) / 4;
    static constexpr int YPS = (KSTEPS * 4) | 1;                    // odd, >= padded pixel count
    static constexpr int LDS_FLOATS = 64 * XPS + 16 * YPS + 64;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

// x [n][cin][H][W], dy [n][cout][H][W] dense; dw [cout][cin][3][3] (pre-zeroed, atomically added).
// grid: (cout/16, ceil(cin/64), batch slices)
template <int H, int W>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dw, int n, int cin, int cout) {
    using G = WgradGeo<H, W>;
    constexpr int HW = G::HW;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xt = sm + 32;                    // [64][XPS], origin shifted so (row -1, col -1) is in bounds
    float* yt = sm + 32 + 64 * G::XPS;      // [16][YPS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int cot = blockIdx.x;
    const int ci0 = blockIdx.y * 64;
    const int nci = min(64, cin - ci0);     // input channels of this workgroup (may be < 64)

    for (int i = tid; i < G::LDS_FLOATS; i += 256) sm[i] = 0.f;   // halos / unused channels / tail pixels stay 0

    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* arow = yt + j * G::YPS + q;                 // A: co = j, pixel = 4s + q
    const float* brow = xt + (wave * 16 + j) * G::XPS;       // B: ci = wave*16 + j

    for (int b = blockIdx.z; b < n; b += gridDim.z) {
        __syncthreads();
        const float* xb = x + ((size_t)b * cin + ci0) * HW;
        for (int idx = tid; idx < nci * HW; idx += 256) {
            const int c = idx / HW, rem = idx - c * HW;
            const int yy = rem / W, xx = rem - yy * W;
            xt[c * G::XPS + (yy + 1) * G::RS + xx + 1] = xb[idx];
        }
        const float* yb = dy + ((size_t)b * cout + cot * 16) * HW;
        for (int idx = tid; idx < 16 * HW; idx += 256) {
            const int c = idx / HW, rem = idx - c * HW;
            yt[c * G::YPS + rem] = yb[idx];
        }
        __syncthreads();
        if (wave * 16 < nci) {               // wave-uniform: this wave's 16 input channels exist
            for (int s = 0; s < G::KSTEPS; s++) {
                const float a = arow[4 * s];
                int p = 4 * s + q;
                p = p < HW ? p : HW - 1;     // tail lanes: dY there is 0, any in-bounds address will do
                const int yy = p / W, xx = p - yy * W;
                const float* bp = brow + yy * G::RS + xx;    // (yy + ky) * RS + xx + kx, origin (-1,-1)
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++)
                        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[ky * G::RS + kx], acc[ky * 3 + kx],
                                                                                0, 0, 0);
            }
        }
    }
    // D lane l, reg r: co = cot*16 + 4q + r, ci = ci0 + wave*16 + j
    const int ci = ci0 + wave * 16 + j;
    if (wave * 16 < nci && ci < cin) {
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int co = cot * 16 + q * 4 + r;
                atomicAdd(dw + ((size_t)co * cin + ci) * 9 + t, acc[t][r]);
            }
    }
}

}  // namespace apz
