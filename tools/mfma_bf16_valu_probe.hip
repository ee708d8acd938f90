// Probe for the next design step (HISTORY.md section 9, "fp32-accurate split on the bf16 matrix pipe"): does VALU work
// hide in the shadow of v_mfma_f32_16x16x32_bf16 on gfx950 (it does NOT under v_mfma_f32_16x16x4_f32:
// tools/mfma_valu_probe.hip), and what does one such MFMA cost per wave with 1 / 2 waves per SIMD?
// Loop of 16 MFMAs (independent accumulators) with K `v_fma_f32` after each.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, bool BF16>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
    bf16x8 av, bv;
    for (int i = 0; i < 8; i++) { av[i] = (__bf16)(a + i); bv[i] = (__bf16)(b - i); }
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = a + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) {
            if (BF16) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[m], 0, 0, 0);
            else acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(m + k) & 7]) : "v"(b), "v"(a));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, bool BF16>
void run(float* out, int threads) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000;
    hipLaunchKernelGGL((probe<K, BF16>), dim3(256), dim3(threads), 0, 0, out, 50);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((probe<K, BF16>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waves_per_simd = threads / 256.0;
    const double ns = ms * 1e6 / (iters * 16.0 * waves_per_simd);   // SIMD time per MFMA (+ its K VALU)
    const double flop = BF16 ? 2.0 * 16 * 16 * 32 : 2.0 * 16 * 16 * 4;
    printf("%s K=%2d waves/SIMD=%.0f: %6.2f ns per MFMA (+%d VALU) = %7.1f TFLOP/s on 256 CUs x 4 SIMDs\n", BF16 ? "bf16 16x16x32" : "f32  16x16x4 ", K,
           waves_per_simd, ns, K, flop / ns * 1024 / 1e3);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    for (int threads = 256; threads <= 512; threads += 256) {
        run<0, false>(out, threads); run<4, false>(out, threads); run<8, false>(out, threads);
        run<0, true>(out, threads); run<1, true>(out, threads); run<2, true>(out, threads); run<4, true>(out, threads);
        run<8, true>(out, threads); run<16, true>(out, threads);
    }
    return 0;
}
