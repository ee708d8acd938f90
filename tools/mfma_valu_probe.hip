// Probe: how many independent VALU instructions fit in the shadow of a v_mfma_f32_16x16x4_f32?
// Loop of 16 MFMAs (independent accumulators) with K VALU FMAs after each; 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int DEP>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = a + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) {
            // DEP: the MFMA's A operand is the VALU result just produced (as in the on-the-fly weight transform)
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(DEP ? v[m & 7] : a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(m + k) & 7]) : "v"(b), "v"(a));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int DEP>
void run(float* out, int threads) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    hipLaunchKernelGGL((probe<K, DEP>), dim3(256), dim3(threads), 0, 0, out, 10);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((probe<K, DEP>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waves_per_simd = threads / 256.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * 16.0 * waves_per_simd);   // SIMD cycles per MFMA
    printf("K=%d dep=%d waves/SIMD=%.0f: %.1f cycles per MFMA (+%d VALU)\n", K, DEP, waves_per_simd, cyc, K);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    for (int threads = 256; threads <= 512; threads += 256) {
        run<0, 0>(out, threads); run<1, 0>(out, threads); run<2, 0>(out, threads); run<3, 0>(out, threads);
        run<4, 0>(out, threads); run<6, 0>(out, threads); run<8, 0>(out, threads); run<12, 0>(out, threads);
        run<2, 1>(out, threads); run<4, 1>(out, threads);
    }
    return 0;
}
