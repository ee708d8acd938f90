// Does vector work hide behind v_mfma_f32_32x32x16_bf16 when ONE wave per SIMD issues both (trunk15_wino3b.h's regime)?
// Loop of 18 MFMAs in dependent chains of three (same accumulator, as the kernel issues them) with K independent vector
// instructions after each: v_fma_f32 (VALU=0), v_cvt_pk_bf16_f32 + v_and + v_sub (VALU=1: the split's mix), ds_read_b128 (VALU=2).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    __shared__ f32x4 sm[1024];
    f32x16 acc[6];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
    bf16x8 av, bv;
    for (int i = 0; i < 8; i++) { av[i] = (__bf16)(a + i); bv[i] = (__bf16)(b - i); }
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = a + i;
    sm[threadIdx.x] = f32x4{a, b, a, b};
    __syncthreads();
    f32x4 ld[4] = {};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 18; m++) {
            acc[m / 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[m / 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(m + k) & 7]) : "v"(b), "v"(a));
                if (KIND == 1) {
                    if (k % 3 == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[(m + k) & 7]) : "v"(b), "v"(a));
                    if (k % 3 == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[(m + k) & 7]));
                    if (k % 3 == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[(m + k) & 7]) : "v"(b));
                }
                if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[k & 3]) : "v"((unsigned)(threadIdx.x * 16)));
            }
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 6; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
    for (int i = 0; i < 8; i++) s += v[i];
    for (int i = 0; i < 4; i++) s += ld[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int KIND>
void run(float* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 3000;
    hipLaunchKernelGGL((probe<K, KIND>), dim3(256), dim3(256), 0, 0, out, 50);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((probe<K, KIND>), dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ns = ms * 1e6 / (iters * 18.0);
    printf("kind %d K=%2d: %6.2f ns per MFMA 32x32x16 (+%d instr) = %7.1f TFLOP/s\n", KIND, K, ns, K, 32768.0 / ns * 1024 / 1e3);
}

int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    run<0, 0>(out); run<2, 0>(out); run<4, 0>(out); run<5, 0>(out); run<6, 0>(out); run<8, 0>(out); run<12, 0>(out);
    run<3, 1>(out); run<6, 1>(out); run<9, 1>(out);
    run<1, 2>(out); run<2, 2>(out); run<3, 2>(out);
    return 0;
}
