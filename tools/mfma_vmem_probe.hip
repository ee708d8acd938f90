// Do operand deliveries into VGPRs (buffer loads from L2, ds_read_b128 from LDS) overlap with v_mfma_f32_32x32x16_f16 on the
// same SIMD?  One 512-thread workgroup per CU (two waves per SIMD), every wave runs ITER iterations of
//   NL buffer_load_dwordx4 (1 KB per wave each, L2-resident stream) + ND ds_read_b128 + NM MFMAs (operands: registers that
//   no load of the iteration writes; the loaded data is only consumed by a dummy asm at the end of the iteration)
// and reports cycles per iteration for the three arms (loads only, MFMAs only, both).  gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_vmem_probe.hip -o tools/_build/mfma_vmem_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NL, int ND, int NM>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ w, float* __restrict__ out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 512) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 8u << 20, 0x00020000);
    f16x8 a = {(_Float16)1.f, (_Float16)0.5f, 0, 0, 0, 0, 0, (_Float16)(lane * 0.01f)}, b = {(_Float16)0.25f, 0, 0, (_Float16)(lane * 0.02f), 0, 0, 0, 0};
    f32x16 acc[4] = {};
    f32x4 ld[NL > 0 ? NL : 1], dd[ND > 0 ? ND : 1];
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned so = (blockIdx.x & 7) * (1u << 20) + wave * 65536u;     // each wave its own 64 KB stream inside an XCD's 1 MB
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NL; i++) ld[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, so + ((it * NL + i) & 63) * 1024u, 0));
#pragma unroll
        for (int i = 0; i < ND; i++) dd[i] = *reinterpret_cast<const f32x4*>(&lds[((wave * ND + i) * 256 + lane * 4) & 16383]);
#pragma unroll
        for (int i = 0; i < NM; i++) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NL; i++) asm volatile("" ::"v"(ld[i]));
#pragma unroll
        for (int i = 0; i < ND; i++) asm volatile("" ::"v"(dd[i]));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int k = 0; k < 4; k++)
        for (int v = 0; v < 16; v++) s += acc[k][v];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NL, int ND, int NM>
void run(const float* w, float* out, unsigned long long* cyc, const char* name) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NL, ND, NM>), dim3(256), dim3(512), 0, 0, w, out, cyc, 200);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NL, ND, NM>), dim3(256), dim3(512), 0, 0, w, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s loads %d ds_reads %d mfma %d per iteration: %.0f cycles / iteration (%.1f ns); per CU: %.1f B/clk from L2, MFMA pipe %.2f\n", name, NL, ND, NM,
           (double)c / iters, ms * 1e6 / iters, 8.0 * NL * 1024 / ((double)c / iters), NM * 2 * 32.0 / ((double)c / iters));
}

int main() {
    float *w, *out;
    unsigned long long* cyc;
    hipMalloc(&w, 8u << 20); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    hipMemset(w, 0, 8u << 20);
    run<8, 0, 0>(w, out, cyc, "loads only");
    run<0, 0, 16>(w, out, cyc, "mfma only");
    run<8, 0, 16>(w, out, cyc, "loads + mfma");
    run<4, 0, 16>(w, out, cyc, "half loads + mfma");
    run<0, 16, 0>(w, out, cyc, "ds_read only");
    run<0, 16, 16>(w, out, cyc, "ds_read + mfma");
    run<8, 16, 16>(w, out, cyc, "loads + ds_read + mfma");
    run<8, 16, 0>(w, out, cyc, "loads + ds_read");
    run<16, 0, 16>(w, out, cyc, "2x loads + mfma");
    run<16, 0, 0>(w, out, cyc, "2x loads only");
    return 0;
}
