// Timing of stem15_kernel<1, 4> at the north_star shape (8192 x 4 x 15 x 15 -> 128 channels) with the experiment
// switches of trunk15_ring.h (tools/: measurement aid).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ialphapig_amd/csrc -DAPZ_STEM_OCC=2 -DAPZ_STEM_SCH=16 -DAPZ_STEM_CTB=2 \
//         tools/stem_bench.hip -o tools/_build/stem_bench_a
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "trunk15_ring.h"

int main() {
    const int n = 8192;
    float *in, *wpk, *bias, *out;
    hipMalloc(&in, (size_t)n * 4 * 225 * 4);
    hipMalloc(&wpk, 8 * 1 * 9 * 64 * 4);
    hipMalloc(&bias, 128 * 4);
    hipMalloc(&out, (size_t)n * 128 * 240 * 4);
    std::vector<float> h((size_t)n * 4 * 225);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u >> 8) & 1);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> w(8 * 9 * 64, 0.01f), bv(128, 0.1f);
    hipMemcpy(wpk, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, bv.data(), bv.size() * 4, hipMemcpyHostToDevice);
    constexpr int lds = apz::stem15_lds_bytes<1>();
    hipFuncSetAttribute((const void*)apz::stem15_kernel<1, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int mult : {2, 3, 4}) {
        float best = 1e9, sum = 0;
        for (int it = 0; it < 14; it++) {
            hipEventRecord(a);
            hipLaunchKernelGGL((apz::stem15_kernel<1, 4, false>), dim3(256 * mult), dim3(256), lds, 0, in, wpk, bias, out, n, 4, 0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (it >= 4) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("OCC %d SCH %d CTB %d grid %4d: best %.1f us  mean %.1f us  (%.3f of 8 TB/s by the mean)\n", APZ_STEM_OCC, APZ_STEM_SCH,
               APZ_STEM_CTB, 256 * mult, best * 1e3, sum / 10 * 1e3, 973228032.0 / (sum / 10 * 1e-3) / 8e12);
    }
    // the real stem (C_in = 9, matrix-pipe-bound), at the north_star batch and at the self-play path's 512 boards
    {
        float *in9, *w9;
        hipMalloc(&in9, (size_t)n * 9 * 225 * 4);
        hipMalloc(&w9, 8 * 3 * 9 * 64 * 4);
        hipMemset(in9, 0, (size_t)n * 9 * 225 * 4);
        std::vector<float> ww(8 * 3 * 9 * 64, 0.01f);
        hipMemcpy(w9, ww.data(), ww.size() * 4, hipMemcpyHostToDevice);
        constexpr int lds9 = apz::stem15_lds_bytes<3>();
        hipFuncSetAttribute((const void*)apz::stem15_kernel<3, 9, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds9);
        for (int nb : {8192, 512}) {
            float best = 1e9, sum = 0;
            for (int it = 0; it < 14; it++) {
                hipEventRecord(a);
                hipLaunchKernelGGL((apz::stem15_kernel<3, 9, false>), dim3(512), dim3(256), lds9, 0, in9, w9, bias, out, nb, 9, 0);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (it >= 4) { best = ms < best ? ms : best; sum += ms; }
            }
            printf("C_in 9, CTB %d, n %4d: best %.1f us  mean %.1f us\n", APZ_STEM_CTB9, nb, best * 1e3, sum / 10 * 1e3);
        }
    }
    return 0;
}
