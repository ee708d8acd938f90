// Timing of wgrad_wino2_kernel alone (csrc/wgrad_wino2.h) on the GPU box: HIP events over launches that rotate through
// operand sets larger than the Infinity Cache.  Built several times with -DAPZ_WGW2_NO_TRANSFORM=1 / -DAPZ_WGW2_NO_MFMA=1
// (the kernel's measurement switches) to see what the launch's skeleton (DMA stream + barriers + epilogue) costs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I alphapig_amd/csrc tools/wgrad_kernel_bench.hip -o tools/_build/wgrad_kernel_bench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wgrad_wino3.h"

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

int main(int argc, char** argv) {
    using T2 = apz::WgradWino2;
    const int ROT = 4;
    CK(hipFuncSetAttribute((const void*)apz::wgrad_wino2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T2::LDS_BYTES));
    for (int n : {64, 128, 131, 512}) {
        const size_t plane = (size_t)n * 128 * 240;
        std::vector<float> h(plane);
        for (size_t i = 0; i < plane; i++) h[i] = (i % 16 == 15) ? 0.f : (float)((i * 2654435761u >> 8) % 2001) / 1000.f - 1.f;
        float *x[ROT], *dy[ROT], *scratch, *dw;
        for (int r = 0; r < ROT; r++) {
            CK(hipMalloc((void**)&x[r], plane * 4));
            CK(hipMalloc((void**)&dy[r], plane * 4));
            CK(hipMemcpy(x[r], h.data(), plane * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dy[r], h.data(), plane * 4, hipMemcpyHostToDevice));
        }
        const int spx = std::max(1, std::min((n + 7) / 8, 256 / (8 * T2::BLOCKS))), slices = 8 * spx;
        CK(hipMalloc((void**)&scratch, (size_t)slices * apz::WgradWino::SCRATCH_FLOATS_PER_SLICE * 4));
        CK(hipMalloc((void**)&dw, 128 * 128 * 9 * 4));
        hipEvent_t a_ev, b_ev;
        CK(hipEventCreate(&a_ev));
        CK(hipEventCreate(&b_ev));
        auto run = [&](int it, bool finish) {
            hipLaunchKernelGGL(apz::wgrad_wino2_kernel, dim3(T2::BLOCKS * slices), dim3(T2::THREADS), T2::LDS_BYTES, 0, x[it % ROT],
                               dy[it % ROT], scratch, n, spx);
            if (finish)
                hipLaunchKernelGGL(apz::wgrad_wino_finish_kernel, dim3(128 * 128 * 9 / 4 / 256), dim3(256), 0, 0, scratch, slices, dw);
        };
        for (int f = 0; f < 2; f++) {
            for (int it = 0; it < 5; it++) run(it, f);
            CK(hipDeviceSynchronize());
            const int iters = 40;
            CK(hipEventRecord(a_ev));
            for (int it = 0; it < iters; it++) run(it, f);
            CK(hipEventRecord(b_ev));
            CK(hipEventSynchronize(b_ev));
            float ms;
            CK(hipEventElapsedTime(&ms, a_ev, b_ev));
            printf("n=%d transform=%d mfma=%d %s: %.1f us\n", n, !APZ_WGW2_NO_TRANSFORM, !APZ_WGW2_NO_MFMA,
                   f ? "kernel + finish" : "kernel", ms * 1e3 / iters);
        }
        // ---- wgrad_wino3_kernel: the same slices -> the same bits as wgrad_wino2_kernel (same transform expressions, same
        // accumulation order); then timed with its own decomposition (16 blocks x slices = one workgroup per CU)
        {
            using T3 = apz::WgradWino3;
            CK(hipFuncSetAttribute((const void*)apz::wgrad_wino3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
            std::vector<float> a(128 * 128 * 9), b(128 * 128 * 9);
            run(0, true);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(a.data(), dw, a.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemset(dw, 0, a.size() * 4));
            hipLaunchKernelGGL(apz::wgrad_wino3_kernel, dim3(T3::BLOCKS * slices), dim3(T3::THREADS), T3::LDS_BYTES, 0, x[0], dy[0],
                               scratch, n, spx);
            hipLaunchKernelGGL(apz::wgrad_wino_finish_kernel, dim3(128 * 128 * 9 / 4 / 256), dim3(256), 0, 0, scratch, slices, dw);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(b.data(), dw, b.size() * 4, hipMemcpyDeviceToHost));
            size_t diff = 0;
            double worst = 0, scale = 0;
            for (size_t i = 0; i < a.size(); i++) {
                diff += a[i] != b[i];
                worst = std::max(worst, (double)std::fabs(a[i] - b[i]));
                scale = std::max(scale, (double)std::fabs(a[i]));
            }
            printf("n=%d wgrad_wino3 vs wgrad_wino2 at %d slices: %zu of %zu values differ (worst %.3g, scale %.3g) %s\n", n, slices, diff,
                   a.size(), worst, scale, diff == 0 ? "ok" : "MISMATCH");
            const int spx3 = std::max(1, std::min((n + 7) / 8, 256 / (8 * T3::BLOCKS))), slices3 = 8 * spx3;
            auto run3 = [&](int it, bool finish) {
                hipLaunchKernelGGL(apz::wgrad_wino3_kernel, dim3(T3::BLOCKS * slices3), dim3(T3::THREADS), T3::LDS_BYTES, 0, x[it % ROT],
                                   dy[it % ROT], scratch, n, spx3);
                if (finish)
                    hipLaunchKernelGGL(apz::wgrad_wino_finish_kernel, dim3(128 * 128 * 9 / 4 / 256), dim3(256), 0, 0, scratch, slices3, dw);
            };
#ifdef APZ_WGW3_STAMPS
            {
                run3(0, false);
                CK(hipDeviceSynchronize());
                unsigned long long st[8][6];
                CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(apz::apz_wgw3_stamps), sizeof(st)));
                printf("n=%d cycles per wave of workgroup 0 (dma, barrier A, transform, barrier B, mfma, epilogue):\n", n);
                for (int w = 0; w < 8; w++)
                    printf("  wave %d: %8llu %8llu %8llu %8llu %8llu %8llu\n", w, st[w][0], st[w][1], st[w][2], st[w][3], st[w][4], st[w][5]);
            }
#endif
            for (int f = 0; f < 2; f++) {
                for (int it = 0; it < 5; it++) run3(it, f);
                CK(hipDeviceSynchronize());
                const int iters = 40;
                CK(hipEventRecord(a_ev));
                for (int it = 0; it < iters; it++) run3(it, f);
                CK(hipEventRecord(b_ev));
                CK(hipEventSynchronize(b_ev));
                float ms;
                CK(hipEventElapsedTime(&ms, a_ev, b_ev));
                printf("n=%d wgrad_wino3 (%d slices) transform=%d mfma=%d %s: %.1f us\n", n, slices3, !APZ_WGW3_NO_TRANSFORM, !APZ_WGW3_NO_MFMA,
                       f ? "kernel + finish" : "kernel", ms * 1e3 / iters);
            }
        }
        for (int r = 0; r < ROT; r++) {
            CK(hipFree(x[r]));
            CK(hipFree(dy[r]));
        }
        CK(hipFree(scratch));
        CK(hipFree(dw));
    }
    return 0;
}
