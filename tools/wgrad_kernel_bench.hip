// Timing of wgrad_wino2_kernel alone (csrc/wgrad_wino2.h) on the GPU box: HIP events over launches that rotate through
// operand sets larger than the Infinity Cache.  Built several times with -DAPZ_WGW2_NO_TRANSFORM=1 / -DAPZ_WGW2_NO_MFMA=1
// (the kernel's measurement switches) to see what the launch's skeleton (DMA stream + barriers + epilogue) costs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I alphapig_amd/csrc tools/wgrad_kernel_bench.hip -o tools/_build/wgrad_kernel_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wgrad_wino2.h"

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
    for (int n : {128, 512}) {
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
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
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
            CK(hipEventRecord(a));
            for (int it = 0; it < iters; it++) run(it, f);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            printf("n=%d transform=%d mfma=%d %s: %.1f us\n", n, !APZ_WGW2_NO_TRANSFORM, !APZ_WGW2_NO_MFMA,
                   f ? "kernel + finish" : "kernel", ms * 1e3 / iters);
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
