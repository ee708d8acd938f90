// Timing of wgrad_wino3_kernel alone (csrc/wgrad_wino3.h) on the GPU box: HIP events over launches that rotate through
// operand sets larger than the Infinity Cache.  Built with -DAPZ_WGW3_NO_TRANSFORM=1 / -DAPZ_WGW3_NO_MFMA=1 (the kernel's
// measurement switches) to see what the phases cost, and with -DAPZ_WGW3_STAMPS for per-wave cycle counts of workgroup 0.
// (Correctness: tests/test_gpu_train.py against float64 autograd.)
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

int main() {
    using T3 = apz::WgradWino3;
    const int ROT = 4;
    CK(hipFuncSetAttribute((const void*)apz::wgrad_wino3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)apz::wgrad_wino3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
    const int only_n = getenv("APZ_WGKB_N") ? atoi(getenv("APZ_WGKB_N")) : 0;     // (PMC runs: one batch size)
    for (int n : {64, 128, 131, 512}) {
        if (only_n && n != only_n) continue;
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
        const int spx = std::max(1, std::min((n + 7) / 8, 256 / (8 * T3::BLOCKS))), slices = 8 * spx;
        CK(hipMalloc((void**)&scratch, (size_t)slices * apz::WgradWino::SCRATCH_FLOATS_PER_SLICE * 4));
        CK(hipMalloc((void**)&dw, 128 * 128 * 9 * 4));
        hipEvent_t a_ev, b_ev;
        CK(hipEventCreate(&a_ev));
        CK(hipEventCreate(&b_ev));
        bool buf = false;
        auto run = [&](int it, bool finish) {
            if (buf)
                hipLaunchKernelGGL(apz::wgrad_wino3_kernel<true>, dim3(T3::BLOCKS * slices), dim3(T3::THREADS), T3::LDS_BYTES, 0, x[it % ROT],
                                   dy[it % ROT], scratch, n, spx);
            else
                hipLaunchKernelGGL(apz::wgrad_wino3_kernel<false>, dim3(T3::BLOCKS * slices), dim3(T3::THREADS), T3::LDS_BYTES, 0, x[it % ROT],
                                   dy[it % ROT], scratch, n, spx);
            if (finish)
                hipLaunchKernelGGL(apz::wgrad_wino_finish_kernel, dim3(128 * 128 * 9 / 4 / 256), dim3(256), 0, 0, scratch, slices, dw);
        };
#ifdef APZ_WGW3_STAMPS
        {
            run(0, false);
            CK(hipDeviceSynchronize());
            unsigned long long st[8][6];
            CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(apz::apz_wgw3_stamps), sizeof(st)));
            printf("n=%d cycles per wave of workgroup 0 (loop ends, -, transform, barrier, mfma, epilogue):\n", n);
            for (int w = 0; w < 8; w++)
                printf("  wave %d: %8llu %8llu %8llu %8llu %8llu %8llu\n", w, st[w][0], st[w][1], st[w][2], st[w][3], st[w][4], st[w][5]);
        }
#endif
        for (int f = 0; f < 12; f++) {
            buf = (f >> 1) & 1;                       // rounds alternate plain loads / buffer loads (kernel, then kernel + finish, each)
            for (int it = 0; it < 5; it++) run(it, f & 1);
            CK(hipDeviceSynchronize());
            const int iters = 40;
            CK(hipEventRecord(a_ev));
            for (int it = 0; it < iters; it++) run(it, f & 1);
            CK(hipEventRecord(b_ev));
            CK(hipEventSynchronize(b_ev));
            float ms;
            CK(hipEventElapsedTime(&ms, a_ev, b_ev));
            printf("n=%d (%d slices) transform=%d mfma=%d %s %s: %.1f us  (%.3f of the fp32 matrix peak)\n", n, slices, !APZ_WGW3_NO_TRANSFORM,
                   !APZ_WGW3_NO_MFMA, buf ? "buffer loads" : "plain loads", (f & 1) ? "kernel + finish" : "kernel", ms * 1e3 / iters, n * 9216.0 * 2048.0 / (ms * 1e-3 / iters) / 157.3e12);
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
