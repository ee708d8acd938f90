// Timing harness for trunk15_wino_kernel variants (ablation macros APZ_WINO_ABL_*): random data,
// HIP events.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Ialphapig_amd/csrc [-D...] tools/wino_ablate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "trunk15_wino.h"
#include "trunk15_wino2.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
#ifdef APZ_HARNESS_V1
    using T = apz::Wino15;
#define KERN apz::trunk15_wino_kernel
#else
    using T = apz::Wino2;
#define KERN apz::trunk15_wino2_kernel
#endif
    const char* tag = argc > 1 ? argv[1] : "base";
    int sizes[3] = {512, 1024, 4096};
    CK(hipFuncSetAttribute((const void*)KERN<true>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)KERN<false>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
    const int nmax = 4096;
    const size_t act = (size_t)nmax * 128 * 240;
    float *in, *res, *out, *upk, *bias;
    CK(hipMalloc(&in, act * 4)); CK(hipMalloc(&res, act * 4)); CK(hipMalloc(&out, act * 4));
    CK(hipMalloc(&upk, T::UPK_FLOATS * 4)); CK(hipMalloc(&bias, 128 * 4 + 512 + 4 * 8 * 8 * 8 + 1024));
    std::vector<float> h(act);
    srand(1);
    for (size_t i = 0; i < act; i++) h[i] = ((i & 15) == 15) ? 0.f : (rand() % 1000) * 1e-5f;
    CK(hipMemcpy(in, h.data(), act * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(res, h.data(), act * 4, hipMemcpyHostToDevice));
    std::vector<float> u(T::UPK_FLOATS);
    for (auto& v : u) v = ((rand() % 2000) - 1000) * 1e-5f;
    CK(hipMemcpy(upk, u.data(), u.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 512));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int si = 0; si < 3; si++) {
        const int n = sizes[si];
#ifdef APZ_HARNESS_V1
        const int grid = n < 256 ? n : 256;
#else
        const int grid = (n + 1) / 2 < 256 ? (n + 1) / 2 : 256;
#endif
        for (int resid = 0; resid < 2; resid++) {
            auto launch = [&]() {
                if (resid) hipLaunchKernelGGL((KERN<true>), dim3(grid), dim3(512), T::LDS_BYTES, 0, in, upk, bias, res, out, n);
                else hipLaunchKernelGGL((KERN<false>), dim3(grid), dim3(512), T::LDS_BYTES, 0, in, upk, bias, res, out, n);
            };
            for (int i = 0; i < 5; i++) launch();
            CK(hipEventRecord(a, 0));
            const int iters = 20;
            for (int i = 0; i < iters; i++) launch();
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            const double us = ms * 1e3 / iters, boards_per_wg = (double)n / 256.0;   // boards per CU
            printf("%-10s n=%5d resid=%d: %8.1f us  %6.1f us/board-slot  mfma-util %.1f%%  alg %.0f TF\n", tag, n, resid, us, us / boards_per_wg,
                   100.0 * (9216.0 * 32 / 4 / 2.25e3) / (us / boards_per_wg), 2.0 * n * 128 * 128 * 9 * 225 / us / 1e6);
        }
    }
#ifdef APZ_HARNESS_TWO_STREAMS
    {
        // do two independent launch chains on two streams hide each other's inter-kernel gaps?
        const int n = 512, grid = 256, iters = 40;
        hipStream_t s1, s2;
        CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        float *in2 = in + (size_t)2048 * 128 * 240, *out2 = out + (size_t)2048 * 128 * 240, *res2 = res + (size_t)2048 * 128 * 240;
        for (int mode = 0; mode < 2; mode++) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a, s1));
            for (int i = 0; i < iters; i++) {
                hipStream_t st = (mode == 1 && (i & 1)) ? s2 : s1;
                const bool second = i & 1;
                hipLaunchKernelGGL((KERN<true>), dim3(grid), dim3(512), T::LDS_BYTES, st, second ? in2 : in, upk, bias,
                                   second ? res2 : res, second ? out2 : out, n);
            }
            CK(hipStreamSynchronize(s2));
            CK(hipEventRecord(b, s1));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            printf("two-chain test, %s: %.1f us per launch\n", mode ? "two streams" : "one stream", ms * 1e3 / iters);
        }
    }
#endif
#ifdef APZ_WINO_STAMPS
    {
        // stamps of the last launch (n = 4096?) are overwritten per launch: rerun n = 512, resid = 1 once
        const int n = 512, grid = 256;
        hipLaunchKernelGGL((KERN<true>), dim3(grid), dim3(512), T::LDS_BYTES, 0, in, upk, bias, res, out, n);
        CK(hipDeviceSynchronize());
        unsigned long long hst[4 * 8 * 8];
        CK(hipMemcpy(hst, (char*)bias + 1024, sizeof(hst), hipMemcpyDeviceToHost));
        const char* names[8] = {"prologue", "barrier", "staging", "transform", "mfma", "epi0", "epi1", "total"};
        for (int wg = 0; wg < 2; wg++)
            for (int w = 0; w < 8; w++) {
                printf("wg %d wave %d:", wg, w);
                for (int i = 0; i < 8; i++) printf(" %s %.1f", names[i], hst[(wg * 8 + w) * 8 + i] / 100.0);   // 100 MHz -> us
                printf("\n");
            }
    }
#endif
    CK(hipGetLastError());
    return 0;
}
