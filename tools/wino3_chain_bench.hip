// trunk15_wino3_chain_kernel (all trunk layers of a forward in one launch) against the same layers as 2 x blocks
// launches of trunk15_wino3_kernel: outputs must be BIT-identical (same arithmetic, same order), timing of both.
// The chain's hand-off is exercised on purpose under uneven load: odd / ragged batches, repeated launches without
// resetting the flags (epoch counter), buffers that are reused every third layer.
// The chain kernel is an EXPERIMENT that did not pay: its header is tools/wino3_chain/trunk15_wino3.h, a fork
// of the product kernel; the product keeps one launch per layer.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Itools/wino3_chain -Ialphapig_amd/csrc tools/wino3_chain_bench.hip -o tools/_build/wino3_chain_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "trunk15_wino2.h"
#include "trunk15_wino3.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    using T2 = apz::Wino2;
    using T3 = apz::Wino3;
    const int NL = argc > 1 ? atoi(argv[1]) : 20;
    CK(hipFuncSetAttribute((const void*)apz::trunk15_wino3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)apz::trunk15_wino3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)apz::trunk15_wino3_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int num_cu = prop.multiProcessorCount;
    const int nmax = 1030;
    const size_t act = (size_t)nmax * 128 * 240;
    float *x0, *a[3], *b[3], *upk, *bias;
    CK(hipMalloc(&x0, act * 4));
    for (int i = 0; i < 3; i++) { CK(hipMalloc(&a[i], act * 4)); CK(hipMalloc(&b[i], act * 4)); }
    CK(hipMalloc(&upk, (size_t)NL * T2::UPK_FLOATS * 4)); CK(hipMalloc(&bias, (size_t)NL * 128 * 4));
    unsigned* flags; int* err;
    CK(hipMalloc(&flags, (size_t)(nmax + 2) * 4));
    CK(hipMemset(flags, 0, (size_t)(nmax + 2) * 4));
    CK(hipHostMalloc(&err, 4));
    *err = 0;
    std::vector<float> h(act);
    srand(1);
    for (size_t i = 0; i < act; i++) h[i] = ((i & 15) == 15) ? 0.f : ((rand() % 2000) - 600) * 1e-3f;
    CK(hipMemcpy(x0, h.data(), act * 4, hipMemcpyHostToDevice));
    {   // weights small enough that twenty layers stay bounded (the residual stream grows slowly)
        std::vector<float> u((size_t)NL * T2::UPK_FLOATS);
        for (auto& v : u) v = ((rand() % 2000) - 1000) * 4e-6f;
        CK(hipMemcpy(upk, u.data(), u.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> hb((size_t)NL * 128);
        for (auto& v : hb) v = ((rand() % 2000) - 1000) * 1e-4f;
        CK(hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    }
    unsigned epoch = 0;
    auto run_layers = [&](float** w, int n) {      // reference: one launch per layer, buffers rotate as in run_trunk()
        float *x = w[0], *t = w[1], *y = w[2];
        const int grid = std::min((n + 1) / 2, num_cu);
        for (int l = 0; l < NL; l++) {
            const float* u = upk + (size_t)l * T2::UPK_FLOATS;
            const float* bb = bias + (size_t)l * 128;
            if (!(l & 1)) {
                hipLaunchKernelGGL((apz::trunk15_wino3_kernel<false>), dim3(grid), dim3(512), T3::LDS_BYTES, 0, x, u, bb, nullptr, t, n);
            } else {
                hipLaunchKernelGGL((apz::trunk15_wino3_kernel<true>), dim3(grid), dim3(512), T3::LDS_BYTES, 0, t, u, bb, x, y, n);
                std::swap(x, y);
            }
        }
        return (NL & 1) ? t : x;
    };
    auto run_chain = [&](float** w, int n) -> float* {
        const int grid = apz::wino3_chain_grid(n, num_cu);
        if (grid == 0) return nullptr;
        apz::Wino3Chain ca;
        ca.act[0] = w[0]; ca.act[1] = w[1]; ca.act[2] = w[2];
        ca.upk = upk; ca.bias = bias; ca.flags = flags; ca.err = err; ca.epoch = ++epoch; ca.nlayers = NL; ca.poll_limit = 200000;
        hipLaunchKernelGGL(apz::trunk15_wino3_chain_kernel, dim3(grid), dim3(512), T3::LDS_BYTES, 0, ca, n);
        const int blocks = NL / 2;
        return (NL & 1) ? w[1] : ((blocks & 1) ? w[2] : w[0]);
    };
    int bad = 0;
    const int sizes[] = {512, 64, 33, 515, 1030, 96, 512};
    for (int si = 0; si < 7; si++) {
        const int n = sizes[si];
        const size_t cnt = (size_t)n * 128 * 240;
        for (int rep = 0; rep < (n == 512 ? 3 : 1); rep++) {
            for (int i = 0; i < 3; i++) { CK(hipMemset(a[i], 0, cnt * 4)); CK(hipMemset(b[i], 0, cnt * 4)); }
            CK(hipMemcpy(a[0], x0, cnt * 4, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(b[0], x0, cnt * 4, hipMemcpyDeviceToDevice));
            float* ra = run_layers(a, n);
            CK(hipDeviceSynchronize());
            float* rb = run_chain(b, n);
            if (!rb) { printf("chain n=%5d: not taken (grid 0)\n", n); break; }
            CK(hipDeviceSynchronize());
            std::vector<float> va(cnt), vb(cnt);
            CK(hipMemcpy(va.data(), ra, cnt * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(vb.data(), rb, cnt * 4, hipMemcpyDeviceToHost));
            size_t diff = 0, nan = 0, first = (size_t)-1;
            double maxv = 0;
            for (size_t i = 0; i < cnt; i++) {
                if (!(va[i] == va[i])) nan++;
                if (memcmp(&va[i], &vb[i], 4)) { diff++; if (first == (size_t)-1) first = i; }
                maxv = std::max(maxv, (double)std::fabs(va[i]));
            }
            printf("chain n=%5d rep %d grid %3d: %zu of %zu words differ (first board %zu ch %zu), max|ref| %.3f nan %zu err %d  %s\n", n, rep,
                   apz::wino3_chain_grid(n, num_cu), diff, cnt, first == (size_t)-1 ? 0 : first / (128 * 240),
                   first == (size_t)-1 ? 0 : (first / 240) % 128, maxv, nan, *err, (diff == 0 && nan == 0 && *err == 0) ? "ok" : "MISMATCH");
            if (diff || nan || *err) bad++;
            *err = 0;
        }
    }
    if (!getenv("APZ_NO_TIMING")) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int n = 512;
        for (int rep = 0; rep < 3; rep++)
            for (int kind = 0; kind < 2; kind++) {
                for (int i = 0; i < 3; i++) kind ? (void)run_chain(b, n) : (void)run_layers(a, n);
                CK(hipEventRecord(e0, 0));
                const int iters = 10;
                for (int i = 0; i < iters; i++) kind ? (void)run_chain(b, n) : (void)run_layers(a, n);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / iters;
                printf("%s n=%d %d layers: %8.1f us per forward = %6.1f us per layer  executed-MFMA %.3f of 157.3 TF  err %d\n", kind ? "chain " : "layers", n, NL,
                       us, us / NL, (double)n * 9216 * 2048 * NL / us / 1e6 / 157.3, *err);
            }
    }
    CK(hipGetLastError());
    printf(bad ? "RESULT: MISMATCH\n" : "RESULT: ok\n");
    return bad ? 2 : 0;
}
