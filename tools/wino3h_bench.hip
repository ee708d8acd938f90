// Timing + cross-check harness for trunk15_wino3h_kernel (the 2 x fp16 split trunk, round 6) beside trunk15_wino3b_kernel
// (3 x bf16) and the exact-fp32 trunk15_wino3_kernel: random data, HIP events, outputs compared with a naive
// double-precision kernel of the same Winograd-domain definition on the SAME fp32 weights U; the errors of all three
// against it are printed side by side.  APZ_ACT_SCALE=<x> multiplies the activations (range test of the fp16 split).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ialphapig_amd/csrc [-DAPZ_WINO3H_STAMPS] tools/wino3h_bench.hip -o tools/_build/wino3h_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "trunk15_wino3b.h"
#include "trunk15_wino3h.h"

// Y = A^T [sum_ci U (.) B^T d B] A + bias (+ resid), ReLU -- the definition, in double.  upk: wino_common.h's fp32 layout.
__global__ void wino_ref_kernel(const float* __restrict__ in, const float* __restrict__ upk, const float* __restrict__ bias,
                                const float* __restrict__ res, double* __restrict__ out, int n, int resid) {
    const long gid = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (gid >= (long)n * 128 * 16) return;
    const int tile = (int)(gid & 15), co = (int)((gid >> 4) & 127), bd = (int)(gid >> 11);
    const int ty = tile >> 2, tx = tile & 3;
    const double Bt[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                             {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
    const double At[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
    double M[6][6];
    for (int i = 0; i < 6; i++)
        for (int k = 0; k < 6; k++) M[i][k] = 0.0;
    const int cot = co >> 4, j = co & 15;
    for (int ci = 0; ci < 128; ci++) {
        double d[6][6], t[6][6];
        const float* pl = in + ((size_t)bd * 128 + ci) * 240;
        for (int i = 0; i < 6; i++)
            for (int k = 0; k < 6; k++) {
                const int r = 4 * ty - 1 + i, c = 4 * tx - 1 + k;
                d[i][k] = (r >= 0 && r < 15 && c >= 0 && c < 15) ? (double)pl[r * 16 + c] : 0.0;
            }
        for (int i = 0; i < 6; i++)
            for (int k = 0; k < 6; k++) {
                double a = 0;
                for (int x = 0; x < 6; x++) a += Bt[i][x] * d[x][k];
                t[i][k] = a;
            }
        const int c4 = ci >> 2, q = ci & 3;
        for (int i = 0; i < 6; i++)
            for (int k = 0; k < 6; k++) {
                double v = 0;
                for (int x = 0; x < 6; x++) v += t[i][x] * Bt[k][x];
                const int ph = i / 3, ii = i % 3;
                M[i][k] += (double)upk[((((size_t)cot * 2 + ph) * 32 + c4) * 64 + q * 16 + j) * 20 + 6 * ii + k] * v;
            }
    }
    double* op = out + ((size_t)bd * 128 + co) * 240;
    const float* rp = res + ((size_t)bd * 128 + co) * 240;
    for (int a = 0; a < 4; a++)
        for (int e = 0; e < 4; e++) {
            const int r = 4 * ty + a, c = 4 * tx + e;
            if (r >= 15) continue;
            double acc = 0;
            for (int i = 0; i < 6; i++)
                for (int k = 0; k < 6; k++) acc += At[a][i] * M[i][k] * At[e][k];
            acc += (double)bias[co];
            if (resid && c < 15) acc += (double)rp[r * 16 + c];
            op[r * 16 + c] = c < 15 ? (acc > 0 ? acc : 0.0) : 0.0;
        }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// does the fp16 MFMA keep subnormal inputs?  A = 1.0 in k slot 0 of every row, B = a subnormal fp16 (2^-20) there
__global__ void denorm_probe_kernel(float* o) {
    apz::f16x8 a = {}, b = {};
    if (threadIdx.x < 32) {
        a[0] = (_Float16)1.0f;
        b[0] = __builtin_bit_cast(_Float16, (unsigned short)0x0010);   // 16 x 2^-24 = 2^-20
    }
    apz::f32x16h c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) o[0] = c[0];
}

int main(int argc, char** argv) {
    using T2 = apz::WinoPack;
    using T3 = apz::Wino3;
    using TB = apz::Wino3B;
    using TH = apz::Wino3H;
    const bool quick = getenv("APZ_NO_TIMING") != nullptr;
    const float act_scale = getenv("APZ_ACT_SCALE") ? (float)atof(getenv("APZ_ACT_SCALE")) : 1.f;
    CK(hipFuncSetAttribute((const void*)apz::trunk15_wino3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)apz::trunk15_wino3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)apz::trunk15_wino3b_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, TB::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)apz::trunk15_wino3b_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, TB::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)apz::trunk15_wino3h_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, TH::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)apz::trunk15_wino3h_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, TH::LDS_BYTES));
    const int nmax = 2048;
    const size_t act = (size_t)nmax * 128 * 240;
    float *in, *res, *out[3], *upk, *bias, *biash, *probe;
    double* outd;
    void *upkb, *upkh;
    unsigned* flag;
    CK(hipMalloc(&in, act * 4)); CK(hipMalloc(&res, act * 4));
    for (int k = 0; k < 3; k++) CK(hipMalloc(&out[k], act * 4));
    const int nref = 1030;
    CK(hipMalloc(&outd, (size_t)nref * 128 * 240 * 8));
    CK(hipMalloc(&upk, T2::UPK_FLOATS * 4)); CK(hipMalloc(&bias, 128 * 4)); CK(hipMalloc(&biash, 256 * 4));
    CK(hipMalloc(&upkb, TB::UPK_BYTES)); CK(hipMalloc(&upkh, TH::UPK_BYTES));
    CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64)); CK(hipMalloc(&probe, 64));
    {
        hipLaunchKernelGGL(denorm_probe_kernel, dim3(1), dim3(64), 0, 0, probe);
        float pv = -1.f;
        CK(hipMemcpy(&pv, probe, 4, hipMemcpyDeviceToHost));
        printf("probe: fp16 MFMA with a subnormal B input 2^-20 x 1.0 -> %.9g (%s)\n", pv, pv == 9.5367431640625e-07f ? "subnormals kept" : "SUBNORMALS FLUSHED");
    }
    std::vector<float> h(act), hr(act);
    srand(1);
    // activations after a ReLU: a third zeros, the rest up to 1.4; residual of either sign
    for (size_t i = 0; i < act; i++) h[i] = ((i & 15) == 15) ? 0.f : act_scale * std::max(0.f, ((rand() % 2000) - 600) * 1e-3f);
    for (size_t i = 0; i < act; i++) hr[i] = ((i & 15) == 15) ? 0.f : ((rand() % 2000) - 1000) * 1e-3f;
    CK(hipMemcpy(in, h.data(), act * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(res, hr.data(), act * 4, hipMemcpyHostToDevice));
    std::vector<float> u(T2::UPK_FLOATS, 0.f);
    // U[pos][co][ci] random at the scale a folded 128 x 128 layer has (output channels at different scales, as BatchNorm
    // folding leaves them); kept in double for the splits, rounded for the fp32 pack
    std::vector<double> ud((size_t)36 * 128 * 128);
    for (size_t i = 0; i < ud.size(); i++) {
        const int co = (int)((i / 128) % 128);
        ud[i] = (((rand() % 20000) - 10000) * 2e-6 + ((rand() % 1000) - 500) * 1e-9) * (co % 7 == 3 ? 1.0 / 64 : (co % 5 == 1 ? 4.0 : 1.0));
    }
    auto u_of = [&](int co, int ci, int pos) { return (double)(float)ud[((size_t)pos * 128 + co) * 128 + ci]; };   // all kernels get the fp32 value
    for (int co = 0; co < 128; co++)
        for (int ci = 0; ci < 128; ci++)
            for (int pos = 0; pos < 36; pos++) {
                const int i = pos / 6, k = pos % 6, half = i / 3, cot = co >> 4, jj = co & 15, qq = ci & 3, c4 = ci >> 2;
                u[((((size_t)cot * 2 + half) * 32 + c4) * 64 + (qq * 16 + jj)) * 20 + (i - 3 * half) * 6 + k] = (float)u_of(co, ci, pos);
            }
    CK(hipMemcpy(upk, u.data(), u.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint16_t> ub, uh;
    apz::wino3b_pack_host(u_of, ub);
    CK(hipMemcpy(upkb, ub.data(), TB::UPK_BYTES, hipMemcpyHostToDevice));
    std::vector<float> hb(256);
    for (int i = 0; i < 128; i++) hb[i] = ((rand() % 2000) - 1000) * 1e-4f;
    apz::wino3h_pack_host(u_of, uh, hb.data() + 128);
    CK(hipMemcpy(upkh, uh.data(), TH::UPK_BYTES, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(biash, hb.data(), 1024, hipMemcpyHostToDevice));

    // kern: 0 = fp16 x 2, 1 = bf16 x 3, 2 = exact fp32
    auto launch = [&](int kern, int resid, int grid, int n, float* o) {
        if (kern == 0) {
            if (resid) hipLaunchKernelGGL((apz::trunk15_wino3h_kernel<true>), dim3(grid), dim3(512), TH::LDS_BYTES, 0, in, upkh, biash, res, o, n, flag);
            else hipLaunchKernelGGL((apz::trunk15_wino3h_kernel<false>), dim3(grid), dim3(512), TH::LDS_BYTES, 0, in, upkh, biash, res, o, n, flag);
        } else if (kern == 1) {
            if (resid) hipLaunchKernelGGL((apz::trunk15_wino3b_kernel<true>), dim3(grid), dim3(512), TB::LDS_BYTES, 0, in, upkb, bias, res, o, n);
            else hipLaunchKernelGGL((apz::trunk15_wino3b_kernel<false>), dim3(grid), dim3(512), TB::LDS_BYTES, 0, in, upkb, bias, res, o, n);
        } else {
            if (resid) hipLaunchKernelGGL((apz::trunk15_wino3_kernel<true>), dim3(grid), dim3(512), T3::LDS_BYTES, 0, in, upk, bias, res, o, n);
            else hipLaunchKernelGGL((apz::trunk15_wino3_kernel<false>), dim3(grid), dim3(512), T3::LDS_BYTES, 0, in, upk, bias, res, o, n);
        }
    };
    const char* kname[3] = {"f16x2", "bf16x3", "fp32"};

    if (getenv("APZ_PROFILE")) {          // rocprofv3 runs: 12 launches of each variant at 512 boards, nothing else
        for (int it = 0; it < 12; it++)
            for (int kern = 0; kern < 3; kern++)
                for (int resid = 0; resid < 2; resid++) launch(kern, resid, 256, 512, out[kern]);
        CK(hipDeviceSynchronize());
        printf("RESULT PROFILE\n");
        return 0;
    }
    // ---- cross-check against the naive double kernel at ragged sizes
    int bad = 0;
    const int check_sizes[7] = {1, 7, 64, 96, 512, 515, 1030};
    std::vector<float> ha[3];
    std::vector<double> hd;
    for (int ci = 0; ci < (quick ? 3 : 7); ci++) {
        const int n = check_sizes[ci];
        const int grid = getenv("APZ_GRID") ? atoi(getenv("APZ_GRID")) : apz::wino3_grid(n, 256);
        for (int resid = 0; resid < 2; resid++) {
            const size_t cnt = (size_t)n * 128 * 240;
            for (int k = 0; k < 3; k++) CK(hipMemset(out[k], 0xff, cnt * 4));
            hipLaunchKernelGGL(wino_ref_kernel, dim3((unsigned)((n * 128 * 16 + 255) / 256)), dim3(256), 0, 0, in, upk, bias, res, outd, n, resid);
            for (int k = 0; k < 3; k++) launch(k, resid, grid, n, out[k]);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            hd.resize(cnt);
            CK(hipMemcpy(hd.data(), outd, cnt * 8, hipMemcpyDeviceToHost));
            unsigned fl = 0;
            CK(hipMemcpy(&fl, flag, 4, hipMemcpyDeviceToHost));
            CK(hipMemset(flag, 0, 4));
            double emax[3] = {0, 0, 0}, ss[3] = {0, 0, 0}, scale = 0;
            size_t worst = 0, nonfinite = 0;
            for (int k = 0; k < 3; k++) {
                ha[k].resize(cnt);
                CK(hipMemcpy(ha[k].data(), out[k], cnt * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < cnt; i++) {
                    if ((i % 240) / 16 >= 15) continue;
                    if (!std::isfinite(ha[k][i])) { if (k == 0) nonfinite++; continue; }
                    const double d = std::fabs((double)ha[k][i] - hd[i]);
                    if (d > emax[k]) { emax[k] = d; if (k == 0) worst = i; }
                    ss[k] += d * d;
                    if (k == 0) scale = std::max(scale, std::fabs(hd[i]));
                }
            }
            const bool ok = nonfinite == 0 && emax[0] < 2e-5 * std::max(1.0, scale) && (fl == 0);
            if (!ok) bad++;
            printf("check n=%5d resid=%d grid=%d: f16x2 max err %.3e rms %.3e | bf16x3 %.3e rms %.3e | fp32 wino3 %.3e rms %.3e | scale %.2f nonfinite %zu flag %u %s (worst at board %zu ch %zu row %zu col %zu: %.6f vs %.6f)\n",
                   n, resid, grid, emax[0], std::sqrt(ss[0] / cnt), emax[1], std::sqrt(ss[1] / cnt), emax[2], std::sqrt(ss[2] / cnt), scale, nonfinite, fl,
                   ok ? "OK" : "MISMATCH", worst / (128 * 240), (worst / 240) % 128, (worst % 240) / 16, worst % 16, ha[0][worst], hd[worst]);
        }
    }
    if (quick) { printf("RESULT %s\n", bad ? "MISMATCH" : "OK"); return bad ? 2 : 0; }

    // ---- the overflow guard: activations far beyond the fp16 range must raise the flag
    {
        std::vector<float> big(2 * 128 * 240, 3000.f);
        CK(hipMemcpy(in, big.data(), big.size() * 4, hipMemcpyHostToDevice));
        launch(0, 1, apz::wino3_grid(2, 256), 2, out[0]);
        CK(hipDeviceSynchronize());
        unsigned fl = 0;
        CK(hipMemcpy(&fl, flag, 4, hipMemcpyDeviceToHost));
        CK(hipMemset(flag, 0, 4));
        printf("guard: activations of 3000 (|V| up to 3e5) -> flag %u %s\n", fl, fl ? "OK" : "MISSED");
        if (!fl) bad++;
        CK(hipMemcpy(in, h.data(), big.size() * 4, hipMemcpyHostToDevice));
        // ... and ONE activation beyond the range (every other one ordinary), at every pixel position class of a tile in turn:
        // the kernel looks at the four corner outputs of a tile only (trunk15_wino3h.h, APZH_CHK4)
        int missed = 0, tried = 0;
        for (int y = 0; y < 15; y += 1)
            for (int x = 0; x < 15; x += (y % 3 == 0 ? 1 : 4)) {
                std::vector<float> one(h.begin(), h.begin() + 2 * 128 * 240);
                const int ch = (7 * y + 3 * x) % 128, bd = (x + y) & 1;
                one[((size_t)bd * 128 + ch) * 240 + y * 16 + x] = 2.0e5f;
                CK(hipMemcpy(in, one.data(), one.size() * 4, hipMemcpyHostToDevice));
                launch(0, (x + y) % 2, apz::wino3_grid(2, 256), 2, out[0]);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(&fl, flag, 4, hipMemcpyDeviceToHost));
                CK(hipMemset(flag, 0, 4));
                tried++;
                if (!fl) missed++;
            }
        printf("guard: one activation of 2e5 among ordinary ones, %d positions -> %d missed %s\n", tried, missed, missed ? "MISSED" : "OK");
        if (missed) bad++;
        CK(hipMemcpy(in, h.data(), big.size() * 4, hipMemcpyHostToDevice));
    }

    // ---- timing: interleaved rounds of the three kernels on the same data
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int sizes[5] = {128, 256, 512, 1024, 2048};
    for (int si = 0; si < 5; si++) {
        const int n = sizes[si];
        const int grid = getenv("APZ_GRID") ? atoi(getenv("APZ_GRID")) : apz::wino3_grid(n, 256);
        float best[3][2], sum[3][2];
        for (int k = 0; k < 3; k++) for (int r = 0; r < 2; r++) best[k][r] = 1e9f, sum[k][r] = 0;
        const int rounds = 6, iters = 20;
        for (int r = 0; r < rounds; r++)
            for (int kern = 0; kern < 3; kern++)
                for (int resid = 0; resid < 2; resid++) {
                    for (int it = -3; it < iters; it++) {
                        if (it == 0) CK(hipEventRecord(e0, 0));
                        launch(kern, resid, grid, n, out[kern]);
                    }
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    best[kern][resid] = std::min(best[kern][resid], ms / iters);
                    sum[kern][resid] += ms / iters;
                }
        printf("time n=%5d grid=%3d:", n, grid);
        for (int k = 0; k < 3; k++)
            printf(" %s %.1f / %.1f us (mean %.1f / %.1f)%s", kname[k], best[k][0] * 1e3, best[k][1] * 1e3, sum[k][0] / rounds * 1e3, sum[k][1] / rounds * 1e3, k < 2 ? " |" : "\n");
    }
#ifdef APZ_WINO3H_STAMPS
    {
        unsigned long long st[4 * 8 * 12];
        for (int resid = 1; resid >= 0; resid--) {
        printf("stamps of one launch, 512 boards, resid=%d\n", resid);
        for (int it = 0; it < 4; it++) launch(0, resid, 256, 512, out[0]);
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(apz::apz_wino3h_stamps), sizeof st));
        for (int w = 0; w < 8; w++)
            printf("stamps wg0 wave %d: start-wait %llu prologue %llu chunk-barriers %llu chunks %llu | epilogue: barrier1 %llu M-write+resid %llu barrier2 %llu gather+transform %llu staging+stores %llu | total %llu cycles in %.2f us (%.2f GHz)\n", w,
                   st[w * 12 + 5], st[w * 12 + 0], st[w * 12 + 1], st[w * 12 + 2], st[w * 12 + 4], st[w * 12 + 8], st[w * 12 + 10], st[w * 12 + 9], st[w * 12 + 3],
                   st[w * 12 + 7], st[w * 12 + 6] * 0.01, st[w * 12 + 7] / (st[w * 12 + 6] * 10.0));
        }
    }
#endif
    printf("RESULT %s\n", bad ? "MISMATCH" : "OK");
    return bad ? 2 : 0;
}
