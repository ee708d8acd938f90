// Timing + cross-check harness for trunk15_wino3_kernel: random data, HIP events, outputs compared with a naive
// double-precision kernel of the same Winograd-domain definition (wino_ref_kernel below: one thread per (board, output
// channel, tile)) and, for board 0, with a CPU loop.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Ialphapig_amd/csrc [-DAPZ_WINO3_STAMPS] tools/wino3_bench.hip -o tools/_build/wino3_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "trunk15_wino3.h"

// Y = A^T [sum_ci U (.) B^T d B] A + bias (+ resid), ReLU -- the definition, in double, no tiling tricks.
// in / res / out: rows16 [n][128][15][16]; upk: wino_common.h's packed layout.
__global__ void wino_ref_kernel(const float* __restrict__ in, const float* __restrict__ upk, const float* __restrict__ bias,
                                const float* __restrict__ res, float* __restrict__ out, int n, int resid) {
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
    float* op = out + ((size_t)bd * 128 + co) * 240;
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
            op[r * 16 + c] = c < 15 ? (float)(acc > 0 ? acc : 0.0) : 0.f;
        }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    using T2 = apz::WinoPack;
    using T3 = apz::Wino3;
    (void)argc; (void)argv;
    CK(hipFuncSetAttribute((const void*)apz::trunk15_wino3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)apz::trunk15_wino3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)(apz::trunk15_wino3_kernel<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)(apz::trunk15_wino3_kernel<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
    const int nmax = 4096;
    const size_t act = (size_t)nmax * 128 * 240;
    float *in, *res, *out, *out2, *upk, *bias;
    CK(hipMalloc(&in, act * 4)); CK(hipMalloc(&res, act * 4)); CK(hipMalloc(&out, act * 4)); CK(hipMalloc(&out2, act * 4));
    CK(hipMalloc(&upk, T2::UPK_FLOATS * 4)); CK(hipMalloc(&bias, 128 * 4));
    std::vector<float> h(act), hr(act);
    srand(1);
    for (size_t i = 0; i < act; i++) h[i] = ((i & 15) == 15) ? 0.f : ((rand() % 2000) - 600) * 1e-3f;
    for (size_t i = 0; i < act; i++) hr[i] = ((i & 15) == 15) ? 0.f : ((rand() % 2000) - 1000) * 1e-3f;
    CK(hipMemcpy(in, h.data(), act * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(res, hr.data(), act * 4, hipMemcpyHostToDevice));
    std::vector<float> u(T2::UPK_FLOATS);
    for (auto& v : u) v = ((rand() % 2000) - 1000) * 2e-5f;
    CK(hipMemcpy(upk, u.data(), u.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hb(128);
    for (auto& v : hb) v = ((rand() % 2000) - 1000) * 1e-4f;
    CK(hipMemcpy(bias, hb.data(), 512, hipMemcpyHostToDevice));

    // ---- CPU reference (double) for board 0, a few channels: Y = A^T [sum_ci U (.) B^T d B] A + bias (+ resid), ReLU
    auto cpu_plane = [&](int co, bool resid, std::vector<double>& yout) {
        static const double Bt[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                        {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
        static const double At[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
        yout.assign(240, 0.0);
        for (int ty = 0; ty < 4; ty++)
            for (int tx = 0; tx < 4; tx++) {
                double M[6][6] = {{0}};
                for (int ci = 0; ci < 128; ci++) {
                    double d[6][6], t[6][6], V[6][6];
                    for (int i = 0; i < 6; i++)
                        for (int k = 0; k < 6; k++) {
                            const int r = 4 * ty - 1 + i, c = 4 * tx - 1 + k;
                            d[i][k] = (r >= 0 && r < 15 && c >= 0 && c < 15) ? h[(size_t)ci * 240 + r * 16 + c] : 0.0;
                        }
                    for (int i = 0; i < 6; i++)
                        for (int k = 0; k < 6; k++) {
                            double a = 0;
                            for (int x = 0; x < 6; x++) a += Bt[i][x] * d[x][k];
                            t[i][k] = a;
                        }
                    for (int i = 0; i < 6; i++)
                        for (int k = 0; k < 6; k++) {
                            double a = 0;
                            for (int x = 0; x < 6; x++) a += t[i][x] * Bt[k][x];
                            V[i][k] = a;
                        }
                    // upk: [cot 8][ph 2][c4 32][lane 64][20], lane = q*16 + j: co = cot*16 + j, ci = c4*4 + q, index 6*ii + k of row 3*ph + ii
                    const int cot = co / 16, j = co % 16, c4 = ci / 4, q = ci % 4;
                    for (int i = 0; i < 6; i++)
                        for (int k = 0; k < 6; k++) {
                            const int ph = i / 3, ii = i % 3;
                            const double U = u[((((size_t)cot * 2 + ph) * 32 + c4) * 64 + q * 16 + j) * 20 + 6 * ii + k];
                            M[i][k] += U * V[i][k];
                        }
                }
                for (int a = 0; a < 4; a++)
                    for (int e = 0; e < 4; e++) {
                        double acc = 0;
                        for (int i = 0; i < 6; i++)
                            for (int k = 0; k < 6; k++) acc += At[a][i] * M[i][k] * At[e][k];
                        const int r = 4 * ty + a, c = 4 * tx + e;
                        if (r < 15 && c < 15) {
                            acc += hb[co];
                            if (resid) acc += hr[(size_t)co * 240 + r * 16 + c];
                            yout[r * 16 + c] = acc > 0 ? acc : 0.0;
                        }
                    }
            }
    };
    // ---- cross-check against the naive kernel at ragged sizes (odd batch, fewer pairs than CUs, several pairs per workgroup)
    int bad = 0;
    const int check_sizes[7] = {1, 7, 64, 96, 512, 515, 1030};
    for (int ci = 0; ci < (getenv("APZ_NO_TIMING") ? 2 : 7); ci++) {
        const int n = check_sizes[ci];
        const int grid3 = apz::wino3_grid(n, 256);
        for (int resid = 0; resid < 2; resid++) {
            CK(hipMemset(out, 0xff, (size_t)n * 128 * 240 * 4));
            CK(hipMemset(out2, 0, (size_t)n * 128 * 240 * 4));
            hipLaunchKernelGGL(wino_ref_kernel, dim3((n * 128 * 16 + 255) / 256), dim3(256), 0, 0, in, upk, bias, res, out2, n, resid);
            if (resid)
                hipLaunchKernelGGL((apz::trunk15_wino3_kernel<true>), dim3(grid3), dim3(512), T3::LDS_BYTES, 0, in, upk, bias, res, out, n);
            else
                hipLaunchKernelGGL((apz::trunk15_wino3_kernel<false>), dim3(grid3), dim3(512), T3::LDS_BYTES, 0, in, upk, bias, res, out, n);
            CK(hipDeviceSynchronize());
#ifdef APZ3_DEBUG_X
            if (n == 1 && !resid) {
                static float dbg[8 * 4 * 2 * 64 * 4];
                CK(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(apz::apz_wino3_dbg), sizeof(dbg)));
                for (int w = 0; w < 8; w++)
                    for (int r = 0; r < 4; r++) {
                        int bad = 0, first = -1;
                        for (int l = 0; l < 64; l++)
                            for (int e = 0; e < 4; e++) {
                                const float sent = dbg[(((w * 4 + r) * 2 + 0) * 64 + l) * 4 + e];
                                const float recv = dbg[((((w ^ 4) * 4 + r) * 2 + 1) * 64 + l) * 4 + e];
                                if (sent != recv) { bad++; if (first < 0) first = l * 4 + e; }
                            }
                        printf("   X wave %d r %d: P2 sent vs received by wave %d: %d differ (first lane %d e %d)\n", w, r, w ^ 4, bad, first / 4, first & 3);
                    }
            }
#endif
            const size_t cnt = (size_t)n * 128 * 240;
            std::vector<float> a(cnt), b(cnt);
            CK(hipMemcpy(a.data(), out, cnt * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), out2, cnt * 4, hipMemcpyDeviceToHost));
            double maxd = 0, maxv = 0;
            size_t nan = 0, pad_bad = 0;
            for (size_t i = 0; i < cnt; i++) {
                if (!(a[i] == a[i])) { nan++; continue; }
                if ((i & 15) == 15 && a[i] != 0.f) pad_bad++;
                const double d = std::fabs((double)a[i] - (double)b[i]);
                if (d > maxd) maxd = d;
                if (std::fabs(b[i]) > maxv) maxv = std::fabs(b[i]);
            }
            if (ci == 0) {                      // both kernels against the CPU reference, board 0
                const int chans[6] = {0, 8, 9, 10, 57, 121};
                for (int cc = 0; cc < 6; cc++) {
                    std::vector<double> yr;
                    cpu_plane(chans[cc], resid != 0, yr);
                    double e3 = 0, e2 = 0;
                    for (int i = 0; i < 240; i++) {
                        e3 = std::max(e3, std::fabs(yr[i] - a[(size_t)chans[cc] * 240 + i]));
                        e2 = std::max(e2, std::fabs(yr[i] - b[(size_t)chans[cc] * 240 + i]));
                    }
                    printf("   cpu ref co %3d resid %d: max|wino3-cpu| %.3e  max|naive-cpu| %.3e\n", chans[cc], resid, e3, e2);
                }
            }
            const bool ok = nan == 0 && pad_bad == 0 && maxd < 2e-5 * (1.0 + maxv);
            if (!ok) {                          // where: histogram by channel tile / board parity / row, first few elements
                int by_ct[8] = {0}, by_par[2] = {0}, by_row[15] = {0}, by_col[16] = {0}, shown = 0;
                for (size_t i = 0; i < cnt; i++) {
                    const double d = std::fabs((double)a[i] - (double)b[i]);
                    if (!(d < 2e-5 * (1.0 + maxv))) {
                        const int col = i & 15, row = (i / 16) % 15, co = (i / 240) % 128, bd = (int)(i / (240 * 128));
                        by_ct[co / 16]++; by_par[bd & 1]++; by_row[row]++; by_col[col]++;
                        if (shown++ < 6) printf("   bd %d co %d row %d col %d: wino3 %g naive %g\n", bd, co, row, col, a[i], b[i]);
                    }
                }
                if (n == 1 && getenv("APZ_DUMP")) {
                    for (int co = 8; co < 11; co++) {
                        printf("   plane co %d (wino3 | naive)\n", co);
                        for (int row = 0; row < 15; row++) {
                            printf("    ");
                            for (int col = 0; col < 16; col++) printf("%7.2f", a[(size_t)co * 240 + row * 16 + col]);
                            printf("  |");
                            for (int col = 0; col < 16; col++) printf("%7.2f", b[(size_t)co * 240 + row * 16 + col]);
                            printf("\n");
                        }
                    }
                }
                printf("   by co/16:"); for (int k = 0; k < 8; k++) printf(" %d", by_ct[k]);
                printf("  by board parity: %d %d\n   by row:", by_par[0], by_par[1]);
                for (int k = 0; k < 15; k++) printf(" %d", by_row[k]);
                printf("\n   by col:"); for (int k = 0; k < 16; k++) printf(" %d", by_col[k]);
                printf("\n");
            }
            printf("check n=%5d resid=%d: max|wino3-naive| %.3e (max|ref| %.3f) nan %zu pad %zu  %s\n", n, resid, maxd, maxv, nan, pad_bad,
                   ok ? "ok" : "MISMATCH");
            if (!ok) bad++;
            // QUARTER items (four workgroups per pair, 32 output channels each): the SAME bits, on the multiple-of-32 grid
            // the launcher uses and on an odd grid (whole pairs per workgroup, quarter after quarter)
            bool quarter = false;
            const int gq = apz::wino3_grid(n, 256, &quarter);
            for (int gi = 0; gi < (quarter ? 2 : 0); gi++) {
                const int grid = gi == 0 ? gq : 7;
                CK(hipMemset(out2, 0xff, cnt * 4));
                if (resid) hipLaunchKernelGGL((apz::trunk15_wino3_kernel<true, true, true>), dim3(grid), dim3(512), T3::LDS_BYTES, 0, in, upk, bias, res, out2, n);
                else hipLaunchKernelGGL((apz::trunk15_wino3_kernel<false, true, true>), dim3(grid), dim3(512), T3::LDS_BYTES, 0, in, upk, bias, res, out2, n);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(b.data(), out2, cnt * 4, hipMemcpyDeviceToHost));
                size_t diff = 0;
                for (size_t i = 0; i < cnt; i++)
                    if ((i % 240) / 16 < 15 && memcmp(&a[i], &b[i], 4)) diff++;
                printf("   quarter items n=%d resid=%d grid=%d: %zu values differ from the 64-channel items %s\n", n, resid, grid, diff, diff ? "MISMATCH" : "ok");
                if (diff) bad++;
            }
        }
    }

    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int sizes[7] = {32, 64, 128, 128, 512, 1024, 4096};
    for (int rep = 0; rep < (getenv("APZ_NO_TIMING") ? 0 : 2); rep++)
    for (int si = 0; si < 7; si++) {
        const int n = sizes[si];
        bool quarter = false;
        const int gq = apz::wino3_grid(n, 256, &quarter);
        const bool useq = quarter && si != 3;                 // 128 boards twice: quarter items, then the 64-channel items
        const int grid3 = useq ? gq : apz::wino3_grid(n, 256);
        const char* tag = useq ? "quarter" : "half";
        for (int kern = 1; kern < 2; kern++)
        for (int resid = 0; resid < 2; resid++) {
            auto launch = [&]() {
                if (useq) {
                    if (resid) hipLaunchKernelGGL((apz::trunk15_wino3_kernel<true, true, true>), dim3(grid3), dim3(512), T3::LDS_BYTES, 0, in, upk, bias, res, out, n);
                    else hipLaunchKernelGGL((apz::trunk15_wino3_kernel<false, true, true>), dim3(grid3), dim3(512), T3::LDS_BYTES, 0, in, upk, bias, res, out, n);
                } else {
                    if (resid) hipLaunchKernelGGL((apz::trunk15_wino3_kernel<true>), dim3(grid3), dim3(512), T3::LDS_BYTES, 0, in, upk, bias, res, out, n);
                    else hipLaunchKernelGGL((apz::trunk15_wino3_kernel<false>), dim3(grid3), dim3(512), T3::LDS_BYTES, 0, in, upk, bias, res, out, n);
                }
            };
            for (int i = 0; i < 5; i++) launch();
            CK(hipEventRecord(a, 0));
            const int iters = 20;
            for (int i = 0; i < iters; i++) launch();
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            const double us = ms * 1e3 / iters;
            printf("%-8s %s n=%5d resid=%d: %8.1f us  executed-MFMA %.1f TF (%.3f of 157.3)  alg %.0f TF\n", tag, "wino3", n, resid, us,
                   (double)n * 9216 * 2048 / us / 1e6, (double)n * 9216 * 2048 / us / 1e6 / 157.3, 2.0 * n * 128 * 128 * 9 * 225 / us / 1e6);
        }
    }
#ifdef APZ_WINO3_STAMPS
    {
        const int n = 512, grid3 = 256;
        hipLaunchKernelGGL((apz::trunk15_wino3_kernel<true>), dim3(grid3), dim3(512), T3::LDS_BYTES, 0, in, upk, bias, res, out, n);
        CK(hipDeviceSynchronize());
        unsigned long long hst[4 * 8 * 8];
        CK(hipMemcpyFromSymbol(hst, HIP_SYMBOL(apz::apz_wino3_stamps), sizeof(hst)));
        const char* names[8] = {"prologue", "barrier", "body", "epi-compute", "epi-barrier", "epi-resid", "epi-store", "total"};
        for (int wg = 0; wg < 2; wg++)
            for (int w = 0; w < 8; w++) {
                printf("wg %d wave %d:", wg, w);
                for (int i = 0; i < 8; i++)
                    if (names[i][0] != '-') printf(" %s %.1f", names[i], hst[(wg * 8 + w) * 8 + i] / 100.0);   // 100 MHz -> us
                printf("\n");
            }
    }
#endif
    CK(hipGetLastError());
    printf(bad ? "RESULT: MISMATCH\n" : "RESULT: ok\n");
    return bad ? 2 : 0;
}
