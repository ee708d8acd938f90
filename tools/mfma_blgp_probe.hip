// Do the gfx950 fp16 MFMAs honour the BLGP field (B-matrix lane-group pattern of the MAI encoding) and CBSZ / ABID?
// For v_mfma_f32_32x32x16_f16 and v_mfma_f32_16x16x32_f16: D = A B with random A, B and blgp = 0 .. 7; the host works out,
// for each candidate lane permutation of B's registers (identity, broadcast lanes 0-31, broadcast lanes 32-63, rotations
// by 16, broadcast of one 16-lane group), which one the hardware applied.  gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/mfma_blgp_probe.hip -o tools/_build/mfma_blgp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BLGP>
__global__ void k32(const f16x8* a, const f16x8* b, f32x16* d) {
    const int lane = threadIdx.x;
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[lane], b[lane], acc, 0, 0, BLGP);
    d[lane] = acc;
}
template <int BLGP>
__global__ void k16(const f16x8* a, const f16x8* b, f32x4* d) {
    const int lane = threadIdx.x;
    f32x4 acc = {};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[lane], b[lane], acc, 0, 0, BLGP);
    d[lane] = acc;
}

// lane permutation candidates: src lane of B for destination lane l
static int perm(int which, int l) {
    switch (which) {
        case 0: return l;
        case 1: return l & 31;              // lanes 0-31 to both halves
        case 2: return 32 + (l & 31);       // lanes 32-63 to both halves
        case 3: return (l + 16) & 63;       // rotate down by 16
        case 4: return (l + 48) & 63;       // rotate up by 16
        case 5: return l & 15;
        case 6: return 16 + (l & 15);
        case 7: return 32 + (l & 15);
        case 8: return 48 + (l & 15);
        case 9: return (l + 32) & 63;       // swap halves
    }
    return l;
}
static const char* pname[] = {"identity", "bcast lanes 0-31", "bcast lanes 32-63", "rotate -16", "rotate +16", "bcast 0-15", "bcast 16-31",
                              "bcast 32-47", "bcast 48-63", "swap halves"};

int main() {
    std::vector<_Float16> ha(64 * 8), hb(64 * 8);
    srand(5);
    for (auto& x : ha) x = (_Float16)((rand() % 17 - 8) * 0.25f);
    for (auto& x : hb) x = (_Float16)((rand() % 13 - 6) * 0.5f);
    f16x8 *da, *db;
    f32x16* d32;
    f32x4* d16;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&d32, 64 * 64); hipMalloc(&d16, 64 * 16);
    hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
    // 32x32x16: A lane (i = l & 31, kg = l >> 5) k = 8 kg + e; B lane (j = l & 31, kg) ; D lane (j = l & 31, g = l >> 5) reg r: row 8 (r / 4) + 4 g ... use the
    // generic formula: row = (r / 4) * 8 + g * 4 + (r % 4)
    for (int blgp = 0; blgp < 8; blgp++) {
        std::vector<float> out(64 * 16);
        switch (blgp) {
#define L(B) case B: hipLaunchKernelGGL(k32<B>, dim3(1), dim3(64), 0, 0, da, db, d32); break;
            L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)
#undef L
        }
        hipMemcpy(out.data(), d32, 64 * 64, hipMemcpyDeviceToHost);
        int found = -1;
        for (int w = 0; w < 10 && found < 0; w++) {
            double worst = 0;
            for (int l = 0; l < 64; l++)
                for (int r = 0; r < 16; r++) {
                    const int j = l & 31, g = l >> 5, row = (r / 4) * 8 + g * 4 + (r % 4);
                    double s = 0;
                    for (int kg = 0; kg < 2; kg++)
                        for (int e = 0; e < 8; e++) s += (double)ha[(kg * 32 + row) * 8 + e] * (double)hb[perm(w, kg * 32 + j) * 8 + e];
                    worst = std::fmax(worst, std::fabs(s - out[l * 16 + r]));
                }
            if (worst < 1e-3) found = w;
        }
        printf("v_mfma_f32_32x32x16_f16 blgp=%d: B as %s\n", blgp, found < 0 ? "NONE of the candidates" : pname[found]);
    }
    for (int blgp = 0; blgp < 8; blgp++) {
        std::vector<float> out(64 * 4);
        switch (blgp) {
#define L(B) case B: hipLaunchKernelGGL(k16<B>, dim3(1), dim3(64), 0, 0, da, db, d16); break;
            L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)
#undef L
        }
        hipMemcpy(out.data(), d16, 64 * 16, hipMemcpyDeviceToHost);
        int found = -1;
        for (int w = 0; w < 10 && found < 0; w++) {
            double worst = 0;
            for (int l = 0; l < 64; l++)
                for (int r = 0; r < 4; r++) {
                    const int j = l & 15, g = l >> 4, row = 4 * g + r;
                    double s = 0;
                    for (int kg = 0; kg < 4; kg++)
                        for (int e = 0; e < 8; e++) s += (double)ha[(kg * 16 + row) * 8 + e] * (double)hb[perm(w, kg * 16 + j) * 8 + e];
                    worst = std::fmax(worst, std::fabs(s - out[l * 4 + r]));
                }
            if (worst < 1e-3) found = w;
        }
        printf("v_mfma_f32_16x16x32_f16 blgp=%d: B as %s\n", blgp, found < 0 ? "NONE of the candidates" : pname[found]);
    }
    return 0;
}
