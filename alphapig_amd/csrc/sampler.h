// Root move sampling on the GPU (opt-in "perf" mode; the parity mode keeps the reference's
// NumPy legacy-MT19937 sampling on the host, which no GPU generator can reproduce bit for bit).
//
// One wavefront per game, everything a wavefront reduction:
//   pi     = softmax(1/temp * log(visits + 1e-10))                 mcts_alphaZero.py:13-16, :152-155
//   noise ~ Dirichlet(alpha) over the root's children              mcts_alphaZero.py:200
//   move  ~ Categorical((1-eps) * pi + eps * noise)                mcts_alphaZero.py:198-201
// Dirichlet = normalised Gamma(alpha) draws (Marsaglia-Tsang with the alpha<1 boost), uniforms
// from a stateless counter hash (splitmix64 of (seed, row key, lane, draw)).  The row key is supplied by the
// caller -- the self-play engine passes (global game index, ply), so a game's draws depend on nothing but the
// seed, the game and the ply: not on the batch row, the rank or the step it happened to be sampled in (two ranks
// with the same seed never share noise) -- or defaults to (step, row).  Validated statistically in
// tests/test_gpu_sampler.py (pi to 1e-6 of the float64 host softmax; chi-square on move
// frequencies; Dirichlet first and second moments).
#pragma once
#include <hip/hip_runtime.h>

#include "heads.h"

namespace apz {

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct HashRng {
    unsigned long long key, ctr;
    __device__ float uniform() {   // (0, 1)
        const unsigned long long r = splitmix64(key + (ctr++) * 0xD1342543DE82EF95ull);
        return ((float)(r >> 40) + 0.5f) * (1.0f / 16777216.0f);
    }
    __device__ float normal() {
        const float u1 = uniform(), u2 = uniform();
        return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530718f * u2);
    }
    __device__ float gamma(float a) {   // Marsaglia & Tsang 2000; a < 1 via Gamma(a+1) * U^(1/a)
        const float boost = (a < 1.0f) ? powf(uniform(), 1.0f / a) : 1.0f;
        const float aa = (a < 1.0f) ? a + 1.0f : a;
        const float d = aa - 1.0f / 3.0f, c = rsqrtf(9.0f * d);
        for (int it = 0; it < 64; it++) {
            const float x = normal();
            float v = 1.0f + c * x;
            if (v <= 0.0f) continue;
            v = v * v * v;
            const float u = uniform();
            if (logf(u) < 0.5f * x * x + d - d * v + d * logf(v)) return d * v * boost;
        }
        return d * boost;
    }
};

__device__ __forceinline__ float wave_excl_scan(float v, int lane) {
    float s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(s, o);
        if (lane >= o) s += t;
    }
    return s - v;
}

// visits [G][HW] int32 (-1: no child), pi [G][HW] f32, moves [G] int32.  HW <= 256.
__global__ __launch_bounds__(64) void root_sample_kernel(const int* __restrict__ visits, float* __restrict__ pi,
                                                         int* __restrict__ moves, int G, int HW, float inv_temp,
                                                         float alpha, float eps, unsigned long long seed,
                                                         unsigned long long step, const unsigned long long* __restrict__ keys) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G) return;
    const int* vrow = visits + (size_t)g * HW;
    float x[4], p[4];
    bool has[4];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = lane * 4 + k;
        const int v = (i < HW) ? vrow[i] : -1;
        has[k] = v >= 0;
        x[k] = has[k] ? inv_temp * logf((float)v + 1e-10f) : -INFINITY;
        m = fmaxf(m, x[k]);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        p[k] = has[k] ? expf(x[k] - m) : 0.f;
        s += p[k];
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
    const unsigned long long rowkey = keys ? splitmix64(keys[g] ^ 0xA0761D6478BD642Full) : step * 0x100000001B3ull + (unsigned long long)g;
    HashRng rng{splitmix64(seed ^ splitmix64(rowkey)) + (unsigned long long)lane * 0x632BE59BD9B4E019ull, 0};
    float nz[4], ns = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        p[k] *= inv;
        const int i = lane * 4 + k;
        if (i < HW) pi[(size_t)g * HW + i] = p[k];
        nz[k] = (has[k] && eps > 0.f) ? rng.gamma(alpha) : 0.f;
        ns += nz[k];
    }
    ns = wave_sum(ns);
    const float ninv = ns > 0.f ? 1.0f / ns : 0.f;
    float w[4], ls = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        w[k] = has[k] ? (1.0f - eps) * p[k] + eps * nz[k] * ninv : 0.f;
        ls += w[k];
    }
    const float before = wave_excl_scan(ls, lane);
    const float total = __shfl(before + ls, 63);
    // one uniform per game: hashed from (seed, row key) only, identical in every lane
    HashRng grng{keys ? splitmix64(seed + 0x5851F42D4C957F2Dull) ^ splitmix64(rowkey + 0x9E3779B97F4A7C15ull)
                      : splitmix64(seed + 0x5851F42D4C957F2Dull * (step + 1)) ^ splitmix64((unsigned long long)g + 0x9E3779B97F4A7C15ull), 0};
    const float u = grng.uniform() * total;
    int pick = 0x7fffffff, last = -1;
    float c = before;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (has[k]) {
            last = lane * 4 + k;
            c += w[k];
            if (c > u && pick == 0x7fffffff) pick = lane * 4 + k;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        pick = min(pick, __shfl_xor(pick, o));
        last = max(last, __shfl_xor(last, o));
    }
    if (lane == 0) moves[g] = (pick == 0x7fffffff) ? last : pick;   // rounding guard: last child
}

}  // namespace apz
