// The chunk loop of trunk15_wino3h_kernel reduced to its weight ring: per slot one 1 KB buffer load (L2-resident weights, the
// same addresses on every CU of an XCD, as in the kernel) into a ring of RING registers and two v_mfma_f32_32x32x16_f16 that
// read the unit loaded RING - 1 slots earlier; optionally a workgroup barrier every 9 slots.  512 threads, one workgroup per
// CU.  Prints ns per 9 slots ("chunk") -- the kernel's chunk takes ~1500 ns, its MFMAs 550 ns, its loads alone ~700 ns.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/mfma_ring_probe.hip -o tools/_build/mfma_ring_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned long long probe_t[2];
template <int RING, bool BARRIER, bool DEP, int NM>
__global__ __launch_bounds__(512) void probe(const void* __restrict__ w, float* __restrict__ out, int chunks, int mode) {
    unsigned long long t_loop = 0, t_ep = 0, t_last = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, 64u << 20, 0x00020000);
    f16x8 b = {(_Float16)0.25f, 0, 0, (_Float16)(lane * 0.02f), 0, 0, 0, 0};
    f16x8 fixed = {(_Float16)1.f, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc[9] = {};
    f16x8 af[RING];
    // mode 0: each wave its own 144 KB stream (16 chunks x 9 KB), all CUs the same (1.18 MB per XCD); mode 1: two halves as in
    // the kernel (blocks b and b + 8 differ: 2.36 MB per XCD); mode 2: every CU of an XCD its own copy (37.7 MB per XCD: beyond L2)
    const unsigned blk = blockIdx.x;
    const unsigned wbase = wave * 147456u + (mode == 1 ? ((blk >> 3) & 1) * 1179648u : (mode == 2 ? (blk >> 3) * 1179648u : 0u));
    auto uload = [&](int u, int slot) { af[slot] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, wbase + (unsigned)(u % 144) * 1024u, 0)); };
#pragma unroll
    for (int u = 0; u < RING - 1; u++) uload(u, u);
    int u0 = 0;
    for (int c = 0; c < chunks; c += 2) {
#pragma unroll
        for (int par = 0; par < 2; par++) {
            if (BARRIER) __syncthreads();
#pragma unroll
            for (int k = 0; k < 9; k++) {
                constexpr int dummy = 0;
                const int slot = (par * 9 + k) % RING;
#pragma unroll
                for (int m = 0; m < NM; m++) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(DEP ? af[slot] : fixed, b, acc[k], 0, 0, 0);
                uload(u0 + par * 9 + k + RING - 1, (par * 9 + k + RING - 1) % RING);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        u0 += 18;
        if ((mode & 16) && ((c + 2) & 15) == 0) {       // an item's epilogue: 128 KB read + 128 KB written per workgroup, private
            { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); t_loop += now - t_last; t_last = now; }
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, 0xfffffff0u, 0x00020000);
            const unsigned abase = (64u << 20) + blk * (2u << 20) + ((unsigned)(c >> 4) & 1) * (1u << 20);
            for (int i = 0; i < 16; i++) {
                const unsigned off = abase + (unsigned)(i * 8 + wave) * 1024u + lane * 16;
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 v;
                if (mode & 32) v = __builtin_amdgcn_raw_buffer_load_b128(ra, off, 0, 2);
                else v = __builtin_amdgcn_raw_buffer_load_b128(ra, off, 0, 0);
                v[0] += 1;
                if (mode & 32) __builtin_amdgcn_raw_buffer_store_b128(v, ra, off + (512u << 10), 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b128(v, ra, off + (512u << 10), 0, 0);
            }
            __syncthreads();
            { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); t_ep += now - t_last; t_last = now; }
        }
        if (!DEP) {
#pragma unroll
            for (int s = 0; s < RING; s++) asm volatile("" ::"v"(af[s]));
        }
    }
    float s = 0;
    for (int q = 0; q < RING; q++) s += (float)af[q][0];
    for (int k = 0; k < 9; k++)
        for (int v = 0; v < 16; v++) s += acc[k][v];
    out[blockIdx.x * 512 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) { probe_t[0] = t_loop; probe_t[1] = t_ep; }
}

template <int RING, bool BARRIER, bool DEP, int NM>
void run(const void* w, float* out, const char* name, int mode = 0) {
    const int chunks = 3200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<RING, BARRIER, DEP, NM>), dim3(256), dim3(512), 0, 0, w, out, 320, mode);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<RING, BARRIER, DEP, NM>), dim3(256), dim3(512), 0, 0, w, out, chunks, mode);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long pt[2];
    hipMemcpyFromSymbol(pt, HIP_SYMBOL(probe_t), 16);
    if (mode & 16) printf("      in-kernel: loops %.0f ns per chunk, epilogue phases %.1f us each\n", pt[0] * 10.0 / chunks, pt[1] * 0.01 / (chunks / 16));
    printf("mode %d %-40s ring %d barrier %d dependent %d mfma/slot %d: %.0f ns per chunk (9 slots)\n", mode, name, RING, (int)BARRIER, (int)DEP, NM, ms * 1e6 / chunks);
}

int main() {
    void* w;
    float* out;
    hipMalloc(&w, (64u << 20) + (512u << 20)); hipMalloc(&out, 256 * 512 * 4);
    hipMemset(w, 0, (64u << 20) + (512u << 20));
    run<6, true, true, 2>(w, out, "kernel-like");
    run<6, false, true, 2>(w, out, "no barrier");
    run<9, true, true, 2>(w, out, "ring 9");
    run<9, false, true, 2>(w, out, "ring 9, no barrier");
    run<6, true, false, 2>(w, out, "MFMAs independent of the loads");
    run<6, true, true, 0>(w, out, "loads only");
    run<6, true, false, 0>(w, out, "loads only, drained per 2 chunks");
    run<6, true, true, 1>(w, out, "one MFMA per slot");
    run<6, true, true, 4>(w, out, "four MFMAs per slot");
    run<12, true, true, 2>(w, out, "ring 12");
    run<12, false, true, 2>(w, out, "ring 12, no barrier");
    run<6, true, true, 2>(w, out, "kernel-like, two halves", 1);
    run<6, true, true, 2>(w, out, "two halves + epilogue traffic", 1 + 16);
    run<9, true, true, 2>(w, out, "ring 9, two halves + epilogue traffic", 1 + 16);
    run<6, true, true, 2>(w, out, "two halves + epilogue traffic, nt", 1 + 16 + 32);
    run<6, true, false, 2>(w, out, "independent, two halves + ep traffic", 1 + 16);
    run<6, true, true, 2>(w, out, "kernel-like, a copy per CU", 2);
    run<9, true, true, 2>(w, out, "ring 9, a copy per CU", 2);
    run<12, true, true, 2>(w, out, "ring 12, a copy per CU", 2);
    run<6, true, true, 0>(w, out, "loads only, a copy per CU", 2);
    return 0;
}
