// HBM write-bandwidth ceiling on this box: streaming stores of 1 GB by persistent workgroups in the access shapes
// the stem kernel could use (tools/: measurement aid, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o write_bw_probe write_bw_probe.hip && ./write_bw_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: plain 16-B stores, a wave writes 1 KiB contiguous; mode 1: the same, nontemporal;
// mode 2: 960-B planes (60 lanes), nontemporal, 16 planes per wave burst (the stem's epilogue shape)
template <int MODE>
__global__ __launch_bounds__(256) void fill(float* out, size_t n16, float v, int stride16 = 7680, int rot = 0) {
    const f32x4 val = {v, v, v, v};
    if (MODE < 2) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
            if (MODE == 0)
                reinterpret_cast<f32x4*>(out)[i] = val;
            else
                __builtin_nontemporal_store(val, reinterpret_cast<f32x4*>(out) + i);
        }
    } else if (MODE == 2 || MODE == 3) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const size_t planes = n16 / 60;           // 960-B planes
        for (size_t p0 = ((size_t)blockIdx.x * 4 + wave) * 16; p0 + 16 <= planes; p0 += (size_t)gridDim.x * 64) {
            if (lane < 60)
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    if (MODE == 2)
                        __builtin_nontemporal_store(val, reinterpret_cast<f32x4*>(out) + (p0 + c) * 60 + lane);
                    else
                        reinterpret_cast<f32x4*>(out)[(p0 + c) * 60 + lane] = val;
                }
        }
    } else if (MODE == 4 || MODE == 5) {
        // a workgroup owns one board (7680 16-B pieces = 128 planes) at a time and streams it front to back:
        // mode 4: wave w writes planes 4 i + w (60 lanes); mode 5: wave w writes KiB 4 i + w (64 lanes)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const size_t boards = n16 / 7680;
        for (size_t b = blockIdx.x; b < boards; b += gridDim.x) {
            f32x4* base = reinterpret_cast<f32x4*>(out) + b * 7680;
            if (MODE == 4) {
                if (lane < 60)
#pragma unroll 8
                    for (int i = 0; i < 32; i++) base[(4 * i + wave) * 60 + lane] = val;
            } else {
#pragma unroll 6
                for (int i = 0; i < 30; i++) base[(4 * i + wave) * 64 + lane] = val;
            }
        }
    } else if (MODE == 9) {
        // mode 5 (a workgroup streams its own board in 4 KiB steps) with the boards `stride16` 16-byte pieces apart
        // and workgroup k starting its sweep `rot * k` steps into the board (wrapping)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const size_t boards = n16 / stride16;
        for (size_t b = blockIdx.x; b < boards; b += gridDim.x) {
            f32x4* base = reinterpret_cast<f32x4*>(out) + b * stride16;
            const int start = (rot * (int)blockIdx.x) % 30;
            for (int i = 0; i < 30; i++) {
                const int ii = (i + start) % 30;
                base[(4 * ii + wave) * 64 + lane] = val;
            }
        }
    } else if (MODE == 7 || MODE == 8) {
        // a workgroup writes 16 planes (15 KiB = one 16-channel tile of one board) per step, its four waves taking
        // planes 4 i + w.  mode 7: chunks dealt round-robin over the workgroups; mode 8: XCD x (workgroup id mod 8)
        // takes the boards == x (mod 8), eight consecutive local workgroups the eight chunks of one board
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const size_t chunks = n16 / 960;
        const int per_xcd = gridDim.x / 8, xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        for (size_t it = 0;; it++) {
            size_t chunk;
            if (MODE == 7) {
                chunk = it * gridDim.x + blockIdx.x;
            } else {
                const size_t board = (it * (per_xcd / 8) + (l >> 3)) * 8 + xcd;
                chunk = board * 8 + (l & 7);
            }
            if (chunk >= chunks) break;
            f32x4* base = reinterpret_cast<f32x4*>(out) + chunk * 960;
            if (lane < 60)
#pragma unroll
                for (int i = 0; i < 4; i++) base[(4 * i + wave) * 60 + lane] = val;
        }
    } else {
        // mode 6: the stem today: wave w writes planes 32 w .. 32 w + 31 of its workgroup's board, nontemporal
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const size_t boards = n16 / 7680;
        for (size_t b = blockIdx.x; b < boards; b += gridDim.x) {
            f32x4* base = reinterpret_cast<f32x4*>(out) + b * 7680 + wave * 32 * 60;
            if (lane < 60)
#pragma unroll 8
                for (int c = 0; c < 32; c++) __builtin_nontemporal_store(val, base + c * 60 + lane);
        }
    }
}

int main() {
    const size_t bytes = (size_t)8192 * 128 * 960 + (size_t)8192 * 4096 * 4;     // room for padded strides
    float* d;
    hipMalloc(&d, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto run = [&](const char* label, auto launch) {
        float best = 1e9, sum = 0;
        for (int it = 0; it < 12; it++) {
            hipEventRecord(a);
            launch();
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (it >= 2) { best = ms < best ? ms : best; sum += ms; }
        }
        const double gb = 8192.0 * 122880;
        printf("%-44s best %.1f us (%.2f TB/s)  mean %.1f us (%.2f TB/s)\n", label, best * 1e3, gb / best / 1e9, sum / 10 * 1e3,
               gb / (sum / 10) / 1e9);
    };
    const size_t n16 = (size_t)8192 * 7680;
    for (int grid : {256, 512}) {
        char lab[128];
        snprintf(lab, sizeof lab, "grid %d mode 0 (1 MiB window)", grid);
        run(lab, [&] { hipLaunchKernelGGL(fill<0>, dim3(grid), dim3(256), 0, 0, d, n16, 1.0f, 7680, 0); });
        snprintf(lab, sizeof lab, "grid %d mode 6 (stem today)", grid);
        run(lab, [&] { hipLaunchKernelGGL(fill<6>, dim3(grid), dim3(256), 0, 0, d, n16, 1.0f, 7680, 0); });
        for (int stride_blocks : {30, 31, 32, 33}) {
            for (int rot : {0, 1, 7}) {
                const int stride16 = stride_blocks * 256;
                snprintf(lab, sizeof lab, "grid %d mode 9 stride %d x 4 KiB rot %d", grid, stride_blocks, rot);
                run(lab, [&] { hipLaunchKernelGGL(fill<9>, dim3(grid), dim3(256), 0, 0, d, (size_t)8192 * stride16, 1.0f, stride16, rot); });
            }
        }
    }
    return 0;
}
