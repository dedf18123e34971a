// micro-benchmark: cost of writing one 256x256 bf16 C tile per CU (all 256 CUs at once, C row stride = N*2 bytes)
//   pattern A: the MFMA-register image as it is (per store instruction: 16 rows x 32 B, 8 B per lane)
//   pattern B: row-contiguous 16 B per lane (per instruction: 2 rows x 512 B) -- what an LDS-staged epilogue would issue
// Build: hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int PATTERN>
__global__ __launch_bounds__(512) void k_store(uint16_t* C, int N, int reps) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tm = blockIdx.x >> 4, tn = blockIdx.x & 15;        // 16 x 16 tiles
    const int wm = wave >> 2, wn = wave & 3, g = lane >> 4, mi = lane & 15;
    for (int r = 0; r < reps; ++r) {
        if (PATTERN == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = tm * 256 + wm * 128 + j * 16 + mi, n = tn * 256 + wn * 64 + i * 16 + g * 4;
                    *(u32x2*)(C + (size_t)m * N + n) = u32x2{(unsigned)(r + i), (unsigned)j};
                }
        } else {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = wave * 32 + it * 2 + (lane >> 5), c16 = lane & 31;
                *(u32x4*)(C + (size_t)(tm * 256 + row) * N + tn * 256 + c16 * 8) = u32x4{(unsigned)r, (unsigned)it, 0u, 1u};
            }
        }
    }
}
int main() {
    const int N = 4096;
    uint16_t* C; (void)hipMalloc(&C, (size_t)4096 * N * 2 * 2);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int pat = 0; pat < 2; ++pat)
        for (int reps : {1, 4}) {
            float best = 1e30f;
            for (int t = 0; t < 6; ++t) {
                (void)hipEventRecord(e0);
                for (int q = 0; q < 10; ++q) {
                    if (pat == 0) hipLaunchKernelGGL(k_store<0>, dim3(256), dim3(512), 0, 0, C, N, reps);
                    else hipLaunchKernelGGL(k_store<1>, dim3(256), dim3(512), 0, 0, C, N, reps);
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (t && ms < best) best = ms;
            }
            printf("pattern %c, %d tile-writes per launch: %6.2f us per launch  (%5.2f TB/s)\n", pat ? 'B' : 'A', reps, best * 100, 256.0 * 131072 * reps / (best * 100) / 1e6);
        }
    return 0;
}
