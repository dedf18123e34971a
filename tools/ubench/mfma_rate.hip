// MFMA rate microbenchmark: 16x16x32 vs 32x32x16 bf16 on a 128x64 wave tile (8 A + 4 B fragment registers per 32 k,
// 32 resp. 16 MFMAs), 8 waves per CU (2 per SIMD), 256 blocks; operands constant-ish or pseudo-random N(0,1)-like bf16.
// The chip is power-limited under random-data MFMA load, so the sustained rate depends on the operand bits -- and the
// question here is whether the instruction shape changes that (fewer operand-register reads per flop with 32x32x16).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u4;
__device__ bf16x8 mk(unsigned& h, bool rnd, int i) {
    u4 w;
    for (int q = 0; q < 4; ++q) {
        h = h * 1664525u + 1013904223u;
        const unsigned lo = 0x3f00u | ((h >> 8) & 0x80ffu), hi = 0x3f00u | ((h >> 16) & 0x80ffu);
        w[q] = rnd ? (lo | (hi << 16)) : (0x3f803f80u + (unsigned)i);
    }
    return __builtin_bit_cast(bf16x8, w);
}
template <bool RND> __global__ __launch_bounds__(512) void k16(float* out, int iters) {
    unsigned h = (threadIdx.x + 1u) * 2654435761u ^ (blockIdx.x * 40503u);
    bf16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = mk(h, RND, i);
    for (int i = 0; i < 4; ++i) b[i] = mk(h, RND, i + 8);
    f32x4 acc[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[i], a[j], acc[i][j], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <bool RND> __global__ __launch_bounds__(512) void k32(float* out, int iters) {
    unsigned h = (threadIdx.x + 1u) * 2654435761u ^ (blockIdx.x * 40503u);
    bf16x8 a[2][4], b[2][2];                          // [k sub-step of 16][32-row / 32-col fragment]
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 4; ++i) a[s][i] = mk(h, RND, i);
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 2; ++i) b[s][i] = mk(h, RND, i + 8);
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[s][i], a[s][j], acc[i][j], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    const double fl = 2.0 * 128 * 64 * 32 * (double)iters * 8 * 256;     // per launch: every wave does a 128x64x32 step per iteration
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
#define RUN(K, label) (void)hipEventRecord(e0); hipLaunchKernelGGL(K, dim3(256), dim3(512), 0, 0, out, iters); (void)hipEventRecord(e1); \
        (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); printf("%-34s %7.1f TF\n", label, fl / ms / 1e9);
        RUN(k16<false>, "16x16x32, constant operands")
        RUN(k32<false>, "32x32x16, constant operands")
        RUN(k16<true>, "16x16x32, random operands")
        RUN(k32<true>, "32x32x16, random operands")
        {   // one wave per SIMD (256 threads): can a single wave keep the matrix pipe full?  (half the waves -> half the flops)
            (void)hipEventRecord(e0); hipLaunchKernelGGL(k16<false>, dim3(256), dim3(256), 0, 0, out, iters); (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%-34s %7.1f TF\n", "16x16x32, constant, 1 wave/SIMD", 0.5 * fl / ms / 1e9);
        }
    }
    return 0;
}
