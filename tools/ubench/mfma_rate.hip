// MFMA issue-rate microbenchmark: 16x16x32 vs 32x32x16 bf16, 8 waves per CU (2 per SIMD), 256 blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int N> __global__ __launch_bounds__(512) void k16(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i * 0.01f); }
    f32x4 acc[N];
    for (int i = 0; i < N; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < N; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int N> __global__ __launch_bounds__(512) void k32(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i * 0.01f); }
    f32x16 acc[N];
    for (int i = 0; i < N; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < N; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        int iters = 20000; float ms;
        hipEventRecord(e0); hipLaunchKernelGGL(k16<32>, dim3(256), dim3(512), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        double fl = 2.0 * 16 * 16 * 32 * 32.0 * iters * 8 * 256;
        printf("16x16x32 x32 acc : %.1f TF (%.2f ms)\n", fl / ms / 1e9, ms);
        hipEventRecord(e0); hipLaunchKernelGGL(k32<8>, dim3(256), dim3(512), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        fl = 2.0 * 32 * 32 * 16 * 8.0 * iters * 8 * 256;
        printf("32x32x16 x8 acc  : %.1f TF (%.2f ms)\n", fl / ms / 1e9, ms);
    }
    return 0;
}
