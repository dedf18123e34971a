// micro-benchmark: what one K-step (64 k of a 256x256 tile: 64 MFMA 16x16x32 + 24 ds_read_b128 + 8 wave-DMA per wave,
// 8 waves) costs when the three streams are merely co-issued with no data dependence between DMA and reads
// (upper bound for the GEMM main loop). Variants switch the streams on/off.
// Build: hipcc --offload-arch=gfx950 -O3 -o mix_rate mix_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ void dma16(const u32x4& desc, uint32_t lds, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(voff), "s"(lds), "s"(desc) : "memory");
}
__device__ __forceinline__ u32x4 mkdesc(const void* p, size_t bytes) {
    const uint64_t a = (uint64_t)p;
    return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, (uint32_t)bytes, 0x00020000u};
}
template <bool DMA, bool RD, bool MMA, int KEEP, bool BAR, bool RND = false>
__global__ __launch_bounds__(512) void k_mix(const char* A, const char* B, int K, int iters, float* out) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const u32x4 da = mkdesc(A, (size_t)4096 * K * 2), db = mkdesc(B, (size_t)4096 * K * 2);
    const uint32_t lbase = (uint32_t)(uintptr_t)smem;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tm = (xcd & 3) * 4 + (idx & 3), tn = (xcd >> 2) * 8 + (idx >> 2);
    const int KT = K / 64;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[8], fb[4];
    // RND: operands = pseudo-random bf16 in (-2, 2) (N(0,1)-like bit activity) instead of zeros: MFMA power, hence the
    // sustained clock, depends on the data
    unsigned h = (threadIdx.x + 1u) * 2654435761u ^ (blockIdx.x * 40503u);
    auto rnd8 = [&]() {
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        u4 w;
        for (int q = 0; q < 4; ++q) {
            h = h * 1664525u + 1013904223u;
            const unsigned lo = 0x3f00u | ((h >> 8) & 0x80ffu), hi = 0x3f00u | ((h >> 16) & 0x80ffu);   // sign + 0.5..2 magnitude
            w[q] = RND ? (lo | (hi << 16)) : 0u;
        }
        return __builtin_bit_cast(bf16x8, w);
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = rnd8();
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = rnd8();
    typedef __attribute__((address_space(3))) bf16x8* lp;
    for (int it = 0; it < iters; ++it) {
        const int kt = it % KT;
        const uint32_t stage = it & 1;
        const char __attribute__((address_space(3)))* rd = (const char __attribute__((address_space(3)))*)smem + (stage ^ 1) * 65536 + lane * 16;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (RD) {
                    fa[c] = *(lp)(rd + (half * 12 + c) * 1024 + wave * 128);
                    if (c < 4) fb[c] = *(lp)(rd + (half * 12 + 8 + c) * 1024 + 32768);
                }
                if (DMA && half == 1) {
                    const int chunk = c * 8 + wave;
                    const bool isB = chunk >= 32;
                    const int c2 = isB ? chunk - 32 : chunk;
                    const int row = c2 * 8 + lane / 8;
                    const uint32_t voff = (uint32_t)(((size_t)((isB ? tn : tm) * 256 + row) * K + (size_t)kt * 64) * 2 + (lane % 8) * 16);
                    dma16(isB ? db : da, lbase + stage * 65536 + chunk * 1024, voff);
                }
                if (MMA) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[(c >> 1) * 8 + (c & 1) * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[c >> 1], fa[(c & 1) * 4 + e], acc[(c >> 1) * 8 + (c & 1) * 4 + e], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (half == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP * 8) : "memory");
                if (BAR) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 32; ++i) s += acc[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[tid] = s[0];
    if (blockIdx.x == 0 && tid == 0) {     // average shader clock over the kernel: core-clock counter vs the 100 MHz wall clock
        const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
        ((unsigned long long*)out)[512] = c1 - c0;
        ((unsigned long long*)out)[513] = w1 - w0;
    }
}
template <typename F>
static float time_ms(F launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    return best;
}
int main() {
    char* buf; (void)hipMalloc(&buf, 2ull << 30); (void)hipMemset(buf, 0, 2ull << 30);
    float* out; (void)hipMalloc(&out, 16384);
    const int K = 4096, iters = 2000;
    char *A = buf, *B = buf + (1ull << 30);
#define RUN(DMA, RD, MMA, KEEP, BAR, label, ...)                                                                     \
    {                                                                                                              \
        (void)hipFuncSetAttribute((const void*)k_mix<DMA, RD, MMA, KEEP, BAR, ##__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
        float ms = time_ms([&] { hipLaunchKernelGGL((k_mix<DMA, RD, MMA, KEEP, BAR, ##__VA_ARGS__>), dim3(256), dim3(512), 131072, 0, A, B, K, iters, out); }); \
        unsigned long long ck[2]; (void)hipMemcpy(ck, (char*)out + 4096, 16, hipMemcpyDeviceToHost);                 \
        printf("%-44s %7.1f ns per K-step  (= %6.0f TF if it were a GEMM)  shader clock %.2f GHz\n", label, ms * 1e6 / iters, 256.0 * 2 * 256 * 256 * 64 / (ms * 1e6 / iters) / 1e3, ck[1] ? 0.1 * ck[0] / ck[1] : 0.0); \
    }
    RUN(false, false, true, 0, false, "MFMA only (zero operands)")
    RUN(false, false, true, 0, false, "MFMA only (random operands)", true)
    RUN(false, false, true, 0, true, "MFMA + barrier (random operands)", true)
    RUN(false, true, true, 0, false, "MFMA + ds_read")
    RUN(false, true, true, 0, true, "MFMA + ds_read + barrier")
    RUN(true, false, false, 0, true, "DMA only, wait all at mid + barrier")
    RUN(true, false, false, 1, true, "DMA only, 1 step in flight at mid + barrier")
    RUN(true, false, true, 0, true, "MFMA + DMA (wait all at mid) + barrier")
    RUN(true, false, true, 1, true, "MFMA + DMA (1 step in flight) + barrier")
    RUN(true, true, false, 0, true, "ds_read + DMA (wait all) + barrier")
    RUN(true, true, true, 0, true, "MFMA + ds_read + DMA (wait all) + barrier")
    RUN(true, true, true, 0, true, "MFMA + ds_read + DMA (wait all), random", true)
    RUN(true, true, true, 1, true, "MFMA + ds_read + DMA (1 in flight) + barrier")
    RUN(true, true, true, 1, false, "MFMA + ds_read + DMA (1 in flight), no barrier")
    return 0;
}
