// micro-benchmark: sustained (L2 / Infinity-Cache / HBM) -> LDS DMA rate per CU.
//   part 1: contiguous 64 KiB slabs (best case), 1 or 2 stages in flight
//   part 2: the GEMM's real access shape -- a 256x256 output tile per CU reading a [256 x BK] K-major slab of A and of B
//           (row stride = K*2 bytes) per step, tiles placed 4x8 per XCD like the kernel's grouped order;
//           BK=64 (8 rows x 128 B per wave-DMA) vs BK=32 (16 rows x 64 B, i.e. half-line requests), with the number
//           of steps kept in flight as in a 2-stage / 4-slot / 5-slot ring.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_rate dma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ void dma16(const u32x4& desc, uint32_t lds, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(voff), "s"(lds), "s"(desc) : "memory");
}
__device__ __forceinline__ u32x4 mkdesc(const void* p, size_t bytes) {
    const uint64_t a = (uint64_t)p;
    return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, (uint32_t)(bytes > 0xffffffffull ? 0xffffffffu : bytes), 0x00020000u};
}
template <int INFLIGHT>
__global__ __launch_bounds__(512) void k_contig(const char* buf, size_t bytes, int iters, int nslab, int share) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const u32x4 desc = mkdesc(buf, bytes);
    const uint32_t lbase = (uint32_t)(uintptr_t)smem;
    const int grp = (blockIdx.x >> 3) / share + (blockIdx.x & 7) * 1000;
    for (int it = 0; it < iters; ++it) {
        const uint32_t slab = (uint32_t)(((uint64_t)grp * 7919u + it) % nslab);
        const uint32_t stage = it & 1;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint32_t chunk = p * 8 + wave;
            dma16(desc, lbase + stage * 65536 + chunk * 1024, slab * 65536u + chunk * 1024 + lane * 16);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// BK = 64 or 32; RING = LDS slots of [512 rows x BK] (A and B slabs together); KEEP = steps left in flight at the wait
template <int BK, int RING, int KEEP, bool SWZ = false>
__global__ __launch_bounds__(512) void k_gemm(const char* A, const char* B, int M, int N, int K, int iters) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const u32x4 da = mkdesc(A, (size_t)M * K * 2), db = mkdesc(B, (size_t)N * K * 2);
    const uint32_t lbase = (uint32_t)(uintptr_t)smem;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;          // 32 tiles per XCD: 4 rows x 8 cols
    const int tm = (xcd & 3) * 4 + (idx & 3), tn = (xcd >> 2) * 8 + (idx >> 2);
    constexpr int STEP_BYTES = 512 * BK * 2;                        // A + B slabs of one step
    constexpr int PIECES = STEP_BYTES / 8192;                       // wave-DMAs per thread per step (8 or 4)
    constexpr int ROWB = BK * 2, RPC = 1024 / ROWB;                 // bytes per row, rows per 1 KiB chunk
    const int KT = K / BK;
    for (int it = 0; it < iters; ++it) {
        const int kt = it % KT;
        const uint32_t slot = it % RING;
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const int chunk = p * 8 + wave;                        // 0 .. 2*256*ROWB/1024
            const bool isB = chunk >= (256 * ROWB / 1024);
            const int c2 = isB ? chunk - 256 * ROWB / 1024 : chunk;
            const int row = c2 * RPC + lane / (ROWB / 16);
            const int sl = lane % (ROWB / 16);
            const int slot = SWZ ? (sl ^ ((row >> 1) & (ROWB / 16 - 1))) : sl;    // the GEMM's XOR swizzle: lanes of a row fetch its 16-B pieces permuted
            const uint32_t voff = (uint32_t)(((size_t)((isB ? tn : tm) * 256 + row) * K + (size_t)kt * BK) * 2 + slot * 16);
            const uint32_t l = __builtin_amdgcn_readfirstlane(lbase + slot * STEP_BYTES + chunk * 1024);
            if (p * 8 >= 256 * ROWB / 1024) dma16(db, l, voff); else dma16(da, l, voff);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP * PIECES) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
#define SETSMEM(f) (void)hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)
int main() {
    SETSMEM(k_contig<8>); SETSMEM(k_contig<0>);
    size_t cap = 3ull << 30;
    char* buf; (void)hipMalloc(&buf, cap); (void)hipMemset(buf, 1, cap);
    const int iters = 2000;
    struct Case { const char* name; size_t bytes; int share; } cases[] = {
        {"contig 8 MiB (L2), private", 8ull << 20, 1}, {"contig 192 MiB (MALL), private", 192ull << 20, 1},
        {"contig 192 MiB, 4 CUs share", 192ull << 20, 4}, {"contig 3 GiB (HBM), private", 3ull << 30, 1}};
    for (auto& c : cases) {
        const int nslab = (int)(c.bytes / 65536);
        float m0 = time_ms([&] { hipLaunchKernelGGL(k_contig<0>, dim3(256), dim3(512), 131072, 0, buf, c.bytes, iters, nslab, c.share); });
        float m1 = time_ms([&] { hipLaunchKernelGGL(k_contig<8>, dim3(256), dim3(512), 131072, 0, buf, c.bytes, iters, nslab, c.share); });
        printf("%-34s 1 in flight: %6.1f B/ns/CU (%7.1f ns/64KiB)   2 in flight: %6.1f B/ns/CU (%7.1f ns/64KiB)\n", c.name,
               iters * 65536.0 / (m0 * 1e6), m0 * 1e6 / iters, iters * 65536.0 / (m1 * 1e6), m1 * 1e6 / iters);
    }
    const int M = 16 * 256, N = 16 * 256, K = 4096;                 // 4x4 XCD grid of 4x8-tile patches: 16 x 16 tiles
    char *A = buf, *B = buf + (1ull << 30);
#define RUN(BK, RING, KEEP, ...)                                                                                             \
    {                                                                                                                     \
        SETSMEM((k_gemm<BK, RING, KEEP, ##__VA_ARGS__>));                                                                              \
        const int n = iters * (64 / BK);                                                                                  \
        float ms = time_ms([&] { hipLaunchKernelGGL((k_gemm<BK, RING, KEEP, ##__VA_ARGS__>), dim3(256), dim3(512), 512 * BK * 2 * RING, 0, A, B, M, N, K, n); }); \
        printf("gemm-shaped%s BK=%d ring=%d keep=%d steps in flight: %6.1f B/ns/CU, %7.1f ns per 64-K of a 256x256 tile\n", sizeof(#__VA_ARGS__) > 1 ? " swizzled" : "", BK, RING, KEEP,  \
               n * 512.0 * BK * 2 / (ms * 1e6), ms * 1e6 / iters);                                                        \
    }
    RUN(64, 2, 0) RUN(64, 2, 1) RUN(64, 2, 0, true) RUN(64, 2, 1, true) RUN(32, 4, 1) RUN(32, 4, 2) RUN(32, 4, 3) RUN(32, 5, 3) RUN(32, 5, 4)
    return 0;
}
