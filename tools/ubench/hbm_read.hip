// micro-benchmark: HBM read rate of a weight STREAMER (the decode-step GEMV) as a function of the per-instruction access shape.
//   pattern 0: wave instruction = 1 KiB contiguous (lane l reads 16 B at l*16)                     -- the ideal stream
//   pattern 1: wave instruction = 16 rows x 64 B (lane (n = l&15, kg = l>>4): row n, bytes kg*16)  -- nv_gemv_bf16 today:
//              the MFMA B-operand layout straight from a row-major [N, K] matrix; half-line requests
//   pattern 2: wave instruction = 8 rows x 128 B (lane (n = l>>3, c = l&7))                         -- full lines, strided rows
// Work split as in the GEMV: block = 16 rows (pattern 2: 16 rows as two instructions), 8 waves split K; matrices rotate over
// a footprint > Infinity Cache.  Build: hipcc --offload-arch=gfx950 -O3 -o hbm_read hbm_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int PAT, int UNROLL>
__global__ __launch_bounds__(512) void k_read(const char* __restrict__ W, int N, int Kbytes, unsigned* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = (long)blockIdx.x * 16;
    const int per = Kbytes / 8;                                   // bytes of each row this wave covers
    const int k0 = wave * per;
    u32x4 acc = {0, 0, 0, 0};
    if (PAT == 0) {
        // the block's 16 rows x Kbytes as one contiguous slab (a pre-tiled matrix): wave slice = 16*per bytes
        const char* p = W + row0 * Kbytes + (long)wave * 16 * per + lane * 16;
        const int steps = 16 * per / 1024;
        for (int s = 0; s + UNROLL <= steps; s += UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load((const u32x4*)(p + (long)(s + u) * 1024));
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
        }
    } else if (PAT == 1) {
        const char* p = W + (row0 + (lane & 15)) * Kbytes + k0 + (lane >> 4) * 16;
        const int steps = per / 64;
        for (int s = 0; s + UNROLL <= steps; s += UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load((const u32x4*)(p + (long)(s + u) * 64));
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
        }
    } else {
        const char* p = W + (row0 + (lane >> 3)) * Kbytes + k0 + (lane & 7) * 16;
        const int steps = per / 128;
        for (int s = 0; s + UNROLL / 2 <= steps; s += UNROLL / 2) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL / 2; ++u) {
                v[2 * u] = __builtin_nontemporal_load((const u32x4*)(p + (long)(s + u) * 128));
                v[2 * u + 1] = __builtin_nontemporal_load((const u32x4*)(p + 8l * Kbytes + (long)(s + u) * 128));
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = 1;
}

template <int PAT, int UNROLL>
void run(const char* name, char* buf, size_t total, int N, int K, unsigned* out) {
    const int Kbytes = K * 2;
    const size_t mat = (size_t)N * Kbytes;
    const int nmat = (int)(total / mat);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k_read<PAT, UNROLL>), dim3(N / 16), dim3(512), 0, 0, buf + (size_t)(i % nmat) * mat, N, Kbytes, out);
    hipDeviceSynchronize();
    const int iters = 40;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_read<PAT, UNROLL>), dim3(N / 16), dim3(512), 0, 0, buf + (size_t)(i % nmat) * mat, N, Kbytes, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  %-34s unroll %2d: %7.1f us  %5.2f TB/s\n", name, UNROLL, ms / iters * 1e3, mat / (ms / iters * 1e-3) / 1e12);
}

int main() {
    const size_t total = 3ull << 30;
    char* buf; unsigned* out;
    hipMalloc(&buf, total); hipMalloc(&out, 1 << 20);
    hipMemset(buf, 1, total);
    const int shapes[4][2] = {{12288, 4096}, {4096, 4096}, {22016, 4096}, {4096, 11008}};
    for (auto& sh : shapes) {
        printf("N=%d K=%d (bf16, %.0f MB)\n", sh[0], sh[1], sh[0] * (double)sh[1] * 2 / 1e6);
        run<0, 4>("contiguous 1 KiB / instruction", buf, total, sh[0], sh[1], out);
        run<0, 8>("contiguous 1 KiB / instruction", buf, total, sh[0], sh[1], out);
        run<0, 16>("contiguous 1 KiB / instruction", buf, total, sh[0], sh[1], out);
        run<1, 4>("16 rows x 64 B (gemv today)", buf, total, sh[0], sh[1], out);
        run<1, 8>("16 rows x 64 B (gemv today)", buf, total, sh[0], sh[1], out);
        run<1, 16>("16 rows x 64 B (gemv today)", buf, total, sh[0], sh[1], out);
        run<2, 8>("8 rows x 128 B", buf, total, sh[0], sh[1], out);
        run<2, 16>("8 rows x 128 B", buf, total, sh[0], sh[1], out);
    }
    return 0;
}
