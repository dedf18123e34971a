// does the raw-buffer bounds check on gfx950 include the SGPR offset (soffset)?  buffer of 256 B inside a larger
// allocation filled with 7s; lane i loads dword at voff = 4*i with soffset = S.  In-range iff (4*i + S) < 256 when
// soffset is checked; if it is not, lanes with 4*i < 256 read 7s from beyond the 256-B window.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const unsigned* buf, unsigned* out, unsigned S) {
    const uint64_t a = (uint64_t)buf;
    u32x4 d = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, 256u, 0x00020000u};
    unsigned v, voff = threadIdx.x * 4;
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(voff), "s"(d), "s"(S) : "memory");
    out[threadIdx.x] = v;
}
int main() {
    unsigned *buf, *out, h[64];
    (void)hipMalloc(&buf, 1 << 20); (void)hipMalloc(&out, 256);
    (void)hipMemset(buf, 7, 1 << 20);
    for (unsigned S : {0u, 128u, 192u, 256u, 4096u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, buf, out, S);
        (void)hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
        int nz = 0; for (int i = 0; i < 64; ++i) nz += (h[i] != 0);
        printf("soffset=%4u: %2d of 64 lanes non-zero (checked => %d)\n", S, nz, (int)(S >= 256 ? 0 : (256 - S) / 4));
    }
    return 0;
}
