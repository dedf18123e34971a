// Shared device helpers for the NaviLLM gfx950 kernels (CDNA4 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NV_OK 0
#define NV_ERR_ARG (-1)
#define NV_ERR_SHAPE (-2)
#define NV_ERR_LAUNCH (-3)

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(T) __attribute__((address_space(3))) T*
#define GLB_PTR(T) __attribute__((address_space(1))) T*

// ---- bf16 <-> f32, round-to-nearest-even (matches torch's CPU/GPU conversions) ----
// f32 -> bf16 is the gfx950 hardware conversion v_cvt_pk_bf16_f32 (RNE, NaN stays NaN): one instruction per PAIR of
// values where the integer formulation (add 0x7fff + lsb, NaN select, shift) costs ~6 per value -- the softmax-backward
// and row kernels are VALU-bound on exactly this.
typedef __attribute__((ext_vector_type(2))) float nv_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 nv_bf16x2;
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(nv_f32x2{lo, hi}, nv_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }
// round a float to bf16 precision and widen back (a "rounding point")
__device__ __forceinline__ float rbf(float f) { return __uint_as_float(pack2bf(f, 0.f) << 16); }

// ---- buffer resource (bounds-checked, OOB loads return 0) ----
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)bytes, 0x00020000);
}

// ---- LDS DMA (buffer_load_dwordx4 ... lds) issued from inline asm ----------------------------------
// Why asm and not __builtin_amdgcn_raw_ptr_buffer_load_lds: hipcc's waitcnt pass treats every LDS read it
// cannot disambiguate -- in particular ds_read_b64_tr_b16 -- as aliasing ALL pending LDS-DMA and puts an
// `s_waitcnt vmcnt(0)` in front of the first such read of each k-step, i.e. it drains the prefetch pipeline
// every iteration (seen in the .s; cost the dgrad/wgrad GEMMs ~35%).  DMA issued from asm is invisible to
// that pass; completion is tracked by hand with counted `s_waitcnt vmcnt(N)` + s_barrier.
// 16 bytes per lane land at LDS[lds_addr + lane*16]; OOB source lanes write zeros.
__device__ __forceinline__ u32x4 make_desc(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
__device__ __forceinline__ void dma16(const u32x4& desc, uint32_t lds_addr_uniform, uint32_t voff) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(lds_addr_uniform), "s"(desc)
        : "memory");
}
// 4 bytes per lane into a scratch LDS area: used as an L2 PREFETCH (the data is never read; the request
// pulls the 128-B line into this XCD's L2 ahead of the real DMA). Same vmcnt queue as dma16.
__device__ __forceinline__ void dma4(const u32x4& desc, uint32_t lds_addr_uniform, uint32_t voff) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(lds_addr_uniform), "s"(desc)
        : "memory");
}
// variant that leaves M0 overwritten (declared as a clobber, so hipcc re-materialises M0 itself in the rare
// places it needs it): saves two SALU ops per DMA in the GEMM main loop, measured +1..3% there.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16_nosave(const u32x4& desc, uint32_t lds_addr_uniform, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                 :
                 : "v"(voff), "s"(lds_addr_uniform), "s"(desc)
                 : "memory", "m0");
}
// LDS destination = scalar base + compile-time offset, formed straight into M0 (one SALU op, no VALU); source
// offset = voff + soff (soff uniform; it takes part in the hardware bounds check like voff does)
template <int IMM>
__device__ __forceinline__ void dma16_m0imm(const u32x4& desc, uint32_t lds_base_uniform, uint32_t voff, uint32_t soff) {
    asm volatile("s_add_u32 m0, %1, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %4 offen lds"
                 :
                 : "v"(voff), "s"(lds_base_uniform), "s"(desc), "n"(IMM), "s"(soff)
                 : "memory", "m0", "scc");
}
// The same in two halves, so that independent work (a group of MFMAs) can sit between the M0 write and the load that
// consumes it instead of an s_nop: set_m0_imm<IMM>(base) ... dma16_m0set(desc, voff, soff).  Nothing else may write M0
// in between (the GEMM main loop has no other M0 user).
template <int IMM>
__device__ __forceinline__ void set_m0_imm(uint32_t lds_base_uniform) {
    asm volatile("s_add_u32 m0, %0, %1" : : "s"(lds_base_uniform), "n"(IMM) : "memory", "m0", "scc");
}
__device__ __forceinline__ void dma16_m0set(const u32x4& desc, uint32_t voff, uint32_t soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(desc), "s"(soff) : "memory");
}
#pragma clang diagnostic pop
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(LDS_PTR(char) p) { return (uint32_t)(uintptr_t)p; }

// ---- wave reductions (wave = 64) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block reductions for blockDim.x = 64*NW threads; `red` must hold >= NW floats.
// The broadcast result is returned to every thread.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if (NW == 1) return v;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    if (NW == 1) return v;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
    return t;
}

// XCD-aware, bijective block-id remap (8 XCDs, block b runs on XCD b%8): give each XCD a
// contiguous chunk of tile ids so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// hipGetLastError() is per-thread and sticky across *every* runtime call, including the ones torch
// makes (an hipEventQuery that returned hipErrorNotReady, ...): clear it before each launch so that the
// status we report belongs to our own launch.
#define NV_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

static inline int nv_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NV_OK : NV_ERR_LAUNCH;
}

// gemv_stream.hip: the decode-step weight streamer behind nv_gemv_bf16 / nv_gemv_fp8w (library-internal, not part of the C ABI).
// Returns NV_ERR_SHAPE when the shape is outside its fast path (the callers then run their generic kernel).
extern "C" __attribute__((visibility("hidden"))) int nvi_gemv_stream(const void* A, const void* W, const float* scales, void* C,
                                                                    const void* R, int M, int N, int K, int lda, int ldw, int ldc,
                                                                    int ldr, int resid, int fp8, void* stream);
extern "C" int nv_gemv_pre(const void* A, const int* a_rows, const void* W, const float* scales, void* C, const void* R, int M, int N, int K,
                           int lda, int ldw, int ldc, int ldr, int rmsnorm, const void* norm_w, float eps, int swiglu, void* stream);
