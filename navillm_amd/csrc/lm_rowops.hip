// Row-wise bf16 kernels of the visual-token causal LM (all HBM-bound, 16-B vector access):
//   K6  token-embedding gather + visual-token add          models/modified_lm.py:100-110
//   K7c RMSNorm fwd/bwd (fp32 statistics)                   HF LlamaRMSNorm
//       RoPE fwd/bwd on packed qkv (rotate-half, table)     HF apply_rotary_pos_emb
//       SwiGLU fwd/bwd on packed gate|up                     HF LlamaMLP
// Rounding points follow torch's bf16 eager semantics (each elementwise op rounds its result
// to bf16) so that results track the reference bf16 CPU path op for op.
#include "nv_common.h"

namespace {

__device__ __forceinline__ void ld8(const bf16_t* p, float* f) {
    const u32x4 v = *(const u32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void st8(bf16_t* p, const float* f) {
    u32x4 v = {pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])};
    *(u32x4*)p = v;
}

// ---------------------------------------------------------------- embedding + visual tokens
// out[m,:] = table[ids[m],:]                       if vis_idx[m] < 0
//          = bf16( f32(table[ids[m],:]) + vis[vis_idx[m],:] )   otherwise   (single rounding)
__global__ __launch_bounds__(256) void embed_vis_kernel(const bf16_t* __restrict__ table, const int* __restrict__ ids,
                                                        const int* __restrict__ vis_idx, const float* __restrict__ vis,
                                                        bf16_t* __restrict__ out, int M, int d) {
    const int per_row = d / 8;
    const long total = (long)M * per_row;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int m = (int)(i / per_row), c = (int)(i % per_row) * 8;
        float f[8];
        ld8(table + (long)ids[m] * d + c, f);
        const int vi = vis_idx[m];
        if (vi >= 0) {
            const float* vp = vis + (long)vi * d + c;
            const f32x4 a = *(const f32x4*)vp, b = *(const f32x4*)(vp + 4);
            f[0] += a[0]; f[1] += a[1]; f[2] += a[2]; f[3] += a[3];
            f[4] += b[0]; f[5] += b[1]; f[6] += b[2]; f[7] += b[3];
        }
        st8(out + (long)m * d + c, f);
    }
}

// dvis[i,:] = f32(dE[vis_rows[i],:])
__global__ __launch_bounds__(256) void vis_grad_kernel(const bf16_t* __restrict__ dE, const int* __restrict__ vis_rows,
                                                       float* __restrict__ dvis, int nvis, int d) {
    const int per_row = d / 8;
    const long total = (long)nvis * per_row;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int r = (int)(i / per_row), c = (int)(i % per_row) * 8;
        float f[8];
        ld8(dE + (long)vis_rows[r] * d + c, f);
        float* o = dvis + (long)r * d + c;
        *(f32x4*)o = f32x4{f[0], f[1], f[2], f[3]};
        *(f32x4*)(o + 4) = f32x4{f[4], f[5], f[6], f[7]};
    }
}

// Embedding-table gradient: tokens pre-grouped by id on the host (ids come from the CPU
// tokenizer anyway).  Block u owns table row uniq[u]; sums its token rows in fp32 and adds.
//   gtable[uniq[u],:] = bf16( gtable[uniq[u],:] + bf16(sum_{t in seg u} dE[tok[t],:]) )
__global__ __launch_bounds__(256) void embed_grad_kernel(const bf16_t* __restrict__ dE, const int* __restrict__ uniq,
                                                         const int* __restrict__ seg_off, const int* __restrict__ tok,
                                                         bf16_t* __restrict__ gtable, int d) {
    const int u = blockIdx.x;
    const int beg = seg_off[u], end = seg_off[u + 1];
    bf16_t* g = gtable + (long)uniq[u] * d;
    for (int c = threadIdx.x * 8; c < d; c += 256 * 8) {
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, f[8];
        for (int t = beg; t < end; ++t) {
            ld8(dE + (long)tok[t] * d + c, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += f[j];
        }
        ld8(g + c, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += rbf(s[j]);
        st8(g + c, f);
    }
}

// ---------------------------------------------------------------- RMSNorm
// y = bf16( w * bf16( x * rsqrt(mean(x^2)+eps) ) ) ; rstd saved for backward. One row per block.
template <int NW>
__global__ __launch_bounds__(NW * 64) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                              bf16_t* __restrict__ y, float* __restrict__ rstd_out, int d,
                                                              float eps) {
    __shared__ float red[NW];
    const int m = blockIdx.x;
    const bf16_t* xr = x + (long)m * d;
    float ss = 0.f;
    for (int c = threadIdx.x * 8; c < d; c += NW * 64 * 8) {
        float f[8];
        ld8(xr + c, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
    }
    ss = block_sum<NW>(ss, red);
    const float rstd = rsqrtf(ss / (float)d + eps);
    if (threadIdx.x == 0 && rstd_out) rstd_out[m] = rstd;
    for (int c = threadIdx.x * 8; c < d; c += NW * 64 * 8) {
        float f[8], wf[8];
        ld8(xr + c, f);
        ld8(w + c, wf);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = wf[j] * rbf(f[j] * rstd);
        st8(y + (long)m * d + c, f);
    }
}

// Backward of y = w * xhat, xhat = x*rstd:
//   dxhat = dy*w ; dx = rstd * (dxhat - xhat * mean(dxhat*xhat)) ; dx_out = bf16(dx) (+ resid_grad)
//   dw partial[blk, :] = sum_rows dy * bf16(xhat)   (fp32, reduced by rmsnorm_dw_reduce)
// Each block walks rows blk, blk+grid, ... so the dw partials stay per-block in registers.
template <int NW, int VEC>
__global__ __launch_bounds__(NW * 64) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                              const bf16_t* __restrict__ w, const float* __restrict__ rstd,
                                                              const bf16_t* __restrict__ resid_grad, bf16_t* __restrict__ dx,
                                                              float* __restrict__ dw_part, int M, int d) {
    __shared__ float red[NW];
    float dwacc[VEC][8];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int j = 0; j < 8; ++j) dwacc[v][j] = 0.f;
    for (int m = blockIdx.x; m < M; m += gridDim.x) {
        const float rs = rstd[m];
        float dot = 0.f;
        float xh[VEC][8], dxh[VEC][8];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int c = (v * NW * 64 + threadIdx.x) * 8;
            if (c < d) {
                float xf[8], dyf[8], wf[8];
                ld8(x + (long)m * d + c, xf);
                ld8(dy + (long)m * d + c, dyf);
                ld8(w + c, wf);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xh[v][j] = xf[j] * rs;
                    dxh[v][j] = rbf(dyf[j] * wf[j]);   // autograd hands d(xhat) over in bf16
                    dot += dxh[v][j] * xh[v][j];
                    dwacc[v][j] += dyf[j] * rbf(xh[v][j]);
                }
            }
        }
        dot = block_sum<NW>(dot, red) / (float)d;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int c = (v * NW * 64 + threadIdx.x) * 8;
            if (c < d) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = rs * (dxh[v][j] - xh[v][j] * dot);
                if (resid_grad) {
                    float rg[8];
                    ld8(resid_grad + (long)m * d + c, rg);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = rg[j] + rbf(o[j]);
                }
                st8(dx + (long)m * d + c, o);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const int c = (v * NW * 64 + threadIdx.x) * 8;
        if (c < d) {
            float* o = dw_part + (long)blockIdx.x * d + c;
            *(f32x4*)o = f32x4{dwacc[v][0], dwacc[v][1], dwacc[v][2], dwacc[v][3]};
            *(f32x4*)(o + 4) = f32x4{dwacc[v][4], dwacc[v][5], dwacc[v][6], dwacc[v][7]};
        }
    }
}

// gw[c] = bf16( gw[c] + bf16( sum_p part[p,c] ) ) ; block = 16 columns x 16 row slices (d/16 blocks: one per CU at
// d = 4096; the 64-column x 4-slice form had only 64 blocks and took longer than the backward kernel itself)
__global__ __launch_bounds__(256) void dw_reduce_bf16_kernel(const float* __restrict__ part, bf16_t* __restrict__ gw, int P,
                                                             int d) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < d) {
        int p = sl;
        for (; p + 48 < P; p += 64) {
            s0 += part[(long)p * d + c];
            s1 += part[(long)(p + 16) * d + c];
            s2 += part[(long)(p + 32) * d + c];
            s3 += part[(long)(p + 48) * d + c];
        }
        for (; p < P; p += 16) s0 += part[(long)p * d + c];
    }
    red[sl][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && c < d) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[q][cl];
        gw[c] = f2bf(bf2f(gw[c]) + rbf(s));
    }
}

// ---------------------------------------------------------------- RoPE on packed qkv
// qkv: [M, 3*H*hd] ; rotates q and k heads in place. position of row m is m % S (arange(S)
// including left padding).  cs: [maxS, hd] bf16 cos | [maxS, hd] bf16 sin rows as HF
// builds them (halves duplicated).  out = bf16( bf16(x*cos) + bf16(rot_half(x)*sin) );
// sign=-1 gives the transpose (backward).
// With `pos` (int32 [M]) the position of row m is pos[m] instead (KV-cache inference: left-aligned per-sample
// frames, the position_ids of HF's generation path).
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ qkv, const bf16_t* __restrict__ cos_t,
                                                   const bf16_t* __restrict__ sin_t, const int* __restrict__ pos, int M, int S,
                                                   int H, int hd, int ld, float sign) {
    const int half = hd / 2;
    const int per_head = half / 8;                 // 8-wide vectors in the first half
    const long total = (long)M * 2 * H * per_head;  // q and k heads
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int v = (int)(i % per_head);
        long r = i / per_head;
        const int head = (int)(r % (2 * H));       // 0..H-1 = q heads, H..2H-1 = k heads
        const int m = (int)(r / (2 * H));
        const int s = pos ? pos[m] : m % S;
        bf16_t* base = qkv + (long)m * ld + (long)head * hd + v * 8;
        float a[8], b[8], c[8], sn[8];
        ld8(base, a);
        ld8(base + half, b);
        ld8(cos_t + (long)s * hd + v * 8, c);
        ld8(sin_t + (long)s * hd + v * 8, sn);
        float o1[8], o2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o1[j] = rbf(a[j] * c[j]) + rbf(-sign * b[j] * sn[j]);
            o2[j] = rbf(b[j] * c[j]) + rbf(sign * a[j] * sn[j]);
        }
        st8(base, o1);
        st8(base + half, o2);
    }
}

// RoPE of the new rows fused with their scatter into the K/V cache (decode / K/V-reuse steps: nv_rope_rows_bf16 followed by
// nv_scatter_rows_bf16 in one launch): dst[rows[m], :] = [rope(q) | rope(k) | v] of src row m at position pos[m].  Same
// arithmetic as rope_kernel (sign = +1); src is left untouched.
__global__ __launch_bounds__(256) void rope_scatter_kernel(const bf16_t* __restrict__ src, const bf16_t* __restrict__ cos_t,
                                                           const bf16_t* __restrict__ sin_t, const int* __restrict__ pos,
                                                           const int* __restrict__ rows, bf16_t* __restrict__ dst, int M, int H, int hd,
                                                           int ld) {
    const int half = hd / 2;
    const int per_head = half / 8;
    const long n_rope = (long)M * 2 * H * per_head;                // q and k heads: pairs of 8-wide vectors
    const int v_per_row = H * hd / 8;
    const long total = n_rope + (long)M * v_per_row;               // + the v part: plain 16-B copies
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        if (i < n_rope) {
            const int v = (int)(i % per_head);
            long r = i / per_head;
            const int head = (int)(r % (2 * H));
            const int m = (int)(r / (2 * H));
            const int s = pos[m];
            const long off = (long)head * hd + v * 8;
            const bf16_t* base = src + (long)m * ld + off;
            bf16_t* out = dst + (long)rows[m] * ld + off;
            float a[8], b[8], c[8], sn[8];
            ld8(base, a);
            ld8(base + half, b);
            ld8(cos_t + (long)s * hd + v * 8, c);
            ld8(sin_t + (long)s * hd + v * 8, sn);
            float o1[8], o2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o1[j] = rbf(a[j] * c[j]) + rbf(-1.f * b[j] * sn[j]);
                o2[j] = rbf(b[j] * c[j]) + rbf(1.f * a[j] * sn[j]);
            }
            st8(out, o1);
            st8(out + half, o2);
        } else {
            const long k = i - n_rope;
            const int m = (int)(k / v_per_row), c = (int)(k % v_per_row) * 8;
            const long off = 2L * H * hd + c;
            *(u32x4*)(dst + (long)rows[m] * ld + off) = *(const u32x4*)(src + (long)m * ld + off);
        }
    }
}

// ---------------------------------------------------------------- SwiGLU on packed gate|up
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ h, int M, int ff) {
    const int per_row = ff / 8;
    const long total = (long)M * per_row;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int m = (int)(i / per_row), c = (int)(i % per_row) * 8;
        float g[8], u[8];
        ld8(gu + (long)m * 2 * ff + c, g);
        ld8(gu + (long)m * 2 * ff + ff + c, u);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = rbf(silu_f(g[j])) * u[j];
        st8(h + (long)m * ff + c, g);
    }
}

// dgu[:, :ff] = bf16( bf16(dh*u) * silu'(g) ) ; dgu[:, ff:] = bf16( dh * bf16(silu(g)) )
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dh,
                                                         bf16_t* __restrict__ dgu, int M, int ff) {
    const int per_row = ff / 8;
    const long total = (long)M * per_row;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int m = (int)(i / per_row), c = (int)(i % per_row) * 8;
        float g[8], u[8], dv[8], dg[8], du[8];
        ld8(gu + (long)m * 2 * ff + c, g);
        ld8(gu + (long)m * 2 * ff + ff + c, u);
        ld8(dh + (long)m * ff + c, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sg = 1.f / (1.f + expf(-g[j]));
            du[j] = dv[j] * rbf(g[j] * sg);
            dg[j] = rbf(dv[j] * u[j]) * (sg * (1.f + g[j] * (1.f - sg)));
        }
        st8(dgu + (long)m * 2 * ff + c, dg);
        st8(dgu + (long)m * 2 * ff + ff + c, du);
    }
}

// rows gather: out[i,:] = src[rows[i],:]  (bf16, d%8==0)
__global__ __launch_bounds__(256) void gather_rows_bf16_kernel(const bf16_t* __restrict__ src, const int* __restrict__ rows,
                                                               bf16_t* __restrict__ out, int n, int d) {
    const int per_row = d / 8;
    const long total = (long)n * per_row;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int r = (int)(i / per_row), c = (int)(i % per_row) * 8;
        *(u32x4*)(out + (long)r * d + c) = *(const u32x4*)(src + (long)rows[r] * d + c);
    }
}
// rows scatter (unique rows): dst[rows[i],:] = src[i,:]
__global__ __launch_bounds__(256) void scatter_rows_bf16_kernel(const bf16_t* __restrict__ src, const int* __restrict__ rows,
                                                                bf16_t* __restrict__ dst, int n, int d) {
    const int per_row = d / 8;
    const long total = (long)n * per_row;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int r = (int)(i / per_row), c = (int)(i % per_row) * 8;
        *(u32x4*)(dst + (long)rows[r] * d + c) = *(const u32x4*)(src + (long)r * d + c);
    }
}

// x *= scale (bf16, in place or out of place), n % 8 == 0
__global__ __launch_bounds__(256) void scale_bf16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, long n8,
                                                         float scale) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        float f[8];
        ld8(x + i * 8, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= scale;
        st8(out + i * 8, f);
    }
}

// out = bf16(x * g)  or  out = bf16(out + bf16(x * g)), g read from device memory (the loss coefficient that arrives as the
// upstream gradient of the LM loss: it never visits the host)
__global__ __launch_bounds__(256) void scale_dev_bf16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, long n8,
                                                             const float* __restrict__ g, int accumulate) {
    const float scale = *g;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        float f[8];
        ld8(x + i * 8, f);
        if (accumulate) {
            float o[8];
            ld8(out + i * 8, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = o[j] + rbf(f[j] * scale);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] *= scale;
        }
        st8(out + i * 8, f);
    }
}

__global__ __launch_bounds__(256) void kv_grad_accum_kernel(const bf16_t* __restrict__ dqkv, float* __restrict__ acc,
                                                            const int* __restrict__ rows, int n, int d) {
    const int per_row = 2 * d / 8;
    const long total = (long)n * per_row;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const long r = rows[i / per_row];
        const int c = (int)(i % per_row) * 8;
        float f[8];
        ld8(dqkv + r * 3 * d + d + c, f);
        float* a = acc + r * 2 * d + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += f[j];
    }
}
// the first step of an episode: acc[rows] = f32(dqkv K/V columns) -- no read of (and therefore no zero-fill of) the accumulator
__global__ __launch_bounds__(256) void kv_grad_set_kernel(const bf16_t* __restrict__ dqkv, float* __restrict__ acc,
                                                          const int* __restrict__ rows, int n, int d) {
    const int per_row = 2 * d / 8;
    const long total = (long)n * per_row;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const long r = rows[i / per_row];
        const int c = (int)(i % per_row) * 8;
        float f[8];
        ld8(dqkv + r * 3 * d + d + c, f);
        float* a = acc + r * 2 * d + c;
        *(f32x4*)a = f32x4{f[0], f[1], f[2], f[3]};
        *(f32x4*)(a + 4) = f32x4{f[4], f[5], f[6], f[7]};
    }
}
__global__ __launch_bounds__(256) void kv_grad_inject_kernel(bf16_t* __restrict__ dqkv, const float* __restrict__ acc,
                                                             const int* __restrict__ rows, int n, int d) {
    const int per_row = 2 * d / 8;
    const long total = (long)n * per_row;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const long row = i / per_row, r = rows[row];
        const int c = (int)(i % per_row) * 8;
        float f[8];
        bf16_t* q = dqkv + row * 3 * d + d + c;
        ld8(q, f);
        const float* a = acc + r * 2 * d + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += a[j];
        st8(q, f);
    }
}

inline int grid_for(long total, int cap = 256 * 8) {
    long b = (total + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

}  // namespace

extern "C" {

int nv_embed_vis_bf16(const void* table, const int* ids, const int* vis_idx, const float* vis, void* out, int M, int d,
                      void* stream) {
    if (!table || !ids || !vis_idx || !out || (d & 7)) return NV_ERR_ARG;
    if (M == 0) return NV_OK;
    NV_LAUNCH(embed_vis_kernel, dim3(grid_for((long)M * d / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)table, ids, vis_idx, vis, (bf16_t*)out, M, d);
    return nv_check_launch();
}

int nv_vis_grad_f32(const void* dE, const int* vis_rows, float* dvis, int nvis, int d, void* stream) {
    if (!dE || !vis_rows || !dvis || (d & 7)) return NV_ERR_ARG;
    if (nvis == 0) return NV_OK;
    NV_LAUNCH(vis_grad_kernel, dim3(grid_for((long)nvis * d / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dE, vis_rows, dvis, nvis, d);
    return nv_check_launch();
}

int nv_embed_grad_bf16(const void* dE, const int* uniq, const int* seg_off, const int* tok, void* gtable, int n_uniq, int d,
                       void* stream) {
    if (!dE || !uniq || !seg_off || !tok || !gtable || (d & 7)) return NV_ERR_ARG;
    if (n_uniq == 0) return NV_OK;
    NV_LAUNCH(embed_grad_kernel, dim3(n_uniq), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dE, uniq, seg_off,
                       tok, (bf16_t*)gtable, d);
    return nv_check_launch();
}

int nv_rmsnorm_fwd_bf16(const void* x, const void* w, void* y, float* rstd, int M, int d, float eps, void* stream) {
    if (!x || !w || !y || (d & 7)) return NV_ERR_ARG;
    if (M == 0) return NV_OK;
    NV_LAUNCH(rmsnorm_fwd_kernel<4>, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w,
                       (bf16_t*)y, rstd, d, eps);
    return nv_check_launch();
}

// workspace: nv_rmsnorm_bwd_workspace_bytes(d) bytes of fp32 partials
size_t nv_rmsnorm_bwd_workspace_bytes(int d) { return (size_t)1024 * d * sizeof(float); }

int nv_rmsnorm_bwd_bf16(const void* dy, const void* x, const void* w, const float* rstd, const void* resid_grad, void* dx,
                        void* gw, void* workspace, int M, int d, void* stream) {
    if (!dy || !x || !w || !rstd || !dx || !gw || !workspace || (d & 7)) return NV_ERR_ARG;
    if (d > 4 * 256 * 8) return NV_ERR_SHAPE;  // VEC=4 covers d <= 8192
    if (M == 0) return NV_OK;
    // 4 blocks per CU: the kernel is HBM-bound and needs the occupancy to cover latency (88 -> ~45 us at M=5152)
    const int P = M < 1024 ? M : 1024;
    NV_LAUNCH((rmsnorm_bwd_kernel<4, 4>), dim3(P), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)x, (const bf16_t*)w, rstd, (const bf16_t*)resid_grad, (bf16_t*)dx, (float*)workspace, M,
                       d);
    NV_LAUNCH(dw_reduce_bf16_kernel, dim3((d + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, (bf16_t*)gw, P, d);
    return nv_check_launch();
}

int nv_rope_bf16(void* qkv, const void* cos_t, const void* sin_t, int M, int S, int H, int hd, int ld, int backward,
                 void* stream) {
    if (!qkv || !cos_t || !sin_t || (hd & 15) || S <= 0) return NV_ERR_ARG;
    if (M == 0) return NV_OK;
    NV_LAUNCH(rope_kernel, dim3(grid_for((long)M * 2 * H * hd / 16)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)qkv, (const bf16_t*)cos_t, (const bf16_t*)sin_t, (const int*)nullptr, M, S, H, hd, ld,
                       backward ? -1.f : 1.f);
    return nv_check_launch();
}

int nv_rope_rows_bf16(void* qkv, const void* cos_t, const void* sin_t, const int* pos, int M, int H, int hd, int ld, void* stream) {
    if (!qkv || !cos_t || !sin_t || !pos || (hd & 15)) return NV_ERR_ARG;
    if (M == 0) return NV_OK;
    NV_LAUNCH(rope_kernel, dim3(grid_for((long)M * 2 * H * hd / 16)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)qkv, (const bf16_t*)cos_t, (const bf16_t*)sin_t, pos, M, 1, H, hd, ld, 1.f);
    return nv_check_launch();
}

int nv_rope_scatter_rows_bf16(const void* qkv, const void* cos_t, const void* sin_t, const int* pos, const int* rows, void* dst, int M, int H,
                              int hd, int ld, void* stream) {
    if (!qkv || !cos_t || !sin_t || !pos || !rows || !dst || (hd & 15) || ld != 3 * H * hd) return NV_ERR_ARG;
    if (M == 0) return NV_OK;
    NV_LAUNCH(rope_scatter_kernel, dim3(grid_for((long)M * (2 * H * hd / 16 + H * hd / 8))), dim3(256), 0, (hipStream_t)stream,
              (const bf16_t*)qkv, (const bf16_t*)cos_t, (const bf16_t*)sin_t, pos, rows, (bf16_t*)dst, M, H, hd, ld);
    return nv_check_launch();
}

// transpose of nv_rope_rows_bf16 (the rotation by -theta): backward of the per-row-position RoPE
int nv_rope_rows_t_bf16(void* qkv, const void* cos_t, const void* sin_t, const int* pos, int M, int H, int hd, int ld, void* stream) {
    if (!qkv || !cos_t || !sin_t || !pos || (hd & 15)) return NV_ERR_ARG;
    if (M == 0) return NV_OK;
    NV_LAUNCH(rope_kernel, dim3(grid_for((long)M * 2 * H * hd / 16)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)qkv, (const bf16_t*)cos_t, (const bf16_t*)sin_t, pos, M, 1, H, hd, ld, -1.f);
    return nv_check_launch();
}

// Training with a cached prompt prefix (navillm_amd/episode.py): the K/V gradients that the steps of an episode send into the
// prefix rows are summed in fp32 and handed to the prefix's own (deferred) backward.
//   accum : acc[rows[i], :] += f32(dqkv[rows[i], d .. 3d))          (acc [*, 2d] fp32, dqkv [*, 3d] bf16, same row index)
//   inject: dqkv[i, d .. 3d) = bf16(f32(dqkv[i, d .. 3d)) + acc[rows[i], :])    (dqkv packed prefix rows, acc cache rows)
int nv_kv_grad_accum_f32(const void* dqkv, float* acc, const int* rows, int n, int d, void* stream) {
    if (!dqkv || !acc || !rows || (d & 7)) return NV_ERR_ARG;
    if (n == 0) return NV_OK;
    NV_LAUNCH(kv_grad_accum_kernel, dim3(grid_for((long)n * 2 * d / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dqkv, acc, rows,
              n, d);
    return nv_check_launch();
}
int nv_kv_grad_set_f32(const void* dqkv, float* acc, const int* rows, int n, int d, void* stream) {
    if (!dqkv || !acc || !rows || (d & 7)) return NV_ERR_ARG;
    if (n == 0) return NV_OK;
    NV_LAUNCH(kv_grad_set_kernel, dim3(grid_for((long)n * 2 * d / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dqkv, acc, rows, n,
              d);
    return nv_check_launch();
}
int nv_kv_grad_inject_bf16(void* dqkv, const float* acc, const int* rows, int n, int d, void* stream) {
    if (!dqkv || !acc || !rows || (d & 7)) return NV_ERR_ARG;
    if (n == 0) return NV_OK;
    NV_LAUNCH(kv_grad_inject_kernel, dim3(grid_for((long)n * 2 * d / 8)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)dqkv, acc, rows, n, d);
    return nv_check_launch();
}

int nv_swiglu_fwd_bf16(const void* gu, void* h, int M, int ff, void* stream) {
    if (!gu || !h || (ff & 7)) return NV_ERR_ARG;
    if (M == 0) return NV_OK;
    NV_LAUNCH(swiglu_fwd_kernel, dim3(grid_for((long)M * ff / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)gu, (bf16_t*)h, M, ff);
    return nv_check_launch();
}

int nv_swiglu_bwd_bf16(const void* gu, const void* dh, void* dgu, int M, int ff, void* stream) {
    if (!gu || !dh || !dgu || (ff & 7)) return NV_ERR_ARG;
    if (M == 0) return NV_OK;
    NV_LAUNCH(swiglu_bwd_kernel, dim3(grid_for((long)M * ff / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)gu, (const bf16_t*)dh, (bf16_t*)dgu, M, ff);
    return nv_check_launch();
}

int nv_gather_rows_bf16(const void* src, const int* rows, void* out, int n, int d, void* stream) {
    if (!src || !rows || !out || (d & 7)) return NV_ERR_ARG;
    if (n == 0) return NV_OK;
    NV_LAUNCH(gather_rows_bf16_kernel, dim3(grid_for((long)n * d / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, rows, (bf16_t*)out, n, d);
    return nv_check_launch();
}

int nv_scale_bf16(const void* x, void* out, long n, float scale, void* stream) {
    if (!x || !out || (n & 7)) return NV_ERR_ARG;
    if (n == 0) return NV_OK;
    NV_LAUNCH(scale_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)out, n / 8, scale);
    return nv_check_launch();
}

int nv_scale_dev_bf16(const void* x, void* out, long n, const float* scale_dev, int accumulate, void* stream) {
    if (!x || !out || !scale_dev || (n & 7)) return NV_ERR_ARG;
    if (n == 0) return NV_OK;
    NV_LAUNCH(scale_dev_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)out, n / 8, scale_dev, accumulate);
    return nv_check_launch();
}

int nv_scatter_rows_bf16(const void* src, const int* rows, void* dst, int n, int d, void* stream) {
    if (!src || !rows || !dst || (d & 7)) return NV_ERR_ARG;
    if (n == 0) return NV_OK;
    NV_LAUNCH(scatter_rows_bf16_kernel, dim3(grid_for((long)n * d / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, rows, (bf16_t*)dst, n, d);
    return nv_check_launch();
}

}  // extern "C"
