// Heads, losses and the optimizer step (K10, K11, K9, K13 in SURVEY.md §2.2).
//   action / object head   models/nav_model.py:237,445  (Linear d->100 in the LM dtype)
//   action CE              train.py:229 CrossEntropyLoss(ignore_index=-100, reduction='sum')
//   LM token CE            models/modified_lm.py:122-137 (special ids -> -inf, shifted labels, mean)
//   clip + AdamW           train.py:86-89, tools/optims.py:43-45 (bf16 params with bf16 state)
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

// ---------------------------------------------------------------- small head: y[b,n] = x[b,:].W[n,:] + bias[n]
// grid (B), 256 threads; wave w computes outputs n = w, w+4, ...
__global__ __launch_bounds__(256) void head_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                       const bf16_t* __restrict__ bias, bf16_t* __restrict__ y, int d, int N) {
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bf16_t* xr = x + (long)b * d;
    for (int n = wave; n < N; n += 4) {
        float acc = 0.f;
        for (int c = lane * 8; c < d; c += 64 * 8) {
            float xf[8], wf[8];
            ld8(xr + c, xf);
            ld8(W + (long)n * d + c, wf);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += xf[j] * wf[j];
        }
        acc = wave_sum(acc);
        if (lane == 0) y[(long)b * N + n] = f2bf(acc + bf2f(bias[n]));
    }
}
// dx[b,c] = sum_n dy[b,n] W[n,c]
__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ W,
                                                          bf16_t* __restrict__ dx, int d, int N) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += bf2f(dy[(long)b * N + n]) * bf2f(W[(long)n * d + c]);
    dx[(long)b * d + c] = f2bf(acc);
}
// gW[n,c] = bf16(gW + bf16(sum_b dy[b,n] x[b,c])) ; gb[n] = bf16(gb + bf16(sum_b dy[b,n]))
__global__ __launch_bounds__(256) void head_bwd_dw_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                          bf16_t* __restrict__ gW, bf16_t* __restrict__ gb, int B, int d, int N) {
    const int n = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < d) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += bf2f(dy[(long)b * N + n]) * bf2f(x[(long)b * d + c]);
        gW[(long)n * d + c] = f2bf(bf2f(gW[(long)n * d + c]) + rbf(acc));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += bf2f(dy[(long)b * N + n]);
        gb[n] = f2bf(bf2f(gb[n]) + rbf(s));
    }
}

// ---------------------------------------------------------------- action CE (sum, ignore -100), rows are short (G <= ~100)
// one wave per row; loss_rows[b] = -log_softmax(logits[b])[t] (0 if ignored);
// dlogits[b,j] = bf16(gscale * (p_j - [j==t])) (0 if ignored / -inf slots)
__global__ __launch_bounds__(64) void action_ce_kernel(const bf16_t* __restrict__ logits, const long* __restrict__ targets,
                                                       float* __restrict__ loss_rows, bf16_t* __restrict__ dlogits, int G,
                                                       float gscale, const float* __restrict__ gscale_dev) {
    if (gscale_dev) gscale *= gscale_dev[0];
    const int b = blockIdx.x, lane = threadIdx.x;
    const long t = targets[b];
    const bf16_t* lr = logits + (long)b * G;
    float mx = -INFINITY;
    for (int j = lane; j < G; j += 64) mx = fmaxf(mx, bf2f(lr[j]));
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < G; j += 64) s += expf(bf2f(lr[j]) - mx);
    s = wave_sum(s);
    const float lse = mx + logf(s);
    const bool ign = (t < 0 || t >= G);
    if (lane == 0) loss_rows[b] = ign ? 0.f : (lse - bf2f(lr[t]));
    if (dlogits)
        for (int j = lane; j < G; j += 64) {
            float g = 0.f;
            if (!ign) g = gscale * (expf(bf2f(lr[j]) - lse) - (j == t ? 1.f : 0.f));
            dlogits[(long)b * G + j] = f2bf(g);
        }
}

// ---------------------------------------------------------------- LM token CE on materialised logits
// row m: logits[m, 0..V) bf16 (row stride ldl), the `nspecial` ids starting at special0 count as -inf;
// label[m] (already shifted on the host) or -100. loss_rows[m] = nll ; logits are overwritten by
// dlogits = bf16(gscale * (p - onehot)) when write_grad.
__global__ __launch_bounds__(256) void lm_ce_kernel(bf16_t* __restrict__ logits, const int* __restrict__ labels,
                                                    float* __restrict__ loss_rows, int V, int ldl, int special0, int nspecial,
                                                    float gscale, int write_grad) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    bf16_t* lr = logits + (long)m * ldl;
    const int t = labels[m];
    if (t < 0) {
        if (threadIdx.x == 0) loss_rows[m] = 0.f;
        if (write_grad)
            for (int j = threadIdx.x; j < V; j += 256) lr[j] = 0;
        return;
    }
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < V; j += 256) {
        const bool sp = (j >= special0 && j < special0 + nspecial);
        if (!sp) mx = fmaxf(mx, bf2f(lr[j]));
    }
    mx = block_max<4>(mx, red);
    float s = 0.f;
    for (int j = threadIdx.x; j < V; j += 256) {
        const bool sp = (j >= special0 && j < special0 + nspecial);
        if (!sp) s += expf(bf2f(lr[j]) - mx);
    }
    s = block_sum<4>(s, red);
    const float lse = mx + logf(s);
    const float lt = bf2f(lr[t]);
    __syncthreads();
    if (threadIdx.x == 0) loss_rows[m] = lse - lt;
    if (write_grad)
        for (int j = threadIdx.x; j < V; j += 256) {
            const bool sp = (j >= special0 && j < special0 + nspecial);
            const float g = sp ? 0.f : gscale * (expf(bf2f(lr[j]) - lse) - (j == t ? 1.f : 0.f));
            lr[j] = f2bf(g);
        }
}

// ---------------------------------------------------------------- grad norm + AdamW on flat buffers
template <typename T> __device__ __forceinline__ float ldf(const T* p, long i);
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p, long i) { return bf2f(p[i]); }
template <> __device__ __forceinline__ float ldf<float>(const float* p, long i) { return p[i]; }
template <typename T> __device__ __forceinline__ void stf(T* p, long i, float v);
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, long i, float v) { p[i] = f2bf(v); }
template <> __device__ __forceinline__ void stf<float>(float* p, long i, float v) { p[i] = v; }
template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) { return rbf(v); }
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }

// partial[blk] = sum of squares of this block's grid-stride slice (fp32)
template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T* __restrict__ g, long n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    if (sizeof(T) == 2) {
        const long n8 = n / 8;
        for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
            float f[8];
            ld8((const bf16_t*)g + i * 8, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += f[j] * f[j];
        }
        for (long i = n8 * 8 + blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) { const float v = ldf<T>(g, i); s += v * v; }
    } else {
        for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) { const float v = ldf<T>(g, i); s += v * v; }
    }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// out[0] = total_norm, out[1] = clip_coef = min(1, max_norm/(total+1e-6))
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partial, int n, float max_norm,
                                                        float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) {
        const float tot = sqrtf(s);
        out[0] = tot;
        const float c = max_norm / (tot + 1e-6f);
        out[1] = c < 1.f ? c : 1.f;
    }
}

// torch.optim.AdamW single-tensor update, every intermediate rounded to the storage dtype
// (oracle/navillm_oracle.py: adamw_step_), gradient pre-scaled by the clip coefficient.
// one element of torch.optim.AdamW's single-tensor update; every intermediate is rounded to the storage dtype exactly where torch
// rounds it (lerp_, mul_, addcmul_, sqrt / div / add_, addcdiv_)
template <typename T>
__device__ __forceinline__ void adamw_elem(float& pi, float g_raw, float& mi, float& vi, float coef, float decay, float w1, float b2,
                                           float w2, float eps, float step_size, float sqrt_bc2) {
    const float gi = rnd<T>(g_raw * coef);
    pi = rnd<T>(pi * decay);
    mi = rnd<T>(mi + w1 * (gi - mi));                             // lerp_, weight < 0.5 form
    vi = rnd<T>(vi * b2);
    vi = rnd<T>(vi + w2 * gi * gi);                               // addcmul_
    float den = rnd<T>(sqrtf(vi));
    den = rnd<T>(den / sqrt_bc2);
    den = rnd<T>(den + eps);
    pi = rnd<T>(pi - step_size * (mi / den));                     // addcdiv_
}

// ZG: the gradient is ZEROED as it is consumed (optimizer.zero_grad() folded into the update: 2 B per parameter written here instead of
// a separate 13.5 GB fill pass per optimizer step at 7B)
template <typename T, bool ZG>
__global__ __launch_bounds__(256) void adamw_kernel(T* __restrict__ p, T* __restrict__ g, T* __restrict__ m,
                                                    T* __restrict__ v, long n, float decay, float w1, float b2, float w2,
                                                    float eps, float step_size, float sqrt_bc2,
                                                    const float* __restrict__ clip) {
    // scalars arrive already rounded the way torch rounds its python-double hyper-parameters to fp32
    const float coef = clip ? clip[1] : 1.f;
    long done = 0;
    if constexpr (sizeof(T) == 2) {
        // 8 elements = one 16-byte access per array and thread (round 3: the element-wise form issued 7 two-byte accesses per element)
        if (((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0) {
            const long n8 = n / 8;
            for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
                float pf[8], gf[8], mf[8], vf[8];
                ld8((const bf16_t*)p + i * 8, pf);
                ld8((const bf16_t*)g + i * 8, gf);
                ld8((const bf16_t*)m + i * 8, mf);
                ld8((const bf16_t*)v + i * 8, vf);
#pragma unroll
                for (int j = 0; j < 8; ++j) adamw_elem<T>(pf[j], gf[j], mf[j], vf[j], coef, decay, w1, b2, w2, eps, step_size, sqrt_bc2);
                *(u32x4*)((bf16_t*)p + i * 8) = u32x4{pack2bf(pf[0], pf[1]), pack2bf(pf[2], pf[3]), pack2bf(pf[4], pf[5]), pack2bf(pf[6], pf[7])};
                *(u32x4*)((bf16_t*)m + i * 8) = u32x4{pack2bf(mf[0], mf[1]), pack2bf(mf[2], mf[3]), pack2bf(mf[4], mf[5]), pack2bf(mf[6], mf[7])};
                *(u32x4*)((bf16_t*)v + i * 8) = u32x4{pack2bf(vf[0], vf[1]), pack2bf(vf[2], vf[3]), pack2bf(vf[4], vf[5]), pack2bf(vf[6], vf[7])};
                if (ZG) *(u32x4*)((bf16_t*)g + i * 8) = u32x4{0u, 0u, 0u, 0u};
            }
            done = n8 * 8;
        }
    }
    for (long i = done + blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        float pi = ldf<T>(p, i), mi = ldf<T>(m, i), vi = ldf<T>(v, i);
        adamw_elem<T>(pi, ldf<T>(g, i), mi, vi, coef, decay, w1, b2, w2, eps, step_size, sqrt_bc2);
        stf<T>(p, i, pi);
        stf<T>(m, i, mi);
        stf<T>(v, i, vi);
        if (ZG) stf<T>(g, i, 0.f);
    }
}

inline int grid_for(long total, int cap = 256 * 8) {
    long b = (total + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

}  // namespace

extern "C" {

int nv_head_fwd_bf16(const void* x, const void* W, const void* bias, void* y, int B, int d, int N, void* stream) {
    if (!x || !W || !bias || !y || (d & 7)) return NV_ERR_ARG;
    if (B == 0) return NV_OK;
    NV_LAUNCH(head_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)W,
                       (const bf16_t*)bias, (bf16_t*)y, d, N);
    return nv_check_launch();
}
int nv_head_bwd_bf16(const void* dy, const void* x, const void* W, void* dx, void* gW, void* gb, int B, int d, int N,
                     void* stream) {
    if (!dy || !x || !W || !dx || !gW || !gb) return NV_ERR_ARG;
    if (B == 0) return NV_OK;
    hipStream_t st = (hipStream_t)stream;
    NV_LAUNCH(head_bwd_dx_kernel, dim3((d + 255) / 256, B), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)W,
                       (bf16_t*)dx, d, N);
    NV_LAUNCH(head_bwd_dw_kernel, dim3((d + 255) / 256, N), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x,
                       (bf16_t*)gW, (bf16_t*)gb, B, d, N);
    return nv_check_launch();
}
int nv_action_ce_bf16(const void* logits, const long* targets, float* loss_rows, void* dlogits, int B, int G, float gscale,
                      const float* gscale_dev, void* stream) {
    if (!logits || !targets || !loss_rows) return NV_ERR_ARG;
    if (B == 0) return NV_OK;
    NV_LAUNCH(action_ce_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)logits, targets, loss_rows,
                       (bf16_t*)dlogits, G, gscale, gscale_dev);
    return nv_check_launch();
}
int nv_lm_ce_bf16(void* logits, const int* labels, float* loss_rows, int M, int V, int ldl, int special0, int nspecial,
                  float gscale, int write_grad, void* stream) {
    if (!logits || !labels || !loss_rows) return NV_ERR_ARG;
    if (M == 0) return NV_OK;
    NV_LAUNCH(lm_ce_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (bf16_t*)logits, labels, loss_rows, V, ldl,
                       special0, nspecial, gscale, write_grad);
    return nv_check_launch();
}

// partial: >= 2048 floats per call site; is_bf16 selects the element type
int nv_sumsq(const void* g, long n, int is_bf16, float* partial, int* n_partial, void* stream) {
    if (!g || !partial || !n_partial) return NV_ERR_ARG;
    const int blocks = grid_for(is_bf16 ? (n + 7) / 8 : n);
    *n_partial = blocks;
    if (is_bf16)
        NV_LAUNCH(sumsq_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, n, partial);
    else
        NV_LAUNCH(sumsq_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)g, n, partial);
    return nv_check_launch();
}
int nv_clip_coef(const float* partial, int n_partial, float max_norm, float* out2, void* stream) {
    if (!partial || !out2) return NV_ERR_ARG;
    NV_LAUNCH(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n_partial, max_norm, out2);
    return nv_check_launch();
}
static int adamw_launch(void* p, void* g, void* m, void* v, long n, int is_bf16, double lr, double beta1, double beta2, double eps,
                        double wd, int step, const float* clip, int zero_grad, void* stream);
int nv_adamw(void* p, const void* g, void* m, void* v, long n, int is_bf16, double lr, double beta1, double beta2, double eps,
             double wd, int step, const float* clip, void* stream) {
    return adamw_launch(p, (void*)g, m, v, n, is_bf16, lr, beta1, beta2, eps, wd, step, clip, 0, stream);
}
// the same update with `optimizer.zero_grad()` folded in (train.py:88-89: step, then zero_grad): g is zeroed as it is read
int nv_adamw_zero_grad(void* p, void* g, void* m, void* v, long n, int is_bf16, double lr, double beta1, double beta2, double eps,
                       double wd, int step, const float* clip, void* stream) {
    return adamw_launch(p, g, m, v, n, is_bf16, lr, beta1, beta2, eps, wd, step, clip, 1, stream);
}
static int adamw_launch(void* p, void* g, void* m, void* v, long n, int is_bf16, double lr, double beta1, double beta2, double eps,
                        double wd, int step, const float* clip, int zero_grad, void* stream) {
    if (!p || !g || !m || !v || step < 1) return NV_ERR_ARG;
    if (n == 0) return NV_OK;
    // hyper-parameter arithmetic in double, like the python floats torch.optim.AdamW works with
    const float decay = (float)(1.0 - lr * wd), w1 = (float)(1.0 - beta1), b2 = (float)beta2, w2 = (float)(1.0 - beta2);
    const float step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
    const float sbc2 = (float)sqrt(1.0 - pow(beta2, (double)step));
    const float epsf = (float)eps;
    const int blocks = grid_for(n, 256 * 16);
    if (is_bf16 && zero_grad)
        NV_LAUNCH((adamw_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (bf16_t*)p, (bf16_t*)g,
                  (bf16_t*)m, (bf16_t*)v, n, decay, w1, b2, w2, epsf, step_size, sbc2, clip);
    else if (is_bf16)
        NV_LAUNCH((adamw_kernel<bf16_t, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (bf16_t*)p, (bf16_t*)g,
                  (bf16_t*)m, (bf16_t*)v, n, decay, w1, b2, w2, epsf, step_size, sbc2, clip);
    else if (zero_grad)
        NV_LAUNCH((adamw_kernel<float, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)p, (float*)g,
                  (float*)m, (float*)v, n, decay, w1, b2, w2, epsf, step_size, sbc2, clip);
    else
        NV_LAUNCH((adamw_kernel<float, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)p, (float*)g,
                  (float*)m, (float*)v, n, decay, w1, b2, w2, epsf, step_size, sbc2, clip);
    return nv_check_launch();
}

}  // extern "C"
