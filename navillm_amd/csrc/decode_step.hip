// Device-side bookkeeping of greedy decoding (reference: HF generate(do_sample=False) as called from
// models/nav_model.py:324-341,388-402 with the special-id logit mask of models/modified_lm.py:122-124).
//
// One decode step is ~360 dependent kernel launches; with the token choice on the host every step also pays a device->host
// sync, the python index bookkeeping and an upload.  Here the choice and the bookkeeping are two tiny kernels, so a whole step
// (lm_head -> pick -> advance -> embed -> 32 decoder layers) has frozen launch arguments and is replayed from a hipGraph
// (navillm_amd/kvcache.py); the host only polls the `fin` flags, a step or two behind.
//
// state layout (int32, one buffer): tok[B] | fin[B] | len[B] | pos[B] | crow[B] | grow[B] | last[B] | dyn[2] | cnt[1] | overflow[1]
#include "nv_common.h"

namespace {

// grid B, 1024 threads, 8 logits per load: argmax over the valid vocabulary (ids >= V and the special range are excluded; ties ->
// smallest id, like torch.argmax on the CPU), then HF's bookkeeping: finished rows emit `pad`; a row finishes when it emits `eos`.
constexpr int PICK_T = 1024;
__global__ __launch_bounds__(PICK_T) void decode_pick_kernel(const bf16_t* __restrict__ logits, int ldl, int V, int special0, int nspecial,
                                                             int* __restrict__ state, int* __restrict__ out, int max_steps, int B, int eos,
                                                             int pad) {
    __shared__ float sv[PICK_T / 64];
    __shared__ int si[PICK_T / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const bf16_t* row = logits + (long)b * ldl;
    float best = -INFINITY;
    int bi = 0x7fffffff;                                          // "none yet"
    auto take = [&](float x, int v) {
        if (v < V && !(v >= special0 && v < special0 + nspecial) && (bi == 0x7fffffff || x > best || (x == best && v < bi))) { best = x; bi = v; }
    };
    const bool vec = ((ldl & 7) == 0) && ((((uintptr_t)logits) & 15) == 0);
    if (vec) {
        for (int v0 = tid * 8; v0 < V; v0 += PICK_T * 8) {
            const u32x4 q = *(const u32x4*)(row + v0);                 // ldl >= V rounded up by the caller's padding (vocab_pad % 8 == 0)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                take(__uint_as_float(q[j] << 16), v0 + 2 * j);
                take(__uint_as_float(q[j] & 0xffff0000u), v0 + 2 * j + 1);
            }
        }
    } else {
        for (int v = tid; v < V; v += PICK_T) take(bf2f(row[v]), v);
    }
    // wave reduction (value desc, index asc), then across the 16 waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float x = __shfl_xor(best, o, 64);
        const int j = __shfl_xor(bi, o, 64);
        if (j != 0x7fffffff && (bi == 0x7fffffff || x > best || (x == best && j < bi))) { best = x; bi = j; }
    }
    if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < PICK_T / 64; ++w) {
            const float x = sv[w];
            const int j = si[w];
            if (j != 0x7fffffff && (bi == 0x7fffffff || x > best || (x == best && j < bi))) { best = x; bi = j; }
        }
        int* tok = state;
        int* fin = state + B;
        const int cnt = state[7 * B + 2];
        const int nxt = fin[b] ? pad : bi;
        if (!fin[b] && nxt == eos) fin[b] = 1;
        tok[b] = nxt;
        if (cnt < max_steps) out[(long)cnt * B + b] = nxt;
    }
}

// one block: the token picked for sample b goes to cache row b*cap + len[b] at position len[b]; dyn = {max len + 1, 128-aligned
// first query row}; len += 1; cnt += 1.  A full cache (len == cap) sends the row to the junk row B*cap, stops growing, marks the sample
// FINISHED (it emits `pad` from the next step on: its query would be read back from the shared junk row, i.e. be garbage) and
// raises the sticky `overflow` word (unless the sample had already finished), which the host must check -- a full cache is an error
// of the caller, not a silent wrong answer.
__global__ __launch_bounds__(64) void decode_advance_kernel(int* __restrict__ state, int B, int cap) {
    __shared__ int smax, smin;
    const int tid = threadIdx.x;
    if (tid == 0) { smax = 0; smin = 0x7fffffff; }
    __syncthreads();
    int* len = state + 2 * B;
    int* pos = state + 3 * B;
    int* crow = state + 4 * B;
    int* grow = state + 5 * B;
    int* last = state + 6 * B;
    for (int b = tid; b < B; b += 64) {
        const int L = len[b];
        const bool full = L >= cap;
        const int Lc = full ? cap - 1 : L;
        pos[b] = Lc;
        crow[b] = full ? B * cap : b * cap + L;
        grow[b] = b * cap + Lc;
        last[b] = b;
        if (!full) len[b] = L + 1;
        else {
            // a row that had ALREADY finished (eos) only emits `pad` from here on: its cache filling up while other samples still
            // decode is not an error -- every token it emitted is valid
            if (!state[B + b]) state[7 * B + 3] = 1;
            state[B + b] = 1;
        }
        atomicMax(&smax, Lc);
        atomicMin(&smin, Lc);
    }
    __syncthreads();
    if (tid == 0) {
        state[7 * B] = smax + 1;
        state[7 * B + 1] = (smin / 128) * 128;
        state[7 * B + 2] += 1;
    }
}

}  // namespace

extern "C" {

int nv_decode_state_ints(int B) { return B > 0 ? 7 * B + 4 : 0; }

int nv_decode_pick_bf16(const void* logits, int ldl, int V, int special0, int nspecial, int* state, int* out, int max_steps, int B, int eos,
                        int pad, void* stream) {
    if (!logits || !state || !out || B <= 0 || V <= 0 || ldl < V || max_steps < 0 || nspecial < 0) return NV_ERR_ARG;
    NV_LAUNCH(decode_pick_kernel, dim3(B), dim3(PICK_T), 0, (hipStream_t)stream, (const bf16_t*)logits, ldl, V, special0, nspecial, state, out,
              max_steps, B, eos, pad);
    return nv_check_launch();
}

int nv_decode_advance(int* state, int B, int cap, void* stream) {
    if (!state || B <= 0 || cap <= 0) return NV_ERR_ARG;
    NV_LAUNCH(decode_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, B, cap);
    return nv_check_launch();
}

}  // extern "C"
