"""Weight-only fp8 (OCP e4m3fn) deployment of the LM (BASELINE config 5: Vicuna-13B inference).

`NavModel.to_fp8_weight_only()` quantises the decoder's seven Linear weights per layer -- as the four packed GEMM operands
q|k|v, o, gate|up, down -- to one byte per weight plus one fp32 scale per output channel (12.7 of 13.0 B parameters at 13B: 25.4 GB
of bf16 weights become 12.7 GB), and releases the bf16 copies and every gradient buffer.  Inference only: a forward with autograd
enabled raises.

Arithmetic: every GEMM multiplies with the de-quantised weight bf16(s[n] * q[n,k]) -- "the reference run on de-quantised
weights", the semantics tests/golden/g11_fp8_*.npz pin:
  * prefill / K/V-reuse steps (M > 16 rows): `nv_fp8_dequant_rows` writes the operand into a bf16 panel in front of the bf16 MFMA
    GEMM: an HBM-bound pre-pass of 3 B per weight, 5-12 % of the GEMM it feeds at B = 4..8.  Round 3 built and measured the obvious overlap -- the pre-pass of the NEXT Linear on a side stream into the
    other of two panels while the current GEMM runs (`NAVILLM_FP8_OVERLAP=1`; the native K/V-cache layer loop does the same,
    nv_decoder_set_fp8_overlap) -- and it LOSES 1-4 % at 13B / B = 8 (profiles/r03_fp8_overlap_ab.txt: the GEMM's LDS-DMA stream and the
    pre-pass share the same CUs and the same HBM queues; the GEMM slows down by more than the pre-pass costs in line).  It stays as an
    opt-in knob, bit-identical and tested; the default is the in-line pre-pass;
  * decode steps (M <= 16): `nv_gemv_fp8w` streams the codes themselves -- half the bytes per generated token;
  * round 4: GEMMs whose launch plan is a 128 / 160-row tile (K/V-reuse steps with ~100 new rows per sample; at 13B also part of the
    B = 4..8 prefill) run on the CODES (`nv_gemm_fp8w`): the weight tile is DMA'd as bytes and converted on the MFMA fragment path -- by
    default to bf16(s*q), bit-identical to the pre-pass (mode 7); opt-in mode 9 converts unscaled and applies s[n] to the fp32
    accumulator (faster than the bf16 GEMM itself, one rounding per weight off the pinned semantics); shapes the kernel declines keep
    the pre-pass.
"""
import os

import torch

from . import lib as _lib
from . import ops

BF16, F32, U8 = torch.bfloat16, torch.float32, torch.uint8
KINDS = ("qkv", "o", "gate_up", "down")


def quantize_rows(W):
    """bf16 [N,K] (device) -> (codes u8 [N,K], scales fp32 [N])"""
    ops._chk2d(W, BF16)
    N, K = W.shape
    q = torch.empty((N, K), dtype=U8, device=W.device)
    s = torch.empty((N,), dtype=F32, device=W.device)
    _lib.check(ops._L().nv_fp8_quant_rows(W.data_ptr(), q.data_ptr(), s.data_ptr(), N, K, W.stride(0), q.stride(0), ops._st()),
               "nv_fp8_quant_rows")
    return q, s


def dequantize_rows(q, s, out=None):
    N, K = q.shape
    if out is None:
        out = torch.empty((N, K), dtype=BF16, device=q.device)
    _lib.check(ops._L().nv_fp8_dequant_rows(q.data_ptr(), s.data_ptr(), out.data_ptr(), N, K, q.stride(0), out.stride(0), ops._st()),
               "nv_fp8_dequant_rows")
    return out


def decode_table(device):
    t = torch.empty((256,), dtype=BF16, device=device)
    _lib.check(ops._L().nv_fp8_decode_table(t.data_ptr(), ops._st()), "nv_fp8_decode_table")
    return t


def gemv_fp8w(x, q, s, out=None, R=None, epilogue=ops.EPI_STORE):
    M, K = x.shape
    N = q.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=x.device)
    rc = ops._L().nv_gemv_fp8w(x.data_ptr(), q.data_ptr(), s.data_ptr(), out.data_ptr(), ops._p(R), M, N, K, x.stride(0), q.stride(0),
                               out.stride(0), 0 if R is None else R.stride(0), epilogue, ops._st())
    _lib.check(rc, "nv_gemv_fp8w")
    return out


def gemm_fp8w(x, q, s, out=None, R=None, epilogue=ops.EPI_STORE, mode=0, tile_cfg=0):
    """the tile GEMM on the codes (nv_gemm_fp8w) -> out, or None when the shape is outside that kernel's range (the caller then runs
    the de-quantisation pre-pass + the bf16 GEMM)"""
    M, K = x.shape
    N = q.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=x.device)
    rc = ops._L().nv_gemm_fp8w(x.data_ptr(), q.data_ptr(), s.data_ptr(), out.data_ptr(), ops._p(R), M, N, K, x.stride(0), q.stride(0),
                               out.stride(0), 0 if R is None else R.stride(0), epilogue, mode, tile_cfg,
                               ops._gemm_ws(x.device) if ops.SPLITK_TAIL else 0, ops._st())
    if rc == -2:
        return None
    _lib.check(rc, "nv_gemm_fp8w")
    return out


FP8_TILE_GEMM = os.environ.get("NAVILLM_FP8_TILE_GEMM", "1") != "0"     # few-hundred-row GEMMs on the codes themselves (round 4)


class Fp8DecoderWeights:
    def __init__(self, model, resident_bf16=False, gemm_mode=None):
        """gemm_mode: how nv_gemm_fp8w turns codes into MFMA operands -- 7 (default): bf16(s * q), bit-identical to the pre-pass + bf16
        GEMM (three VALU instructions per pair of weights), i.e. exactly the de-quantised-weights semantics fixture G11 pins; 9: the
        codes are converted unscaled (exact: e4m3 fits bf16, one instruction per pair) and s[n] multiplies the fp32 accumulator -- the
        weight is then s * q with one bf16 rounding per weight FEWER than the pinned semantics: results within one output spacing of
        mode 7 per GEMM (tests/test_fp8_gpu.py), the G11 logits 2.0-3.0 spacings from the fixture (mode 7: 2.0-2.5), and ~20 % faster
        than the bf16 GEMM on the same tile (profiles/r04_gemm_fp8_probe.txt) -- an explicit opt-in.  NAVILLM_FP8_GEMM_MODE overrides."""
        cfg, st = model.cfg, model.store
        if gemm_mode is None:
            gemm_mode = int(os.environ.get("NAVILLM_FP8_GEMM_MODE", "7"))
        assert gemm_mode in (7, 9), gemm_mode
        self.gemm_mode = gemm_mode
        # (per object: `linear()` passes the mode per call, KVCacheLM hands it to its native layer loop with nv_decoder_set_fp8_gemm_mode;
        # the process-wide default is left alone -- two models with different modes must not override each other)
        self.codes, self.scales = [], []
        self.resident = [] if resident_bf16 else None          # per layer: the bf16 operands holding bf16(s*q) (views of the flat store)
        with torch.no_grad():
            for i in range(cfg.num_layers):
                p = f"lang_model.model.layers.{i}."
                mats = {"qkv": st.qkv(i), "o": st.p(p + "self_attn.o_proj.weight"), "gate_up": st.gate_up(i),
                        "down": st.p(p + "mlp.down_proj.weight")}
                qs = {k: quantize_rows(v) for k, v in mats.items()}
                self.codes.append({k: v[0] for k, v in qs.items()})
                self.scales.append({k: v[1] for k, v in qs.items()})
                if resident_bf16:
                    for k, W in mats.items():
                        dequantize_rows(qs[k][0], qs[k][1], out=W)
                    self.resident.append(mats)
        n_max = max(q.shape[0] * q.shape[1] for q in self.codes[0].values())
        self._scratch = None if resident_bf16 else torch.empty((n_max,), dtype=BF16, device=model.device)
        # the overlapped pre-pass: two panels, a side stream, "operand ready" / "panel free" events; built on first use
        self.overlap = (not resident_bf16) and os.environ.get("NAVILLM_FP8_OVERLAP", "0") == "1"
        self._n_max, self._device, self._L = n_max, model.device, cfg.num_layers
        self._panels = None
        self.bytes = sum(q.numel() + s.numel() * 4 for c, sc in zip(self.codes, self.scales) for q, s in zip(c.values(), sc.values()))

    def weight(self, i, kind):
        """the de-quantised operand bf16(s*q) in the shared scratch panel (valid until the next call, stream-ordered)"""
        if self.resident is not None:
            return self.resident[i][kind]
        q, s = self.codes[i][kind], self.scales[i][kind]
        return dequantize_rows(q, s, out=self._scratch[:q.numel()].view(q.shape))

    def pipe(self):
        """(panel a, panel b, side stream) of the overlapped pre-pass, shared with the native layer loop (navillm_amd/kvcache.py)"""
        if self._panels is None:
            self._panels = [torch.empty((self._n_max,), dtype=BF16, device=self._device) for _ in range(2)]
            self._side = torch.cuda.Stream(device=self._device)
            self._ev_dq = [torch.cuda.Event() for _ in range(2)]
            self._ev_use = [torch.cuda.Event() for _ in range(2)]
            self._ready = [None, None]
        return self._panels[0], self._panels[1], self._side

    def invalidate(self):
        """someone else (the native layer loop) wrote the panels"""
        if self._panels is not None:
            self._ready = [None, None]

    def _dequant_on_side(self, key, pn):
        q, s = self.codes[key[0]][key[1]], self.scales[key[0]][key[1]]
        with torch.cuda.stream(self._side):
            dequantize_rows(q, s, out=self._panels[pn][:q.numel()].view(q.shape))
            self._ev_dq[pn].record(self._side)
        self._ready[pn] = key

    def _next(self, i, kind):
        k = KINDS.index(kind) + 1
        return (i, KINDS[k]) if k < 4 else ((i + 1) % self._L, KINDS[0])

    def linear(self, x, i, kind, out=None, R=None, epilogue=ops.EPI_STORE):
        q, s = self.codes[i][kind], self.scales[i][kind]
        if x.shape[0] <= 16 and epilogue in (ops.EPI_STORE, ops.EPI_RESID) and q.shape[1] % 64 == 0:
            return gemv_fp8w(x, q, s, out=out, R=R, epilogue=epilogue)
        if FP8_TILE_GEMM and self.resident is None and epilogue in (ops.EPI_STORE, ops.EPI_RESID):
            y = gemm_fp8w(x, q, s, out=out, R=R, epilogue=epilogue, mode=self.gemm_mode)   # None: not its shape -> pre-pass + bf16 GEMM
            if y is not None:
                return y
        if not self.overlap or self.resident is not None:
            return ops.gemm_bf16(ops.NT, x, self.weight(i, kind), out=out, R=R, epilogue=epilogue)
        self.pipe()
        main = torch.cuda.current_stream(self._device)
        key = (i, kind)
        if key in self._ready:
            pn = self._ready.index(key)
        else:
            # not prefetched (first Linear, or the Linears are not visited in layer order): after everything enqueued so far
            pn = 0 if self._ready[0] is None else 1 if self._ready[1] is None else getattr(self, "_last_pn", 1) ^ 1
            self._side.wait_stream(main)
            self._dequant_on_side(key, pn)
        main.wait_event(self._ev_dq[pn])
        y = ops.gemm_bf16(ops.NT, x, self._panels[pn][:q.numel()].view(q.shape), out=out, R=R, epilogue=epilogue)
        self._ev_use[pn].record(main)
        self._last_pn = pn
        nxt = self._next(i, kind)
        if self._ready[pn ^ 1] != nxt:
            self._side.wait_event(self._ev_use[pn ^ 1])      # the GEMM that last read the other panel (never recorded: no wait)
            self._dequant_on_side(nxt, pn ^ 1)
        return y
