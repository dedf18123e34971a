"""Data parallelism over the 8 GPUs of one MI355X node: one process per GPU, gradients exchanged over RCCL/xGMI through
the C-ABI communicator `nv_comm_*` (include/navillm_hip.h).  Replaces tools/distributed.py:105-183 (process group from the
torchrun env) and the DDP wrapper of tools/optims.py:52-54.

Design for xGMI (7 point-to-point links per GPU, no switch): few, large messages, moved ONCE per optimizer step.
  * The flat gradient buffers (navillm_amd/flat.py) make every decoder layer one contiguous ~400 MB bf16 slice: no bucket
    copies, no `find_unused_parameters` bitmap (unused parameters simply carry zero gradients in the flat buffer; which is what
    DDP's flag achieves for `og_head`, nav_model.py:78-80).
  * `reduce="step"` (default): per-step `backward()`s only accumulate locally -- also the up-to-three synced backwards of an
    episode's last step (mp3d_agent.py:756,824,902), which under DDP are three full 13.6 GB all-reduces.  The mean over ranks
    of the ACCUMULATED gradient is taken once: either overlapped with the last backward before the optimizer step
    (`with ddp.final_backward(): loss.backward()` -- layer i's slice is exchanged on a side stream as soon as that backward
    has accumulated layer i's weight gradients, while layers i-1..0 are still computing), or, when no backward was flagged,
    by `flush()`, which `FlatAdamW.clip_grad_norm_/step` call first.  Sum of means == mean of sums.
  * `reduce="backward"`: DDP's semantics literally -- every backward outside `no_sync()` exchanges, overlapped.
  * Per slice: in-place reduce-scatter + all-gather (`algo="rs_ag"`) or one all-reduce (`algo="allreduce"`); `calibrate()`
    times both on a layer slice and keeps the faster (identical result on every rank: decided on the max over ranks).
Transports: the C-ABI communicator `RcclComm` (default on GPUs) or torch.distributed (`NAVILLM_COMM=torch`; the only one for
the CPU/gloo tests).  `no_sync()` and `.module` follow the DDP protocol the agent relies on (mp3d_agent.py:661-676).
"""
import contextlib
import ctypes
import os
import time

import torch
import torch.distributed as dist


def world_info_from_env():
    """tools/distributed.py:38-60 (torchrun variables only)."""
    return int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed_device(backend=None, device_index=None):
    """-> (device, rank, world_size). env:// rendezvous, one GPU per process.  The torch.distributed group is the CONTROL
    plane (rendezvous, barriers, the bench's max-over-ranks clock); gradients travel through `RcclComm`.
    `device_index` overrides LOCAL_RANK (rehearsing the N-rank control flow on a box with fewer GPUs: every rank on GPU 0, gloo)."""
    local_rank, rank, world = world_info_from_env()
    # the host driver of the MI355X nodes only supports dmabuf IPC: without this RCCL's intra-node setup (and CUDA-tensor sharing
    # across processes) fails with `hipIpcGetMemHandle: invalid argument`.  Must be in the environment before the first HIP call
    # of the process to be safe; a launcher-level export is the robust form.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.is_available():
        local_rank = local_rank if device_index is None else device_index
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        backend = backend or "nccl"
    else:
        device = torch.device("cpu")
        backend = backend or "gloo"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo" and os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost", "::1"):
            # one node: gloo's TCP transport otherwise picks its interface by resolving the machine's hostname, which a container's
            # may not do; the loopback interface always exists
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank, **kw)
    return device, rank, world


class RcclComm:
    """The C-ABI communicator (`nv_comm_*`, include/navillm_hip.h): rank 0 draws the RCCL unique id, the other ranks receive
    it through `exchange` (any `bytes -> bytes` broadcast); collectives run on the CURRENT torch stream, in place.

    Default exchange: the already initialised torch.distributed group when there is one (control plane only), else a
    `torch.distributed.TCPStore` on MASTER_PORT+1 that rank 0 keeps alive until every rank has read the id."""

    def __init__(self, rank, world, exchange=None):
        from . import lib as _lib
        self._lib, self._L = _lib, _lib.load()
        self.rank, self.world = rank, world
        self._store = None
        n = self._L.nv_comm_unique_id_bytes()
        uid = ctypes.create_string_buffer(n)
        if rank == 0:
            _lib.check(self._L.nv_comm_unique_id(uid), "nv_comm_unique_id")
        if world > 1:
            exchange = exchange or self._default_exchange
            raw = exchange(uid.raw if rank == 0 else None)
            uid = ctypes.create_string_buffer(bytes(raw), n)
        self._ctx = ctypes.c_void_p()
        _lib.check(self._L.nv_comm_init(ctypes.byref(self._ctx), uid, rank, world), "nv_comm_init")
        if self._store is not None:
            # rank 0 owns the store's server: it may only go away once every rank is past its get()
            self._store.add("nv_comm_ready", 1)
            if rank == 0:
                deadline = time.time() + 300
                while int(self._store.add("nv_comm_ready", 0)) < world and time.time() < deadline:
                    time.sleep(0.01)
            self._store = None

    def _default_exchange(self, payload):
        n = self._L.nv_comm_unique_id_bytes()
        if dist.is_initialized() and dist.get_world_size() == self.world:
            on_gpu = dist.get_backend() == "nccl"
            t = torch.zeros(n, dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
            if self.rank == 0:
                t.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
            dist.broadcast(t, src=0)
            return bytes(t.cpu().numpy().tobytes())
        self._store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + 1,
                                    self.world, self.rank == 0)
        if self.rank == 0:
            self._store.set("nv_comm_uid", payload)
            return payload
        return self._store.get("nv_comm_uid")

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def allreduce_mean_(self, t):
        fn = {torch.bfloat16: self._L.nv_comm_allreduce_bf16, torch.float32: self._L.nv_comm_allreduce_f32}[t.dtype]
        self._lib.check(fn(self._ctx, t.data_ptr(), t.numel(), 1, self._stream()), "nv_comm_allreduce")
        return t

    def reduce_scatter_all_gather_mean_(self, t):
        """mean over ranks in two in-place phases; t.numel() % world == 0"""
        assert t.dtype in (torch.bfloat16, torch.float32) and t.numel() % self.world == 0
        s = self._stream()
        self._lib.check(self._L.nv_comm_reduce_scatter(self._ctx, t.data_ptr(), t.numel(), 1 if t.dtype == torch.bfloat16 else 0, 1, s),
                        "nv_comm_reduce_scatter")
        self._lib.check(self._L.nv_comm_all_gather(self._ctx, t.data_ptr(), t.numel() * t.element_size(), s), "nv_comm_all_gather")
        return t

    def broadcast_(self, t, root=0):
        self._lib.check(self._L.nv_comm_broadcast(self._ctx, t.data_ptr(), t.numel() * t.element_size(), root, self._stream()),
                        "nv_comm_broadcast")
        return t

    def close(self):
        if self._ctx:
            self._L.nv_comm_destroy(self._ctx)
            self._ctx = None


def _allreduce_mean_(t, group=None):
    """torch.distributed transport: in-place mean over ranks on the CURRENT stream (SUM collective, then a 1/world scale;
    for bf16 on the GPU the scale is our own HIP kernel)."""
    world = dist.get_world_size(group)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (NAVILLM_BENCH_REHEARSAL: N ranks sharing one GPU cannot form an RCCL communicator): stage through the host
        h = t.detach().to("cpu", torch.float32)
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_((h / world).to(t.dtype))
        return
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    if t.is_cuda and t.dtype == torch.bfloat16 and t.numel() % 8 == 0:
        from . import ops
        ops.scale_bf16_(t, 1.0 / world)
    else:
        t.mul_(1.0 / world)


class GradSlices:
    """The ordered list of contiguous gradient slices exchanged per reduction."""

    def __init__(self, store):
        self.store = store
        lm = store.grad["lm"]
        L = store.cfg.num_layers
        first_s, _ = store.layer_slice(0)
        _, last_e = store.layer_slice(L - 1)
        self.layer = [lm[s:e] for (s, e) in (store.layer_slice(i) for i in range(L))]
        self.rest = [lm[:first_s], lm[last_e:], store.grad["f32"]]   # embeddings | norm+lm_head+heads | fp32 side

    def all_slices(self):
        """every slice, in the order the overlapped exchange issues them (decoder layers LAST to FIRST, as the backward finishes them, then the
        rest): a rank that exchanges from inside its last backward and a rank that has nothing deferred and exchanges in the optimizer's
        flush must issue the SAME sequence of collectives, or equal-sized layer slices would be paired across different layers"""
        return self.layer[::-1] + self.rest


class NavDataParallel(torch.nn.parallel.DistributedDataParallel):
    """Subclasses DistributedDataParallel ONLY so that the reference's `isinstance(model, DistributedDataParallel)` checks
    (tasks/agents/mp3d_agent.py:661; tools/optims.py:66-67 reads `.module`) accept it unchanged; DDP's constructor, reducer and
    buckets are never built -- the state is a plain nn.Module's plus what is set below."""

    def __init__(self, module, group=None, overlap=True, comm=None, force_sync=False, reduce=None, algo=None):
        torch.nn.Module.__init__(self)
        self.module = module
        self.group = group
        on_gpu = module.store.device.type == "cuda"
        if comm is None and on_gpu and os.environ.get("NAVILLM_COMM", "rccl") != "torch":
            _, rank, world = world_info_from_env()
            if dist.is_initialized():
                rank, world = dist.get_rank(group), dist.get_world_size(group)
            if world > 1 or force_sync:
                comm = RcclComm(rank, world)
        self.force_sync = force_sync     # run the exchange even in a world of one (single-GPU test of the stream/event wiring)
        self.comm = comm                 # the C-ABI communicator; None: torch.distributed (gloo on CPU, NAVILLM_COMM=torch)
        self.overlap = overlap
        self.reduce = reduce or os.environ.get("NAVILLM_DP_REDUCE", "step")
        assert self.reduce in ("step", "backward")
        self.algo = algo or os.environ.get("NAVILLM_DP_ALGO", "rs_ag")
        assert self.algo in ("rs_ag", "allreduce")
        self.require_sync = True
        self._final = False              # inside final_backward(): this backward exchanges, overlapped
        self._pending = False            # reduce="step": local gradients not yet averaged over ranks
        self.slices = GradSlices(module.store)
        self._comm_stream = torch.cuda.Stream() if on_gpu else None
        self._queued = False
        self.calibration = None
        # measurement (bench.py `dp` block): HIP events around every slice's collective on the side stream and around the join of
        # the compute stream with it -> exchange time per optimizer step and the part of it the backward did NOT hide
        self.profile = False
        self._prof_slices, self._prof_joins = [], []
        # back-reference WITHOUT module registration: `module._dp = self` would make the wrapper a child module of the model
        # it wraps (a cycle: .train()/.eval()/.state_dict() then recurse forever)
        object.__setattr__(module, "_dp", self)
        self.broadcast_parameters()

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def named_parameters(self, *args, **kwargs):
        """`torch.nn.utils.clip_grad_norm_(model.parameters(), 40.)` on the WRAPPED model (train.py:87): nn.Module walks the children's
        `_parameters` dicts directly, so `NavModel.named_parameters` never runs -- without this the clip norm would be taken before an
        open accumulation window / automatic episode had handed its gradients to `.grad`, and `FlatAdamW.step()` would then add them
        AFTER the clip coefficient was fixed.  `reduce="step"`: what was accumulated locally is averaged over the
        ranks here as well, so the norm the caller computes is the norm of the gradient the optimizer applies."""
        m = self.module
        hand = getattr(m, "grad_handover", None)
        if hand is not None:
            hand("parameters")
        if getattr(m, "episode", None) is not None:
            m.episode.assert_no_pending_gradients("model.parameters() [e.g. torch.nn.utils.clip_grad_norm_(model.parameters(), ...)]")
        if getattr(self, "_pending", False):
            self.flush()
        return super().named_parameters(*args, **kwargs)

    # ---- transport
    def _world(self):
        if self.comm is not None:
            return self.comm.world
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _reduce(self, t, algo=None):
        if self.comm is not None:
            if (algo or self.algo) == "rs_ag" and t.numel() % self.comm.world == 0:
                self.comm.reduce_scatter_all_gather_mean_(t)
            else:
                self.comm.allreduce_mean_(t)
        else:
            _allreduce_mean_(t, self.group)

    @torch.no_grad()
    def broadcast_parameters(self):
        """DDP's initial rank-0 broadcast."""
        self.module.store.wait_params()
        if self.comm is not None:
            if self.comm.world > 1:
                for t in self.module.store.param.values():
                    self.comm.broadcast_(t, 0)
                probe = torch.zeros(1024, dtype=torch.float32, device=self.module.store.device)
                self._reduce(probe)          # channel setup for the collectives of the first exchange
            return
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            staged = self.module.store.device.type == "cuda" and dist.get_backend(self.group) == "gloo"      # rehearsal only
            for t in self.module.store.param.values():
                if staged:
                    h = t.detach().cpu()
                    dist.broadcast(h, src=0, group=self.group)
                    t.copy_(h)
                else:
                    dist.broadcast(t, src=0, group=self.group)
            if not staged:
                probe = torch.zeros(1024, dtype=torch.float32, device=self.module.store.device)
                dist.all_reduce(probe, group=self.group)

    @torch.no_grad()
    def calibrate(self, iters=5):
        """time all-reduce vs reduce-scatter+all-gather on one decoder-layer gradient slice and keep the faster for every
        slice.  The slice's content is irrelevant and is restored; all ranks take the same decision: each rank's figure is its
        MEDIAN over `iters` timed launches, the decision is made on the max over ranks (exchanged through the communicator).
        A collective that fails raises on the spot: a failure that only some ranks see must not be swallowed, or the ranks would
        go on to issue different sequences of collectives and hang."""
        if self.comm is None or self.comm.world < 2:
            return None
        t = self.slices.layer[0]
        keep = t.clone()
        res = {}
        for algo in ("allreduce", "rs_ag"):
            self._reduce(t, algo)                          # warm the channels of this collective
            torch.cuda.synchronize()
            times = []
            for _ in range(iters):
                t0 = time.perf_counter()
                self._reduce(t, algo)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            mine = sorted(times)[len(times) // 2]
            # max over ranks through the communicator itself: rank r writes slot r (x world), mean-all-reduce, max on the host
            slots = torch.zeros(self.comm.world, dtype=torch.float32, device=t.device)
            slots[self.comm.rank] = mine * self.comm.world
            self.comm.allreduce_mean_(slots)
            res[algo] = float(slots.max().item())
        t.copy_(keep)
        self.algo = "rs_ag" if res["rs_ag"] <= res["allreduce"] else "allreduce"
        nbytes = t.numel() * t.element_size()
        self.calibration = {"slice_bytes": nbytes, "seconds": dict(res), "chosen": self.algo, "iters": iters,
                            "algbw_GBps": {k: round(nbytes / v / 1e9, 1) for k, v in res.items() if v > 0.0}, "errors": {}}
        return self.calibration

    # ---- the DDP protocol
    @contextlib.contextmanager
    def no_sync(self):
        old, self.require_sync = self.require_sync, False
        try:
            yield
        finally:
            self.require_sync = old

    @contextlib.contextmanager
    def final_backward(self):
        """`reduce="step"`: the backward(s) run inside this context are the last before the optimizer step; the exchange is
        launched from inside them, overlapped with the remaining backward GEMMs."""
        old, self._final = self._final, True
        try:
            yield
        finally:
            self._final = old

    def _exchanging(self):
        """does the backward that is starting now exchange gradients?"""
        if not (self._world() > 1 or self.force_sync):
            return False
        if self.reduce == "backward":
            return self.require_sync
        return self._final

    # ---- hooks called from LlamaStack.backward
    def on_backward_begin(self):
        if self.reduce == "step" and not self._final and (self._world() > 1 or self.force_sync):
            self._pending = True            # accumulated locally; flush() or a final_backward() will average it
        if not self._exchanging() or self._queued:
            return
        self._queued = True
        torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def on_deferred_backward_begin(self):
        """a backward that runs OUTSIDE the autograd engine (navillm_amd/episode.py: finish_episode): same bookkeeping as
        on_backward_begin, but no engine callback can be queued -- the caller runs `_finalize()` itself when it is done"""
        if self.reduce == "step" and not self._final and (self._world() > 1 or self.force_sync):
            self._pending = True

    def on_layer_done(self, i, events=()):
        if not self._exchanging() or not self.overlap:
            return
        self._launch(self.slices.layer[i], events)

    def _launch(self, t, events=()):
        """exchange slice `t` on the side stream, ordered after everything enqueued so far on the compute stream and after
        `events` (the wgrad GEMMs of that layer when they run on their own stream); the host does not block."""
        if self._comm_stream is not None:
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            for e in events:
                self._comm_stream.wait_event(e)
            with torch.cuda.stream(self._comm_stream):
                if self.profile:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    self._reduce(t)
                    e1.record()
                    self._prof_slices.append((e0, e1, t.numel() * t.element_size()))
                else:
                    self._reduce(t)
        else:
            self._reduce(t)

    def _join(self):
        """the compute stream waits for the side stream's collectives (end of an exchange)"""
        if self._comm_stream is None:
            return
        if self.profile:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            torch.cuda.current_stream().wait_stream(self._comm_stream)
            b.record()
            self._prof_joins.append((a, b))
        else:
            torch.cuda.current_stream().wait_stream(self._comm_stream)

    def exchange_stats(self, reset=True):
        """(profile = True) per exchange = per optimizer step: time the collectives were busy on the side stream, the bytes they
        moved per rank, and the EXPOSED part -- how long the compute stream sat at the join waiting for them (what the overlap with
        the backward did not hide)."""
        if not self._prof_joins:
            return None
        torch.cuda.synchronize()
        n = len(self._prof_joins)
        busy = sum(a.elapsed_time(b) for a, b, _ in self._prof_slices)
        byts = sum(x for _, _, x in self._prof_slices)
        exposed = sum(a.elapsed_time(b) for a, b in self._prof_joins)
        out = {"exchanges": n, "slices_per_exchange": round(len(self._prof_slices) / n, 1), "bytes_per_exchange": int(byts / n),
               "collective_busy_ms_per_exchange": round(busy / n, 3), "exposed_ms_per_exchange": round(exposed / n, 3),
               "busbw_GBps": (round(byts / (busy * 1e-3) / 1e9 * 2 * (self._world() - 1) / max(self._world(), 1), 1) if busy > 0 else None)}
        if reset:
            self._prof_slices, self._prof_joins = [], []
        return out

    def _finalize(self):
        """end of the autograd pass: exchange what is left, then join the side stream."""
        self._queued = False
        todo = self.slices.rest if self.overlap else self.slices.all_slices()
        for t in todo:
            self._launch(t)
        self._join()
        self._pending = False

    @torch.no_grad()
    def flush(self):
        """`reduce="step"`: average whatever was accumulated locally since the last exchange (no-op when a final_backward()
        already did).  Called by FlatAdamW before the clip / the update."""
        if self._pending:
            self.sync_gradients()

    @torch.no_grad()
    def merge_touched(self):
        """Union over ranks of `FlatStore.touched` (which tensors have ever received a gradient = which tensors AdamW updates,
        navillm_amd/optim.py).  Touching depends on the LOCAL batch (`obj_projector` only when the batch carries objects,
        nav_model.py:111-119; `lm_head` only in the LM-loss modes), while the averaged gradient is the same on every rank: the
        reference's DDP(find_unused_parameters=True) writes the reduced gradient on every rank as soon as ANY rank used the
        parameter (tools/optims.py:52-54), so every rank must start that parameter's AdamW state in the same step or the
        replicas diverge for good.  One tiny MAX-reduction of a 0/1 vector over the parameter names per optimizer step."""
        if self._world() <= 1:
            return
        st = self.module.store
        names = st.names["lm"] + st.names["f32"]
        host = torch.tensor([1.0 if n in st.touched else 0.0 for n in names], dtype=torch.float32)
        if self.comm is not None:
            v = host.to(st.device)
            self.comm.allreduce_mean_(v)             # mean > 0  <=>  some rank touched it
            host = v.cpu()
        else:
            on_dev = st.device.type == "cuda" and dist.get_backend(self.group) != "gloo"
            v = host.to(st.device) if on_dev else host
            dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self.group)
            host = v.cpu()
        st.touched.update(n for n, f in zip(names, host.tolist()) if f > 0.0)

    @torch.no_grad()
    def sync_gradients(self):
        """Explicit one-shot reduction of every slice."""
        if self._world() > 1 or self.force_sync:
            for t in self.slices.all_slices():
                self._launch(t)
            self._join()
        self._pending = False


def dp_preflight(device, rank, world, watchdog_s=240.0, on_hang=None, try_nv_comm=True):
    """A few seconds of collectives before an N-rank job commits to its transport (round 6): RCCL through the
    C-ABI communicator has never run with more than one rank on this build's hardware pool, so the first multi-GPU launch must end in
    a measurement or in an ACTIONABLE error, never in a hang or a bare traceback.

    Requires an initialised torch.distributed group (any backend; the bench uses gloo, so the control plane does not depend on RCCL).
    Stages, each recorded with its outcome and time:
      control_plane      barrier + a tiny all-reduce over the existing group;
      nv_comm_init       `RcclComm(rank, world)`: dlopen'd librccl, unique-id exchange, ncclCommInitRank, the enum self-check all-reduce;
      nv_comm_collective mean all-reduce of a known bf16 vector + reduce-scatter / all-gather of 16 MB, values verified, timed;
      agreement          every rank reports; the transport is nv_comm only if ALL ranks passed;
      torch_nccl_group   (fallback, only if nv_comm failed somewhere) `dist.new_group(backend="nccl")` -- ProcessGroupNCCL is RCCL behind
                         torch's own binding -- with one verified all-reduce; the data path then runs through torch.distributed.
    A watchdog (`watchdog_s`) calls `on_hang(report)` and exits the process when a stage never returns (an RCCL bootstrap that hangs).
    -> (report dict, comm or None, group or None): transport = report["transport"] in {"nv_comm", "torch", None}."""
    import threading
    report = {"rank": rank, "world": world, "stages": [], "transport": None,
              "env": {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "NCCL_", "RCCL_", "HIP_VISIBLE", "ROCR_VISIBLE"))},
              "control_backend": dist.get_backend() if dist.is_initialized() else None}
    state = {"stage": "start", "done": False}

    def hang():
        if state["done"]:
            return
        report["hang_in_stage"] = state["stage"]
        try:
            if on_hang is not None:
                on_hang(report)
        finally:
            os._exit(3)
    timer = threading.Timer(watchdog_s, hang)
    timer.daemon = True
    timer.start()

    def stage(name, fn):
        state["stage"] = name
        t0 = time.perf_counter()
        rec = {"stage": name, "ok": False}
        try:
            out = fn()
            rec["ok"] = True
            if isinstance(out, dict):
                rec.update(out)
            return out
        except BaseException as e:                      # (a failing collective must be REPORTED, and by every rank)
            rec["error"] = f"{type(e).__name__}: {e}"[:600]
            return None
        finally:
            rec["seconds"] = round(time.perf_counter() - t0, 3)
            report["stages"].append(rec)

    def agree(flag):
        """min over ranks of a 0/1 flag through the control plane"""
        t = torch.tensor([1.0 if flag else 0.0])
        if dist.get_backend() == "nccl":
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def control():
        dist.barrier()
        return {"all_ranks_present": agree(True)}
    stage("control_plane", control)
    on_gpu = torch.device(device).type == "cuda"
    comm = None
    if try_nv_comm and on_gpu:
        def init():
            c = RcclComm(rank, world)
            v = int(c._L.nv_comm_rccl_version())
            return {"comm": c, "rccl_version": v}
        r = stage("nv_comm_init", init)
        if r is not None:
            comm = r["comm"]
            report["stages"][-1].pop("comm", None)

        def collective():
            x = torch.full((4096,), float(rank + 1), dtype=torch.bfloat16, device=device)
            comm.allreduce_mean_(x)
            want = sum(range(1, world + 1)) / world
            torch.cuda.synchronize()
            assert abs(float(x.float().mean()) - want) < 0.02 * want, (float(x.float().mean()), want)
            n = (16 << 20) // 2 // world * world
            y = torch.full((n,), float(rank + 1), dtype=torch.bfloat16, device=device)
            comm.reduce_scatter_all_gather_mean_(y)          # warm
            torch.cuda.synchronize()
            y.fill_(float(rank + 1))
            t0 = time.perf_counter()
            comm.reduce_scatter_all_gather_mean_(y)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert abs(float(y.float().mean()) - want) < 0.02 * want, (float(y.float().mean()), want)
            return {"rs_ag_16MB_ms": round(dt * 1e3, 3), "busbw_GBps": round(2 * (world - 1) / world * n * 2 / dt / 1e9, 1)}
        ok = comm is not None and stage("nv_comm_collective", collective) is not None
    else:
        ok = False
        report["stages"].append({"stage": "nv_comm_init", "ok": False, "skipped": "no GPU" if not on_gpu else "not requested"})
    all_ok = stage("agreement", lambda: {"nv_comm_on_every_rank": agree(ok)})
    group = None
    if all_ok is not None and all_ok["nv_comm_on_every_rank"]:
        report["transport"] = "nv_comm"
    else:
        if comm is not None:
            try:
                comm.close()
            except Exception:
                pass
            comm = None

        def fallback():
            g = dist.new_group(backend="nccl" if on_gpu else "gloo")
            x = torch.full((1024,), float(rank + 1), dtype=torch.float32, device=device if on_gpu else "cpu")
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=g)
            if on_gpu:
                torch.cuda.synchronize()
            assert abs(float(x[0]) - sum(range(1, world + 1))) < 1e-3, float(x[0])
            return {"group": g}
        r = stage("torch_nccl_group", fallback)
        if r is not None:
            report["stages"][-1].pop("group", None)
        f_ok = stage("agreement_fallback", lambda: {"torch_group_on_every_rank": agree(r is not None)})
        if f_ok is not None and f_ok["torch_group_on_every_rank"]:
            group = r["group"]
            report["transport"] = "torch"
    state["done"] = True
    timer.cancel()
    return report, comm, group


def broadcast_task_id(task_id, device, group=None, comm=None):
    """tasks/loaders.py:176-179: rank 0 picks the task, everyone follows (1 x int64)."""
    t = torch.tensor([task_id], dtype=torch.int64, device=device)
    if comm is not None and comm.world > 1:
        comm.broadcast_(t, 0)
    elif dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(t, src=0, group=group)
    return int(t.item())
