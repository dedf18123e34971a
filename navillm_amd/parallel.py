"""Data parallelism over the 8 GPUs of one MI355X node: one process per GPU, RCCL through
torch.distributed (backend "nccl" is RCCL on ROCm).  Replaces tools/distributed.py:105-183 (process
group from the torchrun env) and the DDP wrapper of tools/optims.py:52-54.

Design for xGMI (7 point-to-point links per GPU, no switch): few, large messages.  The flat
gradient buffers (navillm_amd/flat.py) make every decoder layer one contiguous ~400 MB bf16 slice:
as soon as layer i's backward has accumulated its weight gradients, its slice is all-reduced on a
side stream while layers i-1..0 are still computing -- 32 large collectives per synced backward, no
bucket copies, no `find_unused_parameters` bitmap (unused parameters simply carry zero gradients in
the flat buffer; which is what DDP's flag achieves for `og_head`, nav_model.py:78-80).

`no_sync()` and `.module` follow the DDP protocol the agent relies on (mp3d_agent.py:661-676).
"""
import contextlib
import os
import torch
import torch.distributed as dist


def world_info_from_env():
    """tools/distributed.py:38-60 (torchrun variables only)."""
    return int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed_device(backend=None):
    """-> (device, rank, world_size). env:// rendezvous, one GPU per process."""
    local_rank, rank, world = world_info_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        backend = backend or "nccl"
    else:
        device = torch.device("cpu")
        backend = backend or "gloo"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank, **kw)
    return device, rank, world


class RcclComm:
    """The C-ABI communicator (`nv_comm_*`, include/navillm_hip.h) for hosts that do not want torch.distributed in
    the data path: rank 0 draws the RCCL unique id, the other ranks receive it through `exchange` (any
    `bytes -> bytes` broadcast: a TCPStore, a file, MPI); collectives run on the CURRENT torch stream, in place.

    `NavDataParallel(model, comm=RcclComm(...))` or `NAVILLM_COMM=rccl` selects it; the default exchange uses a
    `torch.distributed.TCPStore` on MASTER_PORT+1 (control plane only)."""

    def __init__(self, rank, world, exchange=None):
        import ctypes
        from . import lib as _lib
        self._lib, self._L = _lib, _lib.load()
        self.rank, self.world = rank, world
        n = self._L.nv_comm_unique_id_bytes()
        uid = ctypes.create_string_buffer(n)
        if rank == 0:
            _lib.check(self._L.nv_comm_unique_id(uid), "nv_comm_unique_id")
        if world > 1:
            exchange = exchange or self._tcp_exchange
            raw = exchange(uid.raw if rank == 0 else None)
            uid = ctypes.create_string_buffer(raw, n)
        self._ctx = ctypes.c_void_p()
        _lib.check(self._L.nv_comm_init(ctypes.byref(self._ctx), uid, rank, world), "nv_comm_init")

    def _tcp_exchange(self, payload):
        store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + 1,
                              self.world, self.rank == 0)
        if self.rank == 0:
            store.set("nv_comm_uid", payload)
            return payload
        return store.get("nv_comm_uid")

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def allreduce_mean_(self, t):
        fn = {torch.bfloat16: self._L.nv_comm_allreduce_bf16, torch.float32: self._L.nv_comm_allreduce_f32}[t.dtype]
        self._lib.check(fn(self._ctx, t.data_ptr(), t.numel(), 1, self._stream()), "nv_comm_allreduce")
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
    """In-place mean over ranks, enqueued on the CURRENT stream (SUM collective, then a 1/world scale:
    the plainest RCCL call there is; for bf16 the scale is our own HIP kernel)."""
    world = dist.get_world_size(group)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    if t.is_cuda and t.dtype == torch.bfloat16 and t.numel() % 8 == 0:
        from . import ops
        ops.scale_bf16_(t, 1.0 / world)
    else:
        t.mul_(1.0 / world)


class GradSlices:
    """The ordered list of contiguous gradient slices exchanged per synced backward."""

    def __init__(self, store):
        self.store = store
        lm = store.grad["lm"]
        L = store.cfg.num_layers
        first_s, _ = store.layer_slice(0)
        _, last_e = store.layer_slice(L - 1)
        self.layer = [lm[s:e] for (s, e) in (store.layer_slice(i) for i in range(L))]
        self.rest = [lm[:first_s], lm[last_e:], store.grad["f32"]]   # embeddings | norm+lm_head+heads | fp32 side

    def all_slices(self):
        return self.layer + self.rest


class NavDataParallel(torch.nn.Module):
    def __init__(self, module, group=None, overlap=True, comm=None, force_sync=False):
        super().__init__()
        self.module = module
        self.group = group
        if comm is None and os.environ.get("NAVILLM_COMM") == "rccl" and module.store.device.type == "cuda":
            _, rank, world = world_info_from_env()
            comm = RcclComm(rank, world)
        self.force_sync = force_sync     # run the exchange even in a world of one (single-GPU test of the stream/event wiring)
        self.comm = comm                 # None: torch.distributed (RCCL through ProcessGroupNCCL); else the C-ABI communicator
        self.overlap = overlap
        self.require_sync = True
        self.slices = GradSlices(module.store)
        self._comm_stream = torch.cuda.Stream() if module.store.device.type == "cuda" else None
        self._queued = False
        module._dp = self
        self.broadcast_parameters()

    def forward(self, *a, **k):
        return self.module(*a, **k)

    @torch.no_grad()
    def broadcast_parameters(self):
        """DDP's initial rank-0 broadcast."""
        if self.comm is not None:
            if self.comm.world > 1:
                for t in self.module.store.param.values():
                    self.comm.broadcast_(t, 0)
            return
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            for t in self.module.store.param.values():
                dist.broadcast(t, src=0, group=self.group)
            # touch the all-reduce path once (communicator/channel setup) so the first synced backward
            # does not pay for it
            probe = torch.zeros(1024, dtype=torch.float32, device=self.module.store.device)
            dist.all_reduce(probe, group=self.group)

    @contextlib.contextmanager
    def no_sync(self):
        old, self.require_sync = self.require_sync, False
        try:
            yield
        finally:
            self.require_sync = old

    # ---- hooks called from LlamaStack.backward
    def _world(self):
        if self.comm is not None:
            return self.comm.world
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _reduce(self, t):
        if self.comm is not None:
            self.comm.allreduce_mean_(t)
        else:
            _allreduce_mean_(t, self.group)

    def _active(self):
        return self.require_sync and (self._world() > 1 or self.force_sync)

    def on_backward_begin(self):
        if not self._active() or self._queued:
            return
        self._queued = True
        torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def on_layer_done(self, i, events=()):
        if not self._active() or not self.overlap:
            return
        self._launch(self.slices.layer[i], events)

    def _launch(self, t, events=()):
        """all-reduce slice `t` on the side stream, ordered after everything enqueued so far on the compute
        stream and after `events` (the wgrad GEMMs of that layer, which run on their own stream); the host
        does not block."""
        if self._comm_stream is not None:
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            for e in events:
                self._comm_stream.wait_event(e)
            with torch.cuda.stream(self._comm_stream):
                self._reduce(t)
        else:
            self._reduce(t)

    def _finalize(self):
        """end of the autograd pass: reduce what is left, then join the side stream."""
        self._queued = False
        todo = self.slices.rest if self.overlap else self.slices.all_slices()
        for t in todo:
            self._launch(t)
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)

    @torch.no_grad()
    def sync_gradients(self):
        """Explicit one-shot reduction (e.g. once per optimizer step instead of per synced backward;
        mathematically the same mean)."""
        if self._world() > 1:
            for t in self.slices.all_slices():
                self._reduce(t)


def broadcast_task_id(task_id, device, group=None):
    """tasks/loaders.py:176-179: rank 0 picks the task, everyone follows (1 x int64)."""
    t = torch.tensor([task_id], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(t, src=0, group=group)
    return int(t.item())
