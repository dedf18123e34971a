"""CPU, world_size 2, gloo: the data-parallel gradient path (navillm_amd/parallel.py) without a GPU.
A FlatStore on the CPU stands in for the model's HBM buffers; a tiny autograd Function drives the same
hooks LlamaStack.backward drives (on_backward_begin, on_layer_done(i) in reverse layer order, then the
end-of-backward callback)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import tiny_cfg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeModel(torch.nn.Module):
    def __init__(self, cfg):
        super().__init__()
        from navillm_amd.flat import FlatStore
        self.cfg = cfg
        self.store = FlatStore(cfg, "cpu")
        self._dp = None


class _Backward(torch.autograd.Function):
    """writes rank-dependent 'gradients' the way the real backward does: layer by layer, last first"""

    @staticmethod
    def forward(ctx, x, model, rank, scale):
        ctx.model, ctx.rank, ctx.scale = model, rank, scale
        return x * 1.0

    @staticmethod
    def backward(ctx, g):
        m, st = ctx.model, ctx.model.store
        if m._dp is not None:
            m._dp.on_backward_begin()
        for i in reversed(range(m.cfg.num_layers)):
            s, e = st.layer_slice(i)
            st.grad["lm"][s:e] += ctx.scale * (ctx.rank + 1) * (i + 1)
            if m._dp is not None:
                m._dp.on_layer_done(i, [])
        st.grad["f32"] += ctx.scale * (ctx.rank + 1) * 0.5
        first = st.layer_slice(0)[0]
        st.grad["lm"][:first] += ctx.scale * (ctx.rank + 1) * 7.0
        return g, None, None, None


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from navillm_amd.parallel import init_distributed_device, NavDataParallel, broadcast_task_id
    dev, r, w = init_distributed_device(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = tiny_cfg("bf16")
    model = _FakeModel(cfg)
    # rank-dependent parameters: the wrapper must broadcast rank 0's
    for t in model.store.param.values():
        t.fill_(float(rank + 1))
    ddp = NavDataParallel(model, reduce="backward")
    ok = all(bool((t.float() == 1.0).all()) for t in model.store.param.values())
    # the wrapper must not become a child module of the model it wraps (.train()/.eval()/.state_dict() recursed)
    ddp.train(); ddp.eval(); model.train(); model.eval()
    ddp.state_dict(); model.state_dict()
    ok &= model._dp is ddp and "_dp" not in dict(model.named_children())
    # the one-shot exchange (the optimizer's flush) issues the slices in the order the overlapped exchange does -- layers last to first, then
    # the rest -- so that ranks reaching the optimizer step by different paths still pair the same slices (round 6)
    sl = ddp.slices.all_slices()
    ok &= [t.data_ptr() for t in sl[:cfg.num_layers]] == [t.data_ptr() for t in reversed(ddp.slices.layer)] and \
        [t.data_ptr() for t in sl[cfg.num_layers:]] == [t.data_ptr() for t in ddp.slices.rest]
    # the reference's own type check (mp3d_agent.py:661) accepts the wrapper unchanged
    ok &= isinstance(ddp, torch.nn.parallel.DistributedDataParallel) and ddp.module is model
    st = model.store
    x = torch.ones(1, requires_grad=True)
    mean_rank = sum(r_ + 1 for r_ in range(world)) / world          # mean of (rank+1)
    s0, e0 = st.layer_slice(0)

    def all_mean(k):
        good = True
        for i in range(cfg.num_layers):
            s, e = st.layer_slice(i)
            good &= bool(torch.allclose(st.grad["lm"][s:e].float(), torch.full((e - s,), k * mean_rank * (i + 1)), rtol=1e-2))
        good &= bool(torch.allclose(st.grad["f32"], torch.full_like(st.grad["f32"], k * mean_rank * 0.5)))
        good &= bool(torch.allclose(st.grad["lm"][:s0].float(), torch.full((s0,), k * mean_rank * 7.0), rtol=1e-2))
        return good

    def all_local(k):
        return abs(float(st.grad["lm"][s0].float()) - k * (rank + 1) * 1.0) < 1e-6 and \
            bool(torch.allclose(st.grad["f32"], torch.full_like(st.grad["f32"], k * (rank + 1) * 0.5)))

    # ---- reduce="backward" (DDP semantics).  step 1: inside no_sync -> purely local accumulation
    with ddp.no_sync():
        _Backward.apply(x, model, rank, 1.0).sum().backward()
    ok &= all_local(1)
    # step 2: synced backward -> every slice becomes the mean over ranks of the ACCUMULATED gradient
    _Backward.apply(x, model, rank, 1.0).sum().backward()
    ok &= all_mean(2)
    # explicit one-shot reduction is idempotent on already-averaged gradients
    before = st.grad["f32"].clone()
    ddp.sync_gradients()
    ok &= bool(torch.allclose(st.grad["f32"], before))

    # ---- reduce="step" (default): nothing moves until the optimizer step, then exactly once
    st.zero_grad()
    ddp.reduce = "step"
    with ddp.no_sync():
        _Backward.apply(x, model, rank, 1.0).sum().backward()
    _Backward.apply(x, model, rank, 1.0).sum().backward()          # "synced" backwards of the last step: still local
    _Backward.apply(x, model, rank, 1.0).sum().backward()
    ok &= all_local(3) and ddp._pending
    ddp.flush()                                                     # what FlatAdamW.clip_grad_norm_/step call first
    ok &= all_mean(3) and not ddp._pending
    ddp.flush()                                                     # nothing pending: no second averaging
    ok &= all_mean(3)
    # ... or overlapped with the last backward before the step
    st.zero_grad()
    with ddp.no_sync():
        _Backward.apply(x, model, rank, 1.0).sum().backward()
    with ddp.final_backward():
        _Backward.apply(x, model, rank, 1.0).sum().backward()
    ok &= all_mean(2) and not ddp._pending
    ddp.flush()
    ok &= all_mean(2)
    # ---- a backward that runs OUTSIDE the autograd engine (navillm_amd/episode.py::finish_episode in its default form: the steps'
    # backward() calls only record output gradients; finish_episode() walks the layers itself inside final_backward(), hands every
    # finished layer to the wrapper and closes the exchange by hand)
    st.zero_grad()
    with ddp.final_backward():
        ddp.on_deferred_backward_begin()
        for i in reversed(range(cfg.num_layers)):
            s, e = st.layer_slice(i)
            st.grad["lm"][s:e] += 2.0 * (rank + 1) * (i + 1)
            ddp.on_layer_done(i, [])
        st.grad["f32"] += 2.0 * (rank + 1) * 0.5
        st.grad["lm"][:st.layer_slice(0)[0]] += 2.0 * (rank + 1) * 7.0
        ok &= ddp._exchanging()
        ddp._finalize()
    ok &= all_mean(2) and not ddp._pending
    ddp.flush()                                                     # nothing pending: the optimizer's flush must not average again
    ok &= all_mean(2)
    # ... and without final_backward(): nothing moves until the optimizer's flush
    st.zero_grad()
    ddp.on_deferred_backward_begin()
    for i in reversed(range(cfg.num_layers)):
        s, e = st.layer_slice(i)
        st.grad["lm"][s:e] += 1.0 * (rank + 1) * (i + 1)
        ddp.on_layer_done(i, [])
    st.grad["f32"] += 1.0 * (rank + 1) * 0.5
    st.grad["lm"][:st.layer_slice(0)[0]] += 1.0 * (rank + 1) * 7.0
    ok &= all_local(1) and ddp._pending and not ddp._exchanging()
    ddp.flush()
    ok &= all_mean(1) and not ddp._pending
    # ---- which tensors AdamW updates must agree across ranks (`obj_projector` is touched only when the LOCAL batch
    # carries objects): the union over ranks is taken before the optimizer assigns step counts
    st.touched.clear()
    st.touch_layers()
    only_rank1 = st.names["f32"][0]
    if rank == 1:
        st.touch(only_rank1)
    ddp.merge_touched()
    ok &= only_rank1 in st.touched
    got = [None] * world
    dist.all_gather_object(got, sorted(st.touched))
    ok &= got[0] == got[1]
    from navillm_amd.optim import active_segments
    born = {n: 0 for n in st.touched}
    segs = [None] * world
    dist.all_gather_object(segs, active_segments(st, born))
    ok &= segs[0] == segs[1]
    # task-id broadcast (tasks/loaders.py:176-179)
    ok &= broadcast_task_id(5 if rank == 0 else 9, dev) == 5
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_mean_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)], res


def _preflight_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from navillm_amd.parallel import init_distributed_device, NavDataParallel, dp_preflight
    dev, r, w = init_distributed_device(backend="gloo")
    rep, comm, group = dp_preflight(dev, rank, world, watchdog_s=120.0)
    names = [s["stage"] for s in rep["stages"]]
    ok = names == ["control_plane", "nv_comm_init", "agreement", "torch_nccl_group", "agreement_fallback"], names
    ok = ok[0] if isinstance(ok, tuple) else ok
    # no GPU here: the C-ABI communicator is skipped on every rank, the ranks AGREE on the fallback, and the group it hands back works
    ok &= rep["transport"] == "torch" and comm is None and group is not None and rep["control_backend"] == "gloo"
    ok &= rep["stages"][1]["ok"] is False and "skipped" in rep["stages"][1] and rep["stages"][3]["ok"] and rep["stages"][0]["all_ranks_present"]
    cfg = tiny_cfg("bf16")
    model = _FakeModel(cfg)
    ddp = NavDataParallel(model, group=group, reduce="step")
    model.store.grad["f32"].fill_(float(rank + 1))
    ddp._pending = True
    ddp.flush()
    ok &= bool((model.store.grad["f32"] == 1.5).all())
    import json
    json.dumps(rep)                                   # the report goes into the bench's JSON line as it is
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_preflight_picks_a_transport_every_rank_agrees_on_world2_gloo():
    """round 6: `bench.py --gpus N` runs `dp_preflight` before it builds the model.  On CPU ranks the C-ABI
    communicator is not available, so this exercises the other half: every rank records its stages, the ranks agree through the
    control plane, the fallback group is created collectively and verified, and the report is JSON-serialisable."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_preflight_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)], res
