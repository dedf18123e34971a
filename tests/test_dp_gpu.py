"""The data-parallel wrapper on a real GPU with a world of one (the box has a single MI355X): RCCL all-reduces of the flat
gradient slices are launched from inside the backward on the communication stream, ordered after the wgrad GEMMs of the
side stream by events, and joined at the end of the autograd pass.  With one rank every all-reduce is the identity, so
the gradients must equal the plain run's bit for bit -- what this checks is the stream/event/callback wiring under the
real runtime, for both transports (torch.distributed/nccl and the C-ABI nv_comm_*)."""
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _step(model, wrapped, seed):
    from navillm_amd.synthetic import SyntheticEpisodes, nav_step
    from navillm_amd.losses import CrossEntropyLoss
    ep = SyntheticEpisodes(model.cfg, 3, seed=seed, instr_len=150, device=torch.device(DEV))
    crit = CrossEntropyLoss()
    model.zero_grad()
    torch.manual_seed(1)
    for i in range(2):
        nav_step(wrapped, crit, ep, train=True, last=(i == 1))      # step 0 inside no_sync, step 1 synced
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in model.store.grad.items()}


@pytest.mark.parametrize("transport", ["torch", "rccl"])
def test_dp_wrapper_world1_matches_plain_run(transport):
    from navillm_amd import config as nvcfg
    from navillm_amd.nav_model import NavModel
    from navillm_amd.parallel import NavDataParallel, RcclComm
    cfg = nvcfg.NavConfig(hidden_size=512, num_layers=3, num_heads=4, intermediate_size=1408, base_vocab_size=1000,
                          enc_hidden_size=256, enc_num_heads=4, enc_intermediate_size=512, image_feat_size=768)
    model = NavModel(nav_config=cfg, device=torch.device(DEV), seed=4)
    model.train()
    base = _step(model, model, 17)
    comm = None
    if transport == "torch":
        if not dist.is_initialized():
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0,
                                    device_id=torch.device(DEV))
    else:
        comm = RcclComm(0, 1)
    ddp = NavDataParallel(model, comm=comm, force_sync=True)
    got = _step(model, ddp, 17)
    model._dp = None
    for k in base:
        assert torch.equal(base[k], got[k]), f"{transport}: gradient buffer {k} differs from the plain run"
    assert float(base["lm"].float().abs().sum()) > 0
    if comm is not None:
        comm.close()
