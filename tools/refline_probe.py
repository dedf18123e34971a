"""The reference's launch line alone (B = 1 per GPU, accumulation 8: bench.py::reference_launch_extra) without the rest of the bench:
python tools/refline_probe.py [--model vicuna-7b]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    a = bench.parse()
    from navillm_amd.nav_model import NavModel
    from navillm_amd.losses import CrossEntropyLoss
    from navillm_amd.optim import FlatAdamW
    torch.set_num_threads(max(1, min(16, bench.usable_cpus())))
    device = torch.device("cuda:0")
    cfg = bench.make_cfg(a)
    model = NavModel(nav_config=cfg, device=device, seed=0)
    model.train()
    opt = FlatAdamW(model, lr=a.lr)
    out = bench.reference_launch_extra(a, cfg, model, opt, CrossEntropyLoss(), device, 1634)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
