"""Drop-in for `criterion = nn.CrossEntropyLoss(ignore_index=-100, reduction='sum')` (train.py:229),
used on the navigation/object logits at tasks/agents/mp3d_agent.py:750,823."""
import torch
from . import functions as Fn


class CrossEntropyLoss(torch.nn.Module):
    def __init__(self, ignore_index=-100, reduction="sum"):
        super().__init__()
        if ignore_index != -100 or reduction != "sum":
            raise NotImplementedError("the navigation path uses ignore_index=-100, reduction='sum'")

    def forward(self, logits, targets):
        return Fn.ActionCE.apply(logits, targets.to(device=logits.device, dtype=torch.int64).contiguous())
