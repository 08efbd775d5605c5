"""Loss -- host-side mirror of the reference's losses.py (7-35)."""
import numpy as np
import torch

from . import ops


class LossUtils(object):
    @staticmethod
    def cross_entropy_loss(logits, targets, seq_len_target, return_ws=False):
        """logits [(T*B),V] float32 CUDA; targets [T,B] int; seq_len_target [B].
        Masked sparse softmax CE, per-utterance length-normalised, batch mean."""
        dev = logits.device
        tg = targets.to(device=dev, dtype=torch.int32).contiguous()
        ln = torch.as_tensor(np.asarray(seq_len_target).astype(np.int32)).to(dev)
        loss, lse = ops.masked_ce(logits, tg, ln)
        if return_ws:
            return loss, dict(lse=lse, targets=tg, len=ln)
        return loss
