"""Loss -- host-side mirror of the reference's losses.py (7-35)."""
import numpy as np
import torch

from . import ops
from .devcache import dev_i32


class LossUtils(object):
    @staticmethod
    def cross_entropy_loss(logits, targets, seq_len_target, return_ws=False):
        """logits [(T*B),V] float32 CUDA; targets [T,B] int; seq_len_target [B].
        Masked sparse softmax CE, per-utterance length-normalised, batch mean."""
        dev = logits.device
        tg = targets.to(device=dev, dtype=torch.int32).contiguous()
        ln = dev_i32(seq_len_target, dev)
        loss, lse = ops.masked_ce(logits, tg, ln)
        if return_ws:
            return loss, dict(lse=lse, targets=tg, len=ln)
        return loss
