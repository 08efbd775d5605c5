"""Loss -- host-side mirror of the reference's losses.py (7-35)."""
import numpy as np
import torch

from . import ops
from .devcache import dev_i32


class LossUtils(object):
    @staticmethod
    def cross_entropy_loss(logits, targets, seq_len_target, return_ws=False, grad_scale=None):
        """logits [(T*B),V] float32 CUDA; targets [T,B] int; seq_len_target [B].
        Masked sparse softmax CE, per-utterance length-normalised, batch mean.
        grad_scale (device scalar, training): d total_loss / d this loss -- the gradient w.r.t. the logits is then formed in the
        same pass over the logits and rides along in the workspace dict (`dlogits`)."""
        dev = logits.device
        tg = targets.to(device=dev, dtype=torch.int32).contiguous()
        ln = dev_i32(seq_len_target, dev)
        if grad_scale is not None and return_ws:
            loss, lse, dlogits = ops.masked_ce_fwd_bwd(logits, tg, ln, grad_scale)
            return loss, dict(lse=lse, targets=tg, len=ln, dlogits=dlogits)
        loss, lse = ops.masked_ce(logits, tg, ln)
        if return_ws:
            return loss, dict(lse=lse, targets=tg, len=ln)
        return loss
