"""num_utils.py of the reference (sigmoid, softmax over axis 0 of a 1-D array) on the GPU: NumPy in, NumPy out,
float32 arithmetic (csrc/util.hip).  Inside the hot path these are fused into the cell / attention / token kernels;
the module exists so that code written against the reference's helpers keeps working."""
import numpy as np
import torch

from . import _lib, ops


def _run(fn_name, x, n_arg):
    arr = np.asarray(x)
    t = torch.as_tensor(np.ascontiguousarray(arr, dtype=np.float32).reshape(-1), device=ops.default_device())
    y = torch.empty_like(t)
    if t.numel():
        rc = getattr(_lib.lib(), fn_name)(ops._stream(), ops._p(t), ops._p(y), n_arg(t.numel()))
        ops._check(rc, fn_name)
    return y.cpu().numpy().reshape(arr.shape).astype(arr.dtype if arr.dtype.kind == "f" else np.float32)


def sigmoid(x):
    """1 / (1 + exp(-x)) (num_utils.py:6-8)."""
    return _run("asr_sigmoid_f32", x, int)


def softmax(x):
    """exp(x - max) / sum over axis 0 of a 1-D array (num_utils.py:11-14)."""
    if np.asarray(x).ndim != 1:
        raise ValueError("softmax: 1-D input expected (the reference applies it to one score vector)")
    return _run("asr_softmax_f32", x, int)
