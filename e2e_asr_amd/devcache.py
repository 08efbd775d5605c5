"""Content-keyed cache of small integer device tensors (lengths, token ids).

The reference feeds these through tf.data each step; here the same length vector / id
matrix is needed by several kernels per step, and re-uploading pageable host memory both
costs a synchronous copy and is illegal inside hipGraph capture.  Keyed by bytes, so a new
batch simply misses -- and a miss uploads from pinned memory without synchronising."""
import collections

import numpy as np
import torch

_CACHE = collections.OrderedDict()
_MAX = 256


def dev_i32(array, device):
    a = np.ascontiguousarray(np.asarray(array).astype(np.int32))
    key = (str(device), a.shape, a.tobytes())
    t = _CACHE.get(key)
    if t is None:
        if torch.device(device).type == "cuda":
            # pinned staging + asynchronous copy: a pageable source makes the copy synchronous, i.e. the host waits for
            # everything queued on the stream before it, and loses its run-ahead once per new array (a fresh batch with ragged
            # lengths misses ~8 times per step: 10.7 instead of 9.1 ms, scripts/bench_fresh.py).  The caching host allocator keeps
            # the pinned block alive until the copy has run.
            t = torch.from_numpy(a).pin_memory().to(device, non_blocking=True)
        else:
            t = torch.from_numpy(a).to(device)
        _CACHE[key] = t
        if len(_CACHE) > _MAX:
            _CACHE.popitem(last=False)
    else:
        _CACHE.move_to_end(key)
    return t
