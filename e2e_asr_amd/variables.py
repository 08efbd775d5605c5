"""Variable store: the stand-in for the TF graph's global variable collection.

The reference creates variables implicitly inside tf.variable_scope("model") (train.py:184)
and finds them again by NAME (beam_search.py:56-98, tf_utils.py:53-90).  Here all trainable
variables live in ONE flat float32 device buffer (plus flat grad / Adam m / Adam v buffers
of the same layout): the data-parallel gradient exchange is then a single RCCL all-reduce of
one contiguous 42.5 MB bucket and clip+Adam is one fused kernel over the flat buffers.
Each variable is a view into the flat buffer, 16-byte aligned.
"""
import numpy as np
import torch


class VariableStore(object):
    def __init__(self, device):
        self.device = torch.device(device)
        self._specs = []           # (name, shape, offset, numel)
        self._index = {}
        self.flat = None
        self.grad = None
        self.adam_m = None
        self.adam_v = None
        self.adam_slots = {}       # optimizer name -> (m, v)  (LM has its own 'AdamLM', lm_model.py:76)

    # ---- construction -----------------------------------------------------------
    @classmethod
    def from_arrays(cls, arrays, device):
        """arrays: dict name -> np.ndarray (TF layouts).  Order is preserved."""
        st = cls(device)
        off = 0
        for name, a in arrays.items():
            a = np.asarray(a, np.float32)
            st._specs.append((name, tuple(a.shape), off, a.size))
            st._index[name] = len(st._specs) - 1
            off += (a.size + 3) // 4 * 4
        host = np.zeros(off, np.float32)
        for (name, shape, o, n) in st._specs:
            host[o:o + n] = np.asarray(arrays[name], np.float32).reshape(-1)
        st.flat = torch.from_numpy(host).to(st.device)
        return st

    # ---- access ------------------------------------------------------------------
    def names(self):
        return [s[0] for s in self._specs]

    def __contains__(self, name):
        return name in self._index

    def _view(self, buf, name):
        _, shape, o, n = self._specs[self._index[name]]
        return buf[o:o + n].view(shape)

    def __getitem__(self, name):
        return self._view(self.flat, name)

    def get(self, name, default=None):
        return self[name] if name in self._index else default

    def grad_of(self, name):
        self.ensure_grad()
        return self._view(self.grad, name)

    def ensure_grad(self):
        if self.grad is None:
            self.grad = torch.zeros_like(self.flat)

    def ensure_adam(self, slot="Adam"):
        if slot not in self.adam_slots:
            self.adam_slots[slot] = (torch.zeros_like(self.flat), torch.zeros_like(self.flat))
        return self.adam_slots[slot]

    def num_params(self):
        return sum(s[3] for s in self._specs)

    # ---- checkpoint interchange (names/layouts of the TF checkpoint) ----------------
    def to_arrays(self):
        host = self.flat.detach().cpu().numpy()
        return {name: host[o:o + n].reshape(shape).copy() for (name, shape, o, n) in self._specs}

    def assign(self, arrays, strict=False):
        """Name-intersection restore (tf_utils.restore_common_variables, tf_utils.py:53-63):
        variables missing from `arrays` keep their value; shape mismatches are reported."""
        restored = []
        for name, a in arrays.items():
            if name not in self._index:
                if strict:
                    raise KeyError(name)
                continue
            _, shape, _, _ = self._specs[self._index[name]]
            a = np.asarray(a, np.float32)
            if tuple(a.shape) != shape:
                print("Shape wanted: %s, Shape stored: %s for %s" % (shape, a.shape, name))
                continue
            self[name].copy_(torch.from_numpy(a).to(self.device))
            restored.append(name)
        return restored
