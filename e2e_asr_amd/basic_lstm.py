"""basic_lstm.py of the reference on the GPU: `BasicLSTM(weight, bias)(x, (c, h)) -> (new_c, new_h)`, one
BasicLSTMCell step for ONE hypothesis with NumPy in / NumPy out (basic_lstm.py:10-23: [x,h].W + b -> i,j,f,o;
c' = c*sigmoid(f+1) + sigmoid(i)*tanh(j); h' = sigmoid(o)*tanh(c')).  The weights stay resident on the device; the step
is the fused skinny-MFMA cell kernel the decoder uses (csrc/skinny.hip via asr_lstm_cell_fwd), float32."""
import numpy as np
import torch

from . import ops


class BasicLSTM(object):
    def __init__(self, weight, bias, device=None):
        self.device = torch.device(device) if device is not None else ops.default_device()
        self.lstm_w = torch.as_tensor(np.asarray(weight, np.float32), device=self.device).contiguous()
        self.lstm_b = torch.as_tensor(np.asarray(bias, np.float32), device=self.device).contiguous()
        self.H = self.lstm_w.shape[1] // 4

    def __call__(self, x, lstm_state):
        c, h = lstm_state
        dt = np.asarray(x).dtype
        to = lambda a: torch.as_tensor(np.asarray(a, np.float32).reshape(1, -1), device=self.device)
        xt, ct, ht = to(x), to(c), to(h)
        if xt.shape[1] + self.H != self.lstm_w.shape[0] or ct.shape[1] != self.H or ht.shape[1] != self.H:
            raise ValueError("BasicLSTM: x[%d], c[%d], h[%d] do not fit a [%d,%d] kernel" % (
                xt.shape[1], ct.shape[1], ht.shape[1], self.lstm_w.shape[0], self.lstm_w.shape[1]))
        new_c, new_h = ops.lstm_cell(xt, ht, ct, self.lstm_w, self.lstm_b)[:2]
        out = lambda t: t[0].cpu().numpy().astype(dt if dt.kind == "f" else np.float32)
        return out(new_c), out(new_h)
