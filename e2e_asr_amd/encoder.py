"""Encoder of the seq2seq model -- host-side mirror of the reference's encoder.py.

Same constructor, __call__ signature, defaults and flags as reference encoder.py:15-200;
the computation is the HIP path: per layer one MFMA GEMM per direction for the input
projection and one persistent recurrent kernel (csrc/lstm.hip).  Layout decision: every
tensor stays batch-major [B,T,feat] end to end.  The reference's two transposes per layer
(encoder.py:158,164), the fw/bw concat (:83) and the pyramid reshape (:112-115) all
disappear: the kernel writes fw|bw halves in place and [B,T,2H] -> [B,T/2,4H] is a view.
"""
import os

import numpy as np
import torch

from . import ops
from .base_params import BaseParams, Bunch
from .devcache import dev_i32
from .weights import enc_name


class Encoder(BaseParams):
    """Encodes a padded batch of filterbank frames with a pyramidal (Bi)LSTM stack."""

    @classmethod
    def class_params(cls):
        # encoder.py:18-31 (note: use_lstm defaults False here but the CLI flag below is
        # store_true with default=True, so LSTM is what every reference run uses)
        return Bunch(bi_dir=True, hidden_size=256, out_prob=0.9, skip_step=2, initial_res_fac=1,
                     use_lstm=False, stack_cons=1, max_scaling_down=8)

    def __init__(self, params=None, isTraining=True, variables=None):
        self.params = params if params is not None else self.class_params()
        self.isTraining = isTraining
        self.variables = variables
        self.saved = None          # per-layer activations for the backward pass
        self.dropout_seed = 0

    def get_cell(self):
        """encoder.py:42-53.  The cell is realised inside csrc/lstm.hip; only LSTM exists
        (the GRU branch is unreachable from the reference CLI, encoder.py:187)."""
        if not self.params.use_lstm:
            raise NotImplementedError("GRUCell encoder: not on the hot path (reference CLI always sets use_lstm)")
        return "BasicLSTMCell(%d)" % self.params.hidden_size

    def _layer_weights(self, depth):
        v = self.variables
        if self.params.bi_dir:
            return (v[enc_name(depth, "fw", "kernel")], v[enc_name(depth, "fw", "bias")],
                    v[enc_name(depth, "bw", "kernel")], v[enc_name(depth, "bw", "bias")])
        return (v[enc_name(depth, "", "kernel", False)], v[enc_name(depth, "", "bias", False)], None, None)

    def _pyramid_plan(self, T, seq_len_host):
        """encoder.py:94-119: pad (skip - max_len % skip) frames when max_len % skip != 0, then
        reshape; tf.reshape fails when the padded T is not divisible -- same error here."""
        skip = self.params.skip_step
        rem = int(seq_len_host.max()) % skip
        t_out = T + (skip - rem if rem else 0)
        if t_out % skip:
            raise ValueError("pyramid: padded length %d not divisible by skip_step %d" % (t_out, skip))
        return t_out

    def __call__(self, encoder_input, seq_len, num_layers):
        """encoder_input [B,T,F] float32 CUDA; seq_len [B] (host array or tensor);
        num_layers {task: depth}.  Returns (attention_states{depth: [B,T_d,D]},
        time_major_states{depth}, seq_len_inps{depth: host int64 array}) -- encoder.py:122-180."""
        params = self.params
        self.get_cell()
        attention_states, time_major_states, seq_len_inps = {}, {}, {}
        max_depth = 0
        for task, nl in num_layers.items():
            (time_major_states if task == "state" else attention_states)[nl] = None
            max_depth = max(max_depth, nl)
        lens = np.asarray(seq_len.cpu() if torch.is_tensor(seq_len) else seq_len).astype(np.int64)
        x = encoder_input
        res = params.initial_res_fac
        if res > 1:                                               # encoder.py:150-153
            x = x[:, ::res, :].contiguous()
            lens = np.ceil(lens / float(res)).astype(np.int64)
        keep = params.out_prob if self.isTraining else 1.0
        save = self.isTraining
        self.saved = []
        ahead = {}                                                 # {depth: (kx_cat, bias_cat)}: one launch for all layers
        if x.is_cuda and params.bi_dir and ops._KXCAT >= 1 and os.environ.get("ASR_KXCAT_MULTI", "1") != "0":
            todo = [(d,) + tuple(self._layer_weights(d)) for d in range(1, max_depth + 1)]
            todo = [t for t in todo if t[1].shape[1] // 4 in ops.LSTM_KERNEL_H]
            for c0 in range(0, len(todo), 4):
                chunk = todo[c0:c0 + 4]
                for t, kc in zip(chunk, ops.concat_kx_layers([t[1:] for t in chunk])):
                    ahead[t[0]] = kc
        for i in range(max_depth):
            d = i + 1
            B, T, _ = x.shape
            reduce_after = params.skip_step > 1 and i != max_depth - 1 and res < params.max_scaling_down
            t_out = self._pyramid_plan(T, lens) if reduce_after else T
            lens_dev = dev_i32(lens, x.device)
            kf, bf, kb, bb = self._layer_weights(d)
            seed = (self.dropout_seed * 1000003 + d * 7919) & 0x7FFFFFFF
            kx, bc = ahead.get(d, (None, None))
            r = ops.lstm_layer_fwd(x, lens_dev, kf, bf, kb, bb, t_out=t_out, save=save,
                                   keep_prob=keep, seed=seed, kx_cat=kx, bias_cat=bc)
            out = r[0] if save else r
            if save:
                self.saved.append(dict(x=x, lens=lens, lens_dev=lens_dev, gates=r[1], c=r[2], hprev=r[3], kx=getattr(r[1], "kx_cat", None), out=out,
                                       T=T, t_out=t_out, keep=keep, seed=seed, depth=d))
            view = out[:, :T] if t_out != T else out
            if d in time_major_states:
                time_major_states[d] = view.transpose(0, 1)
            if d in attention_states:
                attention_states[d] = view
            seq_len_inps[d] = lens
            if reduce_after:                                       # encoder.py:172-176
                x = out.view(B, t_out // params.skip_step, out.shape[2] * params.skip_step)
                lens = np.ceil(lens / float(params.skip_step)).astype(np.int64)
                res *= params.skip_step
            else:
                x = out
        return attention_states, time_major_states, seq_len_inps

    def backward(self, d_states, variables=None, on_layer_done=None):
        """Gradient of __call__ (tf.gradients through encoder.py:122-180).  d_states: {depth:
        gradient w.r.t. attention_states[depth], [B,T_d,D]}.  Weight gradients are accumulated
        into the flat gradient buffer; the pyramid reshape is again only a view: the input
        gradient of layer d+1 [B,T/2,4H] IS the output gradient of layer d [B,T,2H]."""
        v = self.variables if variables is None else variables
        v.ensure_grad()
        dx = None
        for sv in reversed(self.saved):
            d = sv["depth"]
            out = sv["out"]
            if dx is not None:
                dout = dx.view(out.shape)
                if d in d_states and d_states[d] is not None:
                    dout[:, :sv["T"]].add_(d_states[d])
            else:
                dout = d_states[d]
                if dout.shape[1] != sv["t_out"]:
                    pad = dout.new_zeros(dout.shape[0], sv["t_out"], dout.shape[2])
                    pad[:, :dout.shape[1]] = dout
                    dout = pad
            if self.params.bi_dir:
                names = [enc_name(d, "fw", "kernel"), enc_name(d, "fw", "bias"), enc_name(d, "bw", "kernel"), enc_name(d, "bw", "bias")]
                kf, kb = v[names[0]], v[names[2]]
                g = [v.grad_of(n) for n in names]
            else:
                names = [enc_name(d, "", "kernel", False), enc_name(d, "", "bias", False)]
                kf, kb = v[names[0]], None
                g = [v.grad_of(n) for n in names] + [None, None]
            dx = ops.lstm_layer_bwd(sv["x"], sv["lens_dev"], kf, kb, dout.contiguous(), sv["gates"], sv["c"], sv["hprev"],
                                    g[0], g[1], g[2], g[3], need_dx=d > 1, keep_prob=sv["keep"], seed=sv["seed"],
                                    join=False, kx_cat=sv["kx"])
            if on_layer_done is not None:
                on_layer_done(d)
        self.saved = None
        return dx

    @classmethod
    def add_parse_options(cls, parser):
        # encoder.py:182-200 -- same flags, same defaults
        for flags, kw in (
            (("-out_prob", "--out_prob"), dict(default=0.9, type=float, help="Output keep probability for dropout")),
            (("-use_lstm", "--use_lstm"), dict(default=True, action="store_true", help="LSTM cell (always on)")),
            (("-hsize", "--hidden_size"), dict(default=256, type=int, help="Hidden layer size")),
            (("-skip_step", "--skip_step"), dict(default=2, type=int, help="Frame skipping factor up the stack")),
            (("-init_res_fac", "--initial_res_fac"), dict(default=1, type=int, help="Initial resolution factor")),
            (("-stack_cons",), dict(default=1, type=int, help="Stacking consecutive frames in input")),
            (("-max_scaling_down",), dict(default=8, type=int, help="Maximum reduction in resolution")),
        ):
            parser.add_argument(*flags, **kw)
