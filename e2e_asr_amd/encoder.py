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
from .weights import enc_gru_name, enc_name


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
        self._require_lstm()

    def _require_lstm(self):
        """Kept for callers of earlier rounds: both cells of encoder.py:42-53 are built now (round 5)."""
        return None

    def get_cell(self):
        """encoder.py:42-53.  The cells are realised inside csrc/lstm.hip (BasicLSTMCell: the persistent recurrent kernels) and
        csrc/gru.hip (GRUCell -- the `class_params()` default, encoder.py:27, which the reference CLI always overrides,
        encoder.py:187: a plain one-workgroup-per-utterance kernel, off the measured path)."""
        return ("BasicLSTMCell(%d)" if self.params.use_lstm else "GRUCell(%d)") % self.params.hidden_size

    def _gru_cells(self, depth, grad=False):
        """Per direction (gates kernel, gates bias, candidate kernel, candidate bias) of a GRUCell layer, or their gradients."""
        v = self.variables
        get = v.grad_of if grad else v.__getitem__
        dirs = ("fw", "bw") if self.params.bi_dir else ("",)
        return [tuple(get(enc_gru_name(depth, dr, part, leaf, self.params.bi_dir))
                      for part, leaf in (("gates", "kernel"), ("gates", "bias"), ("candidate", "kernel"), ("candidate", "bias")))
                for dr in dirs]

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
        gru = not params.use_lstm
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
        if not gru and x.is_cuda and params.bi_dir and ops._KXCAT >= 1 and os.environ.get("ASR_KXCAT_MULTI", "1") != "0":
            todo = [(d,) + tuple(self._layer_weights(d)) for d in range(1, max_depth + 1)]
            todo = [t for t in todo if t[1].shape[1] // 4 in ops.LSTM_KERNEL_H]
            for c0 in range(0, len(todo), 4):
                chunk = todo[c0:c0 + 4]
                for t, kc in zip(chunk, ops.concat_kx_layers([t[1:] for t in chunk])):
                    ahead[t[0]] = kc
        # Operands as bf16 planes (csrc/gemm_p3.hip): the recurrent kernels write their outputs (h for the layer
        # above, h_prev, and in the backward dG) as planes, the weights are split once per step, and no GEMM of the encoder splits
        # fp32 operands inside its k-loop.  np_ = 0: off (ASR_P3=0, exact-fp32 mode, CPU tensors).
        np_ = ops.p3_planes() if x.is_cuda and params.bi_dir and not gru else 0
        x_p3 = None                                                 # P3 image of the current layer's input, if any
        wsplit = {}                                                 # {depth: (K_x^T image, unit-major K_x image or None)}: ONE launch
        if np_ and ahead:
            ds = [d for d in sorted(ahead) if d > 1 and ahead[d][0] is not None]
            jobs = [(ahead[d][0], np_, True, 0) for d in ds]
            if save:
                jobs += [(ahead[d][0], np_, False, ahead[d][0].shape[1] // 8) for d in ds]
            if jobs and len(jobs) <= 8:
                outs = ops.p3_split_many(jobs)
                for i, d in enumerate(ds):
                    wsplit[d] = (outs[i], outs[len(ds) + i] if save else None)
        for i in range(max_depth):
            d = i + 1
            B, T, IN = x.shape
            reduce_after = params.skip_step > 1 and i != max_depth - 1 and res < params.max_scaling_down
            t_out = self._pyramid_plan(T, lens) if reduce_after else T
            lens_dev = dev_i32(lens, x.device)
            seed = (self.dropout_seed * 1000003 + d * 7919) & 0x7FFFFFFF
            if gru:                                                # encoder.py:47-48: tf.nn.rnn_cell.GRUCell
                r = ops.gru_layer_fwd(x.contiguous(), lens_dev, self._gru_cells(d), t_out=t_out, save=save, keep_prob=keep, seed=seed)
                out = r[0] if save else r
                if save:
                    self.saved.append(dict(x=x, lens=lens, lens_dev=lens_dev, gru=r[1:], out=out, T=T, t_out=t_out, keep=keep,
                                           seed=seed, depth=d))
                view = out[:, :T] if t_out != T else out
                if d in time_major_states:
                    time_major_states[d] = view.transpose(0, 1)
                if d in attention_states:
                    attention_states[d] = view
                seq_len_inps[d] = lens
                if reduce_after:
                    x = out.view(B, t_out // params.skip_step, out.shape[2] * params.skip_step)
                    lens = np.ceil(lens / float(params.skip_step)).astype(np.int64)
                    res *= params.skip_step
                else:
                    x = out
                continue
            kf, bf, kb, bb = self._layer_weights(d)
            kx, bc = ahead.get(d, (None, None))
            p3 = None
            H = kf.shape[1] // 4
            if np_ and kx is not None and ops.lstm_p3_supported(B, T, IN, H, 2):
                p3 = dict(np=np_)
                if x_p3 is not None and x_p3.np == np_ and IN % 16 == 0:
                    p3["x"] = x_p3
                    p3["kxT"] = wsplit[d][0] if d in wsplit else ops.p3_split(kx, np_, transpose=True)     # K_x^T [8H][in], once per step
                    if d in wsplit and wsplit[d][1] is not None:
                        p3["kxu_ahead"] = wsplit[d][1]
                elif save and d == 1 and IN <= 128:
                    # first layer: the frames as a 128-column image for its weight gradient (the projection itself runs
                    # inside the recurrent kernel)
                    p3["x"] = ops.p3_split(x.reshape(B * T, IN), np_, cols=128)
                if i != max_depth - 1:
                    p3["out"] = ops.p3_alloc(B * t_out, 2 * H, np_, x.device)
                if save and p3.get("x") is not None and p3["x"].cols % 128 == 0 and (d == 1 or IN % 256 == 0):
                    # decided HERE, once, for the backward pass too (p3["bwd"]): with h_prev as planes there is no fp32 h_prev, so the
                    # BPTT of this layer must run on plane operands -- backward() follows the record, it does not re-derive it
                    p3["hprev"] = ops.p3_alloc(B * T, 2 * H, np_, x.device)
                    p3["bwd"] = True
            r = ops.lstm_layer_fwd(x, lens_dev, kf, bf, kb, bb, t_out=t_out, save=save,
                                   keep_prob=keep, seed=seed, kx_cat=kx, bias_cat=bc, p3=p3)
            out = r[0] if save else r
            if save:
                self.saved.append(dict(x=x, lens=lens, lens_dev=lens_dev, gates=r[1], c=r[2], hprev=r[3], kx=getattr(r[1], "kx_cat", None), out=out,
                                       T=T, t_out=t_out, keep=keep, seed=seed, depth=d, p3=p3))
            x_p3 = None
            if p3 is not None and p3.get("out") is not None:
                o = p3["out"]
                skip = params.skip_step if reduce_after else 1
                x_p3 = ops.P3(o.buf, o.rows // skip, o.cols * skip, o.np)      # the pyramid reshape is a view here too
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
            if "gru" in sv:                                        # GRUCell layer (csrc/gru.hip)
                gx, cx, hprev, rh = sv["gru"]
                dx = ops.gru_layer_bwd(sv["x"].contiguous(), sv["lens_dev"], self._gru_cells(d), dout.contiguous(), gx, cx, hprev, rh,
                                       self._gru_cells(d, grad=True), need_dx=d > 1, keep_prob=sv["keep"], seed=sv["seed"])
                if on_layer_done is not None:
                    on_layer_done(d)
                continue
            if self.params.bi_dir:
                names = [enc_name(d, "fw", "kernel"), enc_name(d, "fw", "bias"), enc_name(d, "bw", "kernel"), enc_name(d, "bw", "bias")]
                kf, kb = v[names[0]], v[names[2]]
                g = [v.grad_of(n) for n in names]
            else:
                names = [enc_name(d, "", "kernel", False), enc_name(d, "", "bias", False)]
                kf, kb = v[names[0]], None
                g = [v.grad_of(n) for n in names] + [None, None]
            p3 = sv.get("p3")
            if p3 is not None and p3.get("bwd"):
                B_, T_, IN_ = sv["x"].shape
                H = kf.shape[1] // 4
                p3 = dict(p3)
                p3["dg"] = ops.p3_alloc(B_ * T_, 8 * H, p3["np"], sv["x"].device)
                p3["colmap"] = ops.p3_colmap(H, sv["x"].device)
                if d > 1:       # K_x with unit-major columns, for dX (formed with the forward's weight images, one launch per step)
                    p3["kxu"] = p3.get("kxu_ahead") or ops.p3_split(sv["kx"], p3["np"], unit_major_h=H)
            else:
                p3 = None
            dx = ops.lstm_layer_bwd(sv["x"], sv["lens_dev"], kf, kb, dout.contiguous(), sv["gates"], sv["c"], sv["hprev"],
                                    g[0], g[1], g[2], g[3], need_dx=d > 1, keep_prob=sv["keep"], seed=sv["seed"],
                                    join=False, kx_cat=sv["kx"], p3=p3)
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
