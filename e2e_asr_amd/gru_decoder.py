"""GRUCell attention decoder (`use_lstm` False in the decoder's params) -- decoder.py:49-82, attn_decoder.py:37-172.

With `use_lstm` False the reference builds BOTH decoder cells as `DropoutWrapper(tf.nn.rnn_cell.GRUCell)` (decoder.py:56-63;
`lm_cell = self.get_cell(lm_hidden_size)`, attn_decoder.py:62) and the attention query is the GRU state itself (decoder.py:79-80:
`state.c` only for LSTMs).  No reference flag reaches this (`Decoder.class_params()` says LSTM, decoder.py:32, and main.py has no
option for it) and beam_search.py cannot read such a model, so it is off the hot path: like the MultiRNNCell decoder
(e2e_asr_amd/multi_decoder.py) it runs as a host-composed per-step loop over the library's step kernels -- `asr_gru_layer_fwd/bwd`
with T = 1 and explicit initial / final states (csrc/gru.hip), `asr_linear_fwd`, `asr_attention_fwd`, `asr_next_token` forward;
`asr_attn_bwd` (attention backward alone: the query is not an LSTM cell state), `asr_linear_wt_fwd` backward; the products over
all steps (projections, logits, their weight gradients) as MFMA GEMMs around the loops.  The GRU cells' weight gradients are
accumulated step by step inside `asr_gru_layer_bwd`.  Variables: `rnn/gru_cell/{gates,candidate}/{kernel,bias}` (LM cell, created
first) and `rnn/gru_cell_1/...` (outer cell) -- this build's reading of TF-1.x scoping, as for the stacks.  `num_layers_dec > 1`
with GRU cells is not built (NotImplementedError).
"""
import numpy as np
import torch

from . import ops
from .devcache import dev_i32
from .weights import dec_name


def step_seed(seed, stack, step):
    """Dropout stream of one step of one cell: the mask of row b, unit j is keep_scale(step_seed, b, j) (csrc/gru.hip with T = 1)."""
    base = (int(seed) * 2 + (0 if stack == "lm" else 1)) & 0xFFFFFFFF
    return (base * 2654435761 + 7919 * (int(step) + 1)) & 0x7FFFFFFF


class GruDecoderPath(object):
    def __init__(self, decoder):
        self.dec = decoder
        p = decoder.params
        if int(p.num_layers_dec) > 1:
            raise NotImplementedError("GRUCell decoder stacks (use_lstm False with num_layers_dec > 1)")
        self.simple = p.lm_hidden_size != p.hidden_size_dec          # attn_decoder.py:149-151

    def _get(self, leaf, grad):
        name = dec_name(self.dec.scope, leaf)
        return self.dec.variables.grad_of(name) if grad else self.dec.variables[name]

    def weights(self, grad=False):
        get = lambda leaf: self._get(leaf, grad)
        out_leaf = lambda leaf: leaf.replace("OutputProjection", "OutputProjection2") if self.dec.params.ind_softmax else leaf
        aw = get("AttnW")
        d = dict(emb=get("decoder/embedding"), attn_enc_w=aw.reshape(aw.shape[-2], aw.shape[-1]), attn_v=get("AttnV"),
                 attn_w=get("rnn/Attention/kernel"), attn_b=get("rnn/Attention/bias"),
                 inp_w=get("rnn/InputProjection/kernel"), inp_b=get("rnn/InputProjection/bias"),
                 ap_w=get("rnn/AttnProjection/kernel"), ap_b=get("rnn/AttnProjection/bias"),
                 out_w=get(out_leaf("rnn/OutputProjection/kernel")), out_b=get(out_leaf("rnn/OutputProjection/bias")))
        if self.simple:
            d["sp_w"], d["sp_b"] = get("rnn/SimpleProjection/kernel"), get("rnn/SimpleProjection/bias")
        for key, scope in (("lm", "rnn/gru_cell/"), ("dec", "rnn/gru_cell_1/")):
            d[key] = tuple(get(scope + leaf) for leaf in ("gates/kernel", "gates/bias", "candidate/kernel", "candidate/bias"))
        return d

    # ---- forward -----------------------------------------------------------------------------------------------
    def forward(self, tok, seq_len, enc, enc_len_dev, mode, coin, samp_prob, keep, seed, t_out):
        """tok int32 [T_dec,B] (device); seq_len host int64 [B].  Returns (logits [(T_out*B),V], saved)."""
        w = self.weights()
        B, Te, D = enc.shape
        T = int(t_out)
        dev = enc.device
        H, A = w["attn_w"].shape
        lmH = w["lm"][2].shape[1]
        E = w["emb"].shape[1]
        hf = ops.gemm(enc.reshape(B * Te, D), w["attn_enc_w"]).view(B, Te, A)
        tok = tok[:T].contiguous().clone()
        len_dev = dev_i32(seq_len, dev)
        one = ops._const("full_i32", dev, B, 1)                 # every row runs its single step (finished rows: see below)

        def feedback(i):
            if i < 0 or i + 1 >= T:
                return False
            if mode == 1:
                return True
            return mode == 2 and samp_prob > 0 and not (coin[i] < 1.0 - samp_prob)
        lm_h = dec_h = None
        ctx_prev = torch.zeros((B, D), device=dev)
        steps = []
        for i in range(T):
            e = ops.gather_rows(w["emb"], tok[i])                                             # decoder.py:97-99
            r = ops.gru_layer_fwd(e.view(B, 1, E), one, [w["lm"]], save=True, keep_prob=keep, seed=step_seed(seed, "lm", i),
                                  h0=lm_h, want_last=True)                                    # attn_decoder.py:148
            lm_top, lm_sv, lm_h = r[0].view(B, lmH), r[1:5], r[5]
            lm_out = ops.linear(lm_top, w["sp_w"], w["sp_b"]) if self.simple else lm_top      # :149-151
            x = ops.linear(lm_out, w["inp_w"], w["inp_b"], x2=ctx_prev)                       # :157-158
            # the outer cell: its (dropped) output is never used, only its state (raw_rnn keeps next_state, :111)
            r2 = ops.gru_layer_fwd(x.view(B, 1, E), one, [w["dec"]], save=True, keep_prob=1.0, h0=dec_h, want_last=True)
            dec_sv, dec_h = r2[1:5], r2[5]
            q = dec_h.view(B, H)                                                              # decoder.py:79-80: the state itself
            ctx, alpha = ops.attention(q, w["attn_w"], w["attn_b"], w["attn_v"], hf, enc, enc_len_dev)
            if feedback(i):          # this step's own prediction feeds step i+1: project now (attn_decoder.py:116-145)
                pr = ops.linear(q, w["ap_w"], w["ap_b"], x2=ctx)
                lg = ops.linear(pr, w["out_w"], w["out_b"], zero_from=len_dev, zero_t=i)
                tok[i + 1] = ops.next_token(lg, sample=(mode == 2), seed=seed, step=i)
            steps.append(dict(e=e, lm=lm_sv, lm_top=lm_top, lm_out=lm_out, x=x, dec=dec_sv, q=q, alpha=alpha, ctx=ctx))
            ctx_prev = ctx
        # (finished rows keep stepping instead of being copied through: every output of theirs from then on is zeroed below and
        # masked in the loss, and rows do not interact -- same argument as in multi_decoder.py)
        Q = torch.stack([s["q"] for s in steps]).view(T * B, H)
        CTX = torch.stack([s["ctx"] for s in steps]).view(T * B, D)
        p = ops.gemm(Q, w["ap_w"][:H], w["ap_b"])
        ops.gemm(CTX, w["ap_w"][H:], out=p, accumulate=True)
        logits = ops.gemm(p, w["out_w"], w["out_b"])
        ops.zero_finished_rows(logits, len_dev, T, B)
        saved = dict(steps=steps, tok=tok, hf=hf, Q=Q, CTX=CTX, p=p, T=T, B=B, keep=keep, seed=seed, len_dev=len_dev,
                     enc=enc, enc_len_dev=enc_len_dev, one=one)
        return logits, saved

    # ---- backward ----------------------------------------------------------------------------------------------
    def backward(self, sv, dlogits, denc):
        """Accumulates every weight gradient into the flat gradient buffer and the encoder-state gradient into denc."""
        w, g = self.weights(), self.weights(grad=True)
        steps, T, B, keep, seed, one = sv["steps"], sv["T"], sv["B"], sv["keep"], sv["seed"], sv["one"]
        enc, enc_len_dev, hf = sv["enc"], sv["enc_len_dev"], sv["hf"]
        Te, D = enc.shape[1], enc.shape[2]
        H, A = w["attn_w"].shape
        E = w["emb"].shape[1]
        lmH = w["lm"][2].shape[1]
        P = H if self.simple else lmH
        dev = enc.device
        f = lambda *s: torch.zeros(s, device=dev)
        TB = T * B
        Q, CTX, p = sv["Q"], sv["CTX"], sv["p"]
        dP = ops.gemm(dlogits, w["out_w"], trans_b=True)                                  # [TB,H]
        dQC = ops.gemm(dP, w["ap_w"], trans_b=True)                                       # [TB,H+D] = [dq | dctx]
        ops.gemm(p, dlogits, trans_a=True, out=g["out_w"], accumulate=True)
        ops.colsum(dlogits, g["out_b"])
        ops.gemm(Q, dP, trans_a=True, out=g["ap_w"][:H], accumulate=True)
        ops.gemm(CTX, dP, trans_a=True, out=g["ap_w"][H:], accumulate=True)
        ops.colsum(dP, g["ap_b"])
        dLC = f(T, B, P + D)                                        # [dlm_out | dctx_prev]
        dX = f(T, B, E)                                             # gradient of the InputProjection's output
        dE = f(T, B, E)                                             # gradient of the embedded token
        dctx, dY = f(T, B, D), f(T, B, A)
        dhf, dv_part = f(B, Te, A), f(B, A)
        dec_carry = lm_carry = None
        dQC3 = dQC.view(T, B, H + D)
        for i in range(T - 1, -1, -1):
            last = i == T - 1
            st = steps[i]
            dq = ops.attn_bwd(st["q"], w["attn_w"], w["attn_b"], w["attn_v"], hf, enc, enc_len_dev, st["alpha"], dQC3[i],
                              None if last else dLC[i + 1][:, P:], dhf, dctx[i], dY[i], dv_part)
            gx, cx, hp, rh = st["dec"]
            dx, dec_carry = ops.gru_layer_bwd(st["x"].view(B, 1, E), one, [w["dec"]], dq.view(B, 1, H), gx, cx, hp, rh, [g["dec"]],
                                              need_dx=True, keep_prob=1.0, dh_last=dec_carry, want_dh0=True)
            dX[i] = dx.view(B, E)
            ops.linear_wt(dX[i], w["inp_w"], out=dLC[i])                                  # [dlm_out | dctx_prev] = dx . W_inp^T
            dtop = ops.linear_wt(dLC[i], w["sp_w"], k=P) if self.simple else dLC[i][:, :P].contiguous()
            gx, cx, hp, rh = st["lm"]
            de, lm_carry = ops.gru_layer_bwd(st["e"].view(B, 1, E), one, [w["lm"]], dtop.view(B, 1, lmH), gx, cx, hp, rh, [g["lm"]],
                                             need_dx=True, keep_prob=keep, seed=step_seed(seed, "lm", i), dh_last=lm_carry,
                                             want_dh0=True)
            dE[i] = de.view(B, E)
        # ---- encoder-state gradient: denc[b] += sum_i alpha_i[b]^T . dctx_i[b]  and through hf = enc . AttnW
        ALPHA = torch.stack([s["alpha"] for s in steps])                                   # [T,B,Te]
        ops.gemm_batched(ALPHA, dctx, denc, Te, D, T, B * Te, B * D, D, Te, D, Te * D, B, trans_a=True, accumulate=True)
        ops.gemm(dhf.view(B * Te, A), w["attn_enc_w"], trans_b=True, out=denc.view(B * Te, D), accumulate=True)
        ops.gemm(enc.reshape(B * Te, D), dhf.view(B * Te, A), trans_a=True, out=g["attn_enc_w"], accumulate=True)
        # ---- attention query projection, AttnV
        ops.gemm(Q, dY.view(TB, A), trans_a=True, out=g["attn_w"], accumulate=True)
        ops.colsum(dY.view(TB, A), g["attn_b"])
        ops.colsum(dv_part, g["attn_v"])
        ops.scatter_add_rows(g["emb"], sv["tok"].reshape(-1), dE.view(TB, E))
        # ---- InputProjection: rows [lm_out | ctx_prev]
        dx = dX.view(TB, E)
        LMO = torch.stack([s["lm_out"] for s in steps]).view(TB, P)
        ops.gemm(LMO, dx, trans_a=True, out=g["inp_w"][:P], accumulate=True)
        if T > 1:
            ops.gemm(CTX[:TB - B], dx[B:], trans_a=True, out=g["inp_w"][P:], accumulate=True)
        ops.colsum(dx, g["inp_b"])
        if self.simple:
            dlo = dLC.view(TB, P + D)[:, :P].contiguous()
            LMT = torch.stack([s["lm_top"] for s in steps]).view(TB, lmH)
            ops.gemm(LMT, dlo, trans_a=True, out=g["sp_w"], accumulate=True)
            ops.colsum(dlo, g["sp_b"])
