"""MultiRNNCell attention decoder (`-num_layers_dec L`, L > 1) -- decoder.py:49-82, attn_decoder.py:37-172.

With `num_layers_dec > 1` the reference builds BOTH decoder cells as `MultiRNNCell` stacks of L
`DropoutWrapper(BasicLSTMCell)` layers (decoder.py:66-68; `lm_cell = self.get_cell(lm_hidden_size)`, attn_decoder.py:62):
layer k's input is layer k-1's DROPPED output, the state keeps every layer's plain (c, h), and the attention query is the TOP
layer's cell state (decoder.py:77-80).  The reference's own beam search cannot read such a model (beam_search.py:56-98 maps
the single-layer names only), so this configuration is off the hot path: it runs as a host-composed per-step loop over the
library's step kernels -- `asr_lstm_cell_fwd` (fused gemv + cell + dropout), `asr_linear_fwd`, `asr_attention_fwd`,
`asr_next_token` forward; `asr_attn_cell_bwd` (attention + top-cell backward), `asr_lstm_cell_bwd` (lower layers, LM stack),
`asr_linear_wt_fwd` backward; every product over all steps (projections, logits, weight gradients) as MFMA GEMMs after the
loops.  Everything stays on the caller's stream.  Variable names: weights.multi_cell_leaf.  With lm_hidden_size !=
hidden_size_dec the LM stack's (dropped) top output passes through `rnn/SimpleProjection` before the InputProjection
(attn_decoder.py:149-151), here one `asr_linear_fwd` per step and its gradients as GEMMs over all steps (round 5).
"""
import numpy as np
import torch

from . import ops
from .devcache import dev_i32
from .weights import dec_name, multi_cell_leaf


def layer_seed(seed, stack, k):
    """Dropout stream of layer k of a stack (the mask is keep_scale(layer_seed, step * B + row, unit))."""
    base = (int(seed) * 2 + (0 if stack == "lm" else 1)) & 0xFFFFFFFF
    return (base * 2654435761 + 40503 * (k + 1)) & 0x7FFFFFFF


class MultiLayerPath(object):
    def __init__(self, decoder):
        self.dec = decoder
        p = decoder.params
        self.L = int(p.num_layers_dec)
        self.simple = p.lm_hidden_size != p.hidden_size_dec          # attn_decoder.py:149-151

    # ---- variables ---------------------------------------------------------------------------------------------
    def _name(self, leaf):
        return dec_name(self.dec.scope, leaf)

    def _w(self, leaf):
        return self.dec.variables[self._name(leaf)]

    def _g(self, leaf):
        return self.dec.variables.grad_of(self._name(leaf))

    def _out_leaf(self, leaf):
        if self.dec.params.ind_softmax:
            leaf = leaf.replace("OutputProjection", "OutputProjection2")       # attn_decoder.py:119-122
        return leaf

    def weights(self, grad=False):
        get = self._g if grad else self._w
        aw = get("AttnW")
        d = dict(emb=get("decoder/embedding"), attn_enc_w=aw.reshape(aw.shape[-2], aw.shape[-1]), attn_v=get("AttnV"),
                 attn_w=get("rnn/Attention/kernel"), attn_b=get("rnn/Attention/bias"),
                 inp_w=get("rnn/InputProjection/kernel"), inp_b=get("rnn/InputProjection/bias"),
                 ap_w=get("rnn/AttnProjection/kernel"), ap_b=get("rnn/AttnProjection/bias"),
                 out_w=get(self._out_leaf("rnn/OutputProjection/kernel")), out_b=get(self._out_leaf("rnn/OutputProjection/bias")))
        if self.simple:
            d["sp_w"], d["sp_b"] = get("rnn/SimpleProjection/kernel"), get("rnn/SimpleProjection/bias")
        for stack in ("lm", "dec"):
            d[stack + "_k"] = [get(multi_cell_leaf(stack, k, "kernel")) for k in range(self.L)]
            d[stack + "_b"] = [get(multi_cell_leaf(stack, k, "bias")) for k in range(self.L)]
        return d

    # ---- forward -----------------------------------------------------------------------------------------------
    def _stack_step(self, stack, w, x, gather, prev, keep, seed, step, top_plain):
        """One step of a cell stack.  prev: per layer (c, h) or None at step 0.  -> (top output, per-layer (c, h, hd, gates))."""
        B = (gather if gather is not None else x).shape[0]
        out, inp = [], x
        for k in range(self.L):
            kern, bias = w[stack + "_k"][k], w[stack + "_b"][k]
            Hk = kern.shape[1] // 4
            h_prev = prev[k][1] if prev is not None else torch.zeros((B, Hk), device=kern.device)
            c_prev = prev[k][0] if prev is not None else None
            kp = 1.0 if (top_plain and k == self.L - 1) else keep          # the outer stack's top output is never used
            r = ops.lstm_cell(inp, h_prev, c_prev, kern, bias, gather=gather if k == 0 else None, keep_prob=kp,
                              seed=layer_seed(seed, stack, k), step=step, save_gates=True)
            c, h = r[0], r[1]
            hd = r[2] if kp < 1.0 else h
            out.append((c, h, hd, r[-1]))
            inp = hd
        return inp, out

    def forward(self, tok, seq_len, enc, enc_len_dev, mode, coin, samp_prob, keep, seed, t_out):
        """tok int32 [T_dec,B] (device); seq_len host int64 [B].  Returns (logits [(T_out*B),V], saved)."""
        w = self.weights()
        B, Te, D = enc.shape
        T = int(t_out)
        dev = enc.device
        H, A = w["attn_w"].shape
        V = w["out_w"].shape[1]
        hf = ops.gemm(enc.reshape(B * Te, D), w["attn_enc_w"]).view(B, Te, A)
        tok = tok[:T].contiguous().clone()
        len_dev = dev_i32(seq_len, dev)

        def feedback(i):
            if i < 0 or i + 1 >= T:
                return False
            if mode == 1:
                return True
            return mode == 2 and samp_prob > 0 and not (coin[i] < 1.0 - samp_prob)
        lm_prev = dec_prev = None
        ctx_prev = torch.zeros((B, D), device=dev)
        steps = []
        logits_fb = {}
        for i in range(T):
            lm_top, lm_st = self._stack_step("lm", w, w["emb"], tok[i], lm_prev, keep, seed, i, top_plain=False)
            lm_out = ops.linear(lm_top, w["sp_w"], w["sp_b"]) if self.simple else lm_top      # attn_decoder.py:149-151
            x = ops.linear(lm_out, w["inp_w"], w["inp_b"], x2=ctx_prev)                       # attn_decoder.py:157-158
            _, dec_st = self._stack_step("dec", w, x, None, dec_prev, keep, seed, i, top_plain=True)
            q = dec_st[-1][0]                                                                 # decoder.py:77-80
            ctx, alpha = ops.attention(q, w["attn_w"], w["attn_b"], w["attn_v"], hf, enc, enc_len_dev)
            if feedback(i):          # this step's own prediction feeds step i+1: project now (attn_decoder.py:116-145)
                pr = ops.linear(q, w["ap_w"], w["ap_b"], x2=ctx)
                lg = ops.linear(pr, w["out_w"], w["out_b"], zero_from=len_dev, zero_t=i)
                tok[i + 1] = ops.next_token(lg, sample=(mode == 2), seed=seed, step=i)
                logits_fb[i] = lg
            steps.append(dict(lm=lm_st, lm_out=lm_out, lm_top=lm_top, x=x, dec=dec_st, alpha=alpha, ctx=ctx))
            lm_prev = [(s[0], s[1]) for s in lm_st]
            dec_prev = [(s[0], s[1]) for s in dec_st]
            ctx_prev = ctx
        # everything over all steps at once: p = [q | ctx].W_ap + b, logits = p.W_out + b, zero rows past each length
        Q = torch.stack([s["dec"][-1][0] for s in steps]).view(T * B, H)
        CTX = torch.stack([s["ctx"] for s in steps]).view(T * B, D)
        p = ops.gemm(Q, w["ap_w"][:H], w["ap_b"])
        ops.gemm(CTX, w["ap_w"][H:], out=p, accumulate=True)
        logits = ops.gemm(p, w["out_w"], w["out_b"])
        ops.zero_finished_rows(logits, len_dev, T, B)
        saved = dict(steps=steps, tok=tok, hf=hf, Q=Q, CTX=CTX, p=p, T=T, B=B, keep=keep, seed=seed, len_dev=len_dev,
                     enc=enc, enc_len_dev=enc_len_dev)
        return logits, saved

    # ---- backward ----------------------------------------------------------------------------------------------
    def backward(self, sv, dlogits, denc):
        """Accumulates every weight gradient into the flat gradient buffer and the encoder-state gradient into denc."""
        w, g = self.weights(), self.weights(grad=True)
        steps, T, B, keep, seed = sv["steps"], sv["T"], sv["B"], sv["keep"], sv["seed"]
        enc, enc_len_dev, hf = sv["enc"], sv["enc_len_dev"], sv["hf"]
        L = self.L
        Te, D = enc.shape[1], enc.shape[2]
        H, A = w["attn_w"].shape
        E = w["emb"].shape[1]
        lmH = w["lm_k"][0].shape[1] // 4
        P = H if self.simple else lmH                               # width of the InputProjection's first input
        dev = enc.device
        f = lambda *s: torch.zeros(s, device=dev)
        TB = T * B
        Q, CTX, p = sv["Q"], sv["CTX"], sv["p"]
        # hoisted data gradients and the weight gradients that need only them
        dP = ops.gemm(dlogits, w["out_w"], trans_b=True)                                  # [TB,H]
        dQC = ops.gemm(dP, w["ap_w"], trans_b=True)                                       # [TB,H+D] = [dq | dctx]
        ops.gemm(p, dlogits, trans_a=True, out=g["out_w"], accumulate=True)
        ops.colsum(dlogits, g["out_b"])
        ops.gemm(Q, dP, trans_a=True, out=g["ap_w"][:H], accumulate=True)
        ops.gemm(CTX, dP, trans_a=True, out=g["ap_w"][H:], accumulate=True)
        ops.colsum(dP, g["ap_b"])
        in_dec = [E] + [H] * (L - 1)
        in_lm = [E] + [lmH] * (L - 1)
        dXH = [f(T, B, in_dec[k] + H) for k in range(L)]            # [d input | dh_prev] of outer layer k
        dEH = [f(T, B, in_lm[k] + lmH) for k in range(L)]           # same, LM stack
        dLC = f(T, B, P + D)                                        # [dlm_out | dctx_prev]
        dSP = f(T, B, lmH) if self.simple else None                 # gradient of the LM stack's top output (behind SimpleProjection)
        dctx, dY = f(T, B, D), f(T, B, A)
        dhf, dv_part = f(B, Te, A), f(B, A)
        dc_dec = [f(B, H) for _ in range(L)]
        dc_lm = [f(B, lmH) for _ in range(L)]
        # per layer and step: activated gates (overwritten with dG below), c
        dg = lambda st, k, i: steps[i][st][k][3]
        cc = lambda st, k, i: steps[i][st][k][0]
        dQC3 = dQC.view(T, B, H + D)
        for i in range(T - 1, -1, -1):
            last = i == T - 1
            top = L - 1
            ops.attn_cell_bwd(cc("dec", top, i), w["attn_w"], w["attn_b"], w["attn_v"], hf, enc, enc_len_dev, steps[i]["alpha"],
                              dQC3[i], None if last else dLC[i + 1][:, P:], dhf, dctx[i], dY[i], dv_part, dg("dec", top, i),
                              cc("dec", top, i - 1) if i else None, None if last else dXH[top][i + 1][:, in_dec[top]:],
                              dc_dec[top])
            ops.linear_wt(dg("dec", top, i), w["dec_k"][top], out=dXH[top][i])
            for k in range(L - 2, -1, -1):       # lower outer layers: their dropped output fed layer k+1
                ops.lstm_cell_bwd(dg("dec", k, i), cc("dec", k, i), cc("dec", k, i - 1) if i else None,
                                  dXH[k + 1][i][:, :H], None if last else dXH[k][i + 1][:, in_dec[k]:], dc_dec[k],
                                  keep_prob=keep, seed=layer_seed(seed, "dec", k), step=i)
                ops.linear_wt(dg("dec", k, i), w["dec_k"][k], out=dXH[k][i])
            # [dlm_out | dctx_prev] = dx . W_inp^T
            ops.linear_wt(dXH[0][i], w["inp_w"], out=dLC[i], k=E)
            if self.simple:                      # d lm_top = d lm_out . W_sp^T
                ops.linear_wt(dLC[i], w["sp_w"], out=dSP[i], k=P)
            for k in range(L - 1, -1, -1):       # LM stack, top first
                dout = (dSP[i] if self.simple else dLC[i][:, :P]) if k == L - 1 else dEH[k + 1][i][:, :lmH]
                ops.lstm_cell_bwd(dg("lm", k, i), cc("lm", k, i), cc("lm", k, i - 1) if i else None, dout,
                                  None if last else dEH[k][i + 1][:, in_lm[k]:], dc_lm[k],
                                  keep_prob=keep, seed=layer_seed(seed, "lm", k), step=i)
                ops.linear_wt(dg("lm", k, i), w["lm_k"][k], out=dEH[k][i])
        # ---- encoder-state gradient: denc[b] += sum_i alpha_i[b]^T . dctx_i[b]  and through hf = enc . AttnW
        ALPHA = torch.stack([s["alpha"] for s in steps])                                   # [T,B,Te]
        ops.gemm_batched(ALPHA, dctx, denc, Te, D, T, B * Te, B * D, D, Te, D, Te * D, B, trans_a=True, accumulate=True)
        ops.gemm(dhf.view(B * Te, A), w["attn_enc_w"], trans_b=True, out=denc.view(B * Te, D), accumulate=True)
        ops.gemm(enc.reshape(B * Te, D), dhf.view(B * Te, A), trans_a=True, out=g["attn_enc_w"], accumulate=True)
        # ---- attention query projection, AttnV
        ops.gemm(Q, dY.view(TB, A), trans_a=True, out=g["attn_w"], accumulate=True)
        ops.colsum(dY.view(TB, A), g["attn_b"])
        ops.colsum(dv_part, g["attn_v"])
        # ---- cell stacks: kernel rows [input | h_prev], bias
        for stack, ins, hid in (("dec", in_dec, H), ("lm", in_lm, lmH)):
            for k in range(L):
                dG = torch.stack([steps[i][stack][k][3] for i in range(T)]).view(TB, 4 * hid)
                if k == 0:
                    inp = torch.stack([s["x"] for s in steps]).view(TB, E) if stack == "dec" else \
                        ops.gather_rows(w["emb"], sv["tok"].reshape(-1))
                else:
                    inp = torch.stack([steps[i][stack][k - 1][2] for i in range(T)]).view(TB, hid)
                gk = g[stack + "_k"][k]
                ops.gemm(inp, dG, trans_a=True, out=gk[:ins[k]], accumulate=True)
                if T > 1:
                    hprev = torch.stack([steps[i][stack][k][1] for i in range(T - 1)]).view(TB - B, hid)
                    ops.gemm(hprev, dG[B:], trans_a=True, out=gk[ins[k]:], accumulate=True)
                ops.colsum(dG, g[stack + "_b"][k])
        ops.scatter_add_rows(g["emb"], sv["tok"].reshape(-1), dEH[0].view(TB, E + lmH)[:, :E].contiguous())
        # ---- InputProjection: rows [lm_out | ctx_prev]
        dx = dXH[0].view(TB, E + H)[:, :E].contiguous()
        LMO = torch.stack([s["lm_out"] for s in steps]).view(TB, P)
        ops.gemm(LMO, dx, trans_a=True, out=g["inp_w"][:P], accumulate=True)
        if T > 1:
            ops.gemm(CTX[:TB - B], dx[B:], trans_a=True, out=g["inp_w"][P:], accumulate=True)
        ops.colsum(dx, g["inp_b"])
        if self.simple:                          # SimpleProjection: rows = the LM stack's top output, gradient = d lm_out
            dlo = dLC.view(TB, P + D)[:, :P].contiguous()
            LMT = torch.stack([s["lm_top"] for s in steps]).view(TB, lmH)
            ops.gemm(LMT, dlo, trans_a=True, out=g["sp_w"], accumulate=True)
            ops.colsum(dlo, g["sp_b"])
