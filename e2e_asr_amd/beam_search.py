"""Beam search with LM shallow fusion, batch 1 -- device counterpart of the reference's
beam_search.py (31-338).

The reference runs everything in NumPy float64, one hypothesis at a time.  Here the k live
hypotheses of a step are the rows of one small batch on the GPU: both LSTM cells, the
projections, the attention (encoder states shared by all rows) and the external LM run through
the same HIP step kernels as training.  Scoring follows the reference's definition -- log-softmax +
`log p_dec + lm_weight * log p_lm` in float64 on the device logits, top-k per hypothesis and over
the k*k continuations (:214, :300), parent = idx // k (:306), EOS shrinking the beam (:323-327),
first-max selection of the result (:336) -- and runs in one of two places:

* device-resident (default; `asr_beam_select`): selection and bookkeeping stay on the GPU, the host only enqueues
  three library calls per step and reads the back-pointers once at the end (plus a 4-byte liveness check every 8 steps);
* host (`ASR_BEAM_HOST=1`, and whenever beam_size > 16 or V > 1024): `get_top_k` as the reference structures it, one
  D2H copy of the logits per step and NumPy float64 scoring with `np.argpartition`.

Token indices equal the float64 oracle's unless two candidates tie within float32 logit resolution (~1e-6).
"""
import os
import ctypes as C

import numpy as np
import torch

from . import _lib, data_utils, ops
from .base_params import BaseParams, Bunch
from .beam_entry import BeamEntry


def _log_softmax64(x):
    x = x.astype(np.float64)
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return np.log(e / e.sum(axis=-1, keepdims=True))          # num_utils.softmax then np.log (:196-198)


class BeamSearch(BaseParams):
    @classmethod
    def class_params(cls):
        # beam_search.py:19-29
        return Bunch(beam_size=4, lm_weight=0.0, lm_path="", word_ins_penalty=0, cov_penalty=0.0)

    def __init__(self, ckpt_path, search_params=None, device="cuda:0"):
        """ckpt_path: dict name -> array, or an .npz holding TF-named variables (the reference
        reads a TF checkpoint through tf_utils.get_matching_variables, tf_utils.py:66-90)."""
        self.device = torch.device(device)
        self.search_params = self.class_params() if search_params is None else search_params
        self.dec_params = self.map_dec_variables(self.get_model_params(ckpt_path))
        sp = self.search_params
        self.use_lm = not (sp.lm_path is None or sp.lm_weight == 0.0)
        if not self.use_lm:
            print("No separate LM used")
        # the reference always loads LM params, even when lm_weight == 0 (beam_search.py:45-46)
        lm_src = sp.lm_path if sp.lm_path not in (None, "") else ckpt_path
        self.lm_params = self.map_lm_variables(self.get_model_params(lm_src))
        print("Using a beam size of %d" % sp.beam_size)

    # ---- weights -------------------------------------------------------------------
    def get_model_params(self, ckpt_path):
        """Variables whose name contains 'rnn_decoder_char', skipping optimizer slots."""
        if isinstance(ckpt_path, dict):
            arrays = ckpt_path
        else:
            arrays = dict(np.load(ckpt_path))
        return {k: np.asarray(v) for k, v in arrays.items() if "rnn_decoder_char" in k and "Adam" not in k}

    def _t(self, a):
        return None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    def map_dec_variables(self, var_dict):
        """beam_search.py:53-109 (AttnW squeezed to [D,A])."""
        pre = "model/rnn_decoder_char/"
        g = lambda leaf: var_dict[pre + leaf]
        opt = lambda leaf: var_dict.get(pre + leaf)
        aw = g("AttnW")
        p = Bunch(
            lm_lstm_w=g("rnn/basic_lstm_cell/kernel"), lm_lstm_b=g("rnn/basic_lstm_cell/bias"),
            dec_lstm_w=g("rnn/basic_lstm_cell_1/kernel"), dec_lstm_b=g("rnn/basic_lstm_cell_1/bias"),
            attn_dec_w=g("rnn/Attention/kernel"), attn_dec_b=g("rnn/Attention/bias"),
            inp_w=g("rnn/InputProjection/kernel"), inp_b=g("rnn/InputProjection/bias"),
            attn_proj_w=g("rnn/AttnProjection/kernel"), attn_proj_b=g("rnn/AttnProjection/bias"),
            out_w=g("rnn/OutputProjection/kernel"), out_b=g("rnn/OutputProjection/bias"),
            simple_w=opt("rnn/SimpleProjection/kernel"), simple_b=opt("rnn/SimpleProjection/bias"),
            attn_enc_w=aw.reshape(aw.shape[-2], aw.shape[-1]), attn_v=g("AttnV"), embedding=g("decoder/embedding"))
        total = sum(int(np.prod(v.shape)) for v in p.values() if v is not None)
        print("Total parameters in decoder (in million): %.2f" % (total / 1e6))
        return Bunch({k: self._t(v) for k, v in p.items()})

    def map_lm_variables(self, var_dict):
        """beam_search.py:111-134."""
        pre = "model/rnn_decoder_char/"
        g = lambda leaf: var_dict[pre + leaf]
        opt = lambda leaf: var_dict.get(pre + leaf)
        p = Bunch(lstm_w=g("rnn/basic_lstm_cell/kernel"), lstm_b=g("rnn/basic_lstm_cell/bias"),
                  simple_w=opt("rnn/SimpleProjection/kernel"), simple_b=opt("rnn/SimpleProjection/bias"),
                  out_w=g("rnn/OutputProjection/kernel"), out_b=g("rnn/OutputProjection/bias"),
                  embedding=g("decoder/embedding"))
        return Bunch({k: self._t(v) for k, v in p.items()})

    # ---- one step for all live hypotheses ------------------------------------------------
    def calc_attention(self, encoder_hidden_states):
        """beam_search.py:137-161: the encoder term enc . W_enc is computed once (one MFMA GEMM);
        returns a closure q[k,H] -> (ctx[k,D], alpha[k,T])."""
        enc = encoder_hidden_states
        if enc.ndim == 3:
            enc = enc[0]
        enc = torch.as_tensor(np.ascontiguousarray(enc, dtype=np.float32)).to(self.device) \
            if not torch.is_tensor(enc) else enc.to(self.device, torch.float32).contiguous()
        p = self.dec_params
        hf = ops.gemm(enc, p.attn_enc_w)
        ln = torch.tensor([enc.shape[0]], dtype=torch.int32, device=self.device)

        def attention(q):
            return ops.attention(q, p.attn_dec_w, p.attn_dec_b, p.attn_v, hf, enc, ln, shared=True)
        attention.enc = enc
        return attention

    def top_k_setup_with_lm(self, encoder_hidden_states):
        """beam_search.py:163-221, batched over the live hypotheses of the utterance: ONE library call per step
        (`asr_beam_step`: ten stream-ordered launches) on pre-allocated state buffers, one H2D copy (tokens + parent
        rows) and one D2H copy (both logit vectors) per step; scoring in float64 on the host as the reference does."""
        p, lp, sp = self.dec_params, self.lm_params, self.search_params
        attention = self.calc_attention(encoder_hidden_states)
        enc = attention.enc
        dev = self.device
        L = _lib.lib()
        kmax = max(1, int(sp.beam_size))
        Te, D = enc.shape
        H, A = p.attn_dec_w.shape
        lmH, E, V = p.lm_lstm_w.shape[1] // 4, p.embedding.shape[1], p.out_w.shape[1]
        extE, extH = lp.embedding.shape[1], lp.lstm_w.shape[1] // 4
        extP = lp.simple_w.shape[1] if lp.simple_w is not None else extH
        extV = lp.out_w.shape[1]
        if extV != V:
            raise ValueError("LM vocabulary (%d) differs from the decoder's (%d)" % (extV, V))
        f = lambda *shape: torch.zeros(shape, device=dev, dtype=torch.float32)
        widths = dict(dc=H, dh=H, dlc=lmH, dlh=lmH, lc=extH, lh=extH, ctx=D)
        sets = [dict((n, f(kmax, w)) for n, w in widths.items()) for _ in range(2)]      # [0]: step input, [1]: step output
        cst = [ops._dec_struct(_lib.BeamState, st) for st in sets]
        cw = ops._dec_struct(_lib.DecWeights, dict(
            embedding=p.embedding, attn_enc_w=p.attn_enc_w, attn_v=p.attn_v, attn_w=p.attn_dec_w, attn_b=p.attn_dec_b,
            lm_kernel=p.lm_lstm_w, lm_bias=p.lm_lstm_b, dec_kernel=p.dec_lstm_w, dec_bias=p.dec_lstm_b,
            inp_w=p.inp_w, inp_b=p.inp_b, ap_w=p.attn_proj_w, ap_b=p.attn_proj_b, out_w=p.out_w, out_b=p.out_b,
            simple_w=p.simple_w, simple_b=p.simple_b))
        clm = ops._dec_struct(_lib.LmWeights, dict(embedding=lp.embedding, lstm_kernel=lp.lstm_w, lstm_bias=lp.lstm_b,
                                                  simple_w=lp.simple_w, simple_b=lp.simple_b, out_w=lp.out_w, out_b=lp.out_b))
        clm.E, clm.H, clm.P, clm.V = extE, extH, extP, extV
        hf = ops.gemm(enc, p.attn_enc_w)
        ln = torch.tensor([Te], dtype=torch.int32, device=dev)
        scratch = torch.empty(L.asr_beam_scratch_floats(kmax, Te, H, E, extP), device=dev, dtype=torch.float32)
        logits = torch.empty((2, kmax, V), device=dev, dtype=torch.float32)
        ints = torch.zeros(2 * kmax, dtype=torch.int32, device=dev)                     # [tokens | parent rows]
        keep = (sets, hf, ln, scratch, logits, ints, enc)                                # owned by the closure

        def get_top_k(tokens, rows, beam_size=sp.beam_size):
            """tokens: last token of each live hypothesis; rows: its parent's row in the previous step's output
            (None at step 0: zero states).  -> per hypothesis (indices, model scores, scores)."""
            k = len(tokens)
            host = np.zeros(2 * kmax, np.int32)
            host[:k] = tokens
            if rows is not None:
                host[kmax:kmax + k] = rows
            ints.copy_(torch.from_numpy(host))
            if rows is None:
                for t in sets[0].values():
                    t.zero_()
            else:
                ops._check(L.asr_beam_gather(ops._stream(), ops._p(ints[kmax:]), k, C.byref(cst[1]), C.byref(cst[0]),
                                             H, lmH, extH, D), "asr_beam_gather")
            cd = _lib.DecDims(k, Te, D, A, H, lmH, E, V, 1)
            ops._check(L.asr_beam_step(ops._stream(), C.byref(cw), C.byref(clm), C.byref(cd), ops._p(hf), ops._p(enc), ops._p(ln),
                                       ops._p(ints), C.byref(cst[0]), C.byref(cst[1]), ops._p(scratch),
                                       ops._p(logits[0]), ops._p(logits[1])), "asr_beam_step")
            both = logits[:, :k].cpu().numpy()                                            # one D2H copy per step
            comb = _log_softmax64(both[0]) + sp.lm_weight * _log_softmax64(both[1])       # :208
            score = comb + 0.0                                                            # :210-212
            out = []
            for r in range(k):
                idx = np.argpartition(score[r], -beam_size)[-beam_size:]                  # :214
                out.append((idx, comb[r][idx], score[r][idx]))
            return out
        get_top_k.keep = keep
        get_top_k.structs = (cw, clm, cst, (Te, D, A, H, lmH, E, V, extH))
        return get_top_k

    def _tile_ordered_kernels(self):
        """The three LSTM kernels in the column order the step's tiles read them in (asr_lstm_kernel_tile_order), once per
        weight set: the weights are constants of this object (ASR_BEAM_TILED=0: the TF column order, as before round 3)."""
        if os.environ.get("ASR_BEAM_TILED", "1") == "0":
            return (None, None, None)
        src = (self.dec_params.lm_lstm_w, self.lm_params.lstm_w, self.dec_params.dec_lstm_w)
        # Re-tiled when a weight tensor is REPLACED or updated through torch (its _version moves).  The library's own optimizer
        # writes weights through raw pointers and moves no version counter: a BeamSearch that shares tensors with a model
        # being trained must call invalidate() after the weights change (the reference reads them once from a checkpoint,
        # beam_search.py:42-47, so they are constants there).
        key = tuple((w.data_ptr(), w._version, tuple(w.shape)) for w in src)
        if getattr(self, "_tiled", None) is None or getattr(self, "_tiled_key", None) != key:
            self._tiled_key = key
            L = _lib.lib()
            out = []
            for w in src:
                w = w.contiguous()
                t = torch.empty_like(w)
                ops._check(L.asr_lstm_kernel_tile_order(ops._stream(), ops._p(w), w.shape[0], w.shape[1] // 4, ops._p(t)),
                           "asr_lstm_kernel_tile_order")
                out.append(t)
            self._tiled = tuple(out)
        return self._tiled

    def invalidate(self):
        """Drop the cached tile-ordered weight copies: call after the weight tensors were changed behind torch's back (the
        fused clip+Adam kernel of a model that shares them)."""
        self._tiled = None
        self._tiled_key = None

    def _decode_on_device(self, get_top_k, max_steps=120):
        """The loop of beam_search.py:255-337 with scoring, selection and bookkeeping on the device."""
        sp = self.search_params
        sets, hf, ln, scratch, logits, ints, enc = get_top_k.keep
        cw, clm, cst, dims = get_top_k.structs
        Te, D, A, H, lmH, E, V, extH = dims
        kmax = int(sp.beam_size)
        dev = self.device
        L = _lib.lib()
        cum = torch.zeros(kmax, dtype=torch.float64, device=dev)
        state = torch.tensor([1, kmax, 0, 0], dtype=torch.int32, device=dev)
        bp = torch.zeros((max_steps, kmax, 2), dtype=torch.int32, device=dev)
        fin = torch.zeros((kmax, 2), dtype=torch.int32, device=dev)
        fin_score = torch.zeros(kmax, dtype=torch.float64, device=dev)
        cand = torch.zeros((kmax, 16), dtype=torch.float64, device=dev)
        cand_idx = torch.zeros((kmax, 16), dtype=torch.int32, device=dev)
        book = ops._dec_struct(_lib.BeamBook, dict(ints=ints, cum=cum, state=state, bp=bp, fin=fin, fin_score=fin_score,
                                                   cand=cand, cand_idx=cand_idx))
        host = np.zeros(2 * kmax, np.int32)
        host[0] = data_utils.GO_ID
        ints.copy_(torch.from_numpy(host))
        for t in sets[0].values():
            t.zero_()
        cd = _lib.DecDims(kmax, Te, D, A, H, lmH, E, V, 1)
        st = ops._stream()
        in_place = self.dec_params.simple_w is None and self.lm_params.simple_w is None
        tiled = self._tile_ordered_kernels() if in_place else (None, None, None)
        # ASR_BEAM_PERSIST=1: one persistent launch for the whole utterance (asr_beam_decode, csrc/beam.hip) -- bit-identical
        # to the loop below (same tile bodies; tests/test_gpu_beam.py) but SLOWER on MI355X (108 vs 80 us per token: every
        # hand-over between XCDs costs fabric round trips), so the step-by-step loop is the default
        persistent = False
        if in_place and os.environ.get("ASR_BEAM_PERSIST", "0") == "1":
            barrier = torch.zeros(1, dtype=torch.int32, device=dev)
            n_ws = int(L.asr_beam_decode_ws_floats(C.byref(cd), extH, max_steps))
            ws = getattr(get_top_k, "ring", None)
            if ws is None or ws.numel() < n_ws:
                ws = get_top_k.ring = torch.empty(n_ws, dtype=torch.float32, device=dev)
            rc = L.asr_beam_decode(st, C.byref(cw), C.byref(clm), C.byref(cd), ops._p(hf), ops._p(enc), ops._p(ln), ops._p(ws), n_ws,
                                   max_steps, data_utils.EOS_ID, float(sp.lm_weight), float(sp.word_ins_penalty), C.byref(book),
                                   ops._p(barrier), ops._p(ops._Flag.get(dev)))
            if rc != -3:                      # ASR_EUNSUPPORTED: keep the step-by-step loop
                ops._check(rc, "asr_beam_decode")
                persistent = True
        for s in range(0 if not persistent else max_steps, max_steps):
            if in_place:       # the step kernels read the parents' rows of the previous step's output in place: ping-pong
                a, b = (0, 1) if s % 2 == 0 else (1, 0)
                ops._check(L.asr_beam_step_perm(st, C.byref(cw), C.byref(clm), C.byref(cd), ops._p(hf), ops._p(enc), ops._p(ln),
                                                ops._p(ints), ops._p(ints[kmax:]) if s else None, C.byref(cst[a]), C.byref(cst[b]),
                                                ops._p(scratch), ops._p(logits[0]), ops._p(logits[1]),
                                                ops._p(tiled[0]), ops._p(tiled[1]), ops._p(tiled[2])), "asr_beam_step_perm")
            else:
                if s:
                    ops._check(L.asr_beam_gather(st, ops._p(ints[kmax:]), kmax, C.byref(cst[1]), C.byref(cst[0]), H, lmH, extH, D),
                               "asr_beam_gather")
                ops._check(L.asr_beam_step(st, C.byref(cw), C.byref(clm), C.byref(cd), ops._p(hf), ops._p(enc), ops._p(ln), ops._p(ints),
                                           C.byref(cst[0]), C.byref(cst[1]), ops._p(scratch), ops._p(logits[0]), ops._p(logits[1])),
                           "asr_beam_step")
            ops._check(L.asr_beam_select(st, ops._p(logits[0]), ops._p(logits[1]), V, kmax, max_steps, data_utils.EOS_ID,
                                         float(sp.lm_weight), float(sp.word_ins_penalty), C.byref(book)), "asr_beam_select")
            if s % 8 == 7 and int(state[1].item()) == 0:           # every hypothesis finished (:269)
                break
        n_live, _, n_fin, n_steps = [int(x) for x in state.cpu().numpy()]
        ops.check_device_flag(dev)
        bp_h, fin_h, fin_s, cum_h = bp.cpu().numpy(), fin.cpu().numpy(), fin_score.cpu().numpy(), cum.cpu().numpy()
        # what the loop left behind, for tests that compare the persistent launch with the step-by-step loop bit for bit
        self.last_book = dict(persistent=persistent, n_live=n_live, n_fin=n_fin, n_steps=n_steps, bp=bp_h[:n_steps].copy(),
                              fin=fin_h[:n_fin].copy(), fin_score=fin_s[:n_fin].copy(), cum=cum_h[:n_live].copy())

        def backtrack(step, row):                  # tokens of the hypothesis that entered step `step + 1` as row `row`
            seq = []
            while step >= 0:
                row, tok = bp_h[step, row]
                seq.append(int(tok)); step -= 1
            return seq[::-1]
        # final_output_list = finished (in finishing order) + live rows; first max wins (:334-337)
        cands = [(float(fin_s[j]), ("fin", j)) for j in range(n_fin)] + [(float(cum_h[j]), ("live", j)) for j in range(n_live)]
        best = max(range(len(cands)), key=lambda i: (cands[i][0], -i))
        kind, j = cands[best][1]
        if kind == "fin":
            step, parent = int(fin_h[j, 0]), int(fin_h[j, 1])
            seq = (backtrack(step - 1, parent) if step > 0 else []) + [data_utils.EOS_ID]
        else:
            seq = backtrack(n_steps - 1, j)
        return np.asarray(seq, dtype=np.int64)

    def __call__(self, encoder_hidden_states):
        """Beam search for batch size 1 (beam_search.py:224-338)."""
        sp = self.search_params
        get_top_k = self.top_k_setup_with_lm(encoder_hidden_states)
        V = self.dec_params.out_w.shape[1]
        if os.environ.get("ASR_BEAM_HOST", "0") != "1" and 1 <= int(sp.beam_size) <= 16 and V <= 1024:
            return self._decode_on_device(get_top_k)
        D = encoder_hidden_states.shape[-1]
        k = sp.beam_size
        output_list, final_output_list = [], []
        # step 0 from the GO symbol and zero states (:232-266)
        res = get_top_k([data_utils.GO_ID], None, beam_size=k)
        idx, mscore, _ = res[0]
        rows = []
        for i in range(idx.shape[0]):
            tup = (BeamEntry([int(idx[i])], 0, 0), mscore[i])
            if idx[i] == data_utils.EOS_ID:
                final_output_list.append(tup); k -= 1
            else:
                output_list.append(tup); rows.append(0)
        step_count = 1
        while step_count < 120 and k > 0:                                                   # :269
            res = get_top_k([c.get_last_output() for c, _ in output_list], rows, beam_size=k)   # one row per live hypothesis
            score_list = [r[2] + cs for r, (_, cs) in zip(res, output_list)]                 # :290
            model_score_list = [r[1] + cs for r, (_, cs) in zip(res, output_list)]
            all_scores = np.concatenate(score_list); all_model = np.concatenate(model_score_list)
            all_indices = np.concatenate([r[0] for r in res])
            top = np.argpartition(all_scores, -k)[-k:]                                       # :300
            nxt, tsc = all_indices[top], all_model[top]
            orig = top // k                                                                  # :306
            new_list, rows = [], []
            for j in range(k):
                oc = int(orig[j])
                seq = output_list[oc][0].get_index_seq() + [int(nxt[j])]
                tup = (BeamEntry(seq, oc, oc), tsc[j] + sp.word_ins_penalty * len(seq))        # :320-322
                if nxt[j] == data_utils.EOS_ID:
                    final_output_list.append(tup); k -= 1
                else:
                    new_list.append(tup); rows.append(oc)
            output_list = new_list
            step_count += 1
        final_output_list += output_list                                                     # :334
        best = max(final_output_list, key=lambda t: t[1])                                    # :336
        return np.stack(best[0].get_index_seq(), axis=0)

    @classmethod
    def add_parse_options(cls, parser):
        # beam_search.py:340-350
        parser.add_argument("-beam_size", default=1, type=int, help="Beam size")
        parser.add_argument("-lm_weight", default=0.0, type=float, help="LM weight in decoding")
        parser.add_argument("-lm_path", default="", type=str, help="LM ckpt path")
        parser.add_argument("-cov_penalty", default=0.0, type=float, help="Coverage penalty")
