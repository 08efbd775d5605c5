#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference's NumPy half.

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference,
which never travels to the GPU box):

    python oracle/gen_golden.py            # writes tests/golden/*.npz

What is executed from the reference (imported, never copied):
  * num_utils.sigmoid / num_utils.softmax            (num_utils.py:6-14)
  * basic_lstm.BasicLSTM.__call__                    (basic_lstm.py:14-23)
  * beam_search.BeamSearch.calc_attention            (beam_search.py:137-161)
  * beam_search.BeamSearch.top_k_setup_with_lm -> get_top_k (beam_search.py:163-221)
  * data_utils.get_relevant_words / swbd_utils.reverse_swbd_normalizer (data_utils.py:20-33, swbd_utils.py:7-18)
    -> tests/golden/text_utils.json   (`--text-only` regenerates just these)

beam_search.py imports three non-numeric modules that are absent or need
TensorFlow (bunch, tf_utils, data_utils); they are stubbed in sys.modules with
the minimum non-arithmetic surface (an attribute dict; the three special token
ids 0/1/2 from data_utils.py:13-15).  BeamSearch.__new__ bypasses checkpoint
reading; the parameter Bunches are set by hand.

  * beam_search.BeamSearch.__call__ -- the beam LOOP itself (beam_search.py:224-338)
    -> tests/golden/beam_loop.npz     (`--beam-only` regenerates just these)
    The loop is Python-2 / old-NumPy code in two places: `shape[1]/4` is a float under
    Python 3 and np.zeros rejects it (236-243), and `np.divide(idx, k, dtype=np.int32)`
    (306) was integer division under Python 2's NumPy.  While the reference's own
    __call__ runs, the NAME `np` inside the imported beam_search module is bound to a
    proxy that forwards everything to numpy except those two calls: `zeros` accepts an
    integral float, `divide(..., dtype=<integer>)` floor-divides.  No arithmetic of the
    reference is replaced; every other call is numpy's.
"""
import os
import sys
import types
import builtins

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _import_reference():
    sys.path.insert(0, REF)

    class Bunch(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    m = types.ModuleType("bunch"); m.Bunch = Bunch; sys.modules["bunch"] = m
    m = types.ModuleType("tf_utils"); m.get_matching_variables = lambda *a: {}
    sys.modules["tf_utils"] = m
    m = types.ModuleType("data_utils"); m.PAD_ID, m.GO_ID, m.EOS_ID = 0, 1, 2
    sys.modules["data_utils"] = m
    builtins.xrange = range
    import num_utils, basic_lstm, beam_search   # noqa: E401
    return Bunch, num_utils, basic_lstm, beam_search


def text_fixtures():
    """(6) transcript utilities, run from the reference: data_utils.get_relevant_words (data_utils.py:20-33) and
    swbd_utils.reverse_swbd_normalizer (swbd_utils.py:7-18).  Both modules `import tensorflow` at the top and use it for
    nothing on these two code paths; an EMPTY module object stands in for the name (no arithmetic, no behaviour)."""
    import importlib
    import json
    sys.path.insert(0, REF)
    saved = {k: sys.modules.pop(k, None) for k in ("data_utils", "swbd_utils", "tensorflow")}
    sys.modules["tensorflow"] = types.ModuleType("tensorflow")
    try:
        data_utils = importlib.import_module("data_utils")
        swbd_utils = importlib.import_module("swbd_utils")
        sents = [
            "", "   ", "hello world", "uh i think um it is [noise] fine", "i wa- i was th- there", "-", "a-b c- -d",
            "yeah<sp>right<sp>uh<sp>huh", "<sp><sp>", "uh-huh mm-hm", "[laughter] [vocalized-noise] ew eee ach hee oof er ha",
            "UH Um", "it's  double  spaced", "tab\tseparated\nnewline", "ends with dash-", "hm hmm", "ah aha",
            "! that was @ funny #", "!!", "no tags here", "a!b@c#d", "[laughter] ! [noise]", "# # #", "100% #1 fan",
        ]
        norm = swbd_utils.reverse_swbd_normalizer()
        out = {"get_relevant_words": [], "reverse_swbd_normalizer": [], "normalize_then_filter": [],
               "ignored_words": list(data_utils.IGNORED_WORDS),
               "ids": [int(data_utils.PAD_ID), int(data_utils.GO_ID), int(data_utils.EOS_ID)]}
        for t in sents:
            words, rel = data_utils.get_relevant_words(t)
            out["get_relevant_words"].append({"in": t, "words": list(words), "rel_words": list(rel)})
            out["reverse_swbd_normalizer"].append({"in": t, "out": norm(t)})
            words, rel = data_utils.get_relevant_words(norm(t))           # eval_model.py:91-93 composes them this way
            out["normalize_then_filter"].append({"in": t, "words": list(words), "rel_words": list(rel)})
        with open(os.path.join(OUT, "text_utils.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
    finally:
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


class _Py2NumpyProxy(object):
    """Stands for the name `np` inside the imported beam_search module: numpy, with the two Python-2-era behaviours
    the beam loop relies on (see the module docstring)."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def zeros(shape, *a, **kw):
        if isinstance(shape, float):
            assert shape == int(shape)
            shape = int(shape)
        return np.zeros(shape, *a, **kw)

    @staticmethod
    def divide(x, y, *a, **kw):
        dt = kw.get("dtype")
        if dt is not None and np.issubdtype(np.dtype(dt), np.integer):
            return np.floor_divide(x, y).astype(dt)
        return np.divide(x, y, *a, **kw)


BEAM_LOOP_CASES = [
    # (variant, enc key or recipe, beam k, lm_weight, word_ins_penalty, EOS bias added to OutputProjection/bias[2])
    ("plain", "enc_T100", 4, 0.0, 0.0, 0.0),
    ("plain", "enc_T100", 8, 0.0, -0.2, 0.0),
    ("plain", "enc_T100", 16, 0.1, 0.0, 0.0),
    ("plain", "enc_T100", 4, 0.1, 0.3, 0.0),
    ("plain", "enc_T100", 16, 0.1, 1.5, 0.0),
    ("simple", "enc_T100", 8, 0.1, 0.0, 0.0),
    ("simple", "enc_T7", 4, 0.0, 0.3, 0.0),
    # hypotheses that FINISH (EOS made likelier through its output bias), so k shrinks (beam_search.py:323-327)
    ("plain", "enc_T100", 8, 0.1, 0.3, 1.0),       # 7 of 8 finish within two steps, the survivor runs to 120 and wins
    ("plain", "enc_T100", 16, 0.0, 1.5, 1.0),      # 12 finish
    ("plain", "enc_T100", 16, 0.1, 0.3, 1.0),      # 10 finish
    ("plain", "enc_T7", 4, 0.0, 0.3, 1.5),
    ("plain", "enc_T7", 16, 0.1, 1.5, 1.5),        # 13 finish
    ("simple", "enc_T7", 8, 0.0, 1.5, 1.0),        # all 8 finish at lengths 1..5: loop ends on k == 0, best ends in EOS
    ("simple", "enc_T7", 8, 0.1, 1.5, 1.0),
    ("simple", "enc_T7", 16, 0.0, 1.5, 1.0),       # all 16 finish
    ("simple", "enc_T7", 16, 0.1, 1.5, 1.0),
    ("plain", "enc_T100", 4, 0.0, 0.0, 3.0),       # EOS wins step 0: a one-token result
    ("simple", "enc_T2", 8, 0.0, -0.2, 1.0),
]


def beam_loop_weights(g, eos_bias):
    """Weight dicts of a beam-loop case: the committed decoder_step_<variant>.npz sets, EOS output bias raised."""
    wd = {k[len("w_dec/"):]: np.array(g[k]) for k in g.files if k.startswith("w_dec/")}
    wl = {k[len("w_lm/"):]: np.array(g[k]) for k in g.files if k.startswith("w_lm/")}
    wd["model/rnn_decoder_char/rnn/OutputProjection/bias"][2] += np.float32(eos_bias)
    return wd, wl


def beam_loop_fixtures():
    """(7) the reference's OWN BeamSearch.__call__ (beam_search.py:224-338) on the committed decoder_step weight sets:
    token ids of the best hypothesis, its score, and how many hypotheses finished (the beam shrinks once per EOS)."""
    Bunch, num_utils, basic_lstm, beam_search = _import_reference()
    beam_search.np = _Py2NumpyProxy()
    finished = []
    out = {}
    for ci, (variant, enc_key, k, lm_weight, wip, eos_bias) in enumerate(BEAM_LOOP_CASES):
        g = np.load(os.path.join(OUT, "decoder_step_%s.npz" % variant))
        wd, wl = beam_loop_weights(g, eos_bias)
        bs = beam_search.BeamSearch.__new__(beam_search.BeamSearch)
        bs.dec_params = bs.map_dec_variables(wd)
        bs.lm_params = bs.map_lm_variables(wl)
        bs.search_params = Bunch(beam_size=k, lm_weight=lm_weight, lm_path="", word_ins_penalty=wip, cov_penalty=0.0)
        ids = np.asarray(bs(g[enc_key])).astype(np.int64)
        tag = "case%02d_" % ci
        out[tag + "ids"] = ids
        out[tag + "variant"] = np.array(variant); out[tag + "enc_key"] = np.array(enc_key)
        out[tag + "k"] = np.int64(k); out[tag + "lm_weight"] = np.float64(lm_weight)
        out[tag + "word_ins_penalty"] = np.float64(wip); out[tag + "eos_bias"] = np.float64(eos_bias)
        finished.append((ci, len(ids), int(ids[-1]) == 2))
    out["n_cases"] = np.int64(len(BEAM_LOOP_CASES))
    np.savez_compressed(os.path.join(OUT, "beam_loop.npz"), **out)
    print("beam loop fixtures: (case, length, ends in EOS) =", finished)


def make_decoder_weights(rng, E, H, lmH, D, A, V, simple):
    """Random float32 decoder weights keyed by TF variable name."""
    u = lambda *s: rng.uniform(-0.3, 0.3, s).astype(np.float32)
    pre = "model/rnn_decoder_char/"
    w = {
        pre + "rnn/basic_lstm_cell/kernel": u(E + lmH, 4 * lmH),
        pre + "rnn/basic_lstm_cell/bias": u(4 * lmH),
        pre + "rnn/basic_lstm_cell_1/kernel": u(E + H, 4 * H),
        pre + "rnn/basic_lstm_cell_1/bias": u(4 * H),
        pre + "rnn/Attention/kernel": u(H, A), pre + "rnn/Attention/bias": u(A),
        pre + "rnn/InputProjection/kernel": u((H if simple else lmH) + D, E),
        pre + "rnn/InputProjection/bias": u(E),
        pre + "rnn/AttnProjection/kernel": u(H + D, H), pre + "rnn/AttnProjection/bias": u(H),
        pre + "rnn/OutputProjection/kernel": u(H, V), pre + "rnn/OutputProjection/bias": u(V),
        pre + "AttnW": u(1, 1, D, A), pre + "AttnV": u(A),
        pre + "decoder/embedding": rng.uniform(-1, 1, (V, E)).astype(np.float32),
    }
    if simple:
        w[pre + "rnn/SimpleProjection/kernel"] = u(lmH, H)
        w[pre + "rnn/SimpleProjection/bias"] = u(H)
    return w


def main():
    os.makedirs(OUT, exist_ok=True)
    text_fixtures()
    if "--text-only" in sys.argv:
        print("text fixtures written to", os.path.abspath(OUT))
        return
    if "--beam-only" in sys.argv:
        beam_loop_fixtures()
        return
    Bunch, num_utils, basic_lstm, beam_search = _import_reference()
    rng = np.random.default_rng(20180201)

    # (1) BasicLSTM single steps -------------------------------------------
    d = {}
    for tag, (E, H) in {"e40h128": (40, 128), "e256h256": (256, 256)}.items():
        for dt in ((np.float64, np.float32) if H == 128 else (np.float32,)):  # keep fixtures small
            w = rng.uniform(-0.1, 0.1, (E + H, 4 * H)).astype(dt)
            b = rng.uniform(-0.1, 0.1, (4 * H,)).astype(dt)
            x = rng.standard_normal(E).astype(dt)
            c = rng.standard_normal(H).astype(dt)
            h = np.tanh(rng.standard_normal(H)).astype(dt)
            nc, nh = basic_lstm.BasicLSTM(w, b)(x, (c, h))
            k = "%s_%s_" % (tag, np.dtype(dt).name)
            d.update({k + "w": w, k + "b": b, k + "x": x, k + "c": c, k + "h": h,
                      k + "new_c": nc, k + "new_h": nh})
    np.savez_compressed(os.path.join(OUT, "basic_lstm.npz"), **d)

    # (2) sigmoid / softmax incl. large magnitudes --------------------------
    xs = np.array([-745.0, -100.0, -20.0, -1.0, -1e-8, 0.0, 1e-8, 1.0, 20.0, 100.0, 700.0])
    sm_in = [np.array([1.0, 2.0, 3.0]), np.array([1000.0, 1000.0, 999.0]),
             np.array([-1000.0, -1001.0, 0.0]), rng.standard_normal(1000) * 5]
    with np.errstate(over="ignore"):
        d = {"sig_x": xs, "sig_y": num_utils.sigmoid(xs)}
    for i, v in enumerate(sm_in):
        d["sm_x%d" % i] = v; d["sm_y%d" % i] = num_utils.softmax(v)
    np.savez_compressed(os.path.join(OUT, "num_utils.npz"), **d)

    # (3)+(4)+(5) decoder step, via BeamSearch methods ----------------------
    E, H, lmH, D, A, V = 24, 32, 32, 48, 16, 37
    for simple in (False, True):
        lmH_ = 20 if simple else lmH
        wdec = make_decoder_weights(rng, E, H, lmH_, D, A, V, simple)
        wlm = make_decoder_weights(rng, E, H, lmH_, D, A, V, simple)   # separate LM weight set
        bs = beam_search.BeamSearch.__new__(beam_search.BeamSearch)
        bs.dec_params = bs.map_dec_variables(wdec)
        bs.lm_params = bs.map_lm_variables(wlm)
        out = {("w_dec/" + k): v for k, v in wdec.items()}
        out.update({("w_lm/" + k): v for k, v in wlm.items()})
        # T=1 raises inside the reference (np.squeeze at beam_search.py:154 yields a 0-d
        # array that np.matmul rejects at :157), so the smallest pinned case is T=2.
        for T in (2, 7, 100):
            enc = (rng.standard_normal((T, D)) * 0.5).astype(np.float32)
            out["enc_T%d" % T] = enc
            q = rng.standard_normal(H)
            ctx, alpha = bs.calc_attention(enc)(q)
            out["attn_T%d_q" % T] = q; out["attn_T%d_ctx" % T] = ctx
            out["attn_T%d_alpha" % T] = alpha
        enc = out["enc_T100"]
        for lm_weight in (0.0, 0.1):
            for k in (1, 4, 16):
                bs.search_params = Bunch(beam_size=k, lm_weight=lm_weight, lm_path="",
                                         word_ins_penalty=0, cov_penalty=0.0)
                step = bs.top_k_setup_with_lm(enc)
                x = rng.uniform(-1, 1, E); x_lm = rng.uniform(-1, 1, E)
                st = [(rng.standard_normal(H), np.tanh(rng.standard_normal(H))),
                      (rng.standard_normal(lmH_), np.tanh(rng.standard_normal(lmH_))),
                      (rng.standard_normal(lmH_), np.tanh(rng.standard_normal(lmH_)))]
                cx = rng.standard_normal(D) * 0.3
                idx, ms, sc, nst, ncx = step(x, x_lm, st, cx, beam_size=k)
                order = np.argsort(idx)
                tag = "step_lm%g_k%d_" % (lm_weight, k)
                out.update({tag + "x": x, tag + "x_lm": x_lm, tag + "ctx_in": cx,
                            tag + "idx": idx[order], tag + "model_score": ms[order],
                            tag + "score": sc[order], tag + "ctx_out": ncx})
                for si, nm in enumerate(("dec", "declm", "lm")):
                    out[tag + nm + "_c_in"] = st[si][0]; out[tag + nm + "_h_in"] = st[si][1]
                    out[tag + nm + "_c_out"] = nst[si][0]; out[tag + nm + "_h_out"] = nst[si][1]
            # (5) greedy chain: reference get_top_k with k=1, argmax feedback
            bs.search_params = Bunch(beam_size=1, lm_weight=lm_weight, lm_path="",
                                     word_ins_penalty=0, cov_penalty=0.0)
            step = bs.top_k_setup_with_lm(enc)
            emb, emb_lm = bs.dec_params.embedding, bs.lm_params.embedding
            x, x_lm = emb[1], emb_lm[1]
            st = [(np.zeros(H), np.zeros(H)), (np.zeros(lmH_), np.zeros(lmH_)),
                  (np.zeros(lmH_), np.zeros(lmH_))]
            cx = np.zeros(D)
            toks, scs = [], []
            for _ in range(30):
                idx, ms, sc, st, cx = step(x, x_lm, st, cx, beam_size=1)
                toks.append(int(idx[0])); scs.append(float(ms[0]))
                if idx[0] == 2:
                    break
                x, x_lm = emb[idx[0]], emb_lm[idx[0]]
            out["greedy_lm%g_tokens" % lm_weight] = np.asarray(toks)
            out["greedy_lm%g_scores" % lm_weight] = np.asarray(scs)
        np.savez_compressed(
            os.path.join(OUT, "decoder_step_%s.npz" % ("simple" if simple else "plain")), **out)
    beam_loop_fixtures()
    print("golden vectors written to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
