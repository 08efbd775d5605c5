"""Pin the oracle against golden vectors produced by the REFERENCE's own NumPy
code (oracle/gen_golden.py ran basic_lstm.py, num_utils.py and beam_search.py's
calc_attention / get_top_k in the build container)."""
import os

import numpy as np
import pytest

from oracle import asr_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_sigmoid_softmax_golden(golden_dir):
    g = _load(golden_dir, "num_utils.npz")
    with np.errstate(over="ignore"):
        np.testing.assert_array_equal(O.sigmoid(g["sig_x"]), g["sig_y"])
    for i in range(4):
        np.testing.assert_array_equal(O.softmax(g["sm_x%d" % i]), g["sm_y%d" % i])


@pytest.mark.parametrize("tag", ["e40h128_float64", "e40h128_float32", "e256h256_float32"])
def test_lstm_cell_golden(golden_dir, tag):
    g = _load(golden_dir, "basic_lstm.npz")
    k = tag + "_"
    nc, nh = O.lstm_cell(g[k + "x"], g[k + "c"], g[k + "h"], g[k + "w"], g[k + "b"])
    # same NumPy, same op order -> bit-exact
    np.testing.assert_array_equal(nc, g[k + "new_c"])
    np.testing.assert_array_equal(nh, g[k + "new_h"])
    assert nc.dtype == g[k + "new_c"].dtype


def _weights(g, prefix):
    return {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}


@pytest.mark.parametrize("variant", ["plain", "simple"])
def test_calc_attention_golden(golden_dir, variant):
    g = _load(golden_dir, "decoder_step_%s.npz" % variant)
    p = O.decoder_weights(_weights(g, "w_dec/"))
    for T in (2, 7, 100):
        ctx, alpha = O.calc_attention(g["enc_T%d" % T], p)(g["attn_T%d_q" % T])
        np.testing.assert_array_equal(ctx, g["attn_T%d_ctx" % T])
        np.testing.assert_array_equal(alpha, g["attn_T%d_alpha" % T])


@pytest.mark.parametrize("variant", ["plain", "simple"])
@pytest.mark.parametrize("lm_weight", [0.0, 0.1])
@pytest.mark.parametrize("k", [1, 4, 16])
def test_decoder_step_golden(golden_dir, variant, lm_weight, k):
    g = _load(golden_dir, "decoder_step_%s.npz" % variant)
    p = O.decoder_weights(_weights(g, "w_dec/"))
    lmp = O.lm_weights(_weights(g, "w_lm/"))
    att = O.calc_attention(g["enc_T100"], p)
    t = "step_lm%g_k%d_" % (lm_weight, k)
    st = [(g[t + n + "_c_in"], g[t + n + "_h_in"]) for n in ("dec", "declm", "lm")]
    idx, ms, sc, nst, ctx, _ = O.decoder_step(g[t + "x"], g[t + "x_lm"], st, g[t + "ctx_in"],
                                              p, lmp, att, lm_weight, k)
    order = np.argsort(idx)
    np.testing.assert_array_equal(idx[order], g[t + "idx"])
    np.testing.assert_array_equal(ms[order], g[t + "model_score"])
    np.testing.assert_array_equal(sc[order], g[t + "score"])
    np.testing.assert_array_equal(ctx, g[t + "ctx_out"])
    for si, n in enumerate(("dec", "declm", "lm")):
        np.testing.assert_array_equal(nst[si][0], g[t + n + "_c_out"])
        np.testing.assert_array_equal(nst[si][1], g[t + n + "_h_out"])


@pytest.mark.parametrize("variant", ["plain", "simple"])
@pytest.mark.parametrize("lm_weight", [0.0, 0.1])
def test_beam1_equals_reference_greedy_chain(golden_dir, variant, lm_weight):
    """The restated beam LOOP (beam_search.py:224-338) at k=1 must reproduce the
    greedy chain that gen_golden.py drove with the reference's own get_top_k."""
    g = _load(golden_dir, "decoder_step_%s.npz" % variant)
    wd, wl = _weights(g, "w_dec/"), _weights(g, "w_lm/")
    toks = g["greedy_lm%g_tokens" % lm_weight]
    out, allh = O.beam_search(g["enc_T100"], wd, wl, beam_size=1, lm_weight=lm_weight,
                              max_steps=len(toks), return_all=True)
    np.testing.assert_array_equal(out, toks)
    np.testing.assert_allclose(allh[0][1], g["greedy_lm%g_scores" % lm_weight].sum(), rtol=1e-12)


@pytest.mark.parametrize("variant", ["plain", "simple"])
def test_tf_decoder_eval_equals_reference_greedy_chain(golden_dir, variant):
    """The restated TF-graph decoder (attn_decoder.py:37-172) in eval mode, batch 1,
    full-length encoder, must emit the same argmax tokens as the reference's NumPy
    step chain (main.py:217-223 treats the two as interchangeable at beam 1)."""
    g = _load(golden_dir, "decoder_step_%s.npz" % variant)
    wd = {k: v.astype(np.float64) for k, v in _weights(g, "w_dec/").items()}
    toks = g["greedy_lm0_tokens"]
    enc = g["enc_T100"].astype(np.float64)[None]
    n = len(toks)
    dec_inp = np.ones((n + 1, 1), np.int64)
    logits = O.attn_decoder(dec_inp, [n], enc, [100], wd, is_training=False)
    ids = O.greedy_decode_ids(logits, 1)[0]
    np.testing.assert_array_equal(ids, toks)
    # and the per-step log-prob of the chosen token matches the reference's model score
    lp = logits - np.log(np.exp(logits - logits.max(1, keepdims=True)).sum(1, keepdims=True)) \
        - logits.max(1, keepdims=True)
    np.testing.assert_allclose(lp[np.arange(n), toks], g["greedy_lm0_scores"], rtol=0, atol=1e-6)  # ref does enc.W_enc in f32


def _beam_cases():
    from conftest import beam_loop_cases
    return [pytest.param(c, id=i) for i, c in beam_loop_cases()]


@pytest.mark.parametrize("case", _beam_cases())
def test_beam_loop_equals_reference_beam_search_call(case):
    """The restated beam LOOP at k > 1 against token ids produced by the reference's own BeamSearch.__call__
    (beam_search.py:224-338; oracle/gen_golden.py `beam_loop_fixtures`): word-insertion penalties of both signs, LM
    fusion, and cases where hypotheses finish and the beam shrinks (down to k == 0)."""
    got = O.beam_search(case["enc"], case["wd"], case["wl"], beam_size=case["k"], lm_weight=case["lm_weight"],
                        word_ins_penalty=case["word_ins_penalty"])
    np.testing.assert_array_equal(got, case["ids"])


def test_beam_loop_fixtures_cover_finishing_hypotheses():
    from conftest import beam_loop_cases
    cases = [c for _, c in beam_loop_cases()]
    assert len(cases) >= 15
    assert any(c["ids"][-1] == 2 and 1 < len(c["ids"]) < 120 for c in cases)       # a finished hypothesis wins
    assert {c["k"] for c in cases} >= {4, 8, 16}
    assert {c["word_ins_penalty"] for c in cases} >= {0.0, 0.3, -0.2, 1.5}
