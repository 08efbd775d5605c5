"""CPU oracle for the e2e_asr hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``e2e_asr_amd``) never does: it
fails loudly when the HIP library is missing.

What this is: a NumPy restatement, op for op, of the reference's algorithm
(shtoshni/e2e_asr).  Every function cites the reference ``file:line`` it
follows.  The reference has two halves:

* a NumPy half (``basic_lstm.py``, ``num_utils.py``, ``beam_search.py``) that
  is importable in the build container.  ``oracle/gen_golden.py`` runs it and
  commits input/output vectors under ``tests/golden/``; the functions
  ``sigmoid``, ``softmax``, ``lstm_cell``, ``calc_attention``,
  ``decoder_step`` below are PINNED by those vectors
  (``tests/test_oracle_golden.py``).
* a TensorFlow-1.x graph half (``encoder.py``, ``attn_decoder.py``,
  ``losses.py``, ``seq2seq_model.py``).  TensorFlow is absent here (ordinary
  ModuleNotFoundError, no network), and the reference ships no test that pins
  results at the TF boundary, so ``lstm_layer``, ``bilstm_layer``,
  ``pyramid``, ``encoder``, ``attn_decoder``, ``cross_entropy_loss``,
  ``clip_by_global_norm`` and ``adam_step`` are **parity unpinned** against
  TF itself.  They are pinned indirectly: the decoder loop is built from the
  golden-pinned step (greedy transcript fixtures), the cell from the
  golden-pinned cell, and the TF library semantics assumed (dynamic_rnn
  zero-output/copy-through, reverse_sequence for the backward direction,
  raw_rnn emit/copy-through order, TF Adam epsilon placement) are the
  published TF-1.x ones, listed in DESIGN.md.

All functions are dtype-generic: they compute in the dtype of their inputs
(float32 to mimic the TF graph, float64 to mimic beam_search.py, which
promotes everything to float64 through its ``np.zeros`` states).
"""
from __future__ import annotations

import numpy as np

PAD_ID, GO_ID, EOS_ID = 0, 1, 2  # data_utils.py:13-15


# ----------------------------------------------------------------------------
# num_utils.py
# ----------------------------------------------------------------------------
def sigmoid(x):
    """num_utils.py:6-8 -- 1/(1+exp(-x))."""
    return 1 / (1 + np.exp(-x))


def softmax(x, axis=0):
    """num_utils.py:11-14 -- max-subtracted softmax (reference: axis 0 of a 1-D
    vector, max over the whole array; identical for 1-D input)."""
    e_x = np.exp(x - np.max(x, axis=axis, keepdims=True))
    return e_x / e_x.sum(axis=axis, keepdims=True)


# ----------------------------------------------------------------------------
# basic_lstm.py  (== tf.nn.rnn_cell.BasicLSTMCell, forget_bias=1.0)
# ----------------------------------------------------------------------------
def lstm_cell(x, c, h, w, b):
    """basic_lstm.py:14-23.  x [..., E], c/h [..., H], w [E+H, 4H], b [4H].

    Gate order i, j, f, o; forget bias +1 added at run time; returns (c', h').
    Batched over leading dims (the reference is 1-D; matmul broadcasts).
    """
    xh = np.concatenate((x, h), axis=-1)
    i, j, f, o = np.split(np.matmul(xh, w) + b, 4, axis=-1)
    new_c = c * sigmoid(f + 1) + sigmoid(i) * np.tanh(j)
    new_h = sigmoid(o) * np.tanh(new_c)
    return new_c, new_h


# ----------------------------------------------------------------------------
# tf.nn.rnn_cell.GRUCell (TF 1.x) -- the cell encoder.py:45-48 / decoder.py:56-59 build when use_lstm is False.  NOT in the
# reference's NumPy half and absent from /root/reference: restated from the published TF-1.x implementation
# (rnn_cell_impl.GRUCell.call) -- PARITY UNPINNED, like the rest of the TF-graph half.
#   gate_inputs = [x, h] . W_gates + b_gates (b_gates initialised to 1.0);  r, u = split(sigmoid(gate_inputs), 2)
#   candidate   = [x, r * h] . W_cand + b_cand;  c = tanh(candidate);  h' = u * h + (1 - u) * c
# ----------------------------------------------------------------------------
def gru_cell(x, h, wg, bg, wc, bc):
    """x [..., E], h [..., H], wg [E+H, 2H], bg [2H], wc [E+H, H], bc [H] -> h'."""
    v = sigmoid(np.matmul(np.concatenate((x, h), axis=-1), wg) + bg)
    r, u = np.split(v, 2, axis=-1)
    c = np.tanh(np.matmul(np.concatenate((x, r * h), axis=-1), wc) + bc)
    return u * h + (1 - u) * c


def gru_layer(x_tm, seq_len, wg, bg, wc, bc, reverse=False, keep_mask=None):
    """One direction of encoder.py:55-91 with GRUCell: the same dynamic_rnn semantics as lstm_layer (zero output and
    copied-through state past each length, the bw half walks t = len-1 .. 0, output-only dropout)."""
    T, B, _ = x_tm.shape
    H = wc.shape[1]
    out = np.zeros((T, B, H), x_tm.dtype)
    h = np.zeros((B, H), x_tm.dtype)
    seq_len = np.asarray(seq_len).astype(np.int64)
    for s in range(T):
        t_idx = (seq_len - 1 - s) if reverse else np.full((B,), s, np.int64)
        live = s < seq_len
        if not live.any():
            break
        t_safe = np.where(live, t_idx, 0)
        nh = gru_cell(x_tm[t_safe, np.arange(B)], h, wg, bg, wc, bc)
        h = np.where(live[:, None], nh, h)
        rows = np.nonzero(live)[0]
        out[t_idx[rows], rows] = nh[rows]
    if keep_mask is not None:
        out = out * keep_mask
    return out, h


def enc_gru_var(depth, direction, part, leaf, bi_dir=True):
    """TF variable names of a GRUCell encoder layer: .../gru_cell/{gates,candidate}/{kernel,bias}."""
    return enc_var(depth, direction, "%s/%s" % (part, leaf), bi_dir).replace("basic_lstm_cell", "gru_cell")


# ----------------------------------------------------------------------------
# encoder.py
# ----------------------------------------------------------------------------
def lstm_layer(x_tm, seq_len, w, b, reverse=False, keep_mask=None):
    """One direction of encoder.py:55-91 (tf.nn.dynamic_rnn, time_major=True,
    sequence_length=seq_len, zero initial state).

    x_tm [T, B, in]; seq_len [B] ints.  TF semantics restated:
      * t >= len[b]: emitted output is 0, state is copied through unchanged;
      * reverse=True is the bw half of bidirectional_dynamic_rnn:
        reverse_sequence(x, len) -> rnn -> reverse_sequence(out, len), i.e. the
        cell walks t = len[b]-1 ... 0 and its output lands back at index t.
      * keep_mask (optional [T,B,H] of 0 or 1/keep_prob) is DropoutWrapper's
        output-only dropout (encoder.py:49-52): it scales the emitted h, never
        the recurrent state.
    Returns out [T, B, H] and the final (c, h).
    """
    T, B, _ = x_tm.shape
    H = w.shape[1] // 4
    dt = x_tm.dtype
    out = np.zeros((T, B, H), dt)
    c = np.zeros((B, H), dt)
    h = np.zeros((B, H), dt)
    seq_len = np.asarray(seq_len).astype(np.int64)
    for s in range(T):
        if reverse:
            t_idx = seq_len - 1 - s          # per-utterance time index
        else:
            t_idx = np.full((B,), s, np.int64)
        live = s < seq_len                   # [B]
        if not live.any():
            break
        t_safe = np.where(live, t_idx, 0)
        x_t = x_tm[t_safe, np.arange(B)]     # [B, in]
        nc, nh = lstm_cell(x_t, c, h, w, b)
        lv = live[:, None]
        c = np.where(lv, nc, c)
        h = np.where(lv, nh, h)
        rows = np.nonzero(live)[0]
        out[t_idx[rows], rows] = nh[rows]
    if keep_mask is not None:
        out = out * keep_mask
    return out, (c, h)


def bilstm_layer(x_tm, seq_len, w_fw, b_fw, w_bw, b_bw, keep_fw=None, keep_bw=None):
    """encoder.py:76-84 -- fw and bw halves concatenated on the feature axis."""
    o_fw, _ = lstm_layer(x_tm, seq_len, w_fw, b_fw, False, keep_fw)
    o_bw, _ = lstm_layer(x_tm, seq_len, w_bw, b_bw, True, keep_bw)
    return np.concatenate((o_fw, o_bw), axis=2)


def pyramid(x_bm, seq_len, skip_step=2):
    """encoder.py:94-119.  x_bm [B, T, F] batch-major.

    If max(seq_len) % skip != 0, pad (skip - rem) zero frames (104-110);
    reshape [B,T,F] -> [B,T/skip,F*skip] (112-115); len <- ceil(len/skip)
    (117-118).  tf.reshape raises when the padded T is not divisible; so do we.
    """
    seq_len = np.asarray(seq_len).astype(np.int64)
    B, T, F = x_bm.shape
    rem = int(seq_len.max()) % skip_step
    if rem:
        x_bm = np.concatenate(
            (x_bm, np.zeros((B, skip_step - rem, F), x_bm.dtype)), axis=1)
    Tp = x_bm.shape[1]
    if Tp % skip_step:
        raise ValueError("pyramid: padded T=%d not divisible by %d" % (Tp, skip_step))
    out = x_bm.reshape(B, Tp // skip_step, F * skip_step)
    new_len = np.ceil(seq_len / float(skip_step)).astype(np.int64)
    return out, new_len


def enc_var(depth, direction, leaf, bi_dir=True):
    """TF variable names of encoder.py:73-89 under train.py:184's scope 'model'."""
    if bi_dir:
        return "model/encoder/RNNLayer%d/bidirectional_rnn/%s/basic_lstm_cell/%s" % (
            depth, direction, leaf)
    return "model/encoder/RNNLayer%d/%d/basic_lstm_cell/%s" % (depth, depth, leaf)


def encoder(x_bm, seq_len, weights, num_layers, bi_dir=True, skip_step=2,
            initial_res_fac=1, max_scaling_down=8, keep_masks=None):
    """encoder.py:122-180.  x_bm [B,T,F]; num_layers {task: depth}.

    Returns (attention_states{depth:[B,T_d,D]}, time_major_states{depth},
    seq_len_inps{depth}).  keep_masks: optional {depth: (fw[T,B,H], bw[T,B,H])}.
    """
    attention_states, time_major_states, seq_len_inps = {}, {}, {}
    max_depth = 0
    for task, nl in num_layers.items():
        if task == "state":
            time_major_states[nl] = None
        else:
            attention_states[nl] = None
        max_depth = max(max_depth, nl)
    seq_len = np.asarray(seq_len).astype(np.int64)
    res = initial_res_fac
    if res > 1:                                            # encoder.py:150-153
        x_bm = x_bm[:, ::res, :]
        seq_len = np.ceil(seq_len / float(res)).astype(np.int64)
    enc_in = x_bm
    for i in range(max_depth):
        d = i + 1
        x_tm = np.transpose(enc_in, (1, 0, 2))             # encoder.py:158
        km = keep_masks.get(d) if keep_masks else None
        if enc_gru_var(d, "fw" if bi_dir else "", "gates", "kernel", bi_dir) in weights:      # use_lstm False (encoder.py:45-48)
            dirs = ("fw", "bw") if bi_dir else ("",)
            halves = []
            for k, dr in enumerate(dirs):
                g = lambda part, leaf: weights[enc_gru_var(d, dr, part, leaf, bi_dir)]
                halves.append(gru_layer(x_tm, seq_len, g("gates", "kernel"), g("gates", "bias"), g("candidate", "kernel"),
                                        g("candidate", "bias"), dr == "bw", km[k] if km else None)[0])
            out_tm = np.concatenate(halves, axis=2)
        elif bi_dir:
            out_tm = bilstm_layer(
                x_tm, seq_len,
                weights[enc_var(d, "fw", "kernel")], weights[enc_var(d, "fw", "bias")],
                weights[enc_var(d, "bw", "kernel")], weights[enc_var(d, "bw", "bias")],
                km[0] if km else None, km[1] if km else None)
        else:
            out_tm, _ = lstm_layer(
                x_tm, seq_len, weights[enc_var(d, "", "kernel", False)],
                weights[enc_var(d, "", "bias", False)], False, km[0] if km else None)
        if d in time_major_states:
            time_major_states[d] = out_tm
        out_bm = np.transpose(out_tm, (1, 0, 2))           # encoder.py:164
        if d in attention_states:
            attention_states[d] = out_bm
        seq_len_inps[d] = seq_len
        if skip_step > 1 and i != max_depth - 1 and res < max_scaling_down:  # :172
            enc_in, seq_len = pyramid(out_bm, seq_len, skip_step)
            res *= skip_step
        else:
            enc_in = out_bm
    return attention_states, time_major_states, seq_len_inps


# ----------------------------------------------------------------------------
# attn_decoder.py / decoder.py
# ----------------------------------------------------------------------------
def dec_var(task, leaf):
    """Variable names under model/rnn_decoder_<task>/ (beam_search.py:56-98)."""
    return "model/rnn_decoder_%s/%s" % (task, leaf)


def stack_hidden(entry):
    """Hidden size of a cell-stack entry: (kernel [in+H,4H], bias) = BasicLSTMCell, (wg [in+H,2H], bg, wc [in+H,H], bc) = GRUCell."""
    return entry[2].shape[1] if len(entry) == 4 else entry[0].shape[1] // 4


def decoder_weights(weights, task="char", ind_softmax=False):
    """beam_search.py:53-98 -- name -> role mapping (AttnW squeezed to [D,A]).
    ind_softmax (attn_decoder.py:119-125): the TF-graph decoder's softmax is `rnn/OutputProjection2`, not the
    `rnn/OutputProjection` it otherwise shares with the char LM (lm_encoder.py:108-109).  beam_search.py:75-76 reads
    `OutputProjection` regardless, so beam search callers leave this False."""
    out_scope = "rnn/OutputProjection2/" if ind_softmax else "rnn/OutputProjection/"
    g = lambda leaf: weights[dec_var(task, leaf)]
    opt = lambda leaf: weights.get(dec_var(task, leaf))
    multi = lambda stack, k, leaf: "rnn/multi_rnn_cell%s/cell_%d/basic_lstm_cell/%s" % ("" if stack == "lm" else "_1", k, leaf)
    gru = lambda idx, part, leaf: "rnn/gru_cell%s/%s/%s" % ("" if idx == 0 else "_1", part, leaf)
    if opt(gru(0, "gates", "kernel")) is not None:          # GRUCell decoder (decoder.py:56-59, use_lstm False): stack entries of 4
        cellw = lambda idx: (g(gru(idx, "gates", "kernel")), g(gru(idx, "gates", "bias")),
                             g(gru(idx, "candidate", "kernel")), g(gru(idx, "candidate", "bias")))
        lm_stack, dec_stack = [cellw(0)], [cellw(1)]
    elif opt(multi("lm", 0, "kernel")) is not None:         # MultiRNNCell decoder (decoder.py:66-68, num_layers_dec > 1)
        L = 0
        while opt(multi("lm", L, "kernel")) is not None:
            L += 1
        lm_stack = [(g(multi("lm", k, "kernel")), g(multi("lm", k, "bias"))) for k in range(L)]
        dec_stack = [(g(multi("dec", k, "kernel")), g(multi("dec", k, "bias"))) for k in range(L)]
    else:
        lm_stack = [(g("rnn/basic_lstm_cell/kernel"), g("rnn/basic_lstm_cell/bias"))]
        dec_stack = [(g("rnn/basic_lstm_cell_1/kernel"), g("rnn/basic_lstm_cell_1/bias"))]
    return dict(
        lm_stack=lm_stack, dec_stack=dec_stack,
        lm_lstm_w=lm_stack[0][0], lm_lstm_b=lm_stack[0][1],
        dec_lstm_w=dec_stack[0][0], dec_lstm_b=dec_stack[0][1],
        dec_hidden=stack_hidden(dec_stack[0]), lm_hidden=stack_hidden(lm_stack[0]),
        attn_dec_w=g("rnn/Attention/kernel"), attn_dec_b=g("rnn/Attention/bias"),
        inp_w=g("rnn/InputProjection/kernel"), inp_b=g("rnn/InputProjection/bias"),
        attn_proj_w=g("rnn/AttnProjection/kernel"), attn_proj_b=g("rnn/AttnProjection/bias"),
        out_w=g(out_scope + "kernel"), out_b=g(out_scope + "bias"),
        simple_w=opt("rnn/SimpleProjection/kernel"), simple_b=opt("rnn/SimpleProjection/bias"),
        attn_enc_w=np.squeeze(g("AttnW")) if g("AttnW").ndim == 4 else g("AttnW"),
        attn_v=g("AttnV"), embedding=g("decoder/embedding"))


def attention_tf(q, hf, enc, attn_mask, p):
    """attn_decoder.py:77-93.  q [B,H]; hf [B,Te,A] (= enc . AttnW, :70-73);
    enc [B,Te,D]; attn_mask [B,Te] floats.

    softmax over ALL Te positions, then mask, then renormalise (85-88).
    """
    y = np.matmul(q, p["attn_dec_w"]) + p["attn_dec_b"]              # :80
    s = np.sum(p["attn_v"] * np.tanh(hf + y[:, None, :]), axis=2)    # :82-83
    alpha = softmax(s, axis=1) * attn_mask                           # :85
    alpha = alpha / np.sum(alpha, axis=1, keepdims=True)             # :86-88
    ctx = np.sum(alpha[:, :, None] * enc, axis=1)                    # :92
    return ctx, alpha


def cell_stack(x, states, stack, masks, step):
    """MultiRNNCell.__call__ over DropoutWrapper(BasicLSTMCell) layers (decoder.py:49-72): layer k's input is layer k-1's
    DROPPED output; the state keeps the plain (c, h).  masks: None or per layer an array indexed by step (or None).
    Returns (top dropped output, new states)."""
    new, inp = [], x
    for k, entry in enumerate(stack):
        if len(entry) == 4:          # GRUCell (decoder.py:58-59): the state IS h -- kept as the pair (h, h) so that the
            h = gru_cell(inp, states[k][1], *entry)     # "state selector" [0] of decoder.py:79-80 returns h for it
            new.append((h, h))
        else:
            c, h = lstm_cell(inp, states[k][0], states[k][1], *entry)
            new.append((c, h))
        inp = h if (masks is None or masks[k] is None) else h * masks[k][step]
    return inp, new


def attn_decoder(dec_inp, seq_len, enc, seq_len_inp, weights, task="char",
                 is_training=False, samp_prob=0.0, lm_keep_masks=None,
                 coin=None, sampler=None, return_aux=False, dec_keep_masks=None, ind_softmax=False):
    """attn_decoder.py:37-172 driven by tf.nn.raw_rnn (:166).

    dec_inp [T_dec,B] ints; seq_len [B]; enc [B,Te,D]; seq_len_inp [B].
    Returns logits [(T_out*B), V], time-major flattened (:170), T_out=max(seq_len).

    raw_rnn semantics restated (TF 1.x python/ops/rnn.py raw_rnn body):
      loop_fn(0) -> zero states, lm_input = emb[dec_inp[0]];   (:100-109)
      each iteration t: (out, s') = cell(x, s); then loop_fn(t+1, out, s', loop_state)
      computes attention from get_state(s') = s'.c  (decoder.py:79-80, :114),
      AttnProjection (:116-118), OutputProjection (:124-125), next lm_input
      (:128-145), lm cell (:148), InputProjection (:157-158);
      emit = where(finished_before, 0, logits); s = where(finished_before, s, s');
      loop_state (lm state, context) is NOT copied through.
    Training feedback (:131-145): coin[t] (one scalar for the batch, :132)
    < 1-samp_prob -> ground truth emb[dec_inp[t+1]] else emb[sampler(logits)].
    lm_keep_masks [T_out+1, B, lmH]: DropoutWrapper on the lm cell's output
    (decoder.py:60-63).  The outer cell's dropout is a numerical no-op (its
    output is discarded, only s'.c is used).
    num_layers_dec > 1 (MultiRNNCell, decoder.py:66-68): both cells are stacks; the query is the TOP layer's c
    (decoder.py:77-80); lm_keep_masks / dec_keep_masks are then lists over layers (the outer stack's masks act
    between its layers; the top layer's dropped output is discarded as before).
    """
    p = decoder_weights(weights, task, ind_softmax=ind_softmax)
    emb = p["embedding"]
    dt = enc.dtype
    seq_len = np.asarray(seq_len).astype(np.int64)
    seq_len_inp = np.asarray(seq_len_inp).astype(np.int64)
    B, Te, D = enc.shape
    H, lmH = p["dec_hidden"], p["lm_hidden"]
    V = p["out_w"].shape[1]
    L = len(p["lm_stack"])
    lm_masks = None if lm_keep_masks is None else (list(lm_keep_masks) if L > 1 else [lm_keep_masks])
    dec_masks = None if dec_keep_masks is None else (list(dec_keep_masks) + [None])[:L]
    T_out = int(seq_len.max())
    attn_mask = (np.arange(Te)[None, :] < seq_len_inp[:, None]).astype(dt)   # :60
    hf = np.matmul(enc, p["attn_enc_w"])                                     # :70-73

    def lm_and_input(lm_in, lm_st, ctx, step):
        lm_out, lm_st = cell_stack(lm_in, lm_st, p["lm_stack"], lm_masks, step)   # :148
        if p["simple_w"] is not None:                                          # :149-151
            lm_out = np.matmul(lm_out, p["simple_w"]) + p["simple_b"]
        x = np.matmul(np.concatenate((lm_out, ctx), axis=1), p["inp_w"]) + p["inp_b"]  # :157-158
        return x, lm_st

    # loop_fn(time=0)                                                        :100-109
    st = [(np.zeros((B, H), dt), np.zeros((B, H), dt)) for _ in range(L)]
    lm_st = [(np.zeros((B, lmH), dt), np.zeros((B, lmH), dt)) for _ in range(L)]
    ctx = np.zeros((B, D), dt)
    finished = 0 >= seq_len
    x, lm_st = lm_and_input(emb[dec_inp[0]], lm_st, ctx, 0)
    outs = np.zeros((T_out, B, V), dt)
    aux = dict(alpha=[], ctx=[], q=[], tokens=[])
    t = 0
    while not finished.all():
        _, nst = cell_stack(x, st, p["dec_stack"], dec_masks, t)
        q = nst[-1][0]                                            # decoder.py:77-80: the top layer's c
        ctx, alpha = attention_tf(q, hf, enc, attn_mask, p)                   # :114
        proj = np.matmul(np.concatenate((q, ctx), axis=1), p["attn_proj_w"]) + p["attn_proj_b"]
        logits = np.matmul(proj, p["out_w"]) + p["out_b"]                     # :124-125
        nxt_finished = (t + 1) >= seq_len                                     # :96
        all_fin = nxt_finished.all()                                          # :97
        if not is_training:
            tok = np.argmax(logits, axis=1)                        # decoder.py:149-150
            lm_in = emb[tok]
        else:
            tok = None
            if all_fin:
                lm_in = np.zeros((B, emb.shape[1]), dt)                       # :135
            elif samp_prob > 0 and coin is not None and not (coin[t] < 1 - samp_prob):
                tok = sampler(logits)                              # decoder.py:176-177
                lm_in = emb[tok]
            else:
                lm_in = emb[dec_inp[t + 1]]                                   # :137
        x, lm_st = lm_and_input(lm_in, lm_st, ctx, t + 1)
        fb = finished[:, None]
        outs[t] = np.where(fb, 0, logits)              # raw_rnn emit zero-fill
        st = [(np.where(fb, c0, c1), np.where(fb, h0, h1)) for (c0, h0), (c1, h1) in zip(st, nst)]   # raw_rnn state copy-through
        finished = finished | nxt_finished
        if return_aux:
            aux["alpha"].append(alpha); aux["ctx"].append(ctx); aux["q"].append(q)
            aux["tokens"].append(tok)
        t += 1
    flat = outs.reshape(T_out * B, V)
    return (flat, aux) if return_aux else flat


def create_shifted_targets(dec_inp, seq_len):
    """tf_utils.py:4-12 -- targets = dec_inp[1:]; weights = time-major mask."""
    targets = dec_inp[1:]
    T = targets.shape[0]
    w = (np.arange(T)[:, None] < np.asarray(seq_len)[None, :]).astype(np.float32)
    return targets, w.reshape(-1)


def cross_entropy_loss(logits, targets, seq_len_target):
    """losses.py:7-35.  logits [(T*B),V]; targets [T,B]; seq_len_target [B].

    sparse softmax CE, masked by t < len[b], summed over t, divided by len[b],
    mean over the batch.
    """
    T, B = targets.shape
    seq_len_target = np.asarray(seq_len_target)
    z = logits - logits.max(axis=1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
    cost = -logp[np.arange(T * B), targets.reshape(-1)]
    mask = (np.arange(T)[:, None] < seq_len_target[None, :]).astype(logits.dtype)
    loss = (mask.reshape(-1) * cost).reshape(T, B)
    per_ex = loss.sum(axis=0) / seq_len_target.astype(logits.dtype)
    return per_ex.mean()


def seq2seq_forward(batch, weights, tasks=("char",), num_layers=None, bi_dir=True,
                    is_training=True, max_output=None, avg=True, **dec_kw):
    """seq2seq_model.py:88-157 forward wiring (no dropout, no sampling unless
    passed through dec_kw).  batch: dict logmel [B,T,F], logmel_len, <task> [B,T_dec],
    <task>_len.  Returns dict(outputs, losses, total_loss, enc, enc_len)."""
    num_layers = num_layers or {"char": 4}
    max_output = max_output or {"char": 120, "phone": 250}
    att, _, lens = encoder(batch["logmel"], batch["logmel_len"], weights,
                           {t: num_layers[t] for t in tasks}, bi_dir=bi_dir)
    outputs, losses = {}, {}
    for task in tasks:
        dec_inp = np.transpose(batch[task])                 # seq2seq_model.py:189
        dlen = np.asarray(batch[task + "_len"])
        if not is_training:                                 # :191-193
            dlen = np.ones_like(dlen) * max_output[task]
        d = num_layers[task]
        outputs[task] = attn_decoder(dec_inp, dlen, att[d], lens[d], weights, task,
                                     is_training=is_training, **dec_kw)
        if is_training:
            tgt, _ = create_shifted_targets(dec_inp, dlen)
            losses[task] = cross_entropy_loss(outputs[task], tgt, dlen)
    total = None
    if is_training:
        total = sum(losses.values())
        if avg:
            total = total / float(len(tasks))               # :140-144
    return dict(outputs=outputs, losses=losses, total_loss=total, enc=att, enc_len=lens)


# ----------------------------------------------------------------------------
# optimizer: tf.clip_by_global_norm + tf.train.AdamOptimizer (seq2seq_model.py:137-155)
# ----------------------------------------------------------------------------
def clip_by_global_norm(grads, clip_norm):
    """TF semantics: scale = clip_norm * min(1/global_norm, 1/clip_norm)."""
    gn = np.sqrt(sum(float(np.sum(np.square(g.astype(np.float64)))) for g in grads))
    scale = clip_norm * min(1.0 / gn, 1.0 / clip_norm) if gn > 0 else 1.0
    return [g * np.asarray(scale, g.dtype) for g in grads], gn


def adam_step(var, m, v, g, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m,v EMA; var -= lr_t * m / (sqrt(v) + eps).  `step` is 1-based."""
    lr_t = lr * np.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    var = var - (lr_t * m / (np.sqrt(v) + eps)).astype(var.dtype)
    return var, m, v


# ----------------------------------------------------------------------------
# beam_search.py
# ----------------------------------------------------------------------------
def lm_weights(weights, task="char"):
    """beam_search.py:111-134."""
    g = lambda leaf: weights[dec_var(task, leaf)]
    opt = lambda leaf: weights.get(dec_var(task, leaf))
    return dict(lstm_w=g("rnn/basic_lstm_cell/kernel"), lstm_b=g("rnn/basic_lstm_cell/bias"),
                simple_w=opt("rnn/SimpleProjection/kernel"),
                simple_b=opt("rnn/SimpleProjection/bias"),
                out_w=g("rnn/OutputProjection/kernel"), out_b=g("rnn/OutputProjection/bias"),
                embedding=g("decoder/embedding"))


def calc_attention(enc, p):
    """beam_search.py:137-161 -- returns closure q[H] -> (ctx[D], alpha[T]).
    Unmasked softmax: states are pre-sliced to the true length (eval_model.py:141)."""
    if enc.ndim == 3:
        enc = np.squeeze(enc, axis=0)
    enc_term = np.matmul(enc, p["attn_enc_w"])                                # :148

    def attention(q):
        dec_term = np.matmul(q, p["attn_dec_w"]) + p["attn_dec_b"]            # :151
        s = np.matmul(np.tanh(enc_term + dec_term), p["attn_v"])              # :153-154
        a = softmax(s)
        return np.matmul(a, enc), a                                           # :157
    return attention


def decoder_step(x, x_lm, states, ctx, p, lmp, attention, lm_weight, beam_size):
    """beam_search.py:178-219 (get_top_k) for ONE hypothesis.

    states = [(dec_c,dec_h), (declm_c,declm_h), (lm_c,lm_h)].
    Returns (idx[k], model_score[k], score[k], new_states, ctx', full combined
    log-prob vector).  The order inside idx is np.argpartition's (unspecified);
    callers needing determinism sort.
    """
    dec_state, dec_lm_state, lm_state = states
    dec_lm_state = lstm_cell(x, dec_lm_state[0], dec_lm_state[1], p["lm_lstm_w"], p["lm_lstm_b"])
    o = dec_lm_state[1]
    if p["simple_w"] is not None:
        o = np.matmul(o, p["simple_w"]) + p["simple_b"]
    x_dec = np.matmul(np.concatenate((o, ctx), axis=0), p["inp_w"]) + p["inp_b"]      # :188-189
    dec_state = lstm_cell(x_dec, dec_state[0], dec_state[1], p["dec_lstm_w"], p["dec_lstm_b"])
    ctx, _ = attention(dec_state[0])                                                   # :193
    proj = np.matmul(np.concatenate((dec_state[0], ctx), axis=0), p["attn_proj_w"]) + p["attn_proj_b"]
    log_dec = np.log(softmax(np.matmul(proj, p["out_w"]) + p["out_b"]))                # :196-198
    lm_state = lstm_cell(x_lm, lm_state[0], lm_state[1], lmp["lstm_w"], lmp["lstm_b"])  # :200
    lo = lm_state[1]
    if lmp["simple_w"] is not None:
        lo = np.matmul(lo, lmp["simple_w"]) + lmp["simple_b"]
    log_lm = np.log(softmax(np.matmul(lo, lmp["out_w"]) + lmp["out_b"]))               # :205-207
    comb = log_dec + lm_weight * log_lm                                                # :208
    score = comb + 0.0                                                                 # :210-212
    idx = np.argpartition(score, -beam_size)[-beam_size:]                              # :214
    return idx, comb[idx], score[idx], [dec_state, dec_lm_state, lm_state], ctx, comb


def beam_search(enc, weights, lm_weights_dict=None, beam_size=4, lm_weight=0.0,
                word_ins_penalty=0.0, max_steps=120, task="char", return_all=False):
    """beam_search.py:224-338 restated with Python-3 integer division.

    enc [T,D] (or [1,T,D]).  States start as float64 zeros exactly as the
    reference's np.zeros (236-246) so that all arithmetic promotes to float64.
    Ties: argpartition order is unspecified in the reference; this restatement
    keeps np.argpartition so that it behaves identically on the same NumPy.
    """
    p = decoder_weights(weights, task)
    lmp = lm_weights(lm_weights_dict if lm_weights_dict is not None else weights, task)
    if enc.ndim == 3:
        enc = np.squeeze(enc, axis=0)
    attention = calc_attention(enc, p)
    step = lambda x, xl, st, cx, k: decoder_step(x, xl, st, cx, p, lmp, attention, lm_weight, k)

    x = p["embedding"][GO_ID]; x_lm = lmp["embedding"][GO_ID]                  # :232-233
    h = p["dec_lstm_w"].shape[1] // 4
    hl = p["lm_lstm_w"].shape[1] // 4
    hx = lmp["lstm_w"].shape[1] // 4
    z = lambda n: (np.zeros(n), np.zeros(n))
    zero_attn = np.zeros(enc.shape[1])
    out_list, final_list = [], []
    k = beam_size
    idx, mscore, _, st, cx, _ = step(x, x_lm, [z(h), z(hl), z(hx)], zero_attn, k)   # :255-257
    for i in range(idx.shape[0]):
        tup = ([int(idx[i])], st, cx, mscore[i])
        if idx[i] == EOS_ID:
            final_list.append(tup); k -= 1
        else:
            out_list.append(tup)
    n = 1
    while n < max_steps and k > 0:                                              # :269
        nst, ncx, sl, ml, il = [], [], [], [], []
        for seq, st, cx, cscore in out_list:
            x = p["embedding"][seq[-1]]; x_lm = lmp["embedding"][seq[-1]]
            idx, mscore, score, st2, cx2, _ = step(x, x_lm, st, cx, k)
            nst.append(st2); ncx.append(cx2)
            il.append(idx); sl.append(score + cscore); ml.append(mscore + cscore)
        all_s = np.concatenate(sl); all_m = np.concatenate(ml); all_i = np.concatenate(il)
        top = np.argpartition(all_s, -k)[-k:]                                   # :300
        nxt = all_i[top]; tsc = all_m[top]
        orig = top // k                                                         # :306
        new_list = []
        for j in range(k):
            oc = int(orig[j])
            seq = out_list[oc][0] + [int(nxt[j])]
            tup = (seq, nst[oc], ncx[oc], tsc[j] + word_ins_penalty * len(seq))  # :320-322
            if nxt[j] == EOS_ID:
                final_list.append(tup); k -= 1
            else:
                new_list.append(tup)
        out_list = new_list
        n += 1
    final_list += out_list                                                      # :334
    best = max(final_list, key=lambda t: t[3])                                  # :336
    if return_all:
        return np.asarray(best[0]), [(np.asarray(t[0]), float(t[3])) for t in final_list]
    return np.asarray(best[0])


def greedy_decode_ids(logits_flat, batch_size):
    """eval_model.py:84-87 -- argmax over V, reshape (-1,B), transpose -> [B,T]."""
    ids = np.argmax(logits_flat, axis=1).reshape(-1, batch_size)
    return np.transpose(ids)
