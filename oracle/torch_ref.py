"""Torch (CPU, autograd) restatement of the reference TF graph -- TEST INFRASTRUCTURE.

Same op-for-op structure as oracle/asr_oracle.py (which is the primary oracle and is
pinned by the reference's golden vectors); this twin exists because the product path needs
GRADIENT parity (tf.gradients, seq2seq_model.py:148) and a CPU train-step baseline, and
NumPy has no autograd.  tests/test_oracle_torch_ref.py checks its forward against
asr_oracle on the same inputs, so the gradients it yields are gradients of the pinned
forward.  Per-timestep BasicLSTMCell loops exactly as dynamic_rnn / raw_rnn execute them
(no fused/oneDNN RNN primitive).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.
"""
import numpy as np
import torch

from . import asr_oracle as O


def lstm_cell(x, c, h, w, b):
    """basic_lstm.py:14-23."""
    z = torch.cat((x, h), -1) @ w + b
    i, j, f, o = z.chunk(4, -1)
    nc = c * torch.sigmoid(f + 1) + torch.sigmoid(i) * torch.tanh(j)
    return nc, torch.sigmoid(o) * torch.tanh(nc)


def lstm_layer(x_tm, seq_len, w, b, reverse=False, keep_mask=None):
    """encoder.py:55-91, one direction (see asr_oracle.lstm_layer)."""
    T, B, _ = x_tm.shape
    H = w.shape[1] // 4
    seq_len = torch.as_tensor(np.asarray(seq_len), dtype=torch.long)
    c = x_tm.new_zeros(B, H); h = x_tm.new_zeros(B, H)
    ar = torch.arange(B)
    outs = [None] * T if not reverse else None
    if reverse:
        out = x_tm.new_zeros(T, B, H)
    for s in range(int(seq_len.max())):
        live = (s < seq_len)
        t_idx = (seq_len - 1 - s) if reverse else torch.full((B,), s, dtype=torch.long)
        t_safe = torch.where(live, t_idx, torch.zeros_like(t_idx))
        nc, nh = lstm_cell(x_tm[t_safe, ar], c, h, w, b)
        lv = live[:, None]
        c = torch.where(lv, nc, c); h = torch.where(lv, nh, h)
        if reverse:
            out = out.index_put((t_safe[live], ar[live]), nh[live])
        else:
            outs[s] = torch.where(lv, nh, torch.zeros_like(nh))
    if not reverse:
        z = x_tm.new_zeros(B, H)
        out = torch.stack([o if o is not None else z for o in outs], 0)
    if keep_mask is not None:
        out = out * keep_mask
    return out


def gru_layer(x_tm, seq_len, wg, bg, wc, bc, reverse=False, keep_mask=None):
    """encoder.py:55-91 with tf.nn.rnn_cell.GRUCell, one direction (see asr_oracle.gru_layer)."""
    T, B, _ = x_tm.shape
    H = wc.shape[1]
    seq_len = torch.as_tensor(np.asarray(seq_len), dtype=torch.long)
    h = x_tm.new_zeros(B, H)
    ar = torch.arange(B)
    out = x_tm.new_zeros(T, B, H)
    for s in range(int(seq_len.max())):
        live = (s < seq_len)
        t_idx = (seq_len - 1 - s) if reverse else torch.full((B,), s, dtype=torch.long)
        t_safe = torch.where(live, t_idx, torch.zeros_like(t_idx))
        x = x_tm[t_safe, ar]
        v = torch.sigmoid(torch.cat((x, h), -1) @ wg + bg)
        r, u = v.chunk(2, -1)
        c = torch.tanh(torch.cat((x, r * h), -1) @ wc + bc)
        nh = u * h + (1 - u) * c
        h = torch.where(live[:, None], nh, h)
        out = out.index_put((t_safe[live], ar[live]), nh[live])
    if keep_mask is not None:
        out = out * keep_mask
    return out


def encoder(x_bm, seq_len, W, num_layers, bi_dir=True, skip_step=2, max_scaling_down=8, keep_masks=None):
    """encoder.py:122-180."""
    att, lens = {}, {}
    max_depth = max(num_layers.values())
    seq_len = np.asarray(seq_len).astype(np.int64)
    res, enc_in = 1, x_bm
    for i in range(max_depth):
        d = i + 1
        x_tm = enc_in.transpose(0, 1)
        km = keep_masks.get(d) if keep_masks else (None, None)
        if O.enc_gru_var(d, "fw" if bi_dir else "", "gates", "kernel", bi_dir) in W:
            halves = []
            for k, dr in enumerate(("fw", "bw") if bi_dir else ("",)):
                g = lambda part, leaf: W[O.enc_gru_var(d, dr, part, leaf, bi_dir)]
                halves.append(gru_layer(x_tm, seq_len, g("gates", "kernel"), g("gates", "bias"), g("candidate", "kernel"),
                                        g("candidate", "bias"), dr == "bw", km[k]))
            out = torch.cat(halves, 2)
        elif bi_dir:
            fw = lstm_layer(x_tm, seq_len, W[O.enc_var(d, "fw", "kernel")], W[O.enc_var(d, "fw", "bias")], False, km[0])
            bw = lstm_layer(x_tm, seq_len, W[O.enc_var(d, "bw", "kernel")], W[O.enc_var(d, "bw", "bias")], True, km[1])
            out = torch.cat((fw, bw), 2)
        else:
            out = lstm_layer(x_tm, seq_len, W[O.enc_var(d, "", "kernel", False)], W[O.enc_var(d, "", "bias", False)],
                             False, km[0])
        out_bm = out.transpose(0, 1)
        if d in num_layers.values():
            att[d] = out_bm
        lens[d] = seq_len
        if skip_step > 1 and i != max_depth - 1 and res < max_scaling_down:
            B, T, Fd = out_bm.shape
            rem = int(seq_len.max()) % skip_step
            if rem:
                out_bm = torch.cat((out_bm, out_bm.new_zeros(B, skip_step - rem, Fd)), 1)
            enc_in = out_bm.reshape(B, out_bm.shape[1] // skip_step, Fd * skip_step)
            seq_len = np.ceil(seq_len / float(skip_step)).astype(np.int64)
            res *= skip_step
        else:
            enc_in = out_bm
    return att, lens


def attn_decoder(dec_inp, seq_len, enc, seq_len_inp, W, task="char", lm_keep_masks=None, tokens=None, dec_keep_masks=None,
                 ind_softmax=False):
    """attn_decoder.py:37-172 in training mode.  `tokens` [T_out,B]: the token actually fed
    at each step (teacher or sampled, taken from the product path so both follow one path).
    MultiRNNCell decoders (num_layers_dec > 1): stacks as in asr_oracle.cell_stack; the masks are then lists over layers."""
    g = lambda leaf: W[O.dec_var(task, leaf)]
    opt = lambda leaf: W.get(O.dec_var(task, leaf))
    multi = lambda stack, k, leaf: "rnn/multi_rnn_cell%s/cell_%d/basic_lstm_cell/%s" % ("" if stack == "lm" else "_1", k, leaf)
    gru = lambda idx, part, leaf: "rnn/gru_cell%s/%s/%s" % ("" if idx == 0 else "_1", part, leaf)
    if opt(gru(0, "gates", "kernel")) is not None:           # GRUCell decoder (decoder.py:56-59)
        L = 1
        cellw = lambda idx: (g(gru(idx, "gates", "kernel")), g(gru(idx, "gates", "bias")),
                             g(gru(idx, "candidate", "kernel")), g(gru(idx, "candidate", "bias")))
        lm_stack, dec_stack = [cellw(0)], [cellw(1)]
    elif opt(multi("lm", 0, "kernel")) is not None:
        L = 0
        while opt(multi("lm", L, "kernel")) is not None:
            L += 1
        lm_stack = [(g(multi("lm", k, "kernel")), g(multi("lm", k, "bias"))) for k in range(L)]
        dec_stack = [(g(multi("dec", k, "kernel")), g(multi("dec", k, "bias"))) for k in range(L)]
    else:
        L = 1
        lm_stack = [(g("rnn/basic_lstm_cell/kernel"), g("rnn/basic_lstm_cell/bias"))]
        dec_stack = [(g("rnn/basic_lstm_cell_1/kernel"), g("rnn/basic_lstm_cell_1/bias"))]
    lm_masks = None if lm_keep_masks is None else (list(lm_keep_masks) if L > 1 else [lm_keep_masks])
    dec_masks = None if dec_keep_masks is None else (list(dec_keep_masks) + [None])[:L]

    def stack_step(x, states, stack, masks, step):
        new, inp = [], x
        for k, entry in enumerate(stack):
            if len(entry) == 4:          # GRUCell: state = h, kept as (h, h) (see asr_oracle.cell_stack)
                wg, bg, wc, bc = entry
                hp = states[k][1]
                v = torch.sigmoid(torch.cat((inp, hp), -1) @ wg + bg)
                r, u = v.chunk(2, -1)
                cnd = torch.tanh(torch.cat((inp, r * hp), -1) @ wc + bc)
                h = u * hp + (1 - u) * cnd
                new.append((h, h))
            else:
                c, h = lstm_cell(inp, states[k][0], states[k][1], *entry)
                new.append((c, h))
            inp = h if (masks is None or masks[k] is None) else h * masks[k][step]
        return inp, new
    emb = g("decoder/embedding")
    seq_len = np.asarray(seq_len).astype(np.int64)
    B, Te, D = enc.shape
    H, lmH = O.stack_hidden(dec_stack[0]), O.stack_hidden(lm_stack[0])
    T_out = int(seq_len.max())
    mask = (torch.arange(Te)[None] < torch.as_tensor(np.asarray(seq_len_inp))[:, None]).to(enc.dtype)
    aw = g("AttnW"); aw = aw.reshape(aw.shape[-2], aw.shape[-1])
    hf = enc @ aw
    z = lambda n: enc.new_zeros(B, n)
    st = [(z(H), z(H)) for _ in range(L)]
    lst = [(z(lmH), z(lmH)) for _ in range(L)]
    ctx = z(D)
    dec_inp = torch.as_tensor(np.asarray(dec_inp), dtype=torch.long)
    toks = dec_inp if tokens is None else torch.as_tensor(np.asarray(tokens), dtype=torch.long)
    outs = []
    fin = torch.as_tensor(0 >= seq_len)
    for t in range(T_out):
        lo, lst = stack_step(emb[toks[t]], lst, lm_stack, lm_masks, t)
        if opt("rnn/SimpleProjection/kernel") is not None:
            lo = lo @ g("rnn/SimpleProjection/kernel") + g("rnn/SimpleProjection/bias")
        x = torch.cat((lo, ctx), 1) @ g("rnn/InputProjection/kernel") + g("rnn/InputProjection/bias")
        _, nst = stack_step(x, st, dec_stack, dec_masks, t)
        q = nst[-1][0]
        y = q @ g("rnn/Attention/kernel") + g("rnn/Attention/bias")
        s = (g("AttnV") * torch.tanh(hf + y[:, None, :])).sum(2)
        a = torch.softmax(s, 1) * mask
        a = a / a.sum(1, keepdim=True)
        ctx = (a[:, :, None] * enc).sum(1)
        p = torch.cat((q, ctx), 1) @ g("rnn/AttnProjection/kernel") + g("rnn/AttnProjection/bias")
        osc = "rnn/OutputProjection2/" if ind_softmax else "rnn/OutputProjection/"       # attn_decoder.py:119-125
        logits = p @ g(osc + "kernel") + g(osc + "bias")
        fb = fin[:, None]
        outs.append(torch.where(fb, torch.zeros_like(logits), logits))
        st = [(torch.where(fb, c0, c1), torch.where(fb, h0, h1)) for (c0, h0), (c1, h1) in zip(st, nst)]
        fin = fin | torch.as_tensor((t + 1) >= seq_len)
    return torch.cat(outs, 0)


def cross_entropy_loss(logits, targets, seq_len):
    """losses.py:7-35."""
    T, B = targets.shape
    ln = torch.as_tensor(np.asarray(seq_len))
    ce = torch.nn.functional.cross_entropy(logits, torch.as_tensor(np.asarray(targets), dtype=torch.long).reshape(-1),
                                           reduction="none").reshape(T, B)
    m = (torch.arange(T)[:, None] < ln[None]).to(logits.dtype)
    return ((ce * m).sum(0) / ln.to(logits.dtype)).mean()


def seq2seq_loss(batch, W, tasks=("char",), num_layers=None, bi_dir=True, avg=True, tokens=None,
                 enc_keep_masks=None, lm_keep_masks=None, dec_keep_masks=None, ind_softmax=None):
    """seq2seq_model.py:88-144 in training mode -> (total_loss, {task: loss}, {task: logits}).
    ind_softmax: {task: bool} (attn_decoder.py:119-125)."""
    num_layers = num_layers or {"char": 4}
    x = torch.as_tensor(batch["logmel"])
    att, lens = encoder(x, batch["logmel_len"], W, {t: num_layers[t] for t in tasks}, bi_dir=bi_dir,
                        keep_masks=enc_keep_masks)
    losses, outs = {}, {}
    for task in tasks:
        dec_inp = np.transpose(np.asarray(batch[task]))
        dlen = np.asarray(batch[task + "_len"])
        d = num_layers[task]
        outs[task] = attn_decoder(dec_inp, dlen, att[d], lens[d], W, task,
                                  lm_keep_masks=None if lm_keep_masks is None else lm_keep_masks[task],
                                  tokens=None if tokens is None else tokens[task],
                                  dec_keep_masks=None if dec_keep_masks is None else dec_keep_masks[task],
                                  ind_softmax=bool(ind_softmax and ind_softmax.get(task)))
        T_out = int(dlen.max())
        losses[task] = cross_entropy_loss(outs[task], dec_inp[1:1 + T_out], dlen)
    total = sum(losses.values())
    if avg:
        total = total / float(len(tasks))
    return total, losses, outs


def weights_to_torch(weights, dtype=torch.float64, requires_grad=True):
    return {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad) for k, v in weights.items()}


def train_step_reference(batch, weights, adam_state, step, lr=1e-3, clip=5.0, **kw):
    """One reference optimizer step (seq2seq_model.py:137-155): tf.gradients ->
    clip_by_global_norm(5.0) -> Adam.  Returns (new_weights, new_state, loss, grads, gnorm)."""
    W = weights_to_torch(weights)
    total, _, _ = seq2seq_loss(batch, W, **kw)
    total.backward()
    names = list(weights.keys())
    grads = [W[n].grad.numpy() if W[n].grad is not None else np.zeros_like(weights[n], np.float64) for n in names]
    clipped, gn = O.clip_by_global_norm(grads, clip)
    new_w, new_s = {}, {}
    for n, g in zip(names, clipped):
        m, v = adam_state.get(n, (np.zeros_like(g), np.zeros_like(g)))
        nw, m, v = O.adam_step(np.asarray(weights[n], np.float64), m, v, g, step, lr)
        new_w[n] = nw; new_s[n] = (m, v)
    return new_w, new_s, float(total.item()), dict(zip(names, grads)), gn
