"""Weight naming and initialisation -- the checkpoint interface of the reference.

Variable names and layouts are those of the TF graph (verified
against the literal strings beam_search.py:56-98 reads): LSTM kernels are [in+H, 4H]
row-major with gate order i,j,f,o and NO forget-bias offset baked in; AttnW is
[1,1,D,A].  A weight set is a plain dict name -> array, interchangeable with what
tf_utils.get_matching_variables (tf_utils.py:66-90) returns from a checkpoint.
"""
import numpy as np


def enc_name(depth, direction, leaf, bi_dir=True):
    """encoder.py:73-89 under scope 'model' (train.py:184)."""
    if bi_dir:
        return "model/encoder/RNNLayer%d/bidirectional_rnn/%s/basic_lstm_cell/%s" % (depth, direction, leaf)
    return "model/encoder/RNNLayer%d/%d/basic_lstm_cell/%s" % (depth, depth, leaf)


def enc_gru_name(depth, direction, part, leaf, bi_dir=True):
    """GRUCell encoder layer (encoder.py:45-48, use_lstm False): .../gru_cell/{gates,candidate}/{kernel,bias}."""
    return enc_name(depth, direction, "%s/%s" % (part, leaf), bi_dir).replace("basic_lstm_cell", "gru_cell")


def dec_name(task, leaf):
    return "model/rnn_decoder_%s/%s" % (task, leaf)


def _glorot(rng, shape):
    fan_in, fan_out = shape[-2], shape[-1]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, shape).astype(np.float32)


def encoder_layer_inputs(feat, hidden, bi_dir, depth, skip_step=2, max_scaling_down=8,
                         initial_res_fac=1):
    """Input width of every encoder layer (encoder.py:154-178)."""
    dims, res, cur = [], initial_res_fac, feat
    out_w = hidden * (2 if bi_dir else 1)
    for i in range(depth):
        dims.append(cur)
        if skip_step > 1 and i != depth - 1 and res < max_scaling_down:
            cur = out_w * skip_step
            res *= skip_step
        else:
            cur = out_w
    return dims


def multi_cell_leaf(stack, layer, leaf):
    """Variable leaf of layer `layer` of a MultiRNNCell decoder stack (decoder.py:66-68: num_layers_dec > 1) under
    model/rnn_decoder_<task>/.  raw_rnn's scope is `rnn`; the LM stack is built first (`multi_rnn_cell`), the outer stack
    second (`multi_rnn_cell_1`) -- the order that gives the single-layer names `basic_lstm_cell` / `basic_lstm_cell_1`
    (beam_search.py:56-98).  The reference ships no checkpoint or reader for such a model (its beam search maps the
    single-layer names only), so these names are this build's reading of TF-1.x scoping, not a pinned interface."""
    return "rnn/multi_rnn_cell%s/cell_%d/basic_lstm_cell/%s" % ("" if stack == "lm" else "_1", layer, leaf)


def init_weights(feat=80, hidden=256, bi_dir=True, depth=4, tasks=("char",), vocab=None,
                 emb=256, hidden_dec=256, lm_hidden=256, attn_vec=128, num_layers=None,
                 seed=10, skip_step=2, max_scaling_down=8, initial_res_fac=1, num_layers_dec=1, ind_softmax=None,
                 use_lstm=True, dec_use_lstm=True):
    """Random-init weights of the reference architecture.

    use_lstm False: GRUCell encoder layers (encoder.py:45-48) -- gates kernel [in+H, 2H] / bias [2H] = 1.0 (GRUCell's default bias
    initializer for the gates), candidate kernel [in+H, H] / bias 0; kernels from the scope's U(-0.075, 0.075) like the LSTM's.

    Encoder kernels U(-0.075, 0.075) (encoder.py:74), biases 0 (BasicLSTMCell default),
    embedding U(-1,1) (decoder.py:97-99), everything else Glorot-uniform (TF default for
    get_variable with no initializer on scope 'model', train.py:184).  seed mirrors
    tf.set_random_seed(10) (train.py:169) in spirit; TF's RNG stream itself is not reproducible.

    ind_softmax: {task: bool} (attn_decoder.py:119-125, flag 185-186).  For such a task the decoder's softmax lives in
    `rnn/OutputProjection2/*`; `rnn/OutputProjection/*` next to it is what the char LM creates in the same scope
    (lm_encoder.py:108-109) and what beam_search.py:75-76 reads -- it is created here too so that LMModel and
    BeamSearch find it; without LM steps it receives zero gradients and Adam leaves it as initialised.
    """
    rng = np.random.default_rng(seed)
    vocab = vocab or {"char": 1000, "phone": 50}
    num_layers = num_layers or {"char": depth}
    w = {}
    D = hidden * (2 if bi_dir else 1)
    for d, in_dim in enumerate(encoder_layer_inputs(feat, hidden, bi_dir, depth, skip_step, max_scaling_down,
                                                       initial_res_fac), 1):
        for direction in (("fw", "bw") if bi_dir else ("",)):
            if not use_lstm:
                w[enc_gru_name(d, direction, "gates", "kernel", bi_dir)] = rng.uniform(
                    -0.075, 0.075, (in_dim + hidden, 2 * hidden)).astype(np.float32)
                w[enc_gru_name(d, direction, "gates", "bias", bi_dir)] = np.ones(2 * hidden, np.float32)
                w[enc_gru_name(d, direction, "candidate", "kernel", bi_dir)] = rng.uniform(
                    -0.075, 0.075, (in_dim + hidden, hidden)).astype(np.float32)
                w[enc_gru_name(d, direction, "candidate", "bias", bi_dir)] = np.zeros(hidden, np.float32)
                continue
            w[enc_name(d, direction, "kernel", bi_dir)] = rng.uniform(
                -0.075, 0.075, (in_dim + hidden, 4 * hidden)).astype(np.float32)
            w[enc_name(d, direction, "bias", bi_dir)] = np.zeros(4 * hidden, np.float32)
    ind_tasks = []
    for task in tasks:
        V = vocab[task]
        H, lmH, E, A = hidden_dec, lm_hidden, emb, attn_vec
        P = H if lmH != H else lmH
        w[dec_name(task, "decoder/embedding")] = rng.uniform(-1, 1, (V, E)).astype(np.float32)
        w[dec_name(task, "AttnW")] = _glorot(rng, (1, 1, D, A))
        w[dec_name(task, "AttnV")] = _glorot(rng, (1, A))[0]
        w[dec_name(task, "rnn/Attention/kernel")] = _glorot(rng, (H, A))
        w[dec_name(task, "rnn/Attention/bias")] = np.zeros(A, np.float32)
        w[dec_name(task, "rnn/AttnProjection/kernel")] = _glorot(rng, (H + D, H))
        w[dec_name(task, "rnn/AttnProjection/bias")] = np.zeros(H, np.float32)
        w[dec_name(task, "rnn/OutputProjection/kernel")] = _glorot(rng, (H, V))
        w[dec_name(task, "rnn/OutputProjection/bias")] = np.zeros(V, np.float32)
        if ind_softmax and ind_softmax.get(task):       # drawn AFTER every default variable: other seeds' streams unchanged
            ind_tasks.append((task, H, V))
        if not dec_use_lstm:      # GRUCell decoder (decoder.py:56-59): LM cell `gru_cell`, outer cell `gru_cell_1` (creation order as for the LSTMs)
            for idx, (ein, hh) in enumerate(((E, lmH), (E, H))):
                scope = "rnn/gru_cell%s/" % ("" if idx == 0 else "_1")
                w[dec_name(task, scope + "gates/kernel")] = _glorot(rng, (ein + hh, 2 * hh))
                w[dec_name(task, scope + "gates/bias")] = np.ones(2 * hh, np.float32)
                w[dec_name(task, scope + "candidate/kernel")] = _glorot(rng, (ein + hh, hh))
                w[dec_name(task, scope + "candidate/bias")] = np.zeros(hh, np.float32)
        elif num_layers_dec <= 1:
            w[dec_name(task, "rnn/basic_lstm_cell/kernel")] = _glorot(rng, (E + lmH, 4 * lmH))
            w[dec_name(task, "rnn/basic_lstm_cell/bias")] = np.zeros(4 * lmH, np.float32)
            w[dec_name(task, "rnn/basic_lstm_cell_1/kernel")] = _glorot(rng, (E + H, 4 * H))
            w[dec_name(task, "rnn/basic_lstm_cell_1/bias")] = np.zeros(4 * H, np.float32)
        else:       # MultiRNNCell (decoder.py:66-68): the LM stack is created first, the outer stack second
            for k in range(num_layers_dec):
                w[dec_name(task, multi_cell_leaf("lm", k, "kernel"))] = _glorot(rng, ((E if k == 0 else lmH) + lmH, 4 * lmH))
                w[dec_name(task, multi_cell_leaf("lm", k, "bias"))] = np.zeros(4 * lmH, np.float32)
            for k in range(num_layers_dec):
                w[dec_name(task, multi_cell_leaf("dec", k, "kernel"))] = _glorot(rng, ((E if k == 0 else H) + H, 4 * H))
                w[dec_name(task, multi_cell_leaf("dec", k, "bias"))] = np.zeros(4 * H, np.float32)
        w[dec_name(task, "rnn/InputProjection/kernel")] = _glorot(rng, (P + D, E))
        w[dec_name(task, "rnn/InputProjection/bias")] = np.zeros(E, np.float32)
        if lmH != H:
            w[dec_name(task, "rnn/SimpleProjection/kernel")] = _glorot(rng, (lmH, H))
            w[dec_name(task, "rnn/SimpleProjection/bias")] = np.zeros(H, np.float32)
    for task, H, V in ind_tasks:
        w[dec_name(task, "rnn/OutputProjection2/kernel")] = _glorot(rng, (H, V))
        w[dec_name(task, "rnn/OutputProjection2/bias")] = np.zeros(V, np.float32)
    return w


def synthetic_batch(B=32, T=800, F=80, t_dec=121, vocab=1000, variable_len=False, seed=1234,
                    tasks=("char",)):
    """Synthetic filterbank batch of SURVEY.md section 8d (same keys as
    speech_dataset.py:43-45).  variable_len=False: all lengths = T (roofline run);
    True: lengths U[T/2, T] with len[0] = T (masking run)."""
    rng = np.random.default_rng(seed)
    batch = {"logmel": rng.standard_normal((B, T, F)).astype(np.float32)}
    if variable_len:
        ln = rng.integers(T // 2, T + 1, B)
        ln[0] = T
    else:
        ln = np.full(B, T)
    batch["logmel_len"] = ln.astype(np.int64)
    for task in tasks:
        tl = rng.integers(max(1, (t_dec - 1) // 3), t_dec, B)
        tl[-1] = t_dec - 1
        ids = np.zeros((B, t_dec), np.int64)
        for b in range(B):
            ids[b, 0] = 1                                   # GO
            ids[b, 1:tl[b]] = rng.integers(3, vocab, tl[b] - 1)
            ids[b, tl[b]] = 2                               # EOS
        batch[task] = ids
        batch[task + "_len"] = tl.astype(np.int64)
    batch["utt_id"] = np.arange(B)
    return batch
