"""Character LM encoder -- mirror of the reference's lm_encoder.py (19-111).

Embedding -> dynamic_rnn(BasicLSTMCell) -> [SimpleProjection] -> OutputProjection, created in the
scope `rnn_decoder_char` so that it IS the attention decoder's inner LM LSTM / embedding / softmax
(lm_model.py:102-103).  The recurrence runs on the same persistent LSTM kernel as the acoustic
encoder (one direction); projections are MFMA GEMMs."""
import numpy as np
import torch

from . import ops
from .base_params import BaseParams, Bunch
from .devcache import dev_i32
from .weights import dec_name, multi_cell_leaf


class LMEncoder(BaseParams):
    @classmethod
    def class_params(cls):
        # lm_encoder.py:22-33
        return Bunch(out_prob=0.9, lm_hidden_size=256, proj_size=256, num_layers=1, emb_size=256, vocab_size=1000)

    def __init__(self, isTraining=True, params=None, variables=None, scope="char"):
        self.params = self.class_params() if params is None else params
        self.isTraining = isTraining
        self.variables = variables
        self.scope = scope
        L = int(self.params.num_layers)
        # lm_encoder.py:61-63: MultiRNNCell of L DropoutWrapper(BasicLSTMCell) layers -- here L stacked persistent layers, each
        # layer's input the DROPPED output of the layer below (DropoutWrapper(output_keep_prob) on every cell)
        self.cell = ("MultiRNNCell([BasicLSTMCell(%d)] * %d)" % (self.params.lm_hidden_size, L)) if L > 1 else \
            "BasicLSTMCell(%d)" % self.params.lm_hidden_size
        self.saved = None
        self.dropout_seed = 0

    def _v(self, leaf):
        return self.variables[dec_name(self.scope, leaf)]

    def _cell_leaf(self, k, leaf):
        """Variable leaf of layer k: the single-layer name (shared with the attention decoder's inner LSTM, lm_model.py:102-103),
        or the MultiRNNCell names weights.multi_cell_leaf gives the decoder's LM stack (the same variables by AUTO_REUSE)."""
        if int(self.params.num_layers) == 1:
            return "rnn/basic_lstm_cell/" + leaf
        return multi_cell_leaf("lm", k, leaf)

    def _layer_seed(self, k):
        return (self.dropout_seed * 1000003 + 7919 * (k + 1)) & 0x7FFFFFFF if k else self.dropout_seed

    def __call__(self, lm_inputs, seq_len):
        """lm_inputs [T+1,B] int32 device tensor; seq_len [B] host.  Returns logits [(T*B),V]
        time-major flattened (lm_encoder.py:90-111)."""
        p = self.params
        dev = lm_inputs.device
        ids = lm_inputs[:-1]                                           # :93
        T, B = ids.shape
        emb = self._v("decoder/embedding")
        x = ops.gather_rows(emb, ids.t().contiguous().reshape(-1)).view(B, T, emb.shape[1])
        lens = np.minimum(np.asarray(seq_len).astype(np.int64), T)
        lens_dev = dev_i32(lens, dev)
        keep = p.out_prob if self.isTraining else 1.0
        rs, xs, inp = [], [], x
        for kk in range(int(p.num_layers)):                             # :61-63 (one layer: the plain cell)
            r = ops.lstm_layer_fwd(inp, lens_dev, self._v(self._cell_leaf(kk, "kernel")), self._v(self._cell_leaf(kk, "bias")),
                                   save=self.isTraining, keep_prob=keep, seed=self._layer_seed(kk))
            rs.append(r); xs.append(inp)
            inp = r[0] if self.isTraining else r
        out = inp
        h_tm = out.transpose(0, 1).contiguous().view(T * B, -1)        # T x B x H => (T x B) x H  (:98-99)
        feat = h_tm
        sp = None
        if p.lm_hidden_size != p.proj_size:                            # :104-106
            sp = ops.gemm(h_tm, self._v("rnn/SimpleProjection/kernel"), self._v("rnn/SimpleProjection/bias"))
            feat = sp
        logits = ops.gemm(feat, self._v("rnn/OutputProjection/kernel"), self._v("rnn/OutputProjection/bias"))
        if self.isTraining:
            self.saved = dict(ids=ids, xs=xs, lens_dev=lens_dev, rs=rs, h_tm=h_tm, sp=sp, keep=keep, T=T, B=B)
        return logits

    def backward(self, dlogits):
        """Accumulates into the flat gradient buffer (embedding, LSTM kernel/bias, projections)."""
        v, sv, p = self.variables, self.saved, self.params
        g = lambda leaf: v.grad_of(dec_name(self.scope, leaf))
        feat = sv["sp"] if sv["sp"] is not None else sv["h_tm"]
        ops.gemm(feat, dlogits, trans_a=True, out=g("rnn/OutputProjection/kernel"), accumulate=True)
        ops.colsum(dlogits, g("rnn/OutputProjection/bias"))
        dfeat = ops.gemm(dlogits, self._v("rnn/OutputProjection/kernel"), trans_b=True)
        if sv["sp"] is not None:
            ops.gemm(sv["h_tm"], dfeat, trans_a=True, out=g("rnn/SimpleProjection/kernel"), accumulate=True)
            ops.colsum(dfeat, g("rnn/SimpleProjection/bias"))
            dfeat = ops.gemm(dfeat, self._v("rnn/SimpleProjection/kernel"), trans_b=True)
        T, B = sv["T"], sv["B"]
        dout = dfeat.view(T, B, -1).transpose(0, 1).contiguous()
        dx = dout
        for kk in reversed(range(len(sv["rs"]))):
            _, gates, act, hprev = sv["rs"][kk]
            dx = ops.lstm_layer_bwd(sv["xs"][kk], sv["lens_dev"], self._v(self._cell_leaf(kk, "kernel")), None, dx.contiguous(), gates, act,
                                    hprev, g(self._cell_leaf(kk, "kernel")), g(self._cell_leaf(kk, "bias")), keep_prob=sv["keep"],
                                    seed=self._layer_seed(kk))
        ops.scatter_add_rows(g("decoder/embedding"), sv["ids"].t().contiguous().reshape(-1), dx.view(B * T, -1))
        self.saved = None
