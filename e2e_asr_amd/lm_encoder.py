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
from .weights import dec_name


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
        if self.params.num_layers > 1:
            raise NotImplementedError("MultiRNNCell LM (num_layers > 1) is outside the hot path")
        self.cell = "BasicLSTMCell(%d)" % self.params.lm_hidden_size
        self.saved = None
        self.dropout_seed = 0

    def _v(self, leaf):
        return self.variables[dec_name(self.scope, leaf)]

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
        k, b = self._v("rnn/basic_lstm_cell/kernel"), self._v("rnn/basic_lstm_cell/bias")
        r = ops.lstm_layer_fwd(x, lens_dev, k, b, save=self.isTraining, keep_prob=keep, seed=self.dropout_seed)
        out = r[0] if self.isTraining else r
        h_tm = out.transpose(0, 1).contiguous().view(T * B, -1)        # T x B x H => (T x B) x H  (:98-99)
        feat = h_tm
        sp = None
        if p.lm_hidden_size != p.proj_size:                            # :104-106
            sp = ops.gemm(h_tm, self._v("rnn/SimpleProjection/kernel"), self._v("rnn/SimpleProjection/bias"))
            feat = sp
        logits = ops.gemm(feat, self._v("rnn/OutputProjection/kernel"), self._v("rnn/OutputProjection/bias"))
        if self.isTraining:
            self.saved = dict(ids=ids, x=x, lens_dev=lens_dev, r=r, h_tm=h_tm, sp=sp, keep=keep, T=T, B=B)
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
        _, gates, act, hprev = sv["r"]
        dx = ops.lstm_layer_bwd(sv["x"], sv["lens_dev"], self._v("rnn/basic_lstm_cell/kernel"), None, dout, gates, act, hprev,
                                g("rnn/basic_lstm_cell/kernel"), g("rnn/basic_lstm_cell/bias"), keep_prob=sv["keep"],
                                seed=self.dropout_seed)
        ops.scatter_add_rows(g("decoder/embedding"), sv["ids"].t().contiguous().reshape(-1), dx.view(B * T, -1))
        self.saved = None
