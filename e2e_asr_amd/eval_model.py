"""Evaluation callers of the hot path -- mirror of the parts of eval_model.py that drive it
(greedy_decode 56-118, wp_array_to_sent 249-258).  Text normalisation / WER on words
(data_utils.get_relevant_words, swbd_utils) is outside the hot path; the error reported here is
the token-level edit-distance rate over EOS-trimmed id sequences (word-level when a vocabulary is given)."""
import numpy as np

from . import data_utils
from .base_params import BaseParams, Bunch


def edit_distance(a, b):
    """Levenshtein distance between two sequences (the role of edit_distance.SequenceMatcher)."""
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


class Eval(BaseParams):
    @classmethod
    def class_params(cls):
        return Bunch(best_model_dir="/scratch", vocab_dir="")

    def __init__(self, model, params=None, rev_char_vocab=None):
        self.params = self.class_params() if params is None else params
        self.model = model
        self.rev_char_vocab = rev_char_vocab

    @staticmethod
    def cut_at_eos(ids):
        ids = list(ids)
        return ids[:ids.index(data_utils.EOS_ID)] if data_utils.EOS_ID in ids else ids   # eval_model.py:253-254

    @staticmethod
    def wp_array_to_sent(wp_array, reverse_char_vocab, normalizer=None):
        """eval_model.py:249-258: cut at EOS, join word pieces, U+2581 marks a word start."""
        wp = Eval.cut_at_eos(wp_array)
        pieces = [reverse_char_vocab[i] for i in wp]
        pieces = [p.decode("utf-8") if isinstance(p, bytes) else p for p in pieces]
        sent = "".join(pieces).replace(u"▁", " ").strip()
        return normalizer(sent) if normalizer else sent

    def greedy_decode(self, batches):
        """Greedy hypotheses of the eval graph over `batches` (an iterable of batch dicts, the role of
        the dev-set iterator until OutOfRangeError, eval_model.py:70-108).  Returns the error rate."""
        total_err, total_len = 0, 0
        for batch in batches:
            self.model.forward(batch)
            hyp = self.model.greedy_ids("char").cpu().numpy()                 # [B,T]  (:84-87)
            gold = np.asarray(batch["char"])
            for b in range(hyp.shape[0]):
                g = self.cut_at_eos(gold[b][1:])                              # drop GO
                h = self.cut_at_eos(hyp[b])
                if self.rev_char_vocab is not None:
                    g = self.wp_array_to_sent(g, self.rev_char_vocab).split()
                    h = self.wp_array_to_sent(h, self.rev_char_vocab).split()
                total_err += edit_distance(h, g)
                total_len += len(g)
        return total_err / float(max(total_len, 1))
