"""Evaluation callers of the hot path -- eval_model.py of the reference: greedy decoding of a dev set with word
error rate (56-118), encoder pass + batch-1 beam search with insertion/deletion/substitution counts (120-246),
word-piece ids -> sentence (249-258).  Scoring follows the reference: sentences are split into words, fillers /
noises / partial words are dropped (data_utils.get_relevant_words), WER = sum of word edit distances / sum of gold
words.  Without a vocabulary the error is the token-level edit-distance rate over EOS-trimmed id sequences."""
import os

import numpy as np

from . import data_utils, ops, swbd_utils
from .base_params import BaseParams, Bunch


def edit_distance(a, b):
    """Levenshtein distance between two sequences (the role of edit_distance.SequenceMatcher(...).distance())."""
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def edit_ops(a, b):
    """(distance, insertions, deletions, substitutions) of turning `a` into `b` (eval_model.py:218-229 counts the
    opcodes of SequenceMatcher(decoded, gold)).  The distance is unique; where several optimal alignments exist the
    split into the three kinds follows the backtrace order diagonal, deletion, insertion."""
    a, b = list(a), list(b)
    n, m = len(a), len(b)
    D = np.zeros((n + 1, m + 1), dtype=np.int64)
    D[:, 0] = np.arange(n + 1)
    D[0, :] = np.arange(m + 1)
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            D[i, j] = min(D[i - 1, j] + 1, D[i, j - 1] + 1, D[i - 1, j - 1] + (a[i - 1] != b[j - 1]))
    i, j, ins, dele, sub = n, m, 0, 0, 0
    while i > 0 or j > 0:
        if i > 0 and j > 0 and D[i, j] == D[i - 1, j - 1] + (a[i - 1] != b[j - 1]):
            sub += a[i - 1] != b[j - 1]
            i, j = i - 1, j - 1
        elif i > 0 and D[i, j] == D[i - 1, j] + 1:
            dele += 1
            i -= 1
        else:
            ins += 1
            j -= 1
    return int(D[n, m]), ins, dele, int(sub)


class Eval(BaseParams):
    @classmethod
    def class_params(cls):
        return Bunch(best_model_dir="/scratch", vocab_dir="")

    def __init__(self, model, params=None, rev_char_vocab=None):
        self.params = self.class_params() if params is None else params
        self.model = model
        self.rev_char_vocab = rev_char_vocab if rev_char_vocab is not None else self.load_char_vocab()

    def load_char_vocab(self):
        """vocab_dir/char.vocab, one word piece per line (eval_model.py:51-54); None when there is no such file."""
        vocab_dir = getattr(self.params, "vocab_dir", "") or ""
        p = os.path.join(vocab_dir, "char.vocab")
        if not vocab_dir or not os.path.isfile(p):
            return None
        return data_utils.initialize_vocabulary(p)[1]

    @staticmethod
    def cut_at_eos(ids):
        ids = [int(i) for i in ids]
        return ids[:ids.index(data_utils.EOS_ID)] if data_utils.EOS_ID in ids else ids   # eval_model.py:253-254

    @staticmethod
    def wp_array_to_sent(wp_array, reverse_char_vocab, normalizer=None):
        """eval_model.py:249-258: cut at EOS, join word pieces, U+2581 marks a word start, de-normalise."""
        wp = Eval.cut_at_eos(wp_array)
        pieces = [reverse_char_vocab[i] for i in wp]
        pieces = [p.decode("utf-8") if isinstance(p, bytes) else p for p in pieces]
        sent = "".join(pieces).replace(u"▁", " ").strip()
        return normalizer(sent) if normalizer else sent

    def _check_flag(self):
        """A persistent kernel whose exchange timed out leaves zeros and a device flag: raise instead of scoring
        garbage hypotheses (the host has already synchronised on the ids at this point)."""
        dev = getattr(self.model, "device", None)
        if dev is not None and getattr(dev, "type", str(dev)[:4]) == "cuda":
            ops.check_device_flag(dev)

    def _out_files(self, names):
        d = getattr(self.params, "best_model_dir", "") or ""
        if self.rev_char_vocab is None or not os.path.isdir(d):
            return None
        return [open(os.path.join(d, n), "w") for n in names]

    def greedy_decode(self, batches):
        """Greedy hypotheses of the eval graph over `batches` (a re-iterable of batch dicts: the dev-set iterator until
        OutOfRangeError, eval_model.py:70-108).  Returns the error rate; with a vocabulary and an existing
        best_model_dir also writes gold_asr.txt / decoded_asr.txt / raw_asr.txt (`utt_id<TAB>words`)."""
        rev_normalizer = swbd_utils.reverse_swbd_normalizer()
        files = self._out_files(["gold_asr.txt", "decoded_asr.txt", "raw_asr.txt"])
        total_err, total_len, sent_counter = 0, 0, 0
        try:
            for batch in batches:
                self.model.forward(batch)
                hyp = self.model.greedy_ids("char").cpu().numpy()                 # [B,T]  (:84-87)
                gold = np.asarray(batch["char"])
                utt_ids = batch.get("utt_id", [str(sent_counter + i) for i in range(hyp.shape[0])])
                for b in range(hyp.shape[0]):
                    if self.rev_char_vocab is None:
                        g, h = self.cut_at_eos(gold[b][1:]), self.cut_at_eos(hyp[b])      # drop GO
                    else:
                        gold_asr = self.wp_array_to_sent(gold[b][1:], self.rev_char_vocab, rev_normalizer)
                        dec_asr = self.wp_array_to_sent(hyp[b], self.rev_char_vocab, rev_normalizer)
                        raw_words, h = data_utils.get_relevant_words(dec_asr)
                        _, g = data_utils.get_relevant_words(gold_asr)
                        if files:
                            uid = utt_ids[b].decode("utf-8") if isinstance(utt_ids[b], bytes) else str(utt_ids[b])
                            files[0].write(uid + "\t" + " ".join(g) + "\n")
                            files[1].write(uid + "\t" + " ".join(h) + "\n")
                            files[2].write(uid + "\t" + " ".join(raw_words) + "\n")
                    total_err += edit_distance(g, h)
                    total_len += len(g)
                    sent_counter += 1
        finally:
            for f in files or []:
                f.close()
        self._check_flag()
        return total_err / float(total_len) if total_len else 0.0

    def exec_encoder(self, batches):
        """Encoder side of beam-search evaluation (eval_model.py:120-153): per utterance the char-depth encoder states
        cut to their length, the utterance id and the gold ids without GO."""
        hidden_states_list, utt_id_list, gold_id_list = [], [], []
        depth = self.model.params.num_layers["char"]
        for batch in batches:
            self.model.forward(batch)
            enc = self.model.encoder_hidden_states[depth].cpu().numpy()
            lens = self.model.seq_len_encs[depth]                      # host int64 array (Encoder.__call__)
            lens = np.asarray(lens.cpu() if hasattr(lens, "cpu") else lens)
            gold = np.asarray(batch["char"])
            utt_ids = batch.get("utt_id", [str(len(utt_id_list) + i) for i in range(enc.shape[0])])
            for i in range(enc.shape[0]):
                hidden_states_list.append(enc[i, :int(lens[i])])
                utt_id_list.append(utt_ids[i])
                gold_id_list.append(gold[i][1:])
        self._check_flag()
        return hidden_states_list, utt_id_list, gold_id_list

    def beam_search_decode(self, batches, beam_search, get_counts=False):
        """Beam-search evaluation (eval_model.py:155-246): `beam_search` maps encoder states [T,D] to token ids (a
        `BeamSearch`).  Returns WER (and (ins, del, sub) with get_counts); writes gold.txt / raw_<beam>.txt when
        possible."""
        hidden, utt_ids, golds = self.exec_encoder(batches)
        outputs = [beam_search(h) for h in hidden]
        rev_normalizer = swbd_utils.reverse_swbd_normalizer()
        beam = getattr(getattr(beam_search, "search_params", None), "beam_size", 0)
        files = self._out_files(["gold.txt", "raw_%s.txt" % beam])
        total_err = total_len = ins = dele = sub = 0
        try:
            for uid, gold_ids, out in zip(utt_ids, golds, outputs):
                if self.rev_char_vocab is None:
                    g, h, raw = self.cut_at_eos(gold_ids), self.cut_at_eos(out), None
                else:
                    raw, h = data_utils.get_relevant_words(self.wp_array_to_sent(out, self.rev_char_vocab, rev_normalizer))
                    _, g = data_utils.get_relevant_words(self.wp_array_to_sent(gold_ids, self.rev_char_vocab, rev_normalizer))
                d, i_, d_, s_ = edit_ops(h, g)                       # turn decoded words into gold words (:218)
                total_err, total_len, ins, dele, sub = total_err + d, total_len + len(g), ins + i_, dele + d_, sub + s_
                if files:
                    uid = uid.decode("utf-8") if isinstance(uid, bytes) else str(uid)
                    files[0].write(uid + "\t" + " ".join(g) + "\n")
                    files[1].write(uid + "\t" + " ".join(raw) + "\n")
        finally:
            for f in files or []:
                f.close()
        score = total_err / float(total_len) if total_len else 0.0
        return (score, (ins, dele, sub)) if get_counts else score
