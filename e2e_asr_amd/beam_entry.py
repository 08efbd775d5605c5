"""Hypothesis record of the beam (reference beam_entry.py:1-23): token history plus the row of
the device-resident state batch that holds its decoder / LM states and context vector."""


class BeamEntry(object):
    def __init__(self, index_seq, dec_state, context_vec, cum_attn_probs=None):
        self.index_seq = index_seq
        self.dec_state = dec_state          # here: row index into the step's device state batch
        self.context_vec = context_vec      # here: same row index (context lives in the state batch)
        self.cum_attn_probs = cum_attn_probs

    def get_last_output(self):
        return self.index_seq[-1]

    def get_index_seq(self):
        return self.index_seq

    def get_dec_state(self):
        return self.dec_state

    def get_context_vec(self):
        return self.context_vec

    def get_cum_attn_probs(self):
        return self.cum_attn_probs
