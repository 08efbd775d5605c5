"""Hypothesis record of the beam -- the role of the reference's beam_entry.py (a plain record with five getters).

On the device path a hypothesis does not carry its LSTM states and context vector around: they live in row
`parent_row` of the step's state batch (see BeamSearch), so the record holds the token history and that row index;
the reference's accessor names are kept so that code written against `BeamEntry` keeps working."""


class BeamEntry(object):
    __slots__ = ("index_seq", "dec_state", "context_vec", "cum_attn_probs")

    def __init__(self, index_seq, dec_state, context_vec, cum_attn_probs=None):
        # dec_state / context_vec: the row of the device state batch holding this hypothesis' states and context
        self.index_seq, self.dec_state, self.context_vec, self.cum_attn_probs = index_seq, dec_state, context_vec, cum_attn_probs

    @property
    def parent_row(self):
        return self.dec_state

    def get_last_output(self):
        """Last emitted token id."""
        return self.index_seq[-1]


def _getter(field):
    def get(self):
        return getattr(self, field)
    get.__name__ = "get_" + field
    get.__doc__ = "Accessor kept from the reference's record: `%s`." % field
    return get


for _f in BeamEntry.__slots__:
    setattr(BeamEntry, "get_" + _f, _getter(_f))
