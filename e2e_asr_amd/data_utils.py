"""Vocabulary ids and transcript word filtering -- data_utils.py:8-66 of the reference (TensorFlow-free)."""
import os

_PAD, _GO, _EOS = b"<pad>", b"<go>", b"<eos>"
_START_VOCAB = [_PAD, _GO, _EOS]
PAD_ID = 0
GO_ID = 1
EOS_ID = 2

# fillers and noises excluded from scoring (data_utils.py:17-18)
IGNORED_WORDS = ["[noise]", "[laughter]", "[vocalized-noise]", "uh", "um", "eh", "mm", "hm",
                 "ah", "huh", "ha", "er", "oof", "hee", "ach", "eee", "ew"]


def get_relevant_words(char_str):
    """(all words, words that count for WER): `<sp>` is a space; fillers and partial words (`xyz-`) do not count
    (data_utils.py:20-33)."""
    words = char_str.replace("<sp>", " ").split()
    rel_words = [w for w in words if w not in IGNORED_WORDS and not w.endswith("-")]
    return words, rel_words


def initialize_vocabulary(vocabulary_path):
    """One item per line -> ({item: index}, [items]); items are bytes like the reference's (data_utils.py:35-63)."""
    if not os.path.exists(vocabulary_path):
        raise ValueError("Vocabulary file %s not found." % vocabulary_path)
    with open(vocabulary_path, "rb") as f:
        rev_vocab = [line.strip() for line in f.readlines()]
    return dict((x, y) for y, x in enumerate(rev_vocab)), rev_vocab
