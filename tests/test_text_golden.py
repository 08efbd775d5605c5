"""CPU: transcript utilities against fixtures produced by RUNNING the reference's data_utils.get_relevant_words and
swbd_utils.reverse_swbd_normalizer (oracle/gen_golden.py -> tests/golden/text_utils.json), and Eval.wp_array_to_sent
(eval_model.py:249-258: cut at EOS, join word pieces, U+2581 -> space, strip, de-normalise) composed with them the way
eval_model.py:91-98 scores a hypothesis."""
import json
import os

import pytest

from e2e_asr_amd import data_utils, swbd_utils
from e2e_asr_amd.eval_model import Eval


@pytest.fixture(scope="module")
def fx(golden_dir):
    with open(os.path.join(golden_dir, "text_utils.json")) as f:
        return json.load(f)


def test_constants_match_reference(fx):
    assert data_utils.IGNORED_WORDS == fx["ignored_words"]
    assert [data_utils.PAD_ID, data_utils.GO_ID, data_utils.EOS_ID] == fx["ids"]


def test_get_relevant_words_vs_reference(fx):
    assert len(fx["get_relevant_words"]) >= 20
    for case in fx["get_relevant_words"]:
        words, rel = data_utils.get_relevant_words(case["in"])
        assert list(words) == case["words"], case["in"]
        assert list(rel) == case["rel_words"], case["in"]


def test_reverse_swbd_normalizer_vs_reference(fx):
    norm = swbd_utils.reverse_swbd_normalizer()
    for case in fx["reverse_swbd_normalizer"]:
        assert norm(case["in"]) == case["out"], case["in"]


def test_wp_array_to_sent_composition_vs_reference(fx):
    """Hypothesis ids -> sentence -> scored words, as greedy_decode / beam_search_decode do it: every whitespace word of
    a fixture sentence becomes one word piece `U+2581 + word`; the id sequence carries an EOS followed by junk."""
    norm = swbd_utils.reverse_swbd_normalizer()
    for case, want in zip(fx["reverse_swbd_normalizer"], fx["normalize_then_filter"]):
        words = case["in"].split()
        rev = [b"<pad>", b"<go>", b"<eos>"] + [(u"▁" + w).encode("utf-8") for w in words]
        ids = [3 + i for i in range(len(words))] + [data_utils.EOS_ID, 3, 3]
        sent = Eval.wp_array_to_sent(ids, rev, norm)
        assert sent == norm(" ".join(words))
        got_words, got_rel = data_utils.get_relevant_words(sent)
        assert list(got_words) == want["words"] and list(got_rel) == want["rel_words"], case["in"]
    # pieces inside a word join without a space; no EOS = the whole array
    rev = [b"<pad>", b"<go>", b"<eos>", u"▁he".encode("utf-8"), b"llo", u"▁!".encode("utf-8")]
    assert Eval.wp_array_to_sent([3, 4, 5], rev, norm) == "hello [laughter]"
