"""Switchboard transcript de-normalisation -- swbd_utils.py:7-18: the one-character stand-ins go back to their tags."""
import re

_SWBD = {"!": "[laughter]", "@": "[noise]", "#": "[vocalized-noise]"}
_RE = re.compile("(%s)" % "|".join(map(re.escape, _SWBD.keys())))


def reverse_swbd_normalizer():
    return lambda text: _RE.sub(lambda m: _SWBD[m.group(0)], text)
