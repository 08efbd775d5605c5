"""Special vocabulary ids (reference data_utils.py:13-15) -- the only part of data_utils the hot path needs."""
PAD_ID = 0
GO_ID = 1
EOS_ID = 2
