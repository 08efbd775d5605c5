"""CPU oracle (test infrastructure only) -- see asr_oracle.py header."""
