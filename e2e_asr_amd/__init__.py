"""e2e_asr_amd -- MI355X-native hot path of shtoshni/e2e_asr."""
__all__ = ["ops", "weights"]
