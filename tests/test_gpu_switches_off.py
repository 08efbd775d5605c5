"""The kernels of round 5's second half have an environment switch each that keeps the form they replaced (same-box A/B:
profiles/r05_second_half_ab.log).  The switches are read once per process, so the replaced forms are exercised here in child
processes: the parity tests of the recurrent pair, the decoder chains and the full-batch config-2 / config-4 models must hold
with every switch off -- the A/B's baseline is a correct program, not only a slower one."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

OFF = {"ASR_BPTT_QUAD": "0", "ASR_LSTM_XPRE": "0", "ASR_LSTM_GXL": "0", "ASR_LM_DEFER": "0", "ASR_LM_G4": "0",
       "ASR_CHAIN_BWD_WIDE": "0", "ASR_CHAIN_BWD_ARED": "0", "ASR_EXT_EVENTS": "0", "ASR_DEC_FORK_PRE": "1"}

CASES = {
    "recurrent_pair": ("tests/test_gpu_kernels.py", "test_lstm_layer_fwd or test_lstm_layer_bwd_vs_autograd or test_lstm_properties_full_length_800"),
    "decoder_chains": ("tests/test_gpu_model.py", "test_decoder_chain_path_vs_oracle_and_autograd or "
                                                  "test_decoder_chain_equals_launch_path_under_scheduled_sampling or "
                                                  "test_persistent_lm_chain_equals_per_step_lm_cells or "
                                                  "test_config2_full_batch_persistent_paths_equal_launch_paths"),
    "config2_full_size_and_config4": ("tests/test_gpu_parity2.py", "test_config4_phone_decoder_on_layer2_states_real_widths and 400 and one_launch"),
    "config2_full_size": ("tests/test_gpu_parity3.py", "test_config2_full_size_gradients_vs_autograd"),
}


@pytest.mark.parametrize("family", sorted(CASES))
def test_parity_holds_with_the_replaced_kernel_forms(family):
    path, expr = CASES[family]
    env = dict(os.environ)
    env.update(OFF)
    r = subprocess.run([sys.executable, "-m", "pytest", path, "-x", "-q", "-m", "gpu", "-k", expr, "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, "switches off, %s:\n%s" % (family, out[-4000:])
    assert " passed" in out and " failed" not in out, out[-2000:]
