"""GPU: hunt tag-protocol bugs ON PURPOSE (review item, round 4).  Two bugs of the class "a poller takes stale / memset contents of
an exchange slot" were found by accident (round 2: overlapping persistent launches; round 4: the one-launch training decoder's
`p` exchange numbered by the step index).  libe2e_asr_hip_hunt.so is the same library with every PUBLISH and every POLL of the
persistent kernels preceded by a coin toss that puts one wave in eight to sleep for ~4 us (csrc/common.h ASR_RACE_HUNT): late
publishers send pollers into slots that still hold the previous step's (or the memset's) contents, late pollers let their
publishers run ahead into the two-deep parity buffers.  A correct tag protocol gives the same answers, only slower.

Each case runs the existing parity tests of one persistent kernel family -- the ones that compare it with the per-step launch
path, the float64 oracle or autograd -- in a child pytest process that loads the hunt library (a process binds one library).
What those tests hold, they hold against /root/reference's arithmetic: encoder.py:55-91 (recurrent pair), attn_decoder.py:76-162
(decoder chains), lm_encoder.py:90-111 (LM chain)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    # persistent kernel family -> (test file, -k expression)
    "lstm_rec_fwd": ("tests/test_gpu_kernels.py", "test_lstm_layer_fwd or test_lstm_properties_full_length_800"),
    "lstm_rec_bwd": ("tests/test_gpu_kernels.py", "test_lstm_layer_bwd_vs_autograd"),
    "decoder_greedy_inference": ("tests/test_gpu_model.py", "test_greedy_decoder_one_launch_equals_per_step_path_and_oracle or "
                                                            "test_greedy_decoder_kernel_ragged_emit_lengths_and_tiny_shapes"),
    "decoder_greedy_training": ("tests/test_gpu_parity2.py", "test_training_decoder_one_launch_equals_segment_chain_path or "
                                                             "test_training_decoder_one_launch_teacher_forced_vs_oracle_and_autograd"),
    "decoder_chain_fwd_bwd": ("tests/test_gpu_model.py", "test_decoder_chain_path_vs_oracle_and_autograd or "
                                                         "test_decoder_chain_equals_launch_path_under_scheduled_sampling or "
                                                         "test_decoder_chain_long_encoder_one_row_groups or "
                                                         "test_persistent_lm_chain_equals_per_step_lm_cells"),
    "decoder_chain_bwd_two_passes": ("tests/test_gpu_parity2.py", "test_config4_phone_decoder_on_layer2_states_real_widths"),
    "config2_all_chains": ("tests/test_gpu_model.py", "test_config2_full_batch_persistent_paths_equal_launch_paths"),
}


def test_product_library_is_not_the_hunt_build():
    from e2e_asr_amd import _lib
    assert _lib.lib().asr_race_hunt_build() == 0
    assert os.path.exists(os.path.join(ROOT, "e2e_asr_amd", "csrc", "libe2e_asr_hip_hunt.so")), \
        "the race-hunt debug library is missing: run __graft_entry__.build()"


@pytest.mark.parametrize("family", sorted(CASES))
def test_parity_tests_hold_with_randomly_delayed_publishers_and_pollers(family):
    path, expr = CASES[family]
    env = dict(os.environ)
    env["ASR_LIB_VARIANT"] = "hunt"
    env["ASR_EXPECT_HUNT"] = "1"                 # tests/conftest.py: the child asserts that it really bound the hunt build
    r = subprocess.run([sys.executable, "-m", "pytest", path, "-x", "-q", "-m", "gpu", "-k", expr, "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, "hunt build, %s:\n%s" % (family, out[-4000:])
    assert " passed" in out and " failed" not in out, out[-2000:]


_TIME_ONE_LAYER = r"""
import sys, torch
sys.path.insert(0, %r)
from e2e_asr_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
B, T, IN, H = 32, 200, 1024, 256
x = torch.randn(B, T, IN, device=dev, generator=g) * 0.1
ln = torch.full((B,), T, dtype=torch.int32, device=dev)
kf = (torch.rand(IN + H, 4 * H, device=dev, generator=g) - 0.5) * 0.15
kb = (torch.rand(IN + H, 4 * H, device=dev, generator=g) - 0.5) * 0.15
bz = torch.zeros(4 * H, device=dev)
for _ in range(2):
    ops.lstm_layer_fwd(x, ln, kf, bz, kb, bz)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    out = ops.lstm_layer_fwd(x, ln, kf, bz, kb, bz)
e1.record(); torch.cuda.synchronize()
ops.check_device_flag(dev)
print("MS %%.4f SUM %%.9e" %% (e0.elapsed_time(e1) / 5, float(out.double().abs().sum())))
"""


def test_hunt_build_really_delays_and_gives_the_same_bits():
    """The delays exist (one BiLSTM layer is at least 1.5x slower on the hunt build) and change nothing: the recurrence's outputs
    are bit-identical (same arithmetic, only the arrival order of the exchanged values moves)."""
    res = {}
    for variant in ("", "hunt"):
        env = dict(os.environ)
        env["ASR_LIB_VARIANT"] = variant
        r = subprocess.run([sys.executable, "-c", _TIME_ONE_LAYER % ROOT], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=600)
        out = r.stdout.decode(errors="replace")
        assert r.returncode == 0, out[-3000:]
        line = [l for l in out.splitlines() if l.startswith("MS ")][-1].split()
        res[variant] = (float(line[1]), line[3])
    assert res["hunt"][0] > 1.5 * res[""][0], res
    assert res["hunt"][1] == res[""][1], res
