"""GPU: repeat-run detectors for the persistent kernels' forward-progress assumption (every workgroup of a kernel resident).

Round 2 found an intermittent 2-second exchange time-out (about 1 train step in 100) whenever the number of workgroup
groups was not a multiple of 8 (e.g. 30 or 29 utterances: groups spread over the XCDs) and the LM chain's BPTT on the side
stream ran next to the encoder's BPTT: two persistent kernels from two streams starved each other's dispatch.  The decoder
backward now makes the caller's stream wait for the LM chain's BPTT (csrc/decoder_bwd.hip); this test holds that down."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.parametrize("B,T,t_dec,steps", [(30, 83, 27, 400), (29, 120, 12, 200), (7, 64, 9, 100)])
def test_train_steps_never_time_out_with_unaligned_group_counts(B, T, t_dec, steps):
    import bench
    from e2e_asr_amd import ops
    from e2e_asr_amd.weights import synthetic_batch
    dev = torch.device(DEV)
    model = bench.build_model(dev, training=True)
    for it in range(steps):
        b = synthetic_batch(B=B, T=T, F=80, t_dec=t_dec, vocab=1000, variable_len=True, seed=it)
        losses = model.step(b)
        if it % 50 == 49:
            ops.check_device_flag(dev)                      # raises on an exchange time-out
            assert np.isfinite(float(losses["char"].item()))
    ops.check_device_flag(dev)
    assert torch.isfinite(model.variables.flat).all()


@pytest.mark.parametrize("B,T,steps", [(13, 700, 60), (32, 800, 30)])
def test_multitask_steps_with_long_phone_memory_never_time_out(B, T, steps):
    """BASELINE config 4's shape family: char decoder on depth 4 + phone decoder on depth 2, i.e. T/2 = 350 / 400 encoder
    positions for the phone decoder -- the one-launch training kernel with 16 positions per workgroup (round 4) next to the
    8-position instantiation of the char decoder and the one-utterance-per-group backward chain, ragged lengths, scheduled
    sampling and dropout on, repeated steps."""
    import bench
    from e2e_asr_amd import ops
    from e2e_asr_amd.weights import synthetic_batch
    dev = torch.device(DEV)
    model = bench.build_model(dev, training=True, multitask_depth=2)
    for it in range(steps):
        b = synthetic_batch(B=B, T=T, F=80, t_dec=20 + it % 11, vocab=1000, variable_len=True, seed=it, tasks=("char",))
        bp = synthetic_batch(B=B, T=T, F=80, t_dec=35 + it % 17, vocab=bench.V_PHONE, variable_len=True, seed=1000 + it, tasks=("phone",))
        b["phone"], b["phone_len"] = bp["phone"], bp["phone_len"]
        losses = model.step(b)
        if it % 10 == 9:
            ops.check_device_flag(dev)
            assert np.isfinite(float(losses["char"].item())) and np.isfinite(float(losses["phone"].item()))
        ws = model.decoder["phone"].saved["ws"] if getattr(model.decoder["phone"], "saved", None) else None
        assert ws is None or ws.get("greedy_ws") is not None          # the one-launch kernel took the long memory
    ops.check_device_flag(dev)
    assert torch.isfinite(model.variables.flat).all()
