import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _hunt_build_is_bound():
    """Child processes of tests/test_gpu_race_hunt.py: the library this process binds must be the race-hunt debug build."""
    if os.environ.get("ASR_EXPECT_HUNT") == "1":
        from e2e_asr_amd import _lib
        assert os.environ.get("ASR_LIB_VARIANT") == "hunt" and _lib.lib().asr_race_hunt_build() == 1
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box gives a 16-CPU share of a
    256-CPU host; torch's default thread count would oversubscribe it 16 times and the float64 twins of the gradient tests ran
    2-3x slower on some boxes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
    except (OSError, ValueError, IndexError):
        pass
    return max(1, n)


@pytest.fixture(scope="session", autouse=True)
def _torch_threads():
    try:
        import torch
        torch.set_num_threads(min(16, _usable_cpus()))
    except Exception:
        pass
    yield


def beam_loop_cases(golden_dir=GOLDEN):
    """Cases of tests/golden/beam_loop.npz: outputs of the reference's OWN BeamSearch.__call__ (beam_search.py:224-338,
    run by oracle/gen_golden.py).  Yields (id string, dict(enc, wd, wl, k, lm_weight, word_ins_penalty, ids))."""
    import numpy as np
    g = np.load(os.path.join(golden_dir, "beam_loop.npz"))
    sets = {}
    for ci in range(int(g["n_cases"])):
        t = "case%02d_" % ci
        variant, enc_key = str(g[t + "variant"]), str(g[t + "enc_key"])
        if variant not in sets:
            sets[variant] = np.load(os.path.join(golden_dir, "decoder_step_%s.npz" % variant))
        w = sets[variant]
        wd = {k[len("w_dec/"):]: np.array(w[k]) for k in w.files if k.startswith("w_dec/")}
        wl = {k[len("w_lm/"):]: np.array(w[k]) for k in w.files if k.startswith("w_lm/")}
        wd["model/rnn_decoder_char/rnn/OutputProjection/bias"][2] += np.float32(g[t + "eos_bias"])
        case = dict(enc=w[enc_key], wd=wd, wl=wl, k=int(g[t + "k"]), lm_weight=float(g[t + "lm_weight"]),
                    word_ins_penalty=float(g[t + "word_ins_penalty"]), ids=g[t + "ids"])
        yield "%s-%s-k%d-lm%g-wip%g-eos%g" % (variant, enc_key, case["k"], case["lm_weight"], case["word_ins_penalty"],
                                               float(g[t + "eos_bias"])), case
