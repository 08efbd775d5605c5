"""Checkpoint interchange: weights under the reference's TF variable names.

The reference saves TF checkpoints (train.py:202-203,353-371) and reads them back by literal
variable name (beam_search.py:56-98; tf_utils.restore_common_variables, tf_utils.py:53-63).  TF is
not a dependency here; the container is an .npz whose keys are exactly those names (plus
`<name>/Adam`, `<name>/Adam_1` optimizer slots and `global_step`, `learning_rate`), so a script
with TensorFlow can convert in either direction with ckpt_reader.get_tensor / tf.train.Saver.
A path ending in `.safetensors` selects that container instead (same keys; `/` is legal in its tensor names)."""
import os

import numpy as np


def save(path, variables, global_step=0, learning_rate=None, extra=None):
    arrays = variables.to_arrays()
    for slot, (m, v) in variables.adam_slots.items():
        hm, hv = m.detach().cpu().numpy(), v.detach().cpu().numpy()
        for (name, shape, off, n) in variables._specs:
            arrays["%s/%s" % (name, slot)] = hm[off:off + n].reshape(shape).copy()
            arrays["%s/%s_1" % (name, slot)] = hv[off:off + n].reshape(shape).copy()
    arrays["global_step"] = np.asarray(global_step, np.int64)
    if learning_rate is not None:
        arrays["learning_rate"] = np.asarray(learning_rate, np.float64)
    for k, v in (extra or {}).items():
        arrays[k] = np.asarray(v)
    if path.endswith(".safetensors"):
        from safetensors.numpy import save_file
        tmp = path + ".tmp"
        save_file({k: np.ascontiguousarray(v) for k, v in arrays.items()}, tmp)
        os.replace(tmp, path)
        return path
    tmp = path + ".tmp.npz"
    np.savez(tmp, **arrays)
    os.replace(tmp, path if path.endswith(".npz") else path + ".npz")
    return path if path.endswith(".npz") else path + ".npz"


def load(path):
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file
        return dict(load_file(path))
    return dict(np.load(path if path.endswith(".npz") else path + ".npz"))


def restore(variables, path, with_optimizer=True):
    """Full restore (tf.train.Saver().restore, train.py:215). Returns (global_step, learning_rate)."""
    import torch
    arrays = load(path)
    variables.assign({k: v for k, v in arrays.items() if k in variables}, strict=False)
    if with_optimizer:
        slots = set(k.rsplit("/", 1)[1] for k in arrays if k.rsplit("/", 1)[-1] in ("Adam", "AdamLM"))
        for slot in slots:
            m, v = variables.ensure_adam(slot)
            for (name, shape, off, n) in variables._specs:
                if "%s/%s" % (name, slot) in arrays:
                    m[off:off + n].copy_(torch.from_numpy(arrays["%s/%s" % (name, slot)].reshape(-1)).to(m.device))
                    v[off:off + n].copy_(torch.from_numpy(arrays["%s/%s_1" % (name, slot)].reshape(-1)).to(v.device))
    lr = float(np.asarray(arrays["learning_rate"]).reshape(-1)[0]) if "learning_rate" in arrays else None
    return int(np.asarray(arrays.get("global_step", 0)).reshape(-1)[0]), lr


def load_scalars(path, keys):
    """The non-tensor state a TF checkpoint of the reference carries besides weights and slots (epoch counters, the LM's
    step counter and learning-rate variable): {key: python scalar} for the keys present."""
    arrays = load(path)
    return {k: arrays[k].reshape(-1)[0].item() for k in keys if k in arrays}


def restore_common_variables(variables, path):
    """tf_utils.restore_common_variables (tf_utils.py:53-63): warm-start by NAME INTERSECTION
    (-pretrain_lm_path / -pretrain_phone_path, train.py:208-211); optimizer slots are skipped like
    the reference skips names containing 'Adam' (tf_utils.py:86-89)."""
    arrays = {k: v for k, v in load(path).items() if "Adam" not in k}
    restored = variables.assign(arrays, strict=False)
    for name in restored:
        print("Using pre-trained: %s" % name)
    return restored


def get_matching_variables(var_name_substr, path):
    """tf_utils.get_matching_variables (tf_utils.py:66-90)."""
    return {k: v for k, v in load(path).items() if var_name_substr in k and "Adam" not in k}
