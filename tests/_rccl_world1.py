"""Helper of tests/test_gpu_dp_train.py::test_rccl_world1_steps_with_and_without_overlap -- run as a child process.

RCCL really initialised (backend "nccl", world size 1, 127.0.0.1) on the one GPU of the box; ASR_DP_FORCE_EXCHANGE=1 makes
DataParallel issue its collectives although there is nobody to exchange with, so the real torch.distributed / RCCL calls
(stream hand-over, async work handles, the exchange stream of the tail overlap) run next to the persistent kernels of a
config-2-width train step.  A sum over one rank is the identity: weights must equal the plain model's."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")
    os.environ["ASR_DP_FORCE_EXCHANGE"] = "1"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from e2e_asr_amd import ops
    from e2e_asr_amd.attn_decoder import AttnDecoder
    from e2e_asr_amd.parallel import DataParallel
    from e2e_asr_amd.seq2seq_model import Seq2SeqModel
    from e2e_asr_amd.weights import synthetic_batch

    def params():
        p = Seq2SeqModel.class_params()
        p.num_layers = {"char": 4}; p.max_output = {"char": 30}
        p.encoder_params.use_lstm = True; p.encoder_params.out_prob = 1.0
        dp = AttnDecoder.class_params()
        dp.out_prob_dec = 1.0; dp.samp_prob = 0.0; dp.vocab_size = 1000
        p.decoder_params = {"char": dp}
        return p
    batches = [synthetic_batch(B=32, T=160 + 16 * i, F=80, t_dec=21, vocab=1000, variable_len=bool(i & 1), seed=100 + i)
               for i in range(4)]
    grads, losses = {}, {}
    for mode in ("plain", "blocking", "overlap", "overlap_bf16"):
        m = Seq2SeqModel(None, True, params(), device="cuda:0", feat_length=80, seed=6)
        d = None
        if mode != "plain":
            d = DataParallel(m, overlap=mode.startswith("overlap"), grad_dtype="bf16" if mode.endswith("bf16") else "f32")
            assert d.world == 1 and d.force_exchange and (d.overlap or mode == "blocking")
        # the gradient after the exchange of the FIRST step (identical weights in every mode): a sum over one rank is the identity
        m.forward(batches[0]); m.backward()
        if d is not None:
            d.all_reduce_grads(m.variables.grad)
        torch.cuda.synchronize()
        grads[mode] = m.variables.grad.cpu().numpy().copy()
        m.apply_gradients()
        # and a few whole steps: real RCCL calls next to the persistent kernels, no time-out, no hang
        ls = []
        for b in batches[1:]:
            ls.append(m.step(b)["char"].item())
        torch.cuda.synchronize()
        ops.check_device_flag(dev)
        losses[mode] = ls
        assert np.isfinite(m.variables.flat.cpu().numpy()).all()
    scale = np.abs(grads["plain"]).max()
    # (not bit-equal even plain vs plain: the split-K weight-gradient GEMMs sum their slices with float atomics)
    assert np.abs(grads["blocking"] - grads["plain"]).max() < 1e-5 * scale
    assert np.abs(grads["overlap"] - grads["plain"]).max() < 1e-5 * scale
    assert np.abs(grads["overlap_bf16"] - grads["plain"]).max() < 1e-2 * scale      # bfloat16 on the wire: 8 significand bits
    for mode in ("blocking", "overlap", "overlap_bf16"):
        np.testing.assert_allclose(losses[mode], losses["plain"], rtol=2e-2)
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
    dist.barrier()
    dist.destroy_process_group()
    print("rccl world-1 ok")


if __name__ == "__main__":
    main()
