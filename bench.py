#!/usr/bin/env python3
"""bench.py -- encoder+decoder frames/sec at batch 32 x 800 frames x 80 mel (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one synthetic batch already resident in HBM:
  --mode train : forward (teacher forcing + scheduled sampling + dropout) + backward +
                 [RCCL all-reduce] + global-norm clip + Adam     (train.py:297-299)
  --mode fwd   : forward only (training-mode graph, loss included)
Workload = BASELINE config 2: 4-layer pyramidal BiLSTM(256) + attention decoder(256), V=1000,
B=32 per GPU, T=800, F=80, fp32, synthetic data, random-init weights.  Weak scaling: per-GPU
batch fixed, `value` = N * B * T / max-over-ranks step time.
Adds `roofline` (dominant kernel, HIP-event timed on its launch stream) and `cpu_baseline`
(the CPU oracle timed on this host's cores on a bounded sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, T, F, TDEC, V = 32, 800, 80, 121, 1000
H = 256
# algorithmic FLOPs (SURVEY.md 8d / BASELINE.md section 2)
ENC_LAYER_T = [800, 400, 200, 100]
ENC_LAYER_IN = [80, 1024, 1024, 1024]
REC_FLOP_FWD = sum(2 * H * 4 * H * 2 * t * B for t in ENC_LAYER_T)                  # 50.33 GFLOP
PROJ_FLOP_FWD = sum(2 * i * 4 * H * 2 * t * B for t, i in zip(ENC_LAYER_T, ENC_LAYER_IN))  # 102.34 GFLOP
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: matrix/vector FP32 peak
PEAK_HBM_GBS = 8000.0


def build_model(dev, training=True, multitask_depth=None):
    """Config 2 (default) or, with multitask_depth = d, BASELINE config 4: + an auxiliary phone decoder (V = 50, up to 250
    output steps, main.py:127-129) on the encoder states of depth d (`-nlp`; BASELINE says layer 2)."""
    from e2e_asr_amd.seq2seq_model import Seq2SeqModel
    p = Seq2SeqModel.class_params()
    p.encoder_params.use_lstm = True          # the reference CLI always sets it (encoder.py:187)
    if multitask_depth is not None:
        from e2e_asr_amd.attn_decoder import AttnDecoder
        p.tasks = ["char", "phone"]
        p.num_layers = {"char": 4, "phone": int(multitask_depth)}
        dp = AttnDecoder.class_params(); dp.vocab_size = V_PHONE
        p.decoder_params = {"char": AttnDecoder.class_params(), "phone": dp}
    return Seq2SeqModel(None, isTraining=training, params=p, device=dev, feat_length=F, seed=10)


V_PHONE, TDEC_PHONE = 50, 251


def config4_batch(variable_len=False, seed=1234):
    from e2e_asr_amd.weights import synthetic_batch
    b = synthetic_batch(B=B, T=T, F=F, t_dec=TDEC, vocab=V, variable_len=variable_len, seed=seed, tasks=("char",))
    bp = synthetic_batch(B=B, T=T, F=F, t_dec=TDEC_PHONE, vocab=V_PHONE, variable_len=variable_len, seed=seed + 77, tasks=("phone",))
    b["phone"], b["phone_len"] = bp["phone"], bp["phone_len"]
    return b


def decoder_flop_fwd(te, steps, vocab, D=512, A=128, E=256, Hd=256):
    """Algorithmic forward FLOPs of one attention decoder over a batch: per step and utterance the two
    cells, InputProjection, query projection, scores, context, AttnProjection, OutputProjection; + hf = enc.AttnW once."""
    per_step = 2 * (E + Hd) * 4 * Hd * 2 + 2 * (Hd + D) * E + 2 * Hd * A + 2 * te * A + 2 * te * D + 2 * (Hd + D) * Hd + 2 * Hd * vocab
    return B * (steps * per_step + 2 * te * D * A)


def gemm_roofline(dev, form="nn"):
    """Second roofline line, for the throughput-bound kernel family (all GEMMs together are ~30 % of the GPU time):
    the layer-2 input projection [B*T/2, 1024] x [1024, 4H] of config 2, alone on the chip, HIP events on the
    current stream (the stream the kernel is launched on).  form "tn": the weight gradient of the same layer, dK_x = X^T . dG
    ([1024, B*T/2]^T x [B*T/2, 4H], split-K with float atomics -- the largest row of the kernel statistics); "nt": its data
    gradient dX = dG . K_x^T over both directions ([B*T/2, 8H] x [1024, 8H]^T)."""
    import torch
    from e2e_asr_amd import ops
    M, N, K = B * ENC_LAYER_T[1], 4 * H, ENC_LAYER_IN[1]
    ta = tb = False
    if form == "tn":
        M, N, K, ta = ENC_LAYER_IN[1], 4 * H, B * ENC_LAYER_T[1], True
    elif form == "nt":
        M, N, K, tb = B * ENC_LAYER_T[1], ENC_LAYER_IN[1], 8 * H, True
    a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev)
    c = torch.zeros(M, N, device=dev)
    npl = ops.p3_planes()            # > 0: the encoder's GEMMs run on operands their producers wrote as bf16 planes (csrc/gemm_p3.hip)
    if npl:
        if ta:
            ap, bp = ops.p3_split(a, npl), ops.p3_split(b, npl)
            run = lambda: ops.gemm_p3_rr(ap, bp, out=c, accumulate=True)
        else:
            ap, bp = ops.p3_split(a, npl), ops.p3_split(b, npl, transpose=not tb)
            run = lambda: ops.gemm_p3_kk(ap, bp, None, out=c)
    else:
        run = lambda: ops.gemm(a, b, None, ta, tb, out=c, accumulate=ta)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    prec = ops.get_gemm_precision()
    bf16 = prec != "f32"
    split = ops.get_gemm_split() and not bf16
    # peaks (MI355X_MICROARCH.md): dense bf16 MFMA ~2.5 PFLOP/s; fp32-input MFMA 157.3 TFLOP/s (1/16 of it).  The split3
    # kernel issues 6 bf16 MFMA products per fp32-equivalent product: its ceiling is 2500 / 6 = 416.7 TFLOP/s fp32-equivalent.
    peak = (2500.0 if prec == "bf16" else 2500.0 / 3.0) if bf16 else (2500.0 / 6.0 if split else PEAK_F32_MFMA_TFLOPS)
    F = form.upper()
    name = ("gemm_planes_kernel<%s, 1 plane> (fp32 operands rounded to bf16 on the way into LDS)" % F if prec == "bf16" else
            "gemm_planes_kernel<%s, 2 planes> (bf16x2: three bf16 MFMA products per product; peak = bf16 dense peak / 3)" % F if bf16 else
            "gemm_planes_kernel<%s, 3 planes> = split3 (fp32-accurate: operands split exactly into 3 bf16 planes, 6 bf16 MFMA products, fp32 "
            "accumulate; achieved/peak in fp32-equivalent TFLOP/s, peak = bf16 dense peak / 6)" % F if split else
            "gemm_f32_kernel<%s,128,full> (v_mfma_f32_32x32x2_f32)" % F)
    if npl:
        name = ("gemm_p3_kernel<%s, %d plane%s> (operands pre-split into bf16 planes by their producers: LDS-DMA staging, no conversion "
                "in the k-loop; csrc/gemm_p3.hip)" % ("RR" if ta else "KK", npl, "" if npl == 1 else "s"))
    what = {"nn": "layer-2 input projection", "tn": "layer-2 weight gradient dK_x (split-K, float atomics)",
            "nt": "layer-2 data gradient dX (both directions, K = 8H)"}[form]
    return {"bound": "mfma", "kernel": "%s %s %dx%dx%d" % (name, what, M, N, K),
            "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "avg_launch_ms": ms,
            "vs_fp32_mfma_peak": tf / PEAK_F32_MFMA_TFLOPS}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a container on a 128-core host
    is often limited to a share of it; os.cpu_count() would then oversubscribe the thread pool many times over)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / float(per) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(budget_s=90.0, multitask_depth=None):
    """CPU 'port' baseline, SURVEY 8(d) protocol: the torch twin of the oracle (oracle/torch_ref.py: per-timestep
    BasicLSTMCell loops with masking exactly as dynamic_rnn/raw_rnn run them, float32, autograd backward, TF clip +
    Adam arithmetic) doing full train steps on a bounded sample of the same workload -- the full batch of 32 utterances,
    same architecture, T=96 frames / 15 target tokens instead of 800/120 -- one warm-up + up to 3 timed iterations
    (median), weight conversion outside the timer, with torch.set_num_threads(1) (faithful to train.py:178:
    intra_op_parallelism_threads=1) and with all usable host cores.  A second size (T=192 / 30 tokens) at the faster
    thread setting checks that the per-frame cost is flat in T (every layer is linear in T), so the sample stands for the
    800-frame batch.  `value` is the FASTER setting (the conservative choice for the >=10x target).  Time-boxed: every
    leg stops adding iterations once its share of `budget_s` is spent, and progress goes to stderr."""
    import torch as th
    from e2e_asr_amd.weights import init_weights, synthetic_batch
    from oracle import torch_ref as R
    tasks = ("char",) if multitask_depth is None else ("char", "phone")
    nl = {"char": 4} if multitask_depth is None else {"char": 4, "phone": int(multitask_depth)}
    w = {k: v.astype(np.float32) for k, v in init_weights(seed=10, tasks=tasks, vocab={"char": V, "phone": V_PHONE}).items()}
    ncpu = _usable_cpus()
    prev_threads = th.get_num_threads()
    t_start = time.perf_counter()

    def run(nthreads, Ts, tdec, iters, leg_budget):
        th.set_num_threads(nthreads)
        batch = synthetic_batch(B=B, T=Ts, F=F, t_dec=tdec, vocab=V)
        if multitask_depth is not None:                       # phone targets: twice the char length (250 vs 120 at full size)
            bp = synthetic_batch(B=B, T=Ts, F=F, t_dec=2 * tdec, vocab=V_PHONE, seed=99, tasks=("phone",))
            batch["phone"], batch["phone_len"] = bp["phone"], bp["phone_len"]
        W = R.weights_to_torch(w, dtype=th.float32)          # conversion outside the timer
        times, t_leg = [], time.perf_counter()
        for it in range(iters + 1):                          # iteration 0 = warm-up
            t0 = time.perf_counter()
            for p in W.values():
                p.grad = None
            total, _, _ = R.seq2seq_loss(batch, W, tasks=tasks, num_layers=nl)
            total.backward()
            with th.no_grad():      # clip_by_global_norm + Adam (cost only; arithmetic as seq2seq_model.py:137-155)
                gn = th.sqrt(sum((p.grad.double() ** 2).sum() for p in W.values()))
                sc = 5.0 / max(float(gn), 5.0)
                for p in W.values():
                    g = p.grad * sc
                    m = 0.1 * g; v = 0.001 * g * g
                    p -= 1e-3 * m / (v.sqrt() + 1e-8)
            dt_it = time.perf_counter() - t0
            if it:
                times.append(dt_it)
            sys.stderr.write("[bench cpu_baseline] threads=%d T=%d iteration %d: %.2f s\n" % (nthreads, Ts, it, dt_it))
            sys.stderr.flush()
            if it and time.perf_counter() - t_leg > leg_budget:
                break
            if not it and dt_it > leg_budget:                # the warm-up alone spent the leg: keep it as the one sample
                times.append(dt_it)
                break
        med = float(np.median(times))
        return {"threads": nthreads, "frames": B * Ts, "T": Ts, "target_tokens": tdec - 1, "timed_iterations": len(times),
                "median_s": med, "min_s": float(min(times)), "frames_per_s": B * Ts / med}

    runs = [run(nt, 96, 16, 3, budget_s * 0.3) for nt in sorted(set([1, ncpu]))]
    best = max(runs, key=lambda r: r["frames_per_s"])
    flat = None
    left = budget_s - (time.perf_counter() - t_start)
    if left > 4.5 * best["median_s"]:                         # twice the frames: warm-up + at least one timed iteration fit
        flat = run(best["threads"], 192, 31, 2, left * 0.8)
    th.set_num_threads(prev_threads)
    one = [r for r in runs if r["threads"] == 1][0]
    allc = [r for r in runs if r["threads"] == ncpu][0]
    return dict(value=best["frames_per_s"], unit="frames/s", cores=best["threads"], kind="port",
                sample="full train steps (fwd+bwd+clip+Adam; float32 torch twin of the oracle, per-timestep loops) on 32 "
                       "utterances x 96 frames x 80 mel, 15 target tokens%s; 1 warm-up + up to 3 timed iterations, median; "
                       "value = faster of {1 thread, all usable cores}" % ("" if multitask_depth is None else " (+ 31 phone tokens on depth %d)" % multitask_depth),
                cpu_model=_cpu_model(), host_cores=os.cpu_count(), usable_cores=ncpu,
                threads_1_frames_per_s=one["frames_per_s"], threads_all_frames_per_s=allc["frames_per_s"],
                runs=runs, flatness_check=flat,
                per_frame_cost_ratio_T192_vs_T96=((flat["min_s"] / flat["frames"]) / (best["min_s"] / best["frames"])) if flat else None,
                flatness_note="ratio of the fastest iterations (host noise only ever adds time); 1.0 = per-frame cost independent of T")


def run_config5(args, real_stdout):
    """BASELINE config 5: batch-1 beam search, width 16, LM shallow fusion (beam_search.py:224-338), one MI355X.  One "step" =
    one utterance decoded on the device from encoder states resident in HBM ([100, 512], 120 emitted tokens with random
    weights).  `value` = utterances / s.  cpu_baseline = the float64 NumPy oracle's beam search, 1 thread, ONE utterance
    (the reference's own loop is pure Python over NumPy, 1 thread)."""
    from e2e_asr_amd import ops
    from e2e_asr_amd.beam_search import BeamSearch
    from e2e_asr_amd.weights import init_weights
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    wd = {k: v for k, v in init_weights(seed=3).items() if "rnn_decoder_char" in k}
    wl = {k: v for k, v in init_weights(seed=4).items() if "rnn_decoder_char" in k}
    sp = BeamSearch.class_params()
    sp.beam_size = 16; sp.lm_weight = 0.1; sp.lm_path = wl
    bs = BeamSearch(wd, sp)
    te, D = 100, 512
    encs = [torch.as_tensor((rng.standard_normal((te, D)) * 0.3).astype(np.float32)).to(dev) for _ in range(max(args.steps, 1))]
    for i in range(max(args.warmup, 1)):
        bs(encs[i % len(encs)])
    torch.cuda.synchronize(); ops.check_device_flag(dev)
    t0 = time.perf_counter()
    outs = [bs(e) for e in encs[:args.steps]]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.check_device_flag(dev)
    tokens = sum(len(o) for o in outs)
    us_per_token = dt / max(tokens, 1) * 1e6
    # per emitted token the k = 16 live hypotheses read every decoder and LM weight once (two cells of each model, the
    # projections, the output softmax) and the utterance's enc / hf once: the algorithmic bytes of one beam step
    E, Hd, A, lmH = 256, 256, 128, 256
    wbytes = 4 * ((E + lmH) * 4 * lmH * 2 + (E + Hd) * 4 * Hd + (lmH + D) * E + Hd * A + (Hd + D) * Hd + Hd * V * 2)
    step_bytes = wbytes + 4 * te * (D + A)
    out = {
        "metric": "beam-search utterances/sec (beam 16, LM shallow fusion lm_weight 0.1, batch 1)", "value": args.steps / dt,
        "unit": "utterances/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "replicas only (beam_search.py is batch 1; BASELINE north_star keeps it single-GPU)",
        "vs_baseline": None, "dtype": "f32 step kernels; float64 scoring and selection on the device (beam_search.py computes in float64)",
        "data": "synthetic",
        "config": {"workload": "config5: batch-1 beam search width 16 + LM shallow fusion (lm_weight 0.1, separate LM weight set), encoder "
                               "states [100, 512] resident in HBM, %d emitted tokens per utterance" % (tokens // max(len(outs), 1)),
                   "mode": "decode", "global_batch": 1, "parallelism": "single GPU"},
        "us_per_emitted_token": us_per_token,
        "roofline": {"bound": "hbm", "kernel": "asr_beam_step + asr_beam_select (ten step launches + selection per emitted token)",
                     "achieved": step_bytes / (us_per_token * 1e-6) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": step_bytes / (us_per_token * 1e-6) / 1e9 / PEAK_HBM_GBS, "traffic": None,
                     "algorithmic_bytes_per_step": step_bytes,
                     "note": "a chain of ~11 dependent small kernels per token at batch 16 (rocprofv3: their own durations sum to ~59 of "
                             "the 61 us -- scoring 13, merge 7, attention 10, five skinny GEMV launches 5-7 each; every kernel starts "
                             "with dependent memory round trips): the byte rate is reported, the bound is dependent-access latency "
                             "(DESIGN section 10)"},
    }
    if not args.no_cpu_baseline:
        from oracle import asr_oracle as O
        torch.set_num_threads(1)
        enc0 = encs[0].cpu().numpy()
        t1 = time.perf_counter()
        ref = O.beam_search(enc0, wd, wl, beam_size=16, lm_weight=0.1)
        dtc = time.perf_counter() - t1
        sys.stderr.write("[bench cpu_baseline] oracle beam search: %.1f s for one utterance\n" % dtc)
        out["cpu_baseline"] = {"value": 1.0 / dtc, "unit": "utterances/s", "cores": 1, "kind": "port",
                               "sample": "ONE utterance ([100,512] states, beam 16, lm_weight 0.1, %d tokens) through the float64 NumPy "
                                         "oracle of beam_search.py:224-338, 1 thread" % len(ref),
                               "cpu_model": _cpu_model(), "seconds_per_utterance": dtc,
                               "ids_equal_device": bool(np.array_equal(ref, outs[0]))}
    os.write(real_stdout, (json.dumps(out) + "\n").encode())


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: start N fresh child ranks (one process per GPU, RCCL) under
    torch.distributed.run and relay rank 0's JSON line.  Runs BEFORE anything in this process touches the GPU (never
    re-exec a process that has initialised HIP)."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="auto", choices=["auto", "train", "fwd", "eval"],
                    help="train: full step (default); fwd: training graph forward incl. loss; eval: inference graph "
                         "(isTraining=False: greedy argmax feedback, max_output=120 steps, eval_model.py:56-118)")
    ap.add_argument("--variable-len", action="store_true", help="lengths U[400,800] (masking run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "bf16x2"],
                    help="bf16 = BASELINE config 3's per-GPU workload: bf16 MFMA operands in the GEMMs (fp32 accumulate, state, "
                         "recurrences); the default line is the fp32 config 2")
    ap.add_argument("--gemm", default="split3", choices=["split3", "exact"],
                    help="fp32 products of whole tiles: split3 = on the bf16 matrix pipe by exact 3-way operand splitting "
                         "(default, fp32-accurate); exact = v_mfma_f32_32x32x2_f32 everywhere")
    ap.add_argument("--host-input", action="store_true",
                    help="PCIe-inclusive variant (never the headline value): the filterbank batch starts every step in pinned "
                         "host memory and is copied to HBM inside the timed region")
    ap.add_argument("--graph", action="store_true", help="EXPERIMENT: replay one captured step as a hipGraph (step-varying scalars frozen)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs: 2 = the headline line (default, unchanged); 3 = the same model with --dtype bf16; "
                         "4 = multitask (+ phone decoder on encoder depth --nlp); 5 = batch-1 beam search 16 + LM fusion")
    ap.add_argument("--nlp", type=int, default=2, help="config 4: encoder depth the phone decoder reads (BASELINE: layer 2; reference default 3)")
    args = ap.parse_args()
    if args.config == 3:
        args.dtype = "bf16"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)                                  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (args.gpus == 1 and world == 1):
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("ASR_BENCH_SPAWN_TEST") == "1":
        # CPU rehearsal of the launch path only (tests/test_bench_spawn.py): rendezvous over gloo, one all-reduce, the
        # JSON line from rank 0 -- no GPU, no model, nothing timed
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"spawn_test": True, "n_gpus": world, "sum_of_ones": float(t.item())}))
        dist.destroy_process_group()
        return
    # stdout carries ONE line, the JSON: libraries write there too (RCCL prints a version banner at init), so everything else
    # of this process goes to stderr from here on and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.config == 5:
        if world > 1:
            raise SystemExit("bench.py --config 5: beam search is single-GPU (replicas only)")
        run_config5(args, real_stdout)
        return
    mt = args.nlp if args.config == 4 else None
    dist = None
    if world > 1 or os.environ.get("ASR_FORCE_DIST") == "1":      # (ASR_FORCE_DIST: rehearse the RCCL calls on one GPU under torchrun)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from e2e_asr_amd import ops
    from e2e_asr_amd.weights import synthetic_batch
    ops.set_gemm_precision(args.dtype)
    ops.set_gemm_split(args.gemm == "split3")
    model = build_model(dev, training=args.mode != "eval", multitask_depth=mt)
    has_train = hasattr(model, "step")
    mode = args.mode if args.mode != "auto" else ("train" if has_train else "fwd")
    if world > 1:
        from e2e_asr_amd.parallel import DataParallel
        # broadcast weights, hook the grad all-reduce (config 3 = bf16: the exchange carries bfloat16, fp32 master gradient)
        DataParallel(model, grad_dtype="bf16" if args.dtype == "bf16" else "f32")
    batch = (synthetic_batch(B=B, T=T, F=F, t_dec=TDEC, vocab=V, variable_len=args.variable_len, seed=1234 + rank) if mt is None
             else config4_batch(args.variable_len, seed=1234 + rank))
    # inputs resident in HBM before the timed region
    batch = {k: (torch.as_tensor(v).to(dev) if k == "logmel" else v) for k, v in batch.items()}

    feed = None
    if args.host_input:       # every step's batch starts in HOST memory and is staged one batch ahead (e2e_asr_amd/prefetch.py)
        from e2e_asr_amd.prefetch import DevicePrefetcher
        host_batch = dict(batch); host_batch["logmel"] = batch["logmel"].cpu().numpy()

        def endless():
            while True:
                yield host_batch
        feed = iter(DevicePrefetcher(endless(), dev))

    def one_step():
        b = next(feed) if feed is not None else batch
        if mode == "train":
            model.step(b)
        else:
            model.forward(b)

    # BENCH_r04's 10.34 ms (20 x 7.78 + 51 ms): ONE generation-2 pass of CPython's cyclic garbage collector (26-40 ms over the
    # ~172 000 container objects that torch + the model keep alive; scripts/host_stall.py, profiles/r05_host_stall.txt) landed in
    # the 0.16-s timed window: the enqueue of one step took 40 ms and the GPU ran dry.  Whether a run meets one depends on the
    # allocation count since the interpreter started (generation-2 threshold 10 x 10 x 700), so it came and went with unrelated
    # edits.  The step itself makes no cyclic garbage: collect once and move the survivors to the permanent generation --
    # exactly what Train.train does before its loop (e2e_asr_amd/train.py) -- and count what still runs in the window.
    # BEFORE the warm-up steps, and the timed region's events are created here too: between the last warm-up step and the
    # timed region lies nothing but the bracket the contract asks for (a 40-ms collection there let the GPU's clocks drop and
    # the first timed step ran 8.9 instead of 7.8 ms).
    import gc
    gc.collect()
    gc.freeze()
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    ops.check_device_flag(dev)
    if args.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            one_step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            one_step()
        eager_step = one_step
        one_step = g.replay
        one_step(); torch.cuda.synchronize()
    # Which kernel family the `roofline` object reports is decided BEFORE the timed region, from two profiled steps: inside the
    # region only that family's launches carry events (every event pair costs the launch stream a few microseconds -- with all
    # families on, +0.10 ms per 7.2-ms step, profiles/r05_bench_prof_overhead.txt); the other families' times
    # (`phases_ms_per_step`, `side_stream_tail_ms`) come from PHASE_STEPS extra steps behind the region.
    FAMILIES = ("lstm_rec_fwd", "lstm_rec_bwd", "decoder_fwd", "decoder_bwd", "optim")

    def read_families(nsteps):
        r = {}
        for f in FAMILIES:
            ms, n = ops.prof_read(f)
            r[f] = (ms / nsteps, n / float(nsteps))
        return r

    dom_family = None
    if not args.graph:
        ops.prof_enable(True)
        for _ in range(2):
            one_step()
        torch.cuda.synchronize()
        pre = read_families(2)
        ops.prof_enable(False)
        dom_family = "lstm_rec_bwd" if (mode == "train" and pre["lstm_rec_bwd"][0] > pre["lstm_rec_fwd"][0]) else "lstm_rec_fwd"
        if mt is not None and mode == "train" and max(pre["decoder_fwd"][0], pre["decoder_bwd"][0]) > pre[dom_family][0]:
            dom_family = "decoder_bwd" if pre["decoder_bwd"][0] > pre["decoder_fwd"][0] else "decoder_fwd"
        # (ASR_BENCH_PROF=all / none: every family's events / no events inside the timed region -- the overhead measurement)
        prof_env = os.environ.get("ASR_BENCH_PROF", "")
        ops.prof_enable(prof_env != "none", only=None if prof_env == "all" else (dom_family,))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # self-diagnosis of the timed region (no synchronisation inside it): an event at every step's end on the launch stream,
    # read after the loop, and the host's clock when each step's enqueue returned.  An outlier step, a host-bound loop
    # (host_enqueue_ms ~ the whole region) and lost side-stream overlap (side_stream_tail_ms) are three different pictures.
    host_marks = []
    gc_log, gc_t = [], [0.0]

    def gc_cb(phase, info):
        if phase == "start":
            gc_t[0] = time.perf_counter()
        else:
            gc_log.append((info["generation"], (time.perf_counter() - gc_t[0]) * 1e3))
    gc.callbacks.append(gc_cb)
    t0 = time.perf_counter()
    step_ev[0].record()
    for i in range(args.steps):
        one_step()
        step_ev[i + 1].record()
        host_marks.append(time.perf_counter())
    t_enqueued = time.perf_counter()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.callbacks.remove(gc_cb)
    step_ms = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]
    host_step_ms = [(b - a) * 1e3 for a, b in zip([t0] + host_marks[:-1], host_marks)]
    diag = {"step_ms": [round(x, 4) for x in step_ms],
            "step_ms_median": float(np.median(step_ms)) if step_ms else None, "step_ms_max": max(step_ms) if step_ms else None,
            "step_ms_min": min(step_ms) if step_ms else None,
            "gpu_span_ms": step_ev[0].elapsed_time(step_ev[-1]) if step_ms else None,
            "host_enqueue_ms": (t_enqueued - t0) * 1e3,
            "host_enqueue_ms_per_step": [round(x, 4) for x in host_step_ms],
            "host_enqueue_ms_per_step_median": float(np.median(host_step_ms)) if host_step_ms else None,
            "wall_ms": dt * 1e3,
            "gc_passes_in_timed_region": [{"generation": g, "ms": round(ms, 3)} for g, ms in gc_log],
            "gc_ms_in_timed_region": sum(ms for _, ms in gc_log)}
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ops.check_device_flag(dev)
    comm = None
    if dist is not None and mode == "train":
        # the exchange alone: all-reduce of a buffer the size of the flat fp32 gradient, bus bandwidth by the
        # ring formula 2(N-1)/N * bytes / time, next to the 7 x ~153 GB/s of xGMI links per GPU.  Outside the timed region.
        try:
            cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
            buf = torch.zeros(model.variables.flat.numel(), device=dev, dtype=cdt)
            for _ in range(3):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            dist.barrier()
            tc = time.perf_counter()
            for _ in range(10):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            dtc = (time.perf_counter() - tc) / 10
            nbytes = buf.numel() * buf.element_size()
            comm = {"op": "all_reduce(sum) of the flat gradient as %s" % ("bfloat16 (fp32 master)" if args.dtype == "bf16" else "fp32"), "bytes": nbytes, "ms": dtc * 1e3,
                    "busbw_GBps": 2.0 * (world - 1) / max(world, 1) * nbytes / dtc / 1e9,
                    "link_peak_GBps": 153.0, "links_per_gpu": 7,
                    "placement": "one blocking all-reduce after backward (ASR_DP_OVERLAP=1: per-layer buckets under the BPTT)"}
            del buf
        except Exception as e:      # never lose the bench line to the side measurement
            comm = {"error": repr(e)}
    dom_timed_ms, dom_timed_n = ops.prof_read(dom_family) if dom_family else (0.0, 0)      # the roofline family: events of the timed region
    ops.prof_enable(False)
    PHASE_STEPS = min(5, args.steps)
    tails = []
    fam = {f: (0.0, 0.0) for f in FAMILIES}
    if not args.graph and PHASE_STEPS > 0:
        ops.prof_enable(True)
        for _ in range(PHASE_STEPS):
            one_step()
        torch.cuda.synchronize()
        fam = read_families(PHASE_STEPS)
        tails = ops.prof_read_each("side_tail")
        ops.prof_enable(False)
        ops.check_device_flag(dev)
    if dom_family and dom_timed_n:
        fam[dom_family] = (dom_timed_ms / args.steps, dom_timed_n / float(args.steps))
    # (per-step family times scaled to the timed region's step count: the expressions below divide by args.steps)
    rec_ms, rec_n = fam["lstm_rec_fwd"][0] * args.steps, int(round(fam["lstm_rec_fwd"][1] * args.steps))
    recb_ms, recb_n = fam["lstm_rec_bwd"][0] * args.steps, int(round(fam["lstm_rec_bwd"][1] * args.steps))
    decf_ms, decb_ms, opt_ms = fam["decoder_fwd"][0] * args.steps, fam["decoder_bwd"][0] * args.steps, fam["optim"][0] * args.steps
    if tails:
        # per step: from the launch stream reaching asr_side_join (everything of the backward pass enqueued in front of it done)
        # to the end of the side stream's work.  ~0.6-0.8 ms = the layer-1 weight gradients that follow the last BPTT; a
        # value near the side stream's whole load (~2.5 ms) means the two streams did not overlap.
        diag["side_stream_tail_ms"] = [round(x, 4) for x in tails]
        diag["side_stream_tail_ms_median"] = float(np.median(tails))
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    ms_step = dt / args.steps * 1e3
    frames = world * B * T
    value = frames / (dt / args.steps)
    dec_flop = 14.25e9 if mt is None else 14.25e9 + decoder_flop_fwd(T >> (mt - 1), TDEC_PHONE - 1, V_PHONE)
    flop_per_frame = (REC_FLOP_FWD + PROJ_FLOP_FWD + dec_flop) / (B * T) * (3 if mode == "train" else 1)
    # dominant kernel family by GPU time: the persistent recurrent LSTM pair (4 launches/step each,
    # T=800/400/200/100); report the slower of the two.  Same algorithmic FLOPs (h.K_h resp. dG.K_h^T).
    rec_per_step_ms = rec_ms / args.steps
    recb_per_step_ms = recb_ms / args.steps
    use_bwd = (dom_family == "lstm_rec_bwd") if dom_family in ("lstm_rec_fwd", "lstm_rec_bwd") else (mode == "train" and recb_per_step_ms > rec_per_step_ms)
    dom_ms = recb_per_step_ms if use_bwd else rec_per_step_ms
    v2 = os.environ.get("ASR_LSTM_V2", "1") != "0"
    g4 = v2 and os.environ.get("ASR_LSTM_G4", "1") != "0" and 2 * B * 4 <= 256       # csrc/lstm.hip: groups of four workgroups when the batch fits
    dom_name = (("lstm_rec_bwd4_kernel (persistent BPTT, groups of four workgroups)" if g4 else
                 "lstm_rec_bwd2_kernel<256,2> (persistent BPTT, version 2)" if v2 else "lstm_rec_bwd_ag_kernel<256,2> (persistent BPTT)") if use_bwd else
                ("lstm_rec_fwd4_kernel (persistent recurrent LSTM, groups of four workgroups)" if g4 else
                 "lstm_rec_fwd2_kernel<256,2> (persistent recurrent LSTM, version 2)" if v2 else "lstm_rec_fwd_kernel<256,32,2> (persistent recurrent LSTM)"))
    achieved = REC_FLOP_FWD / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else None
    dom_launches, dom_flop, chain_steps = 4.0, REC_FLOP_FWD, sum(ENC_LAYER_T)
    if mt is not None and mode == "train":
        # config 4: the two decoders' persistent chains outweigh each recurrent kernel -- report the larger decoder family
        # (decoder_fwd = both training-graph decoders; decoder_bwd = both backward chains incl. their GEMMs)
        dec_fwd_step, dec_bwd_step = decf_ms / args.steps, decb_ms / args.steps
        if dom_family in ("decoder_fwd", "decoder_bwd") or (dom_family is None and max(dec_fwd_step, dec_bwd_step) > dom_ms):
            use_dec_bwd = (dom_family == "decoder_bwd") if dom_family else dec_bwd_step > dec_fwd_step
            dom_ms = dec_bwd_step if use_dec_bwd else dec_fwd_step
            dom_flop = dec_flop * (2 if use_dec_bwd else 1)
            dom_name = ("asr_attn_decoder_bwd (persistent backward chains of the char and phone decoders + their GEMMs)" if use_dec_bwd else
                        "asr_attn_decoder_fwd (char and phone: one-launch training decoders; phone memory Te = %d, 16 positions per workgroup)" % (T >> (mt - 1)))
            achieved = dom_flop / (dom_ms * 1e-3) / 1e12
            dom_launches, chain_steps = 2.0, (TDEC - 1) + (TDEC_PHONE - 1)
    # HBM traffic per launch cannot be read by the process that is being timed (the PMC passes are separate rocprofv3
    # runs, MI355X_MICROARCH.md): the committed summary of those passes is quoted, and its provenance is stated.
    traffic, traffic_source = None, None
    for name in (() if mt is not None else ("traffic_r05.json", "traffic_r04.json", "traffic_r03.json", "traffic_r02.json", "traffic_r01.json")):
        tp = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("lstm_rec_bwd_bytes_per_launch" if use_bwd else "lstm_rec_fwd_bytes_per_launch")
                traffic_source = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not measured by this run)" % name
                break
            except Exception:
                traffic = None
    out = {
        "metric": "encoder+decoder frames/sec at batch32x800frx80mel" + ("" if mt is None else " (config 4: multitask)"), "value": value, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32" if args.dtype == "f32" else "bf16 MFMA operands in the GEMMs (%s); fp32 accumulate, recurrences, attention, loss, Adam" % (
            "one operand plane in the encoder's products, two in the decoder's" if args.dtype == "bf16" else "two planes, three products")),
        "gemm_path": (args.dtype if args.dtype != "f32" else
                      "split3: fp32 in / fp32 out / fp32-accurate, evaluated on the bf16 MFMA pipe by exact 3-way operand "
                      "splitting (tests/test_gpu_gemm_split.py holds its error to the v_mfma_f32_32x32x2_f32 kernel's)"
                      if args.gemm == "split3" else "exact: v_mfma_f32_32x32x2_f32"),
        "data": "synthetic",
        "config": {"workload": (("config2" if args.dtype == "f32" else "config3 (per-GPU)") if mt is None else
                                "config4 (per-GPU; + phone decoder V=50, 250 steps, on encoder depth %d = %d positions)" % (mt, T >> (mt - 1))) +
                               ": 4-layer pyramidal BiLSTM(256)+attn decoder(256), V=1000, per-GPU batch "
                               "32x800x80, %s step%s" % (
                                   "full train (fwd+bwd+clip+Adam%s)" % ("+RCCL all-reduce" if world > 1 else "")
                                   if mode == "train" else ("forward-only (training graph incl. loss)" if mode == "fwd" else
                                                            "inference graph (greedy decode, 120 steps)"),
                                   ", variable lengths" if args.variable_len else ", all lengths 800"),
                   "mode": mode, "global_batch": world * B, "frames_per_utt": T, "parallelism": "dp%d" % world},
        "model_tflops": value * flop_per_frame / 1e12,
        "roofline": {"bound": "mfma", "kernel": dom_name + ", 4 launches/step",
                     "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": (achieved / PEAK_F32_MFMA_TFLOPS) if achieved else None, "traffic": traffic,
                     "traffic_source": traffic_source,
                     "avg_launch_ms": dom_ms / dom_launches, "launches": (recb_n if use_bwd else rec_n) if dom_launches == 4.0 else int(2 * args.steps),
                     "serial_chain_steps": chain_steps, "us_per_recurrent_step": dom_ms * 1e3 / chain_steps,
                     "note": "latency-bound serial chain of %d dependent steps; fp32 VALU/MFMA peak is the same 157.3 TF" % chain_steps},
        "phases_ms_per_step": {"lstm_rec_fwd": rec_per_step_ms, "lstm_rec_bwd": recb_ms / args.steps,
                               "decoder_fwd": decf_ms / args.steps, "decoder_bwd": decb_ms / args.steps},
    }
    out["phases_ms_per_step"]["optimizer"] = opt_ms / args.steps
    out["phases_source"] = ("`%s` (the roofline family): HIP events around its launches inside the timed region; the other families and "
                            "side_stream_tail_ms: %d extra steps behind the timed region with every family's events on (not timed)" % (dom_family, PHASE_STEPS))
    out.update(diag)
    sum_len = int(np.sum(batch["logmel_len"])) * world
    out["frames_true_sum_len_per_s"] = sum_len / (dt / args.steps)      # rate on the true sum of lengths next to padded B*T
    out["frames_padded_per_step"], out["frames_true_per_step"] = frames, sum_len
    if args.host_input:
        out["input_residency"] = ("PCIe-INCLUSIVE variant: every step's logmel starts in host memory, is pinned and copied to HBM "
                                  "inside the timed region, one batch ahead of the step on a copy stream (e2e_asr_amd/prefetch.py)")
    else:
        out["input_residency"] = ("logmel resident in HBM before the timed region; the same batch every step, so token ids and "
                                  "lengths are uploaded once (devcache); the PCIe-inclusive rate is `--host-input` (DESIGN.md section 8)")
    if comm is not None:
        out["comm"] = comm
    out["roofline_gemm"] = gemm_roofline(dev)
    out["roofline_gemm_tn"] = gemm_roofline(dev, "tn")
    out["roofline_gemm_nt"] = gemm_roofline(dev, "nt")
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(multitask_depth=mt)
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
