"""Recurrent kernels alone (no co-running side-stream GEMMs): us per step of one BiLSTM layer, B=32, H=256.
Diagnostic only; the bench line's numbers come from bench.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
dev = torch.device("cuda:0")
B, H = int(os.environ.get("B", 32)), int(os.environ.get("H", 256))
ops.set_gemm_precision(os.environ.get("PREC", "f32"))      # PREC=bf16: recurrent products on the bf16 matrix pipe
flush = torch.zeros(1 << 28, device=dev)       # 1 GiB of floats
for T, IN in ((800, 80), (400, 1024), (100, 1024)):
    x = torch.randn(B, T, IN, device=dev) * 0.3
    ln = torch.full((B,), T, dtype=torch.int32, device=dev)
    if os.environ.get("RAGGED") == "1":          # lengths U[T/2, T], the longest first (as bench.py --variable-len)
        g = torch.Generator().manual_seed(5)
        ln = torch.randint(T // 2, T + 1, (B,), generator=g, dtype=torch.int32)
        ln[0] = T
        ln = ln.to(dev)
    k = [torch.randn(IN + H, 4 * H, device=dev) * 0.05 for _ in range(2)]
    bz = [torch.zeros(4 * H, device=dev) for _ in range(2)]
    dk = [torch.zeros_like(k[0]) for _ in range(2)]
    db = [torch.zeros_like(bz[0]) for _ in range(2)]
    ops.prof_enable(True)
    n = 6
    for it in range(n + 2):
        if it == 2:
            torch.cuda.synchronize(); ops.prof_enable(False); ops.prof_enable(True)
        p3f = p3b = None
        kxc = bc = None
        if os.environ.get("P3") == "1":       # outputs / dG as bf16 planes, the GEMMs on plane operands (csrc/gemm_p3.hip)
            kxc = torch.cat([k[0][:IN], k[1][:IN]], 1); bc = torch.cat(bz)
            xp = ops.p3_split(x.reshape(B * T, IN), 3, cols=128 if IN < 128 else 0)
            p3f = dict(np=3, x=xp, out=ops.p3_alloc(B * T, 2 * H, 3, dev), hprev=ops.p3_alloc(B * T, 2 * H, 3, dev))
            if IN >= 128:
                p3f["kxT"] = ops.p3_split(kxc, 3, transpose=True)
            p3b = dict(p3f, dg=ops.p3_alloc(B * T, 8 * H, 3, dev), colmap=ops.p3_colmap(H, dev), kxu=ops.p3_split(kxc, 3, unit_major_h=H))
        out, gates, act, hp = ops.lstm_layer_fwd(x, ln, k[0], bz[0], k[1], bz[1], save=True, kx_cat=kxc, bias_cat=bc, p3=p3f)
        dout = torch.ones_like(out)
        if os.environ.get("FLUSH") == "1":      # evict the Infinity Cache: in the train step the saved activations are ~8 ms old
            flush.add_(1.0)
        torch.cuda.synchronize()
        ops.lstm_layer_bwd(x, ln, k[0], k[1], dout, gates, act, hp, dk[0], db[0], dk[1], db[1], need_dx=IN % 256 == 0, join=True,
                           kx_cat=kxc, p3=p3b)
        torch.cuda.synchronize()
    f_ms, f_n = ops.prof_read("lstm_rec_fwd")
    b_ms, b_n = ops.prof_read("lstm_rec_bwd")
    print("T=%4d in=%4d  fwd %.3f us/step (%d launches)   bwd %.3f us/step (%d launches)   per layer call: fwd %.1f us  bwd %.1f us" % (
        T, IN, f_ms / f_n / T * 1e3, f_n, b_ms / b_n / T * 1e3, b_n, f_ms / n * 1e3, b_ms / n * 1e3))
