"""Run-to-run determinism of ONE BiLSTM layer forward + backward through the C-ABI (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2e_asr_amd import ops
dev = torch.device("cuda:0")
B, H, T, IN = 32, 256, int(os.environ.get("T", 160)), int(os.environ.get("IN", 1024))
torch.manual_seed(0)
x = torch.randn(B, T, IN, device=dev) * 0.3
ln = torch.full((B,), T, dtype=torch.int32, device=dev)
k = [torch.randn(IN + H, 4 * H, device=dev) * 0.05 for _ in range(2)]
bz = [torch.zeros(4 * H, device=dev) for _ in range(2)]
dout = torch.randn(B, T, 2 * H, device=dev)
res = []
for run in range(3):
    dk = [torch.zeros_like(k[0]) for _ in range(2)]
    db = [torch.zeros_like(bz[0]) for _ in range(2)]
    out, gates, act, hp = ops.lstm_layer_fwd(x, ln, k[0], bz[0], k[1], bz[1], save=True)
    torch.cuda.synchronize()
    dx = ops.lstm_layer_bwd(x, ln, k[0], k[1], dout.clone(), gates, act, hp, dk[0], db[0], dk[1], db[1], need_dx=True, join=True)
    torch.cuda.synchronize()
    ops.check_device_flag(dev)
    res.append(dict(out=out.clone(), act=act.clone(), hp=hp.clone(), dG=gates.clone(), dx=dx.clone(), dk0=dk[0], dk1=dk[1], db0=db[0], db1=db[1]))
for run in (1, 2):
    print("run %d vs 0:" % run, {n: "%.3g" % float((res[run][n] - res[0][n]).abs().max() / res[0][n].abs().max()) for n in res[0]})
