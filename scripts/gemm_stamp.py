"""EXPERIMENT: phase stamps of the split3 GEMM (a library built with -DASR_EXP_STAMP; ASR_LIB_PATH points at it)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from e2e_asr_amd import ops, _lib
M, N, K, ta, tb = [int(x) for x in sys.argv[1:6]]
dev = torch.device("cuda:0")
a = torch.randn((K, M) if ta else (M, K), device=dev)
b = torch.randn((N, K) if tb else (K, N), device=dev)
c = torch.zeros(M, N, device=dev)
for _ in range(3):
    ops.gemm(a, b, None, bool(ta), bool(tb), out=c, accumulate=bool(ta))
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ["ASR_LIB_PATH"])
n = 8192
buf = (ctypes.c_ulonglong * n)()
assert L.asr_gemm_dbg_read(buf, n) == 0
d = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2)[:2048]
bar = (d[:, 0] >> np.uint64(32)).astype(np.float64); read = (d[:, 0] & np.uint64(0xffffffff)).astype(np.float64)
mma = (d[:, 1] >> np.uint64(32)).astype(np.float64); nst = ((d[:, 1] >> np.uint64(24)) & np.uint64(0xff)).astype(np.float64)
tot = (d[:, 1] & np.uint64(0xffffff)).astype(np.float64)
sync = np.frombuffer(buf, dtype=np.uint64)[4096:4096 + 2048].astype(np.float64)
ok = nst > 0
print("shape", M, N, K, ta, tb, "waves", ok.sum(), "k-tiles", nst[ok].mean())
print("   of barrier+loop: __syncthreads wait %.0f, global-load issue etc. %.0f" % ((sync[ok] / nst[ok]).mean(), ((bar[ok] - sync[ok]) / nst[ok]).mean()))
print("per k-tile cycles: barrier+loop %.0f  frag-read wait %.0f  mfma+split (after frags) %.0f  | step total %.0f | kernel total per k-tile %.0f" % (
    (bar[ok] / nst[ok]).mean(), (read[ok] / nst[ok]).mean(), ((mma[ok] - read[ok]) / nst[ok]).mean(), ((bar[ok] + mma[ok]) / nst[ok]).mean(), (tot[ok] / nst[ok]).mean()))
