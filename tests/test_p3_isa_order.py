"""CPU: the LDS-DMA ordering of gemm_p3_kernel's k-loop, checked on the ISA hipcc emits for gfx950 (scripts/check_p3_isa.py).
The DMA is issued from inline asm, so nothing but the kernel's own counted `s_waitcnt vmcnt` + `s_barrier` orders a fragment read
behind the DMA that fills its stage; the check fails the build if a compiler change ever moves an LDS read or a DMA across them.
(Product arithmetic held by that ordering: the input projections of encoder.py:78-81 in bf16 mode.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_every_barrier_follows_its_counted_wait_and_no_read_or_dma_crosses_it():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_p3_isa.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    assert "instantiations checked" in out
