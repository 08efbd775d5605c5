"""Mechanical check of the LDS-DMA ordering in gemm_p3_kernel's k-loop, on the ISA hipcc emits (no GPU needed).

Why it is needed: the LDS-DMA (`global_load_lds_dwordx4`) is issued from inline asm, so the COMPILER does not know that it writes
LDS (csrc/gemm_p3.hip p3_glds16: with the builtin hipcc drained the DMA queue in front of every LDS read).  The only things that
order a fragment read behind the DMA that fills its stage are therefore the kernel's own: a counted `s_waitcnt vmcnt(N)` (this
wave's DMAs of tile t+1 have landed: vector-memory operations retire in issue order, MI355X_MICROARCH.md "s_waitcnt vmcnt") and
the raw `s_barrier` behind it (every other wave's have).  That holds if and only if, in the emitted code,
  (1) every `s_barrier` of the k-loop has an `s_waitcnt vmcnt(..)` in front of it with NO LDS read and NO DMA between the two, and
  (2) no fragment read of tile t+1 was scheduled ABOVE the barrier of step t (into step t-1), and no DMA of step t above it
      (it would overwrite the stage tile t-1's readers may still be reading): between two consecutive barriers of the steady loop
      there are exactly ND fragment reads (x2 for the transposing 64-bit reads of the RR form), JW DMAs and NM MFMAs.
The asm statements carry "memory" clobbers and `__builtin_amdgcn_s_barrier` orders memory operations, so the compiler may not move an
LDS read across them -- this script checks that it did not, for every product instantiation, each time the library is built.

Usage: python scripts/check_p3_isa.py [gemm_p3.s]   (without an argument it compiles csrc/gemm_p3.hip to assembly in a temp dir)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly(path=None):
    if path:
        return open(path).read()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "gemm_p3.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "gemm_p3.hip", "-o", out],
                       cwd=os.path.join(ROOT, "e2e_asr_amd", "csrc"), check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def functions(asm):
    """{demangled template arguments: [instruction lines]} of every gemm_p3_kernel instantiation."""
    out, cur, name = {}, None, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w*gemm_p3_kernel\w*):", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE).stdout.decode().strip()
            cur = []
            out[name] = cur
            continue
        if cur is not None:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                cur = None
                continue
            cur.append(line)
    return out


def classify(line):
    s = line.strip()
    if not s or s.startswith(";") or s.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", s):
            return "L", s.split(":")[0]
        return None
    op = s.split()[0]
    if op == "s_barrier":
        return "B", s
    if op == "s_waitcnt" and "vmcnt" in s:
        return "W", s
    if op.startswith("global_load_lds"):
        return "D", s
    if op.startswith("ds_read"):
        return "R", s
    if op.startswith("v_mfma"):
        return "M", s
    if op.startswith("s_cbranch") or op == "s_branch":
        return "J", s
    return None


def check(name, lines):
    m = re.search(r"gemm_p3_kernel<(\w+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", name)
    RR, NP, KS, WM, WN, DBG, S = m.group(1) == "true", *map(int, m.groups()[1:])
    if DBG:
        return None
    NW, BM, BN = WM * WN, 64 * WM, 128 * WN
    if RR:
        PA, PB = 16 * KS * (BM // 8 * NP), 16 * KS * (BN // 8 * NP)
    else:
        PA, PB = BM * 2 * KS * NP, BN * 2 * KS * NP
    JW = PA // 64 // NW + PB // 64 // NW
    ND = 6 * NP * KS * (2 if RR else 1)
    NM = 8 * (6 if NP == 3 else 3 if NP == 2 else 1) * KS
    ev = [c for c in (classify(l) for l in lines) if c]
    # the steady loop: the backward branch whose body holds two barriers, JW * 2 DMAs
    labels = {e[1]: i for i, e in enumerate(ev) if e[0] == "L"}
    loops = []
    for i, e in enumerate(ev):
        if e[0] == "J":
            tgt = e[1].split()[-1]
            if tgt in labels and labels[tgt] < i:
                body = ev[labels[tgt]:i]
                if sum(1 for b in body if b[0] == "B") == 2 and sum(1 for b in body if b[0] == "D") == 2 * JW:
                    loops.append(body)
    assert len(loops) == 1, "%s: expected one steady loop with two barriers, found %d" % (name, len(loops))
    body = [e for e in loops[0] if e[0] in "BWDRM"]
    # rotate so that the body starts at its first barrier; segments = [barrier, next barrier)
    b0 = next(i for i, e in enumerate(body) if e[0] == "B")
    pre, rot = body[:b0], body[b0:] + body[:b0]
    segs, cur = [], None
    for e in rot:
        if e[0] == "B":
            cur = []
            segs.append(cur)
        else:
            cur.append(e[0])
    res = []
    for k, seg in enumerate(segs):
        # (1) the barrier that ENDS this segment: the last memory-ordering event in front of it is the counted wait
        tail = "".join(seg).rstrip("M")
        assert tail.endswith("W"), "%s: segment %d does not end with s_waitcnt vmcnt in front of its barrier: ...%s" % (name, k, tail[-12:])
        nr, nd, nm = seg.count("R"), seg.count("D"), seg.count("M")
        assert (nr, nd) == (ND, JW), "%s: segment %d has %d reads / %d DMAs, expected %d / %d" % (name, k, nr, nd, ND, JW)
        res.append((nr, nd, nm))
    # (MFMAs work on registers: the scheduler may and does move a few across a barrier; only their total is fixed)
    assert sum(r[2] for r in res) == 2 * NM, "%s: %d MFMAs in the loop body, expected %d" % (name, sum(r[2] for r in res), 2 * NM)
    return dict(kernel=name, reads_per_kstep=ND, dmas_per_kstep=JW, mfmas_per_kstep=NM, segments=res)


def main():
    asm = assembly(sys.argv[1] if len(sys.argv) > 1 else None)
    n = 0
    for name, lines in sorted(functions(asm).items()):
        r = check(name, lines)
        if r:
            n += 1
            print("ok  %-60s per k-step: %d fragment reads, %d LDS-DMAs, %d MFMAs; every barrier directly behind its counted vmcnt wait" % (
                name[name.index("<"):name.index(">") + 1], r["reads_per_kstep"], r["dmas_per_kstep"], r["mfmas_per_kstep"]))
    assert n >= 9, "only %d product instantiations found" % n
    print("%d instantiations checked" % n)


if __name__ == "__main__":
    main()
