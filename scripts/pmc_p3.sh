#!/bin/bash
# PMC passes over one P3 GEMM (scripts/p3_one.py): MFMA busy, wait buckets, LDS conflicts.  usage: pmc_p3.sh <outdir> kk|rr M N K
OUT=$1; shift
mkdir -p "$OUT"; export TMPDIR=/tmp
RP="rocprofv3 --kernel-trace --output-format csv"
$RP --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d "$OUT/sq" -o p -- python3 scripts/p3_one.py "$@" > "$OUT/sq.log" 2>&1
$RP --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -d "$OUT/lds" -o p -- python3 scripts/p3_one.py "$@" > "$OUT/lds.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
csv.field_size_limit(1 << 30)
for pd in ("sq", "lds"):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
    for f in glob.glob(os.path.join(sys.argv[1], pd, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0][-70:]
            if "gemm_p3_kernel" not in n: continue
            a = agg[n][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for n, cs in agg.items():
        c = {k: v[1] / v[0] for k, v in cs.items()}
        ns = next(iter(cs.values())); dur = ns[2] / ns[0]
        print(pd, n, "launches", ns[0], "avg_us %.1f" % (dur / 1e3))
        for k, v in sorted(c.items()): print("    %-34s %.4g" % (k, v))
        if "GRBM_GUI_ACTIVE" in c:
            clk = c["GRBM_GUI_ACTIVE"] / 8 / dur
            print("    effective clock GHz %.3f" % clk)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c: print("    MFMA busy fraction %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)))
        if "SQ_WAVE_CYCLES" in c:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC"):
                if k in c: print("    %s / WAVE_CYCLES %.3f" % (k, c[k] / c["SQ_WAVE_CYCLES"]))
        if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]: print("    bank conflict / idx active %.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]))
PY
