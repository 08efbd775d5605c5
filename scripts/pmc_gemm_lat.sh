#!/bin/bash
# Where do the split3 GEMM's waves wait?  Average VMEM / LDS latency (LEVEL / INSTS) and the wait buckets, one pass per group.
OUT=${1:-gpurun_out/pmc_gemm_lat}
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1 || true
RP="rocprofv3 --kernel-trace --output-format csv"
for shape in "12800 1024 1024 0 0" "1024 2048 12800 1 0"; do
  tag=$(echo $shape | tr ' ' '_')
  $RP --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d "$OUT/vmem_$tag" -o g -- python3 scripts/gemm_one.py $shape > "$OUT/vmem_$tag.log" 2>&1
  $RP --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$OUT/lds_$tag" -o g -- python3 scripts/gemm_one.py $shape > "$OUT/lds_$tag.log" 2>&1
  $RP --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT/issue_$tag" -o g -- python3 scripts/gemm_one.py $shape > "$OUT/issue_$tag.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections,os
out=sys.argv[1]
for d in sorted(glob.glob(out+"/*/")):
    agg=collections.defaultdict(lambda:[0,0.0,0.0])
    for f in glob.glob(d+"**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_planes" not in r["Kernel_Name"]: continue
            a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"]); a[2]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    if agg:
        print(os.path.basename(d.rstrip("/")), {k:round(v[1]/v[0],1) for k,v in agg.items()}, "avg_us", round(next(iter(agg.values()))[2]/next(iter(agg.values()))[0]/1e3,1))
PY
