#!/bin/bash
# rocprofv3 evidence for profiles/ (rounds 2, 3).  One counter group per pass (MI355X_MICROARCH.md: SQ 8 slots, TCC 4 -- FETCH_SIZE
# and WRITE_SIZE do not fit one pass; GRBM independent), --kernel-trace only, the program directly after `--`.
# usage (on the GPU box, from the repo root):  bash scripts/pmc_collect.sh <outdir under gpurun_out>
set -e
OUT=${1:-gpurun_out/pmc_r04}
mkdir -p "$OUT"
export TMPDIR=/tmp
RP="rocprofv3 --kernel-trace --output-format csv"
# (A) GEMM: MFMA busy / issue stalls, split3 and exact fp32 kernels on the layer-3 input projection (6400x1024x1024: small
#     enough that the 32-bit per-dispatch counters do not saturate)
ASR_GEMM_SPLIT=1 $RP --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d "$OUT/gemm_split_sq" -o g -- python3 scripts/gemm_one.py 6400 1024 1024 0 0 > "$OUT/gemm_split_sq.log" 2>&1
ASR_GEMM_SPLIT=0 $RP --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d "$OUT/gemm_exact_sq" -o g -- python3 scripts/gemm_one.py 6400 1024 1024 0 0 > "$OUT/gemm_exact_sq.log" 2>&1
ASR_GEMM_SPLIT=1 $RP --pmc FETCH_SIZE -d "$OUT/gemm_split_fetch" -o g -- python3 scripts/gemm_one.py 6400 1024 1024 0 0 > "$OUT/gemm_split_fetch.log" 2>&1
ASR_GEMM_SPLIT=1 $RP --pmc WRITE_SIZE -d "$OUT/gemm_split_write" -o g -- python3 scripts/gemm_one.py 6400 1024 1024 0 0 > "$OUT/gemm_split_write.log" 2>&1
# (B) the persistent recurrent pair: where the wave cycles go (parked in waits vs issuing)
$RP --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE -d "$OUT/lstm_sq" -o l -- python3 scripts/pmc_lstm.py 400 3 > "$OUT/lstm_sq.log" 2>&1
# (C) HBM traffic per kernel over whole train steps (separate FETCH / WRITE passes)
$RP --pmc FETCH_SIZE -d "$OUT/step_fetch" -o s -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/step_fetch.log" 2>&1
$RP --pmc WRITE_SIZE -d "$OUT/step_write" -o s -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/step_write.log" 2>&1
# (D) kernel time statistics of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/step_stats" -o s -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/step_stats.log" 2>&1
find "$OUT" -name "*_kernel_trace.csv" -size +20M -delete
ls -R "$OUT" | head -60
