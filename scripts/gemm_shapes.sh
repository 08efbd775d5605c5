# one train step's GEMM shapes in launch order (ASR_GEMM_LOG) -> gpurun_out/<tag>_shapes.txt, plus the kernel timeline
TAG=${1:-shapes}
R=$GRAFT_REPO_ROOT
ASR_GEMM_LOG=1 python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 2> $R/gpurun_out/${TAG}_raw.txt > /dev/null
python3 - <<PY
lines = [l for l in open("$R/gpurun_out/${TAG}_raw.txt") if l.startswith("gemm ")]
# two steps (warmup + timed) -> keep the last half
n = len(lines) // 2
open("$R/gpurun_out/${TAG}_shapes.txt", "w").writelines(lines[-n:] if n else lines)
PY
bash $R/scripts/prof_tl_env.sh ${TAG}_tl
