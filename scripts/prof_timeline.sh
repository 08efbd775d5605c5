# kernel timeline of a few train steps -> gpurun_out/prof_tl/tl_results.db (read with scripts/timeline.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_tl -o tl -- python3 $R/bench.py --no-cpu-baseline --steps 8 --warmup 4 > $R/gpurun_out/prof_tl.log 2>&1
