# kernel timeline of the char-LM train step (scripts/bench_lm.py) -> gpurun_out/prof_lm/tl_results.db (read with scripts/timeline.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_lm -o tl -- python3 $R/scripts/bench_lm.py > $R/gpurun_out/prof_lm.log 2>&1
