# kernel timeline of a few train steps under given environment settings: bash scripts/prof_tl_env.sh <tag> [bench args...]
# -> gpurun_out/prof_tl_<tag>/tl_results.db + gpurun_out/timeline_<tag>.txt.  (rocprofv3 gets the program itself after `--`.)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl_$tag -o tl -- python3 $R/bench.py --no-cpu-baseline --steps 8 --warmup 4 "$@" > $R/gpurun_out/prof_tl_$tag.log 2>&1
cd $R && python3 scripts/timeline.py $(find gpurun_out/prof_tl_$tag -name "tl_results.db" | head -1) > gpurun_out/timeline_$tag.txt 2>&1
find gpurun_out/prof_tl_$tag -name "*.db" -size +40M -delete
