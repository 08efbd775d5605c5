# kernel timeline of a few train steps under the caller's environment -> gpurun_out/<tag>.txt   usage: prof_tl_env.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$TAG -o tl -- python3 $R/bench.py --no-cpu-baseline --steps 8 --warmup 4 "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
python3 $R/scripts/timeline.py $R/gpurun_out/prof_$TAG/tl_results.db > $R/gpurun_out/$TAG.txt 2>&1
rm -rf $R/gpurun_out/prof_$TAG
