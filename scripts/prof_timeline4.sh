cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl4 -o tl -- python3 $R/bench.py --config 4 --no-cpu-baseline --steps 6 --warmup 3 > $R/gpurun_out/prof_tl4.log 2>&1
