cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_loop -o tl -- python3 $R/scripts/bench_train_loop.py 40 mem > $R/gpurun_out/prof_loop.log 2>&1
