cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_final -o s -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/stats_final.log 2>&1
find $R/gpurun_out/stats_final -name "*_kernel_trace.csv" -delete
