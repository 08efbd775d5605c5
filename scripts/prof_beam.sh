cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_beam -o b -- python3 $R/scripts/bench_beam.py > $R/gpurun_out/prof_beam.log 2>&1
find $R/gpurun_out/prof_beam -name "*_kernel_trace.csv" -delete
