# Per-kernel times of the beam search (config 5): step-by-step loop (ASR_BEAM_PERSIST=0) and the persistent launch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ASR_BEAM_PERSIST=${ASR_BEAM_PERSIST:-0}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_beam -o b -- python3 $R/scripts/bench_beam.py > $R/gpurun_out/prof_beam.log 2>&1
find $R/gpurun_out/prof_beam -name "*_kernel_trace.csv" -delete
