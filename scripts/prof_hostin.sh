cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/prof_hostin -o hi -- python3 $R/bench.py --no-cpu-baseline --host-input --steps 20 --warmup 5 > $R/gpurun_out/prof_hostin.log 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/prof_res -o res -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $R/gpurun_out/prof_res.log 2>&1
ls $R/gpurun_out/prof_hostin/*
