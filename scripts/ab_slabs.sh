mkdir -p gpurun_out/r05e
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity3.py tests/test_gpu_gemm_p3.py -x -q -m gpu > gpurun_out/r05e/tests.log 2>&1
tail -8 gpurun_out/r05e/tests.log
for i in 1 2; do
for m in 1 0; do
ASR_WGRAD_SLABS=$m python bench.py --no-cpu-baseline > gpurun_out/r05e/bench_c2_slabs$m.$i.json 2>> gpurun_out/r05e/bench.err
ASR_WGRAD_SLABS=$m python bench.py --no-cpu-baseline --config 3 > gpurun_out/r05e/bench_c3_slabs$m.$i.json 2>> gpurun_out/r05e/bench.err
done; done
ASR_WGRAD_SLABS=1 python bench.py --no-cpu-baseline --config 4 > gpurun_out/r05e/bench_c4_slabs1.json 2>> gpurun_out/r05e/bench.err
ASR_WGRAD_SLABS=0 python bench.py --no-cpu-baseline --config 4 > gpurun_out/r05e/bench_c4_slabs0.json 2>> gpurun_out/r05e/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['ms_per_step'],3), round(d['step_ms_median'],3), 'tail', round(d.get('side_stream_tail_ms_median',0),3), 'tn', round(d['roofline_gemm_tn']['achieved'],1))
    except Exception as e: print(f, 'ERR', e)
PY
