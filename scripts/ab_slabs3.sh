mkdir -p gpurun_out/r05g
for i in 1 2 3; do
ASR_WGRAD_SLABS=0 python bench.py --no-cpu-baseline --config 3 > gpurun_out/r05g/c3_a.$i.json 2>> gpurun_out/r05g/bench.err
ASR_WGRAD_SLABS=2 python bench.py --no-cpu-baseline --config 3 > gpurun_out/r05g/c3_b.$i.json 2>> gpurun_out/r05g/bench.err
ASR_WGRAD_SLABS=2 ASR_P3_WGRAD256=1 python bench.py --no-cpu-baseline --config 3 > gpurun_out/r05g/c3_c.$i.json 2>> gpurun_out/r05g/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05g/c3_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['ms_per_step'],3), round(d['step_ms_median'],3), 'tail', round(d.get('side_stream_tail_ms_median',0),3), 'tn', round(d['roofline_gemm_tn']['achieved'],1))
PY
