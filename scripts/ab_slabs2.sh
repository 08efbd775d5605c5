mkdir -p gpurun_out/r05g
for i in 1 2; do for m in 0 2; do
ASR_WGRAD_SLABS=$m python bench.py --no-cpu-baseline > gpurun_out/r05g/bench_c2_slabs$m.$i.json 2>> gpurun_out/r05g/bench.err
ASR_WGRAD_SLABS=$m python bench.py --no-cpu-baseline --config 3 > gpurun_out/r05g/bench_c3_slabs$m.$i.json 2>> gpurun_out/r05g/bench.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05g/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['ms_per_step'],3), round(d['step_ms_median'],3), 'tail', round(d.get('side_stream_tail_ms_median',0),3), 'tn', round(d['roofline_gemm_tn']['achieved'],1), {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})
PY
