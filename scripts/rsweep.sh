for R in 1 2 4 8; do
  ASR_LSTM_R=$R timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --mode train > gpurun_out/r$R.log 2>&1
  tail -1 gpurun_out/r$R.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('R=$R', round(d['ms_per_step'],2), d['phases_ms_per_step'])"
done
