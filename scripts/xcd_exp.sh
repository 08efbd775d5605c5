for v in 0 1; do
  if [ $v = 1 ]; then export ASR_LSTM_FASTXCD=1; else unset ASR_LSTM_FASTXCD; fi
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/x$v.log 2>&1
  tail -1 gpurun_out/x$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fast=$v', round(d['ms_per_step'],2), d['phases_ms_per_step'])"
  timeout -k 10 200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "lstm_layer_fwd" 2>&1 | tail -1
done
