for sk in default 2 3 4 12; do
  if [ $sk = default ]; then unset ASR_GEMM_SPLITK; else export ASR_GEMM_SPLITK=$sk; fi
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('splitk=$sk', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phases_ms_per_step'].items()}, round(d['roofline']['us_per_recurrent_step'],3))"
done
