run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$*', round(d['ms_per_step'],3), round(d['value']), {k:round(v,3) for k,v in d['phases_ms_per_step'].items()}, 'gemm', round(d['roofline_gemm']['achieved'],1))"; }
run
run --host-input
run --dtype bf16
run --variable-len
run --mode eval
run --mode fwd
run --gemm exact
