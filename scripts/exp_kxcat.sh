for k in 0 1 2; do
ASR_KXCAT=$k python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('kxcat=$k', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phases_ms_per_step'].items()})"
ASR_KXCAT=$k python bench.py --no-cpu-baseline --mode fwd 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('  fwd', round(d['ms_per_step'],3))"
done
