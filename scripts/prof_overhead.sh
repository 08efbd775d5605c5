# what the HIP events of bench.py's phase timing cost the timed region (one box): none / the roofline family only (default) / all
# -> gpurun_out/r05_bench_prof_overhead.txt
L=gpurun_out/r05_bench_prof_overhead.txt; : > $L
for rep in 1 2 3; do for v in none "" all; do
  ASR_BENCH_PROF=$v python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('events in the timed region: %-28s ms_per_step %.3f  median step %.3f' % ('$v' or 'roofline family only (default)', d['ms_per_step'], d['step_ms_median']))" >> $L
done; done
cat $L
