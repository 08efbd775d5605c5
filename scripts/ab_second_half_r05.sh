# same-box A/B of the second half of round 5: every switch off (the first half's kernels), each on alone, all on (HEAD default)
# -> gpurun_out/r05_second_half_ab.log   (env: a later assignment of the same variable wins)
L=gpurun_out/r05_second_half_ab.log; : > $L
OFF="ASR_BPTT_QUAD=0 ASR_LSTM_XPRE=0 ASR_LSTM_GXL=0 ASR_LM_DEFER=0 ASR_CHAIN_BWD_WIDE=0 ASR_CHAIN_BWD_ARED=0 ASR_EXT_EVENTS=0 ASR_DEC_FORK_PRE=1"
echo "## OFF = $OFF" >> $L
echo "## config 2 (python bench.py --no-cpu-baseline), ms per step and phases; rows: OFF, OFF + one switch, HEAD defaults" >> $L
run() { lbl=$1; shift; for rep in 1 2; do env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-34s %.3f ms  %s' % ('$lbl', d['ms_per_step'], {k: round(v, 2) for k, v in d['phases_ms_per_step'].items()}))" >> $L; done; }
run "all off" $OFF
run "+ ASR_BPTT_QUAD=1" $OFF ASR_BPTT_QUAD=1
run "+ ASR_LSTM_XPRE=1" $OFF ASR_LSTM_XPRE=1
run "+ ASR_LSTM_GXL=1" $OFF ASR_LSTM_GXL=1
run "+ ASR_LM_DEFER=1" $OFF ASR_LM_DEFER=1
run "+ ASR_CHAIN_BWD_ARED=1" $OFF ASR_CHAIN_BWD_ARED=1
run "+ ASR_EXT_EVENTS=1 FORK_PRE=0" $OFF ASR_EXT_EVENTS=1 ASR_DEC_FORK_PRE=0
run "HEAD defaults" ASR_NOTHING=1
echo "## config 3 (--config 3)" >> $L
BENCH_ARGS="--config 3"; run "all off" $OFF; run "HEAD defaults" ASR_NOTHING=1
echo "## config 4 (--config 4)" >> $L
BENCH_ARGS="--config 4"; run "all off" $OFF; run "+ ASR_CHAIN_BWD_WIDE=1" $OFF ASR_CHAIN_BWD_WIDE=1; run "+ WIDE=1 ARED=1" $OFF ASR_CHAIN_BWD_WIDE=1 ASR_CHAIN_BWD_ARED=1; run "HEAD defaults" ASR_NOTHING=1
BENCH_ARGS=
echo "## recurrent kernels alone (scripts/bench_lstm.py): old mappings (QUAD, XPRE, GXL off), then HEAD" >> $L
ASR_BPTT_QUAD=0 ASR_LSTM_XPRE=0 ASR_LSTM_GXL=0 python scripts/bench_lstm.py 2>&1 | grep "T=" >> $L
python scripts/bench_lstm.py 2>&1 | grep "T=" >> $L
cat $L
