# same-box A/B of the second half of round 5: every switch off (the first half's kernels), each on alone, all on (HEAD default)
# -> gpurun_out/r05_second_half_ab.log
L=gpurun_out/r05_second_half_ab.log; : > $L
OFF="ASR_BPTT_QUAD=0 ASR_LSTM_XPRE=0 ASR_LM_DEFER=0 ASR_CHAIN_BWD_WIDE=0"
echo "## config 2 (python bench.py --no-cpu-baseline), ms per step and event-timed phases" >> $L
bash scripts/ab_bench.sh "$OFF" "ASR_BPTT_QUAD=1 ASR_LSTM_XPRE=0 ASR_LM_DEFER=0" "ASR_BPTT_QUAD=0 ASR_LSTM_XPRE=1 ASR_LM_DEFER=0" "ASR_BPTT_QUAD=0 ASR_LSTM_XPRE=0 ASR_LM_DEFER=1" "ASR_BPTT_QUAD=1 ASR_LSTM_XPRE=1 ASR_LM_DEFER=1" >> $L 2>&1
echo "## config 3 (--config 3)" >> $L
BENCH_ARGS="--config 3" bash scripts/ab_bench.sh "$OFF" "ASR_BPTT_QUAD=1 ASR_LSTM_XPRE=1 ASR_LM_DEFER=1" >> $L 2>&1
echo "## config 4 (--config 4)" >> $L
BENCH_ARGS="--config 4" bash scripts/ab_bench.sh "$OFF" "ASR_BPTT_QUAD=1 ASR_LSTM_XPRE=1 ASR_LM_DEFER=1 ASR_CHAIN_BWD_WIDE=0" "ASR_BPTT_QUAD=1 ASR_LSTM_XPRE=1 ASR_LM_DEFER=1 ASR_CHAIN_BWD_WIDE=1" >> $L 2>&1
echo "## recurrent kernels alone (scripts/bench_lstm.py): old mappings, then HEAD" >> $L
ASR_BPTT_QUAD=0 ASR_LSTM_XPRE=0 python scripts/bench_lstm.py 2>&1 | grep "T=" >> $L
python scripts/bench_lstm.py 2>&1 | grep "T=" >> $L
cat $L
