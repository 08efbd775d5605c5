# everything the round's write-up quotes, on ONE box at HEAD: gpurun_out/final_r05/
O=gpurun_out/final_r05; mkdir -p $O
python bench.py > $O/bench_train.json 2> $O/bench_train.err
python bench.py --config 3 > $O/bench_config3.json 2> $O/bench_config3.err
python bench.py --config 4 > $O/bench_config4.json 2> $O/bench_config4.err
python bench.py --config 5 > $O/bench_config5.json 2> $O/bench_config5.err
bash scripts/bench_modes.sh > $O/bench_modes.log 2>&1
ASR_WGRAD_SLABS=1 python bench.py --no-cpu-baseline > $O/bench_train_deterministic.json 2>/dev/null
bash scripts/prof_tl_env.sh c2 && cp gpurun_out/timeline_c2.txt $O/timeline_train_step.txt
bash scripts/prof_tl_env.sh c3 --config 3 && cp gpurun_out/timeline_c3.txt $O/timeline_config3.txt
bash scripts/prof_tl_env.sh c4 --config 4 && cp gpurun_out/timeline_c4.txt $O/timeline_config4.txt
python scripts/host_time.py > $O/host_time.txt 2>&1
python scripts/bench_lstm.py > $O/bench_lstm.txt 2>&1
python scripts/bench_gemm.py > $O/gemm_bench.log 2>&1
bash scripts/prof_stats.sh && cp $(find gpurun_out/stats_final -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
cat $O/bench_modes.log
