# several long soaks (different seeds), one log per seed; prints which ones failed and how
mkdir -p gpurun_out/soak
for seed in "$@"; do
  timeout -k 10 300 python scripts/soak.py $seed 4000 > gpurun_out/soak/s$seed.log 2>&1
  echo "seed $seed rc=$? : $(tail -1 gpurun_out/soak/s$seed.log | cut -c1-160)"
done
