# long runs at HEAD (random shapes, multitask, bf16) and the same under the race-hunt debug library: gpurun_out/r05_soak.log
L=gpurun_out/r05_soak.log; : > $L
run() { echo "## $*" >> $L; timeout -k 10 600 "$@" 2>&1 | grep -v amdgpu.ids | tail -2 >> $L; }
run python scripts/soak.py 71 20000
run python scripts/soak.py 72 12000 bf16
run python scripts/soak_multi.py 31 4000 2 900
run python scripts/soak_fixed.py 30 83 27 4000
export ASR_LIB_VARIANT=hunt
echo "## ---- race-hunt debug library (random ~4 us delays in front of every publish and poll)" >> $L
run python scripts/soak.py 73 1500
run python scripts/soak.py 74 1000 bf16
run python scripts/soak_multi.py 32 300 2 900
cat $L
