# a long soak at the round's HEAD: gpurun_out/r05_soak_long.log
L=gpurun_out/r05_soak_long.log; : > $L
run() { echo "## $*" >> $L; timeout -k 10 1000 "$@" 2>&1 | grep -v amdgpu.ids | tail -1 >> $L; }
run python scripts/soak.py 171 100000
run python scripts/soak.py 172 40000 bf16
run python scripts/soak_multi.py 131 15000 2 900
export ASR_LIB_VARIANT=hunt
echo "## ---- race-hunt debug library" >> $L
run python scripts/soak.py 173 8000
run python scripts/soak_multi.py 132 2000 2 900
cat $L
