# a long soak at the round's HEAD: gpurun_out/r05_soak_long.log (every run appends its progress lines directly: a run behind a
# pipe into tail looks hung to gpurun's silence watchdog)
L=gpurun_out/r05_soak_long.log; : > $L
run() { echo "## $*" >> $L; timeout -k 10 700 "$@" 2>&1 | grep --line-buffered -v amdgpu.ids | awk 'NR % 200 == 0 || /ok|fail|Error|error/ { print; fflush() }' >> $L; }
run python scripts/soak.py 171 60000
run python scripts/soak.py 172 30000 bf16
run python scripts/soak_multi.py 131 12000 2 900
export ASR_LIB_VARIANT=hunt
echo "## ---- race-hunt debug library" >> $L
run python scripts/soak.py 173 8000
run python scripts/soak_multi.py 132 2000 2 900
grep -v "^step" $L
