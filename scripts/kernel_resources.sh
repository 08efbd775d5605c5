#!/bin/sh
# Register / scratch / LDS report of every kernel of one source: scripts/kernel_resources.sh lstm_bwd.hip [extra hipcc flags]
cd "$(dirname "$0")/../e2e_asr_amd/csrc"
src=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$src" -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | \
  awk '/remark: Function Name:/ {name=$5} /remark: +VGPRs:/ {v=$4} /remark: +AGPRs:/ {a=$4} /ScratchSize/ {s=$5} /Occupancy/ {o=$5} /LDS Size/ {print name, "VGPR", v, "AGPR", a, "scratch", s, "occ", o, "LDS", $6}' | \
  c++filt | sed 's/asr:://g; s/(asr::[A-Za-z]*)//; s/(.*Args)//' | sort
rm -f /tmp/kr_$$.o
