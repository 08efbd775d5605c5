#!/bin/sh
# Build libe2e_asr_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o libe2e_asr_hip.so \
    *.hip $EXTRA_SRCS
