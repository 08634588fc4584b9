#!/bin/bash
# usage: build_lab.sh name "-DFLAGS"   -> tools/exp/lab_<name>  (cross-compiled for gfx950)
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -I../../include -I../../speaker_follower_amd/csrc $2 gemm_lab.hip -o lab_$1 2>&1 | grep -E "error" -A5 | head -10
