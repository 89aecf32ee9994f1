#!/bin/bash
# Rebuild upr_api.hip alone (C-ABI, linearisation, line search, generic QP kernels) and relink libupright_mi.so with the
# production QP kernel's objects as they are (upright_amd/csrc/build/upr_qp3_part*.o; __graft_entry__.build_engine makes all).
set -e
cd "$(dirname "$0")/.."
B=upright_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed "$@" -c upright_amd/csrc/upr_api.hip -o $B/upr_api.o 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o upright_amd/libupright_mi.so $B/upr_api.o $B/upr_qp3_part*.o -lhiprtc -ldl
