#!/bin/bash
# Instrumented build of the linearisation / line-search kernels (phase stamps: -DUPR_LIN_PROF -DUPR_LS_PROF) into
# upright_amd/libupright_mi_prof.so: upr_api.hip alone is recompiled, the production QP kernel's objects are linked as built.
#   tools/build_prof_lin.sh && gpurun -- 'UPR_LIB=libupright_mi_prof.so python tools/dbg_ls.py; UPR_LIB=libupright_mi_prof.so python tools/dbg_lin.py'
set -e
cd "$(dirname "$0")/.."
B=upright_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -DUPR_LIN_PROF -DUPR_LS_PROF "$@" -c upright_amd/csrc/upr_api.hip -o $B/upr_api_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o upright_amd/libupright_mi_prof.so $B/upr_api_prof.o $B/upr_qp3_part*.o -lhiprtc -ldl
ls -l upright_amd/libupright_mi_prof.so
