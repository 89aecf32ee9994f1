#!/bin/bash
# Instrumented build of the engine for the matrix-sweep work/wait table (upr_qp3.h, UPR_QP3_PROF_MAT).  It goes to
# its own file and is loaded only with UPR_LIB=libupright_mi_prof.so; the production library is not touched.
#   tools/build_prof.sh && gpurun -- 'UPR_LIB=libupright_mi_prof.so python tools/dbg_profile.py 1024 256 mat'
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DUPR_MONOLITHIC -DUPR_QP3_PROF -DUPR_QP3_PROF_MAT \
    -o upright_amd/libupright_mi_prof.so upright_amd/csrc/upr_api.hip -lhiprtc -ldl
ls -l upright_amd/libupright_mi_prof.so
