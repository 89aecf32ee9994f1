#!/bin/bash
# Experiment build of the headline kernel only (-DUPR_HEADLINE_ONLY: ~45 s instead of 2.5 min) into libupright_mi_exp.so;
# extra -D flags are passed through.  Run with UPR_LIB=libupright_mi_exp.so.  Never the production library.
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DUPR_MONOLITHIC -DUPR_HEADLINE_ONLY "$@" \
    -o upright_amd/${OUT:-libupright_mi_exp.so} upright_amd/csrc/upr_api.hip -lhiprtc -ldl 2>&1 | grep -E "error" || true
ls -l upright_amd/${OUT:-libupright_mi_exp.so}
