#!/bin/bash
# collision rows of the linearisation kernel: passes per workgroup; line search: staging and ranking A/B
export TMPDIR=/tmp
for rp in 1 2 3; do for c in config3 config5; do echo -n "ROW_PASSES=$rp "; UPR_LIN_ROW_PASSES=$rp timeout 300 python tools/r4_lin.py $c 2>&1 | tail -1; done; done
for c in config3 config4; do
  echo -n "order on,  stage on : "; timeout 300 python tools/r4_lin.py $c 2>&1 | tail -1
  echo -n "order OFF, stage on : "; UPR_QP_ORDER=0 timeout 300 python tools/r4_lin.py $c 2>&1 | tail -1
  echo -n "order on,  stage OFF: "; UPR_LS_STAGE_FULL=0 timeout 300 python tools/r4_lin.py $c 2>&1 | tail -1
done
