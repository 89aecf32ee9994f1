#!/bin/bash
# Evidence set of the final tree (GPU box, through gpurun from the repo root): tools/r6_evidence.sh <tag>
#   tools/profile_all.sh (bench line, kernel stats, counters per workload) + tools/r6_prof.sh (QP phase cycles) + the phase
#   cycles of the linearisation / line-search kernels (instrumented library of tools/build_prof_lin.sh, when it is there)
TAG=${1:-r06e}
export TMPDIR=/tmp
bash tools/profile_all.sh $TAG > gpurun_out/${TAG}_profile_all.log 2>&1
bash tools/r6_prof.sh $TAG > /dev/null 2>&1
if [ -f upright_amd/libupright_mi_prof.so ]; then
  {
  echo "# tools/dbg_lin.py, tools/dbg_ls.py on upright_amd/libupright_mi_prof.so (tools/build_prof_lin.sh: -DUPR_LIN_PROF -DUPR_LS_PROF): cycles per workgroup,"
  echo "# lane 0's counter reads summed with atomics -- at B = 1024 the atomics of 768 / 1024 workgroups queue up and inflate every phase; B = 128 is the undisturbed view"
  for B in 128 1024; do
    echo "## linearisation kernel, headline shape, B = $B"; UPR_LIB=libupright_mi_prof.so python tools/dbg_lin.py headline $B 2>/dev/null | tail -8
    echo "## line-search kernel, headline shape, B = $B"; UPR_LIB=libupright_mi_prof.so python tools/dbg_ls.py $B 2>/dev/null | tail -13
  done
  echo "## launch times against the batch size (tools/exp_lin_b.py, production library)"; python tools/exp_lin_b.py 2>/dev/null | tail -9
  } > gpurun_out/${TAG}_lin_ls_phases.txt
fi
tail -5 gpurun_out/${TAG}_profile_all.log
head -c 400 gpurun_out/${TAG}_bench.json; echo
cat gpurun_out/${TAG}_lin_ls_phases.txt | head -60
