#!/bin/bash
# per-phase cycle tables (tools/dbg_profile.py: run-time instantiation with -DUPR_QP3_PROF) of the three shapes -> gpurun_out/<tag>_phase_cycles.txt
export TMPDIR=/tmp
TAG=${1:-r06}
OUT=gpurun_out/${TAG}_phase_cycles.txt
mkdir -p gpurun_out
{
echo "# tools/dbg_profile.py (upr_batch_qp_profile; run-time instantiation with -DUPR_QP3_PROF): cycles per IPM iteration of one instance as seen by lane 0 of each wave; every counter read costs ~290 cycles"
echo "## headline (B = 1024)"; python tools/dbg_profile.py 1024 2>/dev/null
echo "## headline, the update's parts in slots 6 - 9 (-DUPR_QP3_PROF_FLAT=1)"; UPR_JIT_FLAGS="-DUPR_QP3_PROF_FLAT=1" python tools/dbg_profile.py 1024 2>/dev/null
echo "## configs[3]: upright_robust 8-corner (B = 256)"; python tools/dbg_profile.py 256 config4 2>/dev/null
echo "## configs[2]: box_arch + 20 collision rows (B = 256, cold: 30 IPM iterations)"; python tools/dbg_profile.py 256 config3 2>/dev/null
} > $OUT
cat $OUT
