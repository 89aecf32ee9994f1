#!/bin/bash
# per-phase cycle tables of the production QP kernel for the three shapes (profiles/<tag>_phase_cycles.txt)
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out
{
  echo "# tools/dbg_profile.py (upr_batch_qp_profile): cycles per IPM iteration of one instance as seen by lane 0 of each wave; every counter read costs ~290 cycles"
  echo "# matrix sweep (two-wave form): wave 0 = columns (factorisation, V, K, fused predictor vector sweep), wave 1 = blocks of P, waves 2-3 = Vc of the next knot / shares of Vc'Vc (multi-body shapes)"
  echo "#   'mat: phase 1' = up to barrier A, 'aug. Cholesky' = A .. before B (wave 0: factorisation; wave 1: P+ b partial sums, A'P+A + Q~ + Vc'Vc), 'V, K store' = wait at B, 'P update' = B .. end of knot"
  echo "## headline (B = 1024)"; python3 tools/dbg_profile.py 1024 256 2>&1 | grep -v "amdgpu.ids\|a phase ends"
  echo "## configs[3]: upright_robust 8-corner (B = 256)"; python3 tools/dbg_profile.py 256 256 config4 2>&1 | grep -v "amdgpu.ids\|a phase ends"
  echo "## configs[2]: box_arch + 20 collision rows (B = 256, cold: 30 IPM iterations)"; python3 tools/dbg_profile.py 256 256 config3 2>&1 | grep -v "amdgpu.ids\|a phase ends"
} > gpurun_out/${TAG}_phase_cycles.txt
cat gpurun_out/${TAG}_phase_cycles.txt
