#!/bin/bash
# round 6: SQ counters of the LDS stage machine at a few lengths (tools/pmc_sq.py, two --pmc passes)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
for n in 592 1110 2368; do
  python "$GRAFT_REPO_ROOT/tools/pmc_sq.py" --sweep --dtype f32 --sizes $n > "$GRAFT_REPO_ROOT/gpurun_out/r6/pmc_sq_lsm_f32_$n.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r6/pmc_sq_lsm_f32_$n.err"
  cat "$GRAFT_REPO_ROOT/gpurun_out/r6/pmc_sq_lsm_f32_$n.json"
done
