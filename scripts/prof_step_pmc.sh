#!/bin/bash
# Hardware counters for EVERY kernel of the flagship training step (each kernel
# "justified with rocprof counters"): three rocprofv3 --pmc passes over bench.py (SQ block,
# FETCH_SIZE, WRITE_SIZE + L2 hit/miss; never combined with tracing).  Condense with
#   python scripts/pmc_summary.py gpurun_out/pmc_step > profiles/<name>.md
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_step
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run_pass () {
  local name=$1; shift
  timeout -s KILL 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 > "$OUT/$name.log" 2>&1
  echo "pass $name done"
}
run_pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
