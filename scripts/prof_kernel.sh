#!/bin/bash
# Counter-backed justification of one kernel shape: runs benchmarks/kernel_bench.py
# for the given shape ids under rocprofv3 in separate passes (PMC slot limits: SQ 8, TCC 4; never
# combined with tracing), then a plain kernel-trace pass.  Outputs CSVs under gpurun_out/pmc_<ids>/.
#   scripts/prof_kernel.sh a3,c4
set -eo pipefail
IDS=${1:-a3}
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/pmc_${IDS//,/_}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run_pass () {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/benchmarks/kernel_bench.py" --only "$IDS" --iters 3 > "$OUT/$name.log" 2>&1
}
run_pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/benchmarks/kernel_bench.py" --only "$IDS" --iters 3 > "$OUT/trace.log" 2>&1
echo "profiles in $OUT"
