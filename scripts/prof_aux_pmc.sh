#!/bin/bash
# Hardware counters for the kernels that are not part of the flagship step (K15 LARS, K17 pack,
# K18 xGMI all-reduce in simulate mode, K20 layout conversion, K21 label rank, K19 Philox): the
# GPU tests of those kernels run under rocprofv3 --pmc, one pass per counter group.
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_aux
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
SEL="lars or pack_scale or xgmi or nchw or topk or philox or softmax or avgpool or linear"
run_pass () {
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "$SEL" > "$OUT/$name.log" 2>&1
  echo "pass $name done"
}
run_pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
