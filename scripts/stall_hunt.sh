#!/bin/bash
# Hunt for the rare ~1 s stall inside the timed loop: many short bench processes with a watchdog
# that dumps every thread's stack when one step takes longer than 0.25 s on the host.
#   scripts/stall_hunt.sh N [tag]     (environment is passed through to bench.py)
set -o pipefail
mkdir -p gpurun_out
N=${1:-40}
tag=${2:-default}
out=gpurun_out/stall_hunt_$tag.log
: > $out
for i in $(seq 1 $N); do
  timeout -k 10 120 python -u bench.py --stall-watch 0.25 > gpurun_out/stall_one.log 2>&1 || { echo "run $i failed" | tee -a $out; tail -20 gpurun_out/stall_one.log; exit 1; }
  ms=$(tail -1 gpurun_out/stall_one.log | grep -o 'ms_per_step": [0-9.]*' | head -1)
  echo "run $i $ms" >> $out
  if grep -q "stall-watch" gpurun_out/stall_one.log; then
    echo "==== stall in run $i ====" >> $out
    grep -v amdgpu.ids gpurun_out/stall_one.log | head -60 >> $out
  fi
done
echo "$tag: $(grep -c '==== stall' $out) stalls in $N runs"
