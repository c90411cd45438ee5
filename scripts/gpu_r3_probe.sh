#!/bin/bash
# timing probes: the step without (some of) the weight-gradient GEMMs - how much of the step is theirs
set -o pipefail
OUT=gpurun_out/${TAG:-r3probe}
mkdir -p $OUT
for v in "" all late early; do
  DDL_PROBE_SKIP_WGRAD=$v timeout -k 10 200 python -u bench.py --steps 20 --warmup 6 > $OUT/skip_${v:-none}.log 2>&1 || { tail -20 $OUT/skip_${v:-none}.log; exit 1; }
  echo "skip=${v:-none} $(tail -1 $OUT/skip_${v:-none}.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["final_loss"])')"
done
