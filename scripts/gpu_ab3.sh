#!/bin/bash
# alternate several env variants on one box: VARIANTS="A=1;B=2 C=3;..." (semicolon separated, each a space separated env list)
OUT=gpurun_out/ab3
mkdir -p $OUT
IFS=';' read -ra VS <<< "$VARIANTS"
for round in 1 2; do
  i=0
  for v in "${VS[@]}"; do
    i=$((i+1))
    env $v timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/v$i.log 2> $OUT/v$i.err || { tail -5 $OUT/v$i.err; continue; }
    python - <<PY
import json; d=json.loads(open("$OUT/v$i.log").read().strip().splitlines()[-1]); print("round $round [$v]", d["ms_per_step"], d["final_loss"])
PY
  done
done
