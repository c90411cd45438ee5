#!/bin/bash
# quick bench: N alternating runs of the current build; extra env through AB_ENV (applied to the B runs)
OUT=gpurun_out/ab
mkdir -p $OUT
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/a$i.log 2> $OUT/a$i.err || { tail -20 $OUT/a$i.err; exit 1; }
python - <<PY
import json; d=json.loads(open("$OUT/a$i.log").read().strip().splitlines()[-1]); print("A$i", d["ms_per_step"], d["host_issue_ms_per_step"], d["final_loss"])
PY
if [ -n "$AB_ENV" ]; then
env $AB_ENV timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/b$i.log 2> $OUT/b$i.err || { tail -20 $OUT/b$i.err; exit 1; }
python - <<PY
import json; d=json.loads(open("$OUT/b$i.log").read().strip().splitlines()[-1]); print("B$i ($AB_ENV)", d["ms_per_step"], d["host_issue_ms_per_step"], d["final_loss"])
PY
fi
done
