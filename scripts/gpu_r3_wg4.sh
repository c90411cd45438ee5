#!/bin/bash
# kernel-level times (rocprofv3) of the weight-gradient GEMM and its slab reduction, per shape
set -o pipefail
OUT=gpurun_out/${TAG:-r3wg4}
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in c4 c3 d4 d5 c1; do
for cfg in "0 256" "1 256" "1 160"; do
  set -- $cfg
  (cd /tmp && DDL_WGRAD_BIG=$1 DDL_WGRAD_BIG_TARGET=$2 DDL_WGRAD_BIG_PIPE=1 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${shape}_$1_$2 -o k -- python3 $R/benchmarks/kernel_bench.py --iters 10 --only $shape > $R/$OUT/p.log 2>&1) || { tail -5 $OUT/p.log; exit 1; }
  echo "$shape big=$1 target=$2: $(python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/p_${shape}_$1_$2/k_kernel_stats.csv")))
out=[]
for r in rows:
    n=r["Name"]
    if "wgrad" in n or "splitk" in n:
        out.append("%s x%s avg %.1f us"%(n.split("(")[0][-40:],r["Calls"],float(r["AverageNs"])/1e3))
print("; ".join(out))
PY
)"
done
done
