#!/bin/bash
# round-2 wrap-up session: whole GPU suite, smoke, flagship bench (x3), forced-exchange bench, secondary configs, kernel stats
OUT=gpurun_out/r2final
mkdir -p $OUT
step() {
    local name=$1 tmo=$2; shift 2
    timeout -k 10 "$tmo" "$@" > $OUT/$name.log 2> $OUT/$name.err
    local rc=$?
    echo "$name rc=$rc" | tee -a $OUT/summary.txt
    case $rc in 124|137|134|139) echo "stopping after $name"; tail -20 $OUT/$name.log $OUT/$name.err; exit $rc;; esac
    return $rc
}
step gpu_tests 1100 python -u -m pytest tests -m gpu -q --timeout 600 --timeout-method thread
tail -5 $OUT/gpu_tests.log
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
tail -2 $OUT/smoke.log
for i in 1 2 3; do step bench$i 300 python bench.py --steps 20 --warmup 5; cut -c1-330 $OUT/bench$i.log; done
step bench_default 300 python bench.py
cut -c1-330 $OUT/bench_default.log
HVD_FORCE_COLLECTIVES=1 step bench_forced 300 python bench.py --steps 20 --warmup 5
grep metric $OUT/bench_forced.log | cut -c1-1500
step lars 300 python bench.py --steps 20 --warmup 5 --optimizer lars
cut -c1-200 $OUT/lars.log
for m in resnet18 resnet34 resnet101 resnet152 resnet200; do step $m 400 python bench.py --model $m --steps 8 --warmup 3; cut -c1-330 $OUT/$m.log; done
step b512 400 python bench.py --steps 10 --warmup 3 --batch 512
cut -c1-200 $OUT/b512.log
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -- python3 $R/bench.py --steps 6 --warmup 3 > $R/$OUT/prof.log 2>&1)
echo "prof rc=$?"
python scripts/trace_gaps.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) | head -6
find $OUT/prof -name "*kernel_trace.csv" -size +30M -delete
exit 0
