#!/bin/bash
# HBM traffic of the training step (rocprofv3 FETCH_SIZE / WRITE_SIZE summed over all kernels of a
# 4-step run), for the default build and for an A/B variant given as VARIANT_ENV.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/traffic
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run_pass () {
  local name=$1 ctr=$2; shift 2
  env "$@" timeout -s KILL 300 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/$name" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run_pass new_fetch FETCH_SIZE X=1 || exit 1
run_pass new_write WRITE_SIZE X=1 || exit 1
run_pass old_fetch FETCH_SIZE ${VARIANT_ENV:-DDL_BN3_LINEAR=0} || exit 1
run_pass old_write WRITE_SIZE ${VARIANT_ENV:-DDL_BN3_LINEAR=0} || exit 1
cd $REPO
python - <<'PY'
import csv, glob, collections
def total(name, ctr):
    f = glob.glob("gpurun_out/traffic/%s/**/*counter_collection.csv" % name, recursive=True)[0]
    tot = 0.0; per = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            v = float(r["Counter_Value"]); tot += v; per[r["Kernel_Name"][:70]] += v
    return tot, per
for tag in ("new", "old"):
    f, pf = total(tag + "_fetch", "FETCH_SIZE"); w, pw = total(tag + "_write", "WRITE_SIZE")
    print("%s: FETCH %.2f GB  WRITE %.2f GB  per step (4 steps incl. warm-up): %.2f GB" % (tag, f / 1e6, w / 1e6, (f + w) / 1e6 / 4))
    if tag == "new":
        top = sorted(pf.items(), key=lambda kv: -(kv[1] + pw.get(kv[0], 0)))[:12]
        for k, v in top:
            print("   %-72s fetch %.2f write %.2f GB/step" % (k, v / 1e6 / 4, pw.get(k, 0) / 1e6 / 4))
PY
