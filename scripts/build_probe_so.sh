#!/bin/bash
# build a second kernel library with extra -D flags (timing probes) next to the product one:
#   scripts/build_probe_so.sh benchmarks/native/_ddl_kernels_probe.so -DDDL_CROWS_PROBE
# and load it with DDL_KERNELS_SO=<path>
set -e
OUT=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/probe
objs=""
for s in $R/csrc/kernels/*.hip; do
  o=$R/build/probe/$(basename $s).o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-math-errno -ffp-contract=fast -fno-gpu-sanitize -Wno-unused-result -I $R/csrc/kernels "$@" -c $s -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT $objs
echo built $OUT
