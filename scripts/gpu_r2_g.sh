#!/bin/bash
# round-2 GPU session G: launcher rehearsals (torchrun N=1; N=2 on one GPU through the xGMI kernel), examples, deep nets
OUT=gpurun_out/r2g
mkdir -p $OUT
step() {
    local name=$1 tmo=$2; shift 2
    timeout -k 10 "$tmo" "$@" > $OUT/$name.log 2> $OUT/$name.err
    local rc=$?
    echo "$name rc=$rc" | tee -a $OUT/summary.txt
    case $rc in 124|137|134|139) echo "stopping after $name"; tail -20 $OUT/$name.log $OUT/$name.err; exit $rc;; esac
    return $rc
}
step tr_n1 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3
grep metric $OUT/tr_n1.log | cut -c1-300; tail -3 $OUT/tr_n1.err
DDL_RCCL_SKIP_COMM=1 HVD_ALLREDUCE_ALGO=twoshot DDL_COMM_STREAM_PRIORITY=low HVD_COLLECTIVE_TIMEOUT_S=60 \
  step tr_n2 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --batch 64
grep metric $OUT/tr_n2.log | cut -c1-1200; tail -5 $OUT/tr_n2.err | cut -c1-300
step examples 900 bash scripts/gpu_examples.sh
tail -12 $OUT/examples.log
step r152 300 python bench.py --model resnet152 --steps 6 --warmup 3
grep metric $OUT/r152.log | cut -c1-330
exit 0
