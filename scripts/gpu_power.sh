#!/bin/bash
# clocks and power while the flagship runs (is the step power-limited?)
set -o pipefail
OUT=gpurun_out/${TAG:-power}
mkdir -p $OUT
rocm-smi --showclocks --showpower --showtemp > $OUT/idle.txt 2>&1
timeout -k 10 200 python -u bench.py --steps 1500 --warmup 6 > $OUT/bench.log 2>&1 &
BP=$!
sleep 10
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp > $OUT/load_$i.txt 2>&1
  sleep 1
done
wait $BP
tail -1 $OUT/bench.log | cut -c1-250
grep -h -i "sclk\|mclk\|fclk\|power\|Temp" $OUT/idle.txt | head -12
echo ---
for i in 1 3 5; do grep -h -i "sclk\|mclk\|fclk\|power\|junction" $OUT/load_$i.txt | head -8; echo --; done
