#!/bin/bash
# sample clocks and power while the device loop runs
python bench.py --steps 30000 --warmup 10 --no-cpu --no-ml > gpurun_out/clk_bench.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | head -8
  echo "--"
  sleep 2
done
wait $BP
echo idle:
sleep 3
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr -s ' ' | head -6
