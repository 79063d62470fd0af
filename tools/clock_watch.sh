#!/bin/bash
# Sample GPU clock and power with rocm-smi while bench.py runs (one render of configs[2] takes ~6 s).
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
BP=$!
for i in $(seq 1 14); do
  sleep 2
  echo "--- t=$((i*2))s"; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|Average Graphics\|Socket" | head -6
done
wait $BP
cut -c1-160 gpurun_out/clock_bench.json
