#!/bin/bash
# The round's measurement set for BASELINE configs[2] on one MI355X (run through gpurun):
#   bench line, rocprofv3 kernel-trace + stats of the same command, PMC passes (VALU issue, FETCH_SIZE, WRITE_SIZE), statistics build.
# Usage: bash tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>_*; copy what should be judged into profiles/
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python3 bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt -o p --output-format csv -- python3 bench.py --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_kt.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $O/${TAG}_pmc_sq -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-worst-case > $O/${TAG}_bench_pmc_sq.json 2> $O/${TAG}_pmc_sq.err
rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-worst-case > $O/${TAG}_bench_pmc_fetch.json 2> $O/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-worst-case > $O/${TAG}_bench_pmc_write.json 2> $O/${TAG}_pmc_write.err
python3 tools/render_once.py rpl_cylm 1920 1080 16 10000 build/libspath_hip_stats.so > $O/${TAG}_filter_stats.log 2>&1 || true
tail -c 2500 $O/${TAG}_bench_default.json
