#!/bin/bash
# The round's measurement set on one MI355X (run through gpurun): BASELINE configs[2] -- bench line, rocprofv3 kernel-trace + stats of
# the same command, PMC passes (VALU issue, waits / matrix pipe / LDS, FETCH_SIZE, WRITE_SIZE), statistics build -- and the declared
# slices of configs[3] / configs[4] (tools/slice_once.py) under kernel trace and the same PMC passes + L2 hit/miss.
# Usage: bash tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>_*; tools/publish_profiles.sh <tag> copies what is judged into profiles/
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
B="python3 bench.py"
$B > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
# (--no-worst-case: the large-triangle scene is the same kernel; its one short launch would sit in the kernel's average)
rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt -o p --output-format csv -- $B --no-cpu-baseline --no-worst-case --no-valu-microbench > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_kt.err
P="--steps 1 --warmup 0 --no-cpu-baseline --no-worst-case --no-valu-microbench"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $O/${TAG}_pmc_sq -o p --output-format csv -- $B $P > $O/${TAG}_bench_pmc_sq.json 2> $O/${TAG}_pmc_sq.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/${TAG}_pmc_wait -o p --output-format csv -- $B $P > $O/${TAG}_bench_pmc_wait.json 2> $O/${TAG}_pmc_wait.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE -d $O/${TAG}_pmc_cls -o p --output-format csv -- $B $P > $O/${TAG}_bench_pmc_cls.json 2> $O/${TAG}_pmc_cls.err
rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -o p --output-format csv -- $B $P > $O/${TAG}_bench_pmc_fetch.json 2> $O/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -o p --output-format csv -- $B $P > $O/${TAG}_bench_pmc_write.json 2> $O/${TAG}_pmc_write.err
python3 tools/render_once.py rpl_cylm 1920 1080 16 10000 build/libspath_hip_stats.so > $O/${TAG}_filter_stats.log 2>&1 || true
for S in config3 config4; do
  python3 tools/slice_once.py $S 64 > $O/${TAG}_${S}_plain.json 2> $O/${TAG}_${S}_plain.err
  rocprofv3 --kernel-trace --stats -d $O/${TAG}_${S}_kt -o p --output-format csv -- python3 tools/slice_once.py $S 64 > $O/${TAG}_${S}_kt.json 2> $O/${TAG}_${S}_kt.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $O/${TAG}_${S}_pmc_sq -o p --output-format csv -- python3 tools/slice_once.py $S 64 > $O/${TAG}_${S}_pmc_sq.json 2> $O/${TAG}_${S}_pmc_sq.err
  rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_${S}_pmc_fetch -o p --output-format csv -- python3 tools/slice_once.py $S 64 > $O/${TAG}_${S}_pmc_fetch.json 2> $O/${TAG}_${S}_pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_${S}_pmc_write -o p --output-format csv -- python3 tools/slice_once.py $S 64 > $O/${TAG}_${S}_pmc_write.json 2> $O/${TAG}_${S}_pmc_write.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/${TAG}_${S}_pmc_tcc -o p --output-format csv -- python3 tools/slice_once.py $S 64 > $O/${TAG}_${S}_pmc_tcc.json 2> $O/${TAG}_${S}_pmc_tcc.err || true
done
tail -c 2500 $O/${TAG}_bench_default.json
