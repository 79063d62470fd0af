#!/bin/bash
# Copy the set collected by tools/collect_profiles.sh <tag> (gpurun_out/<tag>_*) into profiles/ under the round's names and
# refresh the JSONs bench.py reads (valu_issue, hbm_traffic, filter_stats), every entry stamped with the source hash of the library
# the measurement ran on (the bench line's library_source_hash).  Usage: bash tools/publish_profiles.sh <tag> [round] [kernel]
set -e
TAG=${1:?tag}; RND=${2:-3}; KERNEL=${3:-rpl_cylm}
R=$(printf "r%02d" "$RND")
cd "$(dirname "$0")/.."
O=gpurun_out
cp $O/${TAG}_bench_default.json profiles/${R}_bench_default.json
cp $O/${TAG}_bench_under_rocprof.json profiles/${R}_bench_default_under_rocprof.json
cp $O/${TAG}_kt/p_kernel_stats.csv profiles/${R}_bench_default_kernel_stats.csv
cp $O/${TAG}_kt/p_kernel_trace.csv profiles/${R}_bench_default_kernel_trace.csv
cp $O/${TAG}_pmc_sq/p_counter_collection.csv profiles/${R}_bench_pmc_sq_counters.csv
cp $O/${TAG}_bench_pmc_sq.json profiles/${R}_bench_pmc_sq_run.json
cp $O/${TAG}_pmc_wait/p_counter_collection.csv profiles/${R}_bench_pmc_wait_mfma_lds_counters.csv
cp $O/${TAG}_pmc_fetch/p_counter_collection.csv profiles/${R}_bench_pmc_FETCH_SIZE.csv
cp $O/${TAG}_pmc_write/p_counter_collection.csv profiles/${R}_bench_pmc_WRITE_SIZE.csv
grep -v "amdgpu.ids" $O/${TAG}_filter_stats.log > profiles/${R}_filter_stats_build.log
HASH=$(python3 -c "import json,sys; print(json.loads([l for l in open('$O/${TAG}_bench_pmc_sq.json') if l.startswith('{')][-1])['library_source_hash'])")
python3 tools/valu_issue_from_pmc.py profiles/${R}_bench_pmc_sq_counters.csv profiles/${R}_bench_pmc_sq_run.json $RND
if [ -f $O/${TAG}_pmc_cls/p_counter_collection.csv ]; then
  cp $O/${TAG}_pmc_cls/p_counter_collection.csv profiles/${R}_bench_pmc_valu_classes.csv
  cp $O/${TAG}_bench_pmc_cls.json profiles/${R}_bench_pmc_valu_classes_run.json
  python3 tools/valu_classes_from_pmc.py profiles/${R}_bench_pmc_valu_classes.csv profiles/${R}_bench_pmc_valu_classes_run.json $RND
fi
python3 tools/hbm_traffic_from_pmc.py profiles/${R}_bench_pmc_FETCH_SIZE.csv profiles/${R}_bench_pmc_WRITE_SIZE.csv 10000tris_1920x1080x256_g1_${KERNEL} $RND $HASH
python3 tools/filter_stats_from_log.py profiles/${R}_filter_stats_build.log 10000 1920 1080 ${KERNEL} "statistics build (-DSP_FILTER_STATS) of the same kernel, configs[2] frame at 16 spp (profiles/${R}_filter_stats_build.log)"
for S in config3 config4; do
  cp $O/${TAG}_${S}_plain.json profiles/${R}_${S}_slice_run.json
  cp $O/${TAG}_${S}_kt/p_kernel_stats.csv profiles/${R}_${S}_slice_kernel_stats.csv
  cp $O/${TAG}_${S}_pmc_sq/p_counter_collection.csv profiles/${R}_${S}_pmc_sq_counters.csv
  cp $O/${TAG}_${S}_pmc_fetch/p_counter_collection.csv profiles/${R}_${S}_pmc_FETCH_SIZE.csv
  cp $O/${TAG}_${S}_pmc_write/p_counter_collection.csv profiles/${R}_${S}_pmc_WRITE_SIZE.csv
  [ -f $O/${TAG}_${S}_pmc_tcc/p_counter_collection.csv ] && cp $O/${TAG}_${S}_pmc_tcc/p_counter_collection.csv profiles/${R}_${S}_pmc_TCC_hit_miss.csv || true
  python3 tools/slice_profiles.py $O/${TAG}_${S}_pmc_sq.json profiles/${R}_${S}_pmc_sq_counters.csv profiles/${R}_${S}_pmc_FETCH_SIZE.csv profiles/${R}_${S}_pmc_WRITE_SIZE.csv profiles/${R}_${S}_pmc_TCC_hit_miss.csv $RND
done
