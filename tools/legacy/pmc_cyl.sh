set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in rpl_cyl2s rpl_cyl4s rpl_filter2s; do python tools/render_once.py $v 1920 1080 16 10000 build/libspath_hip_stats.so; done 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_cyl_stats.log
for v in rpl_cyl2s rpl_cyl4s; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS -d gpurun_out/pmcA_$v -o p --output-format csv -- python3 tools/render_once.py $v > gpurun_out/pmcA_$v.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d gpurun_out/pmcB_$v -o p --output-format csv -- python3 tools/render_once.py $v > gpurun_out/pmcB_$v.log 2>&1
python tools/pmc_summary.py gpurun_out/pmcA_$v k_pt_filter > gpurun_out/r02_pmc_$v.txt; python tools/pmc_summary.py gpurun_out/pmcB_$v k_pt_filter >> gpurun_out/r02_pmc_$v.txt
done
cat gpurun_out/r02_cyl_stats.log gpurun_out/r02_pmc_*.txt
