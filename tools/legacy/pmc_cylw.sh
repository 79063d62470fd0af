set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in rpl_cyl4s rpl_cylw4s; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS -d gpurun_out/pmcC_$v -o p --output-format csv -- python3 tools/render_once.py $v > gpurun_out/pmcC_$v.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d gpurun_out/pmcD_$v -o p --output-format csv -- python3 tools/render_once.py $v > gpurun_out/pmcD_$v.log 2>&1
echo "== $v"; python tools/pmc_summary.py gpurun_out/pmcC_$v k_pt_filter; python tools/pmc_summary.py gpurun_out/pmcD_$v k_pt_filter
done
