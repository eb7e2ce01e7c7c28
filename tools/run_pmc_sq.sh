set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sq
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/sq/a -- python3 tools/pmc_conv.py all > gpurun_out/sq/a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d gpurun_out/sq/b -- python3 tools/pmc_conv.py all > gpurun_out/sq/b.log 2>&1
