# SQ counters of the data-gradient kernels with the 128x128 four-wave plane tile forced (BDVCIL_PL_TILE=5): the conflict-free LDS image
# on the sites the round-1 conv_dgrad_x3 kernels run by default -> gpurun_out/r03/sq_counters_tile5.tsv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sq5; mkdir -p gpurun_out/sq5 gpurun_out/r03
export BDVCIL_PL_TILE=5
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/sq5/a -- python3 tools/pmc_conv.py dgrad > gpurun_out/sq5/a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d gpurun_out/sq5/b -- python3 tools/pmc_conv.py dgrad > gpurun_out/sq5/b.log 2>&1
python tools/pmc_summarize.py gpurun_out/sq5/a > gpurun_out/sq5_a.tsv
python tools/pmc_summarize.py gpurun_out/sq5/b > gpurun_out/sq5_b.tsv
python tools/pmc_derive.py gpurun_out/sq5_a.tsv gpurun_out/sq5_b.tsv > gpurun_out/r03/sq_counters_tile5.tsv
rm -rf gpurun_out/sq5
cut -f1-8 gpurun_out/r03/sq_counters_tile5.tsv
