# SQ counters of the sparse-conv kernels on the real layer shapes (GPU box); $1 = variant (-1 = default)
R=$GRAFT_REPO_ROOT; V=${1:--1}; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INST_LEVEL_VMEM --output-format csv -d $R/gpurun_out/pmc6 -- python3 $R/tools/sconv_sweep.py $V > $R/gpurun_out/pmc6.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmc7 -- python3 $R/tools/sconv_sweep.py $V > $R/gpurun_out/pmc7.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmc8 -- python3 $R/tools/sconv_sweep.py $V > $R/gpurun_out/pmc8.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $R/gpurun_out/pmc9 -- python3 $R/tools/sconv_sweep.py $V > $R/gpurun_out/pmc9.log 2>&1
tail -2 $R/gpurun_out/pmc6.log $R/gpurun_out/pmc7.log $R/gpurun_out/pmc8.log $R/gpurun_out/pmc9.log | cut -c1-160
