#!/bin/bash
# level-0 counter sweep (five PMC passes, kernel-trace only): instruction mix, pipe activity, lane utilisation, LDS conflicts, waits
set -e
cd $GRAFT_REPO_ROOT
bash tools/pmc.sh l0_p1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS" --steps 3 --warmup 1 > gpurun_out/l0_p1.txt
bash tools/pmc.sh l0_p2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES" --steps 3 --warmup 1 > gpurun_out/l0_p2.txt
bash tools/pmc.sh l0_p3 "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES" --steps 3 --warmup 1 > gpurun_out/l0_p3.txt
bash tools/pmc.sh l0_p4 "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL" --steps 3 --warmup 1 > gpurun_out/l0_p4.txt
bash tools/pmc.sh l0_p5 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" --steps 3 --warmup 1 > gpurun_out/l0_p5.txt
echo done
