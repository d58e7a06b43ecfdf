#!/bin/bash
# rocprofv3 counter passes (one run per pass, counters only with --kernel-trace), outputs under gpurun_out/pmc/
# p5/p6: typed VALU instruction counters of gfx950 (rocprofv3 -L): executed FP64 / FP32 work of the kernels
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${PMC_OUT:-gpurun_out/pmc}
PROG=${PMC_PROG:-tools/prof_run.py}
mkdir -p $OUT
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name --pmc "$@" -- python3 $PROG > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; exit 1; }; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU && \
run p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA && \
run p3 FETCH_SIZE GRBM_GUI_ACTIVE && \
run p4 WRITE_SIZE && \
run p5 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 && \
run p6 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE && echo all passes ok
