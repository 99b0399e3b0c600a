#!/bin/bash
# SQ counter passes for the frame kernels (run on the GPU box through gpurun):
# issue utilisation, stall split and LDS bank conflicts of one launch per kernel.
R=$GRAFT_REPO_ROOT
O=${KWY_MEASURE_OUT:-$R/gpurun_out/pmc_sq}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
set -e
# one wave of 16 pairs on ONE stream, kernel by kernel: the analysis launches take 16 utterances (33 616 frames) each
CMD="python $R/bench.py --driver serial --batch 16 --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $O/pass1 -- $CMD > $O/pass1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA \
  --kernel-trace --output-format csv -d $O/pass2 -- $CMD > $O/pass2.log 2>&1
python $R/tools/pmc_sq_summary.py $O > $O/summary.txt
echo done > $O/DONE
