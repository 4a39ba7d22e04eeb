set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_exp18
mkdir -p $O
MS_LDE_VIRTUAL=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/material -- python3 tools/wide_bench.py --proofs 1 > $O/material.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/virtual -- python3 tools/wide_bench.py --proofs 1 > $O/virtual.log 2>&1
python3 tools/pmc_summary.py $O/material $O/virtual | grep -A8 "LeafHashKernel<GL, 1, true"
