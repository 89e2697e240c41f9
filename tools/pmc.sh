#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>   -- one rocprofv3 --pmc pass per counter group, CSVs under gpurun_out/pmc_<tag>/
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
groups=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE"
 "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_IFETCH"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_PENDING_STALL_CYCLES_sum"
)
i=0
for g in "${groups[@]}"; do
  timeout -s KILL 120 rocprofv3 --pmc $g --output-format csv -d $R/gpurun_out/pmc_$tag/g$i -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-traffic "$@" > $R/gpurun_out/pmc_$tag/g$i.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_$tag
