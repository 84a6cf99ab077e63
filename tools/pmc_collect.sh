#!/bin/bash
# Usage (on the GPU box): bash tools/pmc_collect.sh <outdir-under-gpurun_out> <cell>
# One rocprofv3 --pmc pass per counter group (never combined with trace options).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
CELL=${2:-0.035}
export TMPDIR=/tmp
cd /tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $group --output-format csv -d $OUT/g$i -- python3 $R/tools/one_icp.py $CELL > $OUT.g$i.log 2>&1 || echo "group $i failed"
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU
GROUPS
cd $R
python3 tools/pmc_summary.py $OUT
