#!/bin/bash
# HBM-side traffic of k_nn_accum per launch (bench.py's roofline.traffic).  GPU box only:
#   bash tools/pmc_traffic.sh <cell>
# Separate rocprofv3 --pmc passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2), never mixed
# with trace options; then tools/pmc_traffic.py turns the CSVs into profiles/pmc_traffic.json.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_traffic
CELL=${1:-0.0}
export TMPDIR=/tmp
cd /tmp
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $group --output-format csv -d $OUT/g$i -- python3 $R/tools/one_icp.py $CELL > $OUT.g$i.log 2>&1 || echo "group $i failed"
done
cd $R
python3 tools/pmc_traffic.py $OUT gpurun_out/pmc_traffic.json
