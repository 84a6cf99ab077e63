#!/bin/bash
# A/B of dynamic instruction counts: bash tools/pmc_ab.sh <variant-name> [cell]   (GPU box only)
# One rocprofv3 --pmc pass for the default build and one for lib/libmapalign_<variant>.so.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1
CELL=${2:-0.0}
export TMPDIR=/tmp
cd /tmp
G="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
timeout -k 10 240 rocprofv3 --pmc $G --output-format csv -d $R/gpurun_out/pmcab/default/g1 -- python3 $R/tools/one_icp.py $CELL > $R/gpurun_out/pmcab.default.log 2>&1
export MAPALIGN_LIB=$R/3d-localisation-and-mapping_amd/lib/libmapalign_$V.so
timeout -k 10 240 rocprofv3 --pmc $G --output-format csv -d $R/gpurun_out/pmcab/$V/g1 -- python3 $R/tools/one_icp.py $CELL > $R/gpurun_out/pmcab.$V.log 2>&1
cd $R
echo "== default"; python3 tools/pmc_summary.py gpurun_out/pmcab/default | grep -A9 "k_nn_accum"
echo "== $V"; python3 tools/pmc_summary.py gpurun_out/pmcab/$V | grep -A9 "k_nn_accum"
