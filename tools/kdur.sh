#!/bin/bash
# kernel durations (rocprofv3 kernel trace of one alignment) for the default build and the named variants
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for v in default "$@"; do
  if [ "$v" = default ]; then unset MAPALIGN_LIB; else export MAPALIGN_LIB=$R/3d-localisation-and-mapping_amd/lib/libmapalign_$v.so; fi
  rm -rf $R/gpurun_out/kdur/$v
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kdur/$v -- python3 $R/tools/one_icp.py 0.0 > $R/gpurun_out/kdur.$v.log 2>&1
  echo "== $v"; python3 $R/tools/prof_summary.py $R/gpurun_out/kdur/$v | grep -E "k_reduce_solve|k_nn_accum" | cut -c1-120
done
