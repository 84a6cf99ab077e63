#!/bin/bash
# per-iteration k_nn_accum durations (rocprofv3 kernel trace) for the default build and the named variants
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for v in default "$@"; do
  if [ "$v" = default ]; then unset MAPALIGN_LIB; else export MAPALIGN_LIB=$R/3d-localisation-and-mapping_amd/lib/libmapalign_$v.so; fi
  rm -rf $R/gpurun_out/iterab/$v
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/iterab/$v -- python3 $R/tools/one_icp.py 0.0 > $R/gpurun_out/iterab.$v.log 2>&1
  echo "== $v"; python3 $R/tools/iter_times.py $R/gpurun_out/iterab/$v
done
