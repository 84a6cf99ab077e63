#!/bin/bash
# k_unproject duration (rocprofv3 kernel trace) of the default build and every lib/libmapalign_<name>.so variant,
# configs[2] frame (1280 x 960) via tools/c3_time.py.  GPU box only.
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for lib in "" $(ls $R/3d-localisation-and-mapping_amd/lib/ | grep libmapalign_ | sed 's/libmapalign_//; s/.so//'); do
  rm -rf $R/gpurun_out/prof_unproj
  if [ -n "$lib" ]; then export MAPALIGN_LIB=$R/3d-localisation-and-mapping_amd/lib/libmapalign_$lib.so; else unset MAPALIGN_LIB; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_unproj -- python3 $R/tools/c3_time.py > /dev/null 2>&1
  echo "${lib:-default}: $(python3 $R/tools/prof_summary.py $R/gpurun_out/prof_unproj | grep k_unproject | cut -c1-140)"
done
