#!/bin/bash
# Rehearsals and soaks that get committed under profiles/ at the end of a round (GPU box):
#   bash tools/round_records.sh r05 A     rehearsals of the N > 1 bench on one GPU, peer-to-peer exchange, shard / map timings
#   bash tools/round_records.sh r05 B     soaks, big pairs, loop fuzz
# Output: gpurun_out/records_<tag>/ .  Every step is bounded by its own timeout; a failed step is named, the rest still run
# only if it did not time out (a timed-out GPU step ends the script).
set -u
TAG=${1:-r05}
PART=${2:-A}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/records_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
step() {   # step <seconds> <outfile> <command...>
  local secs=$1 out=$2; shift 2
  echo "[$(date +%H:%M:%S)] $*"
  timeout -k 10 $secs "$@" > $out 2> $out.err
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMED OUT: $* — stopping here"; exit 1; fi
  [ $rc -ne 0 ] && echo "rc=$rc: $*  (see $out.err)"
  return 0
}
if [ "$PART" = A ]; then
  step 200 $OUT/${TAG}_bench_rehearse_sharded_one_rank.json python3 bench.py --rehearse-sharded --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs
  step 300 $OUT/${TAG}_bench_n2_rehearsal_one_gpu.json python3 bench.py --gpus 2 --backend gloo --force-device 0 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs
  : > $OUT/${TAG}_p2p_rehearsals_one_gpu.jsonl
  for args in "--mode procs --ranks 2" "--mode procs --ranks 4" "--mode threads --ranks 2" "--mode threads --ranks 4" "--mode timeout" \
              "--mode procs --ranks 2 --points 100000 --iters 60 --tol 1e-9 --reps 400" \
              "--mode mixed --ranks 8 --procs 4 --reps 2" "--mode mixed --ranks 16 --procs 4 --reps 2"; do
    step 300 $OUT/p2p.tmp python3 tools/p2p_rehearsal.py $args
    tail -1 $OUT/p2p.tmp >> $OUT/${TAG}_p2p_rehearsals_one_gpu.jsonl
  done
  rm -f $OUT/p2p.tmp $OUT/p2p.tmp.err
  step 200 $OUT/${TAG}_shard_iteration_times.txt python3 tools/shard_iter_times.py
  step 200 $OUT/${TAG}_map_time.txt python3 tools/map_time.py 24
  rm -rf $R/gpurun_out/prof_map
  ( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_map -- python3 $R/tools/map_time.py 24 > $R/gpurun_out/prof_map.log 2>&1 ) || echo "map_time under rocprof failed"
  python3 tools/prof_summary.py $R/gpurun_out/prof_map | grep -E "k_unproject|k_sort|k_scan|k_voxel|per-kernel" | cut -c1-200 > $OUT/${TAG}_map_builder_kernel_trace.txt
else
  step 400 $OUT/soak.txt python3 tools/soak.py
  step 200 $OUT/soak_many.txt python3 tools/soak_many.py
  step 200 $OUT/soak_staging.txt python3 tools/soak_staging.py
  step 300 $OUT/soak_sharded.txt python3 tools/soak_sharded.py
  cat $OUT/soak.txt $OUT/soak_many.txt $OUT/soak_staging.txt $OUT/soak_sharded.txt > $OUT/${TAG}_soak.txt
  step 500 $OUT/big.txt python3 tools/big_pair.py 2 4 8 16 32
  step 400 $OUT/fuzz.txt python3 tools/fuzz_icp.py
  cat $OUT/big.txt $OUT/fuzz.txt > $OUT/${TAG}_big_pairs_and_loop_fuzz.txt
fi
find $OUT -name "*.err" -size 0 -delete
ls -la $OUT
