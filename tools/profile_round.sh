#!/bin/bash
# One-stop profile collection for a round (GPU box):  bash tools/profile_round.sh r02
# Writes the summaries that get committed under profiles/ into gpurun_out/profiles_<tag>/ :
#   <tag>_kernel_trace_stats.txt       rocprofv3 --kernel-trace --stats of a short bench.py run
#   <tag>_bench_under_rocprof.json     the JSON line that profiled run printed
#   <tag>_per_iteration_kernel_us.txt  k_nn_accum / k_reduce_solve durations of one alignment, launch order
#   <tag>_c3_kernel_trace.txt          kernels of the depth path (k_unproject) from tools/c3_time.py
#   pmc_traffic.json                   HBM-side bytes of k_nn_accum per launch (separate --pmc passes), stamped
#                                      with the kernel-source hash bench.py checks
#   <tag>_pmc_k_nn_accum.txt           SQ / TA / TCP / TCC counters per launch (separate --pmc passes)
#   <tag>_set_pair_timeline_now.txt    kernels and copies of one warm ma_set_pair in time order (+ host time stamps of both
#                                      threads if lib/libmapalign_pt.so, the -DMA_PAIR_TIMES=1 build, is there)
# PMC passes are never combined with trace options; every rocprofv3 runs `python3 <script>` directly.
set -u
TAG=${1:-r02}
STEPS=${2:-123456}     # which of the six steps to run, e.g. 123 or 45
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp

if [[ $STEPS == *1* ]]; then
echo "[1/5] kernel trace + stats of bench.py"
rm -rf $R/gpurun_out/prof_bench
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- \
  python3 $R/bench.py --steps 5 --warmup 1 --headline-only > $OUT/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof_bench.err \
  || echo "bench under rocprof failed"
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_bench $OUT/${TAG}_kernel_trace_stats.txt > /dev/null
fi

if [[ $STEPS == *2* ]]; then
echo "[2/5] per-iteration kernel durations"
rm -rf $R/gpurun_out/prof_iter
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_iter -- python3 $R/tools/one_icp.py 0.0 > $R/gpurun_out/prof_iter.log 2>&1 \
  || echo "one_icp under rocprof failed"
python3 $R/tools/iter_times.py $R/gpurun_out/prof_iter > $OUT/${TAG}_per_iteration_kernel_us.txt
fi

if [[ $STEPS == *3* ]]; then
echo "[3/5] depth path"
rm -rf $R/gpurun_out/prof_c3
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/tools/c3_time.py > $R/gpurun_out/prof_c3.log 2>&1 \
  || echo "c3 under rocprof failed"
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_c3 | grep -E "k_unproject|k_nn_accum|k_reduce_solve|per-kernel" | cut -c1-200 > $OUT/${TAG}_c3_kernel_trace.txt
fi

if [[ $STEPS == *4* ]]; then
echo "[4/5] HBM-side traffic of k_nn_accum"
bash $R/tools/pmc_traffic.sh 0.0 > $R/gpurun_out/pmc_traffic.log 2>&1
cp $R/gpurun_out/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null || echo "no traffic json"
fi

if [[ $STEPS == *5* ]]; then
echo "[5/5] SQ / TA / TCP / TCC counters"
bash $R/tools/pmc_collect.sh pmc_$TAG 0.0 > $R/gpurun_out/pmc_collect.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_$TAG $OUT/${TAG}_pmc_k_nn_accum.txt > /dev/null 2>&1 || echo "pmc summary failed"
fi

if [[ $STEPS == *6* ]]; then
echo "[6] one warm ma_set_pair in time order"
rm -rf $R/gpurun_out/prof_pair
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/prof_pair -- python3 $R/tools/pair_prof.py > $R/gpurun_out/prof_pair.log 2>&1 \
  || echo "pair_prof under rocprof failed"
python3 $R/tools/pair_timeline.py $R/gpurun_out/prof_pair > $OUT/${TAG}_set_pair_timeline_now.txt
PT=$R/3d-localisation-and-mapping_amd/lib/libmapalign_pt.so
if [ -f $PT ]; then
  MAPALIGN_LIB=$PT timeout -k 10 120 python3 $R/tools/pair_host_times.py 2> $R/gpurun_out/pair_host_times.txt
  { echo; echo "## host time stamps of both threads, no profiler (tools/pair_host_times.py, -DMA_PAIR_TIMES=1 build)"; tail -14 $R/gpurun_out/pair_host_times.txt; } >> $OUT/${TAG}_set_pair_timeline_now.txt
fi
fi

ls -la $OUT
