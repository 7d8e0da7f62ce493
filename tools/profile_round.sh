#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the default bench under rocprofv3
# (kernel trace + stats), then the counter passes the roofline needs.  Counter
# passes use --pmc with --kernel-trace only (no other trace domains).
#   usage: bash tools/profile_round.sh r01 [a|b]     (two gpurun calls of at most 1200 s: a = the bench's trace and counter
#   passes, b = the side runs; without the second argument both, where the time allows)
# Output under gpurun_out/profile_<tag>/ ; tools/summarize_profiles.py turns it
# into the committed files under profiles/.
TAG=${1:-r01}
PART=${2:-ab}
OUT=gpurun_out/profile_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [[ $PART == *a* ]]; then rm -rf "$OUT"; fi
mkdir -p "$OUT"
BENCH="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
if [[ $PART == *a* ]]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.json 2> $OUT/stats.err || echo "stats pass failed"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM" \
           "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$i -- $BENCH > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err || echo "pmc pass $i failed"
done
# the point-to-line solver (north_star's; SLAM_ICP_P2L): the same passes of `bench.py --mode p2l`
BENCH_P2L="$BENCH --mode p2l"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_p2l -- $BENCH_P2L > $OUT/stats_p2l.json 2> $OUT/stats_p2l.err || echo "p2l stats pass failed"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM" \
           "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmcp2l_$i -- $BENCH_P2L > $OUT/pmcp2l_$i.json 2> $OUT/pmcp2l_$i.err || echo "p2l pmc pass $i failed"
done
fi
if [[ $PART == *b* ]]; then
# the reference's own grid update (MLS::addToOccupancy's endpoint loops): kernel trace + HBM traffic of bench.py's endpoint leg alone
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_endpoints -- python3 tools/profile_endpoints.py > $OUT/endpoints.json 2> $OUT/endpoints.err || echo "endpoint stats failed"
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmcep_$i -- python3 tools/profile_endpoints.py > /dev/null 2>> $OUT/endpoints.err || echo "endpoint pmc pass failed"
done
# the run-length merged raycast (measured slower: the counters say why) and the single-scan / config-3 path
for PMC in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_merge -- $BENCH --raycast merge > $OUT/pmc_merge.json 2> $OUT/pmc_merge.err || echo "pmc merge pass failed"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_merge -- $BENCH --raycast merge > $OUT/stats_merge.json 2> $OUT/stats_merge.err || echo "stats merge failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- python3 tools/bench_config3.py 20 > $OUT/config3_profiled.json 2> $OUT/config3.err || echo "config 3 stats failed"
# round 6: the C++ adapter's DEFAULT path (setSceneCloud + doICPMatch per cloud, tests/cpp/ccicp_sequence.cpp) under the kernel trace --
# the binary itself, so that the trace holds a match's launches and nothing else (launches per match: tools/summarize_profiles.py)
rm -rf /tmp/c3cpp && SLAM_C3_KEEP_DIR=/tmp/c3cpp python3 tools/bench_config3.py 20 > /dev/null 2>> $OUT/config3.err
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/stats_c3cpp" -- /tmp/c3cpp/ccicp_sequence /tmp/c3cpp 20 10 2 seq > "$GRAFT_REPO_ROOT/$OUT/config3_cpp_profiled.json" 2>> "$GRAFT_REPO_ROOT/$OUT/config3.err") || echo "config 3 C++ trace failed"
# round 6: the spread form one scan at a time (kernel time by HIP events, every pose against the oracle), and where an iteration's time goes
timeout -k 10 300 python3 tools/spread_time.py 20 > $OUT/spread_time.json 2> $OUT/spread.err || echo "spread_time failed"
SLAM_AMD_MEASURE=1 SLAM_SPREAD_STAMPS=1 timeout -k 10 300 python3 tools/spread_time.py 5 > $OUT/spread_stamps.json 2>> $OUT/spread.err || echo "spread stamps failed (measurement build missing?)"
[ -x tools/exp/latency_chase ] && ./tools/exp/latency_chase > $OUT/latency_chase.txt 2>&1
timeout -k 10 600 python3 bench.py --config 3 > $OUT/config3.json 2>> $OUT/config3.err || echo "config 3 failed"
# the N > 1 path with two ranks on this one GPU (the library's merge over its host-staged communicator, gloo carrying the buffers)
timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --one-device --steps 10 --warmup 3 --no-cpu-baseline > $OUT/two_ranks_one_gpu.json 2> $OUT/two_ranks.err || echo "two-rank rehearsal failed"
# ... and with ONE rank over a real RCCL communicator (the N > 1 pipeline: merge_begin / merge_finish in every step), with and without the merge
timeout -k 10 600 python3 bench.py --force-dist --steps 50 --warmup 5 --no-extras --no-cpu-baseline > $OUT/force_dist.json 2> $OUT/force_dist.err || echo "force-dist failed"
timeout -k 10 600 python3 bench.py --force-dist --no-merge --steps 50 --warmup 5 --no-extras --no-cpu-baseline > $OUT/force_dist_no_merge.json 2>> $OUT/force_dist.err || echo "force-dist --no-merge failed"
timeout -k 10 600 python3 bench.py --force-dist --reg-cu-cap 1 --steps 50 --warmup 5 --no-extras --no-cpu-baseline > $OUT/force_dist_cu_cap1.json 2>> $OUT/force_dist.err || echo "force-dist --reg-cu-cap failed"
timeout -k 10 600 python3 bench.py --mode p2l --steps 50 --warmup 5 --no-extras > $OUT/p2l.json 2> $OUT/p2l.err || echo "p2l bench failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_build -- python3 tools/profile_build.py > $OUT/build.txt 2> $OUT/build.err || echo "build stats failed"
timeout -k 10 600 python3 bench.py --config 5 > $OUT/config5.json 2> $OUT/config5.err || echo "config 5 failed"
timeout -k 10 600 python3 bench.py --config 4 --no-extras --no-cpu-baseline > $OUT/config4.json 2> $OUT/config4.err || echo "config 4 failed"
# the un-profiled bench line, for reference beside the profiled one
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
fi
ls $OUT
