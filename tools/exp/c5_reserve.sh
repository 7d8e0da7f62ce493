#!/bin/bash
# config 5 with two registration streams in pairs that leave K CUs of every XCD to the rebuild (EXPERIMENT: SLAM_MAPPER_REG_RESERVE)
OUT=gpurun_out/c5_reserve.txt
: > $OUT
run() { # label env args
  v=$(env $2 timeout -k 10 200 python3 bench.py --config 5 --stream-scans 10240 $3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.4f ms per chunk of %s  %.1f M points/s  %s' % (d['ms_per_step'], d['config'].get('chunk_scans', '?'), d['value']/1e6, json.dumps(d.get('rebuilds', d.get('config5', {}).get('rebuilds', '')))[:120]))")
  echo "$1: $v" >> $OUT
}
for rep in 1 2; do
  run "default (one stream, one scan per WG, 256)" "X=1" ""
  run "two streams pairs, no reserve, 256" "X=1" "--reg-streams 2 --pair-scans 2"
  run "two streams pairs, reserve 1/XCD, chunk 248" "SLAM_MAPPER_REG_RESERVE=1" "--reg-streams 2 --pair-scans 2 --chunk 248"
  run "two streams pairs, reserve 2/XCD, chunk 240" "SLAM_MAPPER_REG_RESERVE=2" "--reg-streams 2 --pair-scans 2 --chunk 240"
  run "two streams pairs, reserve 2/XCD, chunk 256" "SLAM_MAPPER_REG_RESERVE=2" "--reg-streams 2 --pair-scans 2"
  run "two streams pairs, reserve 4/XCD, chunk 224" "SLAM_MAPPER_REG_RESERVE=4" "--reg-streams 2 --pair-scans 2 --chunk 224"
done
cat $OUT
