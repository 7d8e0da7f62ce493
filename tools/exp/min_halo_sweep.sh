#!/bin/bash
# slam_icp_params::list_min_halo: config 2 against a 5 k / 10 k / 20 k-point model, config 5 with one and two registration streams (round 5)
cd $GRAFT_REPO_ROOT
O=gpurun_out/min_halo
mkdir -p $O
line() { python - "$1" "$2" <<P
import json,sys
n,f=sys.argv[1],sys.argv[2]
try:
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    ti=d["config"].get("target_index") or d["config"].get("icp_index") or {}
    print("%-34s %.4f ms  %.1f M pts/s  err %.4f  pitch %s halo %s %s" % (n, d["ms_per_step"], d["value"]/1e6, d["max_pose_error_m"], ti.get("list_pitch"), ti.get("list_halo"), json.dumps(d["config"].get("mapper",""))))
except Exception as e:
    print(n, "failed", e)
P
}
for H in -1 0.08 0.1 0.125 0.15 0.2 0.3; do
  for M in 5000 10000; do
    timeout -k 10 200 python bench.py --map-points $M --list-min-halo $H --no-extras --no-cpu-baseline --steps 40 --warmup 5 > $O/c2_${M}_$H.json 2> $O/c2_${M}_$H.err
    line "config2 M=$M halo>=$H" $O/c2_${M}_$H.json
  done
  timeout -k 10 200 python bench.py --config 5 --stream-scans 10240 --list-min-halo $H > $O/c5_$H.json 2> $O/c5_$H.err
  line "config5 1 lane halo>=$H" $O/c5_$H.json
  
done
