#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 120 python tools/config5_trace.py background=${1:-1} > gpurun_out/c5t_$i.txt 2>&1
  head -1 gpurun_out/c5t_$i.txt
done
