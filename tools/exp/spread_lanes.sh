mkdir -p gpurun_out/r06
for v in 128 256 512; do SLAM_TILE_LANES=$v SLAM_AMD_MEASURE=1 python tools/spread_time.py 20 > gpurun_out/r06/spread_lanes$v.json 2> gpurun_out/r06/spread_lanes$v.err || exit 1; done
