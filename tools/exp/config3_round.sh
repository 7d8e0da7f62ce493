# config 3 after a change to the chain: the chain's tests, the bench (three drivers), a kernel trace of the C++ sequence
mkdir -p gpurun_out/r06; R=$PWD
timeout -k 10 900 python -m pytest tests/test_gseg.py tests/test_ccicp.py tests/test_gpu_ccicp_chain.py tests/test_gpu_cpp_adapters.py tests/test_ros_shims.py tests/test_gpu_icp_spread.py -x -q -m gpu > gpurun_out/r06/t_chain.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06/t_chain.log; tail -4 gpurun_out/r06/t_chain.log
grep -q "rc=0" gpurun_out/r06/t_chain.log || exit 1
python tools/bench_config3.py 50 > gpurun_out/r06/config3.json 2> gpurun_out/r06/config3.err || exit 1
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/r06/c3trace && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06/c3trace -o c3 -- python3 $R/tools/bench_config3.py 20 > /dev/null 2>&1; ls $R/gpurun_out/r06/c3trace | head
# the C++ adapter's default path (setSceneCloud + doICPMatch per cloud) under the kernel trace: launches per match
cd $R && rm -rf /tmp/c3cpp && SLAM_C3_KEEP_DIR=/tmp/c3cpp python tools/bench_config3.py 20 > /dev/null 2>&1
cd /tmp && rm -rf $R/gpurun_out/r06/c3cpp_trace && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/c3cpp_trace -o c3cpp -- /tmp/c3cpp/ccicp_sequence /tmp/c3cpp 20 10 2 seq > $R/gpurun_out/r06/c3cpp_profiled.json 2> $R/gpurun_out/r06/c3cpp.err; ls $R/gpurun_out/r06/c3cpp_trace | head -5
