cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_t5.log 2>&1; tail -4 gpurun_out/r2_t5.log
timeout -k 10 300 python tools/sweep.py --frames 8192 "" "QPSK_PIPE_V=1" 2>&1 | grep -v amdgpu > gpurun_out/r2_sweep9.log; cat gpurun_out/r2_sweep9.log
timeout -k 10 300 python tools/sweep.py --frames 4096 "" "QPSK_PIPE_V=1" "QPSK_PIPE_LAYOUT_LO=0x11111 QPSK_PIPE_LAYOUT_HI=0x000111" "QPSK_PIPE_LAYOUT_LO=0x01122 QPSK_PIPE_LAYOUT_HI=0x000011" 2>&1 | grep -v amdgpu > gpurun_out/r2_sweep9b.log; cat gpurun_out/r2_sweep9b.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err; cat gpurun_out/r2_bench1.json | cut -c1-400
timeout -k 10 300 python bench.py --frames 8192 --cpu-frames 0 > gpurun_out/r2_bench8192.json 2> gpurun_out/r2_bench8192.err; cat gpurun_out/r2_bench8192.json | cut -c1-400
timeout -k 10 300 python tools/bench_config5.py > gpurun_out/r2_config5.log 2>&1; tail -3 gpurun_out/r2_config5.log
QPSK_PIPE_V=1 timeout -k 10 300 python tools/bench_config5.py > gpurun_out/r2_config5_v1.log 2>&1; tail -3 gpurun_out/r2_config5_v1.log
