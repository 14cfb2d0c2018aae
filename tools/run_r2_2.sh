set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/fir_wave_profile.py "pipe_v=2" "pipe_v=2 pipe_nf=8" "pipe_v=1" > gpurun_out/r2_prof1.log 2>&1
cat gpurun_out/r2_prof1.log
