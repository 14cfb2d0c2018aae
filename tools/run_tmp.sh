cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r06_gputests.log 2>&1; tail -6 gpurun_out/r06_gputests.log
