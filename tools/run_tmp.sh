cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r06_gputests.log 2>&1; tail -4 gpurun_out/r06_gputests.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/measure_r06.sh a
