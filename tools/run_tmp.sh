cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 500 python3 tools/power_probe.py 4096:0:lean_pair=0 4096:0:lean_pair=1 8192:0:lean_pair=0 8192:0:lean_pair=2 6144:0:lean_pair=0 6144:0:lean_pair=2 4096:0:lean_pair=0 4096:0:lean_pair=1 8192:0:lean_pair=0 8192:0:lean_pair=2 > $O/power_pair.txt 2>&1 || echo FAILED
grep -E "child|amd-smi" $O/power_pair.txt | cut -c1-330
