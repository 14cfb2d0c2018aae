cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r06_gputests.log 2>&1; tail -6 gpurun_out/r06_gputests.log
timeout -k 10 600 python3 bench.py --no-timing-modes --no-config5 --no-streams --cpu-frames 0 > $O/bench_b.json 2> $O/bench_b.err || tail -20 $O/bench_b.err
python3 -c "
import json;d=json.load(open('$O/bench_b.json'))
print(d['ms_per_step'], d.get('sustained'))
print(d['shard_8192']['ms_per_step'], d['shard_8192'].get('sustained'))
print(d.get('gather'))"
