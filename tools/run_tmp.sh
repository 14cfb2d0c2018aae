cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r06_gputests.log 2>&1; tail -5 gpurun_out/r06_gputests.log
timeout -k 10 900 python3 bench.py > $O/bench_c.json 2> $O/bench_c.err || tail -20 $O/bench_c.err
python3 -c "
import json;d=json.load(open('$O/bench_c.json'))
print('c2', d['ms_per_step'], d['roofline']['frac'], d.get('sustained_ms_per_step'))
print('c3', d['config3']['ms_per_step'], d['config3']['kernels'])
h=d['hist']; print('hist', h['ms_per_step'], h['kernels'], h.get('two_launch_route'), h['symbol_mismatches'], h['index_mismatches'])
print('shard', d['shard_8192']['ms_per_step'], d['shard_8192'].get('sustained_ms_per_step'))
print('gather', {k:v for k,v in d['gather'].items() if 'ms' in k})
print('parity', d['parity'])
"
