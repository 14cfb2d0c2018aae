cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O; : > $O/driver_style_ab.txt
for i in 1 2 3 4 5 6; do
  for p in 0 3; do
    QPSK_LEAN_PAIR=$p python3 bench.py --steps 20 --warmup 5 --no-timing-modes --no-config5 --no-streams --no-shard --cpu-frames 0 --no-sustained --no-gather --no-parity 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('pair $p  ms_per_step %.4f  event span %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))" >> $O/driver_style_ab.txt
  done
done
cat $O/driver_style_ab.txt
