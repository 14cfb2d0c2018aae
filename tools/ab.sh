# measurement helper: same-box A/B on the bench workload (run on the GPU box)
#   bash tools/ab.sh "ENV1=.. ENV2=.." "ENVa=.." ...     one bench run per argument (its words are the environment), twice
#   EXTRA="--frames 8192" bash tools/ab.sh ...
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-parity $EXTRA 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('ms/step %.4f  kernel_ms %.4f frac %.3f' % (d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac']))"; }
for i in 1 2; do
  for cfg in "$@"; do run $cfg; done
done
