# measurement helper: same-box A/B of library builds on the bench workload (run on the GPU box)
#   bash tools/ab.sh "<env for all runs>" lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
COMMON=$1; shift
run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-parity $EXTRA 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('ms/step %.4f  kernel_ms %.4f frac %.3f' % (d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac']))"; }
for i in 1 2; do
  for lib in "$@"; do run $COMMON QPSK_HIP_LIB=$GRAFT_REPO_ROOT/qpsk_amd/$lib; done
done
