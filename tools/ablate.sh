# measurement helper: ablations of rx_fused_pipe_kernel on the bench workload (run on the GPU box)
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py --steps 10 --warmup 2 --cpu-frames 0 --no-parity $EXTRA 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('ms/step %.4f  kernel_ms %.4f frac %.3f' % (d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac']))"; }
run A=0
run QPSK_PIPE_DBG=3
EXTRA="--frame-size 16448" run A=0
EXTRA="--frame-size 16448" run QPSK_PIPE_DBG=3
EXTRA="--frame-size 16448" run QPSK_PIPE_DBG=1
EXTRA="--frame-size 16448" run QPSK_PIPE_DBG=2
EXTRA="--frames 8192" run A=0
EXTRA="--frames 8192" run QPSK_PIPE_NF=2
