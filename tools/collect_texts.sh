#!/bin/bash
# Text summaries of a tools/measure_all.sh session -> profiles/<tag>_*.txt (the CSV / JSON side is tools/collect_profiles.py).
#   bash tools/collect_texts.sh r03
tag=${1:-r03}; O=gpurun_out; P=profiles
cd "$(dirname "$0")/.."
for n in bench bench20 bench8192 bench_gpus2_shared; do [ -s $O/$n.json ] && [ ! -e $O/$n.failed ] && cp $O/$n.json $P/${tag}_$n.json; done
python3 tools/collect_hist.py gpurun_out sq_4096 > /tmp/sq4096.txt; python3 tools/collect_hist.py gpurun_out sq_8192 > /tmp/sq8192.txt
{
echo "SQ counters of the two receive kernels per launch (rocprofv3 --pmc, three passes over bench.py --frames N --steps 5 each, tools/measure_all.sh ->"
echo "python tools/collect_hist.py gpurun_out sq_N; counters summed over the chip, SQ_* cycle counters in quad-cycles)."
echo
echo "rx_lean_kernel, 8192 frames x 16384 samples, layout 1, 5, 5, 5 units on SIMDs 0-3 (ten FIR waves):"
cat /tmp/sq8192.txt
echo
echo "Derived: SQ_INSTS_VALU / 131,072 unit-chunks = ~738 per unit-chunk: 602 in the FIR stream (508 filter) + the prologue's share + 112 of the serial"
echo "wave (rx_pipe2_kernel, round 2: 816).  SQ_WAVE_CYCLES x 4 / 2816 waves (11 per workgroup) = ~480 k shader cycles per wave; SQ_ACTIVE_INST_VALU x 4"
echo "/ 1024 SIMDs = ~378 k cycles per SIMD = 76 % of the waves' lifetime (73 % for rx_pipe2_kernel) -- at a clock the board's 1400 W limit sets"
echo "(${tag}_power.txt), which is why the kernel got only 4-6 % faster when the layout took 14.5 % of its cycles per chunk round away.  LDS:"
echo "SQ_LDS_IDX_ACTIVE x 4 / 256 CUs = ~247 k cycles per CU = 51 %, bank conflicts 16 % of them."
echo
echo "rx_fused_pipe_kernel, 4096 frames (config 2):"
cat /tmp/sq4096.txt
} > $P/${tag}_sq_counters.txt
{
echo "Board power and shader clock while one receive-kernel shape runs back to back (tools/power_probe.py: rocm-smi sampled from a parent"
echo "process, the kernel looping in a child for 5 s).  MI355X, gpurun box.  Idle: 240-260 W."
echo
grep -v "amdgpu.ids\|^idle" $O/power_probe.log
echo
echo "Measurement build, rx_lean_kernel's streams with a part left out (WRONG results; QPSK_PIPE_DBG 1: no filter multiplies/adds, 16384: no window"
echo "reads after the first two blocks, 32768: no flush arithmetic, 49153: all three):"
echo
grep -v "amdgpu.ids\|^idle" $O/power_ablations.log
echo
[ -s $O/r3_power_c2.log ] && { echo "Config 2 (rx_fused_pipe_kernel), measurement build: QPSK_PIPE_DBG 1 = no filter arithmetic, 2 = no recurrence, 3 = neither:"; echo; cat $O/r3_power_c2.log; echo; }
echo "Reading.  Both shapes run AT the board's power limit (1400 W): 8192 frames at 1.86-1.93 GHz, config 2 at 1360-1395 W and 2.25-2.30 GHz"
echo "(the part's maximum is 2.40 GHz).  Without the filter's arithmetic the same kernel draws 1190-1210 W at 2.395 GHz and takes 0.243-0.247 ms: that is"
echo "the memory side's floor for this access pattern at a 128 KB frame pitch (${tag}_pitch_sweep.txt), 7-9 % under the full kernel."
echo "With the clock set by power, cycles saved buy little time (the 1, 5, 5, 5 layout: -14.5 % cycles per chunk round, -4 to -6 % time);"
echo "what a part costs in TIME is what it costs in energy: the flush 3.5-4 %, the filter's LDS reads 2-4 %, not parking the once-read input in the"
echo "caches (nontemporal loads) 1.5 %, the filter arithmetic the rest above the floor.  Config 2 sits where its two limits meet: the recurrence alone"
echo "(no filter arithmetic: 966 W, 2.39 GHz) takes 0.144 ms, the filter side alone (no recurrence, still at the limit: 1364 W) 0.152 ms, both 0.157."
} > $P/${tag}_power.txt
{
echo "qpsk_rx_batch_pitched on the bench signal, 8192 frames, frames (16384 + extra) samples apart, interleaved in one process (tools/pitch_sweep.py)."
echo "Product kernel:"
grep -v amdgpu $O/pitch_sweep.log
echo
echo "Measurement build, the stream without filter arithmetic, window reads and flush (QPSK_PIPE_DBG 49153) = what the memory side delivers to this access pattern:"
grep -v amdgpu $O/pitch_sweep_floor.log
[ -s $O/r3_stride.log ] && { echo; echo "Earlier session of this round, uniform-noise input (tools/stride_probe.py), same floor build:"; grep "dbg 49153" $O/r3_stride.log; }
echo
echo "Reading.  At a pitch of 128 KB every frame's sample n maps to the same HBM channels and a batch kernel streams sample n of all its frames at"
echo "about the same time: the floor is 4.36-4.56 TB/s against 4.7-5.2 TB/s at pitches that are not a multiple of 4 KB x 32.  The product kernel does"
echo "not see it (the same time at every pitch): it is not the memory side that limits it (${tag}_power.txt).  bench.py keeps the packed layout."
} > $P/${tag}_pitch_sweep.txt
[ -s $O/ubench_fetch.log ] && cp $O/ubench_fetch.log $P/${tag}_ubench_fetch.txt
{
echo "tools/lean_profile.py (measurement build: s_memtime stamps between the phases of a unit of rx_lean_kernel's FIR streams, the serial wave's time inside"
echo "and outside its ring stream, the 100 MHz clock for wall time), workgroups 0 and 77, after 200 untimed launches.  Shader cycles per 64-symbol chunk."
echo
grep -v amdgpu $O/lean_profile_8192.log
echo
grep -v amdgpu $O/lean_profile_4096.log
} > $P/${tag}_lean_profile.txt
grep -v amdgpu $O/config5.log > $P/${tag}_config5.txt; grep -v amdgpu $O/dropin.log > $P/${tag}_dropin_rx_frame.txt; grep -v amdgpu $O/fir_fast.log > $P/${tag}_fir_fast.txt
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
ps = sorted(glob.glob('gpurun_out/prof_streams3/**/*kernel_trace.csv', recursive=True), key=lambda p: __import__('os').path.getmtime(p))
if ps:
    rows = [r for r in csv.DictReader(open(ps[-1])) if 'qpsk' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    t0 = int(rows[0]['Start_Timestamp'])
    out = ["Streaming mode, 4096 streams x 16384-sample blocks, histogram timing (tools/bench_streams.py under rocprofv3 --kernel-trace): every launch of the",
           "large kernels in order, complex input first (6 blocks), then PCM input (6 blocks).  profiles/%s_streams_tx_kernel_stats.csv has the totals." % tag, ""]
    for r in rows:
        n = r['Kernel_Name'].split('(')[0].replace('qpsk::', '')
        if n in ('costas_pipe_kernel', 'rrc_fir_kernel', 'mixer_kernel', 'timing_hist8_kernel'):
            out.append("%-22s start %9.1f us   %8.1f us" % (n, (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    out += ["", "The spread of costas_pipe_kernel (VERDICT r2, weak 8: 171 us ... 1107 us over 12 calls) is two block shapes, not noise: the FIRST block after",
            "qpsk_streams_reset() runs the loop over the zeroed decimated_frame[] of the one-block delay (qpsk.c:186-197, SURVEY Q6) -- 2048 symbols that are",
            "exactly zero per stream.  A zero detector input is the case the hand-scheduled stream does not handle (sgn(0) = -1, costas_loop.c:44-47): every",
            "16-step group is handed to the C++ step, ~6 x slower (1.08 / 1.16 ms).  Every later block takes 176-202 us.  Per block of a running stream:",
            "complex input 0.63-0.73 (rrc_fir_kernel) + 0.13 (timing_hist8_kernel) + 0.18-0.20 ms; PCM input + 0.33-0.35 (mixer_kernel)."]
    open('profiles/%s_streams_per_call.txt' % tag, 'w').write("\n".join(out) + "\n")
PY
ls $P/${tag}_* | wc -l
