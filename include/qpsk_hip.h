/*
 * qpsk_hip.h -- C ABI of libqpsk_hip.so, the MI355X (gfx950) implementation of
 * the MonsieurETM/QPSK receive path:
 *
 *     rrc_fir()  ->  timing estimate  ->  Costas loop  ->  symbol slicer
 *     (reference qpsk.c:88-218, rrc_fir.c:17-30, costas_loop.c:44-74)
 *
 * Two layers, both plain C (pointers and sizes only, no C++/torch types):
 *
 *  1. qpsk_dropin.h -- the reference's own function signatures (rrc_fir,
 *     rrc_make, the costas_loop.h API, fft/fftn/ifft/ifftn, rx_frame,
 *     qpsk_demod) as process-wide singletons, exactly as the reference has
 *     them; they run on the GPU through layer 2.
 *
 *  2. this file -- context-carrying BATCHED entry points, the form in which
 *     the path is fast: many independent frames (or streams) per call, data
 *     already resident in HBM, one HIP stream per context.
 *
 * Conventions
 *   - complex float  == float[2]  (re, im) == HIP float2   (rrc_fir.h:16)
 *   - complex double == double[2]                          (fft.h:46-49)
 *   - every "d_" pointer is DEVICE memory on the context's GPU; "h_" is host.
 *   - every function returns QPSK_OK (0) or a negative qpsk_status; the text
 *     of the last error of the calling thread is qpsk_last_error().
 *     The reference itself has no error paths (all void); anything that could
 *     only fail here (no GPU, bad shape, HIP error) is reported, never
 *     silently computed on the CPU: there is NO host fallback in this library.
 *   - calls on one context are ordered on its stream; outputs are valid after
 *     qpsk_ctx_sync() (or after synchronising the stream the caller supplied).
 *
 * Stream ordering (the contract)
 *   The library enqueues every kernel and copy of a context on THAT context's stream and on nothing else; it does
 *   not order its work against any other stream of the caller.  So:
 *     - a buffer handed to a call must have been produced on the context's stream, or the caller must have ordered
 *       its producer before the call (event / synchronisation);
 *     - memory that a stream-ordered allocator (PyTorch's caching allocator, hipMallocAsync pools) recycles must be
 *       recycled in the context's stream order too: give the library the allocator's stream (qpsk_ctx_create /
 *       qpsk_ctx_set_stream), or synchronise before handing over recycled memory.  A library stream that the
 *       allocator does not know writes into memory whose previous contents queued kernels of the allocator's stream
 *       still have to read (this build's round-1 abort: DESIGN.md section 1);
 *     - outputs may be read by other streams only after an event / synchronisation on the context's stream.
 *   NULL means the HIP default stream, with its usual implicit ordering against blocking streams.
 */
#ifndef QPSK_HIP_H
#define QPSK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QPSK_NTAPS 127 /* rrc_fir.h:13 */

typedef enum {
    QPSK_OK = 0,
    QPSK_ERR_NO_DEVICE = -1, /* no HIP device / device index out of range */
    QPSK_ERR_ARG = -2,       /* null pointer, non-positive size, frame_size % cycles != 0, ... */
    QPSK_ERR_HIP = -3,       /* a HIP runtime call or a kernel launch failed */
    QPSK_ERR_ALLOC = -4,
    QPSK_ERR_STATE = -5,     /* call sequence error (e.g. stream call on a context made for 0 streams) */
    QPSK_ERR_RANGE = -6      /* a Costas loop phase left the range the bounded 2 pi wrap covers (|phase| > ~25,000 rad:
                                input some 10^5 times the modem's working amplitude, or non-finite).  The reference's
                                phase_wrap() (costas_loop.c:61-67) spins |phase| / 2 pi times there and never returns once
                                |phase| >= 2^27; a GPU wave must not, so the call fails instead */
} qpsk_status;

/* How the decimation offset ("index", qpsk.c:105,173-180,190) is chosen. */
typedef enum {
    QPSK_TIMING_HIST = 0,  /* the reference's amplitude-histogram heuristic, qpsk.c:127-180 */
    QPSK_TIMING_FIXED = 1, /* caller-supplied offset: the bandwidth-bound fused kernel */
    QPSK_TIMING_FFT = 2    /* symbol-rate spectral line of |y|^2 via the radix-2 FFT (new design, no
                              reference counterpart: SURVEY section 0, 8(a) A7) */
} qpsk_timing_mode;

/* The reference's compile-time #defines and main()'s literals as run-time
 * parameters (qpsk.h:16-23, qpsk.c:302,308). */
typedef struct {
    double fs;          /* FS, sample rate in Hz                        qpsk.h:16 */
    double rs;          /* RS, symbol rate in Hz; CYCLES = (int)(fs/rs) qpsk.h:17,21 */
    int frame_size;     /* FRAME_SIZE, complex samples per frame/block  qpsk.h:23 */
    float rrc_alpha;    /* third argument of rrc_make()                 qpsk.c:308 */
    float loop_bw;      /* create_control_loop(loop_bw, min, max)       qpsk.c:302 */
    float min_freq;
    float max_freq;
    int timing_mode;    /* qpsk_timing_mode */
    int fixed_index;    /* used by QPSK_TIMING_FIXED, 0 <= fixed_index */
} qpsk_params;

typedef struct qpsk_ctx qpsk_ctx;

const char *qpsk_last_error(void);
const char *qpsk_version(void);
int qpsk_device_count(void);

/* main()'s defaults: FS 9600, RS 2400, FRAME_SIZE 512, alpha .35, loop TAU/100, clamp +-1 (qpsk.c:302,308) */
void qpsk_params_default(qpsk_params *p);

/* device < 0: current HIP device.  stream: the hipStream_t every call of this context is enqueued on;
 * NULL = the HIP default stream (the caller owns the stream and keeps it alive). */
int qpsk_ctx_create(qpsk_ctx **out, int device, const qpsk_params *p, void *stream);
void qpsk_ctx_destroy(qpsk_ctx *ctx);
int qpsk_ctx_sync(qpsk_ctx *ctx);
/* the kernels' status word WITHOUT a synchronisation: QPSK_OK, or what a kernel that has completed so far flagged (as qpsk_ctx_sync()
 * would report it).  For callers that order their own streams and events against the context's (include below: MULTI does). */
int qpsk_ctx_check(qpsk_ctx *ctx);
int qpsk_ctx_set_stream(qpsk_ctx *ctx, void *stream);
/* Kernel-geometry selection for tests and measurements (never needed for results: every geometry computes the same
 * bits).  Names: "QPSK_PIPE_V", "QPSK_PIPE_G", "QPSK_PIPE_NF", "QPSK_PIPE_LAYOUT_LO", "QPSK_PIPE_LAYOUT_HI", "QPSK_PIPE_DBG" (layout
 * bits, csrc/kernels.h), "QPSK_FUSED_G", "QPSK_FUSED_S", "QPSK_FUSED_LDS", "QPSK_FUSED_GENERIC", "QPSK_HIST_GENERIC" (1: histogram
 * timing through the rrc_fir + scan kernels instead of the fused scan kernel, 2: with the any-CYCLES scan), "QPSK_FIR_GENERIC" (1: the
 * compiler-scheduled full-rate filter and LDS-tap streams also for symmetric taps), "QPSK_FFT_FUSED" (0: the FFT timing estimate always as
 * a launch of its own), "QPSK_STREAM_BLOCK" (0: streams never take the one-launch-per-block kernel, 1: whenever the shape allows),
 * "QPSK_STREAM_POLL" (0: qpsk_streams_rx_pcm_host waits with hipStreamSynchronize), "QPSK_STREAM_SCAN" (streams with histogram
 * timing: 1 = mixer + filter + scan as one kernel whatever the stream count, 0 = never), "QPSK_STREAM_CARRIER" (0: that kernel runs every
 * stream's carrier recurrence although all streams share one), "QPSK_LEAN_DMA" (0: rx_lean_kernel stages its filter windows through
 * registers even where it would use LDS-DMA: frames with an even timing offset; 2: LDS-DMA, but one window per FIR wave even where the LDS
 * has room for one per two-frame unit), "QPSK_HIST_ONEPASS" (histogram timing: 0 = always the two-launch route -- timing scan, then the receive kernel -- 1 = the one-pass
 * route, rx_hist_kernel on the previous batch's majority index plus a fall-back pass over the frames it missed, whenever a guess exists;
 * unset: that route only while every frame of the context's last histogram-mode batch sat on that batch's majority index -- a missed
 * frame costs more than the route saves), "QPSK_EST_WAVES" (hardware waves that share the
 * in-launch FFT timing estimate), "QPSK_LEAN_PAIR" (rx_lean_kernel's serial wave: 0 = one lane per Costas loop, 1 = two lanes per
 * loop -- they share the step's sine / cosine polynomial chains -- in workgroups of up to 16 frames, 2 = up to 32; the library's own choice is up to 24); value < 0 = back to
 * the library's own choice.
 * Environment variables of the same names are read once, by qpsk_ctx_create(), as the context's initial values;
 * no other call reads the environment, and none of them can change a result. */
int qpsk_ctx_set_tuning(qpsk_ctx *ctx, const char *name, int value);
int qpsk_ctx_cycles(const qpsk_ctx *ctx);   /* CYCLES */
int qpsk_ctx_nsym(const qpsk_ctx *ctx);     /* FRAME_SIZE / CYCLES */
/* Name of the receive kernel the context's last qpsk_rx_batch() / qpsk_rx_batch_bw() launched ("" before the first
 * call): which of the library's geometries served that batch shape.  For measurement records (bench.py), not
 * results -- every geometry computes the same bits.  The string is static storage. */
const char *qpsk_ctx_last_kernel(const qpsk_ctx *ctx);

/* Host-side copies of what rrc_make() / create_control_loop() produced for this context. */
int qpsk_ctx_get_taps(const qpsk_ctx *ctx, float h_taps[QPSK_NTAPS]);
int qpsk_ctx_get_gains(const qpsk_ctx *ctx, float *h_alpha, float *h_beta);
/* Replace them (the reference lets the caller do both: rrc_make(), set_alpha()/set_beta()). */
int qpsk_ctx_set_taps(qpsk_ctx *ctx, const float h_taps[QPSK_NTAPS]);
int qpsk_ctx_set_loop(qpsk_ctx *ctx, float alpha, float beta, float min_freq, float max_freq);

/* -------------------------------------------------------------------------
 * Batch of INDEPENDENT frames -- the hot path.
 *
 * For each of nframes frames of frame_size complex samples this is, bit for
 * bit, what the reference computes with a fresh process:
 *       rx_frame(frame); rx_frame(zeros);      (qpsk.c:88-218; complex input
 * enters at the rrc_fir() call, qpsk.c:125; the second call is the flush that
 * the one-block pipeline delay of qpsk.c:186-197 needs)
 * taking costas_frame[], the slicer bits and the loop state after the second
 * call.
 *
 *   d_in      [nframes][frame_size] complex float
 *   d_sym     [nframes][nsym] uint8   (bits[1]<<1)|bits[0] of qpsk_demod(), qpsk.c:74-79,270
 *   d_freq    [nframes] float         get_frequency() after the frame  (rad/symbol)
 *   d_phase   [nframes] float         get_phase()
 *   d_costas  [nframes][nsym] complex float, costas_frame[] (qpsk.c:197)   -- may be NULL
 *   d_index   [nframes] int32, the decimation offset used                  -- may be NULL
 *   d_hz      [nframes] float, fbb_offset_freq = freq*RS/TAU (qpsk.c:217)  -- may be NULL
 * ------------------------------------------------------------------------- */
int qpsk_rx_batch(qpsk_ctx *ctx, const float *d_in, int nframes, uint8_t *d_sym, float *d_freq,
                  float *d_phase, float *d_costas, int32_t *d_index, float *d_hz);

/* The same with the frames frame_pitch complex samples apart in d_in (frame_pitch >= frame_size, even; the samples between
 * two frames are never read).  Why a caller would: with frames a power of two apart (16384 samples = 128 KB) every frame's
 * sample n sits in the same HBM channel group, and a batch kernel streams sample n of ALL its frames at about the same time --
 * measured on MI355X, the memory side then delivers 4.4 TB/s to this access pattern against 5.2 TB/s at a pitch of
 * frame_size + 512 samples.  Every timing mode. */
int qpsk_rx_batch_pitched(qpsk_ctx *ctx, const float *d_in, long long frame_pitch, int nframes, uint8_t *d_sym,
                          float *d_freq, float *d_phase, float *d_costas, int32_t *d_index, float *d_hz);

/* The same with nbw Costas loops per frame sharing one FIR pass (loop
 * bandwidth sweep, README.md:12).  Outputs are [nframes][nbw][...]. */
int qpsk_rx_batch_bw(qpsk_ctx *ctx, const float *d_in, int nframes, const float *h_loop_bw, int nbw,
                     uint8_t *d_sym, float *d_freq, float *d_phase, int32_t *d_index);

/* -------------------------------------------------------------------------
 * The stages on their own (each is what the corresponding reference function
 * computes, batched).
 * ------------------------------------------------------------------------- */

/* rrc_fir() (rrc_fir.c:17-30) on nframes independent delay lines.
 *   d_memory [nframes][127] complex float  in/out (may be NULL: zero history, not written back)
 *   d_in, d_out [nframes][length] complex float; d_out may equal d_in only if nframes*length
 *   fits the library's staging (it copies first); prefer separate buffers. */
int qpsk_rrc_fir_batch(qpsk_ctx *ctx, float *d_memory, const float *d_in, float *d_out, int nframes, int length);

/* The same filter by overlap-save with 512-point FFTs (algorithms/fft.h:44) -- FAST, NOT EXACT: the
 * transform changes the summation order of rrc_fir.c:22-26, so the output agrees with qpsk_rrc_fir_batch to ~1e-6 of the
 * frame's peak (bounds asserted in tests/test_gpu_parity.py::test_rrc_fir_fast_error_bounds), not bit for bit.  The
 * library itself never uses it: qpsk_rx_batch and the streams keep the exact kernels, whose symbols sit on decision
 * boundaries.  d_out must not alias d_in; d_memory as above (updated exactly: it is a copy of input samples). */
int qpsk_rrc_fir_batch_fast(qpsk_ctx *ctx, float *d_memory, const float *d_in, float *d_out, int nframes, int length);

/* timing histogram (qpsk.c:127-180) of nframes filtered blocks -> d_index[nframes]; d_hist, if not NULL,
 * receives hist_i[k] + hist_q[k], k = 0..7 (qpsk.c:175, locals of rx_frame) as [nframes][8] */
int qpsk_timing_hist_batch(qpsk_ctx *ctx, const float *d_filtered, int nframes, int32_t *d_index, int32_t *d_hist);

/* The histogram timing estimate (qpsk.c:127-180) straight from UNFILTERED frames: rrc_fir() with a fresh delay line
 * and the scan fused in one kernel, the filtered samples never leaving the CU (what qpsk_rx_batch runs in
 * QPSK_TIMING_HIST mode).  Needs CYCLES = 8, frame_size a multiple of 256, 16-byte aligned input; same results as
 * qpsk_rrc_fir_batch() followed by qpsk_timing_hist_batch().  d_index [nframes]; d_hist [nframes][8] or NULL. */
int qpsk_timing_scan_batch(qpsk_ctx *ctx, const float *d_in, int nframes, int32_t *d_index, int32_t *d_hist);

/* The FFT timing estimate alone (QPSK_TIMING_FFT; NEW DESIGN, the reference never calls fft.c: SURVEY section 0).
 *   d_index     [nframes] int32
 *   d_filtered  [nframes][512] complex float, may be NULL: the 512 rrc_fir() outputs (samples 128..639 of the frame,
 *               fresh delay line) the estimator looks at -- bit for bit what qpsk_rrc_fir_batch() returns there
 *   d_spectrum  [nframes][512] complex double, may be NULL: fftn(|y|^2, 512) as fft.c:110-120 returns it -- bit for
 *               bit qpsk_fft_batch() of the same 512 values
 * so that everything below the final argmax rule is reference-pinned code (rrc_fir.c:17-30, fft.c:98-120). */
int qpsk_timing_fft_batch(qpsk_ctx *ctx, const float *d_in, int nframes, int32_t *d_index, float *d_filtered,
                          double *d_spectrum);

/* The same estimate by the kernel qpsk_rx_batch() runs in QPSK_TIMING_FFT mode: of the 512-point transform only the
 * butterflies the symbol-rate bin X[512 / CYCLES] depends on are evaluated (511 of the recursion's fft.c:55-63 steps,
 * each with the full transform's operands, twiddle and order).
 *   d_bin  [nframes] complex double, may be NULL: that bin -- bit for bit d_spectrum[f][512 / CYCLES] above. */
int qpsk_timing_fft_bin_batch(qpsk_ctx *ctx, const float *d_in, int nframes, int32_t *d_index, float *d_filtered,
                              double *d_bin);

/* Costas loop + slicer (qpsk.c:196-212) over already decimated symbols.
 *   d_symbols_in [nframes][nsym] complex float;  d_state [nframes][2] float (phase, freq) in/out,
 *   NULL = start from (0,0) and do not write back. */
int qpsk_costas_batch(qpsk_ctx *ctx, const float *d_symbols_in, int nframes, int nsym, float *d_state,
                      uint8_t *d_sym, float *d_costas);

/* fftn()/ifftn() (fft.c:110-136) on nbatch independent length-n transforms, n a power of two up to 2^21 (one
 * workgroup per transform in LDS up to 8192 points, two passes over global memory above).
 *   d_in, d_out [nbatch][n] complex double, may be the same array; forward is scaled by 1/n, inverse is not
 *   (fft.c:105-107). */
int qpsk_fft_batch(qpsk_ctx *ctx, const double *d_in, double *d_out, int nbatch, int n, int inverse);

/* -------------------------------------------------------------------------
 * STREAMS: nstreams modems advancing one block per call with all state
 * carried, i.e. consecutive rx_frame() calls (qpsk.c:344-354): FIR delay
 * line, previous block's symbols, Costas phase/frequency, mixer phase.
 * Results are those of the PREVIOUS block (qpsk.c:186-197), as in the
 * reference.
 *
 * Error contract (all-or-poisoned).  A stream call enqueues several kernels.  If it fails between its launches (QPSK_ERR_HIP,
 * QPSK_ERR_ALLOC), or a kernel of a stream call reports that it gave up a bounded wait (QPSK_ERR_HIP from the call that
 * synchronises next), the carried state of ALL the context's streams is undefined: every later stream call -- the three
 * qpsk_streams_rx_*() and qpsk_streams_set/get_loop_state() -- returns QPSK_ERR_STATE until qpsk_streams_reset() has
 * completed successfully (a reset that itself fails leaves them refused).  Argument errors and QPSK_ERR_RANGE (a NaN / Inf
 * sample, a loop phase beyond the bounded wrap: flagged NUMBERS, the kernels completed) do not poison, and neither does a
 * failing batch call while no stream work is in flight.
 * ------------------------------------------------------------------------- */
int qpsk_streams_reset(qpsk_ctx *ctx, int nstreams, double mixer_hz);
/* the carried Costas state, h_state[nstreams][2] = (phase, freq): set_phase()/set_frequency() and
 * get_phase()/get_frequency() for every stream at once (costas_loop.c:117-132,148-150) */
int qpsk_streams_set_loop_state(qpsk_ctx *ctx, const float *h_state);
int qpsk_streams_get_loop_state(qpsk_ctx *ctx, float *h_state);
/* complex input (enters at qpsk.c:125) */
int qpsk_streams_rx_cplx(qpsk_ctx *ctx, const float *d_in, uint8_t *d_sym, float *d_freq, float *d_phase,
                         float *d_costas, int32_t *d_index);
/* int16 PCM input, mixed to complex on the GPU (qpsk.c:114-120) */
int qpsk_streams_rx_pcm(qpsk_ctx *ctx, const int16_t *d_pcm, uint8_t *d_sym, float *d_freq, float *d_phase,
                        float *d_costas, int32_t *d_index);
/* The same with HOST buffers on both sides -- the reference's call pattern (qpsk.c:344-354: fread a block, rx_frame()):
 * one pinned copy up (PCM + loop state), the kernels, one copy down (symbols, costas_frame[], loop state, index), ONE
 * synchronisation.  h_pcm [nstreams][frame_size]; h_loop_io [nstreams][2] (phase, freq), read before and written
 * after the block, NULL = the carried state; h_sym [nstreams][nsym]; h_costas [nstreams][nsym][2] or NULL;
 * h_index [nstreams] or NULL.  This is what the drop-in rx_frame() runs on. */
int qpsk_streams_rx_pcm_host(qpsk_ctx *ctx, const int16_t *h_pcm, float *h_loop_io, uint8_t *h_sym, float *h_costas,
                             int32_t *h_index);

/* -------------------------------------------------------------------------
 * TRANSMITTERS: nstreams independent modulators advancing
 * one block per call with the reference's carried state -- the tx_filter
 * delay line (rrc_fir.c:17, qpsk.c:243) and the carrier phase fbb_tx_phase
 * (qpsk.c:45,249-253).  One call = one qpsk_packet_mod() (qpsk.c:273-285)
 * per transmitter.
 * ------------------------------------------------------------------------- */
/* fbb_tx_phase = cmplx(0.0f); fbb_tx_rect = cmplx(TAU * tx_hz / FS); tx_filter zeroed (qpsk.c:316,320;
 * the shipped main() uses tx_hz = CENTER + 50.0) */
int qpsk_tx_reset(qpsk_ctx *ctx, int nstreams, double tx_hz);
/* d_symbols [nstreams][nsym] uint8, the dibit (tx_bits[s] << 1) | tx_bits[s+1] of qpsk.c:277-281 (the value
 * qpsk_rx_batch writes for the same symbol); d_pcm [nstreams][nsym*CYCLES] int16 as tx_frame() returns them
 * (qpsk.c:259-261), may be NULL; d_baseband [nstreams][nsym*CYCLES][2] float, the shaped complex signal
 * before the up-mix (qpsk.c:243), may be NULL -- not both */
int qpsk_tx_symbols(qpsk_ctx *ctx, const uint8_t *d_symbols, int nsym, int16_t *d_pcm, float *d_baseband);

/* -------------------------------------------------------------------------
 * Bit-level stages after the slicer (algorithms/ of the reference, which its qpsk.c does
 * not call yet), batched over independent packets.
 * ------------------------------------------------------------------------- */

/* crc16() (crc16.c:11-23: init 0xFFFF, polynomial 0x1021, no reflection, no final xor) of npackets packets of
 * nbytes bytes each: d_data [npackets][nbytes] -> d_crc [npackets] uint16 */
int qpsk_crc16_batch(qpsk_ctx *ctx, const uint8_t *d_data, int npackets, int nbytes, uint16_t *d_crc);

/* interleave() (interleave.c:33-78), in place on each packet of nbytes bytes: bit i goes to bit (b*i) mod nbits,
 * b = the largest table prime below nbits (the table ends at 347); dir 0 = INTERLEAVE, 1 = DEINTERLEAVE
 * (interleave.h:10-11).  nbytes*8 must be < 65536 as in the reference (uint16_t nbits). */
int qpsk_interleave_batch(qpsk_ctx *ctx, uint8_t *d_data, int npackets, int nbytes, int dir);

/* scramble() (bit-scramble.c:57-84) on every 2-bit symbol of npackets frames of nsym symbols, in place, the
 * 15-bit register reloaded with SEED 0x4A80 at the start of each frame (bit-scramble.c:11-13 "The Sync Seed is
 * reset at the start of each frame"); additive, so scrambling twice restores the input. */
int qpsk_scramble_batch(qpsk_ctx *ctx, uint8_t *d_sym, int npackets, int nsym);

/* -------------------------------------------------------------------------
 * MULTI: a batch of independent frames sharded over the GPUs of one node
 * (SURVEY.md 8(e)).  It stands where the reference has
 *     while (fread(frame, ...)) rx_frame(frame);                 qpsk.c:344-354
 * over process-global per-frame state (qpsk.c:36-53, costas_loop.c:13-23):
 * every frame is its own modem here, shard r of N takes the contiguous
 * frames [r F / N, (r + 1) F / N), one qpsk_ctx + one host thread + two
 * streams per shard, NO collective and no traffic between devices.  Results
 * (1 byte per symbol, freq and phase per frame) come back per device over
 * PCIe into pinned memory on the shard's second stream -- while the next
 * step's kernel runs -- and are written at the shard's place of the caller's
 * host arrays.
 *
 *   devices[ndev]   HIP ordinals, one shard each; repeats are allowed (two
 *                   shards on one GPU: a rehearsal on a one-GPU box)
 *   qpsk_multi_load total_frames frames of frame_size complex samples;
 *                   h_in = [total_frames][frame_size][2] float on the host
 *                   (uploaded, shard by shard) or NULL = the shards' device
 *                   buffers are allocated and left to the caller
 *                   (qpsk_multi_shard returns them; or
 *                   qpsk_multi_use_device_input lends one of the caller's)
 *   rx_begin(slot)  every shard: qpsk_rx_batch into result slot 0 / 1, then
 *                   its copy-back; returns once everything is ENQUEUED
 *   rx_end(slot, h_sym [total][nsym], h_freq [total], h_phase [total])
 *                   waits for that slot's copy-back on every device, checks
 *                   the kernels' status, concatenates (any of the three may
 *                   be NULL)
 * Pipelined: begin(0); begin(1); end(0); begin(0); end(1); ... -- the
 * copy-back of a step overlaps the kernel of the next.  A 16384-sample
 * frame returns 2056 bytes: 16.1 MiB per 8192-frame step, ~0.3 ms over PCIe
 * Gen5 x16 -- as long as the step's kernel; the copy-back, not the kernel,
 * bounds a host that wants every step's symbols (examples/shard_devices.c,
 * bench.py `gather`).  One caller thread at a time per qpsk_multi (the
 * object runs its own thread per shard; its entry points are not reentrant).
 * ------------------------------------------------------------------------- */
typedef struct qpsk_multi qpsk_multi;
int qpsk_multi_create(qpsk_multi **out, const int *devices, int ndev, const qpsk_params *p);
void qpsk_multi_destroy(qpsk_multi *mj);
int qpsk_multi_shards(const qpsk_multi *mj);
/* (a failed load -- QPSK_ERR_ALLOC -- frees what it had allocated and leaves the job without frames: rx_begin then returns QPSK_ERR_ARG) */
int qpsk_multi_load(qpsk_multi *mj, long long total_frames, const float *h_in);
/* shard r: its device, first frame and frame count, context and device input buffer (any pointer may be NULL) */
int qpsk_multi_shard(qpsk_multi *mj, int r, int *device, long long *first, long long *count, qpsk_ctx **ctx, float **d_in);
int qpsk_multi_use_device_input(qpsk_multi *mj, int r, const float *d_in);
/* Direct mode for a slot: its copy-back goes by DMA straight to the caller's arrays (each shard at its place) instead of to the library's
 * pinned staging, and qpsk_multi_rx_end(slot, NULL, NULL, NULL) only waits -- no concatenating memcpy on the host (16 MiB per 8192-frame
 * step: ~1.4 ms of one host core, five times the kernel).  The arrays must be page-locked (qpsk_host_alloc) and stay valid while the
 * slot is used; all NULL = back to staging. */
int qpsk_multi_set_direct_output(qpsk_multi *mj, int slot, uint8_t *h_sym, float *h_freq, float *h_phase);
/* Packed mode: the symbols come back FOUR PER BYTE -- h_sym rows of ceil(nsym / 4) bytes, byte k = sym[4k] | sym[4k+1] << 2 | sym[4k+2] << 4 |
 * sym[4k+3] << 6 (qpsk_pack_symbols on the device, 16 MiB -> 4 MiB per 8192-frame step): the copy-back drops under the kernel's time and
 * a gathered step costs what the kernel costs.  qpsk_unpack_symbols_host() gives a byte per symbol again where a caller wants it. */
int qpsk_multi_set_packed(qpsk_multi *mj, int on);
int qpsk_pack_symbols(qpsk_ctx *ctx, const uint8_t *d_sym, long long nrows, int nsym, uint8_t *d_packed);
int qpsk_unpack_symbols_host(const uint8_t *h_packed, long long nrows, int nsym, uint8_t *h_sym);
int qpsk_host_alloc(void **h_ptr, size_t bytes);      /* page-locked host memory, usable from every device */
int qpsk_host_free(void *h_ptr);
int qpsk_multi_rx_begin(qpsk_multi *mj, int slot);
int qpsk_multi_rx_end(qpsk_multi *mj, int slot, uint8_t *h_sym, float *h_freq, float *h_phase);

/* -------------------------------------------------------------------------
 * Small helpers so that a C host needs nothing but this library.
 * ------------------------------------------------------------------------- */
int qpsk_dev_alloc(qpsk_ctx *ctx, void **d_ptr, size_t bytes);
int qpsk_dev_free(qpsk_ctx *ctx, void *d_ptr);
int qpsk_dev_upload(qpsk_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int qpsk_dev_download(qpsk_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);

/* Self-test hook: order-independent 64-bit hash of the device sin/cos (the glibc-exact routine the
 * Costas kernel uses) over the float bit patterns [first, first+count) in both signs; tests compare
 * it with the same hash of the CPU oracle over the same range. */
int qpsk_selftest_sincos_hash(qpsk_ctx *ctx, uint32_t first, uint32_t count, unsigned long long *h_out);

/* Test hook (the stream error paths, tests/test_gpu_parity.py): stores `code` in the context's kernel status word, as a kernel that gave
 * up (1), left the bounded phase range (2) or ended on a non-finite loop state (3) would; the context's next synchronising call reports it. */
int qpsk_test_inject_status(qpsk_ctx *ctx, int code);
/* Test hook (the one-pass histogram route): synchronises, then out[5] = {the guess the next histogram-mode call will take (-1: none), frames
 * the last one-pass call's guess missed, and the statistics the host steers by: majority index, frames, frames off the majority or missed by the guess} */
int qpsk_test_hist_state(qpsk_ctx *ctx, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* QPSK_HIP_H */
