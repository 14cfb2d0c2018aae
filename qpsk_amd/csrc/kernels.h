/* kernels.h -- launch interface between the C-ABI layer (api.cpp) and kernels.hip */
#ifndef QPSK_KERNELS_H
#define QPSK_KERNELS_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace qpsk {

constexpr int LOOKBACK = 128;  /* >= 126 samples of FIR history, kept 16-byte friendly */
constexpr int LOOKAHEAD = 8;   /* the decimation offset can reach 7 (qpsk.c:173-180) */
constexpr int MAX_LDS_BYTES = 160 * 1024;
constexpr int MAX_INDEX = 7;
/* values a kernel stores in the context's status word (FusedArgs::status; api.cpp check_status) */
constexpr int STATUS_PIPE_TIMEOUT = 1;   /* the in-LDS producer/consumer pipeline exhausted its bounded spins */
constexpr int STATUS_PHASE_RANGE = 2;    /* a loop phase beyond the bounded 2 pi wrap (qpsk_device.h, phase_wrap) */
constexpr int STATUS_NONFINITE = 3;      /* a loop ended on a NaN / Inf phase or frequency: the input held a non-finite sample */

struct FusedArgs {
    const float2 *x;        /* [nframes] frames of frame_size samples, frame_pitch samples apart */
    size_t frame_pitch;     /* >= frame_size (qpsk_rx_batch: = frame_size) */
    int nframes, frame_size, cycles, nsym;
    int G, S;               /* frames per workgroup, symbols per chunk */
    int mixed;              /* rx_fused_pipe_kernel, set by its launcher: 0 = every FIR wave filters 4 frames, 4 symbols per
                               lane; 1 = 16 frames per workgroup, the last four by two waves of 2 frames with 2 symbols
                               per lane; 2 = every FIR wave 2 frames with 2 symbols per lane (see the kernel) */
    int share_simd0;        /* rx_pipe2_kernel, set by its launcher: a FIR wave sits beside the serial wave (priorities, see there) */
    const int32_t *index;   /* [nframes] or NULL -> fixed_index */
    int fixed_index;
    /* rx_fused_pipe_kernel, full 16-frame workgroups only (set by api.cpp when the shape allows it): the FFT timing estimate
     * runs INSIDE the launch -- every hardware wave estimates two of the workgroup's frames before the pipeline starts
     * (timing_fft_wave.h) -- instead of in a launch of its own in front.  est_tw: size-512 twiddle table, est_cs: the CYCLES
     * candidate phases (both host-built, api.cpp); index_out [nframes] or NULL receives the indices.  NULL = off */
    const double2 *est_tw, *est_cs;
    int32_t *index_out;
    int lean_twowin;        /* rx_lean_kernel, set by launch_rx_lean when the LDS allows it (workgroups of up to 16 frames): every two-frame unit has
                               a window of its own, so a two-unit FIR wave's DMAs run a whole unit ahead (fir_lean_loop2_dma2w) */
    int lean_dma;           /* rx_lean_kernel: 1 = FIR waves whose frames all have an even decimation offset stage their windows by LDS-DMA
                               (fir_lean_asm.h, the _dma loops); 0 = always through registers.  Same bits either way ("QPSK_LEAN_DMA") */
    int est_waves;          /* rx_lean_kernel with the FFT timing estimate inside the launch: hardware waves launched for it (0 = the library's 12, the most the register budget allows) ("QPSK_EST_WAVES") */
    int lean_pair;          /* rx_lean_kernel: the serial wave runs every loop in TWO lanes that share the step's sin/cos polynomial chains
                               (costas_asm.h, QPSK_BODY_P): 1 = in workgroups of up to 16 frames, 2 = up to 32, 3 = up to 24 (the library's rule), 0 = never; launch_rx_lean turns the wish into 0 / 1.  Same bits ("QPSK_LEAN_PAIR") */
    int dbg;                /* layout variants of the pipeline kernel, all bit-exact (qpsk_ctx_set_tuning "QPSK_PIPE_DBG"):
                               4 no spare waves, 8 C++ Costas step, 16 serial wave chunk by chunk (no stream across the ring hand-overs), 64 FIR waves that share a SIMD keep their hardware order (16-frame kernel), 128 one lane mapping for all FIR waves (the plain
                               layout).  Measurement build only (-DQPSK_PIPE_PROFILE; masked off by api.cpp otherwise):
                               1 skip FIR arithmetic, 2 skip the Costas recurrence (both change the result), 32 print the
                               cycle accounting of workgroup 0's FIR waves */
    const float2 *dsrc;     /* costas_pipe_kernel only: decimated symbols, rows dstride symbols apart */
    int dstride;
    /* costas_pipe_kernel, streaming mode (qpsk.c:186-191): once symbol i of a row has been taken, its slot is
     * refilled with the next block's pick refill[frame][i*cycles + index[frame]] (0 past the block);
     * refill rows are frame_size samples apart.  NULL = leave dsrc alone */
    const float2 *refill;
    float2 *refill_dst;     /* = dsrc, writable */
    int refill_planar;      /* refill rows are [cycles][nsym] planes (phase-major: what stream_scan_kernel leaves) instead of frame_size samples */
    /* rx_fused_kernel as the fall-back pass of the one-pass histogram route (rx_hist_kernel, rx_fused.hip): the launch covers the frames
     * frame_list[0 .. *frame_list_count) instead of 0 .. nframes (both in device memory: the count is known only on the device; workgroups
     * beyond it retire at once); inputs, per-frame index and every output are addressed by the listed frame number.  NULL = off */
    const int32_t *frame_list, *frame_list_count;
    const float *taps;      /* [127] */
    const float *gains;     /* [nbw][2] alpha, beta */
    int nbw;
    float min_freq, max_freq;
    double rs;
    const float *state_in;  /* [nframes][nbw][2] phase, freq or NULL */
    float *state_out;       /* same or NULL */
    uint8_t *sym;           /* [nframes][nbw][nsym] */
    uint8_t *sym_pad;       /* rx_lean_kernel on a batch whose last workgroup is not full: [G][nsym] bytes for the pad frames' symbols (and for
                               the last real frame of an odd batch, whose two-frame unit reaches past the end); NULL = whole workgroups only */
    float *freq, *phase;    /* [nframes][nbw] or NULL */
    float2 *costas;         /* [nframes][nbw][nsym] or NULL */
    float *hz;              /* [nframes][nbw] or NULL */
    int *status;            /* the context's status word (pinned host memory): a kernel stores STATUS_* there when a
                               call's results are invalid (pipeline timeout, phase beyond the bounded 2 pi wrap) */
    /* costas_pipe_kernel behind stream_scan_kernel MODE 2 (carrier.h): a spare wave of workgroup 0 finishes the carrier table of the
     * next block.  carrier_state NULL = no spare wave */
    float *carrier_state;
    float2 *carrier_tab;
    int carrier_frame;      /* samples per block of that table */
};

size_t fused_lds_bytes(int G, int S, int cycles, int nbw);
int prepare_kernels(void);
int launch_rx_fused(const FusedArgs &a, hipStream_t s);
/* rx_fused.hip: the producer/consumer pipeline kernel (CYCLES = 8 only) */
size_t pipe_lds_bytes(int NF, int nbw);
int pipe_frames(int NF);               /* frames of a workgroup with NF FIR waves: 4 per wave */
int pipe_cycles(void);
int pipe_max_nf(void);                 /* FIR waves per workgroup: 4 (16 frames) */
int prepare_pipe_kernel(void);
int launch_rx_fused_pipe(const FusedArgs &a, int NF, int *status, hipStream_t s);
int launch_costas_pipe(const FusedArgs &a, int NF, int *status, hipStream_t s);
/* rx_fused.hip: the pipeline kernel for up to 32 frames per workgroup (two-frame units, per-wave windows) */
size_t pipe2_lds_bytes(int G, int nwin, int nbw);      /* nwin = FIR waves (one window each) */
int pipe2_max_fir(void);
int pipe2_max_frames(void);
int pipe2_max_units_per_wave(void);
int pipe2_max_hw_waves(void);
unsigned long long pipe2_default_layout(int NU);        /* 4 bits per hardware wave: units it owns; 0 if NU does not fit */
int launch_rx_pipe2(const FusedArgs &a, int G, unsigned long long layout, int *status, hipStream_t s);
/* rx_fused.hip: the same pipeline with the FIR waves' chunk loop as one hand-written stream (fir_lean_asm.h) */
size_t lean_lds_bytes(int G, int nwin);
unsigned long long lean_default_layout(int NU);         /* rx_lean_kernel: 1, 5, 5, 5 units on SIMDs 0-3 for a full workgroup */
bool lean_shape_ok(const FusedArgs &a, int G);          /* one loop per frame, whole chunks, whole even workgroups, ... */
int launch_rx_lean(const FusedArgs &a, int G, unsigned long long layout, int *status, hipStream_t s);
bool lean_est_ok(const FusedArgs &a, int G, unsigned long long layout);   /* the FFT timing estimate inside rx_lean_kernel's launch fits this geometry */
/* rx_fused.hip: the reference's histogram timing mode in ONE pass (timing scan + receive path on a guessed index, verified per frame) */
bool rx_hist_shape_ok(const FusedArgs &a);
int launch_rx_hist(const FusedArgs &a, int32_t *index_true, const int32_t *hint, int32_t *mis_list, int32_t *mis_count, int *status, hipStream_t s);
int launch_index_majority(const int32_t *index, int nframes, int32_t *hint, int32_t *mis_count, int32_t *h_stats, hipStream_t s);
int launch_rrc_fir(const float *x, const float *memory, float *y, const float *taps, int nframes, int length,
                   hipStream_t s, size_t in_pitch = 0);      /* in_pitch: samples between input frames (0 = length) */
/* firstream.hip: the same filter as the generated stream fir_full8s_asm.h (SYMMETRIC taps only: the caller checks) */
int launch_rrc_fir_stream(const float *x, const float *memory, float *y, const float *taps, int nframes, int length,
                          hipStream_t s, size_t in_pitch = 0, int ncu = 0);
int launch_delay_line(const float *x, float *memory, int nframes, int length, hipStream_t s);
/* firfast.hip: overlap-save FIR, 512-point fp32 FFTs (not a parity path); H, tw: [512][2] floats from qpsk_host_fir_fast_tables */
int launch_rrc_fir_fast(const float *x, const float *memory, float *y, const float *H, const float *tw, int nframes, int length,
                        hipStream_t s);
int rrc_fir_fast_hop(void);
int rrc_fir_fast_nfft(void);
int launch_timing_hist(const float *y, int nframes, int frame_size, int cycles, int32_t *index, int32_t *hist,
                       bool generic, hipStream_t s);   /* generic: always the any-CYCLES scan (test knob) */
int launch_costas(const float *d, int nframes, int nsym, int dstride, int nbw, const float *gains, float min_freq,
                  float max_freq, const float *state_in, float *state_out, uint8_t *sym, float *costas, int *status,
                  hipStream_t s);
int launch_decimate(const float *filtered, const int32_t *index, float *dec, int nstreams, int frame_size,
                    int cycles, int nsym, hipStream_t s);
int launch_mixer(const int16_t *pcm, float *out, float *state, int nstreams, int frame_size, hipStream_t s);
int launch_fft(const double *in, double *out, const double *tw, int nbatch, int n, int log2n, int inverse,
               hipStream_t s);                 /* n <= fft_lds_max_n(): one workgroup per transform, LDS-resident */
int fft_lds_max_n(void);
int fft_big_block(void);
int launch_fft_big(const double *in, double *out, const double *twb, const double *twn, int nbatch, int n, int log2n,
                   int inverse, hipStream_t s);   /* larger powers of two, two passes; in != out */
/* timing_scan.hip: full-rate FIR + histogram scan fused through LDS (CYCLES = 8, frame_size % timing_scan_tile() == 0) */
size_t timing_scan_lds_bytes(void);
int timing_scan_tile(void);
int prepare_timing_scan(void);
int launch_timing_scan(const float *x, int nframes, int frame_size, const float *taps, int32_t *index, int32_t *hist,
                       int *status, hipStream_t s, size_t pitch = 0, bool symmetric = false);   /* symmetric taps: the SGPR-tap stream */
/* timing_fft.hip */
int launch_timing_fft(const float *x, int nframes, int frame_size, int cycles, const float *taps, const double *tw,
                      const double *cs, int32_t *index, float *yout, double *Xout, double *Xk, hipStream_t s, size_t pitch = 0,
                      int ncu = 0, bool symmetric = false);   /* optional, for the parity tests: yout [nframes][512][2] the estimator's filtered samples,
                      Xout [nframes][512][2] the whole spectrum (selects the variant that runs the full transform beside the
                      pruned one), Xk [nframes][2] the symbol-rate bin as the pruned transform delivers it */
int timing_fft_nfft(void);
int timing_fft_first(void);
/* streamblock.hip: one launch per rx_frame() block and stream (few, short streams: the reference's call pattern) */
struct StreamBlockArgs {
    const int16_t *pcm;     /* [n][frame_size] PCM (device memory or mapped pinned host memory), or NULL: */
    const float2 *cplx;     /* [n][frame_size] complex samples at the rrc_fir() call (qpsk.c:125) */
    float *mixer;           /* [n][4] carrier phase and step (PCM input), in/out */
    float2 *memory;         /* [n][127] delay lines, in/out */
    float2 *dec;            /* [n][nsym] decimated_frame[]: the previous block's picks in, this block's out */
    float *loop;            /* [n][2] phase, freq in/out */
    const float *loop_in;   /* [n][2] or NULL: loop state to start from instead of `loop` (host staging) */
    float *loop_out;        /* [n][2] or NULL: a second copy of the final state (host staging) */
    const float *taps, *gains;   /* [127]; [2] alpha, beta */
    float min_freq, max_freq;
    int frame_size, cycles, nsym, hist_timing, fixed_index;
    uint8_t *sym;           /* [n][nsym] */
    float2 *costas;         /* [n][nsym] or NULL */
    int32_t *index;         /* [n] or NULL */
    float2 *filtered;       /* [n][frame_size] or NULL */
    int *status;
    unsigned *done;         /* mapped pinned host counter or NULL: every wave adds 1 behind its last store (system scope), so the host
                               can watch the block finish instead of going through the stream's completion signal */
    /* the streams' ONE carrier (carrier.h; set together or not at all): this block's phases from ctab instead of each stream's own
     * recurrence (the longest stretch of wave 1's chain), the next block's into ctab_next by a third wave of workgroup 0 beside the
     * Costas wave; `mixer` is then neither read nor written */
    const float2 *ctab;
    float2 *ctab_next;
    float *cstate;
};
/* Blocks small enough travel INSIDE the kernel arguments (the dispatch packet's argument buffer is device memory the host writes
 * through the PCIe aperture: posted writes), so the kernel never reads host memory -- on this pool a read of pinned host memory
 * from the GPU takes 8-14 us, the whole budget of a 512-sample block [measured, profiles/r04_stream_block.txt] */
struct StreamBlockInline {
    static constexpr int MAX_STREAMS = 8, MAX_SAMPLES = 1024;      /* n <= 8 streams, n * frame_size <= 1024 samples (2 KB) */
    float loop[2 * MAX_STREAMS];
    uint4 pcm[MAX_SAMPLES / 8];
};
size_t stream_block_lds_bytes(int frame_size, int nsym);
int stream_block_max_frame(void);
int prepare_stream_block(void);
int launch_stream_block(const StreamBlockArgs &a, int nstreams, hipStream_t s, const StreamBlockInline *inl = nullptr);   /* inl: PCM and, if
                        a.loop_in is set (to anything), the loop state come from *inl instead of a.pcm / a.loop_in */
/* streamscan.hip: PCM -> mix -> rrc_fir() -> histogram timing of running streams in one kernel (CYCLES = 8 or 4, frame_size %
 * stream_scan_tile() == 0, symmetric taps); yout [n][cycles][frame_size / cycles] planar by decimation phase; updates mixer [n][4], memory [n][127] */
int stream_scan_tile(void);
int prepare_stream_scan(void);
int launch_stream_scan(const int16_t *pcm, const float *x, float *mixer, float *memory, float *yout, const float *taps, int32_t *index,
                       int nstreams, int frame_size, int *status, hipStream_t s, const float *ctab = nullptr, float *ctab_next = nullptr,
                       float *cstate = nullptr, unsigned cseq = 0, int cycles = 8);
/* the shared carrier of MODE 2 (streamscan.hip): cstate [8] = {phase the next table starts from, fbb_rx_rect, phase the last table
 * started from, the relay counter of stream_scan_kernel's spare waves (an unsigned int, wrapping: zero it with cseq = 0), unused}; a table = the frame_size
 * phases of one block; cseq = the number of MODE 2 launches since that counter was zeroed.  ctab != NULL selects MODE 2: this block's phases from ctab,
 * the next block's into ctab_next (by a spare wave of workgroup 0), cstate advanced; the per-stream mixer[] is then not touched */
int launch_carrier_table(float *cstate, float *tab, int frame_size, bool rest_only, hipStream_t s);   /* whole table, or what stream_scan_kernel left (carrier.h) */
int launch_carrier_broadcast(const float *cstate, float *mixer, int nstreams, hipStream_t s);   /* exactly one of pcm / x (complex blocks: no mixer) */
/* bitstages.hip */
int launch_crc16(const uint8_t *data, int npackets, int nbytes, uint16_t *crc, hipStream_t s);
int launch_interleave(uint8_t *data, int npackets, int nbytes, unsigned b, int dir, hipStream_t s);
int launch_scramble(uint8_t *sym, const uint8_t *keystream, int npackets, int nsym, hipStream_t s);
int launch_pack_dibits(const uint8_t *sym, uint8_t *packed, size_t nrows, int nsym, hipStream_t s);      /* sym 16-byte aligned when nsym % 16 == 0 */
/* txchain.hip */
int tx_history_symbols(void);          /* symbols of state per transmitter (uint8 each, 4 = none yet) */
int launch_tx_shape(const uint8_t *sym, uint8_t *hist, const float *taps, float *sig, int nstreams, int nsym,
                    int cycles, hipStream_t s);   /* shaped baseband of nsym symbols per transmitter; updates hist */
int launch_tx_upmix(const float *sig, int16_t *pcm, float *state, int nstreams, int length, hipStream_t s);
int launch_fill_i32(int32_t *p, int n, int32_t v, hipStream_t s);
int launch_sincos_hash(uint32_t first, uint32_t count, unsigned long long *acc, hipStream_t s);

} // namespace qpsk
#endif
