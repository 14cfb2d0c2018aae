/*
 * qpsk_oracle.h -- TEST INFRASTRUCTURE: CPU restatement of the reference
 * receive path (MonsieurETM/QPSK).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product (qpsk_amd/, include/)
 * never does.
 *
 * Parity status: PINNED.  Every function below is compared bit for bit with
 * the reference itself, compiled untouched into oracle/_ref/ (oracle/Makefile,
 * oracle/ref_harness.c), by tests/test_oracle_vs_ref.py in the build
 * container, and with the committed fixtures under tests/golden/ (generated
 * from the reference by tools/make_golden.py) everywhere else.
 *
 * All complex data are interleaved float pairs (re, im) == C "complex float".
 * Citations are file:line in /root/reference.
 */
#ifndef QPSK_ORACLE_H
#define QPSK_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QO_NTAPS 127            /* rrc_fir.h:13 */
#define QO_GAIN 1.85            /* rrc_fir.h:14 (double) */
#define QO_PI 3.14159265358979323846 /* qpsk.h:25-27 / math.h M_PI */
#define QO_TAU (2.0 * QO_PI)    /* qpsk.h:29 */
#define QO_ROT45 0x1.6a09e6p-1f /* cosf == sinf of (float)(M_PI/4), qpsk.h:30, qpsk.c:75 */

enum { QO_TIMING_HIST = 0, QO_TIMING_FIXED = 1, QO_TIMING_FFT = 2 };

/* ---- RRC taps and FIR (rrc_fir.c) ---- */
void qo_rrc_make(float fs, float rs, float alpha, float *taps);                       /* rrc_fir.c:32-76 */
void qo_rrc_fir(const float *taps, float *memory, float *sample, int length);         /* rrc_fir.c:17-30 */

/* ---- Costas loop (costas_loop.c), state in a struct instead of file statics ---- */
typedef struct {
    float phase, freq;         /* costas_loop.c:13-14 */
    float max_freq, min_freq;  /* :16-17 */
    float damping, loop_bw;    /* :19-20 */
    float alpha, beta;         /* :22-23 */
} qo_costas;

void qo_costas_create(qo_costas *c, float loop_bw, float min_freq, float max_freq);   /* :31-42 */
float qo_phase_detector(float re, float im);                                          /* :44-47 */
void qo_update_gains(qo_costas *c);                                                   /* :49-54 */
void qo_advance_loop(qo_costas *c, float error);                                      /* :56-59 */
void qo_phase_wrap(qo_costas *c);                                                     /* :61-67 */
void qo_frequency_limit(qo_costas *c);                                                /* :69-74 */
void qo_set_loop_bandwidth(qo_costas *c, float bw);                                   /* :79-87 */
void qo_set_damping_factor(qo_costas *c, float df);                                   /* :89-97 */
void qo_set_alpha(qo_costas *c, float a);                                             /* :99-106 */
void qo_set_beta(qo_costas *c, float b);                                              /* :108-115 */
void qo_set_frequency(qo_costas *c, float f);                                         /* :117-125 */
void qo_set_phase(qo_costas *c, float p);                                             /* :127-132 */

/* one Costas step on one symbol (qpsk.c:197-209): writes the de-rotated
 * symbol, returns the slicer decision (bits[1]<<1)|bits[0] */
int qo_costas_step(qo_costas *c, float d_re, float d_im, float *z_re, float *z_im);
int qo_demod(float re, float im);                                                     /* qpsk.c:74-79 */

/* ---- timing histogram (qpsk.c:90-108,127-180) on one filtered block ---- */
int qo_timing_index(const float *filtered, int frame_size, int cycles);
int qo_timing_hist(const float *filtered, int frame_size, int cycles, int hist[8]); /* + hist_i[k]+hist_q[k], qpsk.c:175 */

/* ---- FFT timing estimate: NOT in the reference (parity unpinned by it); definition in
 * qpsk_amd/csrc/timing_fft.hip, restated here for the parity tests. x = one fresh frame BEFORE the FIR ---- */
int qo_timing_fft_index(const float *taps, const float *x, int L, int cycles);

/* ---- one modem instance: everything rx_frame() keeps in globals ---- */
typedef struct {
    double fs, rs;             /* qpsk.h:16-17 */
    int cycles;                /* qpsk.h:21: (int)(FS/RS) */
    int frame_size;            /* qpsk.h:23 */
    int nsym;                  /* frame_size / cycles */
    int timing_mode, fixed_index;
    float taps[QO_NTAPS];      /* rrc_fir.c:12 */
    qo_costas loop;
    float rx_filter[2 * QO_NTAPS]; /* qpsk.c:37 */
    float mix_phase[2], mix_rect[2]; /* qpsk.c:48-49 */
    float offset_hz;           /* qpsk.c:51,217 */
    int last_index;            /* qpsk.c:105 (local there) */
    float *input_frame;        /* qpsk.c:39, frame_size complex */
    float *decimated;          /* qpsk.c:40, 2*nsym complex */
    float *costas_frame;       /* qpsk.c:41, nsym complex */
    uint8_t *symbols;          /* slicer decisions of the last call, nsym */
} qo_modem;

qo_modem *qo_modem_new(double fs, double rs, int frame_size, float rrc_alpha, float loop_bw,
                       float min_freq, float max_freq, int timing_mode, int fixed_index);
void qo_modem_free(qo_modem *m);
void qo_modem_reset(qo_modem *m); /* fresh-process state, taps and loop gains kept */
void qo_modem_set_mixer(qo_modem *m, const float *phase_rect4);
void qo_mixer_from_hz(double hz, double fs, float *rect2); /* qpsk.c:342: cmplxconj(TAU*hz/FS) */
void qo_rx_frame_pcm(qo_modem *m, const int16_t *in);  /* qpsk.c:88-218 */
void qo_rx_frame_cplx(qo_modem *m, const float *in);   /* same from qpsk.c:125 on */

/* ---- batch of INDEPENDENT frames: per frame "fresh modem; rx_frame(frame);
 * rx_frame(zeros)", results of the second call.
 * Any of sym/freq/phase/costas/index/hz may be NULL. threads<=0: all cores. */
void qo_rx_batch(double fs, double rs, int frame_size, float rrc_alpha, float loop_bw,
                 float min_freq, float max_freq, int timing_mode, int fixed_index,
                 const float *in, int nframes, uint8_t *sym, float *freq, float *phase,
                 float *costas, int32_t *index, float *hz, int threads);

/* the same when several loop bandwidths share one FIR pass (config 5):
 * outputs are [nframes][nbw][...] */
void qo_rx_batch_bw(double fs, double rs, int frame_size, float rrc_alpha, const float *loop_bws,
                    int nbw, float min_freq, float max_freq, int timing_mode, int fixed_index,
                    const float *in, int nframes, uint8_t *sym, float *freq, float *phase,
                    int32_t *index, int threads);

/* ---- transmit side used as stimulus (qpsk.c:225-285) ---- */
typedef struct {
    float taps[QO_NTAPS];
    float tx_filter[2 * QO_NTAPS];
    float phase[2], rect[2];
    int cycles;
} qo_tx;
void qo_tx_init(qo_tx *t, double fs, double rs, float rrc_alpha, double tx_hz);
int qo_tx_symbols(qo_tx *t, int16_t *samples, const int *bits, int nsym); /* qpsk.c:273-285,225-264 */

/* ---- algorithms/fft.c ---- */
void qo_fftn(const double *in, double *out, int n);   /* fft.c:110-120 */
void qo_ifftn(const double *in, double *out, int n);  /* fft.c:130-136 */

/* ---- algorithms/ bit-level stages ---- */
uint16_t qo_crc16(const uint8_t *data, int length);          /* crc16.c:11-23 */
void qo_interleave(uint8_t *inout, int nbytes, int dir);     /* interleave.c:33-78 */
void qo_scramble_init(uint16_t *mem);                        /* bit-scramble.c:46-55 */
void qo_scramble(uint8_t *sym, uint16_t *mem);               /* bit-scramble.c:57-69 */

/* single-precision sin/cos as the reference's libm computes them */
void qo_sincosf(float x, float *s, float *c);
float qo_cabsf(float re, float im);

#ifdef __cplusplus
}
#endif
#endif
