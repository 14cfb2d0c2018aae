/*
 * qpsk_dropin.h -- the reference's own C entry points, served by the GPU.
 *
 * A program written against the reference's headers keeps its calls; it
 * includes this header instead of rrc_fir.h / costas_loop.h / algorithms/fft.h
 * and links libqpsk_hip.so instead of rrc_fir.c / costas_loop.c / fft.c.
 * Signatures, argument meaning, in-place behaviour and the process-wide
 * singleton state (one modem per process, not re-entrant) are the
 * reference's.  Each prototype cites the declaration it replaces
 * (file:line in the reference tree).
 *
 * C and C++: the reference's headers carry extern "C" guards (rrc_fir.h:7-9,
 * costas_loop.h:9-12, fft.h:32-35); so does this one.  In C the complex types
 * are <complex.h>'s, as in the reference.  A C++ translation unit gets the
 * pointer-taking functions with the compiler's _Complex types (an extension
 * both g++ and hipcc/clang++ accept; same layout, float[2] / double[2]); the
 * two functions that take a complex float BY VALUE -- phase_detector(),
 * qpsk_demod() -- are declared for C only.
 *
 * The names the reference's caller takes from those headers are here too:
 * NTAPS, GAIN (rrc_fir.h:13-14: qpsk.c:36-37 sizes tx_filter[NTAPS] /
 * rx_filter[NTAPS] with them) and NFFT (fft.h:44).
 *
 * qpsk.c keeps rx_frame() and qpsk_demod() file-static (qpsk.c:24-25).  A
 * caller that keeps those definitions and wants only the primitives from the
 * GPU defines QPSK_DROPIN_PRIMITIVES_ONLY before the include: the two names
 * are then not declared here (patch A).
 *
 * What the reference fixed with #defines (FS, RS, CENTER, FRAME_SIZE,
 * qpsk.h:16-23) is set once with qpsk_dropin_configure(); without that call
 * the shipped values apply (9600, 2400, 1500, 512).
 *
 * Failure: the reference's functions return void and cannot fail.  Here a
 * missing GPU or a HIP error is fatal: the message goes to stderr and the
 * process aborts.  Nothing is ever computed on the CPU instead.
 */
#ifndef QPSK_DROPIN_H
#define QPSK_DROPIN_H

#include <stdint.h>
#include "qpsk_hip.h"

#ifdef __cplusplus
typedef float _Complex qpsk_dropin_cfloat;
typedef double _Complex qpsk_dropin_cdouble;
extern "C" {
#else
#include <complex.h>
typedef complex float qpsk_dropin_cfloat;
typedef complex double qpsk_dropin_cdouble;
#endif

/* ---- the constants of rrc_fir.h and fft.h ------------------------------ */
#ifndef NTAPS
#define NTAPS 127                                                          /* rrc_fir.h:13 */
#endif
#ifndef GAIN
#define GAIN 1.85                                                          /* rrc_fir.h:14 */
#endif
#ifndef NFFT
#define NFFT 512                                                           /* fft.h:44 */
#endif
/* a caller that had its own NTAPS / GAIN / NFFT before this header is accepted only if they are the reference's: rrc_fir()
 * reads and writes memory[127] whatever the caller thinks NTAPS is, fft()/ifft() transform 512 points */
#if NTAPS != 127 || NFFT != 512
#error "qpsk_dropin.h: NTAPS / NFFT are defined with other values than the reference's (rrc_fir.h:13: 127, fft.h:44: 512); the library's rrc_fir() and fft() use those"
#endif
/* GAIN is a floating constant: the preprocessor cannot compare it and C11 has no constant expression for it; C++ checks it */
#ifdef __cplusplus
#define QPSK_DROPIN_CHECK_GAIN static_assert(GAIN == 1.85, "qpsk_dropin.h: GAIN is defined with another value than the reference's (rrc_fir.h:14)")
QPSK_DROPIN_CHECK_GAIN;
#undef QPSK_DROPIN_CHECK_GAIN
#endif

/* ---- configuration that the reference hard-codes ---------------------- */
/* p->fs, rs, frame_size replace FS, RS, FRAME_SIZE (qpsk.h:16-23); center_hz replaces CENTER
 * (qpsk.h:18) in fbb_rx_rect = cmplxconj(TAU * CENTER / FS) (qpsk.c:342).  Resets all modem state.
 * Does NOT build taps or the loop: call rrc_make() and create_control_loop() as main() does
 * (qpsk.c:302,308). */
int qpsk_dropin_configure(const qpsk_params *p, double center_hz);
/* device to use (default: current HIP device); call before anything else */
int qpsk_dropin_set_device(int device);
void qpsk_dropin_shutdown(void);

/* ---- rrc_fir.h --------------------------------------------------------- */
void rrc_fir(qpsk_dropin_cfloat memory[], qpsk_dropin_cfloat sample[], int length); /* rrc_fir.h:16 */
void rrc_make(float fs, float rs, float alpha);                           /* rrc_fir.h:17 */

/* ---- costas_loop.h ----------------------------------------------------- */
void create_control_loop(float loop_bw, float min_freq, float max_freq);  /* costas_loop.h:16 */
#ifndef __cplusplus
float phase_detector(complex float sample);                               /* costas_loop.h:17 */
#endif
void update_gains(void);                                                  /* costas_loop.h:18 */
void advance_loop(float error);                                           /* costas_loop.h:19 */
void phase_wrap(void);                                                    /* costas_loop.h:20 */
void frequency_limit(void);                                               /* costas_loop.h:21 */
void set_loop_bandwidth(float);                                           /* costas_loop.h:25 */
void set_damping_factor(float);                                           /* costas_loop.h:26 */
void set_alpha(float);                                                    /* costas_loop.h:27 */
void set_beta(float);                                                     /* costas_loop.h:28 */
void set_frequency(float);                                                /* costas_loop.h:29 */
void set_phase(float);                                                    /* costas_loop.h:30 */
void set_max_freq(float);                                                 /* costas_loop.h:31 */
void set_min_freq(float);                                                 /* costas_loop.h:32 */
float get_loop_bandwidth(void);                                           /* costas_loop.h:36 */
float get_damping_factor(void);                                           /* costas_loop.h:37 */
float get_alpha(void);                                                    /* costas_loop.h:38 */
float get_beta(void);                                                     /* costas_loop.h:39 */
float get_frequency(void);                                                /* costas_loop.h:40 */
float get_phase(void);                                                    /* costas_loop.h:41 */
float get_max_freq(void);                                                 /* costas_loop.h:42 */
float get_min_freq(void);                                                 /* costas_loop.h:43 */

/* ---- algorithms/fft.h -------------------------------------------------- */
#define QPSK_NFFT NFFT
void fft(qpsk_dropin_cdouble *in, qpsk_dropin_cdouble *out);              /* fft.h:46 */
void fftn(qpsk_dropin_cdouble *in, qpsk_dropin_cdouble *out, int n);      /* fft.h:47 */
void ifft(qpsk_dropin_cdouble *in, qpsk_dropin_cdouble *out);             /* fft.h:48 */
void ifftn(qpsk_dropin_cdouble *in, qpsk_dropin_cdouble *out, int n);     /* fft.h:49 */

/* ---- qpsk.c (file-static there, exported equivalents here) ------------- */
#ifndef QPSK_DROPIN_PRIMITIVES_ONLY
#ifndef __cplusplus
void qpsk_demod(complex float symbol, int bits[]);                        /* qpsk.c:24,74-79 */
#endif
void rx_frame(int16_t in[]);                                              /* qpsk.c:25,88-218 */
#endif

/* what rx_frame() leaves in the reference's globals (qpsk.c:41,51) */
const qpsk_dropin_cfloat *qpsk_dropin_costas_frame(void);   /* costas_frame[FRAME_SIZE/CYCLES] */
const uint8_t *qpsk_dropin_symbols(void);              /* (bits[1]<<1)|bits[0] per symbol, qpsk.c:209 */
float qpsk_dropin_offset_freq(void);                   /* fbb_offset_freq, qpsk.c:217 */
int qpsk_dropin_timing_index(void);                    /* index, qpsk.c:105,173-180 */

#ifdef __cplusplus
}
#endif

#endif /* QPSK_DROPIN_H */
