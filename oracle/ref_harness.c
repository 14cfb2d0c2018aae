/*
 * ref_harness.c -- TEST INFRASTRUCTURE.  Builds the UNMODIFIED reference
 * sources, where they lie under /root/reference, into the oracle/_ref/ libraries so
 * that the oracle restatement (oracle/qpsk_oracle.c) and the golden fixtures
 * (tests/golden/) can be pinned against the reference itself.
 *
 * Nothing from the reference is copied: this translation unit #includes the
 * reference .c files by path (-I/root/reference) and only adds
 *   - the variant parameters: FS / RS / FRAME_SIZE are unguarded #defines in
 *     qpsk.h:16-23; qpsk.h is "#pragma once", so including it first and then
 *     re-#defining the three names makes every later use in qpsk.c pick up
 *     the variant (CYCLES = (int)(FS/RS), qpsk.h:21, follows);
 *   - a pass-through hook on the rrc_fir() call inside rx_frame() (qpsk.c:125)
 *     so that a test can hand rx_frame() an arbitrary COMPLEX input block
 *     (the BASELINE configs start from complex baseband, rx_frame() itself
 *     starts from int16 PCM, qpsk.c:88,114-118);
 *   - plain-pointer accessors for the file-scope state (qpsk.c:36-53) and the
 *     static functions (qpsk.c:24-29).
 * main() (qpsk.c:289) is renamed and never run (it seeds from time(0) and
 * overflows its own frame[] buffer, qpsk.c:294,329 / SURVEY Q13).
 *
 * This file is only compiled in the build container (the reference does not
 * exist on the GPU box); see oracle/Makefile.
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <complex.h>
#include <math.h>

#include "qpsk.h" /* reference, via -I: consumed once (#pragma once) */

#ifndef REF_FS
#error "build with -DREF_FS= -DREF_RS= -DREF_FRAME_SIZE="
#endif
#undef FS
#undef RS
#undef FRAME_SIZE
#define FS REF_FS
#define RS REF_RS
#define FRAME_SIZE REF_FRAME_SIZE

#include "rrc_fir.c"     /* rrc_fir(), rrc_make(), static coeffs[] */
#include "costas_loop.c" /* the 6 loop functions + 16 accessors */

static void ref_hook_rrc_fir(complex float memory[], complex float sample[], int length);
#define rrc_fir ref_hook_rrc_fir
#define main ref_main_unused
#include "qpsk.c" /* rx_frame(), qpsk_demod(), tx path, all globals */
#undef main
#undef rrc_fir

#include "algorithms/fft.c"
#include "algorithms/crc16.c"
#include "algorithms/interleave.c"
#include "algorithms/bit-scramble.c"

/* ---- complex-input injection ------------------------------------------ */
static const float *g_inject; /* FRAME_SIZE interleaved (re,im) or NULL */

static void ref_hook_rrc_fir(complex float memory[], complex float sample[], int length)
{
    if (g_inject && memory == rx_filter) {
        for (int i = 0; i < length; i++)
            sample[i] = CMPLXF(g_inject[2 * i], g_inject[2 * i + 1]);
    }
    rrc_fir(memory, sample, length);
}

/* ---- parameters -------------------------------------------------------- */
void ref_params(double *fs, double *rs, int *frame_size, int *cycles, int *ntaps)
{
    *fs = FS;
    *rs = RS;
    *frame_size = FRAME_SIZE;
    *cycles = CYCLES;
    *ntaps = NTAPS;
}

/* State reset + the initialisation statements of main(), qpsk.c:302,308,316,
 * 320,341,342, with the literals turned into arguments. */
void ref_reset(float loop_bw, float min_freq, float max_freq, float alpha, double tx_hz, double rx_hz)
{
    memset(tx_filter, 0, sizeof tx_filter);
    memset(rx_filter, 0, sizeof rx_filter);
    memset(input_frame, 0, sizeof input_frame);
    memset(decimated_frame, 0, sizeof decimated_frame);
    memset(costas_frame, 0, sizeof costas_frame);
    fbb_offset_freq = 0.0f;
    d_error = 0.0f;
    /* costas_loop.c statics start at zero in a fresh process; set_frequency()
     * inside create_control_loop() compares against d_max/d_min_freq of the
     * PREVIOUS configuration, so clear those first to get fresh-process
     * behaviour (costas_loop.c:31-42, 117-125). */
    d_phase = d_freq = d_max_freq = d_min_freq = d_damping = d_loop_bw = d_alpha = d_beta = 0.0f;
    create_control_loop(loop_bw, min_freq, max_freq);
    rrc_make(FS, RS, alpha);
    fbb_tx_phase = cmplx(0.0f);
    fbb_tx_rect = cmplx(TAU * tx_hz / FS);
    fbb_rx_phase = cmplx(0.0f);
    fbb_rx_rect = cmplxconj(TAU * rx_hz / FS);
    g_inject = NULL;
    scramble_init(both);
}

/* ---- taps and the FIR as a free function ------------------------------- */
void ref_rrc_make(float fs, float rs, float alpha) { rrc_make(fs, rs, alpha); }
void ref_get_taps(float *out) { memcpy(out, coeffs, sizeof coeffs); }
void ref_rrc_fir(float *memory, float *sample, int length)
{
    rrc_fir((complex float *)memory, (complex float *)sample, length);
}

/* ---- receive path ------------------------------------------------------ */
void ref_rx_frame_pcm(const int16_t *in)
{
    int16_t *tmp = malloc(sizeof(int16_t) * FRAME_SIZE);
    memcpy(tmp, in, sizeof(int16_t) * FRAME_SIZE);
    g_inject = NULL;
    rx_frame(tmp);
    free(tmp);
}

void ref_rx_frame_cplx(const float *in)
{
    int16_t *zeros = calloc(FRAME_SIZE, sizeof(int16_t));
    g_inject = in;
    rx_frame(zeros);
    g_inject = NULL;
    free(zeros);
}

void ref_get_input_frame(float *out) { memcpy(out, input_frame, sizeof input_frame); }
void ref_get_rx_filter(float *out) { memcpy(out, rx_filter, sizeof rx_filter); }
void ref_get_decimated(float *out) { memcpy(out, decimated_frame, sizeof(complex float) * 2 * (FRAME_SIZE / CYCLES)); }
void ref_set_decimated(const float *in) { memcpy(decimated_frame, in, sizeof(complex float) * 2 * (FRAME_SIZE / CYCLES)); }
void ref_get_costas(float *out) { memcpy(out, costas_frame, sizeof(complex float) * (FRAME_SIZE / CYCLES)); }
float ref_get_phase(void) { return get_phase(); }
float ref_get_freq(void) { return get_frequency(); }
float ref_get_alpha(void) { return get_alpha(); }
float ref_get_beta(void) { return get_beta(); }
float ref_get_offset_hz(void) { return fbb_offset_freq; }
void ref_get_mixer(float *out)
{
    out[0] = crealf(fbb_rx_phase);
    out[1] = cimagf(fbb_rx_phase);
    out[2] = crealf(fbb_rx_rect);
    out[3] = cimagf(fbb_rx_rect);
}
void ref_set_mixer(const float *in)
{
    fbb_rx_phase = CMPLXF(in[0], in[1]);
    fbb_rx_rect = CMPLXF(in[2], in[3]);
}

/* slicer (static qpsk_demod, qpsk.c:74-79); symbol index = (bits[1]<<1)|bits[0],
 * the inverse of qpsk_mod(), qpsk.c:270 */
void ref_demod(float re, float im, int *bits) { qpsk_demod(CMPLXF(re, im), bits); }
void ref_get_symbols(uint8_t *out)
{
    int bits[2];
    for (int i = 0; i < FRAME_SIZE / CYCLES; i++) {
        qpsk_demod(costas_frame[i], bits);
        out[i] = (uint8_t)((bits[1] << 1) | bits[0]);
    }
}

/* ---- Costas scalar API by plain floats --------------------------------- */
float ref_phase_detector(float re, float im) { return phase_detector(CMPLXF(re, im)); }
void ref_costas_state(float *out)
{
    out[0] = d_phase; out[1] = d_freq; out[2] = d_max_freq; out[3] = d_min_freq;
    out[4] = d_damping; out[5] = d_loop_bw; out[6] = d_alpha; out[7] = d_beta;
}

/* ---- transmit path as stimulus (qpsk.c:225-285); nsym <= 4096 per call -- */
int ref_tx_symbols(int16_t *samples, const int *bits, int nsym)
{
    return qpsk_packet_mod(samples, (int *)bits, nsym);
}
/* complex baseband of the same modulator, before the upmix: zero-stuff + RRC
 * (qpsk.c:232-243), for building complex test frames with reference code */
int ref_tx_baseband(float *out, const int *bits, int nsym)
{
    int dibit[2];
    complex float *sig = malloc(sizeof(complex float) * (size_t)nsym * CYCLES);
    for (int i = 0, s = 0; i < nsym; i++, s += 2) {
        dibit[0] = bits[s + 1] & 0x1;
        dibit[1] = bits[s] & 0x1;
        sig[i * CYCLES] = qpsk_mod(dibit);
        for (int j = 1; j < CYCLES; j++)
            sig[i * CYCLES + j] = 0.0f;
    }
    rrc_fir(tx_filter, sig, nsym * CYCLES);
    memcpy(out, sig, sizeof(complex float) * (size_t)nsym * CYCLES);
    free(sig);
    return nsym * CYCLES;
}

/* ---- algorithms/ ------------------------------------------------------- */
void ref_fftn(const double *in, double *out, int n) { fftn((complex double *)in, (complex double *)out, n); }
void ref_ifftn(const double *in, double *out, int n) { ifftn((complex double *)in, (complex double *)out, n); }
void ref_fft(const double *in, double *out) { fft((complex double *)in, (complex double *)out); }
void ref_ifft(const double *in, double *out) { ifft((complex double *)in, (complex double *)out); }
int ref_nfft(void) { return NFFT; }
uint16_t ref_crc16(const uint8_t *d, int n) { return crc16(d, n); }
void ref_interleave(uint8_t *d, int nbytes, int dir) { interleave(d, nbytes, dir); }
void ref_scramble_init(int sr) { scramble_init((SRegister)sr); }
int ref_scramble(uint8_t *d, int sr) { return scramble(d, (SRegister)sr); }
