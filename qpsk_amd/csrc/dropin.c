/*
 * dropin.c -- the reference's function names as process-wide singletons on
 * top of the batched GPU API (include/qpsk_dropin.h).  Plain C11 so that the
 * by-value "complex float" parameters have the reference's ABI.
 *
 * The per-sample work of rrc_fir(), rx_frame() and fft*() runs in
 * kernels.hip.  The scalar loop API (create_control_loop ... get_min_freq,
 * costas_loop.c:31-154) is the loop's control surface: it edits the one
 * (phase, freq, gains, limits) record that rx_frame() hands to the GPU
 * before each block and reads back after it, so a host program sees the same
 * state transitions the reference's file statics go through.
 */
#include "../../include/qpsk_dropin.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define TAU_D (2.0 * 3.14159265358979323846) /* qpsk.h:29 */

/* internal extras of api.cpp */
int qpsk_streams_set_loop_state(qpsk_ctx *ctx, const float *h_state);
int qpsk_streams_get_loop_state(qpsk_ctx *ctx, float *h_state);

static struct {
    qpsk_ctx *ctx;
    int device;
    qpsk_params prm;
    double center_hz;
    int configured;
    int have_taps;
    float taps[QPSK_NTAPS];
    /* costas_loop.c:13-23 */
    float phase, freq, max_freq, min_freq, damping, loop_bw, alpha, beta;
    /* streaming buffers for rx_frame() */
    int streams_ready;
    void *d_pcm, *d_sym, *d_costas, *d_index;
    complex float *costas_frame;
    uint8_t *symbols;
    int index;
    float offset_freq;
    /* staging for rrc_fir()/fft*() */
    void *d_a, *d_b, *d_m;
    size_t cap_a, cap_b;
} S = {.device = -1};

static void die(const char *what)
{
    fprintf(stderr, "qpsk_hip drop-in: %s failed: %s\n", what, qpsk_last_error());
    abort();
}
#define MUST(call) do { if ((call) != QPSK_OK) die(#call); } while (0)

static void free_buffers(void)
{
    if (!S.ctx) return;
    qpsk_dev_free(S.ctx, S.d_pcm); qpsk_dev_free(S.ctx, S.d_sym); qpsk_dev_free(S.ctx, S.d_costas);
    qpsk_dev_free(S.ctx, S.d_index); qpsk_dev_free(S.ctx, S.d_a); qpsk_dev_free(S.ctx, S.d_b);
    qpsk_dev_free(S.ctx, S.d_m);
    S.d_pcm = S.d_sym = S.d_costas = S.d_index = S.d_a = S.d_b = S.d_m = NULL;
    S.cap_a = S.cap_b = 0;
    free(S.costas_frame); free(S.symbols);
    S.costas_frame = NULL; S.symbols = NULL;
    S.streams_ready = 0;
}

static void ensure_ctx(void)
{
    if (S.ctx) return;
    if (!S.configured) {
        qpsk_params_default(&S.prm);
        S.center_hz = 1500.0; /* CENTER, qpsk.h:18 */
        S.configured = 1;
    }
    MUST(qpsk_ctx_create(&S.ctx, S.device, &S.prm, NULL));
    if (S.have_taps)
        MUST(qpsk_ctx_set_taps(S.ctx, S.taps));
}

int qpsk_dropin_set_device(int device)
{
    if (S.ctx) { free_buffers(); qpsk_ctx_destroy(S.ctx); S.ctx = NULL; }
    S.device = device;
    return QPSK_OK;
}

int qpsk_dropin_configure(const qpsk_params *p, double center_hz)
{
    if (!p) return QPSK_ERR_ARG;
    if (S.ctx) { free_buffers(); qpsk_ctx_destroy(S.ctx); S.ctx = NULL; }
    S.prm = *p;
    S.center_hz = center_hz;
    S.configured = 1;
    S.have_taps = 0;
    S.index = 0;
    S.offset_freq = 0.0f;
    ensure_ctx();
    return QPSK_OK;
}

void qpsk_dropin_shutdown(void)
{
    if (S.ctx) { free_buffers(); qpsk_ctx_destroy(S.ctx); S.ctx = NULL; }
}

static void grow(void **p, size_t *cap, size_t bytes)
{
    if (*cap >= bytes) return;
    if (*p) MUST(qpsk_dev_free(S.ctx, *p));
    MUST(qpsk_dev_alloc(S.ctx, p, bytes));
    *cap = bytes;
}

/* ------------------------------------------------------------ rrc_fir.h */
void rrc_make(float fs, float rs, float alpha)
{
    /* host arithmetic as in the reference (rrc_fir.c:32-76), then uploaded */
    extern void qpsk_host_rrc_taps(float, float, float, float *);
    qpsk_host_rrc_taps(fs, rs, alpha, S.taps);
    S.have_taps = 1;
    ensure_ctx();
    MUST(qpsk_ctx_set_taps(S.ctx, S.taps));
}

void rrc_fir(complex float memory[], complex float sample[], int length)
{
    if (length <= 0) return;
    ensure_ctx();
    const size_t bytes = sizeof(complex float) * (size_t)length;
    grow(&S.d_a, &S.cap_a, bytes);
    grow(&S.d_b, &S.cap_b, bytes);
    if (!S.d_m) MUST(qpsk_dev_alloc(S.ctx, &S.d_m, sizeof(complex float) * QPSK_NTAPS));
    MUST(qpsk_dev_upload(S.ctx, S.d_a, sample, bytes));
    MUST(qpsk_dev_upload(S.ctx, S.d_m, memory, sizeof(complex float) * QPSK_NTAPS));
    MUST(qpsk_rrc_fir_batch(S.ctx, (float *)S.d_m, (const float *)S.d_a, (float *)S.d_b, 1, length));
    MUST(qpsk_dev_download(S.ctx, sample, S.d_b, bytes));
    MUST(qpsk_dev_download(S.ctx, memory, S.d_m, sizeof(complex float) * QPSK_NTAPS));
}

/* -------------------------------------------------------- costas_loop.h */
void update_gains(void)
{
    extern void qpsk_host_loop_gains(float, float, float *, float *);
    qpsk_host_loop_gains(S.damping, S.loop_bw, &S.alpha, &S.beta);
}

void phase_wrap(void)
{
    while (S.phase > TAU_D) S.phase -= TAU_D;
    while (S.phase < -TAU_D) S.phase += TAU_D;
}

void frequency_limit(void)
{
    if (S.freq > S.max_freq) S.freq = S.max_freq;
    else if (S.freq < S.min_freq) S.freq = S.min_freq;
}

float phase_detector(complex float sample)
{
    const float re = crealf(sample), im = cimagf(sample);
    return (re > 0.0f ? 1.0f : -1.0f) * im - (im > 0.0f ? 1.0f : -1.0f) * re;
}

void advance_loop(float error)
{
    S.freq = S.freq + S.beta * error;
    S.phase = S.phase + S.freq + S.alpha * error;
}

/* the range checks of the reference setters have no effect (the value is stored
 * unconditionally afterwards, costas_loop.c:79-115); kept that way on purpose */
void set_loop_bandwidth(float bw) { S.loop_bw = bw; update_gains(); }
void set_damping_factor(float df) { S.damping = df; update_gains(); }
void set_alpha(float a) { S.alpha = a; }
void set_beta(float b) { S.beta = b; }
void set_frequency(float f)
{
    if (f > S.max_freq) S.freq = S.max_freq;
    else if (f < S.min_freq) S.freq = S.min_freq;
    else S.freq = f;
}
void set_phase(float p) { S.phase = p; phase_wrap(); }
void set_max_freq(float f) { S.max_freq = f; }
void set_min_freq(float f) { S.min_freq = f; }
float get_loop_bandwidth(void) { return S.loop_bw; }
float get_damping_factor(void) { return S.damping; }
float get_alpha(void) { return S.alpha; }
float get_beta(void) { return S.beta; }
float get_frequency(void) { return S.freq; }
float get_phase(void) { return S.phase; }
float get_max_freq(void) { return S.max_freq; }
float get_min_freq(void) { return S.min_freq; }

void create_control_loop(float loop_bw, float min_freq, float max_freq)
{
    set_phase(0.0f);
    set_frequency(0.0f);
    set_max_freq(max_freq);
    set_min_freq(min_freq);
    set_damping_factor(sqrtf(2.0f) / 2.0f);
    set_loop_bandwidth(loop_bw);
}

/* ------------------------------------------------------ algorithms/fft.h */
static void fft_any(complex double *in, complex double *out, int n, int inverse)
{
    ensure_ctx();
    const size_t bytes = sizeof(complex double) * (size_t)n;
    grow(&S.d_a, &S.cap_a, bytes);
    grow(&S.d_b, &S.cap_b, bytes);
    MUST(qpsk_dev_upload(S.ctx, S.d_a, in, bytes));
    MUST(qpsk_fft_batch(S.ctx, (const double *)S.d_a, (double *)S.d_b, 1, n, inverse));
    MUST(qpsk_dev_download(S.ctx, out, S.d_b, bytes));
}
void fft(complex double *in, complex double *out) { fft_any(in, out, QPSK_NFFT, 0); }
void fftn(complex double *in, complex double *out, int n) { fft_any(in, out, n, 0); }
void ifft(complex double *in, complex double *out) { fft_any(in, out, QPSK_NFFT, 1); }
void ifftn(complex double *in, complex double *out, int n) { fft_any(in, out, n, 1); }

/* ---------------------------------------------------------------- qpsk.c */
void qpsk_demod(complex float symbol, int bits[])
{
    /* one symbol: symbol * cmplx(ROTATE45), both parts 0x1.6a09e6p-1 (qpsk.c:75) */
    const float k = 0x1.6a09e6p-1f, re = crealf(symbol), im = cimagf(symbol);
    bits[0] = (re * k - im * k) < 0.0f;
    bits[1] = (re * k + im * k) < 0.0f;
}

void rx_frame(int16_t in[])
{
    ensure_ctx();
    const int L = S.prm.frame_size, N = qpsk_ctx_nsym(S.ctx);
    if (!S.streams_ready) {
        MUST(qpsk_streams_reset(S.ctx, 1, S.center_hz)); /* qpsk.c:341-342 */
        S.costas_frame = calloc((size_t)N, sizeof(complex float));
        S.symbols = calloc((size_t)N, 1);
        S.streams_ready = 1;
    }
    (void)L;
    /* the loop's control surface (costas_loop.h setters) may have changed gains, limits or state since the last
     * block: limits are kernel arguments, gains go up only when they changed, the state travels with the block.
     * One upload, one download, one synchronisation per block (qpsk_streams_rx_pcm_host). */
    float st[2] = {S.phase, S.freq};
    int32_t idx = 0;
    MUST(qpsk_ctx_set_loop(S.ctx, S.alpha, S.beta, S.min_freq, S.max_freq));
    MUST(qpsk_streams_rx_pcm_host(S.ctx, in, st, S.symbols, (float *)S.costas_frame, &idx));
    S.phase = st[0];
    S.freq = st[1];
    S.index = idx;
    S.offset_freq = (float)((double)S.freq * S.prm.rs / TAU_D); /* qpsk.c:217 */
}

const complex float *qpsk_dropin_costas_frame(void) { return S.costas_frame; }
const uint8_t *qpsk_dropin_symbols(void) { return S.symbols; }
float qpsk_dropin_offset_freq(void) { return S.offset_freq; }
int qpsk_dropin_timing_index(void) { return S.index; }
