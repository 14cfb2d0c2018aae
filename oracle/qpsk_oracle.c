/*
 * qpsk_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of the reference
 * receive path, see qpsk_oracle.h for the rules on who may use it and for the
 * parity status (pinned against oracle/_ref and tests/golden).
 *
 * The reference works on C99 "complex float" values; here every complex
 * operation is written out on the two float components in the order gcc's
 * C front end lowers it (real x complex is component-wise, complex x complex
 * is (ac - bd, ad + bc), no FMA contraction because the reference Makefile
 * builds -std=c11, Makefile:7).  Build this file -ffp-contract=off.
 */
#include "qpsk_oracle.h"
#include "oracle_sincosf.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void qo_sincosf(float x, float *s, float *c) { oracle_sincosf(x, s, c); }

/* cabsf() == glibc hypotf (sysdeps/ieee754/flt-32/e_hypotf.c): exact squares
 * in double, one rounded add, correctly rounded double sqrt, narrowed. */
float qo_cabsf(float re, float im)
{
    return (float)sqrt((double)re * (double)re + (double)im * (double)im);
}

/* ------------------------------------------------------------------ taps */

/* rrc_fir.c:32-76.  Types decide the rounding: every name below is float,
 * QO_PI and QO_GAIN are double, so each sub-expression is evaluated in the
 * type C's usual arithmetic conversions give it. */
void qo_rrc_make(float fs, float rs, float alpha, float *taps)
{
    const float spb = fs / rs;
    float scale = 0.f;
    const int mid = QO_NTAPS / 2;

    for (int i = 0; i < QO_NTAPS; i++) {
        const float xi = (float)(i - mid);
        const float x1 = (float)(QO_PI * (double)xi / (double)spb);
        float x2 = 4.f * alpha * xi / spb;
        float x3 = x2 * x2 - 1.f;
        float num, den;

        if (fabsf(x3) >= 0.000001f) {
            const float cs = oracle_cosf((1.f + alpha) * x1);
            if (i != mid) {
                const float sn = oracle_sinf((1.f - alpha) * x1);
                num = cs + sn / (4.f * alpha * xi / spb);
            } else {
                num = (float)((double)cs + (double)(1.f - alpha) * QO_PI / (double)(4.f * alpha));
            }
            den = (float)((double)x3 * QO_PI);
        } else {
            if (alpha == 1.f) {
                taps[i] = -1.f;
                scale += taps[i];
                continue;
            }
            x3 = (1.f - alpha) * x1;
            x2 = (1.f + alpha) * x1;
            const float s2 = oracle_sinf(x2), c3 = oracle_cosf(x3), s3 = oracle_sinf(x3);
            const double t1 = (double)(s2 * (1.f + alpha)) * QO_PI;
            const double t2 = (double)c3 * ((double)(1.f - alpha) * QO_PI * (double)spb) / (double)(4.f * alpha * xi);
            const double t3 = (double)(s3 * spb * spb / (4.f * alpha * xi * xi));
            num = (float)(t1 - t2 + t3);
            den = (float)((double)-32.f * QO_PI * (double)alpha * (double)alpha * (double)xi / (double)spb);
        }
        taps[i] = 4.f * alpha * num / den;
        scale += taps[i];
    }
    for (int i = 0; i < QO_NTAPS; i++)
        taps[i] = (float)(((double)taps[i] * QO_GAIN) / (double)scale);
}

/* ------------------------------------------------------------------- FIR */

/* one output of rrc_fir.c:22-28 from a window w[0..126] of (re,im) pairs,
 * w[126] being the newest sample */
static inline void fir_point(const float *taps, const float *w, float *o_re, float *o_im)
{
    float yr = 0.0f, yi = 0.0f;
    for (int i = 0; i < QO_NTAPS; i++) {
        yr = yr + w[2 * i] * taps[i];
        yi = yi + w[2 * i + 1] * taps[i];
    }
    *o_re = (float)((double)yr * QO_GAIN);
    *o_im = (float)((double)yi * QO_GAIN);
}

/* rrc_fir.c:17-30, in place on sample[], delay line in memory[127] */
void qo_rrc_fir(const float *taps, float *memory, float *sample, int length)
{
    if (length <= 0)
        return;
    const int H = QO_NTAPS - 1;
    float *ext = malloc(sizeof(float) * 2 * (size_t)(H + length));
    memcpy(ext, memory + 2, sizeof(float) * 2 * H); /* memory[0] is shifted out by the first step */
    memcpy(ext + 2 * H, sample, sizeof(float) * 2 * (size_t)length);
    for (int j = 0; j < length; j++)
        fir_point(taps, ext + 2 * j, &sample[2 * j], &sample[2 * j + 1]);
    /* the delay line now holds the last 127 samples that went through it */
    if (length >= QO_NTAPS) {
        memcpy(memory, ext + 2 * (H + length - QO_NTAPS), sizeof(float) * 2 * QO_NTAPS);
    } else {
        memmove(memory, memory + 2 * length, sizeof(float) * 2 * (size_t)(QO_NTAPS - length));
        memcpy(memory + 2 * (QO_NTAPS - length), ext + 2 * H, sizeof(float) * 2 * (size_t)length);
    }
    free(ext);
}

/* ---------------------------------------------------------------- Costas */

void qo_update_gains(qo_costas *c)
{
    const float denom = (1.0f + (2.0f * c->damping * c->loop_bw)) + (c->loop_bw * c->loop_bw);
    c->alpha = (4.0f * c->damping * c->loop_bw) / denom;
    c->beta = (4.0f * c->loop_bw * c->loop_bw) / denom;
}

void qo_phase_wrap(qo_costas *c)
{
    /* float state compared with, and stepped by, the DOUBLE 2*pi */
    while ((double)c->phase > QO_TAU)
        c->phase = (float)((double)c->phase - QO_TAU);
    while ((double)c->phase < -QO_TAU)
        c->phase = (float)((double)c->phase + QO_TAU);
}

void qo_frequency_limit(qo_costas *c)
{
    if (c->freq > c->max_freq)
        c->freq = c->max_freq;
    else if (c->freq < c->min_freq)
        c->freq = c->min_freq;
}

/* the "validation" branches of the reference setters are dead: the value is
 * stored unconditionally afterwards (costas_loop.c:79-115) */
void qo_set_loop_bandwidth(qo_costas *c, float bw) { c->loop_bw = bw; qo_update_gains(c); }
void qo_set_damping_factor(qo_costas *c, float df) { c->damping = df; qo_update_gains(c); }
void qo_set_alpha(qo_costas *c, float a) { c->alpha = a; }
void qo_set_beta(qo_costas *c, float b) { c->beta = b; }
void qo_set_frequency(qo_costas *c, float f)
{
    if (f > c->max_freq)
        c->freq = c->max_freq;
    else if (f < c->min_freq)
        c->freq = c->min_freq;
    else
        c->freq = f;
}
void qo_set_phase(qo_costas *c, float p) { c->phase = p; qo_phase_wrap(c); }

void qo_costas_create(qo_costas *c, float loop_bw, float min_freq, float max_freq)
{
    memset(c, 0, sizeof *c); /* file statics of a fresh process */
    qo_set_phase(c, 0.0f);
    qo_set_frequency(c, 0.0f);
    c->max_freq = max_freq;
    c->min_freq = min_freq;
    qo_set_damping_factor(c, sqrtf(2.0f) / 2.0f);
    qo_set_loop_bandwidth(c, loop_bw);
}

float qo_phase_detector(float re, float im)
{
    return (re > 0.0f ? 1.0f : -1.0f) * im - (im > 0.0f ? 1.0f : -1.0f) * re;
}

void qo_advance_loop(qo_costas *c, float error)
{
    c->freq = c->freq + c->beta * error;
    c->phase = c->phase + c->freq + c->alpha * error;
}

int qo_demod(float re, float im)
{
    /* symbol * (cos45 + j sin45), both constants QO_ROT45 */
    const float rr = re * QO_ROT45 - im * QO_ROT45;
    const float ri = re * QO_ROT45 + im * QO_ROT45;
    return ((ri < 0.0f) << 1) | (rr < 0.0f);
}

int qo_costas_step(qo_costas *c, float d_re, float d_im, float *z_re, float *z_im)
{
    float sn, cs;
    oracle_sincosf(c->phase, &sn, &cs);
    /* d * (cos - j sin): w = (cs, -sn); (a+jb)(c+jd) = (ac - bd) + j(ad + bc) */
    const float wr = cs + sn * -0.0f, wi = sn * -1.0f;
    const float zr = d_re * wr - d_im * wi;
    const float zi = d_re * wi + d_im * wr;
    *z_re = zr;
    *z_im = zi;
    qo_advance_loop(c, qo_phase_detector(zr, zi));
    qo_phase_wrap(c);
    qo_frequency_limit(c);
    return qo_demod(zr, zi);
}

/* ---------------------------------------------------------------- timing */

int qo_timing_index(const float *x, int frame_size, int cycles)
{
    int hist[8];
    return qo_timing_hist(x, frame_size, cycles, hist);
}

/* the same, also handing out hist_i[k] + hist_q[k] (qpsk.c:175), which the reference only keeps in locals */
int qo_timing_hist(const float *x, int frame_size, int cycles, int hist[8])
{
    float max_i = 0.0f, max_q = 0.0f, av_i = 0.0f, av_q = 0.0f;
    int hist_i[8] = {0}, hist_q[8] = {0};

    for (int i = 0; i + cycles <= frame_size; i += cycles) {
        for (int j = 0; j < cycles; j++) {
            av_i += fabsf(x[2 * (i + j)]);
            av_q += fabsf(x[2 * (i + j) + 1]);
        }
        av_i /= (float)cycles; /* never reset between symbols (Q2) */
        av_q /= (float)cycles;
        if (av_i > max_i) max_i = av_i;
        if (av_q > max_q) max_q = av_q;
        const float hv_i = max_i / 8.0f, hv_q = max_q / 8.0f;
        for (int k = 1; k < 8; k++)
            if (av_i <= hv_i * (float)k) { hist_i[k]++; break; }
        for (int k = 1; k < 8; k++)
            if (av_q <= hv_q * (float)k) { hist_q[k]++; break; }
    }
    int hmax = 0, index = 0;
    for (int k = 0; k < 8; k++) {
        const int h = hist_i[k] + hist_q[k];
        hist[k] = h;
        if (h > hmax) { hmax = h; index = k; }
    }
    return index;
}

/* FFT timing estimate (NEW DESIGN, no reference counterpart -- parity unpinned by the reference; the
 * transform underneath is the pinned qo_fftn()).  Definition: qpsk_amd/csrc/timing_fft.hip.
 * x = one fresh frame (before the FIR), L samples. */
int qo_timing_fft_index(const float *taps, const float *x, int L, int cycles)
{
    enum { N0 = 128, NFFT = 512 };
    float w[2 * QO_NTAPS];
    double *p = calloc(2 * NFFT, sizeof(double)), *X = calloc(2 * NFFT, sizeof(double));
    for (int m = 0; m < NFFT; m++) {
        const int n = N0 + m;
        /* window of the fresh delay line at sample n: x[n-126 .. n], zeros past the end of the frame */
        for (int k = 0; k < QO_NTAPS; k++) {
            const int s = n - (QO_NTAPS - 1) + k;
            w[2 * k] = (s >= 0 && s < L) ? x[2 * s] : 0.0f;
            w[2 * k + 1] = (s >= 0 && s < L) ? x[2 * s + 1] : 0.0f;
        }
        float yr, yi;
        fir_point(taps, w, &yr, &yi);
        const double pr = (double)yr * (double)yr, pi = (double)yi * (double)yi;
        p[2 * m] = pr + pi;
    }
    qo_fftn(p, X, NFFT);
    const double xr = X[2 * (NFFT / cycles)], xi = X[2 * (NFFT / cycles) + 1];
    int best = 0;
    double hmax = xr * cos(QO_TAU * 0.0 / (double)cycles) - xi * sin(QO_TAU * 0.0 / (double)cycles);
    for (int i = 1; i < cycles; i++) {
        const double a = QO_TAU * (double)i / (double)cycles;
        const double c = xr * cos(a) - xi * sin(a);
        if (c > hmax) { hmax = c; best = i; }
    }
    free(p);
    free(X);
    return best;
}

/* ----------------------------------------------------------------- modem */

void qo_mixer_from_hz(double hz, double fs, float *rect2)
{
    float sn, cs;
    oracle_sincosf((float)(QO_TAU * hz / fs), &sn, &cs);
    rect2[0] = cs + sn * -0.0f;
    rect2[1] = sn * -1.0f;
}

qo_modem *qo_modem_new(double fs, double rs, int frame_size, float rrc_alpha, float loop_bw,
                       float min_freq, float max_freq, int timing_mode, int fixed_index)
{
    qo_modem *m = calloc(1, sizeof *m);
    m->fs = fs;
    m->rs = rs;
    m->cycles = (int)(fs / rs);
    m->frame_size = frame_size;
    m->nsym = frame_size / m->cycles;
    m->timing_mode = timing_mode;
    m->fixed_index = fixed_index;
    qo_rrc_make((float)fs, (float)rs, rrc_alpha, m->taps);
    qo_costas_create(&m->loop, loop_bw, min_freq, max_freq);
    m->input_frame = calloc((size_t)frame_size * 2, sizeof(float));
    m->decimated = calloc((size_t)m->nsym * 4, sizeof(float));
    m->costas_frame = calloc((size_t)m->nsym * 2, sizeof(float));
    m->symbols = calloc((size_t)m->nsym, 1);
    m->mix_phase[0] = 1.0f;
    m->mix_rect[0] = 1.0f;
    return m;
}

void qo_modem_free(qo_modem *m)
{
    if (!m) return;
    free(m->input_frame);
    free(m->decimated);
    free(m->costas_frame);
    free(m->symbols);
    free(m);
}

void qo_modem_reset(qo_modem *m)
{
    memset(m->rx_filter, 0, sizeof m->rx_filter);
    memset(m->input_frame, 0, sizeof(float) * 2 * (size_t)m->frame_size);
    memset(m->decimated, 0, sizeof(float) * 4 * (size_t)m->nsym);
    memset(m->costas_frame, 0, sizeof(float) * 2 * (size_t)m->nsym);
    memset(m->symbols, 0, (size_t)m->nsym);
    m->loop.phase = 0.0f;
    m->loop.freq = 0.0f;
    m->offset_hz = 0.0f;
    m->last_index = 0;
}

void qo_modem_set_mixer(qo_modem *m, const float *pr)
{
    m->mix_phase[0] = pr[0]; m->mix_phase[1] = pr[1];
    m->mix_rect[0] = pr[2];  m->mix_rect[1] = pr[3];
}

/* qpsk.c:125-217 once input_frame holds the complex block */
static void rx_tail(qo_modem *m)
{
    const int N = m->nsym, C = m->cycles, L = m->frame_size;
    /* the FFT estimate (new design, see qo_timing_fft_index) looks at the block BEFORE the filter and, starting
     * at sample 128, never reaches back into the carried delay line: it is the same function of the block in
     * the streaming mode as for an independent frame */
    const int fft_index = m->timing_mode == QO_TIMING_FFT ? qo_timing_fft_index(m->taps, m->input_frame, L, C) : 0;
    qo_rrc_fir(m->taps, m->rx_filter, m->input_frame, L);

    int index = (m->timing_mode == QO_TIMING_FIXED) ? m->fixed_index
                : (m->timing_mode == QO_TIMING_FFT) ? fft_index
                                                    : qo_timing_index(m->input_frame, L, C);
    m->last_index = index;

    for (int i = 0; i < N; i++) {
        const int e = N + i, src = i * C + index;
        m->decimated[2 * i] = m->decimated[2 * e];
        m->decimated[2 * i + 1] = m->decimated[2 * e + 1];
        /* a read past the block (index >= cycles, Q5) is DEFINED as 0 here */
        m->decimated[2 * e] = src < L ? m->input_frame[2 * src] : 0.0f;
        m->decimated[2 * e + 1] = src < L ? m->input_frame[2 * src + 1] : 0.0f;
    }
    for (int i = 0; i < N; i++)
        m->symbols[i] = (uint8_t)qo_costas_step(&m->loop, m->decimated[2 * i], m->decimated[2 * i + 1],
                                                &m->costas_frame[2 * i], &m->costas_frame[2 * i + 1]);
    m->offset_hz = (float)((double)m->loop.freq * m->rs / QO_TAU);
}

void qo_rx_frame_cplx(qo_modem *m, const float *in)
{
    memcpy(m->input_frame, in, sizeof(float) * 2 * (size_t)m->frame_size);
    rx_tail(m);
}

void qo_rx_frame_pcm(qo_modem *m, const int16_t *in)
{
    float pr = m->mix_phase[0], pi = m->mix_phase[1];
    const float rr = m->mix_rect[0], ri = m->mix_rect[1];
    for (int i = 0; i < m->frame_size; i++) {
        const float nr = pr * rr - pi * ri;
        const float ni = pr * ri + pi * rr;
        pr = nr;
        pi = ni;
        const float v = (float)in[i] / 16384.0f;
        m->input_frame[2 * i] = pr * v;
        m->input_frame[2 * i + 1] = pi * v;
    }
    const float mag = qo_cabsf(pr, pi);
    m->mix_phase[0] = pr / mag;
    m->mix_phase[1] = pi / mag;
    rx_tail(m);
}

/* ----------------------------------------------------- independent batch */

/* decimated FIR of one fresh frame: output i is the full-rate output at
 * sample i*cycles+index (zero history before the frame, zero past its end) */
static void fir_decimate_fresh(const float *taps, const float *x, int L, int C, int index, int N, float *d)
{
    float w[2 * QO_NTAPS];
    for (int i = 0; i < N; i++) {
        const int n = i * C + index;
        if (n >= L) { d[2 * i] = 0.0f; d[2 * i + 1] = 0.0f; continue; }
        const int first = n - (QO_NTAPS - 1);
        const float *win;
        if (first >= 0) {
            win = x + 2 * (size_t)first;
        } else {
            memset(w, 0, sizeof(float) * 2 * (size_t)(-first));
            memcpy(w + 2 * (-first), x, sizeof(float) * 2 * (size_t)(n + 1));
            win = w;
        }
        fir_point(taps, win, &d[2 * i], &d[2 * i + 1]);
    }
}

void qo_rx_batch_bw(double fs, double rs, int frame_size, float rrc_alpha, const float *loop_bws,
                    int nbw, float min_freq, float max_freq, int timing_mode, int fixed_index,
                    const float *in, int nframes, uint8_t *sym, float *freq, float *phase,
                    int32_t *index_out, int threads)
{
    const int C = (int)(fs / rs), L = frame_size, N = L / C;
    float taps[QO_NTAPS];
    qo_rrc_make((float)fs, (float)rs, rrc_alpha, taps);
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel num_threads(threads)
    {
        float *d = malloc(sizeof(float) * 2 * (size_t)N);
        float *filt = NULL;
        float mem[2 * QO_NTAPS];
#pragma omp for schedule(dynamic, 1)
        for (int f = 0; f < nframes; f++) {
            const float *x = in + 2 * (size_t)f * L;
            int index = fixed_index;
            if (timing_mode == QO_TIMING_HIST) {
                if (!filt) filt = malloc(sizeof(float) * 2 * (size_t)L);
                memcpy(filt, x, sizeof(float) * 2 * (size_t)L);
                memset(mem, 0, sizeof mem);
                qo_rrc_fir(taps, mem, filt, L);
                index = qo_timing_index(filt, L, C);
                for (int i = 0; i < N; i++) {
                    const int src = i * C + index;
                    d[2 * i] = src < L ? filt[2 * src] : 0.0f;
                    d[2 * i + 1] = src < L ? filt[2 * src + 1] : 0.0f;
                }
            } else {
                if (timing_mode == QO_TIMING_FFT)
                    index = qo_timing_fft_index(taps, x, L, C);
                fir_decimate_fresh(taps, x, L, C, index, N, d);
            }
            if (index_out) index_out[f] = index;
            for (int b = 0; b < nbw; b++) {
                qo_costas loop;
                qo_costas_create(&loop, loop_bws[b], min_freq, max_freq);
                const size_t o = (size_t)f * nbw + b;
                float zr, zi;
                for (int i = 0; i < N; i++) {
                    const int s = qo_costas_step(&loop, d[2 * i], d[2 * i + 1], &zr, &zi);
                    if (sym) sym[o * N + i] = (uint8_t)s;
                }
                if (freq) freq[o] = loop.freq;
                if (phase) phase[o] = loop.phase;
            }
        }
        free(d);
        free(filt);
    }
}

void qo_rx_batch(double fs, double rs, int frame_size, float rrc_alpha, float loop_bw,
                 float min_freq, float max_freq, int timing_mode, int fixed_index,
                 const float *in, int nframes, uint8_t *sym, float *freq, float *phase,
                 float *costas, int32_t *index_out, float *hz, int threads)
{
    const int C = (int)(fs / rs), L = frame_size, N = L / C;
    float taps[QO_NTAPS];
    qo_rrc_make((float)fs, (float)rs, rrc_alpha, taps);
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel num_threads(threads)
    {
        float *d = malloc(sizeof(float) * 2 * (size_t)N);
        float *filt = NULL;
        float mem[2 * QO_NTAPS];
#pragma omp for schedule(dynamic, 1)
        for (int f = 0; f < nframes; f++) {
            const float *x = in + 2 * (size_t)f * L;
            int index = fixed_index;
            if (timing_mode == QO_TIMING_HIST) {
                if (!filt) filt = malloc(sizeof(float) * 2 * (size_t)L);
                memcpy(filt, x, sizeof(float) * 2 * (size_t)L);
                memset(mem, 0, sizeof mem);
                qo_rrc_fir(taps, mem, filt, L);
                index = qo_timing_index(filt, L, C);
                for (int i = 0; i < N; i++) {
                    const int src = i * C + index;
                    d[2 * i] = src < L ? filt[2 * src] : 0.0f;
                    d[2 * i + 1] = src < L ? filt[2 * src + 1] : 0.0f;
                }
            } else {
                if (timing_mode == QO_TIMING_FFT)
                    index = qo_timing_fft_index(taps, x, L, C);
                fir_decimate_fresh(taps, x, L, C, index, N, d);
            }
            if (index_out) index_out[f] = index;
            qo_costas loop;
            qo_costas_create(&loop, loop_bw, min_freq, max_freq);
            float zr, zi;
            for (int i = 0; i < N; i++) {
                const int s = qo_costas_step(&loop, d[2 * i], d[2 * i + 1], &zr, &zi);
                if (sym) sym[(size_t)f * N + i] = (uint8_t)s;
                if (costas) {
                    costas[2 * ((size_t)f * N + i)] = zr;
                    costas[2 * ((size_t)f * N + i) + 1] = zi;
                }
            }
            if (freq) freq[f] = loop.freq;
            if (phase) phase[f] = loop.phase;
            if (hz) hz[f] = (float)((double)loop.freq * rs / QO_TAU);
        }
        free(d);
        free(filt);
    }
}

/* -------------------------------------------------------------------- TX */

void qo_tx_init(qo_tx *t, double fs, double rs, float rrc_alpha, double tx_hz)
{
    memset(t, 0, sizeof *t);
    t->cycles = (int)(fs / rs);
    qo_rrc_make((float)fs, (float)rs, rrc_alpha, t->taps);
    float sn, cs;
    oracle_sincosf(0.0f, &sn, &cs); /* cmplx(0.0f), qpsk.c:316 */
    t->phase[0] = cs + sn * 0.0f;
    t->phase[1] = sn * 1.0f;
    oracle_sincosf((float)(QO_TAU * tx_hz / fs), &sn, &cs); /* qpsk.c:320 */
    t->rect[0] = cs + sn * 0.0f;
    t->rect[1] = sn * 1.0f;
}

int qo_tx_symbols(qo_tx *t, int16_t *samples, const int *bits, int nsym)
{
    /* Gray map, qpsk.c:58-63,270,278-279 */
    static const float cre[4] = {1.0f, 0.0f, 0.0f, -1.0f};
    static const float cim[4] = {0.0f, 1.0f, -1.0f, 0.0f};
    const int C = t->cycles, n = nsym * C;
    float *sig = calloc((size_t)n * 2, sizeof(float));
    for (int i = 0, s = 0; i < nsym; i++, s += 2) {
        const int k = ((bits[s] & 1) << 1) | (bits[s + 1] & 1);
        sig[2 * (i * C)] = cre[k];
        sig[2 * (i * C) + 1] = cim[k];
    }
    qo_rrc_fir(t->taps, t->tx_filter, sig, n);
    float pr = t->phase[0], pi = t->phase[1];
    for (int i = 0; i < n; i++) {
        const float nr = pr * t->rect[0] - pi * t->rect[1];
        const float ni = pr * t->rect[1] + pi * t->rect[0];
        pr = nr;
        pi = ni;
        /* signal[i] *= phase; only the real part is kept (qpsk.c:250,260) */
        const float re = sig[2 * i] * pr - sig[2 * i + 1] * pi;
        samples[i] = (int16_t)(re * 16384.0f);
    }
    const float mag = qo_cabsf(pr, pi);
    t->phase[0] = pr / mag;
    t->phase[1] = pi / mag;
    free(sig);
    return n;
}

/* ------------------------------------------------------------------- FFT */

/* fft.c:38-64 / 66-96: recursive even/odd split, twiddle from libm cos/sin
 * per butterfly, product written out on real parts */
static void fft_rec(double *v, int n, double sgn)
{
    if (n <= 1) return;
    const int h = n / 2;
    double *tmp = malloc(sizeof(double) * 2 * (size_t)n);
    double *ve = tmp, *vo = tmp + 2 * h;
    for (int k = 0; k < h; k++) {
        ve[2 * k] = v[4 * k];         ve[2 * k + 1] = v[4 * k + 1];
        vo[2 * k] = v[4 * k + 2];     vo[2 * k + 1] = v[4 * k + 3];
    }
    fft_rec(ve, h, sgn);
    fft_rec(vo, h, sgn);
    for (int m = 0; m < h; m++) {
        const double a = QO_TAU * (double)m / (double)n;
        const double wr = cos(a), wi = sgn * sin(a); /* -sin forward (exact negation), +sin inverse */
        const double zr = wr * vo[2 * m] - wi * vo[2 * m + 1];
        const double zi = wr * vo[2 * m + 1] + wi * vo[2 * m];
        v[2 * m] = ve[2 * m] + zr;
        v[2 * m + 1] = ve[2 * m + 1] + zi;
        v[2 * (m + h)] = ve[2 * m] - zr;
        v[2 * (m + h) + 1] = ve[2 * m + 1] - zi;
    }
    free(tmp);
}

void qo_fftn(const double *in, double *out, int n)
{
    if (out != in) memmove(out, in, sizeof(double) * 2 * (size_t)n);
    fft_rec(out, n, -1.0);
    /* "out[i] / (double)n" on a complex double: component-wise division */
    for (int i = 0; i < 2 * n; i++)
        out[i] = out[i] / (double)n;
}

void qo_ifftn(const double *in, double *out, int n)
{
    if (out != in) memmove(out, in, sizeof(double) * 2 * (size_t)n);
    fft_rec(out, n, 1.0);
}

/* ------------------------------------------------------------ bit stages */

uint16_t qo_crc16(const uint8_t *data, int length)
{
    uint16_t crc = 0xFFFF;
    for (int i = 0; i < length; i++) {
        uint8_t x = (uint8_t)((crc >> 8) ^ data[i]);
        x ^= (uint8_t)(x >> 4);
        crc = (uint16_t)((crc << 8) ^ ((uint16_t)(x << 12)) ^ ((uint16_t)(x << 5)) ^ (uint16_t)x);
    }
    return crc;
}

void qo_interleave(uint8_t *inout, int nbytes, int dir)
{
    static const uint16_t primes[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59,
        61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157,
        163, 167, 173, 179, 181, 191, 193, 197, 199, 211, 223, 227, 229, 233, 239, 241, 251, 257,
        263, 269, 271, 277, 281, 283, 293, 307, 311, 313, 317, 331, 337, 347};
    const int np = (int)(sizeof primes / sizeof primes[0]);
    const uint16_t nbits = (uint16_t)(nbytes * 8);
    uint8_t *out = calloc((size_t)nbytes, 1);
    int idx = 1;
    while (idx < np && primes[idx] < nbits)
        idx++;
    const uint32_t b = primes[idx - 1];
    for (uint32_t n = 0; n < nbits; n++) {
        uint32_t i = n, j = (b * n) % nbits;
        if (dir == 1) { uint32_t t = j; j = i; i = t; }
        const uint32_t bit = (inout[i / 8] >> (i % 8)) & 1u;
        out[j / 8] |= (uint8_t)(bit << (j % 8));
    }
    memcpy(inout, out, (size_t)nbytes);
    free(out);
}

void qo_scramble_init(uint16_t *mem) { *mem = 0x4A80; }

void qo_scramble(uint8_t *sym, uint16_t *mem)
{
    for (int i = 0; i < 2; i++) {
        const uint16_t so = (uint16_t)(((*mem & 0x2) >> 1) ^ (*mem & 0x1));
        const uint16_t bit = (uint16_t)(((*sym >> i) & 0x1) ^ so);
        *sym = (uint8_t)((*sym & ~(1 << i)) | (bit << i));
        *mem = (uint16_t)((*mem >> 1) | (so << 14));
    }
}
