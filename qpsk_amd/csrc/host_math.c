/*
 * host_math.c -- the once-per-configuration arithmetic of the receive path,
 * done on the host in plain C exactly as the reference does it (SURVEY 8(a)
 * rows A2 and A5: "stays on host; taps uploaded"):
 *
 *   qpsk_host_rrc_taps     root-raised-cosine taps            rrc_fir.c:32-76
 *   qpsk_host_loop_gains   2nd-order loop gains               costas_loop.c:49-54
 *   qpsk_host_rect         mixer step e^{-j 2 pi f/fs}         qpsk.c:342, qpsk.h:36
 *   qpsk_host_twiddles     FFT twiddle table                  fft.c:55-56
 *
 * Built as C11 (-std=c11 => no FMA contraction), libm sinf/cosf/cos/sin as in
 * the reference, so the numbers are the reference's numbers on this machine.
 */
#include "host_math.h"

#include <math.h>

#define RRC_GAIN 1.85                    /* rrc_fir.h:14 */
#define PI_D 3.14159265358979323846      /* M_PI */

void qpsk_host_rrc_taps(float fs, float rs, float alpha, float taps[QPSK_HOST_NTAPS])
{
    const int centre = QPSK_HOST_NTAPS / 2;
    const float spb = fs / rs; /* samples per symbol */
    float sum = 0.f;

    for (int i = 0; i < QPSK_HOST_NTAPS; i++) {
        const float t = (float)(i - centre);                          /* tap position in samples */
        const float ph = (float)(PI_D * (double)t / (double)spb);     /* pi t / T, rounded to float */
        const float q = 4.f * alpha * t / spb;                        /* 4 alpha t / T */
        const float pole = q * q - 1.f;
        float num, den;

        if (fabsf(pole) >= 0.000001f) {
            /* regular point */
            const float cpart = cosf((1.f + alpha) * ph);
            if (i == centre)
                num = (float)((double)cpart + (double)(1.f - alpha) * PI_D / (double)(4.f * alpha));
            else
                num = cpart + sinf((1.f - alpha) * ph) / (4.f * alpha * t / spb);
            den = (float)((double)pole * PI_D);
        } else if (alpha == 1.f) {
            taps[i] = -1.f;
            sum += taps[i];
            continue;
        } else {
            /* t = +-T/(4 alpha): the removable singularity, evaluated by its limit form */
            const float am = (1.f - alpha) * ph;
            const float ap = (1.f + alpha) * ph;
            const double a = (double)(sinf(ap) * (1.f + alpha)) * PI_D;
            const double b = (double)cosf(am) * ((double)(1.f - alpha) * PI_D * (double)spb) / (double)(4.f * alpha * t);
            const double c = (double)(sinf(am) * spb * spb / (4.f * alpha * t * t));
            num = (float)(a - b + c);
            den = (float)((double)-32.f * PI_D * (double)alpha * (double)alpha * (double)t / (double)spb);
        }
        taps[i] = 4.f * alpha * num / den;
        sum += taps[i];
    }
    /* normalise the DC gain to GAIN (the FIR multiplies by GAIN once more, rrc_fir.c:28) */
    for (int i = 0; i < QPSK_HOST_NTAPS; i++)
        taps[i] = (float)(((double)taps[i] * RRC_GAIN) / (double)sum);
}

void qpsk_host_loop_gains(float damping, float loop_bw, float *alpha, float *beta)
{
    const float denom = (1.0f + (2.0f * damping * loop_bw)) + (loop_bw * loop_bw);
    *alpha = (4.0f * damping * loop_bw) / denom;
    *beta = (4.0f * loop_bw * loop_bw) / denom;
}

void qpsk_host_rect(double hz, double fs, float rect[2])
{
    const float a = (float)(2.0 * PI_D * hz / fs);
    const float s = sinf(a);
    rect[0] = cosf(a) + s * -0.0f; /* cosf(v) + sinf(v) * -I, qpsk.h:36 */
    rect[1] = s * -1.0f;
}

/* (cos, sin)(2 pi i / n) for i < n: the candidate timing phases of the FFT timing estimate (timing_fft.hip) */
void qpsk_host_phases(int n, double *cs)
{
    for (int i = 0; i < n; i++) {
        const double a = 2.0 * PI_D * (double)i / (double)n;
        cs[2 * i] = cos(a);
        cs[2 * i + 1] = sin(a);
    }
}

void qpsk_host_twiddles(int n, double *tw)
{
    for (int m = 0; m < n / 2; m++) {
        const double a = 2.0 * PI_D * (double)m / (double)n;
        tw[2 * m] = cos(a);
        tw[2 * m + 1] = sin(a);
    }
}
