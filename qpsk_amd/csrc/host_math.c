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

/* transmit carrier step e^{+j 2 pi f/fs}: fbb_tx_rect = cmplx(TAU * hz / FS)  (qpsk.c:320, qpsk.h:35) */
void qpsk_host_rect_tx(double hz, double fs, float rect[2])
{
    const float a = (float)(2.0 * PI_D * hz / fs);
    const float s = sinf(a);
    rect[0] = cosf(a) + s * 0.0f; /* cosf(v) + sinf(v) * I */
    rect[1] = s * 1.0f;
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

/* the multiplier of the golden-prime interleaver: the largest table prime below nbits (interleave.c:15-23,40-45) */
unsigned qpsk_host_interleave_prime(unsigned nbits)
{
    static const unsigned short primes[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67,
        71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167, 173, 179, 181,
        191, 193, 197, 199, 211, 223, 227, 229, 233, 239, 241, 251, 257, 263, 269, 271, 277, 281, 283, 293, 307, 311,
        313, 317, 331, 337, 347};
    const unsigned count = sizeof primes / sizeof primes[0];
    unsigned k = 1;
    while (k < count && primes[k] < nbits)
        k++;
    return primes[k - 1];
}

/* keystream of the DVB scrambler 1 + x^14 + x^15 from SEED 0x4A80, two bits per symbol (bit-scramble.c:11-69):
 * ks[i] = (second output bit << 1) | first output bit of symbol i */
void qpsk_host_scramble_keystream(unsigned char *ks, int nsym)
{
    unsigned short mem = 0x4A80;
    for (int i = 0; i < nsym; i++) {
        unsigned char k = 0;
        for (int bit = 0; bit < 2; bit++) {
            const unsigned short out = (unsigned short)(((mem & 0x2) >> 1) ^ (mem & 0x1));
            k |= (unsigned char)(out << bit);
            mem = (unsigned short)((mem >> 1) | (out << 14));
        }
        ks[i] = k;
    }
}

void qpsk_host_fir_fast_tables(const float taps[QPSK_HOST_NTAPS], float *H, float *tw)
{
    const int n = 512;
    for (int m = 0; m < n; m++) {
        const double a = 2.0 * PI_D * (double)m / (double)n;
        tw[2 * m] = (float)cos(a);
        tw[2 * m + 1] = (float)-sin(a);
    }
    for (int k = 0; k < n; k++) {
        double re = 0.0, im = 0.0;
        for (int i = 0; i < QPSK_HOST_NTAPS; i++) {          /* h[i] = taps[126 - i]: memory[126] is the newest sample */
            const double a = 2.0 * PI_D * (double)((i * k) % n) / (double)n;
            const double h = (double)taps[QPSK_HOST_NTAPS - 1 - i];
            re += h * cos(a);
            im -= h * sin(a);
        }
        H[2 * k] = (float)(re * RRC_GAIN / (double)n);       /* GAIN again on taps that already sum to GAIN (rrc_fir.c:28, 73-74) */
        H[2 * k + 1] = (float)(im * RRC_GAIN / (double)n);
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
