/* host_math.h -- host-side configuration arithmetic (see host_math.c) */
#ifndef QPSK_HOST_MATH_H
#define QPSK_HOST_MATH_H

#ifdef __cplusplus
extern "C" {
#endif

#define QPSK_HOST_NTAPS 127

void qpsk_host_rrc_taps(float fs, float rs, float alpha, float taps[QPSK_HOST_NTAPS]);
void qpsk_host_loop_gains(float damping, float loop_bw, float *alpha, float *beta);
void qpsk_host_rect(double hz, double fs, float rect[2]);
void qpsk_host_rect_tx(double hz, double fs, float rect[2]);
void qpsk_host_twiddles(int n, double *tw); /* tw[n/2][2] = cos, sin of 2 pi m / n */
void qpsk_host_phases(int n, double *cs);   /* cs[n][2]   = cos, sin of 2 pi i / n */
/* qpsk_rrc_fir_batch_fast (firfast.hip): H[512][2] = GAIN / 512 * DFT512 of h[k] = taps[126 - k] (rrc_fir.c:17-30 as a
 * convolution), tw[512][2] = exp(-2 pi j m / 512); double arithmetic, narrowed to float */
void qpsk_host_fir_fast_tables(const float taps[QPSK_HOST_NTAPS], float *H, float *tw);
unsigned qpsk_host_interleave_prime(unsigned nbits);
void qpsk_host_scramble_keystream(unsigned char *ks, int nsym);

#ifdef __cplusplus
}
#endif
#endif
