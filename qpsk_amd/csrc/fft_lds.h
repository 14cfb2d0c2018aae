/*
 * fft_lds.h -- the butterfly stages of algorithms/fft.c (_fft/_ifft, fft.c:38-96) on a bit-reversed array in
 * LDS.  The reference recursion (even/odd split, recurse, combine) is exactly: bit-reversal permutation, then
 * log2(n) stages; the stage of size m combines e = v[lo], o = v[lo + m/2] with
 *     w = cos(TAU k/m) + (-/+ sin(TAU k/m)) j        (fft.c:55-56 forward, 85-86 inverse)
 *     z = (wr*o.re - wi*o.im) + (wr*o.im + wi*o.re) j   written out on real parts (fft.c:57-58)
 *     v[lo] = e + z,  v[lo + m/2] = e - z             (fft.c:60-63)
 * all in fp64 without contraction.  tw[] is the size-n table (cos, sin)(TAU j/n), j < n/2, built on the host
 * with libm as the reference does per butterfly; stage m reads entry k*(n/m), which is bit-identical to
 * computing the stage's own angle (scaling an argument by a power of two is exact).
 */
#ifndef QPSK_FFT_LDS_H
#define QPSK_FFT_LDS_H

#include <hip/hip_runtime.h>

namespace qpsk {

/* sgn = -1.0 forward, +1.0 inverse; all threads of the workgroup call it; ends with a barrier */
__device__ __forceinline__ void fft_lds_stages(double2 *v, const double2 *__restrict__ tw, int n, int log2n, int tid,
                                               int nthreads, double sgn)
{
    for (int s = 1; s <= log2n; s++) {
        const int half = 1 << (s - 1), stride = n >> s;
        for (int b = tid; b < n / 2; b += nthreads) {
            const int k = b & (half - 1);
            const int lo = ((b >> (s - 1)) << s) + k, hi = lo + half;
            const double2 w = tw[k * stride];
            const double wr = w.x, wi = sgn * w.y;
            const double2 e = v[lo], o = v[hi];
            const double zr = wr * o.x - wi * o.y;
            const double zi = wr * o.y + wi * o.x;
            v[lo] = make_double2(e.x + zr, e.y + zi);
            v[hi] = make_double2(e.x - zr, e.y - zi);
        }
        __syncthreads();
    }
}

} // namespace qpsk
#endif
