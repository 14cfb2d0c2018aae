/*
 * carrier.h -- the streams' carrier while it is ONE carrier for all of them (streamscan.hip MODE 2; api.cpp carrier_shared).
 *
 * fbb_rx_phase *= fbb_rx_rect per sample (qpsk.c:115) is a strictly serial complex recurrence: 16384 dependent steps per 16384-sample
 * block, ~50 cycles each for a wave alone on its SIMD and ~100 beside three busy ones.  It does not depend on the data, and
 * qpsk_streams_reset() gives every stream the same frequency and starting phase: the block's phases are computed ONCE, a block ahead,
 * into a table every workgroup reads -- the first 11/16 of a table by a spare wave of stream_scan_kernel's workgroup 0 (beside that
 * workgroup's filter waves, hidden behind them), the rest by a spare wave of the loop kernel that follows it in the same call
 * (costas_pipe_kernel, whose SIMDs are nearly idle).  State st[8]: [0..1] the phase reached so far, [2..3] fbb_rx_rect, [4..5] the
 * phase the table under construction started from (what every stream's own mixer state would be, carrier_broadcast_kernel).
 */
#ifndef QPSK_CARRIER_H
#define QPSK_CARRIER_H

#include <hip/hip_runtime.h>

namespace qpsk {

/* pairs of samples [from, to) of a block of frame_size samples (from, to even counts of PAIRS are not required; one lane runs this).
 * from == 0 records the block's starting phase; to == frame_size / 2 ends the block with qpsk.c:120's normalisation. */
__device__ __forceinline__ void carrier_block(float *st, float2 *tab, int frame_size, int from, int to)
{
    float2 p = make_float2(st[0], st[1]);
    const float rr = st[2], ri = st[3];
    float nri = -ri;
    asm volatile("" : "+v"(nri));   /* opaque: keeps the compiler from folding the sign back into two packed adds */
    if (from == 0) {
        st[4] = p.x;
        st[5] = p.y;
    }
    float4 *t4 = reinterpret_cast<float4 *>(tab);
#pragma unroll 4
    for (int i = from; i < to; i++) {      /* fbb_rx_phase *= fbb_rx_rect twice (mixer_kernel's step): the phases AFTER each step */
        float2 a = make_float2(p.x * rr, p.y * rr), b = make_float2(p.y * nri, p.x * ri);   /* p.y * (-ri) = -(p.y * ri) exactly */
        const float2 p0 = make_float2(a.x + b.x, a.y + b.y);
        a = make_float2(p0.x * rr, p0.y * rr);
        b = make_float2(p0.y * nri, p0.x * ri);
        p = make_float2(a.x + b.x, a.y + b.y);
        t4[i] = make_float4(p0.x, p0.y, p.x, p.y);
    }
    if (to == frame_size / 2) {
        const float mag = (float)sqrt((double)p.x * (double)p.x + (double)p.y * (double)p.y);   /* qpsk.c:120 */
        p = make_float2(p.x / mag, p.y / mag);
    }
    st[0] = p.x;
    st[1] = p.y;
}

/* where stream_scan_kernel's spare wave stops and the loop kernel's takes over (pairs of samples) */
__device__ __host__ inline int carrier_split(int frame_size) { return ((frame_size / 2) * 11 / 16) & ~3; }

} // namespace qpsk
#endif
