/*
 * firstream.hip -- rrc_fir() (rrc_fir.c:17-30) at full rate on a batch of delay lines, as the generated instruction
 * stream fir_full8s_asm.h: the kernel behind qpsk_rrc_fir_batch(), the drop-in rrc_fir(), the streams and the
 * three-kernel histogram path whenever the filter is symmetric (an RRC filter is; kernels.hip's compiler-scheduled
 * rrc_fir_kernel keeps every other tap set).
 *
 * VALU-bound: 508 unfused fp32 operations per output, 2032 packed instructions per 512 outputs.  What
 * the issue rate depends on [measured, profiles/r03_power_ceiling.txt: 5.2 cycles per packed instruction and SIMD with
 * two waves on it, 4.8 with three, 4.7 with four] is waves per SIMD, so the stream keeps its taps in SGPRs (no tap
 * registers, no tap reads: a third of the LDS instructions gone) and fits 104 VGPRs; the kernel runs four waves per SIMD.
 *
 * One WAVE owns a run of consecutive 512-output tiles of ONE delay line:
 *   - four coalesced 16-byte loads per lane and tile, a tile ahead, into registers;
 *   - the window (126 samples of history + 512 new ones) staged from registers in the image the stream reads (position p
 *     at float2 slot p + 2 (p / 8): lanes 80 bytes apart, pairs are aligned 16-byte words); the history of the next tile IS
 *     the last 128-sample block just loaded (4 VGPRs), the first tile's comes from the caller's delay line
 *     (memory[1..126]; memory[0] is shifted out by the first step, rrc_fir.c:19), from the samples in front of the run,
 *     or is zero (fresh delay line);
 *   - one pass of the stream: lane l's 8 accumulators are outputs 8l .. 8l+7, each summed taps 0..126 in order, unfused;
 *   - the second GAIN in double (rrc_fir.c:28), then lane-major -> sample-major through the same LDS image, so the
 *     stores are coalesced 16-byte words like the loads.
 * Waves never synchronise with each other.  Frames are cut into `parts` runs when the batch has fewer delay lines than
 * the chip has wave slots (a run that does not start its frame re-reads the 126 samples in front of it).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qpsk_device.h"
#include "costas_asm.h"      /* lds_addr() */
#include "fir_full8s_asm.h"
#include "kernels.h"

namespace qpsk {

namespace firs {
constexpr int R = 8, PADS = 2;
constexpr int TILE = 64 * R;               /* 512 outputs per pass of the stream */
constexpr int WSLOTS = 808;                /* float2 slots per wave (timing_fft.hip has the arithmetic) */
constexpr int WAVES = 4;                   /* per workgroup */
constexpr int WAVES_PER_SIMD = 4;
__device__ __host__ constexpr int slot_of(int p) { return p + PADS * (p / R); }
static_assert(slot_of(TILE + HIST - 1) < WSLOTS && slot_of(R * 63) + slot_of(NTAPS + R) + 2 <= WSLOTS, "window geometry");
static_assert(FIR_FULL8S_ASM_END_VGPR <= 104, "four waves per SIMD: 128 VGPRs");

__device__ __forceinline__ void wave_sync()      /* one wave, in-order LDS: only the compiler needs telling */
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
} // namespace firs

__global__ void __launch_bounds__(64 * firs::WAVES) __attribute__((amdgpu_waves_per_eu(firs::WAVES_PER_SIMD, firs::WAVES_PER_SIMD)))
rrc_fir_stream_kernel(const float2 *__restrict__ x, const float2 *__restrict__ memory, float2 *__restrict__ y,
                      const float *__restrict__ taps_g, int nframes, int length, size_t in_pitch, int parts, int tpp, int aligned)
{
    using namespace firs;
    __shared__ __attribute__((aligned(16))) float2 wins[WAVES][WSLOTS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = blockIdx.x * WAVES + wave;
    const int f = unit / parts, part = unit - f * parts;
    if (f >= nframes) return;
    const int ntiles = (length + TILE - 1) / TILE;
    const int t0 = part * tpp, t1 = min(ntiles, t0 + tpp);
    if (t0 >= t1) return;
    float2 *win = wins[wave];
    const unsigned rd_addr = lds_addr(win + (R + PADS) * lane);      /* position 8 lane -> slot 10 lane */
    const float2 *src = x + (size_t)f * in_pitch;
    float2 *dst = y + (size_t)f * length;

    /* lane l holds samples 2l, 2l+1 of each 128-sample block of a tile */
    float4 pre[4], hist;
    auto load_tile = [&](int t) {
        const int n0 = t * TILE;
        if (aligned && n0 + TILE <= length) {
            const float4 *s4 = reinterpret_cast<const float4 *>(src + n0);
#pragma unroll
            for (int j = 0; j < 4; j++) pre[j] = s4[64 * j + lane];
        } else {      /* the frame's last tile, or frames that do not start on 16-byte boundaries */
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int n = n0 + 128 * j + 2 * lane;
                const float2 a = n < length ? src[n] : make_float2(0.0f, 0.0f);
                const float2 b = n + 1 < length ? src[n + 1] : make_float2(0.0f, 0.0f);
                pre[j] = make_float4(a.x, a.y, b.x, b.y);
            }
        }
    };
    {   /* history of the run's first tile: the "block" in front of it, samples n0 - 128 + 2l, + 1 (lane 0's pair is never used) */
        const int n = t0 * TILE - 128 + 2 * lane;
        auto at = [&](int m) {
            if (m >= 0) return src[m];
            return memory && m >= -HIST ? memory[(size_t)f * NTAPS + (NTAPS + m)] : make_float2(0.0f, 0.0f);   /* m = -1 -> memory[126] */
        };
        const float2 a = at(n), b = at(n + 1);
        hist = make_float4(a.x, a.y, b.x, b.y);
    }
    load_tile(t0);
    const int p0 = 2 * lane + HIST;     /* window position of sample 2 lane of the tile: even, so a pair is one aligned word */
    for (int t = t0; t < t1; t++) {
        if (lane >= 1) *reinterpret_cast<float4 *>(win + slot_of(p0 - 128)) = hist;      /* positions 2 lane - 2, 2 lane - 1 */
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<float4 *>(win + slot_of(p0 + 128 * j)) = pre[j];
        hist = pre[3];
        if (t + 1 < t1) load_tile(t + 1);
        wave_sync();
        v2f a0, a1, a2, a3, a4, a5, a6, a7;
        fir_full8s_asm(rd_addr, taps_g, a0, a1, a2, a3, a4, a5, a6, a7);      /* ends with every LDS read returned */
        /* rrc_fir.c:28, then through the image again: output o of the tile at slot o + 2 (o / 8) */
        const float2 y0 = fir_gain(make_float2(a0.x, a0.y)), y1 = fir_gain(make_float2(a1.x, a1.y)),
                     y2 = fir_gain(make_float2(a2.x, a2.y)), y3 = fir_gain(make_float2(a3.x, a3.y)),
                     y4 = fir_gain(make_float2(a4.x, a4.y)), y5 = fir_gain(make_float2(a5.x, a5.y)),
                     y6 = fir_gain(make_float2(a6.x, a6.y)), y7 = fir_gain(make_float2(a7.x, a7.y));
        float4 *wl = reinterpret_cast<float4 *>(win + (R + PADS) * lane);
        wl[0] = make_float4(y0.x, y0.y, y1.x, y1.y);
        wl[1] = make_float4(y2.x, y2.y, y3.x, y3.y);
        wl[2] = make_float4(y4.x, y4.y, y5.x, y5.y);
        wl[3] = make_float4(y6.x, y6.y, y7.x, y7.y);
        wave_sync();
        const int n0 = t * TILE;
        if (aligned && (length & 1) == 0 && n0 + TILE <= length) {
            float4 *d4 = reinterpret_cast<float4 *>(dst + n0);
#pragma unroll
            for (int j = 0; j < 4; j++)
                d4[64 * j + lane] = *reinterpret_cast<const float4 *>(win + slot_of(128 * j + 2 * lane));
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int o = 128 * j + 2 * lane;
                const float4 v = *reinterpret_cast<const float4 *>(win + slot_of(o));
                if (n0 + o < length) dst[n0 + o] = make_float2(v.x, v.y);
                if (n0 + o + 1 < length) dst[n0 + o + 1] = make_float2(v.z, v.w);
            }
        }
        wave_sync();      /* the next tile's staging overwrites the image */
    }
}

/* symmetric taps only (the caller checks).  ncu: compute units of the device */
int launch_rrc_fir_stream(const float *x, const float *memory, float *y, const float *taps, int nframes, int length,
                          hipStream_t s, size_t in_pitch, int ncu)
{
    using namespace firs;
    if (in_pitch == 0) in_pitch = (size_t)length;
    if (ncu < 1) ncu = 256;
    const int ntiles = (length + TILE - 1) / TILE;
    /* runs per frame: enough to give every wave slot of the chip a run when the batch has few delay lines */
    const long long slots = (long long)ncu * 4 * WAVES_PER_SIMD;
    int parts = (int)((slots + nframes - 1) / nframes);
    if (parts > ntiles) parts = ntiles;
    if (parts < 1) parts = 1;
    const int tpp = (ntiles + parts - 1) / parts;
    parts = (ntiles + tpp - 1) / tpp;
    const int aligned = (reinterpret_cast<uintptr_t>(x) % 16) == 0 && (reinterpret_cast<uintptr_t>(y) % 16) == 0 && (in_pitch % 2) == 0;
    const long long units = (long long)nframes * parts;
    hipLaunchKernelGGL(rrc_fir_stream_kernel, dim3((unsigned)((units + WAVES - 1) / WAVES)), dim3(64 * WAVES), 0, s,
                       reinterpret_cast<const float2 *>(x), reinterpret_cast<const float2 *>(memory), reinterpret_cast<float2 *>(y), taps,
                       nframes, length, in_pitch, parts, tpp, aligned);
    return (int)hipGetLastError();
}

} // namespace qpsk
