/*
 * timing_scan.hip -- the reference's histogram timing estimate (qpsk.c:127-180) WITHOUT a filtered block in memory:
 * full-rate rrc_fir() (rrc_fir.c:17-30, fresh delay line) and the amplitude-histogram scan fused in one kernel
 * through LDS.  Output: the decimation index per frame (and the summed histograms, for the tests).
 *
 * Why: the estimate needs EVERY filtered sample (running average over all 8 phases, qpsk.c:131-138), i.e. the
 * VALU-bound full-rate FIR (508 unfused fp32 operations per sample).  As three kernels (rrc_fir_kernel ->
 * timing_hist8_kernel -> pipeline kernel) the histogram mode of qpsk_rx_batch wrote the filtered block (1 x the input),
 * read it back for the scan (1 x) and then filtered the decimated outputs again from the raw input (1 x): ~4 x the
 * algorithmic bytes and 0.96-1.0 ms at config 2, of which 0.135 ms for a scan whose serial chain leaves its SIMD
 * nearly idle.  Here the filtered samples never leave the CU: FIR waves filter 256-sample tiles into a 2-tile LDS
 * ring (I and Q planes), scan waves consume them; the input is read once here and once by the pipeline kernel that
 * follows with the index (2.0 x), nothing but 4 bytes per frame is written.
 *
 * Workgroup = 16 frames, 12 waves, one workgroup per CU at config 2:
 *   waves 0-3    scan waves (one per SIMD), 4 frames each: timing_hist8_kernel's scan as it stands -- 8 lanes per
 *                (frame, I|Q) running the same average/max chain on the same samples (LDS broadcast reads), lane q
 *                owning threshold hv*q (see kernels.hip) -- fed from the ring instead of global memory;
 *   waves 4-11   FIR waves (two per SIMD), 2 frames each, owned outright: 16-byte loads a tile ahead, window staged
 *                from registers (history = the last 128-sample block of the previous tile, 4 VGPRs per frame), the
 *                filter as the generated stream fir_full8_asm.h (8 CONSECUTIVE outputs per lane = one symbol,
 *                1016 + 1016 packed operations, taps 0..126 in order in one fp32 accumulator per output), the
 *                second GAIN (rrc_fir.c:28, in double), then the lane's symbol as two 16-byte words per plane.
 * Counters: ready[FIR wave], consumed[scan wave]; bounded spins; the context's status word on timeout.
 *
 * Conditions (host-checked; everything else takes the three-kernel path): CYCLES = 8, frame_size a multiple of 256,
 * 16-byte aligned input.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qpsk_device.h"
#include "costas_asm.h"      /* lds_addr() */
#include "fir_full8_asm.h"
#include "fir_full8s_asm.h"
static_assert(qpsk::FIR_FULL8_ASM_END_VGPR <= 168, "timing_scan_kernel: 12 waves per workgroup = three per SIMD = at most 168 VGPRs (regenerate fir_full8_asm.h from VGPR 80, tools/gen_fir_asm.py)");
#include "kernels.h"

namespace qpsk {

namespace tscan {
constexpr int C = 8;
constexpr int G = 16;                 /* frames per workgroup */
constexpr int NSCAN = 4, NFIR = 8;    /* scan waves (4 frames each), FIR waves (2 frames each) */
constexpr int UF = 2, QL = 32, R = 8; /* frames per FIR wave, lanes per frame, outputs per lane */
constexpr int TILE = QL * R;          /* 256 outputs per frame per round */
constexpr int DRO = 2;                /* tiles in the output ring */
constexpr int PADS = 2;               /* window: position p at slot p + 2 (p / 8): lanes 80 bytes apart */
constexpr int WPOS = TILE + HIST;     /* 382 window positions per frame */
constexpr int WSF = 480;              /* slots per frame window (positions 0..381 -> slots 0..475) */
constexpr int PITCH = TILE + 4;       /* floats per plane row: 16-byte aligned, 4 banks apart */
constexpr int THREADS = 64 * (NSCAN + NFIR);
constexpr int SPIN_LIMIT = 1 << 24;
__device__ __host__ constexpr int slot_of(int p) { return p + PADS * (p / R); }
static_assert(slot_of(WPOS - 1) < WSF && WSF % 2 == 0, "window geometry");

struct Smem {
    float taps[128];
    int ready[NFIR];
    int consumed[NSCAN];
    int abort_flag;
    int pad_[3];
};

__device__ __forceinline__ bool wait_ge(int *p, int target, int *abort_flag)
{
    int spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}

__device__ __forceinline__ void publish(int *p, int v)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
} // namespace tscan

/* SYM: the filter is symmetric (host-checked): the stream with its taps in SGPRs (fir_full8s_asm.h: no tap reads from LDS) */
template <bool SYM>
__global__ void __launch_bounds__(tscan::THREADS)
timing_scan_kernel(const float2 *__restrict__ x, int nframes, int frame_size, const float *__restrict__ taps_g,
                   int32_t *index, int32_t *hist_out, int *status, size_t pitch)
{
    using namespace tscan;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem *sm = reinterpret_cast<Smem *>(smem_raw);
    float2 *win = reinterpret_cast<float2 *>(smem_raw + sizeof(Smem));                 /* [G][WSF] */
    float *ring = reinterpret_cast<float *>(win + (size_t)G * WSF);                    /* [G][DRO][2 planes][PITCH] */
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f0 = blockIdx.x * G;
    const int ntiles = frame_size / TILE;

    for (int i = tid; i < 128; i += THREADS) sm->taps[i] = i < NTAPS ? taps_g[i] : 0.0f;
    if (tid < NFIR) sm->ready[tid] = 0;
    if (tid < NSCAN) sm->consumed[tid] = 0;
    if (tid == 0) sm->abort_flag = 0;
    __syncthreads();

    if (wave < NSCAN) {
        /* ================================ scan wave: frames 4*wave .. 4*wave+3 of the workgroup ================= */
        __builtin_amdgcn_s_setprio(3);      /* a latency chain: it takes few issue slots and must get them at once */
        const int fl = lane >> 4, comp = (lane >> 3) & 1, q = lane & 7;      /* lane = 16*frame + 8*component + q */
        const int g = 4 * wave + fl;
        const float qf = (float)q;
        float av = 0.0f, mx = 0.0f;
        int cum = 0;
        bool ok = true;
        for (int t = 0; t < ntiles && ok; t++) {
            /* the tile of this lane's frame comes from FIR wave g / 2; every lane waits for its own producer */
            ok = wait_ge(&sm->ready[g >> 1], t + 1, &sm->abort_flag);
            if (!__all(ok)) { ok = false; break; }
            const float4 *row = reinterpret_cast<const float4 *>(ring + ((size_t)(g * DRO + (t % DRO)) * 2 + comp) * PITCH);
#pragma unroll 4
            for (int s = 0; s < TILE / 8; s++) {
                const float4 a = row[2 * s], b = row[2 * s + 1];
                av += fabsf(a.x); av += fabsf(a.y); av += fabsf(a.z); av += fabsf(a.w);      /* qpsk.c:131-136 */
                av += fabsf(b.x); av += fabsf(b.y); av += fabsf(b.z); av += fabsf(b.w);
                av *= 0.125f;                           /* av /= CYCLES (qpsk.c:137-138) */
                if (av > mx) mx = av;                   /* qpsk.c:140-145 */
                const float th = (mx * 0.125f) * qf;    /* (max / 8.0f) * q, qpsk.c:147-165 */
                cum += (av <= th) ? 0 : 1;
            }
            if (lane == 0) publish(&sm->consumed[wave], t + 1);
        }
        if (!ok) {
            if (lane == 0) __hip_atomic_store(status, STATUS_PIPE_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        /* hist[q] = cum[q-1] - cum[q], cum[0] = number of symbols; index = first argmax of hist_i + hist_q (qpsk.c:173-180) */
        if (q == 0) cum = frame_size / 8;
        int h = __shfl_up(cum, 1) - cum;
        if (q == 0) h = 0;
        h += __shfl_xor(h, 8);
        int hmax = 0, best = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int hk = __shfl(h, (lane & ~7) + k);
            if (hk > hmax) { hmax = hk; best = k; }
        }
        if (f0 + g < nframes && comp == 0) {
            if (hist_out) hist_out[(size_t)(f0 + g) * 8 + q] = h;
            if (q == 0) index[f0 + g] = best;
        }
        return;
    }

    /* ==================================== FIR wave: frames 2*w, 2*w+1 of the workgroup ======================== */
    const int w = wave - NSCAN;
    const int fl = lane / QL, q = lane % QL;
    const int g = UF * w + fl;
    const bool fv[UF] = {f0 + UF * w < nframes, f0 + UF * w + 1 < nframes};
    const float4 *src[UF];
#pragma unroll
    for (int ff = 0; ff < UF; ff++)
        src[ff] = reinterpret_cast<const float4 *>(x + (size_t)(fv[ff] ? f0 + UF * w + ff : 0) * pitch);      /* frames pitch samples apart */
    float2 *mywin = win + (size_t)(UF * w) * WSF;
    const unsigned rd_addr = lds_addr(mywin + fl * WSF + (R + PADS) * q);   /* position 8q -> slot 10q */
    const unsigned tap_addr = lds_addr(sm->taps);
    /* the lane loads samples 2*lane, 2*lane+1 of each 128-sample block of a frame's tile: window position of sample s
     * of the tile is s + 126 (history: the previous tile's last 126 samples at positions 0..125) */
    const int p0 = 2 * lane + HIST;
    float4 hist[UF], pre[UF][2];
#pragma unroll
    for (int ff = 0; ff < UF; ff++) hist[ff] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);    /* a fresh delay line */
    auto prefetch = [&](int t) {
#pragma unroll
        for (int ff = 0; ff < UF; ff++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                float4 v = load_once(&src[ff][(t * TILE + 128 * j + 2 * lane) >> 1]);      /* read once (qpsk_device.h) */
                if (!fv[ff]) v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                pre[ff][j] = v;
            }
    };
    prefetch(0);
    bool ok = true;
#ifdef QPSK_PIPE_PROFILE
    /* measurement build: where FIR wave w of workgroup 0 spends its cycles (printed per tile) */
    const bool prof = blockIdx.x == 0 && hist_out == nullptr && (nframes & 1) == 1;
    unsigned long long tacc[4] = {0, 0, 0, 0}, tlast = 0;
    auto tick = [&](int k) {
        if (prof) {
            unsigned long long tt;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
            if (k >= 0) tacc[k] += tt - tlast;
            tlast = tt;
        }
    };
    tick(-1);
#else
    auto tick = [](int) {};
#endif
    for (int t = 0; t < ntiles && ok; t++) {
        /* The two FIR waves of a SIMD (w and w ^ 4) are served oldest first: left alone, the older one runs at nearly
         * full rate, finishes its two frames after ~55 % of the kernel and leaves the younger one to run the rest
         * alone at a lone wave's rate (every LDS instruction's latency exposed) [measured: 13.5 k against 25 k
         * cycles per tile].  So the one that is behind gets the higher priority, tile by tile: they stay within a
         * tile of each other and overlap to the end. */
#ifndef QPSK_TSCAN_NO_BALANCE      /* A/B builds only */
        {
            const int pt = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sm->ready[w ^ 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (pt > t) __builtin_amdgcn_s_setprio(2);
            else if (pt < t) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(1);
        }
#endif
        /* window from registers: p0 is even, so a lane's pair is always one aligned 16-byte word of the image */
#pragma unroll
        for (int ff = 0; ff < UF; ff++) {
            float2 *wf = mywin + ff * WSF;
            if (lane >= 1) *reinterpret_cast<float4 *>(wf + slot_of(p0 - 128)) = hist[ff];   /* positions 2*lane - 2, - 1 */
            *reinterpret_cast<float4 *>(wf + slot_of(p0)) = pre[ff][0];
            *reinterpret_cast<float4 *>(wf + slot_of(p0 + 128)) = pre[ff][1];
            hist[ff] = pre[ff][1];
        }
        if (t + 1 < ntiles) prefetch(t + 1);
        tick(0);
        v2f a0, a1, a2, a3, a4, a5, a6, a7;
        if constexpr (SYM) fir_full8s_asm(rd_addr, taps_g, a0, a1, a2, a3, a4, a5, a6, a7);
        else fir_full8_asm(rd_addr, tap_addr, a0, a1, a2, a3, a4, a5, a6, a7);
        tick(1);
        /* the ring slot is free once the scan wave of this frame has finished tile t - DRO */
        if (t >= DRO) ok = wait_ge(&sm->consumed[g >> 2], t - DRO + 1, &sm->abort_flag);
        if (!ok) break;
        tick(2);
        /* rrc_fir.c:28: y * GAIN in double, narrowed; the lane's 8 outputs are one symbol: two 16-byte words per plane */
        const float2 y0 = fir_gain(make_float2(a0.x, a0.y)), y1 = fir_gain(make_float2(a1.x, a1.y)),
                     y2 = fir_gain(make_float2(a2.x, a2.y)), y3 = fir_gain(make_float2(a3.x, a3.y)),
                     y4 = fir_gain(make_float2(a4.x, a4.y)), y5 = fir_gain(make_float2(a5.x, a5.y)),
                     y6 = fir_gain(make_float2(a6.x, a6.y)), y7 = fir_gain(make_float2(a7.x, a7.y));
        float *pi = ring + ((size_t)(g * DRO + (t % DRO)) * 2 + 0) * PITCH + R * q;
        float *pq = pi + PITCH;
        reinterpret_cast<float4 *>(pi)[0] = make_float4(y0.x, y1.x, y2.x, y3.x);
        reinterpret_cast<float4 *>(pi)[1] = make_float4(y4.x, y5.x, y6.x, y7.x);
        reinterpret_cast<float4 *>(pq)[0] = make_float4(y0.y, y1.y, y2.y, y3.y);
        reinterpret_cast<float4 *>(pq)[1] = make_float4(y4.y, y5.y, y6.y, y7.y);
        if (lane == 0) publish(&sm->ready[w], t + 1);
        tick(3);
    }
#ifdef QPSK_PIPE_PROFILE
    if (prof && lane == 0)
        printf("timing_scan FIR wave %d: %d tiles; cycles per tile: staging + prefetch %llu, filter %llu, wait for the scan %llu, gain + ring write %llu\n",
               w, ntiles, tacc[0] / ntiles, tacc[1] / ntiles, tacc[2] / ntiles, tacc[3] / ntiles);
#endif
    if (!ok && lane == 0) __hip_atomic_store(status, STATUS_PIPE_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

size_t timing_scan_lds_bytes(void)
{
    using namespace tscan;
    return sizeof(Smem) + sizeof(float2) * (size_t)G * WSF + sizeof(float) * (size_t)G * DRO * 2 * PITCH;
}

int timing_scan_tile(void) { return tscan::TILE; }

int prepare_timing_scan(void)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(timing_scan_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       MAX_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(timing_scan_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    MAX_LDS_BYTES);
}

/* frame_size % 256 == 0, CYCLES = 8, x 16-byte aligned (host-checked) */
int launch_timing_scan(const float *x, int nframes, int frame_size, const float *taps, int32_t *index, int32_t *hist,
                       int *status, hipStream_t s, size_t pitch, bool symmetric)
{
    using namespace tscan;
    if (pitch == 0) pitch = (size_t)frame_size;
    if (frame_size % TILE != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0 || (pitch & 1)) return (int)hipErrorInvalidValue;
    if (symmetric)
        hipLaunchKernelGGL(timing_scan_kernel<true>, dim3((nframes + G - 1) / G), dim3(THREADS), timing_scan_lds_bytes(), s,
                           reinterpret_cast<const float2 *>(x), nframes, frame_size, taps, index, hist, status, pitch);
    else
        hipLaunchKernelGGL(timing_scan_kernel<false>, dim3((nframes + G - 1) / G), dim3(THREADS), timing_scan_lds_bytes(), s,
                           reinterpret_cast<const float2 *>(x), nframes, frame_size, taps, index, hist, status, pitch);
    return (int)hipGetLastError();
}

} // namespace qpsk
