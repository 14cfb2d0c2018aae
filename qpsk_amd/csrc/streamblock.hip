/*
 * streamblock.hip -- ONE launch per rx_frame() block for few, short streams: the reference's own call pattern
 * (qpsk.c:344-354: fread 512 int16, rx_frame()), where the five-kernel composition of qpsk_streams_rx_pcm (mixer ->
 * rrc_fir -> delay line -> timing scan -> Costas pipeline) is bound by its launches, not by its work.
 *
 * One workgroup per stream, two waves, everything in LDS:
 *   wave 0   the Costas loop + slicer over decimated_frame[0..N) = the PREVIOUS block's picks (qpsk.c:196-212, Q6):
 *            the hand-scheduled stream of the pipeline kernels (costas_asm.h, groups of 16 steps; the C++ step for the
 *            frame's first steps, a partial last group and the groups the stream hands back), which leaves the phase each
 *            step started from; then all 64 lanes turn phases + symbols into costas_frame[] and slicer decisions.
 *   wave 1   this block: PCM -> complex mix (qpsk.c:114-120: the carrier recurrence in lane 0, then all lanes scale and
 *            multiply) -> rrc_fir() with the stream's delay line (rrc_fir.c:17-30: the generated full-rate stream
 *            fir_full8_asm.h, one pass per 512 samples) -> the histogram timing estimate (qpsk.c:127-180: 8 lanes per
 *            component, lane q owning threshold (max / 8) q, as timing_hist8_kernel) or the fixed offset -> the picks that
 *            the NEXT call's loop consumes (qpsk.c:186-191) -> the delay line and carrier phase the block leaves behind.
 * The two waves share nothing but the read of decimated_frame[] at the start (one barrier).
 *
 * Inputs and outputs may live in pinned HOST memory mapped into the device's address space (qpsk_streams_rx_pcm_host hands
 * the kernel its staging buffer directly): 1 KB of PCM in, a few hundred bytes out -- no copy engine, one launch, one
 * synchronisation per block.
 *
 * Served (host-checked): frame_size <= 2048, histogram or fixed timing.  Bit-exact like the kernels it stands in for; the
 * tests compare both paths with the reference's recordings.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qpsk_device.h"
#include "costas_asm.h"
#include "fir_full8_asm.h"
#include "carrier.h"
#include "kernels.h"

namespace qpsk {

/* -DQPSK_SBLK_PROF (A/B builds, tools only): stream 0 prints where its two waves spend their cycles and the shader clock they ran at */
#ifdef QPSK_SBLK_PROF
#define SBLK_STAMP(k) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt[k])::"memory"); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pr[k])::"memory"); } while (0)
#else
#define SBLK_STAMP(k) do { } while (0)
#endif

namespace sblk {
/* a wave's results are out: release them to the host and count the wave */
__device__ __forceinline__ void wave_done(unsigned *done, int lane)
{
    if (!done) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      /* system scope: everything this wave stored is visible before the count */
    if (lane == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr int MAX_L = 2048;
constexpr int TILE = 512, R = 8, PADS = 2, WSLOTS = 808;
__device__ __host__ constexpr int slot_of(int p) { return p + PADS * (p / R); }
/* LDS layout (bytes), all 16-byte aligned */
__device__ __host__ constexpr int tiles_of(int L) { return (L + TILE - 1) / TILE; }
__device__ __host__ constexpr int xs_slots(int L) { return tiles_of(L) * TILE + 128; }      /* positions 0..125 history, 126.. samples, zero padded */
__device__ __host__ constexpr int ds_slots(int N) { return ((N + 15) / 16) * 16 + 4; }       /* the stream fetches a pair past its last group */

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
} // namespace sblk

template <bool INLINE>
__device__ __forceinline__ void stream_block_body(const StreamBlockArgs &a, const StreamBlockInline *inl)
{
    using namespace sblk;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int L = a.frame_size, N = a.nsym, C = a.cycles;
    float *taps = reinterpret_cast<float *>(smem);                                   /* 128 */
    float2 *win = reinterpret_cast<float2 *>(smem + 512);                           /* WSLOTS: the FIR stream's window image */
    float2 *xs = win + WSLOTS;                                                       /* xs_slots(L): history + mixed samples */
    float2 *ys = xs + xs_slots(L);                                                   /* tiles * 512: filtered block */
    float2 *ph = ys + tiles_of(L) * TILE;                                            /* L + 2: carrier phases (PCM input) */
    float2 *dsym = ph + ((L + 2 + 1) & ~1);                                          /* ds_slots(N): previous picks */
    float *zrec = reinterpret_cast<float *>(dsym + ds_slots(N));                    /* ds_slots(N): phase records */
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = blockIdx.x;                                                        /* the stream */
#ifdef QPSK_SBLK_PROF
    unsigned long long pt[8] = {0}, pr[8] = {0};
#endif
    SBLK_STAMP(0);

    if (wave == 0) {
        const float2 *dec = a.dec + (size_t)s * N;
        for (int i = lane; i < ds_slots(N); i += 64) dsym[i] = i < N ? dec[i] : make_float2(0.0f, 0.0f);
    } else {
        for (int i = lane; i < 128; i += 64) taps[i] = i < NTAPS ? a.taps[i] : 0.0f;
    }
    __syncthreads();      /* decimated_frame[] is in LDS: wave 1 may overwrite the global copy from here on */

    if (wave == 2) {      /* the spare wave (launched only with the shared carrier): workgroup 0's runs the carrier of the NEXT block */
        if (s == 0 && lane == 0) {
            __builtin_amdgcn_s_setprio(3);
            carrier_block(a.cstate, a.ctab_next, L, 0, L / 2);
        }
        return;
    }
    SBLK_STAMP(1);
    if (wave == 0) {
        /* ============================ Costas + slicer over the previous block's symbols (qpsk.c:196-212) ========= */
        const float *lin = a.loop_in ? (INLINE ? inl->loop : a.loop_in) : a.loop;
        float phs = lin[2 * s], fr = lin[2 * s + 1];
        const float al = a.gains[0], be = a.gains[1], fmin_ = a.min_freq, fmax_ = a.max_freq;
        const bool fast_clamp = fmin_ < 0.0f && fmax_ > 0.0f;
        bool over = false;
#ifdef QPSK_SBLK_EXPT_NO_LOOP
        if (false) {
#else
        if (lane == 0) {
#endif
            int j = 0;
            auto cstep = [&](int i) {
                float tx, ty; unsigned qq;
                zrec[i] = phs;
                if (fast_clamp) costas_step_t<true>(phs, fr, al, be, fmin_, fmax_, dsym[i], tx, ty, qq, over);
                else costas_step_t<false>(phs, fr, al, be, fmin_, fmax_, dsym[i], tx, ty, qq, over);
            };
            /* a LOADED phase may be -0 (the loop itself never produces one: x + y is -0 only if both are): then the first step
             * takes the form that is exact there too, and the stream starts at symbol 4 (it wants multiples of 4); otherwise
             * the stream runs from the first symbol on */
            if (__float_as_uint(phs) == 0x80000000u || !fast_clamp) {
                Loop s0 = {phs, fr};
                LoopGains lg = {al, be, fmin_, fmax_};
                zrec[0] = phs;
                costas_step<true>(s0, lg, dsym[0], over);
                phs = s0.phase; fr = s0.freq;
                for (j = 1; j < min(4, N); j++) cstep(j);
            }
            if (fast_clamp) {
                constexpr int AG = COSTAS_ASM_GROUP;
                unsigned long long ign = 0ull;      /* this lane excused from the stream's exact-zero test (costas_asm.h, `ign`) */
                while (N - j >= AG) {
                    unsigned da = lds_addr(dsym + j), za = lds_addr(zrec + j);
                    unsigned long long fl = 0ull;
                    bool ran = false;
                    const unsigned want = __builtin_amdgcn_readfirstlane((unsigned)(N - j) / AG);      /* wave-uniform (one lane is active) */
                    unsigned left = want;
                    if (__float_as_uint(fr) != 0x80000000u) {    /* a -0 frequency (loaded state only) stays with the C++ step */
                        left = costas_asm_run(phs, fr, da, za, want, al, be, fmin_, fmax_, fl, ign);
                        ran = true;
                    }
                    j += AG * (int)(want - left);
                    if (left != 0) {    /* the group the stream handed back (exact-zero detector input, double wrap) */
                        if (ran && (fl & ~ign) != 0ull && __float_as_uint(phs) != 0x80000000u) {
                            /* nothing but (+0, +0) in the whole groups ahead (a first block, a squelched input): the stream's step is
                             * the reference's for those -- it takes the group again without the test */
                            const uint4 *p4 = reinterpret_cast<const uint4 *>(dsym + j);
                            unsigned acc = 0;
                            for (int i = 0; i < AG * (int)left / 2; i++) {
                                const uint4 v = p4[i];
                                acc |= v.x | v.y | v.z | v.w;
                            }
                            if (acc == 0u) { ign = 1ull; continue; }
                        }
                        for (int i = 0; i < AG; i++, j++) cstep(j);
                    }
                }
            }
            for (; j < N; j++) cstep(j);
            a.loop[2 * s] = phs;
            a.loop[2 * s + 1] = fr;
            if (a.loop_out) { a.loop_out[2 * s] = phs; a.loop_out[2 * s + 1] = fr; }
            if (over) __hip_atomic_store(a.status, STATUS_PHASE_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else if (!loop_state_finite(phs, fr)) __hip_atomic_store(a.status, STATUS_NONFINITE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        wave_sync();
        SBLK_STAMP(2);
        /* records -> costas_frame[] (qpsk.c:197) and slicer decisions (qpsk.c:74-79), 64 symbols at a time */
        for (int i = lane; i < N; i += 64) {
            const float2 z = i == 0 ? derotate<true>(zrec[0], dsym[0]) : derotate<false>(zrec[i], dsym[i]);
            a.sym[(size_t)s * N + i] = (uint8_t)slicer(z);
            if (a.costas) a.costas[(size_t)s * N + i] = z;
        }
        SBLK_STAMP(3);
#ifdef QPSK_SBLK_PROF
        if (s == 0 && lane == 0)
            printf("wave 0: start -> barrier %llu cycles, loop %llu, flush %llu; %.2f us in all at %.0f MHz\n", pt[1] - pt[0], pt[2] - pt[1], pt[3] - pt[2],
                   (double)(pr[3] - pr[0]) * 0.01, (double)(pt[3] - pt[0]) / ((double)(pr[3] - pr[0]) * 0.01));
#endif
        wave_done(a.done, lane);
        return;
    }

    /* ================================== wave 1: this block ===================================================== */
    /* history: positions 0..125 = memory[1..126] (memory[0] is shifted out by the first step, rrc_fir.c:19) */
    const float2 *mem = a.memory + (size_t)s * NTAPS;
    for (int i = lane; i < HIST; i += 64) xs[i] = mem[i + 1];
    for (int i = HIST + L + lane; i < xs_slots(L); i += 64) xs[i] = make_float2(0.0f, 0.0f);
    if (a.pcm && a.ctab) {
        /* qpsk.c:114-120 with the streams' one carrier: the block's phases are in the table the call before left -- no recurrence here */
        const int16_t *pcm = (INLINE ? reinterpret_cast<const int16_t *>(inl->pcm) : a.pcm) + (size_t)s * L;
        for (int i = lane; i < L; i += 64) {
            const float v = (float)pcm[i] / 16384.0f;
            const float2 c = a.ctab[i];
            xs[HIST + i] = make_float2(c.x * v, c.y * v);      /* qpsk.c:117 */
        }
    } else if (a.pcm) {
        /* qpsk.c:114-120.  The carrier state is fetched and WAITED FOR before the PCM loads are issued (loads return in order: a
         * wait for a later one would wait for the PCM too, which may cross PCIe); the carrier recurrence then runs in their shadow */
        float2 p = make_float2(a.mixer[4 * s], a.mixer[4 * s + 1]);
        float rr = a.mixer[4 * s + 2], ri = a.mixer[4 * s + 3];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(p.x), "+v"(p.y), "+v"(rr), "+v"(ri)::"memory");
        const int16_t *pcm = (INLINE ? reinterpret_cast<const int16_t *>(inl->pcm) : a.pcm) + (size_t)s * L;
        /* 8 samples per lane and load: a 512-sample block is ONE 16-byte load per lane -- one trip over PCIe when the block sits in
         * pinned host memory.  All loads are issued before anything waits (clamped addresses, no per-load branch). */
        constexpr int NV = MAX_L / 512;
        const bool vec = (L & 7) == 0 && (reinterpret_cast<uintptr_t>(pcm) & 15) == 0;
        uint4 pv[NV];
        if (vec) {
            const uint4 *p4 = reinterpret_cast<const uint4 *>(pcm);
#pragma unroll
            for (int k = 0; k < NV; k++) pv[k] = p4[min(k * 64 + lane, L / 8 - 1)];
        }
        SBLK_STAMP(6);
        if (lane == 0) {
            const float nri = -ri;
            /* fbb_rx_phase *= fbb_rx_rect, qpsk.c:115: four products, two sums, unfused.  ONE lane runs this chain, so what counts
             * is the instruction stream: six 4-byte VOP2 instructions per sample (a lone wave issues ~1.56 instruction bytes per
             * cycle, profiles/r03_ubench_fetch.txt; the compiler's packed form, three 8-byte instructions with a dependent pair,
             * ran 50 cycles per sample [measured, profiles/r04_stream_block.txt]) */
            auto step = [&]() {
                const float t0 = p.x * rr, t1 = p.y * nri, t2 = p.y * rr, t3 = p.x * ri;
                p = make_float2(t0 + t1, t2 + t3);
            };
            /* sixteen samples per block of straight-line code: the phase alternates between two register pairs that ARE the payload
             * of the 16-byte store (no moves), the store's address is a VGPR + an immediate: 6.5 instructions per sample */
            int i = 0;
            if (L >= 16) {
                unsigned addr = lds_addr(ph);
                unsigned nblk = __builtin_amdgcn_readfirstlane((unsigned)L / 16u);      /* wave-uniform (one lane is active) */
                i = 16 * (int)nblk;
                /* v[40:43] = (A, B), v44..v47 products: fixed registers owned by the block.  The products of the real part come first
             * (the imaginary part is the instruction just before them): no instruction directly follows its producer */
#define SBLK_MIX2(OFF)                                                                                   \
                "v_mul_f32_e32 v44, %[rr], v42\n\tv_mul_f32_e32 v47, %[ri], v42\n\t"                        \
                "v_mul_f32_e32 v45, %[nri], v43\n\tv_mul_f32_e32 v46, %[rr], v43\n\t"                       \
                "v_add_f32_e32 v40, v44, v45\n\tv_add_f32_e32 v41, v46, v47\n\t"                            \
                "v_mul_f32_e32 v44, %[rr], v40\n\tv_mul_f32_e32 v47, %[ri], v40\n\t"                        \
                "v_mul_f32_e32 v45, %[nri], v41\n\tv_mul_f32_e32 v46, %[rr], v41\n\t"                       \
                "v_add_f32_e32 v42, v44, v45\n\tv_add_f32_e32 v43, v46, v47\n\t"                            \
                "ds_write_b128 %[ad], v[40:43] offset:" #OFF "\n\t"
                asm volatile("v_mov_b32_e32 v42, %[px]\n\tv_mov_b32_e32 v43, %[py]\n"
                             "1:\n\t"
                             SBLK_MIX2(0) SBLK_MIX2(16) SBLK_MIX2(32) SBLK_MIX2(48) SBLK_MIX2(64) SBLK_MIX2(80) SBLK_MIX2(96) SBLK_MIX2(112)
                             "v_add_u32_e32 %[ad], 0x80, %[ad]\n\t"
                             "s_sub_u32 %[n], %[n], 1\n\t"
                             "s_cmp_lg_u32 %[n], 0\n\t"
                             "s_cbranch_scc1 1b\n\t"
                             "v_mov_b32_e32 %[px], v42\n\tv_mov_b32_e32 %[py], v43"
                             : [px] "+v"(p.x), [py] "+v"(p.y), [ad] "+v"(addr), [n] "+s"(nblk)
                             : [rr] "v"(rr), [ri] "v"(ri), [nri] "v"(nri)
                             : "memory", "scc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
#undef SBLK_MIX2
            }
            for (; i < L; i++) { step(); ph[i] = p; }
            SBLK_STAMP(7);
            const float mag = (float)sqrt((double)p.x * (double)p.x + (double)p.y * (double)p.y);   /* qpsk.c:120 */
            a.mixer[4 * s] = p.x / mag;
            a.mixer[4 * s + 1] = p.y / mag;
        }
        wave_sync();
        if (vec) {
#pragma unroll
            for (int k = 0; k < NV; k++) {
                const int g = k * 64 + lane;      /* samples 8 g .. 8 g + 7 */
                if (8 * g < L) {
                    const unsigned w[4] = {pv[k].x, pv[k].y, pv[k].z, pv[k].w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float v0 = (float)(int16_t)(w[e] & 0xffffu) / 16384.0f, v1 = (float)(int16_t)(w[e] >> 16) / 16384.0f;
                        const float4 c = *reinterpret_cast<const float4 *>(ph + 8 * g + 2 * e);
                        *reinterpret_cast<float4 *>(xs + HIST + 8 * g + 2 * e) = make_float4(c.x * v0, c.y * v0, c.z * v1, c.w * v1);   /* qpsk.c:117 */
                    }
                }
            }
        } else {
            for (int i = lane; i < L; i += 64) {
                const float v = (float)pcm[i] / 16384.0f;
                const float2 c = ph[i];
                xs[HIST + i] = make_float2(c.x * v, c.y * v);
            }
        }
    } else {
        const float2 *cx = a.cplx + (size_t)s * L;
        for (int i = lane; i < L; i += 64) xs[HIST + i] = cx[i];
    }
    wave_sync();
    SBLK_STAMP(2);
    /* the delay line the block leaves behind: the last 127 of (memory ++ x) (rrc_fir.c:19-20 applied L times) */
    float2 *memw = a.memory + (size_t)s * NTAPS;
    for (int i = lane; i < NTAPS; i += 64) {
        const int src = i + L - 1;      /* index into xs: (memory ++ x)[j + 1] = xs[j] */
        memw[i] = src >= 0 ? xs[src] : make_float2(0.0f, 0.0f);      /* src < 0 cannot happen (L >= 1) */
    }
    /* rrc_fir(): one pass of the stream per 512 outputs (window position p of a tile = xs[512 t + p]) */
    const unsigned rd_addr = lds_addr(win + (R + PADS) * lane), tap_addr = lds_addr(taps);
    for (int t = 0; t < tiles_of(L); t++) {
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int p = 2 * (lane + 64 * j);
            if (j < 4 || lane < 63) *reinterpret_cast<float4 *>(win + slot_of(p)) = *reinterpret_cast<const float4 *>(xs + TILE * t + p);
        }
        wave_sync();
        v2f a0, a1, a2, a3, a4, a5, a6, a7;
        fir_full8_asm(rd_addr, tap_addr, a0, a1, a2, a3, a4, a5, a6, a7);
        float4 *yl = reinterpret_cast<float4 *>(ys + TILE * t + R * lane);
        const float2 y0 = fir_gain(make_float2(a0.x, a0.y)), y1 = fir_gain(make_float2(a1.x, a1.y)),
                     y2 = fir_gain(make_float2(a2.x, a2.y)), y3 = fir_gain(make_float2(a3.x, a3.y)),
                     y4 = fir_gain(make_float2(a4.x, a4.y)), y5 = fir_gain(make_float2(a5.x, a5.y)),
                     y6 = fir_gain(make_float2(a6.x, a6.y)), y7 = fir_gain(make_float2(a7.x, a7.y));
        yl[0] = make_float4(y0.x, y0.y, y1.x, y1.y);
        yl[1] = make_float4(y2.x, y2.y, y3.x, y3.y);
        yl[2] = make_float4(y4.x, y4.y, y5.x, y5.y);
        yl[3] = make_float4(y6.x, y6.y, y7.x, y7.y);
        wave_sync();
    }
    SBLK_STAMP(3);
    if (a.filtered) {      /* the filtered block itself (tests; qpsk.c:125's input_frame[] after the call) */
        float2 *fo = a.filtered + (size_t)s * L;
        for (int i = lane; i < L; i += 64) fo[i] = ys[i];
    }
    /* timing (qpsk.c:127-180) */
    int best = a.fixed_index;
    if (a.hist_timing) {
        /* lane = 8 comp + q (16 lanes): every lane of a component runs the same average / max chain on the same samples (LDS
         * broadcast reads), lane q counts the symbols whose average is NOT <= (max / 8) q; the thresholds are nondecreasing in
         * q, so hist[q] = cum[q - 1] - cum[q] with cum[0] = the number of symbols (timing_hist8_kernel's argument) */
        const int comp = (lane >> 3) & 1, q = lane & 7;
        const float qf = (float)q, fc = (float)C;
        const float *row = reinterpret_cast<const float *>(ys) + comp;
        float av = 0.0f, mx = 0.0f;
        int cum = 0;
        /* a division by a power of two is a multiplication by its (exact) reciprocal, bit for bit; the divider's ~15
         * instructions per use would otherwise sit on this chain twice per symbol */
        const bool pow2 = (C & (C - 1)) == 0;
        const float rc = 1.0f / fc;
        if (lane < 16) {
            if (pow2 && C == 4) {
                /* the lane's component of a symbol's four samples: two two-address LDS reads, a symbol ahead of their use (the last
                 * fetch lands in the rows behind ys[]: unused) */
                float nx[8];
#pragma unroll
                for (int u = 0; u < 8; u++) nx[u] = row[2 * u];
                auto symbol = [&](float v0, float v1, float v2, float v3) {
                    av += fabsf(v0); av += fabsf(v1); av += fabsf(v2); av += fabsf(v3);      /* qpsk.c:131-136 */
                    av *= 0.25f;                                                              /* qpsk.c:137-138 */
                    mx = __builtin_fmaxf(mx, av);     /* "if (av > max) max = av" (qpsk.c:140-145): the same value, NaN included */
                    const float th = (mx * 0.125f) * qf;                                      /* qpsk.c:147-165 */
                    cum += (av <= th) ? 0 : 1;
                };
                int sy = 0;
                for (; sy + 2 <= N; sy += 2) {      /* two symbols per round, their samples fetched a round ahead */
                    float cur[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) cur[u] = nx[u];
#pragma unroll
                    for (int u = 0; u < 8; u++) nx[u] = row[8 * sy + 16 + 2 * u];
                    symbol(cur[0], cur[1], cur[2], cur[3]);
                    symbol(cur[4], cur[5], cur[6], cur[7]);
                }
                if (sy < N) symbol(nx[0], nx[1], nx[2], nx[3]);
            } else {
                for (int sy = 0; sy < N; sy++) {
                    for (int u = 0; u < C; u++) av += fabsf(row[2 * (sy * C + u)]);
                    av = pow2 ? av * rc : av / fc;
                    if (av > mx) mx = av;
                    const float th = (mx * 0.125f) * qf;      /* max / 8.0f */
                    cum += (av <= th) ? 0 : 1;
                }
            }
        }
        if (q == 0) cum = N;
        int h = __shfl_up(cum, 1) - cum;
        if (q == 0) h = 0;
        h += __shfl_xor(h, 8);
        int hmax = 0;
        best = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int hk = __shfl(h, k);
            if (hk > hmax) { hmax = hk; best = k; }      /* first argmax (qpsk.c:173-180), all-zero block -> ... */
        }
    }
    SBLK_STAMP(4);
    if (lane == 0 && a.index) a.index[s] = best;
    /* decimated_frame[i] <- input_frame[i*CYCLES + index] for the next call (qpsk.c:186-191; a pick past the block is 0, Q5) */
    float2 *decw = a.dec + (size_t)s * N;
    for (int i = lane; i < N; i += 64) {
        const int src = i * C + best;
        decw[i] = src < L ? ys[src] : make_float2(0.0f, 0.0f);
    }
    wave_done(a.done, lane);
    SBLK_STAMP(5);
#ifdef QPSK_SBLK_PROF
    if (s == 0 && lane == 0)
        printf("wave 1: (PCM issue at %llu, recurrence done at %llu) start -> barrier %llu cycles, history + mixer %llu, delay line + filter %llu, timing %llu, picks %llu; %.2f us in all at %.0f MHz\n",
               pt[6] - pt[0], pt[7] - pt[0], pt[1] - pt[0], pt[2] - pt[1], pt[3] - pt[2], pt[4] - pt[3], pt[5] - pt[4], (double)(pr[5] - pr[0]) * 0.01,
               (double)(pt[5] - pt[0]) / ((double)(pr[5] - pr[0]) * 0.01));
#endif
}

__global__ void __launch_bounds__(192)
stream_block_kernel(StreamBlockArgs a)
{
    stream_block_body<false>(a, nullptr);
}

/* the same with the block inside the kernel arguments (kernels.h, StreamBlockInline) */
__global__ void __launch_bounds__(192)
stream_block_inline_kernel(StreamBlockArgs a, StreamBlockInline)
{
    /* read where the dispatch left it (taking the parameter's address would copy its 2 KB to scratch): the second argument
     * follows the first at its own alignment */
    constexpr size_t OFF = (sizeof(StreamBlockArgs) + alignof(StreamBlockInline) - 1) / alignof(StreamBlockInline) * alignof(StreamBlockInline);
#if defined(__HIP_DEVICE_COMPILE__)      /* (the host pass of this file only needs the kernel's signature) */
    const unsigned char *ka = (const unsigned char *)__builtin_amdgcn_kernarg_segment_ptr();
    stream_block_body<true>(a, reinterpret_cast<const StreamBlockInline *>(ka + OFF));
#endif
}

size_t stream_block_lds_bytes(int L, int N)
{
    using namespace sblk;
    return 512 + sizeof(float2) * ((size_t)WSLOTS + xs_slots(L) + (size_t)tiles_of(L) * TILE + ((L + 2 + 1) & ~1) + ds_slots(N)) +
           sizeof(float) * (size_t)ds_slots(N);
}

int stream_block_max_frame(void) { return sblk::MAX_L; }

int prepare_stream_block(void)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(stream_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       MAX_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(stream_block_inline_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    MAX_LDS_BYTES);
}

int launch_stream_block(const StreamBlockArgs &a, int nstreams, hipStream_t s, const StreamBlockInline *inl)
{
    if (a.frame_size < 1 || a.frame_size > sblk::MAX_L || a.nsym < 1 || stream_block_lds_bytes(a.frame_size, a.nsym) > (size_t)MAX_LDS_BYTES ||
        (a.ctab != nullptr) != (a.ctab_next != nullptr) || (a.ctab != nullptr) != (a.cstate != nullptr) || (a.ctab && (a.frame_size & 1)))
        return (int)hipErrorInvalidValue;
    const dim3 threads(a.ctab ? 192 : 128);
    if (inl) {
        if (nstreams > StreamBlockInline::MAX_STREAMS || (long long)nstreams * a.frame_size > StreamBlockInline::MAX_SAMPLES) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(stream_block_inline_kernel, dim3(nstreams), threads, stream_block_lds_bytes(a.frame_size, a.nsym), s, a, *inl);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(stream_block_kernel, dim3(nstreams), threads, stream_block_lds_bytes(a.frame_size, a.nsym), s, a);
    return (int)hipGetLastError();
}

} // namespace qpsk

