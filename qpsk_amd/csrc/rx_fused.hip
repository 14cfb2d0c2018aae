/*
 * rx_fused.hip -- the hot kernel: decimating RRC FIR + Costas loop + slicer in
 * one pass over the input (reference rrc_fir.c:17-30 evaluated only at the
 * samples qpsk.c:190 keeps, then qpsk.c:196-212 / costas_loop.c:44-74).
 *
 * Shape of the problem on MI355X ("rx_fused_pipe"):
 *   - the FIR is throughput work: 127 unfused mul+add pairs per symbol, any
 *     number of symbols in parallel;
 *   - the Costas loop is a strict recurrence per frame (phase -> sincos ->
 *     rotate -> detector -> phase): one wave can issue one VALU instruction
 *     per ~5 cycles, dependent ones every ~8, so a frame's 2048 symbols cost
 *     2048 x (instructions per step) x 5 cycles however many CUs idle.
 * So each workgroup is a producer/consumer pipeline in LDS:
 *   wave 0           the Costas recurrence (costas_asm.h), one lane per (frame, loop); it never waits
 *                    for HBM and leaves one 4-byte record per step in an LDS ring: the phase the step
 *                    started from (four steps per LDS write);
 *   FIR waves        each owns its frames outright (loads, history, filter, and the flush of their
 *                    records: sin/cos of the recorded phase again, de-rotation of the symbol -- still in the
 *                    symbol ring -- to costas_frame[], slicer, coalesced stores), so FIR waves never
 *                    synchronise with each other, only with wave 0 through two monotonic counters in LDS
 *                    (bounded spins; a timeout is reported through *status);
 *   spare waves      hardware waves that would land on wave 0's SIMD exit at once.
 * A full workgroup (16 frames) has three FIR waves of 4 frames (4 symbols per lane) and two of 2 frames
 * (2 symbols per lane): 1, 1.5, 1.5 filter units on the three SIMDs the serial wave leaves free.
 *
 * FIR wave layout of rx_fused_pipe_kernel (Geom<16,4,4,1>; rx_pipe2_kernel further down lays the same pipeline
 * out for 32 frames per workgroup): lane = (frame f of 4) x (q of 16); per chunk of S = 64
 * symbols the lane produces the R = 4 consecutive symbols 4q..4q+3 of its
 * frame with a sliding window: one 8-byte LDS read feeds up to 4 of the 508
 * multiply-adds, taps come from LDS broadcast reads (R + 1 groups of 8 live at
 * a time), each symbol's taps are summed 0..126 in one fp32 accumulator
 * (bit-exactness).
 * LDS image of a frame's window: position p (0 = the oldest sample the
 * chunk needs, i.e. sample chunk_start + index - 126) at float2 slot
 * p + p/32; lanes of one frame are 32 samples apart -> 33 slots -> the 16 x 2
 * lanes of a half wave hit 32 distinct even banks.  The per-frame decimation
 * index is absorbed when the window is WRITTEN (two per-lane base addresses),
 * so every FIR read is "one VGPR base + immediate".
 *
 * HBM traffic: every input sample is read exactly once (16-byte coalesced
 * loads, prefetched one chunk ahead into registers); 1 byte per symbol out.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qpsk_device.h"
#include "costas_asm.h"
#include "costas_asm_lo.h"     /* the same streams with the VGPR block at v84..v127 (rx_hist_kernel: four waves on a SIMD) */
#include "fir_full8s_asm.h"    /* rx_hist_kernel's full-rate filter */
#include "fir_r2_asm.h"
#include "fir_r4_asm.h"
#include "fir_lean_asm.h"      /* WS_LEAN_SLOTS, the streams */
#include "fir_full8_asm.h"     /* the in-launch FFT timing estimate of rx_fused_pipe_kernel */
#include "timing_fft_wave.h"
#include "carrier.h"
#ifdef QPSK_PIPE_PROFILE
#include "fir_lean_prof_asm.h"
#endif
#include "kernels.h"

#ifndef QPSK_PIPE1_ASM
#define QPSK_PIPE1_ASM 1     /* rx_fused_pipe_kernel: FIR steps as the generated streams; 0: the compiler's pinned step (A/B builds) */
#endif

namespace qpsk {

/* register budgets of the generated streams (tools/gen_fir_asm.py, tools/gen_lean_asm.py): kernels with three waves per
 * SIMD have 168 VGPRs; the serial wave's stream (costas_asm.h) owns v100..v143 of its own wave only */
static_assert(FIR_R2_ASM_END_VGPR <= 168, "rx_pipe2_kernel: three FIR waves per SIMD");
static_assert(FIR_R4_ASM_END_VGPR <= 256, "rx_fused_pipe_kernel: two waves per SIMD");

namespace pipe {

constexpr int C = 8;            /* CYCLES this instantiation is built for */
constexpr int DR = 2;            /* depth of the symbol rings in chunks */
constexpr int MAX_WAVES = 16;    /* ready[] counters: FIR waves of rx_fused_pipe_kernel / two-frame units of rx_pipe2_kernel */
constexpr int SPIN_LIMIT = 1 << 24;

/* Result-changing ablation knobs (FusedArgs::dbg bit 0: skip the filter arithmetic, bit 1: skip the recurrence) exist
 * only in the measurement build (make -C qpsk_amd/csrc profile); the product library computes the reference's
 * result whatever the environment says. */
#ifdef QPSK_PIPE_PROFILE
#define QPSK_ABLATE(a, bit) (((a).dbg & (bit)) != 0)
#else
#define QPSK_ABLATE(a, bit) false
#endif

/*
 * Geometry of a workgroup.  The LDS (window + rings per frame) decides how many frames it holds:
 *   Geom<16, 4>  "narrow", the one the host uses: chunks of 64 symbols, 7.3 KB of LDS per frame, 16 frames per
 *                workgroup; the serial wave has a SIMD of its own (spare waves retire at once) and three SIMDs
 *                filter.  Lane mappings of its FIR waves: (frame of 4) x (q of 16) with 4 symbols per lane, or
 *                (frame of 2) x (q of 32) with 2 symbols per lane -- both in one workgroup when it is full (see
 *                the kernel).  Batches above 16 frames per CU go to rx_pipe2_kernel (api.cpp).
 */
template <int QL_, int R_, int MAX_NF_, int SPARE_, bool PINNED_>
struct Geom {
    static constexpr bool PINNED = PINNED_;  /* FIR step with the product/add order pinned by hand (see the FIR loop) */
    static constexpr int QL = QL_;           /* lanes per frame */
    static constexpr int R = R_;             /* symbols per FIR lane per chunk */
    static constexpr int MAX_NF = MAX_NF_;   /* FIR waves per workgroup */
    static constexpr int SPARE = SPARE_;     /* 1: no FIR wave on the serial wave's SIMD (waves 4, 8 are launched and retire at once) */
    static constexpr int FWV = 64 / QL;      /* frames per FIR wave */
    static constexpr int S = R * QL;         /* symbols per chunk */
    static constexpr int CH = S * C;         /* samples per chunk per frame */
    static constexpr int TSTEPS = NTAPS + C * (R - 1);   /* window positions a lane sweeps per chunk */
    static constexpr int PAD = R * C;        /* lanes of a frame are PAD positions apart: position p lives at slot p + p/PAD */
    static constexpr int WL = CH + 128;      /* window positions kept per frame */
    /* padded image (two pad slots per PAD positions, see the FIR waves): 716 slots for the two-symbol lanes; frame
     * stride = 0 (mod 32) slots = 0 (mod 64) banks: the four frames whose lanes share a ds_read_b128 pass start in
     * bank quads 4q, which the 16 lanes of a pass then cover exactly once */
    static constexpr int WSLOTS = ((WL + 2 * (WL / (2 * C)) + 31) / 32) * 32;
    static constexpr int DSTRIDE = DR * S + 2;   /* float2 slots per frame row: 16-byte aligned rows (the Costas wave reads two
                                                    symbols per ds_read_b128), 4 dwords (mod 64) apart: lanes hit different bank quads */
    static constexpr int ZSTRIDE = DR * S + 4;   /* 4-byte records (the phase a step started from) per Costas row: 16-byte
                                                    aligned rows (four records per write), 4 dwords (mod 64) apart: 16 lanes x 16 bytes on 64 banks */
    static constexpr int MAX_THREADS = SPARE ? 512 : 64 * (MAX_NF + 1);   /* with spares: 3 + 2 FIR waves, 3 retiring ones */
    static_assert(QL == 16 && 64 % QL == 0 && 128 % PAD == 0, "slot arithmetic assumes 16 lanes per frame");
};
/* lane mapping of one FIR wave: QL lanes per frame, R symbols per lane */
template <int QL_, int R_>
struct WaveMap {
    static constexpr int QL = QL_, R = R_;
};
using GeomNarrow = Geom<16, 4, 4, 1, true>;
/* (Round 1 also had a "wide" geometry, Geom<16, 2, 8, 0>: 32 frames per workgroup in 32-symbol chunks, per-frame
 * windows, FIR waves beside the serial wave -- 0.365-0.379 ms at 8192 frames.  rx_pipe2_kernel below replaces it.) */

#define QPSK_GEOM_CONSTANTS(GM)                                                                          \
    constexpr int QL = GM::QL, R = GM::R, FWV = GM::FWV, S = GM::S, CH = GM::CH, WSLOTS = GM::WSLOTS,    \
                  DSTRIDE = GM::DSTRIDE, ZSTRIDE = GM::ZSTRIDE, TSTEPS = GM::TSTEPS, PAD = GM::PAD,      \
                  MAX_NF = GM::MAX_NF;                                                                   \
    (void)QL; (void)R; (void)FWV; (void)S; (void)CH; (void)WSLOTS; (void)DSTRIDE; (void)ZSTRIDE;         \
    (void)TSTEPS; (void)PAD; (void)MAX_NF

#ifdef QPSK_PIPE_PROFILE
/* measurement build, dbg bit 23: wall-clock stamps (100 MHz constant clock) of one rx_lean_kernel launch, 16 per workgroup: entry, the serial
 * wave in costas_wave / at its first step / behind its last step / at its end, then the end of hardware wave w in slot 5 + w
 * (tools/lean_timeline.py reads them through qpsk_prof_timeline) */
__device__ unsigned long long g_timeline[1024 * 48];
#define QPSK_TL_STAMP(SLOT) do { if ((a.dbg & 8388608) && lane == 0 && blockIdx.x < 1024) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); g_timeline[48 * blockIdx.x + (SLOT)] = t_; } } while (0)
#endif
struct Smem {
    float taps[128];
    int ready[MAX_WAVES];     /* chunks produced, per FIR wave */
    int consumed;             /* chunks consumed by the Costas wave */
    int abort_flag;
    int pad_[2];
    int est_index[16];        /* rx_fused_pipe_kernel with the FFT timing estimate in the launch: the workgroup's decimation offsets */
};

__device__ __forceinline__ int ld_acquire(const int *p)
{
    int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return v;
}

__device__ __forceinline__ void st_release(int *p, int v)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

/* the context's status word lives in pinned host memory (api.cpp): the host reads it after any synchronisation */
__device__ __forceinline__ void report_status(int *status, int what)
{
    __hip_atomic_store(status, what, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* bounded wait until *p >= target; false if the workgroup gave up */
__device__ __forceinline__ bool wait_ge(int *p, int target, int *abort_flag)
{
    int spins = 0;
    while (ld_acquire(p) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return false;
        }
    }
    return true;
}

} // namespace pipe

using namespace pipe;

/*
 * The serial wave: one lane per (frame, loop).  It waits for chunk c of its frame in the symbol ring (counter
 * ready[]), advances the loop over the chunk's symbols and leaves one 4-byte record per symbol in the record
 * ring (the phase the step started from, four steps per write), then publishes consumed = c + 1.  Shared by rx_fused_pipe_kernel (ring fed by FIR waves) and
 * costas_pipe_kernel (ring fed from already decimated symbols in global memory).
 */
/* which build of the instruction streams the serial wave runs: costas_asm.h's (VGPRs 100..143) or costas_asm_lo.h's (84..127) */
struct StreamHi {
    template <class... A> static __device__ __forceinline__ void ring(A &&...a) { costas_asm_run_ring(a...); }
    template <class... A> static __device__ __forceinline__ void ring_pair(A &&...a) { costas_asm_run_ring_pair(a...); }
    template <class... A> static __device__ __forceinline__ unsigned run(A &&...a) { return costas_asm_run(a...); }
};
struct StreamLo {
    template <class... A> static __device__ __forceinline__ void ring(A &&...a) { costas_asm_run_ring_lo(a...); }
    template <class... A> static __device__ __forceinline__ void ring_pair(A &&...a) { costas_asm_run_ring_pair_lo(a...); }
    template <class... A> static __device__ __forceinline__ unsigned run(A &&...a) { return costas_asm_run_lo(a...); }
};

template <class GM, class ST = StreamHi, class SM>   /* SM: Smem, or rx_lean_kernel's control block (ready[], consumed, abort_flag) */
__device__ __forceinline__ void costas_wave(const FusedArgs &a, SM *sm, const float2 *dring, float *zring, int G,
                                            int f0, int lane, int nchunks, int *status)
{
    QPSK_GEOM_CONSTANTS(GM);
    const int nbw = a.nbw, N = a.nsym;
    /* =============================== Costas + slicer wave ===================================== */
    /* Alone on its SIMD the wave takes priority 3.  rx_pipe2_kernel may place a FIR wave beside it (a.share_simd0):
     * the recurrence has slack there (32 frames per workgroup: the filter is the limit) and a FIR wave that only
     * gets the serial wave's leftovers becomes the slowest of the workgroup, so the priorities are the other way
     * round: this wave 1, the FIR wave beside it 3 [measured: 0.335 -> 0.294 ms at 8192 frames] */
    if (a.share_simd0 != ((a.dbg & 512) != 0)) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(3);
    /* a.lean_pair (rx_lean_kernel, one loop per frame, at most 32 frames per workgroup): lanes 2g and 2g + 1 both carry frame g's loop
     * and share the two polynomial chains of its step between them (costas_asm.h, costas_asm_run_ring_pair); wherever the instruction
     * stream is not running the two lanes simply do the same thing */
    const bool pair = a.lean_pair != 0;
    const bool odd = pair && (lane & 1);
    const int g = pair ? lane >> 1 : lane / nbw, b = pair ? 0 : lane - g * nbw;
    const bool inwg = pair ? lane < 2 * G : lane < G * nbw;
    const bool active = inwg && f0 + g < a.nframes;
    Loop st = {0.0f, 0.0f};
    LoopGains lg = {0.0f, 0.0f, a.min_freq, a.max_freq};
    if (active) {
        lg.alpha = a.gains[2 * b];
        lg.beta = a.gains[2 * b + 1];
        if (a.state_in) {
            st.phase = a.state_in[2 * ((size_t)(f0 + g) * nbw + b)];
            st.freq = a.state_in[2 * ((size_t)(f0 + g) * nbw + b) + 1];
        }
    }
    /* the FIR wave that feeds this lane: 4 frames each; in the mixed workgroup frames 12..15 come two per wave */
    const int gl = inwg ? g : 0;
    const int gw = a.mixed == 2 ? gl / 2 : a.mixed == 1 && gl >= 12 ? 3 + (gl - 12) / 2 : gl / FWV;
    const float2 *dl = dring + (size_t)gl * DSTRIDE;
    float *zl = zring + (size_t)(pair ? gl : lane) * ZSTRIDE;   /* this lane's records: the phase each symbol's step started from */
    /* one median-of-3 instead of two compare/select pairs when the clamp is the usual min < 0 < max */
    const bool fast_clamp = a.min_freq < 0.0f && a.max_freq > 0.0f;
    float ph = st.phase, fr = st.freq;
    bool over = false;   /* a phase beyond the bounded 2 pi wrap (qpsk_device.h, phase_wrap) */
    /* Symbols that are exactly (+0, +0) -- a stream's first block, a squelched input, an all-zero frame -- trip the instruction
     * stream's exact-zero test in every group although its step is the reference's for them (costas_asm.h, `ign`).  When a group
     * is abandoned, the lanes that tripped the test look at the symbols they have ahead in the chunk: nothing but +0.0 there and no
     * -0 in the loop state excuses them from the test for that stretch, and the stream takes the group again.  Without this such a
     * lane sent its whole workgroup through the C++ step, six times the cost (a squelched stream among 16; config 2 with one silent
     * frame: the launch waits for that workgroup). */
    auto zero_run = [&](int from, int to) {       /* ring slots [from, to) of this lane's row, both even: 16-byte reads */
        const uint4 *p = reinterpret_cast<const uint4 *>(dl + from);
        unsigned acc = 0;
        for (int i = 0; i < (to - from) / 2; i++) {
            const uint4 v = p[i];
            acc |= v.x | v.y | v.z | v.w;
        }
        return acc == 0u;
    };
    auto excusable = [&](unsigned long long lanes, int from, int to) {       /* -> the lanes of `lanes` with such a stretch ahead */
        const bool z = ((lanes >> lane) & 1ull) != 0ull && __float_as_uint(ph) != 0x80000000u && __float_as_uint(fr) != 0x80000000u &&
                       zero_run(from, to);
        return (unsigned long long)__ballot(z);
    };
    const float al = lg.alpha, be = lg.beta, fmin_ = a.min_freq, fmax_ = a.max_freq;
    bool ok = true;
#ifdef QPSK_PIPE_PROFILE
    const bool cprof = (a.dbg & 32) && blockIdx.x == 0;
    unsigned long long cw = 0, cs = 0, ct = 0;
    auto ctick = [&](unsigned long long &acc) {
        if (cprof) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            acc += t - ct;
            ct = t;
        }
    };
    unsigned long long dummy = 0;
    ctick(dummy);
#endif
    /* full chunks go through the stream that runs across the ring hand-overs (costas_asm.h, costas_asm_run_ring); a
     * first chunk that starts from a loaded phase of -0, a last partial one and the variants below stay chunk by chunk */
#ifdef QPSK_PIPE_PROFILE
    /* dbg bit 12 (4096): keep the ring stream under dbg 32 and time it (cycles inside the stream / waiting outside it) */
    /* dbg bit 23 (8388608): every workgroup prints the wall-clock stamps of its waves (the launch's timeline across the chip) */
    const bool timeline = (a.dbg & 8388608) != 0;
    const bool ring_prof = (a.dbg & 4096) != 0 && (blockIdx.x == 0 || blockIdx.x == 77 || timeline);
    unsigned long long rp_in = 0, rp_out = 0, rp_t = 0, rp_calls = 0, rp_rt0 = 0, rp_rt1 = 0, rp_rt2 = 0;
    auto rp_real = [&]() {      /* the 100 MHz constant clock: wall time inside the launch */
        unsigned long long t = 0;
        if (ring_prof) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    };
    rp_rt0 = rp_real();
    auto rp_tick = [&](unsigned long long &acc) {
        if (ring_prof) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (rp_t) acc += t - rp_t;
            rp_t = t;
        }
    };
    const bool ring_stream = fast_clamp && !(a.dbg & (2 | 8 | 16 | ((a.dbg & 4096) ? 0 : 32))) && S == 4 * COSTAS_ASM_GROUP && DR == 2;
#else
    const bool ring_stream = fast_clamp && !(a.dbg & (8 | 16)) && S == 4 * COSTAS_ASM_GROUP && DR == 2;
#endif
    /* the frame's first chunk goes through the stream too unless some loop starts from a LOADED phase of -0 (the
     * stream's sin/cos form is exact everywhere else; a -0 frequency is kept away from it group by group below) */
    const int first_ring_chunk = __any(active && __float_as_uint(ph) == 0x80000000u) ? 1 : 0;
    for (int c = 0; c < nchunks && ok; c++) {
        if (ring_stream && c >= first_ring_chunk && (c + 1) * S <= N) {
            const int cfull = N / S;      /* chunks c .. cfull - 1 are whole */
            if (active) {
                constexpr int AG = COSTAS_ASM_GROUP;
                unsigned k = 4u * (unsigned)c;
                const unsigned kend = 4u * (unsigned)cfull;
                const unsigned d_base = lds_addr(dl), z_base = lds_addr(zl);
                const unsigned ready_addr = lds_addr(&sm->ready[gw]), consumed_addr = lds_addr(&sm->consumed);
                unsigned long long ign = 0ull;      /* lanes excused from the zero test up to the end of the chunk they are in */
                while (k < kend) {
                    if ((k & 3u) == 0u) {     /* entering a chunk: the stream left because it was not there yet (or this is the start) */
                        ok = wait_ge(&sm->ready[gw], (int)(k >> 2) + 1, &sm->abort_flag);
                        if (!__all(ok)) { ok = false; break; }
                        if (ign) {            /* still nothing but zeros?  (the stream stops at every chunk end while some lane is excused) */
                            const int at = (int)(k & 7u) * AG;
                            ign = excusable(ign, at, at + S);
                        }
                    }
                    unsigned long long fl = 0ull;
                    bool ran = false;
                    if (!__any(__float_as_uint(fr) == 0x80000000u)) {
                        unsigned ks = __builtin_amdgcn_readfirstlane(k);
                        const unsigned kstop = ign ? min(kend, (ks | 3u) + 1u) : kend;
#ifdef QPSK_PIPE_PROFILE
                        rp_tick(rp_out);
                        if (rp_calls++ == 0) rp_rt1 = rp_real();
#endif
                        if (pair)
                            ST::ring_pair(ph, fr, d_base, z_base, ready_addr, consumed_addr, ks, kstop, al, be, fmin_, fmax_, odd, fl, ign);
                        else
                            ST::ring(ph, fr, d_base, z_base, ready_addr, consumed_addr, ks, kstop, al, be, fmin_, fmax_, fl, ign);
                        ran = true;
#ifdef QPSK_PIPE_PROFILE
                        rp_tick(rp_in);
                        rp_rt2 = rp_real();
#endif
                        k = ks;
                    }
                    if ((!ran || fl != 0ull) && k < kend) {      /* group k abandoned (or never started) */
                        const int at = (int)(k & 7u) * AG;
                        const unsigned long long fresh = fl & ~ign;
                        if (ran && fresh) {
                            const unsigned long long zm = excusable(fresh, at, ((int)(k & 7u) | 3) * AG + AG);
                            if ((fresh & ~zm) == 0ull) { ign |= zm; continue; }      /* every lane that tripped: the stream again */
                        }
                        /* the C++ step, then on */
                        for (int i = 0; i < AG; i++) {
                            float tx, ty; unsigned qq;
                            zl[at + i] = ph;
                            costas_step_t<true>(ph, fr, al, be, fmin_, fmax_, dl[at + i], tx, ty, qq, over);
                        }
                        k++;
                        if ((k & 3u) == 0u && lane == 0) st_release(&sm->consumed, (int)(k >> 2));
                    }
                }
            }
            ok = __all(ok);
            c = cfull - 1;
            continue;
        }
        ok = wait_ge(&sm->ready[gw], c + 1, &sm->abort_flag);
        if (!__all(ok)) { ok = false; break; }
#ifdef QPSK_PIPE_PROFILE
        {
            const unsigned long long before = cw;
            ctick(cw);
            if (cprof && lane == 0 && (a.dbg & 2048) && (c < 4 || c % 8 == 0)) printf("  serial wave: chunk %d waited %llu cycles\n", c, cw - before);
        }
#endif
        const int slot = (c % DR) * S;
        const int cnt = min(S, N - c * S);
        if (active && !QPSK_ABLATE(a, 2)) { /* measurement build only: skip the recurrence */
            int j = 0;
            if (c == 0) { /* a loaded phase may be -0: first step with the form that is exact there too */
                Loop s0 = {ph, fr};
                zl[slot] = ph;
                costas_step<true>(s0, lg, dl[slot], over);
                ph = s0.phase; fr = s0.freq;
                j = 1;
                /* the stream below starts on a symbol number that is a multiple of 4 (16-byte aligned reads of
                 * symbol pairs and writes of four records) */
                for (; j < min(4, cnt); j++) {
                    float tx, ty; unsigned qq;
                    zl[slot + j] = ph;
                    if (fast_clamp) costas_step_t<true>(ph, fr, al, be, fmin_, fmax_, dl[slot + j], tx, ty, qq, over);
                    else costas_step_t<false>(ph, fr, al, be, fmin_, fmax_, dl[slot + j], tx, ty, qq, over);
                }
            }
            /* the wave only advances the loop and leaves each step's starting phase; de-rotation to z (sin/cos
             * again, from that phase), the slicer and costas_frame[] happen in the FIR waves' flush */
            /* the next symbol is fetched from LDS one whole step ahead, so the recurrence never waits for
             * the LDS pipe (which the FIR waves keep busy); reading one slot past the chunk is harmless
             * (next slot or the row's padding element) */
            if (fast_clamp && !(a.dbg & 8)) {
                /* groups of steps in the hand-scheduled stream (costas_asm.h); a group it abandons (exact-zero
                 * detector input, double wrap) is redone here with the C++ step, and so is one that would start
                 * with freq = -0 (loaded state only), where the stream's zero error has the wrong sign */
                constexpr int AG = COSTAS_ASM_GROUP;
                unsigned long long ign = 0ull;      /* lanes excused from the zero test for the rest of this chunk's whole groups */
                while (cnt - j >= AG) {
                    unsigned da = lds_addr(dl + slot + j), za = lds_addr(zl + slot + j);
                    unsigned long long fl = 0ull;
                    bool ran = false;
                    const unsigned want = __builtin_amdgcn_readfirstlane((unsigned)(cnt - j) / AG);   /* wave-uniform */
                    unsigned left = want;
                    if (!__any(__float_as_uint(fr) == 0x80000000u)) {
                        left = ST::run(ph, fr, da, za, want, al, be, fmin_, fmax_, fl, ign);
                        ran = true;
                    }
                    j += AG * (int)(want - left);
                    if (left != 0) {
                        const unsigned long long fresh = fl & ~ign;
                        if (ran && fresh) {
                            const unsigned long long zm = excusable(fresh, slot + j, slot + j + AG * (int)left);
                            if ((fresh & ~zm) == 0ull) { ign |= zm; continue; }
                        }
                        for (int i = 0; i < AG; i++, j++) {
                            float tx, ty; unsigned qq;
                            zl[slot + j] = ph;
                            costas_step_t<true>(ph, fr, al, be, fmin_, fmax_, dl[slot + j], tx, ty, qq, over);
                        }
                    }
                }
            }
            float2 dcur = dl[slot + j];
            if (fast_clamp) {
#pragma unroll 1
                for (; j < cnt; j++) {
                    const float2 dnext = dl[slot + j + 1];
                    float tx, ty; unsigned qq;
                    zl[slot + j] = ph;
                    costas_step_t<true>(ph, fr, al, be, fmin_, fmax_, dcur, tx, ty, qq, over);
                    dcur = dnext;
                }
            } else {
#pragma unroll 4
                for (; j < cnt; j++) {
                    const float2 dnext = dl[slot + j + 1];
                    float tx, ty; unsigned qq;
                    zl[slot + j] = ph;
                    costas_step_t<false>(ph, fr, al, be, fmin_, fmax_, dcur, tx, ty, qq, over);
                    dcur = dnext;
                }
            }
        }
        if (lane == 0) st_release(&sm->consumed, c + 1);
#ifdef QPSK_PIPE_PROFILE
        ctick(cs);
#endif
    }
#ifdef QPSK_PIPE_PROFILE
    if (cprof && lane == 0 && !ring_prof)
        printf("serial wave: %d chunks; cycles per chunk: wait for the FIR waves %llu, steps %llu\n", nchunks, cw / nchunks, cs / nchunks);
    if (timeline) {
        if (lane == 0 && blockIdx.x < 1024) {
            unsigned long long *tl = g_timeline + 48 * blockIdx.x;
            tl[1] = rp_rt0; tl[2] = rp_rt1; tl[3] = rp_rt2; tl[4] = rp_real();
        }
    } else if (ring_prof && lane == 0)
        printf("wg %3d serial wave: %d chunks, %llu entries into the stream; cycles per chunk inside the stream %llu (= %llu per step), "
               "outside it (waiting for the FIR waves, redone groups) %llu; wall time (100 MHz clock): wave start -> first step %.2f us, "
               "first -> last step %.2f us\n", (int)blockIdx.x, nchunks, rp_calls, rp_in / nchunks,
               rp_in / nchunks / S, rp_out / nchunks, (double)(rp_rt1 - rp_rt0) * 0.01, (double)(rp_rt2 - rp_rt1) * 0.01);
#endif
    st.phase = ph; st.freq = fr;
    if (active && ok && !odd) {
        const size_t o = (size_t)(f0 + g) * nbw + b;
        if (a.freq) a.freq[o] = st.freq;
        if (a.phase) a.phase[o] = st.phase;
        if (a.hz) a.hz[o] = (float)((double)st.freq * a.rs / TAU); /* qpsk.c:217 */
        if (a.state_out) { a.state_out[2 * o] = st.phase; a.state_out[2 * o + 1] = st.freq; }
    }
    if (!ok && lane == 0) report_status(status, STATUS_PIPE_TIMEOUT);
    if (over) report_status(status, STATUS_PHASE_RANGE);
    else if (active && ok && !loop_state_finite(ph, fr)) report_status(status, STATUS_NONFINITE);
}

/*
 * Flush of one consumed chunk for one frame by its lanes (lane q owns symbols R q .. R q + R - 1 of the chunk):
 * record = the phase the symbol's step started from; with the symbol itself, still in the symbol ring (the slot
 * is refilled only after this flush, by the same wave), z = symbol x conj(e^{j phase}) = costas_frame[] (qpsk.c:197)
 * exactly as the step formed it, then the slicer (qpsk.c:74-79), R symbols per store.
 */
template <class GM, int RW = GM::R>   /* RW symbols per lane: S / RW lanes per frame */
__device__ __forceinline__ void flush_records(const FusedArgs &a, const float *zring, const float2 *dring, int g, int frame,
                                              int q, int chunk, uint8_t *sym_row = nullptr)
{   /* sym_row (one loop per frame): where this frame's symbols go instead of a.sym + frame * nsym (rx_lean_kernel's pad rows) */
    constexpr int DSTRIDE = GM::DSTRIDE;
    constexpr int R = RW, S = GM::S, ZSTRIDE = GM::ZSTRIDE;
    static_assert(R == 2 || R == 4, "packs 2 or 4 symbols per store");
    const int nbw = a.nbw, N = a.nsym;
    const int slot = (chunk % DR) * S, sym0 = chunk * S;
    const int cnt = min(S, N - sym0);
    for (int b = 0; b < nbw; b++) {
        const int row = g * nbw + b;
        const size_t o = ((size_t)frame * nbw + b) * N + sym0 + R * q;
        float2 z[R];
        uint32_t packed = 0;
        /* the lane's R records and R symbols: rows are 16-byte aligned and R q is a multiple of R, so they come
         * as whole 8- or 16-byte LDS reads */
        float phv[R];
        float2 dv[R];
        {
            const float *zp = zring + (size_t)row * ZSTRIDE + slot + R * q;
            const float2 *dp = dring + (size_t)g * DSTRIDE + slot + R * q;
            if constexpr (R == 4) {
                const float4 p4 = *reinterpret_cast<const float4 *>(zp);
                const float4 da = reinterpret_cast<const float4 *>(dp)[0], db = reinterpret_cast<const float4 *>(dp)[1];
                phv[0] = p4.x; phv[1] = p4.y; phv[2] = p4.z; phv[3] = p4.w;
                dv[0] = make_float2(da.x, da.y); dv[1] = make_float2(da.z, da.w);
                dv[2] = make_float2(db.x, db.y); dv[3] = make_float2(db.z, db.w);
            } else {
                const float2 p2 = *reinterpret_cast<const float2 *>(zp);
                const float4 da = *reinterpret_cast<const float4 *>(dp);
                phv[0] = p2.x; phv[1] = p2.y;
                dv[0] = make_float2(da.x, da.y); dv[1] = make_float2(da.z, da.w);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++)
            z[r] = derotate<false>(phv[r], dv[r]);
        if (chunk == 0 && q == 0)   /* the frame's first symbol: its phase may be a loaded -0 */
            z[0] = derotate<true>(phv[0], dv[0]);
#pragma unroll
        for (int r = 0; r < R; r++)
            packed |= (uint32_t)slicer(z[r]) << (8 * r);
        if (a.sym) {
            uint8_t *so = sym_row ? sym_row + sym0 + R * q : a.sym + o;
            if (R * q + R <= cnt && ((N | sym0) & (R - 1)) == 0) {   /* o is a multiple of R: one aligned store */
                if constexpr (R == 4) *reinterpret_cast<uint32_t *>(so) = packed;
                else *reinterpret_cast<uint16_t *>(so) = (uint16_t)packed;
            } else {
                for (int r = 0; r < R; r++)
                    if (R * q + r < cnt) so[r] = (uint8_t)(packed >> (8 * r));
            }
        }
        if (a.costas) {
            for (int r = 0; r < R; r++)
                if (R * q + r < cnt) a.costas[o + r] = z[r];
        }
    }
}

/* ========================================================================
 * costas_pipe_kernel: the same pipeline with ALREADY DECIMATED symbols as
 * input (qpsk.c:196-212 alone: qpsk_costas_batch, and the streaming mode,
 * where decimated_frame[] carries the previous block, qpsk.c:186-197).
 * Waves 1..NF only move data: 64 symbols per frame per chunk from global
 * memory into the symbol ring, and the flush of consumed records.
 * a.dsrc rows are a.dstride symbols apart; one loop per frame (nbw = 1).
 * ======================================================================== */
__global__ void __launch_bounds__(64 * (GeomNarrow::MAX_NF + 2))
costas_pipe_kernel(FusedArgs a, int *status)
{
    using GM = GeomNarrow;
    QPSK_GEOM_CONSTANTS(GM);
    static_assert(R == 2 || R == 4, "flush_records packs 2 or 4 symbols per store");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem *sm = reinterpret_cast<Smem *>(smem_raw);
    const int NF = (int)blockDim.x / 64 - 1 - (a.carrier_state ? 1 : 0);
    const int G = NF * FWV;
    float2 *dring = reinterpret_cast<float2 *>(smem_raw + sizeof(Smem));     /* [G][DSTRIDE] */
    float *zring = reinterpret_cast<float *>(dring + (size_t)G * DSTRIDE);  /* [G][ZSTRIDE] */
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int N = a.nsym;
    const int f0 = blockIdx.x * G;
    const int nchunks = (N + S - 1) / S;
    if (tid < MAX_WAVES) sm->ready[tid] = 0;
    if (tid == 0) { sm->consumed = 0; sm->abort_flag = 0; }
    __syncthreads();

    if (wave == 0) {
        costas_wave<GM>(a, sm, dring, zring, G, f0, lane, nchunks, status);
        return;
    }
    if (wave == NF + 1) {      /* the spare wave (streams behind stream_scan_kernel MODE 2): the rest of the next block's carrier table */
        if (blockIdx.x == 0 && lane == 0)
            carrier_block(a.carrier_state, a.carrier_tab, a.carrier_frame, carrier_split(a.carrier_frame), a.carrier_frame / 2);
        return;
    }
    const int w = wave - 1;
    const int fl = lane / QL, q = lane % QL;
    const int g = w * FWV + fl;
    const int frame = f0 + g;
    const bool fvalid = frame < a.nframes;
    const float2 *src = a.dsrc + (size_t)(fvalid ? frame : 0) * a.dstride;
    int flushed = 0;
    bool ok = true;
    float2 pre[R];
    auto fetch = [&](int c) {
#pragma unroll
        for (int r = 0; r < R; r++)
            pre[r] = src[min(c * S + R * q + r, N - 1)];     /* clamped: symbols past N are never consumed */
    };
    fetch(0);
    for (int c = 0; c < nchunks; c++) {
        if (c >= DR) {
            ok = wait_ge(&sm->consumed, c - DR + 1, &sm->abort_flag);
            if (!ok) break;
            for (; flushed < c - DR + 1; flushed++)
                if (fvalid) flush_records<GM>(a, zring, dring, g, frame, q, flushed);
        }
        float2 *dw = dring + (size_t)g * DSTRIDE + (c % DR) * S + R * q;
#pragma unroll
        for (int r = 0; r < R; r++)
            dw[r] = pre[r];
        if (c + 1 < nchunks) fetch(c + 1);
        if (lane == 0) st_release(&sm->ready[w], c + 1);
        if (a.refill && fvalid) {
            /* decimated_frame[i] <- input_frame[i*CYCLES + index] of the block just filtered (qpsk.c:186-191):
             * the slots this lane has just emptied, off the hand-over path */
            const float2 *blk = a.refill + (size_t)frame * a.frame_size;
            float2 *row = a.refill_dst + (size_t)frame * a.dstride;
            const int ix = a.index ? a.index[frame] : a.fixed_index;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int i = c * S + R * q + r, s = i * a.cycles + ix;
                if (i < N) {
                    if (a.refill_planar) {      /* the block planar by decimation phase (streamscan.hip): [cycles][symbols]; an index of
                                                  * CYCLES or more (the histogram has 8 bins whatever CYCLES is) reaches into the next symbol */
                        const int pl = ix % a.cycles, sy = i + ix / a.cycles;
                        row[i] = sy < N ? blk[(size_t)pl * N + sy] : make_float2(0.0f, 0.0f);
                    } else {
                        row[i] = s < a.frame_size ? blk[s] : make_float2(0.0f, 0.0f);
                    }
                }
            }
        }
    }
    if (ok) {
        ok = wait_ge(&sm->consumed, nchunks, &sm->abort_flag);
        if (ok)
            for (; flushed < nchunks; flushed++)
                if (fvalid) flush_records<GM>(a, zring, dring, g, frame, q, flushed);
    }
    if (!ok && lane == 0) report_status(status, STATUS_PIPE_TIMEOUT);
}

template <class GM>
__global__ void __launch_bounds__(GM::MAX_THREADS)
rx_fused_pipe_kernel(FusedArgs a, int *status)
{
    QPSK_GEOM_CONSTANTS(GM);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem *sm = reinterpret_cast<Smem *>(smem_raw);
    /* Waves are dealt to the SIMDs round-robin, so waves 4, 8, ... land beside the serial wave 0.  With spare
     * waves the host launches those too and they retire at once: the serial wave keeps SIMD 0 to itself and the
     * FIR waves are hardware waves 1,2,3, 5,6,7, 9,10 (QPSK_PIPE_DBG bit 2 turns the spares off). */
    const bool spares = GM::SPARE && !(a.dbg & 4);
    const int nwaves = (int)blockDim.x / 64;
    const bool mixed = a.mixed != 0;                                        /* 16 frames, two lane mappings (see the FIR waves) */
    const int nfir = spares ? (nwaves - 1) - (nwaves - 1) / 4 : nwaves - 1;   /* FIR waves unless mixed */
    const int G = a.mixed == 1 ? 4 * FWV : a.mixed == 2 ? 2 * nfir : nfir * FWV;   /* frames of the workgroup */
    float2 *win = reinterpret_cast<float2 *>(smem_raw + sizeof(Smem));   /* [G][WSLOTS] */
    float2 *dring = win + (size_t)G * WSLOTS;                              /* [G][DSTRIDE] */
    float *zring = reinterpret_cast<float *>(dring + (size_t)G * DSTRIDE);   /* [G*nbw][ZSTRIDE] records: the phase each
                                                                               symbol's step started from, see flush_records */

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   /* scalar: per-wave frame pointers stay in SGPRs */
    const int L = a.frame_size, N = a.nsym;
    const int f0 = blockIdx.x * G;
    const int nchunks = (N + S - 1) / S;

    /* ---- common prologue: taps, counters, zeroed windows (= fresh delay lines, qpsk.c:37) */
    for (int i = tid; i < 128; i += blockDim.x)
        sm->taps[i] = i < NTAPS ? a.taps[i] : 0.0f;
    if (tid < MAX_WAVES) sm->ready[tid] = 0;
    if (tid == 0) { sm->consumed = 0; sm->abort_flag = 0; }
    __syncthreads();      /* (no window to zero: a fresh delay line is a zero history in the FIR waves' registers) */

    /* ---- BASELINE config 3: the FFT timing estimate inside the launch (timing_fft_wave.h; the host sets a.est_tw only for full
     * 16-frame workgroups = 8 hardware waves).  Every hardware wave -- the serial wave and the two that retire included, all idle
     * until the first chunk exists -- estimates frames 2 wave, 2 wave + 1 of the workgroup: 4 frames per SIMD, the full-rate
     * stream fir_full8_asm.h on 512 samples each, its window in the (not yet used) frame windows' LDS.  No launch of its own, no
     * drain and refill of the chip between the estimate and the pipeline; the indices never leave the CU. */
    if (a.est_tw) {
        float2 *ewin = win + (size_t)wave * tfft::WSLOTS;
        const unsigned erd = lds_addr(ewin + (tfft::R + tfft::PADS) * lane), etap = lds_addr(sm->taps);
        const int fe0 = f0 + 2 * wave;
        /* both frames unconditionally (a frame past the batch's end re-reads the last one and its result is dropped): the samples
         * stay in registers, the second frame's are in flight while the first is filtered */
        const float2 *srcA = a.x + (size_t)min(fe0, a.nframes - 1) * a.frame_pitch;
        const float2 *srcB = a.x + (size_t)min(fe0 + 1, a.nframes - 1) * a.frame_pitch;
        tfft::Pre5 pre = tfft::load_frame5(srcA, lane);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            tfft::stage_frame5(ewin, lane, pre);
            if (i == 0) pre = tfft::load_frame5(srcB, lane);
            tfft::wave_sync();
            v2f e0, e1, e2, e3, e4, e5, e6, e7;
            fir_full8_asm(erd, etap, e0, e1, e2, e3, e4, e5, e6, e7);
            const v2f eacc[tfft::R] = {e0, e1, e2, e3, e4, e5, e6, e7};
            double epv[tfft::R];
            const tfft::cd u = tfft::power_bin(eacc, ewin, lane, a.est_tw, tfft::NFFT / C, nullptr, epv);
            if (lane == 0 && fe0 + i < a.nframes) {
                const int best = tfft::pick_index(u, a.est_cs, C, nullptr);
                sm->est_index[2 * wave + i] = best;
                if (a.index_out) a.index_out[fe0 + i] = best;
            }
        }
        __syncthreads();
    }

    if (wave == 0) {
        costas_wave<GM>(a, sm, dring, zring, G, f0, lane, nchunks, status);
        return;
    }

    /* ======================================= FIR waves =========================================== */
    /* Which frames a hardware wave filters, and with which lane mapping.
     *   plain: FIR wave w owns the 4 frames of group w (with spares, hardware waves 4, 8 retire at once).
     *   mixed (measurement variant of the full narrow workgroup, QPSK_PIPE_DBG bit 7): four 4-frame units on
     *     three SIMDs leave one SIMD with two, so 12 frames go to hardware waves 1-3 in the 4-symbol mapping (one
     *     per SIMD) and the last 4 to hardware waves 6 and 7 (SIMDs 2 and 3), two frames each in the 2-symbol
     *     mapping (32 lanes per frame): 1, 1.5 and 1.5 units per SIMD instead of 2, 1, 1.  Every wave still owns
     *     its frames outright.  Bit-exact like the plain layout, and no faster (see the launcher). */
    int w, gbase, rslot;          /* FIR wave index, first frame slot, its ready[] counter */
    bool half = false;
    if (a.mixed == 2) {   /* several loops per frame: every FIR wave takes two frames, 2 symbols per lane (the flush
                             does sin/cos once per loop and symbol, so the frames are spread over more waves) */
        if (spares && (wave & 3) == 0) return;
        w = spares ? wave - 1 - wave / 4 : wave - 1;
        half = true;
        gbase = 2 * w;
        rslot = w;
    } else if (mixed) {
        if (wave >= 4 && wave <= 5) return;
        half = wave >= 6;
        w = half ? wave - 3 : wave - 1;
        gbase = half ? 12 + 2 * (wave - 6) : 4 * (wave - 1);
        rslot = w;
    } else {
        if (spares && (wave & 3) == 0) return;   /* a wave that would share a SIMD with wave 0 */
        w = spares ? wave - 1 - wave / 4 : wave - 1;   /* FIR wave index = the frame group it owns */
        gbase = w * GM::FWV;
        rslot = w;
    }

  auto fir_wave = [&](auto wmap) {
    /* this wave's lane mapping: QL lanes per frame, R symbols per lane (QL * R = S for every wave of the workgroup) */
    /* window image: position p at slot p + PADS (p / PAD), PADS = 2 -- lanes are 16-byte multiples apart (272 bytes for
     * four symbols per lane, 144 for two), so positions (t, t+1), t even, are one aligned 16-byte word, which is what
     * the hand-scheduled filter streams read (fir_r4_asm.h, fir_r2_asm.h: one ds_read_b128 per two positions) */
    constexpr int PADS = 2;
    constexpr int QL = decltype(wmap)::QL, R = decltype(wmap)::R, FWV = 64 / QL, PAD = R * C,
                  TSTEPS = NTAPS + C * (R - 1);
    static_assert(QL * R == S && 128 % PAD == 0 && PAD >= 2 * C, "chunk size and window padding are the workgroup's");
    const int fl = lane / QL, q = lane % QL;

    /* taps stay in LDS; the FIR loop keeps a rolling set of R + 1 groups of C taps in registers (broadcast reads) */
    const float4 *taps4 = reinterpret_cast<const float4 *>(sm->taps);
    constexpr int NLD = CH / 128;                          /* 16-byte loads per frame per chunk */

    /* what a wave needs to know about its frames: one 16-byte load covers 128 samples of ONE frame, so the
     * wave's 64 lanes sweep a frame in NLD loads; lane `lane` holds samples 2*lane, 2*lane+1 of each 128-block */
    struct Ctx {
        int group, g, frame;       /* ready[] slot, this lane's frame slot in the workgroup, its frame */
        bool fvalid, all_valid;
        float2 *wf;                /* this lane's frame window */
        const float2 *rd;          /* FIR read base: position PAD*q -> slot (PAD+1)*q */
        int wr0[FWV], wr1[FWV];    /* window write slots of the loaded pair, per frame of the wave */
        bool even[FWV];            /* the frame's decimation offset is even: the pair is one aligned 16-byte word */
        bool h0[FWV], h1[FWV];     /* per lane: the pair's first / second sample of the LAST block of a chunk is history of the next */
        float4 hist[FWV];          /* per lane: the pair it loaded from the last block of the previous chunk */
        const float4 *src[FWV];
        bool fv[FWV];
    };
    auto make_ctx = [&]() {
        Ctx cx;
        cx.group = rslot;
        cx.g = gbase + fl;
        cx.frame = f0 + cx.g;
        cx.fvalid = cx.frame < a.nframes;
        cx.all_valid = f0 + gbase + FWV <= a.nframes;
        cx.wf = win + (size_t)cx.g * WSLOTS;
        cx.rd = cx.wf + (PAD + PADS) * q;
#pragma unroll
        for (int ff = 0; ff < FWV; ff++) {
            const int fr = f0 + gbase + ff;
            cx.fv[ff] = fr < a.nframes;
            const int ix = a.est_tw ? (cx.fv[ff] ? sm->est_index[gbase + ff] : 0)
                                    : a.index ? (cx.fv[ff] ? a.index[fr] : 0) : a.fixed_index;   /* decimation offset, < C */
            const int p0 = 2 * lane + 126 - ix;               /* window position of sample 2*lane of the chunk */
            cx.wr0[ff] = (gbase + ff) * WSLOTS + p0 + PADS * (p0 / PAD);
            cx.wr1[ff] = (gbase + ff) * WSLOTS + (p0 + 1) + PADS * ((p0 + 1) / PAD);
            cx.even[ff] = (ix & 1) == 0;
            cx.h0[ff] = p0 >= 128;                            /* position p0 - 128 of the next chunk's window exists */
            cx.h1[ff] = p0 + 1 >= 128;
            cx.hist[ff] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   /* a fresh delay line (qpsk.c:37) */
            cx.src[ff] = reinterpret_cast<const float4 *>(a.x + (size_t)(cx.fv[ff] ? fr : 0) * a.frame_pitch);
        }
        return cx;
    };

    /* L is even on this path (checked by the host), so a 16-byte pair is either inside the frame or past it.
     * Loads are UNCONDITIONAL (a per-load branch would make the compiler wait for each load in turn): a
     * chunk that lies inside every frame of the group -- all but the last one -- loads straight; the tail
     * chunk clamps the address into the frame and zeroes what lies past the end afterwards. */
    auto prefetch = [&](const Ctx &cx, float4 (&pre)[FWV][NLD], int c) {
        if (cx.all_valid && (c + 1) * CH <= L) {
#pragma unroll
            for (int ff = 0; ff < FWV; ff++)
#pragma unroll
                for (int j = 0; j < NLD; j++)
                    pre[ff][j] = load_once(&cx.src[ff][(c * CH + 128 * j + 2 * lane) >> 1]);
        } else {
#pragma unroll
            for (int ff = 0; ff < FWV; ff++) {
#pragma unroll
                for (int j = 0; j < NLD; j++) {
                    const int s = c * CH + 128 * j + 2 * lane;      /* first sample of the pair */
                    const bool in = cx.fv[ff] && s + 1 < L;
                    float4 v = load_once(&cx.src[ff][in ? (s >> 1) : 0]);
                    if (!in) v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    pre[ff][j] = v;
                }
            }
        }
    };

    /* measurement build only (make -C qpsk_amd/csrc profile, then QPSK_PIPE_DBG bit 5): where a FIR wave of
     * workgroup 0 spends its shader cycles.  The product library carries neither the counters nor the printf. */
#ifdef QPSK_PIPE_PROFILE
    const bool prof = (a.dbg & 32) && blockIdx.x == 0;
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tlast = 0;
    auto tick = [&](int k) {
        if (prof) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (k >= 0) tacc[k] += t - tlast;
            tlast = t;
        }
    };
#else
    auto tick = [](int) {};
#endif

    /* Every wave turns its own frames' records into symbols: the flush needs the symbols themselves, and they
     * stay in the ring only until their owner refills the slot -- which it does right after this flush.  (With the
     * (T, quadrant) records of the earlier design any wave could flush for any other, and the waves with slack did;
     * the phase records take 8 bytes per step off the serial wave's LDS writes instead.) */
    const int fbase[1] = {gbase};
    const int nflush = 1;
    auto flush_chunk = [&](int chunk) {
        for (int i = 0; i < nflush; i++) {
            const int g2 = fbase[i] + fl, fr2 = f0 + g2;
            if (fr2 < a.nframes) flush_records<GM, R>(a, zring, dring, g2, fr2, q, chunk);
        }
    };

    /* one chunk of one frame group: flush what the loop has finished with, stage the window, filter, hand over */
    auto run_chunk = [&](Ctx &cx, float4 (&pre)[FWV][NLD], int c, bool prefetch_next) -> bool {
        /* The window is rebuilt from REGISTERS every chunk: the 126 samples of history a frame carries are the tail
         * of the last 128-sample block its wave loaded for the previous chunk (4 VGPRs per frame), written one block
         * below block 0; then the prefetched samples of this chunk.  No copy through LDS, nothing to zero.  An even
         * decimation offset makes a lane's pair one aligned 16-byte word of the image. */
        constexpr int BLK = 128 + PADS * (128 / PAD);
#pragma unroll
        for (int ff = 0; ff < FWV; ff++) {
            if (cx.even[ff]) {      /* wave-uniform: every lane holds a pair of frame ff here */
                if (cx.h0[ff]) *reinterpret_cast<float4 *>(win + cx.wr0[ff] - BLK) = cx.hist[ff];
#pragma unroll
                for (int j = 0; j < NLD; j++)
                    *reinterpret_cast<float4 *>(win + cx.wr0[ff] + BLK * j) = pre[ff][j];
            } else {
                if (cx.h0[ff]) win[cx.wr0[ff] - BLK] = make_float2(cx.hist[ff].x, cx.hist[ff].y);
                if (cx.h1[ff]) win[cx.wr1[ff] - BLK] = make_float2(cx.hist[ff].z, cx.hist[ff].w);
#pragma unroll
                for (int j = 0; j < NLD; j++) {
                    win[cx.wr0[ff] + BLK * j] = make_float2(pre[ff][j].x, pre[ff][j].y);
                    win[cx.wr1[ff] + BLK * j] = make_float2(pre[ff][j].z, pre[ff][j].w);
                }
            }
            cx.hist[ff] = pre[ff][NLD - 1];
        }
        if (prefetch_next) prefetch(cx, pre, c + 1);
        tick(2);

        /* sliding-window FIR: symbol r of this lane uses tap k = t - C*r at window position PAD*q + t */
        const float2 *rd = cx.rd;
        float2 acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = QPSK_ABLATE(a, 1) ? make_float2(0.7f, 0.3f) : make_float2(0.0f, 0.0f);
        /* t = C*tb + u: symbol r needs tap group tb - r (taps C*(tb-r) .. +C-1), so each group of C taps is
         * live for R consecutive blocks */
        static_assert(C == 8 && (R == 4 || R == 2), "the step below is written for C = 8 and R = 2 or 4");
        if constexpr (!GM::PINNED) {
            /* the compiler's own schedule of the same sum (a Geom with PINNED = false; no longer instantiated):
             * config 2 ran 0.2148 ms with it against 0.2088 ms with the pinned order below [measured, same process] */
            if (!QPSK_ABLATE(a, 1)) {
                float tgc[R][C];
#pragma unroll
                for (int tb = 0; tb * C < TSTEPS; tb++) {
                    if (tb * C < NTAPS) {
                        const float4 ta = taps4[2 * tb], tb4 = taps4[2 * tb + 1];
                        tgc[tb % R][0] = ta.x; tgc[tb % R][1] = ta.y; tgc[tb % R][2] = ta.z; tgc[tb % R][3] = ta.w;
                        tgc[tb % R][4] = tb4.x; tgc[tb % R][5] = tb4.y; tgc[tb % R][6] = tb4.z; tgc[tb % R][7] = tb4.w;
                    }
#pragma unroll
                    for (int u = 0; u < C; u++) {
                        const int t = tb * C + u;
                        if (t < TSTEPS) {
                            const float2 v = rd[t + PADS * (t / PAD)];
#pragma unroll
                            for (int r = 0; r < R; r++) {
                                const int k = t - C * r;
                                if (k >= 0 && k < NTAPS) fir_mac(acc[r], v, tgc[(tb - r + R) % R][u]);
                            }
                        }
                    }
                }
            }
        } else if (QPSK_PIPE1_ASM && !QPSK_ABLATE(a, 1)) {
            /* the hand-scheduled streams (generated, tools/gen_fir_asm.py): the sum of the step below, same order */
            if constexpr (R == 2) {
                v2f a0, a1;
                fir_r2_asm(lds_addr(rd), lds_addr(sm->taps), a0, a1);
                acc[0] = make_float2(a0.x, a0.y);
                acc[1] = make_float2(a1.x, a1.y);
            } else {
                v2f a0, a1, a2, a3;
                fir_r4_asm(lds_addr(rd), lds_addr(sm->taps), a0, a1, a2, a3);
                acc[0] = make_float2(a0.x, a0.y); acc[1] = make_float2(a1.x, a1.y);
                acc[R > 2 ? 2 : 0] = make_float2(a2.x, a2.y); acc[R > 2 ? 3 : 1] = make_float2(a3.x, a3.y);
            }
        } else if (!QPSK_ABLATE(a, 1)) { /* measurement build only: skip the filter arithmetic, keep the traffic */
            /* Software pipeline, one block of C window positions deep: the LDS reads of block tb+1 (window values
             * and tap group) are issued before the multiply-adds of block tb.  Groups rotate through R + 1 slots so
             * that the group fetched a block early does not overwrite the one symbol R-1 still needs.
             * Instruction order inside a step is pinned with empty asm statements (they keep their program order
             * and every value they name must exist where they stand): the R products of a step, then the R
             * adds.  A product is then R instructions ahead of its add and an accumulator's adds are 2R apart,
             * more than the ~8 cycles a dependent packed op waits. */
            constexpr int NB = (TSTEPS + C - 1) / C;
            float tg[R + 1][C];
            float2 wv[2][C];
            auto fetch_block = [&](int tb) {
                if (tb * C < NTAPS) {
                    /* (taps through the scalar cache into SGPRs instead -- s_load_dwordx8 from the constant address
                     * space, SGPR operands in the packed multiplies, 43 VGPRs fewer -- measured: 8192 frames
                     * 0.3456 against 0.3514 ms, 4096 frames 0.1951 against 0.1881 ms: the scalar loads share
                     * lgkmcnt with the window reads.  Not kept.) */
                    const float4 ta = taps4[2 * tb], tb4 = taps4[2 * tb + 1];
                    float *g_ = tg[tb % (R + 1)];
                    g_[0] = ta.x; g_[1] = ta.y; g_[2] = ta.z; g_[3] = ta.w;
                    g_[4] = tb4.x; g_[5] = tb4.y; g_[6] = tb4.z; g_[7] = tb4.w;
                }
#pragma unroll
                for (int u = 0; u < C; u++) {
                    const int t = tb * C + u;
                    if (t < TSTEPS) wv[tb & 1][u] = rd[t + PADS * (t / PAD)];
                }
            };
            v2f ac[R];
#pragma unroll
            for (int r = 0; r < R; r++) ac[r] = v2f{acc[r].x, acc[r].y};
            fetch_block(0);
            /* static_for: the compiler does not unroll a loop with an asm statement in it */
            static_for<0, NB>([&](auto tbc) {
                constexpr int tb = decltype(tbc)::value;
                if (tb + 1 < NB) fetch_block(tb + 1);
                static_for<0, C>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    constexpr int t = tb * C + u;
                    if constexpr (t < TSTEPS) {
                        const v2f v = v2f{wv[tb & 1][u].x, wv[tb & 1][u].y};
                        constexpr bool ok0 = t < NTAPS, ok1 = t >= C && t - C < NTAPS,
                                       ok2 = R > 2 && t >= 2 * C && t - 2 * C < NTAPS,
                                       ok3 = R > 2 && t >= 3 * C && t - 3 * C < NTAPS;
                        v2f p0, p1, p2, p3;    /* re*tap, im*tap (rrc_fir.c:24-25) */
                        if constexpr (ok0) p0 = v * tg[(tb + (R + 1)) % (R + 1)][u];
                        if constexpr (ok1) p1 = v * tg[(tb - 1 + (R + 1)) % (R + 1)][u];
                        if constexpr (ok2) p2 = v * tg[(tb - 2 + (R + 1)) % (R + 1)][u];
                        if constexpr (ok3) p3 = v * tg[(tb - 3 + (R + 1)) % (R + 1)][u];
                        if constexpr (ok0 && ok1 && ok2 && ok3) {
                            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                        } else if constexpr (R == 2 && ok0 && ok1) {
                            asm volatile("" : "+v"(p0), "+v"(p1));
                        } else {
                            if constexpr (ok0) asm volatile("" : "+v"(p0));
                            if constexpr (ok1) asm volatile("" : "+v"(p1));
                            if constexpr (ok2) asm volatile("" : "+v"(p2));
                            if constexpr (ok3) asm volatile("" : "+v"(p3));
                        }
                        if constexpr (ok0) ac[0] = ac[0] + p0;
                        if constexpr (ok1) ac[1] = ac[1] + p1;
                        if constexpr (ok2) ac[2] = ac[2] + p2;
                        if constexpr (ok3) ac[3] = ac[3] + p3;
                        if constexpr (R == 4) asm volatile("" : "+v"(ac[0]), "+v"(ac[1]), "+v"(ac[2]), "+v"(ac[3]));
                        else asm volatile("" : "+v"(ac[0]), "+v"(ac[1]));
                    }
                });
            });
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = make_float2(ac[r].x, ac[r].y);
        }
        tick(3);
        /* Only now does the wave need the loop: ring slot c % DR is free once chunk c - DR has been consumed, and
         * that chunk's outputs leave first (staging and filtering above touch neither ring, so the FIR waves run
         * up to two chunks ahead of the loop instead of one) */
        if (c >= DR) {
            if (!wait_ge(&sm->consumed, c - DR + 1, &sm->abort_flag)) return false;
            tick(0);
            flush_chunk(c - DR);
            tick(1);
        }
        /* decimated symbols -> ring; a pick at or past the end of the block is 0 (cannot happen for idx < C) */
        float2 *dw = dring + (size_t)cx.g * DSTRIDE + (c % DR) * S + R * q;
#pragma unroll
        for (int r = 0; r < R; r++)
            dw[r] = fir_gain(acc[r]);
        if (lane == 0) st_release(&sm->ready[cx.group], c + 1);
        tick(4);
        return true;
    };

    Ctx own = make_ctx();
    float4 pre[FWV][NLD];
    prefetch(own, pre, 0);
    bool ok = true;
    tick(-1);
    for (int c = 0; c < nchunks && ok; c++) {
        /* mixed layout: SIMDs 2 and 3 each carry a four-frame wave and a (younger) two-frame wave, and a SIMD serves
         * its oldest wave first -- the two-frame waves were the ones with no slack.  The wave that is behind its SIMD
         * partner goes first, chunk by chunk (as in timing_scan_kernel): 1-2 % at config 2 (0.1707 against 0.1741 ms and
         * 0.1745 against 0.1757 ms in two same-process runs; QPSK_PIPE_DBG bit 64 = without). */
        if (a.mixed == 1 && w >= 1 && !(a.dbg & 64)) {
            const int pt = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sm->ready[w <= 2 ? w + 2 : w - 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (pt > c) __builtin_amdgcn_s_setprio(2);
            else if (pt < c) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(1);
        }
        ok = run_chunk(own, pre, c, c + 1 < nchunks);
    }
    if (ok) {
        ok = wait_ge(&sm->consumed, nchunks, &sm->abort_flag);
        if (ok)
            for (int c = max(0, nchunks - DR); c < nchunks; c++) flush_chunk(c);
    }
    if (!ok && lane == 0) report_status(status, STATUS_PIPE_TIMEOUT);
#ifdef QPSK_PIPE_PROFILE
    if (prof && lane == 0)
        printf("FIR wave %d (%d symbols per lane): %d chunks; cycles per chunk: wait for the loop %llu, flush %llu, history + window %llu, "
               "filter %llu, ring hand-over %llu\n", w, R, nchunks, tacc[0] / nchunks, tacc[1] / nchunks, tacc[2] / nchunks,
               tacc[3] / nchunks, tacc[4] / nchunks);
#endif
  };   /* fir_wave */

    if constexpr (GM::R == 4) {   /* the mixed workgroup exists for 64-symbol chunks only */
        if (half) {
            fir_wave(WaveMap<GM::S / 2, 2>{});
            return;
        }
    }
    fir_wave(WaveMap<GM::QL, GM::R>{});
}


/* ========================================================================
 * rx_pipe2_kernel: the same pipeline laid out for up to 32 frames per workgroup (BASELINE config 4's per-GPU
 * share: 8192 frames = 32 per CU), where the recurrence costs the serial wave no more than for 16 frames and the
 * FILTER decides the time.  Measured on the layouts above: a FIR wave alone on its SIMD issues a
 * packed instruction every 6.2 cycles (every LDS instruction costs it ~15), two on one SIMD 4.07 between them (the
 * SIMD's rate); the serial wave loses half its speed as soon as anything shares its SIMD.  Hence:
 *   wave 0            the serial wave, alone on its SIMD (hardware waves 4 and 8 retire at once);
 *   hardware waves 1-3, 5-7, 9-11   up to nine FIR waves, two or three on each of the other SIMDs;
 *   unit              2 frames x 64 symbols of one chunk, lane = (frame of 2) x (q of 32), 2 symbols per lane.
 *                     FIR wave i owns units i, i + nfir (static), so SIMDs 1-3 carry 6, 5, 5 of a full
 *                     workgroup's 16 units.
 * What makes 32 frames fit the 160 KB: the window (hist + one chunk of samples, 5.4 KB per frame) belongs to the
 * WAVE, not the frame -- a wave stages, filters and hands over one unit after the other in the same 10.9 KB -- and
 * the 126 samples of history a frame carries from chunk to chunk stay in the owner's REGISTERS (they are the last
 * 128-sample block it loaded: 4 VGPRs per frame), so there is no history copy through LDS and no window to zero.
 * Rings, records, flush and the serial wave are those of rx_fused_pipe_kernel (costas_wave, flush_records).
 * ======================================================================== */
#define QPSK_PIPE2_MAXFIR 9  /* FIR waves per workgroup: three per SIMD, so at most 168 VGPRs.  (Two per SIMD with three units
                                each and a deeper LDS prefetch in 256 VGPRs: 0.31 against 0.29 ms at 8192 frames.) */
#ifndef QPSK_PIPE2_ASM
#define QPSK_PIPE2_ASM 1     /* 1: the FIR step as the generated instruction stream fir_r2_asm.h; 0: the compiler's (A/B builds) */
#endif
#ifndef QPSK_PIPE2_DEPTH
#define QPSK_PIPE2_DEPTH 1   /* blocks of 8 window positions fetched ahead of their use (measured: see DESIGN.md 4.1) */
#endif
namespace pipe2 {
constexpr int QL = 32, R = 2, UF = 2;                    /* lanes per frame, symbols per lane, frames per unit */
constexpr int S = GeomNarrow::S, CH = S * C;              /* 64 symbols = 512 samples per chunk per frame */
constexpr int PAD = R * C, PADS = 2;                      /* position p lives at slot p + 2 (p/16): lanes 18 slots = 144 bytes apart, so a
                                                             lane's positions (t, t+1), t even, are one aligned 16-byte word and the
                                                             16 lanes of a ds_read_b128 pass start in 16 different bank quads */
constexpr int TSTEPS = NTAPS + C * (R - 1);               /* 135 window positions per lane per chunk */
constexpr int NLD = CH / 128;                             /* 16-byte loads per frame per chunk */
constexpr int BLK = 128 + PADS * (128 / PAD);             /* slots per 128-sample block */
constexpr int WS = 720;                                   /* slots per frame window: positions 0..637 -> slots 0..715 */
__device__ __host__ constexpr int slot_of(int p) { return p + PADS * (p / PAD); }
constexpr int MAX_UNITS = 16, MAX_FIR = QPSK_PIPE2_MAXFIR, MAX_UW = (MAX_UNITS + MAX_FIR - 1) / MAX_FIR;   /* units per workgroup, FIR waves, units per FIR wave */
constexpr int MAX_THREADS = 64 * (MAX_FIR + 1 + (MAX_FIR - 1) / 3);   /* 12 hardware waves */
static_assert(S == QL * R && CH % 128 == 0 && slot_of(CH + HIST - 1) < WS && WS % 2 == 0 && PAD % 2 == 0 &&
              PAD * (QL - 1) + TSTEPS - 1 <= CH + HIST - 1 - (C - 1), "window geometry: every position a lane reads is written for every index < C");
static_assert(MAX_UNITS <= MAX_WAVES, "one ready[] counter per unit");

/* what a FIR wave keeps per unit (wave-uniform except where noted) */
struct Unit {
    int u;                    /* unit number in the workgroup = its ready[] counter; frames 2u, 2u + 1 */
    bool fv[UF], all_valid;   /* frames inside the batch */
    const float4 *src[UF];
    int ix[UF];               /* the frames' decimation offsets (< C): sample 2*lane of a chunk lives at window position
                                 2*lane + HIST - ix; when ix is even the lane's pair is one aligned 16-byte word */
    float4 hist[UF];          /* per lane: the pair it loaded from the LAST block of the previous chunk */
};
} // namespace pipe2

/* NUW = units of this wave, unrolled: each unit's state (8 VGPRs of history, a few scalars) has its own registers.
 * [Measured and not kept: ONE copy of the unit's code with the state selected per iteration -- 47 KB of kernel
 * instead of 79 KB, no more instruction fetches beyond the cache two CUs share (the unrolled form reads 6 % more
 * bytes than the batch holds, PMC FETCH_SIZE) -- but the register allocator then reloads spilled loop invariants
 * right behind the sample prefetch, and a scratch reload waits for every load issued before it: 0.356 against
 * 0.32 ms at 8192 frames on the same box.] */
template <int NUW>
__device__ __forceinline__ void fir_wave2(const FusedArgs &a, Smem *sm, float2 *mywin, float2 *dring, float *zring, int G,
                                          int widx, int u0, int f0, int lane, int nchunks, int *status)
{
    constexpr int nuw = NUW;
    using namespace pipe2;
    using GM = GeomNarrow;    /* ring geometry (S, DSTRIDE, ZSTRIDE) shared with costas_wave / flush_records */
    constexpr int DSTRIDE = GM::DSTRIDE;
    const int L = a.frame_size;
    const int fl = lane / QL, q = lane % QL;
    const bool simd0 = ((threadIdx.x >> 6) & 3) == 0;       /* a FIR wave placed beside the serial wave (layout) */
    const float4 *taps4 = reinterpret_cast<const float4 *>(sm->taps);
    const float2 *rd = mywin + fl * WS + (PAD + PADS) * q;   /* FIR read base: position PAD*q -> slot (PAD+PADS)*q */

    Unit un[NUW];
#pragma unroll
    for (int ui = 0; ui < NUW; ui++) {
        Unit &U = un[ui];
        U.u = u0 + ui;
        U.all_valid = UF * U.u + UF <= G && f0 + UF * U.u + UF <= a.nframes;
#pragma unroll
        for (int ff = 0; ff < UF; ff++) {
            const int fr = f0 + UF * U.u + ff;
            U.fv[ff] = UF * U.u + ff < G && fr < a.nframes;      /* an odd G leaves the last unit one frame */
            const int ix = a.index ? (U.fv[ff] ? a.index[fr] : 0) : a.fixed_index;   /* decimation offset, < C */
            U.ix[ff] = __builtin_amdgcn_readfirstlane(ix);
            U.src[ff] = reinterpret_cast<const float4 *>(a.x + (size_t)(U.fv[ff] ? fr : 0) * a.frame_pitch);
            U.hist[ff] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);       /* a fresh delay line (qpsk.c:37) */
        }
    }

    /* loads are UNCONDITIONAL (a per-load branch would make the compiler wait for each in turn): a chunk that lies
     * inside every frame of the unit loads straight; the tail chunk clamps the address and zeroes what lies past the
     * end.  L is even on this path (host-checked), so a 16-byte pair is either inside the frame or past it. */
    auto prefetch = [&](const Unit &U, float4 (&pre)[UF][NLD], int c) {
        if (U.all_valid && (c + 1) * CH <= L) {
#pragma unroll
            for (int ff = 0; ff < UF; ff++)
#pragma unroll
                for (int j = 0; j < NLD; j++)
                    pre[ff][j] = load_once(&U.src[ff][(c * CH + 128 * j + 2 * lane) >> 1]);
        } else {
#pragma unroll
            for (int ff = 0; ff < UF; ff++)
#pragma unroll
                for (int j = 0; j < NLD; j++) {
                    const int s = c * CH + 128 * j + 2 * lane;
                    const bool in = U.fv[ff] && s + 1 < L;
                    float4 v = load_once(&U.src[ff][in ? (s >> 1) : 0]);
                    if (!in) v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    pre[ff][j] = v;
                }
        }
    };

#ifdef QPSK_PIPE_PROFILE
    const bool prof = (a.dbg & 32) && blockIdx.x == 0;
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tlast = 0;
    auto tick = [&](int k) {
        if (prof) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (k >= 0) tacc[k] += t - tlast;
            tlast = t;
        }
    };
#else
    auto tick = [](int) {};
#endif

    /* one unit of one chunk: stage the window from registers, start the next loads, filter, flush what the loop has
     * finished with, hand over */
    auto run_unit = [&](Unit &U, float4 (&pre)[UF][NLD], int c, const Unit &next, bool has_next, int cnext) -> bool {
        /* Waves of a SIMD are served oldest first, so the older FIR waves would race two chunks ahead of the loop and
         * then sleep while the youngest, whose chunk the loop is waiting for, crawls along alone (a lone wave pays for
         * every LDS instruction and every dependent result itself).  Priority by need instead: the fewer chunks a
         * wave is ahead of the loop, the higher its priority, so the waves of a SIMD advance together and keep its
         * issue slots full.  (The serial wave runs at priority 3.) */
        if (a.dbg & 256) {
        } else if (simd0 && !(a.dbg & 1024)) {
            __builtin_amdgcn_s_setprio(3);
        } else {
            const int lead = c - __hip_atomic_load(&sm->consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lead <= 0) __builtin_amdgcn_s_setprio(2);
            else if (lead == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        /* history (the previous chunk's last block, one block below block 0: the lanes whose pair lies before
         * the first tap's reach are skipped), then this chunk's samples */
#pragma unroll
        for (int ff = 0; ff < UF; ff++) {
            const int p0 = 2 * lane + HIST - U.ix[ff];             /* window position of sample 2*lane of the chunk */
            float2 *w0 = mywin + ff * WS + slot_of(p0), *w1 = mywin + ff * WS + slot_of(p0 + 1);
            if ((U.ix[ff] & 1) == 0) {      /* wave-uniform: every lane holds a pair of frame ff here */
                if (p0 >= 128) *reinterpret_cast<float4 *>(w0 - BLK) = U.hist[ff];
#pragma unroll
                for (int j = 0; j < NLD; j++)
                    *reinterpret_cast<float4 *>(w0 + BLK * j) = pre[ff][j];
            } else {
                if (p0 >= 128) w0[-BLK] = make_float2(U.hist[ff].x, U.hist[ff].y);
                if (p0 + 1 >= 128) w1[-BLK] = make_float2(U.hist[ff].z, U.hist[ff].w);
#pragma unroll
                for (int j = 0; j < NLD; j++) {
                    w0[BLK * j] = make_float2(pre[ff][j].x, pre[ff][j].y);
                    w1[BLK * j] = make_float2(pre[ff][j].z, pre[ff][j].w);
                }
            }
            U.hist[ff] = pre[ff][NLD - 1];
        }
        if (has_next) prefetch(next, pre, cnext);
        tick(2);

        /* sliding-window FIR, the pinned two-symbol step of rx_fused_pipe_kernel: symbol r of this lane uses tap
         * k = t - C*r at window position PAD*q + t; taps 0..126 in order into one accumulator per symbol */
        v2f ac[R] = {v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}};
#if QPSK_PIPE2_ASM
        if (!QPSK_ABLATE(a, 1)) {
            /* the hand-scheduled stream (fir_r2_asm.h, generated by tools/gen_fir_asm.py): the sum below, same order */
            fir_r2_asm(lds_addr(rd), lds_addr(sm->taps), ac[0], ac[1]);
#else
        if (!QPSK_ABLATE(a, 1)) {
            /* Two symbols per lane leave a lone multiply/add pair per symbol and window position: too little
             * independent work for a wave that waits 8 cycles on every dependent result.  So positions go in PAIRS:
             * the four products of positions t, t+1 (two symbols each), then the four adds -- each accumulator still
             * receives its taps in order 0..126 (rrc_fir.c:22-26), an add follows its product by 4 instructions and
             * the previous add of its accumulator by 2.  Window values and tap groups are fetched from LDS DEPTH
             * blocks of 8 positions ahead of their use. */
            constexpr int NB = (TSTEPS + C - 1) / C, DEPTH = QPSK_PIPE2_DEPTH, NW = DEPTH + 1, NG = R + DEPTH;
            float tg[NG][C];
            float2 wv[NW][C];
            auto fetch_block = [&](int tb) {
                if (tb * C < NTAPS) {
                    const float4 ta = taps4[2 * tb], tb4 = taps4[2 * tb + 1];
                    float *g_ = tg[tb % NG];
                    g_[0] = ta.x; g_[1] = ta.y; g_[2] = ta.z; g_[3] = ta.w;
                    g_[4] = tb4.x; g_[5] = tb4.y; g_[6] = tb4.z; g_[7] = tb4.w;
                }
#pragma unroll
                for (int u = 0; u < C; u += 2) {
                    const int t = tb * C + u;      /* t even: positions t, t + 1 are one aligned 16-byte word */
                    if (t < TSTEPS) {
                        const float4 w4 = *reinterpret_cast<const float4 *>(rd + slot_of(t));
                        wv[tb % NW][u] = make_float2(w4.x, w4.y);
                        wv[tb % NW][u + 1] = make_float2(w4.z, w4.w);
                    }
                }
            };
            static_for<0, DEPTH>([&](auto d) { if (decltype(d)::value < NB) fetch_block(decltype(d)::value); });
            static_for<0, NB>([&](auto tbc) {
                constexpr int tb = decltype(tbc)::value;
                if (tb + DEPTH < NB) fetch_block(tb + DEPTH);
                static_for<0, C / 2>([&](auto uc) {
                    constexpr int u = 2 * decltype(uc)::value;
                    constexpr int t = tb * C + u;              /* positions t and t + 1 */
                    if constexpr (t < TSTEPS) {
                        constexpr bool a0 = t < NTAPS, a1 = t >= C && t - C < NTAPS;                  /* symbol 0 / 1 at t */
                        constexpr bool b0 = t + 1 < NTAPS, b1 = t + 1 >= C && t + 1 - C < NTAPS && t + 1 < TSTEPS;   /* at t + 1 */
                        const v2f va = v2f{wv[tb % NW][u].x, wv[tb % NW][u].y};
                        const v2f vb = v2f{wv[tb % NW][u + 1].x, wv[tb % NW][u + 1].y};
                        v2f pa0, pa1, pb0, pb1;    /* re*tap, im*tap (rrc_fir.c:24-25) */
                        if constexpr (a0) pa0 = va * tg[tb % NG][u];
                        if constexpr (a1) pa1 = va * tg[(tb - 1 + NG) % NG][u];
                        if constexpr (b0) pb0 = vb * tg[tb % NG][u + 1];
                        if constexpr (b1) pb1 = vb * tg[(tb - 1 + NG) % NG][u + 1];
                        if constexpr (a0 && a1 && b0 && b1) {
                            asm volatile("" : "+v"(pa0), "+v"(pa1), "+v"(pb0), "+v"(pb1));
                        } else {
                            if constexpr (a0) asm volatile("" : "+v"(pa0));
                            if constexpr (a1) asm volatile("" : "+v"(pa1));
                            if constexpr (b0) asm volatile("" : "+v"(pb0));
                            if constexpr (b1) asm volatile("" : "+v"(pb1));
                        }
                        if constexpr (a0) ac[0] = ac[0] + pa0;
                        if constexpr (a1) ac[1] = ac[1] + pa1;
                        if constexpr (b0) ac[0] = ac[0] + pb0;
                        if constexpr (b1) ac[1] = ac[1] + pb1;
                        asm volatile("" : "+v"(ac[0]), "+v"(ac[1]));
                    }
                });
            });
#endif
        } else {
            ac[0] = v2f{0.7f, 0.3f}; ac[1] = v2f{0.7f, 0.3f};
        }
        tick(3);
        /* ring slot c % DR is free once chunk c - DR has been consumed, and that chunk's symbols leave first */
        const int g = UF * U.u + fl, frame = f0 + g;
        const bool mine = g < G && frame < a.nframes;     /* this lane's frame exists (rings have G rows) */
        if (c >= DR) {
            if (!wait_ge(&sm->consumed, c - DR + 1, &sm->abort_flag)) return false;
            tick(0);
            /* the flush's addresses are recomputed here every time (the asm statement hides that g never changes):
             * hoisted out of the chunk loop they are spilled, and a scratch reload waits for every load issued before
             * it -- the next unit's samples, just requested (vmcnt counts in order) */
            int gf = g, qf = q;
            asm volatile("" : "+v"(gf), "+v"(qf));
            if (mine) flush_records<GM, R>(a, zring, dring, gf, f0 + gf, qf, c - DR);
            tick(1);
        }
        if (g < G) {   /* the lane's two symbols are one aligned 16-byte word of the ring (rows are 16-byte aligned, R q even) */
            const float2 s0 = fir_gain(make_float2(ac[0].x, ac[0].y)), s1 = fir_gain(make_float2(ac[1].x, ac[1].y));
            int gr = g, qr = q;     /* as above: the ring address is cheaper to recompute than to reload */
            asm volatile("" : "+v"(gr), "+v"(qr));
            *reinterpret_cast<float4 *>(dring + (size_t)gr * DSTRIDE + (c % DR) * S + R * qr) = make_float4(s0.x, s0.y, s1.x, s1.y);
        }
        if (lane == 0) st_release(&sm->ready[U.u], c + 1);
        tick(4);
        return true;
    };

    float4 pre[UF][NLD];
    prefetch(un[0], pre, 0);
    bool ok = true;
    tick(-1);
    for (int c = 0; c < nchunks && ok; c++) {
        static_for<0, NUW>([&](auto uic) {
            constexpr int ui = decltype(uic)::value, nx = (ui + 1) % NUW;
            constexpr bool last_unit = ui + 1 == NUW;
            if (ok) ok = run_unit(un[ui], pre, c, un[nx], !last_unit || c + 1 < nchunks, last_unit ? c + 1 : c);
        });
    }
    if (ok) {
        ok = wait_ge(&sm->consumed, nchunks, &sm->abort_flag);
        if (ok)
#pragma unroll
            for (int ui = 0; ui < NUW; ui++) {
                const int g = UF * (u0 + ui) + fl, frame = f0 + g;
                for (int c = max(0, nchunks - DR); c < nchunks; c++)
                    if (g < G && frame < a.nframes) flush_records<GM, R>(a, zring, dring, g, frame, q, c);
            }
    }
    if (!ok && lane == 0) report_status(status, STATUS_PIPE_TIMEOUT);
#ifdef QPSK_PIPE_PROFILE
    if (prof && lane == 0)
        printf("FIR wave %d (%d units): %d chunks; cycles per chunk: wait for the loop %llu, flush %llu, window staging %llu, "
               "filter %llu, ring hand-over %llu\n", widx, nuw, nchunks, tacc[0] / nchunks, tacc[1] / nchunks, tacc[2] / nchunks,
               tacc[3] / nchunks, tacc[4] / nchunks);
#endif
}

__global__ void __launch_bounds__(pipe2::MAX_THREADS)
rx_pipe2_kernel(FusedArgs a, unsigned long long layout, int nwin, int *status)
{
    using namespace pipe2;
    using GM = GeomNarrow;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem *sm = reinterpret_cast<Smem *>(smem_raw);
    const int G = a.G;                                   /* frames of a workgroup */
    float2 *win = reinterpret_cast<float2 *>(smem_raw + sizeof(Smem));      /* [nwin][UF][WS]: one window per FIR wave */
    float2 *dring = win + (size_t)nwin * UF * WS;                            /* [G][DSTRIDE] */
    float *zring = reinterpret_cast<float *>(dring + (size_t)G * GM::DSTRIDE);   /* [G*nbw][ZSTRIDE] */
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f0 = blockIdx.x * G;
    const int nchunks = (a.nsym + S - 1) / S;

    for (int i = tid; i < 128; i += blockDim.x)
        sm->taps[i] = i < NTAPS ? a.taps[i] : 0.0f;
    if (tid < MAX_WAVES) sm->ready[tid] = 0;
    if (tid == 0) { sm->consumed = 0; sm->abort_flag = 0; }
    __syncthreads();

    if (wave == 0) {
        costas_wave<GM>(a, sm, dring, zring, G, f0, lane, nchunks, status);   /* a.mixed == 2: lane g waits on ready[g / 2] */
        return;
    }
    /* layout: 4 bits per hardware wave = the units it owns (consecutive units, in wave order); 0 = the wave retires at
     * once (the host leaves the serial wave's SIMD -- waves 4, 8 -- empty unless asked otherwise) */
    const int mine = (int)((layout >> (4 * wave)) & 15);
    if (mine == 0) return;
    int u0 = 0, widx = 0;
    for (int v = 1; v < wave; v++) {
        const int cv = (int)((layout >> (4 * v)) & 15);
        u0 += cv;
        widx += cv != 0;
    }
    float2 *mywin = win + (size_t)widx * UF * WS;
    if (mine == 2)
        fir_wave2<2>(a, sm, mywin, dring, zring, G, widx, u0, f0, lane, nchunks, status);
    else
        fir_wave2<1>(a, sm, mywin, dring, zring, G, widx, u0, f0, lane, nchunks, status);
}

size_t pipe2_lds_bytes(int G, int nwin, int nbw)
{
    using GM = GeomNarrow;
    size_t b = sizeof(Smem) + sizeof(float2) * ((size_t)nwin * pipe2::UF * pipe2::WS + (size_t)G * GM::DSTRIDE) +
               sizeof(float) * (size_t)G * nbw * GM::ZSTRIDE;
    return (b + 15) & ~(size_t)15;
}

int pipe2_max_fir(void) { return pipe2::MAX_FIR; }
int pipe2_max_frames(void) { return pipe2::UF * pipe2::MAX_UNITS; }
int pipe2_max_units_per_wave(void) { return pipe2::MAX_UW; }
int pipe2_max_hw_waves(void) { return pipe2::MAX_THREADS / 64; }

/*
 * The default layout for NU units [measured, DESIGN.md 4.1].
 *   NU <= 8 (up to 16 frames per workgroup: the recurrence is the limit): one unit per wave on hardware waves 1-3,
 *     5-7, 9, 10 -- SIMDs 1-3; waves 4 and 8 would sit beside the serial wave and retire at once.
 *   NU > 8 (the filter is the limit): the serial wave leaves three quarters of its SIMD's issue slots unused, so
 *     hardware wave 4 filters too: one unit each on waves 1-7, 9, 10, then a second one on waves 1, 2, ... -- a full
 *     workgroup's 16 units sit 2, 5, 5, 4 on SIMDs 0-3 (nine windows are what the LDS holds beside 32 frames' rings).
 *     The kernel then runs the serial wave at priority 1 and the FIR wave beside it at 3 (see there).
 */
unsigned long long pipe2_default_layout(int NU)
{
    using namespace pipe2;
    int cnt[16] = {0};
    const int hwmax = MAX_THREADS / 64;
    const bool share = NU > 8 && MAX_UW == 2;
    int left = NU;
    for (int round = 0; round < MAX_UW && left > 0; round++)
        for (int w = 1; w < hwmax && left > 0; w++) {
            if ((w & 3) == 0 && !(share && w == 4)) continue;
            if (share && (w == 11 || (round == 1 && w > 7))) continue;
            cnt[w]++;
            left--;
        }
    unsigned long long layout = 0;
    for (int w = 1; w < hwmax; w++) layout |= (unsigned long long)cnt[w] << (4 * w);
    return left == 0 ? layout : 0;
}

/* G frames per workgroup (at most 32, G * nbw <= 64); layout as in the kernel: its counts must add up to the units */
int launch_rx_pipe2(const FusedArgs &a0, int G, unsigned long long layout, int *status, hipStream_t s)
{
    using namespace pipe2;
    if (a0.est_tw) return (int)hipErrorInvalidValue;   /* no in-launch timing estimate in this kernel: it would use fixed_index */
    FusedArgs a = a0;
    const int NU = (G + UF - 1) / UF;
    int units = 0, nwin = 0, hw = 1;
    for (int w = 1; w < 16; w++) {
        const int cw = (int)((layout >> (4 * w)) & 15);
        if (cw > MAX_UW || (cw && w >= MAX_THREADS / 64)) return (int)hipErrorInvalidValue;
        units += cw;
        nwin += cw != 0;
        if (cw) hw = w + 1;
    }
    if (G < 1 || G > UF * MAX_UNITS || G * a.nbw > 64 || units != NU || (layout & 15) ||
        pipe2_lds_bytes(G, nwin, a.nbw) > (size_t)MAX_LDS_BYTES)
        return (int)hipErrorInvalidValue;
    a.G = G;
    a.lean_pair = 0;                                      /* one lane per loop: the paired stream is rx_lean_kernel's */
    a.mixed = 2;                                          /* the serial wave's lane for frame g waits on ready[g / 2] */
    a.share_simd0 = ((layout >> 16) & 15) != 0 || ((layout >> 32) & 15) != 0;   /* hardware waves 4, 8 */
    const int blocks = (a.nframes + G - 1) / G;
    hipLaunchKernelGGL(rx_pipe2_kernel, dim3(blocks), dim3(64 * hw), pipe2_lds_bytes(G, nwin, a.nbw), s, a, layout, nwin, status);
    return (int)hipGetLastError();
}

/* ========================================================================
 * rx_lean_kernel: rx_pipe2_kernel's pipeline (two-frame units, per-wave windows, the same rings, counters and serial
 * wave) with the FIR waves' WHOLE chunk loop as one hand-written instruction stream (fir_lean_asm.h, generated by
 * tools/gen_lean_asm.py): ~600 vector instructions per unit and chunk where the compiler's version of the same work
 * issued 700-800 (508 are the filter), the 64 distinct taps of the symmetric filter in SGPRs (no tap reads: a third
 * of the LDS instructions gone), no fence anywhere -- a hand-over is two LDS writes in program order, so a wave never
 * waits for its symbol stores or for the prefetched samples of its next unit -- and one counted vmcnt wait per unit.
 * What it serves (launch_rx_lean checks; everything else stays with the kernels above): CYCLES = 8, frames of whole
 * 64-symbol chunks (at least two), one loop per frame from a fresh state, no costas_frame[] dump, a symmetric filter,
 * 16-byte aligned frames, workgroups of an even number of frames (the batch's last one may be partly pad: round 6).  Results: the same bits.
 * ======================================================================== */
namespace lean {
/*
 * LDS of rx_lean_kernel (160 KB, filled to the last 200 bytes so that TEN waves can filter: one beside the serial wave
 * with a single unit, three on each of the other SIMDs with 2 + 2 + 1 -- 1, 5, 5, 5 units on SIMDs 0-3):
 *   Ctl                      counters (80 bytes; no taps: the stream keeps them in SGPRs)
 *   windows [nwin][2][WS]    WS = 712 slots per frame: positions 0..633 of the padded image.  A lane reads up to position
 *                            630; the chunk's last samples (up to 637) matter only as the next chunk's history, which lives
 *                            in registers -- the stream drops them (fir_lean_asm.h, stage)
 *   ring rows [rows][ROW]    one row per frame: 128 symbols (1024 bytes), 128 records (512 bytes), 16 bytes of padding: rows
 *                            1552 bytes = 4 banks (mod 64) apart for the serial wave's 16-byte reads and writes, like the
 *                            separate rings of the other kernels (1040- and 528-byte rows) at 16 bytes less per frame.
 *                            rows = max(G, hardware waves): the record half of row w carries hardware wave w's parameter
 *                            block until the first record is written -- which the serial wave does only once EVERY unit has
 *                            handed over its first chunk, i.e. after every FIR wave has read its parameters.
 */
constexpr int PRM_DWORDS = 20;                        /* per-wave parameter block of the stream */
constexpr int HW_WAVES = pipe2::MAX_THREADS / 64;
constexpr int WS = WS_LEAN_SLOTS;
struct Ctl {
    int ready[MAX_WAVES];     /* chunks produced, per unit */
    int consumed;             /* chunks consumed by the serial wave */
    int abort_flag;
    int pad_[2];
};
struct GeomLean : GeomNarrow {                        /* ring geometry seen by costas_wave / flush_records */
    static constexpr int DSTRIDE = 194;               /* float2 slots per row: 1552 bytes */
    static constexpr int ZSTRIDE = 388;               /* floats per row */
};
constexpr int ROW_BYTES = 1552, Z_OFFSET_BYTES = 1024;
static_assert(GeomLean::DSTRIDE * 8 == ROW_BYTES && GeomLean::ZSTRIDE * 4 == ROW_BYTES &&
              Z_OFFSET_BYTES == DR * GeomNarrow::S * 8 && Z_OFFSET_BYTES + DR * GeomNarrow::S * 4 <= ROW_BYTES &&
              ROW_BYTES % 16 == 0 && (ROW_BYTES / 4) % 64 == 4, "ring rows: symbols, records, bank spread");
static_assert(sizeof(Ctl) % 16 == 0 && offsetof(Ctl, ready) == 0 && offsetof(Ctl, consumed) == 64 &&
              offsetof(Ctl, abort_flag) == 68, "fir_lean_asm.h addresses the counters by these offsets");
static_assert(PRM_DWORDS * 4 <= ROW_BYTES - Z_OFFSET_BYTES, "a parameter block fits the record half of a row");
static_assert(FIR_LEAN_END_VGPR <= 168, "three waves per SIMD");
static_assert(FIR_LEAN_WOFF == sizeof(float2) * pipe2::UF * WS, "fir_lean_loop2_dma2w: unit 1's window follows unit 0's");
static_assert(WS % 2 == 0 && pipe2::slot_of(pipe2::PAD * (pipe2::QL - 1) + pipe2::TSTEPS - 1) < WS &&
              pipe2::slot_of(pipe2::CH + HIST - 1) < WS + 16, "window: every position a lane reads, and a frame's spill stays in the next frame's history slots");
__device__ __host__ constexpr int rows_of(int G) { return G > HW_WAVES ? G : HW_WAVES; }
} // namespace lean
constexpr size_t LEAN_EST_LDS_BYTES = 128 * sizeof(float) + 32 * sizeof(int);   /* in-launch FFT estimate: taps, the workgroup's indices */

/* the two symbol rows of unit u of the workgroup whose first frame is f0: the caller's, or -- a unit that reaches past the batch's last
 * frame (the last workgroup of a ragged batch) -- its two rows of the launch's pad buffer (a.sym_pad: G rows; api.cpp copies the one real
 * row of a unit that straddles the end, i.e. the last frame of an odd batch, to its place behind the launch) */
__device__ __forceinline__ uint8_t *lean_unit_rows(const FusedArgs &a, int f0, int u)
{
    const int fa = f0 + pipe2::UF * u;
    return fa + pipe2::UF <= a.nframes ? a.sym + (size_t)fa * a.nsym : a.sym_pad + (size_t)(pipe2::UF * u) * a.nsym;
}

template <int NUW>
__device__ __forceinline__ void fir_wave_lean(const FusedArgs &a, lean::Ctl *sm, unsigned char *rows, float2 *mywin,
                                              int hwave, int u0, int f0, int lane, int nchunks, int *status, const int *est)
{
    using namespace pipe2;
    using GM = lean::GeomLean;
    constexpr int WS = lean::WS;
    float2 *dring = reinterpret_cast<float2 *>(rows);
    float *zring = reinterpret_cast<float *>(rows + lean::Z_OFFSET_BYTES);
    const int N = a.nsym;
    const int fl = lane / QL, q = lane % QL;
    const bool simd0 = (hwave & 3) == 0;                 /* placed beside the serial wave (layout): keeps priority 3 */
    unsigned *prm = reinterpret_cast<unsigned *>(rows + (size_t)hwave * lean::ROW_BYTES + lean::Z_OFFSET_BYTES);
    unsigned ixpack = ((unsigned)(4 * u0) << 16) | (simd0 && !(a.dbg & 1024) ? 0x80000000u : 0u);
#pragma unroll
    for (int ui = 0; ui < NUW; ui++)
#pragma unroll
        for (int ff = 0; ff < UF; ff++) {
            const int fr = f0 + UF * (u0 + ui) + ff;
            /* decimation offset, < C: the estimate made inside this launch (est[], in LDS), the caller's array, or the fixed one */
            const int ix = __builtin_amdgcn_readfirstlane(est ? est[UF * (u0 + ui) + ff] : a.index ? a.index[min(fr, a.nframes - 1)] : a.fixed_index) & 7;
            ixpack |= (unsigned)ix << (4 * (2 * ui + ff));
        }
    /* which stream this wave runs (wave-uniform): LDS-DMA staging needs even offsets (16-byte pairs); a two-unit wave that stages by
     * DMA uses the window per unit the launch has set aside (lean_twowin), every other wave the first of its windows only */
    const bool use_dma = a.lean_dma && (ixpack & 0x1111u) == 0;
    const bool two_windows = NUW == 2 && a.lean_twowin && use_dma;
    LeanLaneAddr w;
#pragma unroll
    for (int ui = 0; ui < 2; ui++) {
        const int u = u0 + (ui < NUW ? ui : 0);
#pragma unroll
        for (int ff = 0; ff < UF; ff++) {
            const int fr = f0 + UF * u + ff;
            const int ix = (int)((ixpack >> (4 * (2 * (ui < NUW ? ui : 0) + ff))) & 7u);
            const int p0 = 2 * lane + HIST - ix;          /* window position of sample 2*lane of a chunk */
            const int wu = (two_windows && ui == 1) ? UF * WS : 0;      /* unit 1's own window, FIR_LEAN_WOFF bytes up */
            w.wr0[ui][ff] = lds_addr(mywin + wu + ff * WS + slot_of(p0)) - 8u * BLK;
            w.wr1[ui][ff] = lds_addr(mywin + wu + ff * WS + slot_of(p0 + 1)) - 8u * BLK;
            if (lane == 0) {
                /* a batch's last workgroup may have fewer than G frames: its pad frames read the batch's last frame (valid memory) and
                 * their symbols go to the launch's pad rows (below); their lanes of the serial wave are switched off */
                size_t fsrc = (size_t)min(fr, a.nframes - 1);
#ifdef QPSK_PIPE_PROFILE
                /* measurement build, dbg bit 21: a workgroup's frames a grid apart instead of adjacent (WRONG frame -> output mapping;
                 * the visit order of the batch in memory is what is measured) */
                if (a.dbg & 2097152) fsrc = (size_t)(UF * u + ff) * gridDim.x + blockIdx.x;
#endif
                unsigned long long src = (unsigned long long)(a.x + fsrc * a.frame_pitch);
#ifdef QPSK_PIPE_PROFILE
                /* measurement build, dbg bit 22: the workgroup's frames as one contiguous tile per chunk (see gen_lean_asm.py, "tiled") */
                if (a.dbg & 4194304) src = (unsigned long long)(a.x + (size_t)f0 * a.frame_pitch) + (unsigned long long)(UF * u + ff) * 4096ull;
#endif
                prm[4 * ui + 2 * ff] = (unsigned)src;
                prm[4 * ui + 2 * ff + 1] = (unsigned)(src >> 32);
            }
        }
        const int g = UF * u + fl;
        w.ring[ui] = lds_addr(dring + (size_t)g * GM::DSTRIDE + R * q);
        w.z[ui] = lds_addr(zring + (size_t)g * GM::ZSTRIDE + R * q);
        if (lane == 0) {
            const unsigned long long sb = (unsigned long long)lean_unit_rows(a, f0, u);
            prm[8 + 2 * ui] = (unsigned)sb;
            prm[8 + 2 * ui + 1] = (unsigned)(sb >> 32);
        }
    }
    w.wlim = lds_addr(mywin + WS + (WS - 2));            /* the second frame's last pair */
    w.wpad = lds_addr(mywin + WS + PAD);                 /* its first pad pair (slots 16, 17): never read */
    if (lane == 0) {
        prm[12] = (unsigned)nchunks;
        prm[13] = ixpack;
        prm[16] = (unsigned)(unsigned long long)a.taps;
        prm[17] = (unsigned)((unsigned long long)a.taps >> 32);
    }
    if (simd0 && !(a.dbg & 1024)) __builtin_amdgcn_s_setprio(3);
    const unsigned rd = lds_addr(mywin + fl * WS + (PAD + PADS) * q);
    int st;
#ifdef QPSK_PIPE_PROFILE
    if (a.dbg & 32) {      /* measurement build: the stream with cycle stamps between the phases of a unit */
        unsigned pf[FIR_LEAN_NPROF];
        unsigned long long t0, t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        if constexpr (NUW == 2)
            st = fir_lean_loop2_prof(lds_addr(prm), rd, 16u * lane, (unsigned)(fl * N + R * q), lds_addr(sm), w, pf);
        else
            st = fir_lean_loop1_prof(lds_addr(prm), rd, 16u * lane, (unsigned)(fl * N + R * q), lds_addr(sm), w, pf);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if ((blockIdx.x == 0 || blockIdx.x == 77) && lane == 0)
            printf("wg %3d FIR wave hw %2d (%d units): %d chunks, %llu cycles in the stream; per chunk: samples %u, stage+loads %u, "
                   "filter+gain %u, wait for the loop %u, flush %u, hand-over %u\n", (int)blockIdx.x, hwave, NUW, nchunks, t1 - t0,
                   pf[0] / nchunks, pf[1] / nchunks, pf[2] / nchunks, pf[3] / nchunks, pf[4] / nchunks, pf[5] / nchunks);
    } else if (a.dbg & (1 | 16384 | 32768 | 65536 | 262144 | 524288 | 1048576 | 4194304)) {
        /* measurement build: streams with a part of the work left out (WRONG results): 1 the filter's multiplies and adds,
         * 16384 its window reads, 32768 the flush's arithmetic, 65536 the window staging writes, 262144 the symbol stores, 524288 the
         * hand-over write of the symbols -- what each costs in time and in energy at the board's power limit (one at a time: the
         * first bit set wins); 1048576: the unit's loads issued frame-alternating (right results) */
#define QPSK_LEAN_ABLATED(SFX)                                                                                          \
        (NUW == 2 ? fir_lean_loop2_##SFX(lds_addr(prm), rd, 16u * lane, (unsigned)(fl * N + R * q), lds_addr(sm), w)    \
                  : fir_lean_loop1_##SFX(lds_addr(prm), rd, 16u * lane, (unsigned)(fl * N + R * q), lds_addr(sm), w))
        if (a.dbg & 1) st = QPSK_LEAN_ABLATED(avalu);
        else if (a.dbg & 16384) st = QPSK_LEAN_ABLATED(alds);
        else if (a.dbg & 32768) st = QPSK_LEAN_ABLATED(aflush);
        else if (a.dbg & 65536) st = QPSK_LEAN_ABLATED(astage);
        else if (a.dbg & 262144) st = QPSK_LEAN_ABLATED(astore);
        else if (a.dbg & 524288) st = QPSK_LEAN_ABLATED(aring);
        else if (a.dbg & 4194304) st = QPSK_LEAN_ABLATED(atiled);
        else st = QPSK_LEAN_ABLATED(aorder);
#undef QPSK_LEAN_ABLATED
    } else
#endif
#ifdef QPSK_PIPE_PROFILE
    QPSK_TL_STAMP(32 + hwave);     /* in front of the stream (parameters written, not yet the DMA table) */
#endif
    if (use_dma) {
        /* Window staging by LDS-DMA (fir_lean_asm.h, the _dma loops): every decimation offset of the wave's frames is even, so the
         * 16-byte pairs of the window image are 16-byte pairs of the input.  The table the stream reads first, left in the still
         * unused window: per frame k = 2 unit + frame the four per-lane source byte offsets of its DMAs -- DMA j's lane l fills
         * window slots S0 + 128 j + 2 l, + 1 of the padded image (slot_of), S0 = the slot of the chunk's first sample; a lane whose
         * pair is a pad pair re-reads its neighbour's samples (the same 16 bytes: no extra traffic) -- and the LDS byte address of S0. */
        unsigned *tab = reinterpret_cast<unsigned *>(mywin);
#pragma unroll
        for (int ui = 0; ui < NUW; ui++)
#pragma unroll
            for (int ff = 0; ff < UF; ff++) {
                const int k = 2 * ui + ff;
                const int ix = (int)((ixpack >> (4 * k)) & 15u);
                const int S0 = slot_of(HIST - ix);
                for (int j = 0; j < 4; j++) {
                    const int sl = S0 + 2 * (64 * j + lane);
                    const int grp = sl / (PAD + PADS);
                    int r = sl % (PAD + PADS);
                    if (r >= PAD) r = PAD - 2;
                    int n = PAD * grp + r - HIST + ix;
                    if (n > CH - 2) n = CH - 2;          /* lanes the last DMA masks off */
                    tab[(4 * k + j) * 64 + lane] = 8u * (unsigned)n;
                }
                tab[(16 + k) * 64 + lane] = lds_addr(mywin + ((two_windows && ui == 1) ? UF * WS : 0) + ff * WS + S0);
            }
        if constexpr (NUW == 2) {
            if (two_windows)
                st = fir_lean_loop2_dma2w(lds_addr(prm), rd, 16u * lane, (unsigned)(fl * N + R * q), lds_addr(sm), lds_addr(tab + lane), w);
            else
                st = fir_lean_loop2_dma(lds_addr(prm), rd, 16u * lane, (unsigned)(fl * N + R * q), lds_addr(sm), lds_addr(tab + lane), w);
        } else
            st = fir_lean_loop1_dma(lds_addr(prm), rd, 16u * lane, (unsigned)(fl * N + R * q), lds_addr(sm), lds_addr(tab + lane), w);
    } else if constexpr (NUW == 2)
        st = fir_lean_loop2(lds_addr(prm), rd, 16u * lane, (unsigned)(fl * N + R * q), lds_addr(sm), w);
    else
        st = fir_lean_loop1(lds_addr(prm), rd, 16u * lane, (unsigned)(fl * N + R * q), lds_addr(sm), w);
    bool ok = st == 0;
    if (ok) {      /* the last two chunks leave as the loop finishes each (the first while it still steps through the second) */
        for (int c = nchunks - DR; c < nchunks && ok; c++) {
            ok = wait_ge(&sm->consumed, c + 1, &sm->abort_flag);
            if (ok)
#pragma unroll
                for (int ui = 0; ui < NUW; ui++) {
                    const int g = UF * (u0 + ui) + fl;
                    flush_records<GM, R>(a, zring, dring, g, f0 + g, q, c, lean_unit_rows(a, f0, u0 + ui) + (size_t)fl * N);
                }
        }
    } else if (lane == 0) {
        __hip_atomic_store(&sm->abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#ifdef QPSK_PIPE_PROFILE
    if ((a.dbg & 8388608) && lane == 0) {
        unsigned long long t;
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (blockIdx.x < 1024 && hwave < 11) g_timeline[48 * blockIdx.x + 5 + hwave] = t;
    }
#endif
    if (!ok && lane == 0) report_status(status, STATUS_PIPE_TIMEOUT);
}

__global__ void __launch_bounds__(pipe2::MAX_THREADS)
rx_lean_kernel(FusedArgs a, unsigned long long layout, int nwin, int *status)
{
    using namespace pipe2;
    using GM = lean::GeomLean;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lean::Ctl *sm = reinterpret_cast<lean::Ctl *>(smem_raw);
    const int G = a.G;
    float2 *win = reinterpret_cast<float2 *>(smem_raw + sizeof(lean::Ctl));          /* [nwin][UF][WS] */
    unsigned char *rows = reinterpret_cast<unsigned char *>(win + (size_t)nwin * UF * lean::WS);   /* [rows_of(G)][ROW_BYTES] */
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f0 = blockIdx.x * G;
    const int nchunks = a.nsym / S;

#ifdef QPSK_PIPE_PROFILE
    if ((a.dbg & 8388608) && tid == 0) {
        unsigned long long t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (blockIdx.x < 1024) g_timeline[48 * blockIdx.x] = t;
    }
#endif
    if (tid < MAX_WAVES) sm->ready[tid] = 0;
    if (tid == 0) { sm->consumed = 0; sm->abort_flag = 0; }
    /* ---- BASELINE config 3: the FFT timing estimate inside the launch (timing_fft_wave.h), as in rx_fused_pipe_kernel: every hardware
     * wave of the workgroup -- the serial wave and waves that own no unit included, all idle until the first chunk exists -- estimates
     * frames wave, wave + nwe, ... of the workgroup (at most MAX_FPW each): one pass of the full-rate stream fir_full8_asm.h over 512
     * samples, the symbol-rate bin of fft.c's transform, the argmax rule; its window lies in the (not yet used) frame windows, the taps
     * and the indices in 640 bytes behind the ring rows.  No launch of its own, no drain and refill of the chip in between. */
    const int *est = nullptr;
    if (a.est_tw) {
        float *etaps = reinterpret_cast<float *>(rows + (size_t)lean::rows_of(G) * lean::ROW_BYTES);
        int *eidx = reinterpret_cast<int *>(etaps + 128);
        for (int i = tid; i < 128; i += blockDim.x) etaps[i] = i < NTAPS ? a.taps[i] : 0.0f;
        __syncthreads();
        const int nw = (int)blockDim.x / 64;
        const int room = (int)((size_t)nwin * UF * lean::WS / tfft::WSLOTS);       /* estimator windows the frame windows hold */
        const int nwe = nw < room ? nw : room;
        if (wave < nwe) {
            float2 *ewin = win + (size_t)wave * tfft::WSLOTS;
            const unsigned erd = lds_addr(ewin + (tfft::R + tfft::PADS) * lane), etap = lds_addr(etaps);
            tfft::Pre5 pre = tfft::load_frame5(a.x + (size_t)min(f0 + wave, a.nframes - 1) * a.frame_pitch, lane);
            for (int fe = wave; fe < G; fe += nwe) {
                tfft::stage_frame5(ewin, lane, pre);
                if (fe + nwe < G) pre = tfft::load_frame5(a.x + (size_t)min(f0 + fe + nwe, a.nframes - 1) * a.frame_pitch, lane);
                tfft::wave_sync();
                v2f e0, e1, e2, e3, e4, e5, e6, e7;
                fir_full8_asm(erd, etap, e0, e1, e2, e3, e4, e5, e6, e7);
                const v2f eacc[tfft::R] = {e0, e1, e2, e3, e4, e5, e6, e7};
                double epv[tfft::R];
                const tfft::cd u = tfft::power_bin(eacc, ewin, lane, a.est_tw, tfft::NFFT / C, nullptr, epv);
                if (lane == 0) {
                    const int best = tfft::pick_index(u, a.est_cs, C, nullptr);
                    eidx[fe] = best;
                    if (a.index_out && f0 + fe < a.nframes) a.index_out[f0 + fe] = best;
                }
                tfft::wave_sync();
            }
        }
        est = eidx;
    }
    __syncthreads();

    if (wave == 0) {   /* a.mixed == 2: lane g waits on ready[g / 2] */
        /* a ragged batch's last workgroup: the serial wave's lanes of the pad frames are off, so nothing makes it wait for THEIR units --
         * and the record half of row w is hardware wave w's parameter block until that wave has read it (lean::, above), which the
         * first hand-over of every unit proves: wait for the pad units' too before the first record is written */
        if (f0 + G > a.nframes) {
            bool ok = true;
            for (int u = (a.nframes - f0 + UF - 1) / UF; u < G / UF && ok; u++) ok = wait_ge(&sm->ready[u], 1, &sm->abort_flag);
            if (!ok) {
                if (lane == 0) report_status(status, STATUS_PIPE_TIMEOUT);
                return;
            }
        }
        costas_wave<GM>(a, sm, reinterpret_cast<const float2 *>(rows), reinterpret_cast<float *>(rows + lean::Z_OFFSET_BYTES), G,
                        f0, lane, nchunks, status);
        return;
    }
    const int mine = wave < 16 ? (int)((layout >> (4 * wave)) & 15) : 0;
    if (mine == 0) return;
#ifdef QPSK_PIPE_PROFILE
    QPSK_TL_STAMP(16 + wave);      /* a FIR wave behind the workgroup's barrier */
#endif
    int u0 = 0, widx = 0;
    for (int v = 1; v < wave; v++) {
        const int cv = (int)((layout >> (4 * v)) & 15);
        u0 += cv;
        widx += a.lean_twowin ? cv : (cv != 0);      /* a window per unit, or one per FIR wave */
    }
    float2 *mywin = win + (size_t)widx * UF * lean::WS;
    if (mine == 2)
        fir_wave_lean<2>(a, sm, rows, mywin, wave, u0, f0, lane, nchunks, status, est);
    else
        fir_wave_lean<1>(a, sm, rows, mywin, wave, u0, f0, lane, nchunks, status, est);
}

size_t lean_lds_bytes(int G, int nwin)
{
    size_t b = sizeof(lean::Ctl) + sizeof(float2) * (size_t)nwin * pipe2::UF * lean::WS + (size_t)lean::rows_of(G) * lean::ROW_BYTES;
    return (b + 15) & ~(size_t)15;
}

/*
 * rx_lean_kernel's layout for NU units (4 bits per hardware wave, as rx_pipe2_kernel's).  A full workgroup (16 units): ONE
 * unit beside the serial wave (hardware wave 4) and 2 + 2 + 1 on the three waves of each other SIMD -- 1, 5, 5, 5.  [Measured,
 * every vector instruction of a wave on SIMD 0 costs the serial wave its 4 cycles, whatever the priorities;
 * with two units there the serial wave takes 239 cycles per step instead of 162 and is the slowest of the workgroup, with
 * one 191, and three SIMDs with five units each take as long.]  Fewer units: rx_pipe2_kernel's layouts.
 */
unsigned long long lean_default_layout(int NU)
{
    if (NU == 8) {
        /* 16 frames per workgroup (BASELINE config 2): the serial wave alone on SIMD 0; 3, 3, 2 units on SIMDs 1-3 as two-unit
         * waves 1, 2, 3 and one-unit waves 5, 6 -- five FIR waves, not eight: a two-unit wave covers the LDS-DMA latency of one unit
         * with the filter of the other.  [measured, profiles/r05_config2_lean.txt, one process: 0.1532 ms against 0.1578 with eight
         * one-unit waves and 0.1621 for rx_fused_pipe_kernel; a unit beside the serial wave: 0.1797] */
        static const int c8[12] = {0, 2, 2, 2, 0, 1, 1, 0, 0, 0, 0, 0};
        unsigned long long l8 = 0;
        for (int w = 1; w < 12; w++) l8 |= (unsigned long long)c8[w] << (4 * w);
        return l8;
    }
    if (NU != pipe2::MAX_UNITS) return pipe2_default_layout(NU);
    static const int cnt[12] = {0, 2, 2, 2, 1, 2, 2, 2, 0, 1, 1, 1};
    unsigned long long layout = 0;
    for (int w = 1; w < 12; w++) layout |= (unsigned long long)cnt[w] << (4 * w);
    return layout;
}

/* what rx_lean_kernel serves (see its header); the caller has checked CYCLES = 8, the alignment and the filter's symmetry */
bool lean_shape_ok(const FusedArgs &a, int G)
{
    return a.nbw == 1 && !a.costas && !a.state_in && !a.state_out && a.nsym % pipe2::S == 0 && a.nsym >= 2 * pipe2::S &&
           a.frame_size == a.nsym * C && G >= 2 && G % 2 == 0 && G <= pipe2::UF * pipe2::MAX_UNITS && a.nframes >= 1 &&
           (a.nframes % G == 0 || a.sym_pad) && !(a.dbg & (8 | 16));
}

/*
 * rx_lean_kernel's launch geometry for G frames per workgroup under `layout`, derived in ONE place (round 5 derived units / windows /
 * estimator waves / LDS twice, in lean_est_ok and in launch_rx_lean: VERDICT r5): the question "does the FFT timing estimate fit inside
 * the launch" and the launch itself read the same numbers.
 */
struct LeanGeometry {
    bool ok;          /* layout and LDS are valid for G */
    bool est_ok;      /* ... and the in-launch FFT timing estimate fits beside them */
    int units, nwin, hw;      /* two-frame units, frame windows, hardware waves launched WITHOUT the estimate */
    int hw_est, nwe;          /* hardware waves launched WITH it (twelve unless the caller says otherwise), and how many of them estimate */
    bool twowin, pair;
    size_t lds, lds_est;
};

static LeanGeometry lean_geometry(const FusedArgs &a, int G, unsigned long long layout)
{
    using namespace pipe2;
    LeanGeometry g{};
    g.hw = 1;
    bool valid = (layout & 15) == 0;
    for (int w = 1; w < 16; w++) {
        const int cw = (int)((layout >> (4 * w)) & 15);
        if (cw > 2 || (cw && w >= MAX_THREADS / 64)) valid = false;
        g.units += cw;
        g.nwin += cw != 0;
        if (cw) g.hw = w + 1;
    }
    valid = valid && G >= 2 && g.units == G / UF && lean_lds_bytes(G, g.nwin) <= (size_t)MAX_LDS_BYTES;
    /* a window per UNIT where the LDS has the room (up to 16 frames per workgroup; a.lean_twowin comes in as the caller's wish) */
    g.twowin = valid && a.lean_twowin && a.lean_dma && g.units > g.nwin && lean_lds_bytes(G, g.units) <= (size_t)MAX_LDS_BYTES;
    if (g.twowin) g.nwin = g.units;
    /* two lanes of the serial wave per loop (costas_wave): a.lean_pair comes in as the caller's wish: 1 = up to 16 frames per workgroup,
     * 2 = wherever the wave has the lanes, 3 = the library's rule: up to 24 frames per workgroup.  [Measured in steady state,
     * profiles/r06_step_cost.txt: 4096 frames 0.1511-0.1520 -> 0.1488-0.1500 ms, 6144 frames 0.2113 -> 0.2055, 8192 frames (32 per
     * workgroup, every lane of the wave busy, the board at its power limit) 0.2531 -> 0.2541-0.2545: not there] */
    g.pair = (a.lean_pair == 2 && 2 * G <= 64) || (a.lean_pair == 1 && G <= 16) || (a.lean_pair == 3 && G <= 24);
    g.lds = lean_lds_bytes(G, g.nwin);
    /* the FFT timing estimate inside the launch: twelve hardware waves share the workgroup's frames (waves without a unit retire
     * after it), at most MAX_FPW frames per wave, estimator windows in the frame windows, taps + indices behind the rows */
    {
        /* twelve hardware waves (three per SIMD: the kernel's register budget) share the workgroup's frames: 0.1654 ms against 0.1692 with
         * eight at config 3, launches interleaved in one process (profiles/r06_config3_estimator_waves.txt) */
        const int want = a.est_waves > 0 ? a.est_waves : 12;
        g.hw_est = g.hw < want ? want : g.hw;
        if (g.hw_est > MAX_THREADS / 64) g.hw_est = MAX_THREADS / 64;
    }
    const int room = (int)((size_t)g.nwin * UF * lean::WS / tfft::WSLOTS);
    g.nwe = g.hw_est < room ? g.hw_est : room;
    g.lds_est = g.lds + LEAN_EST_LDS_BYTES;
    g.ok = valid;
    g.est_ok = valid && a.frame_size >= tfft::N0 + tfft::NFFT && g.nwe >= 1 && (G + g.nwe - 1) / g.nwe <= tfft::MAX_FPW && G <= 32 &&
               g.lds_est <= (size_t)MAX_LDS_BYTES;
    return g;
}

/* can rx_lean_kernel run the FFT timing estimate inside its launch at this geometry?  (api.cpp asks; launch_rx_lean reads the same numbers) */
bool lean_est_ok(const FusedArgs &a, int G, unsigned long long layout)
{
    return lean_geometry(a, G, layout).est_ok;
}

int launch_rx_lean(const FusedArgs &a0, int G, unsigned long long layout, int *status, hipStream_t s)
{
    FusedArgs a = a0;
    const LeanGeometry g = lean_geometry(a, G, layout);
    if (!g.ok || !lean_shape_ok(a, G)) return (int)hipErrorInvalidValue;
    if (a.est_tw && (!g.est_ok || !a.est_cs || a.index)) return (int)hipErrorInvalidValue;
    a.G = G;
    a.mixed = 2;
    a.share_simd0 = ((layout >> 16) & 15) != 0 || ((layout >> 32) & 15) != 0;
    a.lean_twowin = g.twowin;
    a.lean_pair = g.pair;
    hipLaunchKernelGGL(rx_lean_kernel, dim3((a.nframes + G - 1) / G), dim3(64 * (a.est_tw ? g.hw_est : g.hw)), a.est_tw ? g.lds_est : g.lds, s, a,
                       layout, g.nwin, status);
    return (int)hipGetLastError();
}

/* frames of a workgroup with NF FIR waves */
template <class GM>
static int frames_of(int NF)
{
    return NF * GM::FWV;
}

template <class GM>
static size_t lds_bytes_of(int NF, int nbw)
{
    const size_t G = (size_t)frames_of<GM>(NF);
    size_t b = sizeof(Smem) + sizeof(float2) * (G * GM::WSLOTS + G * GM::DSTRIDE) + sizeof(float) * G * nbw * GM::ZSTRIDE;
    return (b + 15) & ~(size_t)15;
}

size_t pipe_lds_bytes(int NF, int nbw)
{
    return lds_bytes_of<GeomNarrow>(NF, nbw);
}

int pipe_frames(int NF) { return frames_of<GeomNarrow>(NF); }
int pipe_cycles(void) { return C; }
int pipe_max_nf(void) { return GeomNarrow::MAX_NF; }

int launch_rx_fused_pipe(const FusedArgs &a0, int NF, int *status, hipStream_t s)
{
    FusedArgs a = a0;
    a.lean_pair = 0;
    /* the full narrow workgroup runs with two lane mappings (see the kernel): no SIMD carries two four-frame
     * units, every FIR wave has slack and the kernel follows the loop.  QPSK_PIPE_DBG bit 7 = the plain layout
     * (4 FIR waves + 1 spare) for A/B runs */
    a.mixed = NF == GeomNarrow::MAX_NF && !(a.dbg & (128 | 4));
    /* several loops per frame (bandwidth sweeps): the flush costs a sin/cos per loop and symbol, so the frames go
     * two to a FIR wave (2 symbols per lane) instead of four */
    if (!a.mixed && a.nbw > 1 && NF <= 3 && !(a.dbg & (128 | 4))) a.mixed = 2;
    const int G = pipe_frames(NF);
    const int blocks = (a.nframes + G - 1) / G;
    const size_t lds = pipe_lds_bytes(NF, a.nbw);
    if (NF < 1 || NF > pipe_max_nf() || lds > (size_t)MAX_LDS_BYTES || G * a.nbw > 64) return (int)hipErrorInvalidValue;
    /* the in-launch FFT timing estimate is written for the full workgroup: 8 hardware waves x 2 frames, windows in the frame windows' LDS */
    if (a.est_tw && (a.mixed != 1 || !a.est_cs || a.frame_size < tfft::N0 + tfft::NFFT || a.index ||
                     (size_t)8 * tfft::WSLOTS > (size_t)G * GeomNarrow::WSLOTS))
        return (int)hipErrorInvalidValue;
    /* hardware waves of a workgroup with NF FIR waves: with spares, FIR wave k is hardware wave k + 1 + k/3 */
    const int nfir = a.mixed == 2 ? 2 * NF : NF;
    auto nwaves = [&](int spare) { return (spare && !(a.dbg & 4)) ? nfir + 1 + (nfir - 1) / 3 : nfir + 1; };
    /* mixed: hardware waves 0 (loop), 1-3 (4 frames each), 4-5 (retire), 6-7 (2 frames each) */
    const dim3 threads(a.mixed == 1 ? 512 : 64 * nwaves(GeomNarrow::SPARE));
    hipLaunchKernelGGL(rx_fused_pipe_kernel<GeomNarrow>, dim3(blocks), threads, lds, s, a, status);
    hipError_t e = hipGetLastError();
    return (int)e;
}

int launch_costas_pipe(const FusedArgs &a0, int NF, int *status, hipStream_t s)
{
    using GM = GeomNarrow;
    FusedArgs a = a0;
    a.lean_pair = 0;
    const int G = NF * GM::FWV;
    const int blocks = (a.nframes + G - 1) / G;
    const size_t lds = sizeof(Smem) + sizeof(float2) * (size_t)G * GM::DSTRIDE + sizeof(float) * (size_t)G * GM::ZSTRIDE;
    hipLaunchKernelGGL(costas_pipe_kernel, dim3(blocks), dim3(64 * (NF + 1 + (a.carrier_state ? 1 : 0))), lds, s, a, status);
    return (int)hipGetLastError();
}

/* ========================================================================
 * rx_hist_kernel: the reference's histogram timing mode (qpsk.c:127-191 + 196-212) in ONE pass over the input (round 6).
 *
 * Until round 5 histogram mode read the batch twice: timing_scan_kernel (full-rate rrc_fir() + the amplitude-histogram scan, 0.60 ms at
 * config 2, VALU-bound) for the index, then the receive kernel, which filters the kept phase AGAIN and runs its 2048-step serial chain in
 * a second launch (0.15 ms).  The index is only known at a frame's END, but it is the same for nearly every frame of a batch and from
 * batch to batch (every clean frame of one configuration lands on one value), so this kernel takes a GUESS -- the majority index of
 * the context's previous histogram-mode call, from device memory -- and runs the receive path on it inside the scan kernel's
 * workgroup, where every filtered sample already is:
 *   waves 0-3    scan waves, timing_scan_kernel's -- their chain leaves their SIMD two thirds idle, so they also FLUSH the chunks the loop
 *                has finished (flush_records: sin/cos of the recorded phase, the reference's four products, slicer; 16 lanes per frame);
 *                at the frame's end they write the true index and, where it differs from the guess, append the frame to the launch's list;
 *   waves 4-11   FIR waves, timing_scan_kernel's (two frames each, full-rate stream, taps in SGPRs) -- a lane's 8 outputs are
 *                one symbol, so the guessed phase's pick is one of its accumulators: it goes to the frame's symbol ring row (8 bytes
 *                per lane and tile; two tiles = one 64-symbol chunk = one hand-over);
 *   (role 12)    the serial wave of the pipeline kernels (costas_wave: the ring stream in its low-register build, two lanes per
 *                loop), with all the slack in the world: a chunk arrives every ~18 us and costs it 4-7.
 * Four waves on a SIMD = 128 VGPRs per wave: the Costas streams come from costas_asm_lo.h (VGPR block at v84..v127); which hardware wave
 * plays which part is dealt by SIMD (histk::role_of).
 * The frames on the list are then redone by the fall-back pass (rx_fused_kernel over the list, with the true indices) in a second,
 * usually empty, launch; a third, one-workgroup launch leaves the batch's majority index as the next call's guess.
 * Host conditions (rx_hist_shape_ok): CYCLES = 8, whole 64-symbol chunks, a symmetric filter, one loop per frame, no costas_frame[].
 * ======================================================================== */
namespace histk {
constexpr int G = 16, NSCAN = 4, NFIR = 8, UF = 2, QL = 32, R = 8;
constexpr int TILE = QL * R;          /* 256 outputs = 32 symbols per frame and round */
constexpr int DRO = 2, PADS = 2, WPOS = TILE + HIST, WSF = 480, PITCH = TILE + 4;
/* Hardware wave i of a workgroup runs on SIMD i % 4.  The serial wave costs its SIMD 2048 steps x 26 instructions x 4 cycles = 0.09 ms of
 * vector issue, a scan wave 0.05, a FIR wave 0.25: with the serial wave as a thirteenth wave beside a scan wave and two FIR waves that SIMD
 * was the kernel's straggler (0.70 ms against timing_scan_kernel's 0.60).  So the roles are dealt by SIMD: SIMD 0 = the serial wave + two
 * FIR waves and NO scan wave, SIMD 1 = two scan waves + two FIR waves, SIMDs 2, 3 = one scan wave + two FIR waves (0.58 / 0.59 / 0.54 /
 * 0.54 ms of issue): 14 hardware waves, of which wave 12 (the fourth on SIMD 0) retires at once. */
constexpr int HW_WAVES = 14;
constexpr int THREADS = 64 * HW_WAVES;
__device__ constexpr int role_of(int hw)      /* 0..3: scan wave, 4..11: FIR wave hw - 4, 12: the serial wave, -1: nothing */
{
    return hw == 0 ? 12 : hw <= 3 ? hw - 1 : hw <= 11 ? hw : hw == 13 ? 3 : -1;
}
__device__ __host__ constexpr int slot_of(int p) { return p + PADS * (p / R); }
static_assert(slot_of(WPOS - 1) < WSF && WSF % 2 == 0, "window geometry (timing_scan.hip)");
static_assert(COSTAS_ASM_LO_END_VGPR <= 128, "rx_hist_kernel: four waves on SIMD 0");
struct Ctl {
    int ready[NFIR];          /* tiles produced, per FIR wave (scan side) */
    int consumed[NSCAN];      /* tiles consumed, per scan wave */
    int flushed[NSCAN];       /* chunks of its four frames a scan wave has turned into symbols (flush_records) */
    lean::Ctl loop;           /* the receive side: chunks handed over per FIR wave, chunks consumed by the serial wave, the abort flag of both */
};
} // namespace histk

__global__ void __launch_bounds__(histk::THREADS)
rx_hist_kernel(FusedArgs a, int32_t *index_true, const int32_t *hint, int32_t *mis_list, int32_t *mis_count, int *status)
{
    using namespace histk;
    using GM = lean::GeomLean;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Ctl *sm = reinterpret_cast<Ctl *>(smem_raw);
    float2 *win = reinterpret_cast<float2 *>(smem_raw + sizeof(Ctl));                  /* [G][WSF] */
    float *ring = reinterpret_cast<float *>(win + (size_t)G * WSF);                    /* [G][DRO][2 planes][PITCH] */
    unsigned char *rows = reinterpret_cast<unsigned char *>(ring + (size_t)G * DRO * 2 * PITCH);      /* [G][lean::ROW_BYTES]: symbols, records */
    float2 *dring = reinterpret_cast<float2 *>(rows);
    float *zring = reinterpret_cast<float *>(rows + lean::Z_OFFSET_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = role_of(__builtin_amdgcn_readfirstlane(tid >> 6));      /* the wave's ROLE (see role_of) */
    const int f0 = blockIdx.x * G;
    const int frame_size = a.frame_size, nframes = a.nframes;
    const int ntiles = frame_size / TILE, nchunks = a.nsym / pipe2::S;
    const int gi = __builtin_amdgcn_readfirstlane(hint[0]) & 7;      /* the guessed decimation offset of every frame of the launch */
    int *abortf = &sm->loop.abort_flag;

    if (tid < NFIR) sm->ready[tid] = 0;
    if (tid < NSCAN) { sm->consumed[tid] = 0; sm->flushed[tid] = 0; }
    if (tid < MAX_WAVES) sm->loop.ready[tid] = 0;
    if (tid == 0) { sm->loop.consumed = 0; sm->loop.abort_flag = 0; }
    __syncthreads();

    if (wave < 0) return;
    if (wave == NSCAN + NFIR) {
        /* ================================ the serial wave ======================================================= */
        costas_wave<GM, StreamLo>(a, &sm->loop, dring, zring, G, f0, lane, nchunks, status);
        return;
    }
    if (wave < NSCAN) {
        /* ================================ scan wave: frames 4*wave .. 4*wave+3 (timing_scan.hip) ================= */
        /* ... and the flush of its four frames (flush_records, 16 lanes per frame, four symbols each): the scan's chain leaves this
         * wave's SIMD two thirds idle, and the chunk the loop has finished is served wherever the wave would otherwise spin or has
         * just finished a tile.  flushed[wave] tells the FIR waves that a symbol-ring slot may be filled again. */
        __builtin_amdgcn_s_setprio(3);
        const int fl = lane >> 4, comp = (lane >> 3) & 1, q = lane & 7;
        const int g = 4 * wave + fl;
        const float qf = (float)q;
        float av = 0.0f, mx = 0.0f;
        int cum = 0, done = 0;
        bool ok = true;
        auto serve_flush = [&]() {
            const int cons = __builtin_amdgcn_readfirstlane(ld_acquire(&sm->loop.consumed));
            while (done < cons && done < nchunks) {
                if (f0 + g < nframes) flush_records<GM, 4>(a, zring, dring, g, f0 + g, lane & 15, done);
                done++;
                if (lane == 0) st_release(&sm->flushed[wave], done);
            }
        };
        for (int t = 0; t < ntiles && ok; t++) {
            /* the tile of this lane's frame comes from FIR wave g / 2; every lane waits for its own producer */
            int spins = 0;
            while (!__all(ld_acquire(&sm->ready[g >> 1]) >= t + 1)) {
                serve_flush();
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_LIMIT || __hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                    __hip_atomic_store(abortf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    ok = false;
                    break;
                }
            }
            if (!ok) break;
            const float4 *row = reinterpret_cast<const float4 *>(ring + ((size_t)(g * DRO + (t % DRO)) * 2 + comp) * PITCH);
#pragma unroll 4
            for (int sidx = 0; sidx < TILE / 8; sidx++) {
                const float4 va = row[2 * sidx], vb = row[2 * sidx + 1];
                av += fabsf(va.x); av += fabsf(va.y); av += fabsf(va.z); av += fabsf(va.w);      /* qpsk.c:131-136 */
                av += fabsf(vb.x); av += fabsf(vb.y); av += fabsf(vb.z); av += fabsf(vb.w);
                av *= 0.125f;                           /* av /= CYCLES (qpsk.c:137-138) */
                if (av > mx) mx = av;                   /* qpsk.c:140-145 */
                const float th = (mx * 0.125f) * qf;    /* (max / 8.0f) * q, qpsk.c:147-165 */
                cum += (av <= th) ? 0 : 1;
            }
            if (lane == 0) st_release(&sm->consumed[wave], t + 1);
            serve_flush();
        }
        /* the chunks still in the rings leave as the loop finishes each */
        while (ok && done < nchunks) {
            ok = wait_ge(&sm->loop.consumed, done + 1, abortf);
            if (ok) serve_flush();
        }
        if (!ok) {
            if (lane == 0) report_status(status, STATUS_PIPE_TIMEOUT);
            return;
        }
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
        if (f0 + g < nframes && comp == 0 && q == 0) {
            index_true[f0 + g] = best;                  /* qpsk.c:173-180 */
            if (best != gi) mis_list[atomicAdd(mis_count, 1)] = f0 + g;      /* the guess was wrong for this frame: the fall-back pass redoes it */
        }
        return;
    }

    /* ==================================== FIR wave: frames 2*w, 2*w+1 of the workgroup (timing_scan.hip) ============ */
    const int w = wave - NSCAN;
    const int fl = lane / QL, q = lane % QL;
    const int g = UF * w + fl;
    const bool fv[UF] = {f0 + UF * w < nframes, f0 + UF * w + 1 < nframes};
    const float4 *src[UF];
#pragma unroll
    for (int ff = 0; ff < UF; ff++)
        src[ff] = reinterpret_cast<const float4 *>(a.x + (size_t)(fv[ff] ? f0 + UF * w + ff : 0) * a.frame_pitch);
    float2 *mywin = win + (size_t)(UF * w) * WSF;
    const unsigned rd_addr = lds_addr(mywin + fl * WSF + (R + PADS) * q);
    const int p0 = 2 * lane + HIST;
    float4 hist[UF], pre[UF][2];
#pragma unroll
    for (int ff = 0; ff < UF; ff++) hist[ff] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto prefetch = [&](int t) {
#pragma unroll
        for (int ff = 0; ff < UF; ff++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                float4 v = load_once(&src[ff][(t * TILE + 128 * j + 2 * lane) >> 1]);
                if (!fv[ff]) v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                pre[ff][j] = v;
            }
    };
    prefetch(0);
    bool ok = true;
    float2 *drow = dring + (size_t)g * GM::DSTRIDE;
    for (int t = 0; t < ntiles && ok; t++) {
        {   /* the two FIR waves of a SIMD stay within a tile of each other (timing_scan.hip) */
            const int pt = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sm->ready[w ^ 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (pt > t) __builtin_amdgcn_s_setprio(2);
            else if (pt < t) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(1);
        }
#pragma unroll
        for (int ff = 0; ff < UF; ff++) {
            float2 *wf = mywin + ff * WSF;
            if (lane >= 1) *reinterpret_cast<float4 *>(wf + slot_of(p0 - 128)) = hist[ff];
            *reinterpret_cast<float4 *>(wf + slot_of(p0)) = pre[ff][0];
            *reinterpret_cast<float4 *>(wf + slot_of(p0 + 128)) = pre[ff][1];
            hist[ff] = pre[ff][1];
        }
        if (t + 1 < ntiles) prefetch(t + 1);
        v2f a0, a1, a2, a3, a4, a5, a6, a7;
        fir_full8s_asm(rd_addr, a.taps, a0, a1, a2, a3, a4, a5, a6, a7);
        if (t >= DRO) ok = wait_ge(&sm->consumed[g >> 2], t - DRO + 1, abortf);
        const int c = t >> 1;               /* the 64-symbol chunk this tile is half of */
        /* the symbol ring holds two chunks: chunk c goes where chunk c - 2 was, once the loop is through that one and this frame's scan
         * wave has flushed it */
        if (ok && !(t & 1) && c >= DR) ok = wait_ge(&sm->flushed[g >> 2], c - 1, abortf);
        if (!ok) break;
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
        /* qpsk.c:190 on the guessed index: symbol 32 t + q of the frame = this lane's output number gi */
        float2 pick;
        switch (gi) {
        case 0: pick = y0; break;
        case 1: pick = y1; break;
        case 2: pick = y2; break;
        case 3: pick = y3; break;
        case 4: pick = y4; break;
        case 5: pick = y5; break;
        case 6: pick = y6; break;
        default: pick = y7; break;
        }
        drow[(c % DR) * pipe2::S + QL * (t & 1) + q] = pick;
        if (lane == 0) {
            st_release(&sm->ready[w], t + 1);
            if (t & 1) __hip_atomic_store(&sm->loop.ready[w], c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      /* behind the release above */
        }
    }
    if (!ok && lane == 0) {
        __hip_atomic_store(abortf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        report_status(status, STATUS_PIPE_TIMEOUT);
    }
}

/* the batch's majority index (first maximum) as the next call's guess, and -- for the host's choice of route next time -- how many of
 * the batch's frames are NOT on it: h_stats (pinned host memory, one 16-byte store) = {majority index, frames, frames off the majority or
 * missed by this call's guess (the larger of the two), 0}.  Also resets the miss counter for the next one-pass call. */
__global__ void __launch_bounds__(1024)
index_majority_kernel(const int32_t *index, int nframes, int32_t *hint, int32_t *mis_count, int32_t *h_stats)
{
    __shared__ int cnt[8];
    if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
    __syncthreads();
    int mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int f = threadIdx.x; f < nframes; f += 1024) {
        const int ix = index[f] & 7;
#pragma unroll
        for (int k = 0; k < 8; k++) mine[k] += ix == k;
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (mine[k]) atomicAdd(&cnt[k], mine[k]);
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        for (int k = 1; k < 8; k++)
            if (cnt[k] > cnt[best]) best = k;
        const int offm = nframes - cnt[best];
        const int missed = mis_count ? mis_count[0] : 0;
        hint[0] = best;
        if (mis_count) mis_count[0] = 0;
        int4 st = make_int4(best, nframes, offm > missed ? offm : missed, 0);
        *reinterpret_cast<int4 *>(h_stats) = st;
    }
}

static size_t rx_hist_lds_bytes(void)
{
    using namespace histk;
    return sizeof(Ctl) + sizeof(float2) * (size_t)G * WSF + sizeof(float) * (size_t)G * DRO * 2 * PITCH + (size_t)G * lean::ROW_BYTES;
}

bool rx_hist_shape_ok(const FusedArgs &a)
{
    return a.nbw == 1 && !a.costas && !a.state_in && !a.state_out && a.cycles == C && a.frame_size == a.nsym * C &&
           a.nsym % pipe2::S == 0 && a.nsym >= 2 * pipe2::S && (a.frame_pitch & 1) == 0 && ((uintptr_t)a.x & 15) == 0 &&
           a.min_freq < 0.0f && a.max_freq > 0.0f && rx_hist_lds_bytes() <= (size_t)MAX_LDS_BYTES;
}

int launch_rx_hist(const FusedArgs &a0, int32_t *index_true, const int32_t *hint, int32_t *mis_list, int32_t *mis_count, int *status, hipStream_t s)
{
    FusedArgs a = a0;
    if (!rx_hist_shape_ok(a) || !index_true || !hint || !mis_list || !mis_count) return (int)hipErrorInvalidValue;
    a.G = histk::G;
    a.mixed = 2;              /* the serial wave's lane of frame g waits on loop.ready[g / 2]: FIR wave g / 2 */
    a.share_simd0 = 0;        /* the serial wave at priority 3, like the scan waves: every FIR wave waits for it through the flush */
    a.lean_pair = a.lean_pair != 0;
    a.index = nullptr;
    a.est_tw = nullptr;
    hipLaunchKernelGGL(rx_hist_kernel, dim3((a.nframes + histk::G - 1) / histk::G), dim3(histk::THREADS), rx_hist_lds_bytes(), s, a, index_true,
                       hint, mis_list, mis_count, status);
    return (int)hipGetLastError();
}

int launch_index_majority(const int32_t *index, int nframes, int32_t *hint, int32_t *mis_count, int32_t *h_stats, hipStream_t s)
{
    hipLaunchKernelGGL(index_majority_kernel, dim3(1), dim3(1024), 0, s, index, nframes, hint, mis_count, h_stats);
    return (int)hipGetLastError();
}

int prepare_pipe_kernel(void)
{
    hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(rx_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_BYTES);
    if (e0 != hipSuccess) return (int)e0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(costas_pipe_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(rx_pipe2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            MAX_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(rx_lean_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            MAX_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(rx_fused_pipe_kernel<GeomNarrow>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_BYTES);
}

} // namespace qpsk

#ifdef QPSK_PIPE_PROFILE
/* measurement build only (libqpsk_hip_prof.so): the stamps of the last rx_lean_kernel launch under dbg bit 23 */
extern "C" __attribute__((visibility("default"))) int qpsk_prof_timeline(unsigned long long *dst, int n)
{
    if (n > 1024 * 48) n = 1024 * 48;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(qpsk::g_timeline), sizeof(unsigned long long) * (size_t)n);
}
#endif
