/*
 * kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the QPSK receive path.
 * Compile with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
 *
 *   rx_fused_kernel       decimating RRC FIR + Costas loop + slicer, one pass over the input
 *                         (reference rrc_fir.c:17-30 at the samples qpsk.c:190 keeps,
 *                          qpsk.c:196-212, costas_loop.c:44-74)            -- the hot kernel
 *   rrc_fir_kernel        full-rate rrc_fir() on a batch of delay lines (rrc_fir.c:17-30)
 *   delay_line_kernel     the delay line rrc_fir() leaves behind (rrc_fir.c:19-20)
 *   timing_hist_kernel    the amplitude-histogram timing estimate (qpsk.c:127-180)
 *   costas_kernel         Costas + slicer over decimated symbols (qpsk.c:196-212)
 *   decimate_kernel       qpsk.c:186-191 for the streaming mode
 *   mixer_kernel          PCM -> complex mix (qpsk.c:114-120)
 *   fft_kernel            radix-2 complex-double FFT (algorithms/fft.c:38-136)
 *
 * Bit-exactness rules: no FMA contraction, FIR taps summed
 * 0..126 in one fp32 accumulator, sin/cos from sincos_f32.h, no shuffle-tree
 * reductions over floating-point sums.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qpsk_device.h"
#include "fft_lds.h"
#include "kernels.h"

namespace qpsk {

/* ========================================================================
 * rx_fused_kernel (V0: chunked, barrier-synchronised)
 *
 * One workgroup owns G consecutive frames and walks them in chunks of S
 * symbols.  Per chunk: (1) all threads stage the S*C new samples of each
 * frame plus the 126-sample history into LDS with coalesced 8-byte loads,
 * (2) all threads compute the G*S decimated FIR outputs (taps in LDS,
 * broadcast reads), (3) wave 0 runs the G*nbw Costas recurrences over the S
 * symbols, one lane per (frame, loop), and stages symbols / costas_frame in
 * LDS, (4) all threads write the staged outputs coalesced.
 *
 * HBM traffic: 8 B read per input sample (+ history re-reads that hit L2),
 * 1 B written per symbol per loop.
 * ======================================================================== */
constexpr int FUSED_THREADS = 256;

__global__ void __launch_bounds__(FUSED_THREADS)
rx_fused_kernel(FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int G = a.G, S = a.S, C = a.cycles, L = a.frame_size, N = a.nsym, nbw = a.nbw;
    const int W = S * C + LOOKBACK + LOOKAHEAD; /* samples staged per frame per chunk */
    float *taps = reinterpret_cast<float *>(smem);                          /* 128 floats */
    float2 *xs = reinterpret_cast<float2 *>(smem + 512);                    /* [G][W] */
    float2 *ds = xs + (size_t)G * W;                                        /* [G][S] decimated symbols */
    float2 *zs = ds + (size_t)G * S;                                        /* [G*nbw][S] costas_frame staging */
    uint8_t *ss = reinterpret_cast<uint8_t *>(zs + (size_t)G * nbw * S);    /* [G*nbw][S] symbol staging */
    int *idx = reinterpret_cast<int *>(ss + (size_t)G * nbw * S);           /* [G], then the G frame numbers */
    int *fid = idx + G;

    const int tid = threadIdx.x;
    const int f0 = blockIdx.x * G;
    /* the frames of this launch: 0 .. nframes, or the listed ones (the fall-back pass of rx_hist_kernel: count known on the device only) */
    const int total = a.frame_list_count ? min(*a.frame_list_count, a.nframes) : a.nframes;
    if (f0 >= total) return;
    const int gcount = min(G, total - f0);

    if (tid < 128)
        taps[tid] = tid < NTAPS ? a.taps[tid] : 0.0f;
    if (tid < G) {
        const int fr = tid < gcount ? (a.frame_list ? a.frame_list[f0 + tid] : f0 + tid) : 0;
        fid[tid] = fr;
        idx[tid] = (tid < gcount) ? (a.index ? a.index[fr] : a.fixed_index) : 0;
    }

    /* Costas lane state (wave 0 only) */
    const int lane_g = tid / nbw, lane_b = tid % nbw;
    const bool costas_lane = tid < gcount * nbw && tid < 64;
    Loop st = {0.0f, 0.0f};
    bool over = false;   /* a phase beyond the bounded 2 pi wrap (qpsk_device.h) */
    LoopGains lg = {0.0f, 0.0f, a.min_freq, a.max_freq};
    __syncthreads();
    if (costas_lane) {
        lg.alpha = a.gains[2 * lane_b];
        lg.beta = a.gains[2 * lane_b + 1];
        if (a.state_in) {
            st.phase = a.state_in[2 * ((size_t)fid[lane_g] * nbw + lane_b)];
            st.freq = a.state_in[2 * ((size_t)fid[lane_g] * nbw + lane_b) + 1];
        }
    }

    const int nchunks = (N + S - 1) / S;
    for (int c = 0; c < nchunks; c++) {
        const int sym0 = c * S;
        const int base = sym0 * C - LOOKBACK; /* sample index of xs[g][0] */

        /* (1) stage samples: zeros before the frame (fresh delay line) and past its end */
        for (int i = tid; i < gcount * W; i += FUSED_THREADS) {
            const int g = i / W, w = i - g * W;
            const int n = base + w;
            float2 v = make_float2(0.0f, 0.0f);
            if (n >= 0 && n < L)
                v = a.x[(size_t)fid[g] * a.frame_pitch + n];
            xs[(size_t)g * W + w] = v;
        }
        __syncthreads();

        /* (2) decimated FIR: output (g, j) is the full-rate output at sample (sym0+j)*C + idx[g] */
        for (int o = tid; o < gcount * S; o += FUSED_THREADS) {
            const int g = o / S, j = o - g * S;
            const int n = (sym0 + j) * C + idx[g];
            float2 d = make_float2(0.0f, 0.0f);
            if (n < L && sym0 + j < N) {
                const float2 *w = xs + (size_t)g * W + (n - HIST - base);
                float2 y = make_float2(0.0f, 0.0f);
#pragma unroll 8
                for (int k = 0; k < NTAPS; k++)
                    fir_mac(y, w[k], taps[k]);
                d = fir_gain(y);
            }
            ds[o] = d;
        }
        __syncthreads();

        /* (3) the serial part: one lane per (frame, loop) */
        if (costas_lane) {
            const int cnt = min(S, N - sym0);
            const float2 *dl = ds + (size_t)lane_g * S;
            for (int j = 0; j < cnt; j++) {
                const float2 z = (c == 0 && j == 0) ? costas_step<true>(st, lg, dl[j], over) : costas_step<false>(st, lg, dl[j], over);
                ss[(size_t)tid * S + j] = (uint8_t)slicer(z);
                if (a.costas)
                    zs[(size_t)tid * S + j] = z;
            }
        }
        __syncthreads();

        /* (4) coalesced write-out of the chunk */
        {
            const int cnt = min(S, N - sym0);
            const int rows = gcount * nbw;
            for (int i = tid; i < rows * cnt; i += FUSED_THREADS) {
                const int r = i / cnt, j = i - r * cnt;
                const size_t o = ((size_t)fid[r / nbw] * nbw + r % nbw) * N + sym0 + j;
                a.sym[o] = ss[(size_t)r * S + j];
                if (a.costas)
                    a.costas[o] = zs[(size_t)r * S + j];
            }
        }
        /* next chunk's staging overwrites xs/ds only after everyone passed the barrier above;
         * ss/zs are rewritten in step (3) of the next chunk, after two more barriers */
    }

    if (costas_lane) {
        const size_t o = (size_t)fid[lane_g] * nbw + lane_b;
        if (a.freq) a.freq[o] = st.freq;
        if (a.phase) a.phase[o] = st.phase;
        if (a.hz) a.hz[o] = (float)((double)st.freq * a.rs / TAU); /* qpsk.c:217 */
        if (a.state_out) {
            a.state_out[2 * o] = st.phase;
            a.state_out[2 * o + 1] = st.freq;
        }
        if (over && a.status) __hip_atomic_store(a.status, STATUS_PHASE_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (a.status && !loop_state_finite(st.phase, st.freq))
            __hip_atomic_store(a.status, STATUS_NONFINITE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

size_t fused_lds_bytes(int G, int S, int cycles, int nbw)
{
    const size_t W = (size_t)S * cycles + LOOKBACK + LOOKAHEAD;
    size_t b = 512 + sizeof(float2) * ((size_t)G * W + (size_t)G * S + (size_t)G * nbw * S);
    b += (size_t)G * nbw * S;      /* ss */
    b = (b + 15) & ~(size_t)15;
    b += sizeof(int) * 2 * (size_t)G;  /* idx, fid */
    return b;
}

/* ========================================================================
 * rrc_fir_kernel: full-rate FIR (rrc_fir.c:17-30).  grid = (tiles, frames);
 * a workgroup produces FIR_TILE = 2048 outputs of one delay line from
 * 2048 + 126 staged inputs; history before the block comes from the
 * caller's delay line (memory[1..126]; memory[0] is shifted out by the first
 * step, rrc_fir.c:19) or is zero.
 *
 * VALU-bound (508 unfused fp32 operations per output), so the
 * inner loop is a register sliding window: a lane produces FR = 8
 * CONSECUTIVE outputs, one 8-byte LDS read feeds up to 8 multiply-add
 * pairs (134 reads for 8 outputs instead of 8 x 127), each output's taps
 * are still summed 0..126 in one fp32 accumulator.  Lanes are 8 samples
 * apart; the LDS image puts sample position p at slot p + p/8, i.e. lanes
 * 9 slots = 18 dwords apart: 32 lanes on 32 distinct even banks.  Taps come
 * from LDS broadcast reads, two groups of 8 live at a time.  Outputs go back
 * through the same padded image so that the global stores are coalesced.
 * ======================================================================== */
constexpr int FIR_THREADS = 256;
constexpr int FR = 8;
constexpr int FIR_TILE = FIR_THREADS * FR;
constexpr int FIR_WIN = FIR_TILE + HIST;
constexpr int FIR_SLOTS = FIR_WIN + FIR_WIN / 8 + 1;

__global__ void __launch_bounds__(FIR_THREADS, 2)
rrc_fir_kernel(const float2 *__restrict__ x, const float2 *__restrict__ memory, float2 *__restrict__ y,
               const float *__restrict__ taps_g, int length, size_t in_pitch)
{
    __shared__ __attribute__((aligned(16))) float taps[128];
    __shared__ float2 xs[FIR_SLOTS];
    const int tid = threadIdx.x, f = blockIdx.y;
    const int n0 = blockIdx.x * FIR_TILE;
    if (tid < 128)
        taps[tid] = tid < NTAPS ? taps_g[tid] : 0.0f;
    for (int i = tid; i < FIR_WIN; i += FIR_THREADS) {
        const int n = n0 - HIST + i;
        float2 v = make_float2(0.0f, 0.0f);
        if (n >= 0) {
            if (n < length) v = x[(size_t)f * in_pitch + n];       /* input frames in_pitch samples apart, output packed */
        } else if (memory) {
            v = memory[(size_t)f * NTAPS + (NTAPS + n)]; /* n = -1 -> memory[126] */
        }
        xs[i + (i >> 3)] = v;
    }
    __syncthreads();

    /* output r of this lane = sample n0 + 8*tid + r = window position 8*tid + r + 126; its tap k sits at
     * position 8*tid + r + k, so step t = r + k reads position 8*tid + t once for all r */
    const float2 *rd = xs + 9 * tid;
    const float4 *taps4 = reinterpret_cast<const float4 *>(taps);
    /* Left to itself the compiler emits each multiply-add as v_pk_mul, s_nop, dependent v_pk_add (357 s_nops and
     * as many stalls in 2032 packed operations).  As in rx_fused.hip the order inside a window position is pinned
     * with empty asm statements: the up to 8 products of the position, then the 8 adds, so that a product is 8
     * instructions ahead of its add and an accumulator's adds are 16 apart.  Window values and tap groups are
     * fetched one block of 8 positions ahead (three tap groups live: a position's 8 outputs reach back into the
     * previous group).  The sum per output is still taps 0..126 in order in one accumulator. */
    constexpr int TS = NTAPS + FR - 1, NB = (TS + 7) / 8;
    static_assert(FR == 8, "the pinned step names 8 products");
    float tg[3][8];
    float2 wv[2][8];
    auto fetch_block = [&](int tb) {
        if (tb * 8 < NTAPS) {
            const float4 ta = taps4[2 * tb], tc = taps4[2 * tb + 1];
            float *g_ = tg[tb % 3];
            g_[0] = ta.x; g_[1] = ta.y; g_[2] = ta.z; g_[3] = ta.w;
            g_[4] = tc.x; g_[5] = tc.y; g_[6] = tc.z; g_[7] = tc.w;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int t = tb * 8 + u;
            if (t < TS) wv[tb & 1][u] = rd[t + (t >> 3)];
        }
    };
    v2f ac[FR];
#pragma unroll
    for (int r = 0; r < FR; r++) ac[r] = v2f{0.0f, 0.0f};
    fetch_block(0);
    static_for<0, NB>([&](auto tbc) {
        constexpr int tb = decltype(tbc)::value;
        if (tb + 1 < NB) fetch_block(tb + 1);
        static_for<0, 8>([&](auto uc) {
            constexpr int u = decltype(uc)::value, t = tb * 8 + u;
            if constexpr (t < TS) {
                const v2f v = v2f{wv[tb & 1][u].x, wv[tb & 1][u].y};
                v2f p[FR];   /* re*tap, im*tap (rrc_fir.c:24-25); output r takes tap k = t - r */
                static_for<0, FR>([&](auto rc) {
                    constexpr int r = decltype(rc)::value, k = t - r;
                    if constexpr (k >= 0 && k < NTAPS) p[r] = v * tg[(k >> 3) % 3][k & 7];
                });
                if constexpr (t >= FR - 1 && t < NTAPS) {
                    asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
                } else {   /* the first 7 and the last 7 positions: fewer outputs are in range */
#define QPSK_PIN_P(r) if constexpr (t - (r) >= 0 && t - (r) < NTAPS) asm volatile("" : "+v"(p[r]))
                    QPSK_PIN_P(0); QPSK_PIN_P(1); QPSK_PIN_P(2); QPSK_PIN_P(3);
                    QPSK_PIN_P(4); QPSK_PIN_P(5); QPSK_PIN_P(6); QPSK_PIN_P(7);
#undef QPSK_PIN_P
                }
                static_for<0, FR>([&](auto rc) {
                    constexpr int r = decltype(rc)::value, k = t - r;
                    if constexpr (k >= 0 && k < NTAPS) ac[r] = ac[r] + p[r];
                });
                asm volatile("" : "+v"(ac[0]), "+v"(ac[1]), "+v"(ac[2]), "+v"(ac[3]), "+v"(ac[4]), "+v"(ac[5]), "+v"(ac[6]), "+v"(ac[7]));
            }
        });
    });
    __syncthreads();
    /* transpose through LDS: lane-major results -> sample-major coalesced stores */
#pragma unroll
    for (int r = 0; r < FR; r++)
        xs[9 * tid + r] = fir_gain(make_float2(ac[r].x, ac[r].y));
    __syncthreads();
#pragma unroll
    for (int j = 0; j < FR; j++) {
        const int i = j * FIR_THREADS + tid, n = n0 + i;
        if (n < length)
            y[(size_t)f * length + n] = xs[i + (i >> 3)];
    }
}

/* memory <- last 127 samples of (memory ++ x)  (rrc_fir.c:19-20 applied length times) */
__global__ void __launch_bounds__(128)
delay_line_kernel(const float2 *__restrict__ x, float2 *memory, int length)
{
    const int f = blockIdx.x, i = threadIdx.x;
    float2 v = make_float2(0.0f, 0.0f);
    if (i < NTAPS) {
        const int src = i + length - NTAPS; /* index into x; negative -> old memory */
        v = src >= 0 ? x[(size_t)f * length + src] : memory[(size_t)f * NTAPS + (i + length)];
    }
    __syncthreads();
    if (i < NTAPS)
        memory[(size_t)f * NTAPS + i] = v;
}

/* ========================================================================
 * timing_hist_kernel: qpsk.c:127-180.  The running average is never reset
 * (Q2) and the thresholds use the running maximum (Q3), so each component of
 * each frame is one serial scan: one lane per (frame, I|Q), 16 frames per
 * (single-wave) workgroup.  The filtered samples come through LDS in tiles
 * of 64 samples per frame: all 64 lanes fetch one frame's tile with a
 * coalesced 8-byte load (prefetched one tile ahead into registers), the 32
 * scanning lanes then read their own row (row pitch 65 slots: the 32 lanes
 * hit 32 different banks).
 * ======================================================================== */
constexpr int TH_FRAMES = 16;
constexpr int TH_TILE = 64;

__global__ void __launch_bounds__(64)
timing_hist_kernel(const float *__restrict__ y, int nframes, int frame_size, int cycles, int32_t *index,
                   int32_t *hist_out)
{
    __shared__ float2 tile[TH_FRAMES][TH_TILE + 1];
    const int lane = threadIdx.x;
    const int f0 = blockIdx.x * TH_FRAMES;
    const int fl = (lane >> 1) & (TH_FRAMES - 1), comp = lane & 1;
    const bool scanning = lane < 2 * TH_FRAMES && f0 + fl < nframes;
    const float2 *y2 = reinterpret_cast<const float2 *>(y);
    const int ntiles = (frame_size + TH_TILE - 1) / TH_TILE;

    float2 pre[TH_FRAMES];
    auto fetch = [&](int t) {
        const int s = min(t * TH_TILE + lane, frame_size - 1);        /* clamped: past-the-end slots are never scanned */
#pragma unroll
        for (int r = 0; r < TH_FRAMES; r++)
            pre[r] = y2[(size_t)min(f0 + r, nframes - 1) * frame_size + s];
    };

    int hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float av = 0.0f, mx = 0.0f;
    const float fc = (float)cycles;
    int j = 0;                                  /* samples of the current symbol already added */
    const float *row = reinterpret_cast<const float *>(&tile[fl][0]) + comp;
    fetch(0);
    for (int t = 0; t < ntiles; t++) {
#pragma unroll
        for (int r = 0; r < TH_FRAMES; r++)
            tile[r][lane] = pre[r];
        if (t + 1 < ntiles) fetch(t + 1);
        __syncthreads();
        if (scanning) {
            const int cnt = min(TH_TILE, frame_size - t * TH_TILE);
            /* one symbol's bookkeeping (qpsk.c:137-166); the bin is found without a data-dependent loop */
            auto symbol_end = [&]() {
                av = av / fc;
                if (av > mx) mx = av;
                const float hv = mx / 8.0f;
                int k = 8;
#pragma unroll
                for (int q = 7; q >= 1; q--)
                    if (av <= hv * (float)q) k = q;          /* the first q that holds wins */
#pragma unroll
                for (int q = 1; q < 8; q++)
                    hist[q] += (q == k) ? 1 : 0;
            };
            if (cycles == 8 && cnt == TH_TILE && j == 0) {
                /* whole symbols inside the tile: fetch a symbol's 8 samples, then the 8 ordered adds */
#pragma unroll 2
                for (int sy = 0; sy < TH_TILE / 8; sy++) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = row[2 * (sy * 8 + u)];
#pragma unroll
                    for (int u = 0; u < 8; u++) av += fabsf(v[u]);
                    symbol_end();
                }
            } else if (cycles == 4 && cnt == TH_TILE && j == 0) {
#pragma unroll 2
                for (int sy = 0; sy < TH_TILE / 4; sy++) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = row[2 * (sy * 4 + u)];
#pragma unroll
                    for (int u = 0; u < 4; u++) av += fabsf(v[u]);
                    symbol_end();
                }
            } else {
                for (int i = 0; i < cnt; i++) {
                    av += fabsf(row[2 * i]);
                    if (++j == cycles) {
                        j = 0;
                        symbol_end();
                    }
                }
            }
        }
        __syncthreads();
    }
    /* lanes 2f and 2f+1 hold hist_i and hist_q of frame f: integer add across the pair */
    int hmax = 0, best = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int h = hist[q] + __shfl_xor(hist[q], 1);
        if (h > hmax) { hmax = h; best = q; }
        if (hist_out && scanning && comp == 0) hist_out[(size_t)(f0 + fl) * 8 + q] = h;
    }
    if (scanning && comp == 0)
        index[f0 + fl] = best;
}

/* ========================================================================
 * timing_hist8_kernel: the same scan for CYCLES = 8 and frames of whole
 * 128-sample tiles, with the scan wave stripped to the recurrence.
 *
 * The scan is latency-bound (one serial chain per frame and component: the
 * running average is 8 dependent adds and a multiply per symbol), so what
 * counts is the instruction stream of the scanning wave (a lone wave issues
 * one VALU op per ~5 cycles, a dependent one after ~8, an LDS op per ~12):
 *   - the 7 threshold compares of a symbol (qpsk.c:147-165) do not depend on
 *     each other, so they go ACROSS lanes instead of down the instruction
 *     stream: 8 lanes per (frame, component), lane q owns threshold hv*q.
 *     Every lane of the group runs the same average/max chain on the same
 *     samples (LDS broadcast reads) and adds one compare + one count;
 *   - hv*q is nondecreasing in q, so "the first q with av <= hv*q" is
 *     1 + #{q : !(av <= hv*q)}: lane q counts cum[q] = #{symbols : !(av <= hv*q)}
 *     and hist[q] = cum[q-1] - cum[q], cum[0] = number of symbols (a NaN
 *     average is counted by every lane and lands in no bin, as in the
 *     reference's compare chain);
 *   - a wave scans 4 frames and stages its own tiles: 16-byte global loads
 *     prefetched a tile ahead, I and Q de-interleaved into two LDS planes so
 *     that a symbol's 8 samples of one component are two ds_read_b128;
 *   - av / 8 is av * 0.125f (both correctly rounded of the same real number).
 * ======================================================================== */
constexpr int TH8_FRAMES = 4;
constexpr int TH8_TILE = 128;
constexpr int TH8_PITCH = TH8_TILE + 4;      /* floats; rows 16-byte aligned, 4 banks apart */

__global__ void __launch_bounds__(64)
timing_hist8_kernel(const float *__restrict__ y, int nframes, int frame_size, int32_t *index, int32_t *hist_out)
{
    __shared__ __attribute__((aligned(16))) float tile[2][TH8_FRAMES][TH8_PITCH];   /* [I|Q][frame][sample] */
    const int lane = threadIdx.x;
    const int f0 = blockIdx.x * TH8_FRAMES;
    const int ntiles = frame_size / TH8_TILE;

    /* staging: lane l carries samples 2l, 2l+1 of every frame's tile */
    const float4 *y4 = reinterpret_cast<const float4 *>(y);
    float4 pre[TH8_FRAMES];
    auto fetch = [&](int t) {
#pragma unroll
        for (int r = 0; r < TH8_FRAMES; r++)
            pre[r] = y4[((size_t)min(f0 + r, nframes - 1) * frame_size + (size_t)t * TH8_TILE) / 2 + lane];
    };

    /* scanning: lane = 16*frame + 8*component + q */
    const int fl = lane >> 4, comp = (lane >> 3) & 1, q = lane & 7;
    const float qf = (float)q;
    const float4 *row = reinterpret_cast<const float4 *>(&tile[comp][fl][0]);
    float av = 0.0f, mx = 0.0f;
    int cum = 0;

    fetch(0);
    for (int t = 0; t < ntiles; t++) {
        __syncthreads();          /* the scan of tile t-1 is over */
#pragma unroll
        for (int r = 0; r < TH8_FRAMES; r++) {
            *reinterpret_cast<float2 *>(&tile[0][r][2 * lane]) = make_float2(pre[r].x, pre[r].z);
            *reinterpret_cast<float2 *>(&tile[1][r][2 * lane]) = make_float2(pre[r].y, pre[r].w);
        }
        if (t + 1 < ntiles) fetch(t + 1);
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < TH8_TILE / 8; s++) {
            const float4 a = row[2 * s], b = row[2 * s + 1];
            av += fabsf(a.x); av += fabsf(a.y); av += fabsf(a.z); av += fabsf(a.w);
            av += fabsf(b.x); av += fabsf(b.y); av += fabsf(b.z); av += fabsf(b.w);
            av *= 0.125f;                           /* av /= CYCLES */
            if (av > mx) mx = av;
            const float th = (mx * 0.125f) * qf;    /* (max / 8.0f) * q */
            cum += (av <= th) ? 0 : 1;
        }
    }
    /* lane q = 0 stands for cum[0] = the number of symbols */
    if (q == 0) cum = frame_size / 8;
    int h = __shfl_up(cum, 1) - cum;           /* hist[q], q >= 1 */
    if (q == 0) h = 0;
    h += __shfl_xor(h, 8);                     /* hist_i + hist_q (qpsk.c:175) */
    int hmax = 0, best = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int hk = __shfl(h, (lane & ~7) + k);
        if (hk > hmax) { hmax = hk; best = k; }
    }
    if (f0 + fl < nframes && comp == 0) {
        if (hist_out) hist_out[(size_t)(f0 + fl) * 8 + q] = h;
        if (q == 0) index[f0 + fl] = best;
    }
}

/* ========================================================================
 * costas_kernel: Costas + slicer over decimated symbols; one lane per
 * (frame, loop).  d: nframes rows of nsym symbols, dstride symbols apart;
 * outputs [nframes][nbw][nsym].
 * ======================================================================== */
__global__ void __launch_bounds__(64)
costas_kernel(const float2 *__restrict__ d, int nframes, int nsym, int dstride, int nbw, const float *__restrict__ gains,
              float min_freq, float max_freq, const float *state_in, float *state_out, uint8_t *sym,
              float2 *costas, int *status)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= nframes * nbw) return;
    const int f = t / nbw, b = t - f * nbw;
    Loop st = {0.0f, 0.0f};
    if (state_in) { st.phase = state_in[2 * t]; st.freq = state_in[2 * t + 1]; }
    const LoopGains lg = {gains[2 * b], gains[2 * b + 1], min_freq, max_freq};
    bool over = false;
    const float2 *p = d + (size_t)f * dstride;
    for (int i = 0; i < nsym; i++) {
        const float2 z = i == 0 ? costas_step<true>(st, lg, p[i], over) : costas_step<false>(st, lg, p[i], over);
        if (sym) sym[(size_t)t * nsym + i] = (uint8_t)slicer(z);
        if (costas) costas[(size_t)t * nsym + i] = z;
    }
    if (state_out) { state_out[2 * t] = st.phase; state_out[2 * t + 1] = st.freq; }
    if (over && status) __hip_atomic_store(status, STATUS_PHASE_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (status && !loop_state_finite(st.phase, st.freq))
        __hip_atomic_store(status, STATUS_NONFINITE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* ========================================================================
 * decimate_kernel: qpsk.c:186-191 for the generic (non-pipeline) streaming
 * path.  dec: [nstreams][nsym] holds the block the NEXT call's Costas loop
 * consumes (the reference's decimated_frame[N..2N), which it moves down at
 * the start of the next rx_frame); run after this call's Costas kernel has
 * read the previous content.  A pick past the block (index >= cycles, Q5) is
 * defined as 0.  The pipeline path refills the slots inside
 * costas_pipe_kernel instead.
 * ======================================================================== */
__global__ void __launch_bounds__(256)
decimate_kernel(const float2 *__restrict__ filtered, const int32_t *__restrict__ index, float2 *dec,
                int nstreams, int frame_size, int cycles, int nsym)
{
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (i >= nsym) return;
    float2 *row = dec + (size_t)f * nsym;
    const int src = i * cycles + index[f];
    row[i] = src < frame_size ? filtered[(size_t)f * frame_size + src] : make_float2(0.0f, 0.0f);
}

/* ========================================================================
 * mixer_kernel: qpsk.c:114-120.  phase *= rect per sample is a serial complex
 * recurrence per stream (a jump-ahead by rect^k would round differently), so
 * one lane per stream, 16 streams per single-wave workgroup; PCM comes in and
 * complex samples go out through LDS tiles of 64 samples so that every global
 * access is a coalesced row (all 64 lanes on one stream's 64 samples).  The
 * block ends with the renormalisation by cabsf (glibc hypotf: exact squares in
 * fp64, one rounded add, correctly rounded sqrt, narrowed).
 * state: [nstreams][4] = phase.re, phase.im, rect.re, rect.im
 * ======================================================================== */
constexpr int MX_STREAMS = 16;
constexpr int MX_TILE = 128;

__global__ void __launch_bounds__(64)
mixer_kernel(const int16_t *__restrict__ pcm, float2 *__restrict__ out, float *state, int nstreams, int frame_size)
{
    /* Two passes per tile of 128 samples.  Serial pass: lanes 0..15 run ONLY the carrier recurrence of their
     * stream and leave the tile's phases in LDS (3 packed VALU ops per sample, two phases per 16-byte store).
     * Parallel pass: all 64 lanes take one stream's row at a time -- a lane owns two neighbouring samples --
     * scale the PCM pair (one 4-byte load, prefetched a tile ahead), multiply by the phases and store 16 bytes:
     * coalesced rows straight from registers, no transposes. */
    __shared__ __attribute__((aligned(16))) float2 ph[MX_STREAMS][MX_TILE + 2];   /* rows 260 dwords apart */
    const int lane = threadIdx.x, f0 = blockIdx.x * MX_STREAMS;
    const bool scanning = lane < MX_STREAMS && f0 + lane < nstreams;
    const int f = min(f0 + (lane & (MX_STREAMS - 1)), nstreams - 1);
    float2 p = make_float2(state[4 * f], state[4 * f + 1]);
    const float rr = state[4 * f + 2], ri = state[4 * f + 3];
    const int ntiles = (frame_size + MX_TILE - 1) / MX_TILE;

    /* the paired accesses need whole rows of streams, an even frame size and aligned bases (wave-uniform) */
    const bool fast = f0 + MX_STREAMS <= nstreams && (frame_size & 1) == 0 &&
                      (reinterpret_cast<uintptr_t>(pcm) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    uint32_t pre[MX_STREAMS];
    auto fetch = [&](int t) {                            /* full tiles on the fast path only */
#pragma unroll
        for (int r = 0; r < MX_STREAMS; r++) {
            const int16_t *rowp = pcm + (size_t)(f0 + r) * frame_size + (size_t)t * MX_TILE;   /* uniform base */
            pre[r] = reinterpret_cast<const uint32_t *>(rowp)[lane];
        }
    };
    float nri = -ri;
    asm volatile("" : "+v"(nri));   /* opaque: keeps the compiler from folding the sign back into two packed adds */
    auto step = [&]() {                                  /* fbb_rx_phase *= fbb_rx_rect, qpsk.c:115 */
        const float2 a = make_float2(p.x * rr, p.y * rr);
        const float2 b = make_float2(p.y * nri, p.x * ri);   /* p.y * (-ri) = -(p.y * ri) exactly: re = a.x - p.y*ri */
        p = make_float2(a.x + b.x, a.y + b.y);
    };
    if (fast && frame_size >= MX_TILE) fetch(0);
    for (int t = 0; t < ntiles; t++) {
        const int cnt = min(MX_TILE, frame_size - t * MX_TILE);
        __syncthreads();                                  /* the parallel pass of tile t-1 has read ph[] */
        if (scanning) {
            float4 *prow = reinterpret_cast<float4 *>(&ph[lane][0]);
            if (cnt == MX_TILE) {
#pragma unroll 8
                for (int i = 0; i < MX_TILE / 2; i++) {
                    step();
                    const float2 p0 = p;
                    step();
                    prow[i] = make_float4(p0.x, p0.y, p.x, p.y);
                }
            } else {
                for (int i = 0; i < cnt; i++) {
                    step();
                    ph[lane][i] = p;
                }
            }
        }
        __syncthreads();
        if (fast && cnt == MX_TILE) {                    /* wave-uniform: no per-row branches, uniform row bases */
            float4 c[MX_STREAMS];
#pragma unroll
            for (int r = 0; r < MX_STREAMS; r++)
                c[r] = *reinterpret_cast<const float4 *>(&ph[r][2 * lane]);
#pragma unroll
            for (int r = 0; r < MX_STREAMS; r++) {
                float2 *orow = out + (size_t)(f0 + r) * frame_size + (size_t)t * MX_TILE;
                const float v0 = (float)(int16_t)(pre[r] & 0xffffu) / 16384.0f;
                const float v1 = (float)(int16_t)(pre[r] >> 16) / 16384.0f;
                reinterpret_cast<float4 *>(orow)[lane] =
                    make_float4(c[r].x * v0, c[r].y * v0, c[r].z * v1, c[r].w * v1);   /* qpsk.c:117 */
            }
            if ((t + 2) * MX_TILE <= frame_size) fetch(t + 1);
        } else {
            for (int i = lane; i < cnt; i += 64) {
#pragma unroll 1
                for (int r = 0; r < MX_STREAMS && f0 + r < nstreams; r++) {
                    const size_t at = (size_t)(f0 + r) * frame_size + (size_t)t * MX_TILE + i;
                    const float v = (float)pcm[at] / 16384.0f;
                    const float2 c = ph[r][i];
                    out[at] = make_float2(c.x * v, c.y * v);
                }
            }
        }
    }
    const float pr = p.x, pi = p.y;
    if (scanning) {
        const float mag = (float)sqrt((double)pr * (double)pr + (double)pi * (double)pi);   /* qpsk.c:120 */
        state[4 * f] = pr / mag;
        state[4 * f + 1] = pi / mag;
    }
}

/* ========================================================================
 * fft_kernel: algorithms/fft.c as an in-LDS iterative radix-2 DIT.  The
 * reference recursion (even/odd split, fft.c:40-52) is the bit-reversal
 * permutation followed by log2(n) butterfly stages; the stage of size m uses
 * w_k = cos(TAU k/m) -/+ j sin(TAU k/m) (fft.c:55-56, 85-86), which equals
 * table entry k*(n/m) of the size-n table bit for bit (scaling the argument
 * by a power of two is exact), so one host-built libm table serves every
 * stage.  One workgroup per transform.
 * ======================================================================== */
__global__ void __launch_bounds__(256)
fft_kernel(const double2 *__restrict__ in, double2 *__restrict__ out, const double2 *__restrict__ tw, int n,
           int log2n, int inverse)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *v = reinterpret_cast<double2 *>(smem);
    const int tid = threadIdx.x;
    const double2 *src = in + (size_t)blockIdx.x * n;
    double2 *dst = out + (size_t)blockIdx.x * n;
    for (int i = tid; i < n; i += blockDim.x) {
        const int r = (int)(__brev((unsigned)i) >> (32 - log2n));
        v[r] = src[i];
    }
    __syncthreads();
    fft_lds_stages(v, tw, n, log2n, tid, (int)blockDim.x, inverse ? 1.0 : -1.0);
    const double dn = (double)n;
    for (int i = tid; i < n; i += blockDim.x) {
        double2 r = v[i];
        if (!inverse) { r.x = r.x / dn; r.y = r.y / dn; } /* fft.c:105-107,117-119 */
        dst[i] = r;
    }
}

__global__ void fill_i32_kernel(int32_t *p, int n, int32_t v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

/* exhaustive self-test support: order-independent hash of sincos_f32 over a range of float bit patterns */
__global__ void sincos_hash_kernel(uint32_t first, uint32_t count, unsigned long long *acc)
{
    unsigned long long h = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint32_t u = first + i;
        for (int sg = 0; sg < 2; sg++) {
            const float y = __uint_as_float(u | ((uint32_t)sg << 31));
            const SinCos r = sincos_f32(y);
            unsigned long long v = ((unsigned long long)__float_as_uint(r.s) << 32) | __float_as_uint(r.c);
            v ^= (unsigned long long)(u | ((uint32_t)sg << 31)) * 0x9E3779B97F4A7C15ull;
            v *= 0xD6E8FEB86659FD93ull;
            v ^= v >> 32;
            h += v;
        }
    }
    atomicAdd(acc, h);
}

/* ------------------------------------------------------------------------ launchers */
#define LAUNCH_CHECK()                          \
    do {                                        \
        hipError_t e_ = hipGetLastError();      \
        if (e_ != hipSuccess) return (int)e_;   \
    } while (0)

int launch_rx_fused(const FusedArgs &a, hipStream_t s)
{
    if (a.est_tw) return (int)hipErrorInvalidValue;    /* no in-launch timing estimate in this kernel: it would use fixed_index */
    const int blocks = (a.nframes + a.G - 1) / a.G;
    const size_t lds = fused_lds_bytes(a.G, a.S, a.cycles, a.nbw);
    hipLaunchKernelGGL(rx_fused_kernel, dim3(blocks), dim3(FUSED_THREADS), lds, s, a);
    LAUNCH_CHECK();
    return 0;
}

__global__ void fft_big_first_kernel(const double2 *__restrict__ in, double2 *__restrict__ out, const double2 *__restrict__ twb,
                                     int n, int log2n, int inverse);
__global__ void fft_big_late_kernel(double2 *__restrict__ data, const double2 *__restrict__ twn, int n, int log2n, int inverse);

int prepare_kernels(void)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rx_fused_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(fft_big_first_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            MAX_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(fft_big_late_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            MAX_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(fft_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_BYTES);
    return (int)e;
}

int launch_rrc_fir(const float *x, const float *memory, float *y, const float *taps, int nframes, int length,
                   hipStream_t s, size_t in_pitch)
{
    dim3 grid((length + FIR_TILE - 1) / FIR_TILE, nframes);
    hipLaunchKernelGGL(rrc_fir_kernel, grid, dim3(FIR_THREADS), 0, s, reinterpret_cast<const float2 *>(x),
                       reinterpret_cast<const float2 *>(memory), reinterpret_cast<float2 *>(y), taps, length,
                       in_pitch ? in_pitch : (size_t)length);
    LAUNCH_CHECK();
    return 0;
}

int launch_delay_line(const float *x, float *memory, int nframes, int length, hipStream_t s)
{
    hipLaunchKernelGGL(delay_line_kernel, dim3(nframes), dim3(128), 0, s, reinterpret_cast<const float2 *>(x),
                       reinterpret_cast<float2 *>(memory), length);
    LAUNCH_CHECK();
    return 0;
}

int launch_timing_hist(const float *y, int nframes, int frame_size, int cycles, int32_t *index, int32_t *hist,
                       bool generic, hipStream_t s)
{
    if (cycles == 8 && frame_size % TH8_TILE == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && !generic) {
        hipLaunchKernelGGL(timing_hist8_kernel, dim3((nframes + TH8_FRAMES - 1) / TH8_FRAMES), dim3(64), 0, s, y,
                           nframes, frame_size, index, hist);
        LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(timing_hist_kernel, dim3((nframes + TH_FRAMES - 1) / TH_FRAMES), dim3(64), 0, s, y, nframes,
                       frame_size, cycles, index, hist);
    LAUNCH_CHECK();
    return 0;
}

int launch_costas(const float *d, int nframes, int nsym, int dstride, int nbw, const float *gains, float min_freq,
                  float max_freq, const float *state_in, float *state_out, uint8_t *sym, float *costas, int *status,
                  hipStream_t s)
{
    const int threads = nframes * nbw;
    hipLaunchKernelGGL(costas_kernel, dim3((threads + 63) / 64), dim3(64), 0, s,
                       reinterpret_cast<const float2 *>(d), nframes, nsym, dstride, nbw, gains, min_freq, max_freq,
                       state_in, state_out, sym, reinterpret_cast<float2 *>(costas), status);
    LAUNCH_CHECK();
    return 0;
}

int launch_decimate(const float *filtered, const int32_t *index, float *dec, int nstreams, int frame_size,
                    int cycles, int nsym, hipStream_t s)
{
    dim3 grid((nsym + 255) / 256, nstreams);
    hipLaunchKernelGGL(decimate_kernel, grid, dim3(256), 0, s, reinterpret_cast<const float2 *>(filtered), index,
                       reinterpret_cast<float2 *>(dec), nstreams, frame_size, cycles, nsym);
    LAUNCH_CHECK();
    return 0;
}

int launch_mixer(const int16_t *pcm, float *out, float *state, int nstreams, int frame_size, hipStream_t s)
{
    hipLaunchKernelGGL(mixer_kernel, dim3((nstreams + MX_STREAMS - 1) / MX_STREAMS), dim3(64), 0, s, pcm,
                       reinterpret_cast<float2 *>(out), state, nstreams, frame_size);
    LAUNCH_CHECK();
    return 0;
}

/* ========================================================================
 * fftn / ifftn beyond what one workgroup's LDS holds (n > 8192, up to the frame lengths of the BASELINE configs:
 * 16384 and 2^20; fft.c:110-136 take any power of two).  The same butterflies in the same order as fft_kernel --
 * the reference's recursion is a bit-reversal followed by log2(n) stages, and WHICH workgroup performs a
 * butterfly does not change its operands -- in two passes over global memory:
 *   fft_big_first_kernel  block b gathers the FFTB = 4096 bit-reversed inputs that make up outputs
 *                         [b*FFTB, (b+1)*FFTB) and runs stages 1..12 in LDS (fft_lds_stages, size-4096 table: a
 *                         stage's angle TAU k/m does not depend on the transform it is part of);
 *   fft_big_late_kernel   stages 13..log2(n) pair elements FFTB*2^j apart: a workgroup takes a tile of 16
 *                         adjacent columns x n/FFTB rows (rows FFTB apart), runs the remaining stages on it in
 *                         LDS with the size-n table (entry m*(n/size), bit-identical to the stage's own angle:
 *                         scaling an angle's argument by a power of two is exact) and stores with the forward
 *                         transform's division by n (fft.c:117-119).
 * ======================================================================== */
constexpr int FFTB_LOG2 = 12, FFTB = 1 << FFTB_LOG2, FFTB_COLS = 16, FFTB_THREADS = 256;

__global__ void __launch_bounds__(FFTB_THREADS)
fft_big_first_kernel(const double2 *__restrict__ in, double2 *__restrict__ out, const double2 *__restrict__ twb, int n,
                     int log2n, int inverse)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *v = reinterpret_cast<double2 *>(smem);
    const int tid = threadIdx.x;
    const double2 *src = in + (size_t)blockIdx.y * n;
    double2 *dst = out + (size_t)blockIdx.y * n;
    const int base = blockIdx.x * FFTB;
    for (int i = tid; i < FFTB; i += FFTB_THREADS)
        v[i] = src[__brev((unsigned)(base + i)) >> (32 - log2n)];
    __syncthreads();
    fft_lds_stages(v, twb, FFTB, FFTB_LOG2, tid, FFTB_THREADS, inverse ? 1.0 : -1.0);
    for (int i = tid; i < FFTB; i += FFTB_THREADS)
        dst[base + i] = v[i];
}

__global__ void __launch_bounds__(FFTB_THREADS)
fft_big_late_kernel(double2 *__restrict__ data, const double2 *__restrict__ twn, int n, int log2n, int inverse)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *t = reinterpret_cast<double2 *>(smem);           /* [rows][FFTB_COLS] */
    const int tid = threadIdx.x;
    const int rows = n >> FFTB_LOG2, lrows = log2n - FFTB_LOG2;
    const int c0 = blockIdx.x * FFTB_COLS;
    double2 *d = data + (size_t)blockIdx.y * n;
    for (int e = tid; e < rows * FFTB_COLS; e += FFTB_THREADS)
        t[e] = d[(size_t)(e / FFTB_COLS) * FFTB + c0 + (e % FFTB_COLS)];
    __syncthreads();
    const double sgn = inverse ? 1.0 : -1.0;
    for (int st = 1; st <= lrows; st++) {
        const int h = 1 << (st - 1);
        const int stride = n >> (FFTB_LOG2 + st);             /* n / (stage size), stage size = 2h * FFTB */
        for (int b = tid; b < (rows / 2) * FFTB_COLS; b += FFTB_THREADS) {
            const int rr = b / FFTB_COLS, c = b % FFTB_COLS;
            const int kr = rr & (h - 1);
            const int lo = (((rr >> (st - 1)) << st) + kr) * FFTB_COLS + c, hi = lo + h * FFTB_COLS;
            const double2 w = twn[(size_t)(kr * FFTB + c0 + c) * stride];   /* k = position inside the stage's first half */
            const double wr = w.x, wi = sgn * w.y;
            const double2 e = t[lo], o = t[hi];
            const double zr = wr * o.x - wi * o.y;
            const double zi = wr * o.y + wi * o.x;
            t[lo] = make_double2(e.x + zr, e.y + zi);
            t[hi] = make_double2(e.x - zr, e.y - zi);
        }
        __syncthreads();
    }
    const double dn = (double)n;
    for (int e = tid; e < rows * FFTB_COLS; e += FFTB_THREADS) {
        double2 r = t[e];
        if (!inverse) { r.x = r.x / dn; r.y = r.y / dn; }    /* fft.c:117-119 */
        d[(size_t)(e / FFTB_COLS) * FFTB + c0 + (e % FFTB_COLS)] = r;
    }
}

int fft_lds_max_n(void) { return 8192; }
int fft_big_block(void) { return FFTB; }

/* n > fft_lds_max_n(): twb = size-FFTB table, twn = size-n table; in != out */
int launch_fft_big(const double *in, double *out, const double *twb, const double *twn, int nbatch, int n, int log2n,
                   int inverse, hipStream_t s)
{
    const int rows = n >> FFTB_LOG2;
    if (n <= FFTB || rows * FFTB_COLS * (int)sizeof(double2) > MAX_LDS_BYTES || in == out) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(fft_big_first_kernel, dim3(n / FFTB, nbatch), dim3(FFTB_THREADS), sizeof(double2) * (size_t)FFTB, s,
                       reinterpret_cast<const double2 *>(in), reinterpret_cast<double2 *>(out),
                       reinterpret_cast<const double2 *>(twb), n, log2n, inverse);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(fft_big_late_kernel, dim3(FFTB / FFTB_COLS, nbatch), dim3(FFTB_THREADS),
                       sizeof(double2) * (size_t)rows * FFTB_COLS, s, reinterpret_cast<double2 *>(out),
                       reinterpret_cast<const double2 *>(twn), n, log2n, inverse);
    LAUNCH_CHECK();
    return 0;
}

int launch_fft(const double *in, double *out, const double *tw, int nbatch, int n, int log2n, int inverse,
               hipStream_t s)
{
    const int threads = n / 2 < 256 ? (n / 2 < 64 ? 64 : n / 2) : 256;
    hipLaunchKernelGGL(fft_kernel, dim3(nbatch), dim3(threads), sizeof(double2) * (size_t)n, s,
                       reinterpret_cast<const double2 *>(in), reinterpret_cast<double2 *>(out),
                       reinterpret_cast<const double2 *>(tw), n, log2n, inverse);
    LAUNCH_CHECK();
    return 0;
}

int launch_fill_i32(int32_t *p, int n, int32_t v, hipStream_t s)
{
    hipLaunchKernelGGL(fill_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, n, v);
    LAUNCH_CHECK();
    return 0;
}

int launch_sincos_hash(uint32_t first, uint32_t count, unsigned long long *acc, hipStream_t s)
{
    hipLaunchKernelGGL(sincos_hash_kernel, dim3(2048), dim3(256), 0, s, first, count, acc);
    LAUNCH_CHECK();
    return 0;
}

} // namespace qpsk
