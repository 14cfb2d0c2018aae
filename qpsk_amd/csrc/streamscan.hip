/*
 * streamscan.hip -- a block of MANY running streams from PCM to the timing index in ONE kernel: the PCM -> complex mix of
 * qpsk.c:114-120, rrc_fir() with the streams' delay lines (qpsk.c:125, rrc_fir.c:17-30) and the reference's histogram timing
 * estimate (qpsk.c:127-180), PCM in at 2 bytes per sample, nothing but the filtered block (for the picks of qpsk.c:186-191), the
 * index and the carried state out.  What follows it is costas_pipe_kernel: the loop over the previous block's picks, whose loader
 * waves take this block's picks from the filtered block.
 *
 * Why: as kernels of their own the mixer costs 0.33 ms per 16384-sample block whatever the number of streams (`phase *= rect` is a
 * serial complex recurrence per stream: 16384 dependent steps during which the chip idles), the mixed block makes a round trip
 * through HBM, and the scan reads the filtered block once more.  Here (timing_scan_kernel's pipeline, timing_scan.hip, plus a mixer):
 *
 * Workgroup = 16 streams = 13 waves, one workgroup per CU at 4096 streams:
 *   waves 0-3    scan waves, 4 streams each: 8 lanes per (stream, I|Q) running the average / max chain on the tiles of an LDS ring;
 *   waves 4-11   filter waves, 2 streams each: PCM pairs (one 4-byte load per lane, stream and 128-sample block, a tile ahead), the
 *                carrier phases of the tile from the phase ring, qpsk.c:117's scale and multiply, window staged from registers
 *                (history = the last mixed 128-sample block, or the stream's delay line), the full-rate stream with the taps in
 *                SGPRs (fir_full8s_asm.h), the second GAIN, the lane's symbol (8 consecutive outputs) to the scan ring AND to the
 *                filtered block in memory, PLANAR BY DECIMATION PHASE ([stream][phase][symbol]: every store a coalesced row, and
 *                the pick of index i later reads one contiguous plane, an eighth of the block);
 *   wave 12      the mixer: lane s runs the carrier recurrence of stream s a tile ahead (three packed operations per sample) and
 *                leaves the tile's 256 phases in the stream's row of a one-tile ring; it starts tile t + 1 once every filter wave
 *                has picked tile t up -- i.e. while they filter it.
 * LDS: windows 61,440 + scan ring 66,560 + phase ring 33,024 + counters: 161 KB of the CU's 160 KiB (163,840 bytes).
 * While all streams share ONE carrier (the state qpsk_streams_reset() leaves, and the carrier does not depend on the data) wave 12 has
 * nothing to do per workgroup: the block's phases come from a table the call before prepared (MODE 2 below, carrier.h).
 *
 * Served (host-checked): CYCLES = 8 or 4 (the description above is for 8; at 4 a lane's 8 outputs are two symbols), frame_size a multiple of 256, symmetric taps, 4-byte aligned PCM rows, 16-byte aligned output.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qpsk_device.h"
#include "costas_asm.h"      /* lds_addr() */
#include "fir_full8s_asm.h"
#include "carrier.h"
#include "kernels.h"

namespace qpsk {

namespace sscan {
constexpr int G = 16;                 /* streams per workgroup */
constexpr int NSCAN = 4, NFIR = 8;    /* scan waves (4 streams each), filter waves (2 streams each); + 1 mixer wave */
constexpr int UF = 2, QL = 32, R = 8; /* streams per filter wave, lanes per stream, outputs per lane */
constexpr int TILE = QL * R;          /* 256 outputs per stream per round */
constexpr int DRO = 2;                /* tiles in the scan ring */
constexpr int PADS = 2;               /* window: position p at slot p + 2 (p / 8): lanes 80 bytes apart */
constexpr int WSF = 480;              /* slots per stream window (positions 0..381 -> slots 0..475) */
constexpr int PITCH = TILE + 4;       /* floats per plane row of the scan ring */
constexpr int PROW = TILE + 2;        /* float2 slots per phase row: 16 bytes of padding (the mixer's lanes store to different banks) */
constexpr int THREADS = 64 * (NSCAN + NFIR + 1);
constexpr int SPIN_LIMIT = 1 << 24;
constexpr int CARRIER_RELAY = 8;      /* MODE 2: workgroups whose spare waves share stream_scan_kernel's stretch of the next carrier table */
__device__ __host__ constexpr int slot_of(int p) { return p + PADS * (p / R); }
static_assert(slot_of(TILE + HIST - 1) < WSF && FIR_FULL8S_ASM_END_VGPR <= 128, "window geometry; 13 waves = four per SIMD");

struct Smem {
    int ready[NFIR];          /* tiles each filter wave has handed to the scan ring */
    int consumed[NSCAN];      /* tiles each scan wave has finished */
    int mcons[NFIR];          /* tiles of phases each filter wave has picked up */
    int mready;               /* tiles of phases the mixer has produced */
    int abort_flag;
    int pad_[2];
};
constexpr size_t LDS_BYTES = sizeof(Smem) + sizeof(float2) * (size_t)G * WSF + sizeof(float) * (size_t)G * DRO * 2 * PITCH +
                             sizeof(float2) * (size_t)G * PROW;
static_assert(sizeof(Smem) % 16 == 0 && LDS_BYTES <= (size_t)MAX_LDS_BYTES, "everything fits the CU's LDS");

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
} // namespace sscan

/* the first block's table (qpsk_streams_reset); the second part of a table when no loop kernel with a spare wave follows */
__global__ void __launch_bounds__(64) carrier_table_kernel(float *st, float2 *tab, int frame_size, int from, int to)
{
    if (threadIdx.x == 0) carrier_block(st, tab, frame_size, from, to);
}

/* leaving the shared carrier: every stream's own mixer state = the phase the next block starts from */
__global__ void __launch_bounds__(256) carrier_broadcast_kernel(const float *st, float *mixer, int nstreams)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f < nstreams) {
        mixer[4 * f] = st[4];
        mixer[4 * f + 1] = st[5];
    }
}

/* MODE 0: the streams' blocks are already complex (qpsk_streams_rx_cplx: they enter at the rrc_fir() call) -- no mixer wave, the
 * filter waves load 16-byte pairs a tile ahead.
 * MODE 1: PCM, every stream with its own carrier state (the mixer wave described above).
 * MODE 2: PCM, ONE carrier for all streams -- what qpsk_streams_reset() sets up (one mixer frequency, one starting phase) and what
 * stays true as long as every block of the streams comes through here: the carrier does not depend on the data.  Its 16384 dependent
 * steps per block are then run ONCE (carrier.h: the first 11/16 by a spare wave of workgroup 0 here, the rest by a spare wave of the
 * loop kernel that follows), for the NEXT block (ctab_next), while every workgroup's filter waves take this block's phases from the
 * table the call before left (ctab, 128 KB, read through the caches): no mixer wave paces the filter waves (256 steps per tile at
 * ~110 cycles beside three busy waves was what MODE 1 waits for) and no phase ring. */
template <int MODE>
__global__ void __launch_bounds__(MODE ? sscan::THREADS : sscan::THREADS - 64)
stream_scan_kernel(const int16_t *__restrict__ pcm, const float2 *__restrict__ x, float *mixer, float2 *memory, float2 *__restrict__ yout,
                   const float *__restrict__ taps_g, int32_t *index, int nstreams, int frame_size, int *status,
                   const float2 *__restrict__ ctab, float2 *ctab_next, float *cstate, unsigned cseq, int cycles)
{
    using namespace sscan;
    constexpr bool PCM = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem *sm = reinterpret_cast<Smem *>(smem_raw);
    float2 *win = reinterpret_cast<float2 *>(smem_raw + sizeof(Smem));                 /* [G][WSF] */
    float *ring = reinterpret_cast<float *>(win + (size_t)G * WSF);                    /* [G][DRO][2 planes][PITCH] */
    float2 *phases = reinterpret_cast<float2 *>(ring + (size_t)G * DRO * 2 * PITCH);  /* [G][PROW] */
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f0 = blockIdx.x * G;
    /* CYCLES = 8 (a lane's 8 outputs are one symbol) or 4 (the reference's shipped rates: two symbols): the filter is the same; the
     * scan's symbols, the planes of the filtered block and the symbol count differ */
    const int ntiles = frame_size / TILE, nsym = frame_size / cycles;

    if (tid < NFIR) { sm->ready[tid] = 0; sm->mcons[tid] = 0; }
    if (tid < NSCAN) sm->consumed[tid] = 0;
    if (tid == 0) { sm->abort_flag = 0; sm->mready = 0; }
    __syncthreads();

    if (MODE == 2 && wave == NSCAN + NFIR) {
        /* ============================ the spare wave: workgroup 0's runs the carrier of the NEXT block ====================== */
        /* The chain costs the SIMD it runs on 12 issue cycles per step, which its workgroup's filter waves miss: spread over the
         * spare waves of the first CARRIER_RELAY workgroups, one stretch each, in turn (a counter in memory: stretch j of block
         * number cseq starts when the counter reads cseq * parts + j; workgroups are dispatched in order, so the one waited for is
         * resident or done).  A relay wave that is not yet on sleeps; the stretch before it takes ~0.06 ms. */
        const int parts = min(CARRIER_RELAY, (int)gridDim.x);
        if ((int)blockIdx.x < parts && lane == 0) {
            const int split = carrier_split(frame_size), per = ((split + parts - 1) / parts + 3) & ~3;
            const int from = min((int)blockIdx.x * per, split), to = min(from + per, split);
            /* unsigned: the launch count since the reset wraps modulo 2^32 with defined arithmetic (2^28 launches of a signed turn were
             * ~3 h of 4096 x 512 blocks); parts is the same at every launch between two resets, so the sequence stays consistent */
            unsigned *ctr = reinterpret_cast<unsigned *>(cstate + 6);
            const unsigned turn = cseq * (unsigned)parts + blockIdx.x;
            int spins = 0;
            bool ok = true;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != turn) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > SPIN_LIMIT) { ok = false; break; }
            }
            if (ok) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __builtin_amdgcn_s_setprio(3);
                carrier_block(cstate, ctab_next, frame_size, from, to);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(ctr, turn + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(status, STATUS_PIPE_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    if (MODE == 1 && wave == NSCAN + NFIR) {
        /* ===================================== the mixer wave: lane s = stream f0 + s ================================= */
        __builtin_amdgcn_s_setprio(3);      /* a latency chain: few issue slots, wanted at once */
        const bool mine = lane < G && f0 + lane < nstreams;
        const int f = min(f0 + (lane < G ? lane : 0), nstreams - 1);
        float2 p = make_float2(mixer[4 * f], mixer[4 * f + 1]);
        const float rr = mixer[4 * f + 2], ri = mixer[4 * f + 3];
        float nri = -ri;
        asm volatile("" : "+v"(nri));   /* opaque: keeps the compiler from folding the sign back into two packed adds */
        auto step = [&]() {                                  /* fbb_rx_phase *= fbb_rx_rect, qpsk.c:115 (mixer_kernel's step) */
            const float2 a = make_float2(p.x * rr, p.y * rr);
            const float2 b = make_float2(p.y * nri, p.x * ri);   /* p.y * (-ri) = -(p.y * ri) exactly: re = a.x - p.y*ri */
            p = make_float2(a.x + b.x, a.y + b.y);
        };
        float4 *prow = reinterpret_cast<float4 *>(phases + (size_t)(lane < G ? lane : 0) * PROW);
        bool ok = true;
        for (int t = 0; t < ntiles && ok; t++) {
            if (t >= 1) {      /* the phase ring holds ONE tile: every filter wave must have picked tile t - 1 up */
                int spins = 0;
                for (;;) {
                    const int c = lane < NFIR ? __hip_atomic_load(&sm->mcons[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0x7fffffff;
                    if (__all(c >= t)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT || __hip_atomic_load(&sm->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        __hip_atomic_store(&sm->abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        ok = false;
                        break;
                    }
                }
                if (!ok) break;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            if (mine) {
#pragma unroll 8
                for (int i = 0; i < TILE / 2; i++) {      /* two phases per 16-byte store */
                    step();
                    const float2 p0 = p;
                    step();
                    prow[i] = make_float4(p0.x, p0.y, p.x, p.y);
                }
            }
            if (lane == 0) publish(&sm->mready, t + 1);
        }
        if (mine && ok) {
            const float mag = (float)sqrt((double)p.x * (double)p.x + (double)p.y * (double)p.y);   /* qpsk.c:120 */
            mixer[4 * f] = p.x / mag;
            mixer[4 * f + 1] = p.y / mag;
        }
        if (!ok && lane == 0) __hip_atomic_store(status, STATUS_PIPE_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }

    if (wave < NSCAN) {
        /* ================================ scan wave: streams 4*wave .. 4*wave+3 of the workgroup (timing_scan_kernel's) ====== */
        /* (workgroup 0's first scan wave shares its SIMD with the carrier wave of MODE 2, the longer chain of the two: it steps back) */
        if (MODE == 2 && blockIdx.x == 0 && wave == 0) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
        const int fl = lane >> 4, comp = (lane >> 3) & 1, q = lane & 7;      /* lane = 16*stream + 8*component + q */
        const int g = 4 * wave + fl;
        const float qf = (float)q;
        float av = 0.0f, mx = 0.0f;
        int cum = 0;
        bool ok = true;
        for (int t = 0; t < ntiles && ok; t++) {
            ok = wait_ge(&sm->ready[g >> 1], t + 1, &sm->abort_flag);
            if (!__all(ok)) { ok = false; break; }
            const float4 *row = reinterpret_cast<const float4 *>(ring + ((size_t)(g * DRO + (t % DRO)) * 2 + comp) * PITCH);
            if (cycles == 8) {
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
            } else {      /* CYCLES = 4: four samples per symbol, / 4 is * 0.25 exactly; the histogram keeps its 8 bins (qpsk.c:130) */
#pragma unroll 4
                for (int s = 0; s < TILE / 4; s++) {
                    const float4 a = row[s];
                    av += fabsf(a.x); av += fabsf(a.y); av += fabsf(a.z); av += fabsf(a.w);
                    av *= 0.25f;
                    if (av > mx) mx = av;
                    const float th = (mx * 0.125f) * qf;
                    cum += (av <= th) ? 0 : 1;
                }
            }
            if (lane == 0) publish(&sm->consumed[wave], t + 1);
        }
        if (!ok) {
            if (lane == 0) __hip_atomic_store(status, STATUS_PIPE_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        if (q == 0) cum = frame_size / cycles;
        int h = __shfl_up(cum, 1) - cum;
        if (q == 0) h = 0;
        h += __shfl_xor(h, 8);
        int hmax = 0, best = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int hk = __shfl(h, (lane & ~7) + k);
            if (hk > hmax) { hmax = hk; best = k; }
        }
        if (f0 + g < nstreams && comp == 0 && q == 0) index[f0 + g] = best;
        return;
    }

    /* ==================================== filter wave: streams 2*w, 2*w+1 of the workgroup ============================== */
    const int w = wave - NSCAN;
    const int fl = lane / QL, q = lane % QL;
    const int g = UF * w + fl;
    const bool fv[UF] = {f0 + UF * w < nstreams, f0 + UF * w + 1 < nstreams};
    const uint32_t *src[UF];      /* PCM pairs */
    const float4 *srcx[UF];       /* complex pairs */
    float2 *mem[UF];
#pragma unroll
    for (int ff = 0; ff < UF; ff++) {
        const int f = fv[ff] ? f0 + UF * w + ff : 0;
        src[ff] = reinterpret_cast<const uint32_t *>(pcm + (size_t)f * frame_size);
        srcx[ff] = reinterpret_cast<const float4 *>(x + (size_t)f * frame_size);
        mem[ff] = memory + (size_t)f * NTAPS;
    }
    float2 *mywin = win + (size_t)(UF * w) * WSF;
    const unsigned rd_addr = lds_addr(mywin + fl * WSF + (R + PADS) * q);   /* position 8q -> slot 10q */
    const int p0 = 2 * lane + HIST;      /* window position of sample 2 lane of a tile */
    float4 hist[UF], pre[UF][2], nxt[UF][2];
    float4 pc[2];                 /* MODE 2: the phases of the lane's two sample pairs, a tile ahead like the PCM */
    uint32_t pv[UF][2];
    const float4 *ctab4 = reinterpret_cast<const float4 *>(ctab);
#pragma unroll
    for (int ff = 0; ff < UF; ff++) {
        /* the delay line (memory[0] is shifted out by the first step, rrc_fir.c:19): the lane's pair of the "block" in front of the
         * first tile = samples 2 lane - 128, + 1 = memory[2 lane - 1], memory[2 lane] (lane 0's pair is never staged) */
        hist[ff] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (fv[ff] && lane >= 1) {
            const float2 a = mem[ff][2 * lane - 1], b = mem[ff][2 * lane];
            hist[ff] = make_float4(a.x, a.y, b.x, b.y);
        }
    }
    auto prefetch = [&](int t) {
        if constexpr (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 2; j++) pc[j] = ctab4[(t * TILE + 128 * j) / 2 + lane];
        }
#pragma unroll
        for (int ff = 0; ff < UF; ff++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if constexpr (PCM) pv[ff][j] = src[ff][(t * TILE + 128 * j) / 2 + lane];      /* samples 128 j + 2 lane, + 1 */
                else nxt[ff][j] = srcx[ff][(t * TILE + 128 * j) / 2 + lane];
            }
    };
    prefetch(0);
    bool ok = true;
    for (int t = 0; t < ntiles && ok; t++) {
        {   /* the two filter waves of a SIMD keep pace (timing_scan_kernel): the one that is behind goes first */
            const int pt = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sm->ready[w ^ 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (pt > t) __builtin_amdgcn_s_setprio(2);
            else if (pt < t) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(1);
        }
        if constexpr (MODE == 2) {
            /* qpsk.c:117 with the shared carrier's phases */
#pragma unroll
            for (int ff = 0; ff < UF; ff++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float4 c = pc[j];
                    const float v0 = (float)(int16_t)(pv[ff][j] & 0xffffu) / 16384.0f, v1 = (float)(int16_t)(pv[ff][j] >> 16) / 16384.0f;
                    pre[ff][j] = fv[ff] ? make_float4(c.x * v0, c.y * v0, c.z * v1, c.w * v1) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
        } else if constexpr (MODE == 1) {
            /* this tile's carrier phases, then qpsk.c:117: input_frame[i] = fbb_rx_phase * ((float) in[i] / 16384.0f) */
            ok = wait_ge(&sm->mready, t + 1, &sm->abort_flag);
            if (!ok) break;
#pragma unroll
            for (int ff = 0; ff < UF; ff++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float4 c = *reinterpret_cast<const float4 *>(phases + (size_t)(UF * w + ff) * PROW + 128 * j + 2 * lane);
                    const float v0 = (float)(int16_t)(pv[ff][j] & 0xffffu) / 16384.0f, v1 = (float)(int16_t)(pv[ff][j] >> 16) / 16384.0f;
                    pre[ff][j] = fv[ff] ? make_float4(c.x * v0, c.y * v0, c.z * v1, c.w * v1) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            if (lane == 0) publish(&sm->mcons[w], t + 1);      /* (the fence waits for the phase reads) */
        } else {
#pragma unroll
            for (int ff = 0; ff < UF; ff++)
#pragma unroll
                for (int j = 0; j < 2; j++) pre[ff][j] = fv[ff] ? nxt[ff][j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
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
        else {      /* the delay line the block leaves behind: the last 127 mixed samples = samples 129..255 of this tile */
#pragma unroll
            for (int ff = 0; ff < UF; ff++)
                if (fv[ff]) {
                    if (lane >= 1) mem[ff][2 * lane - 1] = make_float2(pre[ff][1].x, pre[ff][1].y);
                    mem[ff][2 * lane] = make_float2(pre[ff][1].z, pre[ff][1].w);
                }
        }
        v2f a0, a1, a2, a3, a4, a5, a6, a7;
        fir_full8s_asm(rd_addr, taps_g, a0, a1, a2, a3, a4, a5, a6, a7);
        /* the ring slot is free once the scan wave of this stream has finished tile t - DRO */
        if (t >= DRO) ok = wait_ge(&sm->consumed[g >> 2], t - DRO + 1, &sm->abort_flag);
        if (!ok) break;
        /* rrc_fir.c:28: y * GAIN in double, narrowed; the lane's 8 outputs are one symbol */
        const float2 y0 = fir_gain(make_float2(a0.x, a0.y)), y1 = fir_gain(make_float2(a1.x, a1.y)),
                     y2 = fir_gain(make_float2(a2.x, a2.y)), y3 = fir_gain(make_float2(a3.x, a3.y)),
                     y4 = fir_gain(make_float2(a4.x, a4.y)), y5 = fir_gain(make_float2(a5.x, a5.y)),
                     y6 = fir_gain(make_float2(a6.x, a6.y)), y7 = fir_gain(make_float2(a7.x, a7.y));
        if (fv[fl]) {      /* the filtered block, planar by decimation phase: [stream][phase][symbol] */
            if (cycles == 8) {
                float2 *o = yout + ((size_t)(f0 + g) * R) * nsym + (size_t)t * QL + q;
                o[0 * (size_t)nsym] = y0; o[1 * (size_t)nsym] = y1; o[2 * (size_t)nsym] = y2; o[3 * (size_t)nsym] = y3;
                o[4 * (size_t)nsym] = y4; o[5 * (size_t)nsym] = y5; o[6 * (size_t)nsym] = y6; o[7 * (size_t)nsym] = y7;
            } else {      /* four planes of frame_size / 4 symbols; the lane's two symbols sit side by side in each: 16-byte stores */
                float2 *o = yout + (size_t)(f0 + g) * frame_size + (size_t)t * (2 * QL) + 2 * q;
                *reinterpret_cast<float4 *>(o + 0 * (size_t)nsym) = make_float4(y0.x, y0.y, y4.x, y4.y);
                *reinterpret_cast<float4 *>(o + 1 * (size_t)nsym) = make_float4(y1.x, y1.y, y5.x, y5.y);
                *reinterpret_cast<float4 *>(o + 2 * (size_t)nsym) = make_float4(y2.x, y2.y, y6.x, y6.y);
                *reinterpret_cast<float4 *>(o + 3 * (size_t)nsym) = make_float4(y3.x, y3.y, y7.x, y7.y);
            }
        }
        float *pi = ring + ((size_t)(g * DRO + (t % DRO)) * 2 + 0) * PITCH + R * q;
        float *pq = pi + PITCH;
        reinterpret_cast<float4 *>(pi)[0] = make_float4(y0.x, y1.x, y2.x, y3.x);
        reinterpret_cast<float4 *>(pi)[1] = make_float4(y4.x, y5.x, y6.x, y7.x);
        reinterpret_cast<float4 *>(pq)[0] = make_float4(y0.y, y1.y, y2.y, y3.y);
        reinterpret_cast<float4 *>(pq)[1] = make_float4(y4.y, y5.y, y6.y, y7.y);
        if (lane == 0) publish(&sm->ready[w], t + 1);
    }
    if (!ok && lane == 0) __hip_atomic_store(status, STATUS_PIPE_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int stream_scan_tile(void) { return sscan::TILE; }

int prepare_stream_scan(void)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(stream_scan_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(stream_scan_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(stream_scan_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_BYTES);
}

/* the shared carrier (MODE 2): cstate as kernels.h describes it */
int launch_carrier_table(float *cstate, float *tab, int frame_size, bool rest_only, hipStream_t s)
{
    hipLaunchKernelGGL(carrier_table_kernel, dim3(1), dim3(64), 0, s, cstate, reinterpret_cast<float2 *>(tab), frame_size,
                       rest_only ? carrier_split(frame_size) : 0, frame_size / 2);
    return (int)hipGetLastError();
}

int launch_carrier_broadcast(const float *cstate, float *mixer, int nstreams, hipStream_t s)
{
    hipLaunchKernelGGL(carrier_broadcast_kernel, dim3((nstreams + 255) / 256), dim3(256), 0, s, cstate, mixer, nstreams);
    return (int)hipGetLastError();
}

/* CYCLES = 8 or 4, frame_size % 256 == 0, symmetric taps, 16-byte aligned yout (the caller checks); exactly one of pcm (4-byte aligned rows;
 * mixer state updated) and x (16-byte aligned complex blocks); yout [nstreams][cycles][frame_size / cycles], cycles 8 or 4 */
int launch_stream_scan(const int16_t *pcm, const float *x, float *mixer, float *memory, float *yout, const float *taps, int32_t *index,
                       int nstreams, int frame_size, int *status, hipStream_t s, const float *ctab, float *ctab_next, float *cstate, unsigned cseq,
                       int cycles)
{
    using namespace sscan;
    if (frame_size % TILE != 0 || (cycles != 8 && cycles != 4) || (pcm != nullptr) == (x != nullptr) || (reinterpret_cast<uintptr_t>(pcm) & 3) != 0 ||
        (reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(yout) & 15) != 0 ||
        (reinterpret_cast<uintptr_t>(ctab) & 15) != 0 || (reinterpret_cast<uintptr_t>(ctab_next) & 15) != 0 ||
        (ctab != nullptr && (!pcm || !ctab_next || !cstate)))
        return (int)hipErrorInvalidValue;
    const dim3 grid((nstreams + G - 1) / G);
    if (pcm && ctab)
        hipLaunchKernelGGL(stream_scan_kernel<2>, grid, dim3(THREADS), LDS_BYTES, s, pcm, nullptr, mixer, reinterpret_cast<float2 *>(memory),
                           reinterpret_cast<float2 *>(yout), taps, index, nstreams, frame_size, status,
                           reinterpret_cast<const float2 *>(ctab), reinterpret_cast<float2 *>(ctab_next), cstate, cseq, cycles);
    else if (pcm)
        hipLaunchKernelGGL(stream_scan_kernel<1>, grid, dim3(THREADS), LDS_BYTES, s, pcm, nullptr, mixer, reinterpret_cast<float2 *>(memory),
                           reinterpret_cast<float2 *>(yout), taps, index, nstreams, frame_size, status, nullptr, nullptr, nullptr, 0u, cycles);
    else
        hipLaunchKernelGGL(stream_scan_kernel<0>, grid, dim3(THREADS - 64), LDS_BYTES, s, nullptr, reinterpret_cast<const float2 *>(x), mixer,
                           reinterpret_cast<float2 *>(memory), reinterpret_cast<float2 *>(yout), taps, index, nstreams, frame_size, status,
                           nullptr, nullptr, nullptr, 0u, cycles);
    return (int)hipGetLastError();
}

} // namespace qpsk
