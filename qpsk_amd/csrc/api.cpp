/*
 * api.cpp -- the C ABI of include/qpsk_hip.h: contexts, argument checking,
 * scratch management and kernel sequencing.  No arithmetic of the receive
 * path happens here: per-configuration numbers come from host_math.c (as in
 * the reference, on the host), everything per sample runs in kernels.hip.
 * There is deliberately no CPU fallback: without a usable GPU every entry
 * point fails with QPSK_ERR_NO_DEVICE / QPSK_ERR_HIP.
 */
#include "../../include/qpsk_hip.h"
#include "host_math.h"
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <map>
#include <vector>

using namespace qpsk;

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(QPSK_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define KERNEL_TRY(expr)                                                                      \
    do {                                                                                      \
        (void)hipGetLastError(); /* a stale error of an unrelated earlier call must not be blamed on this launch */ \
        int e_ = (expr);                                                                      \
        if (e_ != 0)                                                                          \
            return fail(QPSK_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString((hipError_t)e_), __FILE__, __LINE__); \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

/*
 * Tuning: choices that select WHICH kernel geometry computes a result, never the result (every geometry is bit-exact
 * against the oracle; tests/test_gpu_parity.py runs them all).  -1 = the library's own choice.  The QPSK_* environment
 * variables of the same names are read ONCE, in qpsk_ctx_create(); qpsk_ctx_set_tuning() overrides them per context.
 * Nothing on a call path reads the environment.
 */
struct Tuning {
    int fused_g = -1, fused_s = -1, fused_lds = -1;   /* generic chunked kernel: frames per workgroup, symbols per chunk, LDS budget */
    int generic = -1;                                 /* 1: the barrier-synchronised generic kernels instead of the pipeline */
    int pipe_nf = -1;                                 /* rx_fused_pipe_kernel: FIR waves per workgroup */
    int pipe_v = -1, pipe_g = -1;                     /* 1: rx_fused_pipe_kernel, 2: rx_pipe2_kernel; frames per workgroup of the latter */
    int layout_lo = -1, layout_hi = -1;               /* rx_pipe2_kernel: units per hardware wave, 4 bits each (waves 1-5 / 6-11) */
    int pipe_variant = -1;                            /* pipeline kernel: layout bits (kernels.h, FusedArgs::dbg) */
    int hist_generic = -1;                            /* histogram estimate: 1 = rrc_fir + scan kernels (no fused scan), 2 = those with the generic scan */
    int fir_generic = -1;                             /* full-rate rrc_fir(): 1 = the compiler-scheduled rrc_fir_kernel also for symmetric taps */
    int stream_scan = -1;                             /* streams from PCM, histogram timing: 1 = stream_scan_kernel (mixer + filter + scan) whatever the stream count, 0 = never */
    int stream_poll = -1;                             /* qpsk_streams_rx_pcm_host on the one-launch kernel: 0 = wait with hipStreamSynchronize instead of watching the kernel's counter */
    int stream_block = -1;                            /* streams: 0 = never the one-launch-per-block kernel (streamblock.hip), 1 = whenever the shape allows, unset: by samples per block of all streams (stream_block_ok: 3.5 M for PCM, 0.4 M for complex input at CYCLES 8) */
    int stream_carrier = -1;                          /* stream_scan_kernel on PCM: 0 = every stream runs its own carrier (mixer wave) even while all streams share one */
    int lean_dma = -1;                                /* rx_lean_kernel: 0 = window staging through registers even where LDS-DMA applies (even decimation offsets), 2 = LDS-DMA with one window per FIR wave */
    int hist_onepass = -1;                            /* histogram timing: 0 = never the one-pass route (rx_hist_kernel on the previous batch's majority index + a fall-back pass), 1 = whenever the shape allows and a guess exists (unset: only while every frame of the last batch sat on its majority index) */
    int est_waves = -1;                               /* rx_lean_kernel, in-launch FFT estimate: hardware waves launched for it */
    int lean_pair = -1;                               /* rx_lean_kernel: 0 = one lane per loop in the serial wave, 1 = two lanes per loop up to 16 frames per workgroup, 2 = up to 32; unset: up to 24, where it pays in steady state (profiles/r06_step_cost.txt) */
    int fft_fused = -1;                               /* FFT timing estimate: 0 = always a launch of its own (1 / unset: inside rx_fused_pipe_kernel's launch for full workgroups) */
};

static const struct { const char *name; int Tuning::*field; } TUNING_KEYS[] = {
    {"QPSK_FUSED_G", &Tuning::fused_g},       {"QPSK_FUSED_S", &Tuning::fused_s},     {"QPSK_FUSED_LDS", &Tuning::fused_lds},
    {"QPSK_FUSED_GENERIC", &Tuning::generic}, {"QPSK_PIPE_NF", &Tuning::pipe_nf},
    {"QPSK_PIPE_DBG", &Tuning::pipe_variant}, {"QPSK_HIST_GENERIC", &Tuning::hist_generic},
    {"QPSK_PIPE_V", &Tuning::pipe_v},         {"QPSK_PIPE_G", &Tuning::pipe_g},
    {"QPSK_PIPE_LAYOUT_LO", &Tuning::layout_lo}, {"QPSK_PIPE_LAYOUT_HI", &Tuning::layout_hi},
    {"QPSK_FIR_GENERIC", &Tuning::fir_generic}, {"QPSK_FFT_FUSED", &Tuning::fft_fused},
    {"QPSK_STREAM_BLOCK", &Tuning::stream_block}, {"QPSK_STREAM_POLL", &Tuning::stream_poll},
    {"QPSK_STREAM_SCAN", &Tuning::stream_scan}, {"QPSK_STREAM_CARRIER", &Tuning::stream_carrier},
    {"QPSK_LEAN_DMA", &Tuning::lean_dma},     {"QPSK_LEAN_PAIR", &Tuning::lean_pair},
    {"QPSK_EST_WAVES", &Tuning::est_waves},   {"QPSK_HIST_ONEPASS", &Tuning::hist_onepass},
};

/* layout bits a product build honours: 4 no spare waves, 8 C++ Costas step, 64/128 lane-mapping variants.  The
 * result-changing ablation bits (1 skip the filter arithmetic, 2 skip the recurrence) and the cycle accounting (32)
 * exist only in the measurement build (make profile, -DQPSK_PIPE_PROFILE) */
#ifdef QPSK_PIPE_PROFILE
static const int PIPE_VARIANT_MASK = ~0;
#else
static const int PIPE_VARIANT_MASK = 4 | 8 | 16 | 64 | 128 | 256 | 512 | 1024 | 8192;
#endif

struct qpsk_ctx {
    int device = 0;
    Tuning tune;
    int ncu = 256;                /* compute units of the device (256 on MI355X) */
    const char *last_kernel = ""; /* the receive kernel the last rx batch launched (qpsk_ctx_last_kernel) */
    bool taps_symmetric = false;  /* taps[k] == taps[126 - k] bit for bit: rx_lean_kernel keeps the 64 distinct ones in SGPRs */
    hipStream_t stream = nullptr; /* caller's stream; nullptr = default stream */
    qpsk_params prm{};
    int cycles = 0, nsym = 0;
    float taps[QPSK_NTAPS];
    float damping = 0.f, alpha = 0.f, beta = 0.f;
    float min_freq = 0.f, max_freq = 0.f;
    float *d_taps = nullptr;     /* 128 floats */
    float *d_gains = nullptr;    /* MAX_BW x (alpha, beta) */
    /* Status word, in pinned host memory that the device can write: a kernel stores a nonzero STATUS_* there when
     * its results are invalid (its in-LDS pipeline exhausted the bounded spins; a loop phase beyond the bounded
     * 2 pi wrap).  The host looks at it after EVERY synchronisation the library performs (check_status), so no
     * entry point that hands results to the caller can return QPSK_OK over invalid ones. */
    unsigned *h_done = nullptr, *d_done = nullptr;   /* stream_block_kernel's wave counter (mapped pinned memory), host / device view */
    unsigned done_expect = 0;
    int *h_status = nullptr;     /* host view */
    int *d_status = nullptr;     /* device view of the same word */
    std::vector<float> h_gains;
    DevBuf index, filtered, mixed, keystream, sympad, mislist;
    /* the one-pass histogram route (rx_hist_kernel): d_hint[0] = the guessed decimation offset = the majority index of the context's last
     * histogram-mode batch (left there by index_majority_kernel, in stream order: no synchronisation), d_hint[1] = the frames the guess
     * missed in the running call; h_hist_stats (pinned, written by that kernel) = {majority, frames, missed} of the last batch whose
     * kernels have completed: the host reads it to stay away from the route while the guess misses many frames */
    int32_t *d_hint = nullptr;
    int32_t *h_hist_stats = nullptr, *d_hist_stats = nullptr;
    bool hint_valid = false;
    int keystream_len = 0;
    std::map<int, double *> twiddles;
    float *d_fast = nullptr;      /* qpsk_rrc_fir_batch_fast: H[512][2] then tw[512][2]; rebuilt when the taps change */
    bool fast_valid = false;
    /* streams */
    int nstreams = 0;
    float *s_memory = nullptr, *s_dec = nullptr, *s_loop = nullptr, *s_mixer = nullptr;
    /* the streams' carrier while it is ONE carrier (streamscan.hip MODE 2): qpsk_streams_reset() gives every stream the same mixer
     * frequency and starting phase, and the carrier does not depend on the data, so it stays one as long as every PCM block goes
     * through stream_scan_kernel.  s_ctab: two tables of frame_size phases (this block's, the next block's), s_cstate: kernels.h.
     * Any other consumer of s_mixer first gets the per-stream state back (carrier_to_streams) and the sharing ends until the next reset */
    float *s_ctab = nullptr, *s_cstate = nullptr;
    bool carrier_shared = false;
    unsigned carrier_blocks = 0;        /* blocks taken from the tables since the reset (which of the two holds the current one) */
    unsigned carrier_scans = 0;         /* stream_scan_kernel launches among them (its relay counter, kernels.h) */
    float *carrier_pending = nullptr;   /* the table stream_scan_kernel has begun: the loop kernel of the same call finishes it (carrier.h) */
    bool stream_work_unchecked = false; /* a stream call has enqueued kernels whose status word no synchronisation has looked at yet */
    bool streams_poisoned = false;      /* a stream call failed between its launches (or a kernel gave up): carried state is undefined until the next reset */
    /* host-pointer streaming call (qpsk_streams_rx_pcm_host: what the drop-in rx_frame() uses): pinned staging on the
     * host, matching arena on the device; sized for nstreams blocks */
    unsigned char *h_stage = nullptr, *d_stage = nullptr;
    size_t stage_cap = 0;
    /* transmitters (N2) */
    int ntx = 0;
    uint8_t *t_hist = nullptr;    /* [ntx][tx_history_symbols()]: the symbols still inside tx_filter */
    float *t_mixer = nullptr;     /* [ntx][4]: carrier phase and step */
    DevBuf tx_b;                  /* shaped baseband when the caller does not want it */
};

static const int MAX_BW = 64;

static int ensure(qpsk_ctx *c, DevBuf &b, size_t bytes)
{
    if (b.cap >= bytes) return QPSK_OK;
    if (b.p) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(QPSK_ERR_ALLOC, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    b.cap = bytes;
    return QPSK_OK;
}

/* after a synchronisation of the context's stream: did a kernel of the calls just completed flag its results? */
static int check_status(qpsk_ctx *c)
{
    const int st = __atomic_exchange_n(c->h_status, 0, __ATOMIC_ACQ_REL);
    if (st != STATUS_PIPE_TIMEOUT) c->stream_work_unchecked = false;
    if (st == STATUS_PIPE_TIMEOUT) {
        /* a kernel gave up a bounded wait.  Running streams are affected only if a STREAM call has enqueued work since the status word was
         * last looked at (device-pointer stream calls do not synchronise: their kernels' verdict arrives here, through whichever call
         * synchronises next); a batch call's timeout with no stream work in flight leaves the streams' carried state alone (ADVICE r5:
         * this function used to poison them for any call) */
        if (c->nstreams > 0 && c->stream_work_unchecked) {
            c->streams_poisoned = true;
            c->carrier_shared = false;
            c->carrier_pending = nullptr;
        }
        c->stream_work_unchecked = false;
        return fail(QPSK_ERR_HIP, "pipeline kernel: producer/consumer wait timed out; results of the calls since the last synchronisation are invalid");
    }
    if (st == STATUS_PHASE_RANGE)
        return fail(QPSK_ERR_RANGE, "Costas loop phase beyond the bounded 2 pi wrap (input amplitude far outside the modem's range; "
                                    "the reference's phase_wrap() would spin or hang, costas_loop.c:61-67); results of the calls since the last synchronisation are invalid");
    if (st == STATUS_NONFINITE)
        return fail(QPSK_ERR_RANGE, "a Costas loop ended on a NaN / Inf state: the input held a non-finite sample (the reference hangs in phase_wrap() "
                                    "on an infinite phase, costas_loop.c:61-67); results of the calls since the last synchronisation are invalid");
    if (st != 0) return fail(QPSK_ERR_HIP, "kernel status %d", st);
    return QPSK_OK;
}

/* test hook (tests/test_gpu_parity.py, the stream error paths): stores a kernel status code in the context's status word as a kernel
 * that gave up would; the next synchronising call of the context reports it */
int qpsk_test_inject_status(qpsk_ctx *c, int code)
{
    if (!c || !c->h_status) return fail(QPSK_ERR_ARG, "null context");
    __atomic_store_n(c->h_status, code, __ATOMIC_RELEASE);
    return QPSK_OK;
}

/* test hook: the one-pass histogram route's books after a synchronisation -- out[0] = the guess the next call will take (-1: none), out[1] = frames
 * the last one-pass call's guess missed, out[2..4] = the statistics the host reads (majority, frames, missed) */
int qpsk_test_hist_state(qpsk_ctx *c, int32_t *out)
{
    if (!c || !out) return fail(QPSK_ERR_ARG, "null argument");
    out[0] = out[1] = out[2] = out[3] = out[4] = -1;
    if (!c->d_hint) return QPSK_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int32_t h[2];
    HIP_TRY(hipMemcpy(h, c->d_hint, sizeof h, hipMemcpyDeviceToHost));
    out[0] = c->hint_valid ? h[0] : -1;
    out[1] = h[1];
    out[2] = c->h_hist_stats[0]; out[3] = c->h_hist_stats[1]; out[4] = c->h_hist_stats[2];
    return QPSK_OK;
}

static int bind(const qpsk_ctx *c)
{
    HIP_TRY(hipSetDevice(c->device));
    return QPSK_OK;
}

/* make d_gains[0] the context's own (alpha, beta) again after a bandwidth sweep replaced it */
static int use_context_gains(qpsk_ctx *c)
{
    const float g[2] = {c->alpha, c->beta};
    if (c->h_gains.size() != 2 || c->h_gains[0] != g[0] || c->h_gains[1] != g[1]) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->h_gains.assign(g, g + 2);
        HIP_TRY(hipMemcpyAsync(c->d_gains, c->h_gains.data(), sizeof g, hipMemcpyHostToDevice, c->stream));
    }
    return QPSK_OK;
}

extern "C" {

const char *qpsk_last_error(void) { return g_err; }
const char *qpsk_version(void) { return "qpsk_hip 0.3 (gfx950)"; }

int qpsk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void qpsk_params_default(qpsk_params *p)
{
    if (!p) return;
    p->fs = 9600.0;
    p->rs = 2400.0;
    p->frame_size = 512;
    p->rrc_alpha = .35f;
    p->loop_bw = (float)(2.0 * 3.14159265358979323846 / 100.0f); /* (TAU / 100.0f), qpsk.c:302 */
    p->min_freq = -1.0f;
    p->max_freq = 1.0f;
    p->timing_mode = QPSK_TIMING_HIST;
    p->fixed_index = 0;
}

static int upload_config(qpsk_ctx *c)
{
    float t128[128];
    memset(t128, 0, sizeof t128);
    memcpy(t128, c->taps, sizeof c->taps);
    c->taps_symmetric = true;
    for (int k = 0; k < QPSK_NTAPS / 2; k++)
        if (memcmp(&c->taps[k], &c->taps[QPSK_NTAPS - 1 - k], sizeof(float)) != 0) c->taps_symmetric = false;
    HIP_TRY(hipMemcpyAsync(c->d_taps, t128, sizeof t128, hipMemcpyHostToDevice, c->stream));
    const float g[2] = {c->alpha, c->beta};
    c->h_gains.assign(g, g + 2);
    HIP_TRY(hipMemcpyAsync(c->d_gains, c->h_gains.data(), sizeof g, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QPSK_OK;
}

int qpsk_ctx_create(qpsk_ctx **out, int device, const qpsk_params *p, void *stream)
{
    if (!out || !p) return fail(QPSK_ERR_ARG, "qpsk_ctx_create: null argument");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(QPSK_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return fail(QPSK_ERR_NO_DEVICE, "device %d out of range (%d devices)", device, ndev);
    if (!(p->fs > 0.0) || !(p->rs > 0.0) || p->frame_size <= 0)
        return fail(QPSK_ERR_ARG, "fs, rs and frame_size must be positive");
    const int cycles = (int)(p->fs / p->rs); /* qpsk.h:21 */
    if (cycles < 1 || cycles > 64) return fail(QPSK_ERR_ARG, "CYCLES = (int)(fs/rs) = %d out of range 1..64", cycles);
    if (p->frame_size % cycles != 0)
        return fail(QPSK_ERR_ARG, "frame_size %d is not a multiple of CYCLES %d", p->frame_size, cycles);
    if (p->timing_mode < QPSK_TIMING_HIST || p->timing_mode > QPSK_TIMING_FFT)
        return fail(QPSK_ERR_ARG, "unknown timing_mode %d", p->timing_mode);
    if (p->fixed_index < 0 || p->fixed_index > MAX_INDEX)
        return fail(QPSK_ERR_ARG, "fixed_index %d outside 0..%d", p->fixed_index, MAX_INDEX);

    HIP_TRY(hipSetDevice(device));
    KERNEL_TRY(prepare_kernels());
    KERNEL_TRY(prepare_pipe_kernel());
    KERNEL_TRY(prepare_timing_scan());
    KERNEL_TRY(prepare_stream_block());
    KERNEL_TRY(prepare_stream_scan());
    qpsk_ctx *c = new qpsk_ctx();
    c->device = device;
    for (const auto &k : TUNING_KEYS) {   /* the only place the environment is read */
        const char *v = getenv(k.name);
        if (v && *v) c->tune.*(k.field) = atoi(v);
    }
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0)
            c->ncu = ncu;
    }
    c->prm = *p;
    c->cycles = cycles;
    c->nsym = p->frame_size / cycles;
    c->stream = (hipStream_t)stream; /* NULL = the HIP default stream, like every HIP library */
    qpsk_host_rrc_taps((float)p->fs, (float)p->rs, p->rrc_alpha, c->taps); /* rrc_make(FS, RS, alpha), qpsk.c:308 */
    c->damping = sqrtf(2.0f) / 2.0f;                                        /* costas_loop.c:38 */
    qpsk_host_loop_gains(c->damping, p->loop_bw, &c->alpha, &c->beta);
    c->min_freq = p->min_freq;
    c->max_freq = p->max_freq;
    if (hipMalloc((void **)&c->d_taps, 128 * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&c->d_gains, MAX_BW * 2 * sizeof(float)) != hipSuccess ||
        hipHostMalloc((void **)&c->h_status, sizeof(int), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->d_status, c->h_status, 0) != hipSuccess) {
        qpsk_ctx_destroy(c);
        return fail(QPSK_ERR_ALLOC, "hipMalloc of configuration buffers failed");
    }
    *c->h_status = 0;
    if (hipHostMalloc((void **)&c->h_done, sizeof(unsigned), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->d_done, c->h_done, 0) != hipSuccess) {
        qpsk_ctx_destroy(c);
        return fail(QPSK_ERR_ALLOC, "pinned counter");
    }
    *c->h_done = 0;
    int rc = upload_config(c);
    if (rc != QPSK_OK) { qpsk_ctx_destroy(c); return rc; }
    *out = c;
    return QPSK_OK;
}

static void free_streams(qpsk_ctx *c)
{
    if (c->h_stage) hipHostFree(c->h_stage);
    hipFree(c->d_stage);
    c->h_stage = c->d_stage = nullptr;
    c->stage_cap = 0;
    hipFree(c->s_memory); hipFree(c->s_dec); hipFree(c->s_loop); hipFree(c->s_mixer); hipFree(c->s_ctab); hipFree(c->s_cstate);
    c->s_memory = c->s_dec = c->s_loop = c->s_mixer = c->s_ctab = c->s_cstate = nullptr;
    c->carrier_shared = false;
    c->carrier_pending = nullptr;       /* points into s_ctab */
    c->nstreams = 0;
}

static void free_transmitters(qpsk_ctx *c)
{
    hipFree(c->t_hist); hipFree(c->t_mixer); hipFree(c->tx_b.p);
    c->t_hist = nullptr;
    c->t_mixer = nullptr;
    c->tx_b = DevBuf();
    c->ntx = 0;
}

void qpsk_ctx_destroy(qpsk_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    hipFree(c->d_taps);
    hipFree(c->d_gains);
    if (c->h_status) hipHostFree(c->h_status);
    if (c->h_done) hipHostFree(c->h_done);
    hipFree(c->index.p);
    hipFree(c->filtered.p);
    hipFree(c->mixed.p);
    hipFree(c->keystream.p);
    hipFree(c->sympad.p);
    hipFree(c->mislist.p);
    if (c->d_hint) hipFree(c->d_hint);
    if (c->h_hist_stats) hipHostFree(c->h_hist_stats);
    for (auto &kv : c->twiddles) hipFree(kv.second);
    hipFree(c->d_fast);
    free_streams(c);
    free_transmitters(c);
    delete c;
}

int qpsk_ctx_sync(qpsk_ctx *c)
{
    if (!c) return fail(QPSK_ERR_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return check_status(c);
}

/* four 2-bit symbols per byte (bitstages.hip): what a host that gathers every step copies back instead of a byte per symbol */
int qpsk_pack_symbols(qpsk_ctx *c, const uint8_t *d_sym, long long nrows, int nsym, uint8_t *d_packed)
{
    if (!c || !d_sym || !d_packed || nrows < 0 || nsym <= 0) return fail(QPSK_ERR_ARG, "qpsk_pack_symbols: bad argument");
    if (bind(c)) return QPSK_ERR_HIP;
    if ((nsym & 15) == 0 && (((uintptr_t)d_sym & 15) || ((uintptr_t)d_packed & 3)))
        return fail(QPSK_ERR_ARG, "qpsk_pack_symbols: d_sym must be 16-byte and d_packed 4-byte aligned");
    KERNEL_TRY(launch_pack_dibits(d_sym, d_packed, (size_t)nrows, nsym, c->stream));
    return QPSK_OK;
}

/* the host side of it: plain C, a format conversion for callers that want a byte per symbol again */
int qpsk_unpack_symbols_host(const uint8_t *h_packed, long long nrows, int nsym, uint8_t *h_sym)
{
    if (!h_packed || !h_sym || nrows < 0 || nsym <= 0) return fail(QPSK_ERR_ARG, "qpsk_unpack_symbols_host: bad argument");
    const size_t pb = (size_t)(nsym + 3) / 4;
    for (long long r = 0; r < nrows; r++) {
        const uint8_t *p = h_packed + (size_t)r * pb;
        uint8_t *o = h_sym + (size_t)r * (size_t)nsym;
        for (int i = 0; i < nsym; i++) o[i] = (uint8_t)((p[i >> 2] >> (2 * (i & 3))) & 3u);
    }
    return QPSK_OK;
}

/* the status word WITHOUT a synchronisation: what the kernels completed so far have flagged (multi.cpp looks at it behind a result
 * slot's copy event, while the next step may already be running on the compute stream) */
int qpsk_ctx_check(qpsk_ctx *c)
{
    if (!c) return fail(QPSK_ERR_ARG, "null context");
    return check_status(c);
}

/* multi.cpp's shard threads report through the CALLING thread's error text */
int qpsk_set_error(int code, const char *msg)
{
    return fail(code, "%s", msg ? msg : "");
}

int qpsk_ctx_set_stream(qpsk_ctx *c, void *stream)
{
    if (!c) return fail(QPSK_ERR_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = (hipStream_t)stream;
    return QPSK_OK;
}

int qpsk_ctx_set_tuning(qpsk_ctx *c, const char *name, int value)
{
    if (!c || !name) return fail(QPSK_ERR_ARG, "null argument");
    for (const auto &k : TUNING_KEYS)
        if (!strcmp(k.name, name)) {
            c->tune.*(k.field) = value < 0 ? -1 : value;
            return QPSK_OK;
        }
    return fail(QPSK_ERR_ARG, "qpsk_ctx_set_tuning: unknown name '%s'", name);
}

int qpsk_ctx_cycles(const qpsk_ctx *c) { return c ? c->cycles : 0; }
int qpsk_ctx_nsym(const qpsk_ctx *c) { return c ? c->nsym : 0; }
const char *qpsk_ctx_last_kernel(const qpsk_ctx *c) { return c ? c->last_kernel : ""; }

int qpsk_ctx_get_taps(const qpsk_ctx *c, float *h)
{
    if (!c || !h) return fail(QPSK_ERR_ARG, "null argument");
    memcpy(h, c->taps, sizeof c->taps);
    return QPSK_OK;
}

int qpsk_ctx_get_gains(const qpsk_ctx *c, float *a, float *b)
{
    if (!c || !a || !b) return fail(QPSK_ERR_ARG, "null argument");
    *a = c->alpha;
    *b = c->beta;
    return QPSK_OK;
}

int qpsk_ctx_set_taps(qpsk_ctx *c, const float *h)
{
    if (!c || !h) return fail(QPSK_ERR_ARG, "null argument");
    if (bind(c)) return QPSK_ERR_HIP;
    memcpy(c->taps, h, sizeof c->taps);
    c->fast_valid = false;
    return upload_config(c);
}

int qpsk_ctx_set_loop(qpsk_ctx *c, float alpha, float beta, float min_freq, float max_freq)
{
    if (!c) return fail(QPSK_ERR_ARG, "null argument");
    if (bind(c)) return QPSK_ERR_HIP;
    /* the limits are kernel arguments; the gains live in device memory and go up only when they change (the
     * drop-in rx_frame() calls this before every block) */
    c->min_freq = min_freq; c->max_freq = max_freq;
    if (alpha == c->alpha && beta == c->beta && !(alpha == 0.0f && beta == 0.0f)) return QPSK_OK;
    c->alpha = alpha; c->beta = beta;
    return upload_config(c);
}

/* ---------------------------------------------------------------- tiling */
static int tuned(int v, int dflt) { return v >= 0 ? v : dflt; }

/* full-rate rrc_fir() of a batch of delay lines (rrc_fir.c:17-30): the generated stream with the taps in SGPRs when the
 * filter is symmetric (firstream.hip), the compiler-scheduled kernel for any other tap set (kernels.hip) */
static int fir_full_rate(qpsk_ctx *c, const float *x, const float *memory, float *y, int nframes, int length, size_t in_pitch = 0)
{
    if (c->taps_symmetric && tuned(c->tune.fir_generic, 0) == 0)
        return launch_rrc_fir_stream(x, memory, y, c->d_taps, nframes, length, c->stream, in_pitch, c->ncu);
    return launch_rrc_fir(x, memory, y, c->d_taps, nframes, length, c->stream, in_pitch);
}

/* frames per workgroup G and symbols per chunk S of rx_fused_kernel */
static void pick_tiling(const qpsk_ctx *c, int nframes, int nbw, int *G, int *S)
{
    int g = nframes / 256;                 /* one workgroup per CU when the batch allows it */
    if (g < 1) g = 1;
    if (g > 64 / nbw) g = 64 / nbw;
    if (g < 1) g = 1;
    g = tuned(c->tune.fused_g, g);
    if (g < 1) g = 1;
    if (g * nbw > 64) g = 64 / nbw;
    int s = tuned(c->tune.fused_s, 64);
    s &= ~7;
    if (s < 8) s = 8;
    const size_t budget = (size_t)tuned(c->tune.fused_lds, 64 * 1024);
    while (s > 8 && fused_lds_bytes(g, s, c->cycles, nbw) > budget) s -= 8;
    while (g > 1 && fused_lds_bytes(g, s, c->cycles, nbw) > (size_t)MAX_LDS_BYTES) g--;
    *G = g;
    *S = s;
}

/* device copy of the size-n twiddle table (cos, sin)(TAU j/n), built once per n with the host's libm (fft.c:55-56) */
static int get_twiddles(qpsk_ctx *c, int n, double **out)
{
    auto it = c->twiddles.find(n);
    if (it != c->twiddles.end()) { *out = it->second; return QPSK_OK; }
    const size_t cnt = n >= 2 ? (size_t)n / 2 : 1;
    std::vector<double> h(2 * cnt, 0.0);
    qpsk_host_twiddles(n, h.data());
    double *tw = nullptr;
    HIP_TRY(hipMalloc((void **)&tw, sizeof(double) * 2 * cnt));
    HIP_TRY(hipMemcpy(tw, h.data(), sizeof(double) * 2 * cnt, hipMemcpyHostToDevice));
    c->twiddles[n] = tw;
    *out = tw;
    return QPSK_OK;
}

/* the FFT timing estimate's host-built tables: size-512 twiddles and the CYCLES candidate phases (libm, like fft.c:55-56) */
static int fft_timing_tables(qpsk_ctx *c, double **tw_out, double **cs_out)
{
    const int C = c->cycles, nfft = timing_fft_nfft();
    if (C < 2 || C > 8 || (C & (C - 1)))
        return fail(QPSK_ERR_ARG, "QPSK_TIMING_FFT needs CYCLES in {2,4,8} (the symbol-rate bin NFFT/CYCLES must be exact and the offset < 8); CYCLES = %d", C);
    if (c->prm.frame_size < timing_fft_first() + nfft)
        return fail(QPSK_ERR_ARG, "QPSK_TIMING_FFT needs frame_size >= %d", timing_fft_first() + nfft);
    double *tw = nullptr;
    int rc = get_twiddles(c, nfft, &tw);
    if (rc) return rc;
    /* candidate phases (cos, sin)(TAU i/CYCLES): entry -C of the twiddle cache (key < 0 cannot collide with an FFT size) */
    double *cs = nullptr;
    auto it = c->twiddles.find(-C);
    if (it == c->twiddles.end()) {
        std::vector<double> full(2 * (size_t)C);
        qpsk_host_phases(C, full.data());
        HIP_TRY(hipMalloc((void **)&cs, sizeof(double) * 2 * C));
        HIP_TRY(hipMemcpy(cs, full.data(), sizeof(double) * 2 * C, hipMemcpyHostToDevice));
        c->twiddles[-C] = cs;
    } else {
        cs = it->second;
    }
    *tw_out = tw;
    *cs_out = cs;
    return QPSK_OK;
}

static int fft_timing_indices(qpsk_ctx *c, const float *d_in, int nframes, int32_t *d_index, float *d_y = nullptr,
                              double *d_X = nullptr, size_t pitch = 0, double *d_bin = nullptr)
{
    double *tw = nullptr, *cs = nullptr;
    int rc = fft_timing_tables(c, &tw, &cs);
    if (rc) return rc;
    KERNEL_TRY(launch_timing_fft(d_in, nframes, c->prm.frame_size, c->cycles, c->d_taps, tw, cs, d_index, d_y, d_X, d_bin, c->stream, pitch,
                                 c->ncu, c->taps_symmetric && tuned(c->tune.fir_generic, 0) == 0));
    return QPSK_OK;
}

/* QPSK_HIST_GENERIC = 1 or 2 keeps the three-kernel path (rrc_fir -> scan -> pipeline), 2 with the any-CYCLES scan too */
static bool scan_fused_ok(const qpsk_ctx *c, const float *d_in)
{
    return c->cycles == 8 && c->prm.frame_size % timing_scan_tile() == 0 && ((uintptr_t)d_in % 16) == 0 &&
           tuned(c->tune.hist_generic, 0) == 0;
}

/* fused_fft: the caller will run the FFT estimate inside the receive launch (rx_fused_pipe_kernel, full workgroups): no
 * launch here, *d_index_out stays NULL; c->index is sized for the indices the kernel leaves */
static int timing_indices(qpsk_ctx *c, const float *d_in, size_t pitch, int nframes, const int32_t **d_index_out, bool fused_fft = false)
{
    *d_index_out = nullptr;
    if (c->prm.timing_mode == QPSK_TIMING_FIXED) return QPSK_OK;
    int rc = ensure(c, c->index, sizeof(int32_t) * (size_t)nframes);
    if (rc) return rc;
    if (fused_fft && c->prm.timing_mode == QPSK_TIMING_FFT) return QPSK_OK;
    if (c->prm.timing_mode == QPSK_TIMING_HIST) {
        /* the fused full-rate FIR + scan kernel keeps the filtered block in LDS (timing_scan.hip): the input is read
         * once here and once by the pipeline kernel that follows, nothing is written but the index */
        if (scan_fused_ok(c, d_in) && (pitch & 1) == 0) {
            KERNEL_TRY(launch_timing_scan(d_in, nframes, c->prm.frame_size, c->d_taps, (int32_t *)c->index.p, nullptr,
                                          c->d_status, c->stream, pitch, c->taps_symmetric && tuned(c->tune.fir_generic, 0) == 0));
            *d_index_out = (const int32_t *)c->index.p;
            return QPSK_OK;
        }
        const size_t bytes = sizeof(float) * 2 * (size_t)nframes * c->prm.frame_size;
        rc = ensure(c, c->filtered, bytes);
        if (rc) return rc;
        KERNEL_TRY(fir_full_rate(c, d_in, nullptr, (float *)c->filtered.p, nframes, c->prm.frame_size, pitch));
        KERNEL_TRY(launch_timing_hist((const float *)c->filtered.p, nframes, c->prm.frame_size, c->cycles,
                                      (int32_t *)c->index.p, nullptr, tuned(c->tune.hist_generic, 0) == 2, c->stream));
        *d_index_out = (const int32_t *)c->index.p;
        return QPSK_OK;
    }
    /* QPSK_TIMING_FFT: symbol-rate line of |y|^2 through the reference's radix-2 FFT (timing_fft.hip) */
    rc = fft_timing_indices(c, d_in, nframes, (int32_t *)c->index.p, nullptr, nullptr, pitch);
    if (rc) return rc;
    *d_index_out = (const int32_t *)c->index.p;
    return QPSK_OK;
}

static int rx_batch_common(qpsk_ctx *c, const float *d_in, long long frame_pitch, int nframes, int nbw, uint8_t *d_sym,
                           float *d_freq, float *d_phase, float *d_costas, int32_t *d_index, float *d_hz)
{
    if (!c || !d_in || !d_sym) return fail(QPSK_ERR_ARG, "qpsk_rx_batch: null context, input or symbol buffer");
    if (nframes <= 0) return fail(QPSK_ERR_ARG, "qpsk_rx_batch: nframes = %d", nframes);
    if (frame_pitch == 0) frame_pitch = c->prm.frame_size;
    if (frame_pitch < c->prm.frame_size)
        return fail(QPSK_ERR_ARG, "qpsk_rx_batch_pitched: frame_pitch %lld below frame_size %d", frame_pitch, c->prm.frame_size);
    if (bind(c)) return QPSK_ERR_HIP;
    /* the pipeline kernel (rx_fused.hip) is built for CYCLES = 8, 16-byte aligned frames and an index
     * below CYCLES; everything else takes the generic chunked kernel (kernels.hip) */
    const bool pipe_ok = c->cycles == pipe_cycles() && (c->prm.frame_size % 2) == 0 && (frame_pitch % 2) == 0 &&
                         ((uintptr_t)d_in % 16) == 0 && nbw * pipe_frames(1) <= 64 &&
                         tuned(c->tune.generic, 0) == 0;
    FusedArgs a{};
    a.x = reinterpret_cast<const float2 *>(d_in);
    a.frame_pitch = (size_t)frame_pitch;
    a.nframes = nframes;
    a.frame_size = c->prm.frame_size;
    a.cycles = c->cycles;
    a.nsym = c->nsym;
    pick_tiling(c, nframes, nbw, &a.G, &a.S);
    a.fixed_index = c->prm.fixed_index;
    a.dbg = tuned(c->tune.pipe_variant, 0) & PIPE_VARIANT_MASK;
    a.lean_dma = tuned(c->tune.lean_dma, 1) != 0;
    a.lean_twowin = tuned(c->tune.lean_dma, 1) != 2 ? 1 : 0;      /* QPSK_LEAN_DMA = 2: LDS-DMA, but one window per FIR wave whatever the LDS allows */
    a.est_waves = tuned(c->tune.est_waves, 0);
    a.lean_pair = tuned(c->tune.lean_pair, 3);      /* 3: the library's own rule (launch_rx_lean) */
    a.taps = c->d_taps;
    a.gains = c->d_gains;
    a.nbw = nbw;
    a.min_freq = c->min_freq;
    a.max_freq = c->max_freq;
    a.rs = c->prm.rs;
    a.sym = d_sym;
    a.freq = d_freq;
    a.phase = d_phase;
    a.costas = reinterpret_cast<float2 *>(d_costas);
    a.hz = d_hz;
    a.status = c->d_status;

    /* ---- which kernel takes the batch: decided ONCE, here, as a plan; the timing estimate's placement (below) reads the plan and
     * the launches execute it -- nothing restates the choice (round 4 restated it for the in-launch FFT estimate and missed a
     * tuning key: QPSK_PIPE_G above 16 sent the batch to rx_lean_kernel with no index computed).
     * The pipeline kernels [measured, DESIGN.md 4.1]: rx_lean_kernel wherever its stream serves the shape in one launch (see below);
     * rx_fused_pipe_kernel (16-frame workgroups, four-symbol FIR lanes, the FFT timing estimate inside the launch) for ragged batches
     * up to 16 frames per CU, several loops per frame, a costas_frame[] dump and config 3; rx_pipe2_kernel for those above 16. */
    enum { K_GENERIC, K_FUSED_PIPE, K_PIPE2, K_LEAN };
    struct Plan {
        int kind = K_GENERIC, nframes = 0, G = 0, nf = 0;
        unsigned long long layout = 0;
    };
    const int pipe_v = tuned(c->tune.pipe_v, nframes > 16 * c->ncu ? 2 : 1);
    /* the pipeline kernels of rounds 1-2 (and the generic chunked kernel): any shape */
    auto plan_general = [&](int nfr, int pv, Plan *pl) -> int {
        pl->nframes = nfr;
        if (pipe_ok && pv == 2) {
            /* rx_pipe2_kernel: up to 32 frames per workgroup, one workgroup per CU when the batch allows it */
            int G = (nfr + c->ncu - 1) / c->ncu;
            if (G > pipe2_max_frames()) G = pipe2_max_frames();
            G = tuned(c->tune.pipe_g, G);
            if (G > pipe2_max_frames()) G = pipe2_max_frames();
            if (G * nbw > 64) G = 64 / nbw;
            if (G < 1) G = 1;
            /* wave layout: the library's (pipe2_default_layout), or the caller's for measurements: 4 bits per hardware wave
             * = units it owns, waves 1-5 in QPSK_PIPE_LAYOUT_LO, 6-11 in QPSK_PIPE_LAYOUT_HI (their sum fixes G's units) */
            unsigned long long layout = 0;
            if (c->tune.layout_lo >= 0 || c->tune.layout_hi >= 0) {
                layout = ((unsigned long long)(unsigned)tuned(c->tune.layout_lo, 0) << 4) |
                         ((unsigned long long)(unsigned)tuned(c->tune.layout_hi, 0) << 24);
                int units = 0, nwin = 0;
                for (int w = 1; w < 16; w++) { const int cw = (int)((layout >> (4 * w)) & 15); units += cw; nwin += cw != 0; }
                if (units < 1 || (G + 1) / 2 != units || pipe2_lds_bytes(G, nwin, nbw) > (size_t)MAX_LDS_BYTES)
                    return fail(QPSK_ERR_ARG, "QPSK_PIPE_LAYOUT_*: %d units on %d waves do not match %d frames per workgroup", units, nwin, G);
            } else {
                for (;;) {   /* many loops per frame: the record rings grow, fewer frames fit */
                    layout = pipe2_default_layout((G + 1) / 2);
                    int nwin = 0;
                    for (int w = 1; w < 16; w++) nwin += ((layout >> (4 * w)) & 15) != 0;
                    if (layout && pipe2_lds_bytes(G, nwin, nbw) <= (size_t)MAX_LDS_BYTES) break;
                    if (G == 1) return fail(QPSK_ERR_ARG, "pipeline geometry does not fit: %d loops per frame", nbw);
                    G--;
                }
            }
            pl->kind = K_PIPE2;
            pl->G = G;
            pl->layout = layout;
        } else if (pipe_ok) {
            /* 16-frame workgroups (four FIR waves of four frames, fewer when the batch gives a CU fewer frames); a batch
             * above 16 frames per CU would run them in rounds */
            auto fits = [&](int nf_) {   /* one lane of the serial wave per (frame, loop); rings grow with the loops */
                return pipe_frames(nf_) * nbw <= 64 && pipe_lds_bytes(nf_, nbw) <= (size_t)MAX_LDS_BYTES;
            };
            const int full = pipe_max_nf();
            int nf = 1;
            while (nf < full && (long long)c->ncu * pipe_frames(nf) < nfr) nf++;
            nf = tuned(c->tune.pipe_nf, nf);
            if (nf < 1) nf = 1;
            if (nf > full) nf = full;
            while (nf > 1 && !fits(nf)) nf--;
            if (!fits(nf))
                return fail(QPSK_ERR_ARG, "pipeline geometry does not fit: nf %d, %d loops per frame", nf, nbw);
            pl->kind = K_FUSED_PIPE;
            pl->nf = nf;
        } else {
            pl->kind = K_GENERIC;
        }
        return QPSK_OK;
    };
    /* rx_lean_kernel (the FIR waves' chunk loop as one hand-written stream) serves whole even workgroups of frames made of
     * whole chunks, one loop per frame, symmetric filter, no costas_frame[] dump; a batch's last partial workgroup and every
     * other shape go to the kernels above.  QPSK_PIPE_V = 3 asks for it at any batch size, 1 / 2 for the older kernels. */
    Plan main_pl, rem_pl;
    bool lean = false;
    if (pipe_ok && c->taps_symmetric && (pipe_v == 3 || c->tune.pipe_v < 0)) {
        int G = (nframes + c->ncu - 1) / c->ncu;
        G = tuned(c->tune.pipe_g, G);
        if (G > pipe2_max_frames()) G = pipe2_max_frames();
        G += G & 1;
        /* above 16 frames per CU: always (the filter is the limit there).  Up to 16 frames per CU both kernels sit on the serial wave,
         * and since round 5 (LDS-DMA staging, a window per unit, two-unit FIR waves) this one is ahead there too -- 0.1519 against 0.1560 ms
         * at config 2, 1-2 % at 1024-3584 frames (profiles/r05_config2_lean.txt) -- for batches it takes in ONE launch (whole workgroups
         * of at least four frames). */
        /* Round 6: a batch that is not whole workgroups rides in the same launch -- the last workgroup's pad frames read the batch's last
         * frame and leave their symbols in a pad buffer (round 5 sent the remainder to a SECOND launch: one more 2048-step serial chain,
         * 0.25 + 0.15 ms for 8193 frames against 0.25 for 8192). */
        const bool wanted = pipe_v == 3 || G >= 4;
        unsigned long long layout = 0;
        if (c->tune.layout_lo >= 0 || c->tune.layout_hi >= 0)
            layout = ((unsigned long long)(unsigned)tuned(c->tune.layout_lo, 0) << 4) |
                     ((unsigned long long)(unsigned)tuned(c->tune.layout_hi, 0) << 24);
        else if (G >= 2)
            layout = lean_default_layout(G / 2);
        FusedArgs am = a;
        if (G >= 2 && nframes % G) {
            int rp = ensure(c, c->sympad, (size_t)G * (size_t)c->nsym);
            if (rp) return rp;
            a.sym_pad = am.sym_pad = (uint8_t *)c->sympad.p;
        }
        if (wanted && layout && lean_shape_ok(am, G)) {
            int nwin = 0, units = 0;
            for (int w = 1; w < 16; w++) { const int cw = (int)((layout >> (4 * w)) & 15); units += cw; nwin += cw != 0; }
            if (units == G / 2 && lean_lds_bytes(G, nwin) <= (size_t)MAX_LDS_BYTES) {
                lean = true;
                main_pl.kind = K_LEAN;
                main_pl.nframes = am.nframes;
                main_pl.G = G;
                main_pl.layout = layout;
            }
        }
    }
    if (!lean) {
        int rp = plan_general(nframes, pipe_v == 3 ? (nframes > 16 * c->ncu ? 2 : 1) : pipe_v, &main_pl);
        if (rp) return rp;
    }

    /* ---- histogram timing in ONE pass (round 6; rx_fused.hip, rx_hist_kernel): where the context has a guess -- the majority index of
     * its previous histogram-mode batch -- the scan kernel's workgroup runs the receive path on that guess while it scans, the frames
     * whose true index differs are redone by a fall-back pass over their list (usually empty), and the batch's own majority becomes
     * the next guess.  No guess yet, a shape the kernel does not serve, or a last batch that missed more than an eighth of its frames
     * (a batch of mixed indices: the fall-back pass would carry it): the two-launch route below. */
    const bool hist_mode = c->prm.timing_mode == QPSK_TIMING_HIST;
    if (hist_mode && pipe_ok && c->taps_symmetric && scan_fused_ok(c, d_in) && c->tune.pipe_v < 0) {      /* (QPSK_PIPE_V asks for one of the receive kernels by name) */
        if (!c->d_hint) {
            HIP_TRY(hipMalloc((void **)&c->d_hint, 2 * sizeof(int32_t)));
            HIP_TRY(hipMemset(c->d_hint, 0, 2 * sizeof(int32_t)));
            HIP_TRY(hipHostMalloc((void **)&c->h_hist_stats, 4 * sizeof(int32_t), hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void **)&c->d_hist_stats, c->h_hist_stats, 0));
            c->h_hist_stats[0] = c->h_hist_stats[1] = c->h_hist_stats[2] = -1;
            c->hint_valid = false;
        }
        FusedArgs ah = a;
        const int op = tuned(c->tune.hist_onepass, -1);
        /* the route pays only while the guess holds for EVERY frame: a frame it misses goes through the fall-back pass, whose one
         * workgroup takes 0.4 ms whatever the count (a serial chain again) against the 0.07 ms the route saves.  Every histogram-mode call
         * leaves behind how many frames of its batch were off the batch's majority index (or missed by its guess): zero = try the route */
        const int seen = __atomic_load_n(&c->h_hist_stats[1], __ATOMIC_ACQUIRE), off_majority = __atomic_load_n(&c->h_hist_stats[2], __ATOMIC_ACQUIRE);
        const bool guess_is_good = seen > 0 && off_majority == 0;
        if (op != 0 && c->hint_valid && (op == 1 || guess_is_good) && rx_hist_shape_ok(ah)) {
            int r1 = ensure(c, c->index, sizeof(int32_t) * (size_t)nframes);
            if (r1) return r1;
            r1 = ensure(c, c->mislist, sizeof(int32_t) * (size_t)nframes);
            if (r1) return r1;
            int32_t *mis_count = c->d_hint + 1;      /* zero: allocation, then every index_majority_kernel */
            KERNEL_TRY(launch_rx_hist(ah, (int32_t *)c->index.p, c->d_hint, (int32_t *)c->mislist.p, mis_count, c->d_status, c->stream));
            /* the fall-back pass: the generic chunked kernel over the listed frames with their true indices (its grid is sized for the
             * whole batch: the list's length is known on the device only; workgroups beyond it retire at once) */
            FusedArgs af = a;
            af.index = (const int32_t *)c->index.p;
            af.frame_list = (const int32_t *)c->mislist.p;
            af.frame_list_count = mis_count;
            af.G = 4;
            af.S = 64;
            KERNEL_TRY(launch_rx_fused(af, c->stream));
            KERNEL_TRY(launch_index_majority((const int32_t *)c->index.p, nframes, c->d_hint, mis_count, c->d_hist_stats, c->stream));
            c->last_kernel = "rx_hist_kernel (one pass on the guessed index) + rx_fused_kernel (fall-back list)";
            if (d_index)
                HIP_TRY(hipMemcpyAsync(d_index, c->index.p, sizeof(int32_t) * (size_t)nframes, hipMemcpyDeviceToDevice, c->stream));
            return QPSK_OK;
        }
    }

    /* ---- the timing estimate.  BASELINE config 3's shape -- FFT estimate, ONE launch of rx_fused_pipe_kernel in full 16-frame
     * workgroups, as the plan says -- runs the estimate inside the receive launch (rx_fused.hip); every other plan gets its indices
     * from a launch in front (timing_fft_kernel / timing_scan_kernel / ...). */
    bool fused_fft = false;
    double *est_tw = nullptr, *est_cs = nullptr;
    if (c->prm.timing_mode == QPSK_TIMING_FFT && rem_pl.nframes == 0 && nbw == 1 && tuned(c->tune.fft_fused, 1) != 0 &&
        c->prm.frame_size >= timing_fft_first() + timing_fft_nfft() &&
        ((main_pl.kind == K_FUSED_PIPE && main_pl.nf == pipe_max_nf() && !(tuned(c->tune.pipe_variant, 0) & (128 | 4))) ||
         (main_pl.kind == K_LEAN && lean_est_ok(a, main_pl.G, main_pl.layout)))) {
        int rt = fft_timing_tables(c, &est_tw, &est_cs);
        if (rt) return rt;
        fused_fft = true;
    }
    const int32_t *idx = nullptr;
    int rc = timing_indices(c, d_in, (size_t)frame_pitch, nframes, &idx, fused_fft);
    if (rc) return rc;
    a.index = idx;
    if (fused_fft) {
        a.est_tw = reinterpret_cast<const double2 *>(est_tw);
        a.est_cs = reinterpret_cast<const double2 *>(est_cs);
        a.index_out = d_index ? (int32_t *)c->index.p : nullptr;
        if (d_index) idx = (const int32_t *)c->index.p;      /* copied to the caller's array behind the launch, below */
    }

    /* ---- the launches */
    auto execute = [&](const FusedArgs &fa, const Plan &pl) -> int {
        /* only rx_fused_pipe_kernel's full workgroups and rx_lean_kernel look at est_tw; any other kernel would demodulate with fixed_index */
        if (fa.est_tw && !(pl.kind == K_FUSED_PIPE && pl.nf == pipe_max_nf()) && pl.kind != K_LEAN)
            return fail(QPSK_ERR_STATE, "internal: in-launch FFT timing estimate planned for a kernel that has none");
        switch (pl.kind) {
        case K_LEAN:
            KERNEL_TRY(launch_rx_lean(fa, pl.G, pl.layout, c->d_status, c->stream));
            c->last_kernel = fa.est_tw ? "rx_lean_kernel (FFT timing estimate inside the launch)" : "rx_lean_kernel";
            break;
        case K_PIPE2:
            KERNEL_TRY(launch_rx_pipe2(fa, pl.G, pl.layout, c->d_status, c->stream));
            c->last_kernel = "rx_pipe2_kernel";
            break;
        case K_FUSED_PIPE:
            KERNEL_TRY(launch_rx_fused_pipe(fa, pl.nf, c->d_status, c->stream));
            c->last_kernel = fa.est_tw ? "rx_fused_pipe_kernel (FFT timing estimate inside the launch)" : "rx_fused_pipe_kernel";
            break;
        default:
            KERNEL_TRY(launch_rx_fused(fa, c->stream));
            c->last_kernel = "rx_fused_kernel";
        }
        return QPSK_OK;
    };
    {
        FusedArgs am = a;
        am.nframes = main_pl.nframes;
        int rc2 = execute(am, main_pl);
        if (rc2) return rc2;
        if (rem_pl.nframes > 0) {
            FusedArgs ar = a;
            const size_t o = (size_t)main_pl.nframes;
            ar.nframes = rem_pl.nframes;
            ar.x += o * a.frame_pitch;
            if (ar.index) ar.index += o;
            ar.sym += o * (size_t)a.nsym;
            if (ar.freq) ar.freq += o;
            if (ar.phase) ar.phase += o;
            if (ar.hz) ar.hz += o;
            const char *lk = c->last_kernel;
            rc2 = execute(ar, rem_pl);
            if (rc2) return rc2;
            c->last_kernel = lk;
        }
    }
    if (hist_mode && c->d_hint && idx && tuned(c->tune.hist_onepass, -1) != 0) {
        /* the batch's majority index as the next histogram-mode call's guess (one workgroup, in stream order) */
        KERNEL_TRY(launch_index_majority(idx, nframes, c->d_hint, c->d_hint + 1, c->d_hist_stats, c->stream));
        c->hint_valid = true;
    }
    if (main_pl.kind == K_LEAN && (nframes & 1)) {
        /* the last frame of an odd batch shares its two-frame unit with a pad frame: the unit's rows are pad rows (rx_fused.hip,
         * lean_unit_rows); its own row goes to its place behind the launch */
        const size_t last0 = (size_t)((nframes - 1) / main_pl.G) * (size_t)main_pl.G;      /* the last workgroup's first frame */
        HIP_TRY(hipMemcpyAsync(d_sym + (size_t)(nframes - 1) * (size_t)c->nsym, (const uint8_t *)c->sympad.p + ((size_t)(nframes - 1) - last0) * (size_t)c->nsym,
                               (size_t)c->nsym, hipMemcpyDeviceToDevice, c->stream));
    }
    if (d_index) {
        if (idx)
            HIP_TRY(hipMemcpyAsync(d_index, idx, sizeof(int32_t) * (size_t)nframes, hipMemcpyDeviceToDevice, c->stream));
        else
            KERNEL_TRY(launch_fill_i32(d_index, nframes, c->prm.fixed_index, c->stream));
    }
    return QPSK_OK;
}

int qpsk_rx_batch(qpsk_ctx *c, const float *d_in, int nframes, uint8_t *d_sym, float *d_freq, float *d_phase,
                  float *d_costas, int32_t *d_index, float *d_hz)
{
    if (c) {
        if (bind(c)) return QPSK_ERR_HIP;
        if (int rg = use_context_gains(c)) return rg;
    }
    return rx_batch_common(c, d_in, 0, nframes, 1, d_sym, d_freq, d_phase, d_costas, d_index, d_hz);
}

int qpsk_rx_batch_pitched(qpsk_ctx *c, const float *d_in, long long frame_pitch, int nframes, uint8_t *d_sym, float *d_freq,
                          float *d_phase, float *d_costas, int32_t *d_index, float *d_hz)
{
    if (c) {
        if (bind(c)) return QPSK_ERR_HIP;
        if (int rg = use_context_gains(c)) return rg;
    }
    if (frame_pitch <= 0) return fail(QPSK_ERR_ARG, "qpsk_rx_batch_pitched: frame_pitch = %lld", frame_pitch);
    return rx_batch_common(c, d_in, frame_pitch, nframes, 1, d_sym, d_freq, d_phase, d_costas, d_index, d_hz);
}

int qpsk_rx_batch_bw(qpsk_ctx *c, const float *d_in, int nframes, const float *h_loop_bw, int nbw, uint8_t *d_sym,
                     float *d_freq, float *d_phase, int32_t *d_index)
{
    if (!c || !h_loop_bw) return fail(QPSK_ERR_ARG, "qpsk_rx_batch_bw: null argument");
    if (nbw < 1 || nbw > MAX_BW) return fail(QPSK_ERR_ARG, "nbw = %d outside 1..%d", nbw, MAX_BW);
    if (bind(c)) return QPSK_ERR_HIP;
    std::vector<float> g(2 * (size_t)nbw);
    for (int b = 0; b < nbw; b++)
        qpsk_host_loop_gains(c->damping, h_loop_bw[b], &g[2 * b], &g[2 * b + 1]); /* set_loop_bandwidth(), costas_loop.c:79-87 */
    if (g != c->h_gains) {
        HIP_TRY(hipStreamSynchronize(c->stream)); /* the previous upload may still read h_gains */
        c->h_gains = g;
        HIP_TRY(hipMemcpyAsync(c->d_gains, c->h_gains.data(), sizeof(float) * g.size(), hipMemcpyHostToDevice, c->stream));
    }
    return rx_batch_common(c, d_in, 0, nframes, nbw, d_sym, d_freq, d_phase, nullptr, d_index, nullptr);
}

/* Costas + slicer over decimated symbols already in device memory (qpsk.c:196-212): the pipeline kernel with
 * loader waves in place of the FIR waves; the one-lane-per-frame kernel only when QPSK_FUSED_GENERIC is set. */
/* Costas + slicer over rows of decimated symbols.  Streaming mode: refill != NULL names the block just filtered
 * ([nframes][nsym*CYCLES]) and refill_index its timing indices; each row of d_symbols is then replaced by that
 * block's picks once the loop has taken the old ones (qpsk.c:186-191). */
static int costas_over_symbols(qpsk_ctx *c, float *d_symbols, int nframes, int nsym, int dstride, float *d_state,
                               uint8_t *d_sym, float *d_costas, const float *refill = nullptr,
                               const int32_t *refill_index = nullptr, bool refill_planar = false)
{
    float *carrier_tab = c->carrier_pending;
    c->carrier_pending = nullptr;
    if (tuned(c->tune.generic, 0)) {
        if (carrier_tab) KERNEL_TRY(launch_carrier_table(c->s_cstate, carrier_tab, c->prm.frame_size, true, c->stream));
        KERNEL_TRY(launch_costas(d_symbols, nframes, nsym, dstride, 1, c->d_gains, c->min_freq, c->max_freq, d_state, d_state,
                                 d_sym, d_costas, c->d_status, c->stream));
        if (refill) {
            if (dstride != nsym) return fail(QPSK_ERR_ARG, "internal: refill needs packed rows");
            KERNEL_TRY(launch_decimate(refill, refill_index, d_symbols, nframes, nsym * c->cycles, c->cycles, nsym, c->stream));
        }
        return QPSK_OK;
    }
    FusedArgs a{};
    a.nframes = nframes;
    a.nsym = nsym;
    a.frame_size = nsym * c->cycles;
    a.cycles = c->cycles;
    a.refill = reinterpret_cast<const float2 *>(refill);
    a.refill_dst = reinterpret_cast<float2 *>(d_symbols);
    a.refill_planar = refill_planar ? 1 : 0;
    a.index = refill_index;
    a.gains = c->d_gains;
    a.nbw = 1;
    a.min_freq = c->min_freq;
    a.max_freq = c->max_freq;
    a.rs = c->prm.rs;
    a.state_in = d_state;
    a.state_out = d_state;
    a.sym = d_sym;
    a.costas = reinterpret_cast<float2 *>(d_costas);
    a.dsrc = reinterpret_cast<const float2 *>(d_symbols);
    a.dstride = dstride;
    a.status = c->d_status;
    if (carrier_tab) {
        a.carrier_state = c->s_cstate;
        a.carrier_tab = reinterpret_cast<float2 *>(carrier_tab);
        a.carrier_frame = c->prm.frame_size;
    }
    a.dbg = tuned(c->tune.pipe_variant, 0) & PIPE_VARIANT_MASK;
    int nf = (nframes + c->ncu * pipe_frames(1) - 1) / (c->ncu * pipe_frames(1));
    if (nf < 1) nf = 1;
    if (nf > pipe_max_nf()) nf = pipe_max_nf();
    KERNEL_TRY(launch_costas_pipe(a, nf, c->d_status, c->stream));
    return QPSK_OK;
}

/* ---------------------------------------------------------------- stages */
int qpsk_rrc_fir_batch(qpsk_ctx *c, float *d_memory, const float *d_in, float *d_out, int nframes, int length)
{
    if (!c || !d_in || !d_out) return fail(QPSK_ERR_ARG, "qpsk_rrc_fir_batch: null argument");
    if (nframes <= 0 || length <= 0) return fail(QPSK_ERR_ARG, "qpsk_rrc_fir_batch: nframes %d length %d", nframes, length);
    if (bind(c)) return QPSK_ERR_HIP;
    const float *src = d_in;
    if (d_in == d_out) {
        /* rrc_fir() works in place (rrc_fir.c:28); a tiled kernel cannot, so filter from a copy */
        const size_t bytes = sizeof(float) * 2 * (size_t)nframes * length;
        int rc = ensure(c, c->mixed, bytes);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(c->mixed.p, d_in, bytes, hipMemcpyDeviceToDevice, c->stream));
        src = (const float *)c->mixed.p;
    }
    KERNEL_TRY(fir_full_rate(c, src, d_memory, d_out, nframes, length));
    if (d_memory) KERNEL_TRY(launch_delay_line(src, d_memory, nframes, length, c->stream));
    return QPSK_OK;
}

int qpsk_rrc_fir_batch_fast(qpsk_ctx *c, float *d_memory, const float *d_in, float *d_out, int nframes, int length)
{
    if (!c || !d_in || !d_out) return fail(QPSK_ERR_ARG, "qpsk_rrc_fir_batch_fast: null argument");
    if (nframes <= 0 || length <= 0) return fail(QPSK_ERR_ARG, "qpsk_rrc_fir_batch_fast: nframes %d length %d", nframes, length);
    if (d_in == d_out) return fail(QPSK_ERR_ARG, "qpsk_rrc_fir_batch_fast: every block reads 126 samples in front of it: separate buffers");
    if (bind(c)) return QPSK_ERR_HIP;
    if (!c->fast_valid) {      /* the filter's spectrum and the twiddles, in double on the host (host_math.c) */
        const int n = rrc_fir_fast_nfft();
        std::vector<float> t(4 * (size_t)n);
        qpsk_host_fir_fast_tables(c->taps, t.data(), t.data() + 2 * n);
        if (!c->d_fast) HIP_TRY(hipMalloc((void **)&c->d_fast, sizeof(float) * t.size()));
        HIP_TRY(hipStreamSynchronize(c->stream));      /* an earlier call may still read the old tables */
        HIP_TRY(hipMemcpy(c->d_fast, t.data(), sizeof(float) * t.size(), hipMemcpyHostToDevice));
        c->fast_valid = true;
    }
    KERNEL_TRY(launch_rrc_fir_fast(d_in, d_memory, d_out, c->d_fast, c->d_fast + 2 * rrc_fir_fast_nfft(), nframes, length, c->stream));
    if (d_memory) KERNEL_TRY(launch_delay_line(d_in, d_memory, nframes, length, c->stream));
    return QPSK_OK;
}

int qpsk_timing_hist_batch(qpsk_ctx *c, const float *d_filtered, int nframes, int32_t *d_index, int32_t *d_hist)
{
    if (!c || !d_filtered || !d_index) return fail(QPSK_ERR_ARG, "qpsk_timing_hist_batch: null argument");
    if (nframes <= 0) return fail(QPSK_ERR_ARG, "nframes = %d", nframes);
    if (bind(c)) return QPSK_ERR_HIP;
    KERNEL_TRY(launch_timing_hist(d_filtered, nframes, c->prm.frame_size, c->cycles, d_index, d_hist,
                                  tuned(c->tune.hist_generic, 0) == 2, c->stream));
    return QPSK_OK;
}

int qpsk_timing_scan_batch(qpsk_ctx *c, const float *d_in, int nframes, int32_t *d_index, int32_t *d_hist)
{
    if (!c || !d_in || !d_index) return fail(QPSK_ERR_ARG, "qpsk_timing_scan_batch: null argument");
    if (nframes <= 0) return fail(QPSK_ERR_ARG, "nframes = %d", nframes);
    if (bind(c)) return QPSK_ERR_HIP;
    if (!(c->cycles == 8 && c->prm.frame_size % timing_scan_tile() == 0 && ((uintptr_t)d_in % 16) == 0))
        return fail(QPSK_ERR_ARG, "qpsk_timing_scan_batch needs CYCLES = 8, frame_size %% %d == 0 and 16-byte aligned input "
                                  "(use qpsk_rrc_fir_batch + qpsk_timing_hist_batch otherwise)", timing_scan_tile());
    KERNEL_TRY(launch_timing_scan(d_in, nframes, c->prm.frame_size, c->d_taps, d_index, d_hist, c->d_status, c->stream, 0,
                                  c->taps_symmetric && tuned(c->tune.fir_generic, 0) == 0));
    return QPSK_OK;
}

int qpsk_timing_fft_batch(qpsk_ctx *c, const float *d_in, int nframes, int32_t *d_index, float *d_filtered, double *d_spectrum)
{
    if (!c || !d_in || !d_index) return fail(QPSK_ERR_ARG, "qpsk_timing_fft_batch: null argument");
    if (nframes <= 0) return fail(QPSK_ERR_ARG, "nframes = %d", nframes);
    if (bind(c)) return QPSK_ERR_HIP;
    return fft_timing_indices(c, d_in, nframes, d_index, d_filtered, d_spectrum);
}

int qpsk_timing_fft_bin_batch(qpsk_ctx *c, const float *d_in, int nframes, int32_t *d_index, float *d_filtered, double *d_bin)
{
    if (!c || !d_in || !d_index) return fail(QPSK_ERR_ARG, "qpsk_timing_fft_bin_batch: null argument");
    if (nframes <= 0) return fail(QPSK_ERR_ARG, "nframes = %d", nframes);
    if (bind(c)) return QPSK_ERR_HIP;
    return fft_timing_indices(c, d_in, nframes, d_index, d_filtered, nullptr, 0, d_bin);
}

int qpsk_costas_batch(qpsk_ctx *c, const float *d_symbols_in, int nframes, int nsym, float *d_state, uint8_t *d_sym,
                      float *d_costas)
{
    if (!c || !d_symbols_in) return fail(QPSK_ERR_ARG, "qpsk_costas_batch: null argument");
    if (nframes <= 0 || nsym <= 0) return fail(QPSK_ERR_ARG, "nframes %d nsym %d", nframes, nsym);
    if (bind(c)) return QPSK_ERR_HIP;
    if (int rg = use_context_gains(c)) return rg;
    /* no refill: the rows are only read */
    return costas_over_symbols(c, const_cast<float *>(d_symbols_in), nframes, nsym, nsym, d_state, d_sym, d_costas);
}

int qpsk_fft_batch(qpsk_ctx *c, const double *d_in, double *d_out, int nbatch, int n, int inverse)
{
    if (!c || !d_in || !d_out) return fail(QPSK_ERR_ARG, "qpsk_fft_batch: null argument");
    if (nbatch <= 0 || n < 1 || (n & (n - 1))) return fail(QPSK_ERR_ARG, "n = %d must be a power of two, nbatch %d > 0", n, nbatch);
    if (n > (1 << 21)) return fail(QPSK_ERR_ARG, "n = %d: transforms above 2^21 points are not supported (the late stages' tile must fit the LDS)", n);
    if (bind(c)) return QPSK_ERR_HIP;
    int log2n = 0;
    while ((1 << log2n) < n) log2n++;
    if (n <= fft_lds_max_n()) {   /* one workgroup per transform, LDS-resident */
        double *tw = nullptr;
        if (int rt = get_twiddles(c, n, &tw)) return rt;
        KERNEL_TRY(launch_fft(d_in, d_out, tw, nbatch, n, log2n, inverse ? 1 : 0, c->stream));
        return QPSK_OK;
    }
    /* two passes over global memory (kernels.hip, fft_big_*); the first gathers bit-reversed, so it cannot run in place */
    double *twb = nullptr, *twn = nullptr;
    if (int rt = get_twiddles(c, fft_big_block(), &twb)) return rt;
    if (int rt = get_twiddles(c, n, &twn)) return rt;
    const double *src = d_in;
    if (d_in == d_out) {
        const size_t bytes = sizeof(double) * 2 * (size_t)nbatch * n;
        int rc = ensure(c, c->mixed, bytes);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(c->mixed.p, d_in, bytes, hipMemcpyDeviceToDevice, c->stream));
        src = (const double *)c->mixed.p;
    }
    KERNEL_TRY(launch_fft_big(src, d_out, twb, twn, nbatch, n, log2n, inverse ? 1 : 0, c->stream));
    return QPSK_OK;
}

/* --------------------------------------------------------------- streams */
static bool stream_scan_ok(const qpsk_ctx *c, bool pcm, bool shared_carrier);
static int streams_usable(qpsk_ctx *c, const char *who);
static bool stream_block_ok(const qpsk_ctx *c, bool pcm = true);

int qpsk_streams_reset(qpsk_ctx *c, int nstreams, double mixer_hz)
{
    if (!c || nstreams <= 0) return fail(QPSK_ERR_ARG, "qpsk_streams_reset: bad argument");
    if (bind(c)) return QPSK_ERR_HIP;
    /* poisoned until this call has COMPLETED: a reset that fails half way (a launch, the last synchronisation) must not leave
     * streams that look usable over half-initialised state */
    c->streams_poisoned = true;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (nstreams != c->nstreams) {
        free_streams(c);
        const size_t n = (size_t)nstreams;
        if (hipMalloc((void **)&c->s_memory, sizeof(float) * 2 * QPSK_NTAPS * n) != hipSuccess ||
            hipMalloc((void **)&c->s_dec, sizeof(float) * 2 * c->nsym * n) != hipSuccess ||
            hipMalloc((void **)&c->s_loop, sizeof(float) * 2 * n) != hipSuccess ||
            hipMalloc((void **)&c->s_ctab, sizeof(float) * 2 * 2 * (size_t)c->prm.frame_size) != hipSuccess ||
            hipMalloc((void **)&c->s_cstate, sizeof(float) * 8) != hipSuccess ||
            hipMalloc((void **)&c->s_mixer, sizeof(float) * 4 * n) != hipSuccess) {
            free_streams(c);
            return fail(QPSK_ERR_ALLOC, "hipMalloc of stream state failed");
        }
        c->nstreams = nstreams;
    }
    const size_t n = (size_t)nstreams;
    HIP_TRY(hipMemsetAsync(c->s_memory, 0, sizeof(float) * 2 * QPSK_NTAPS * n, c->stream));
    HIP_TRY(hipMemsetAsync(c->s_dec, 0, sizeof(float) * 2 * c->nsym * n, c->stream));
    HIP_TRY(hipMemsetAsync(c->s_loop, 0, sizeof(float) * 2 * n, c->stream));
    /* fbb_rx_phase = cmplx(0.0f); fbb_rx_rect = cmplxconj(TAU * hz / FS)  (qpsk.c:341-342) */
    float rect[2];
    qpsk_host_rect(mixer_hz, c->prm.fs, rect);
    std::vector<float> m(4 * n);
    for (size_t i = 0; i < n; i++) { m[4 * i] = 1.0f; m[4 * i + 1] = 0.0f; m[4 * i + 2] = rect[0]; m[4 * i + 3] = rect[1]; }
    HIP_TRY(hipMemcpyAsync(c->s_mixer, m.data(), sizeof(float) * m.size(), hipMemcpyHostToDevice, c->stream));
    /* one carrier for all of them: the first block's phases now, every later block's by the block before it (streamscan.hip MODE 2) */
    const float cst[8] = {1.0f, 0.0f, rect[0], rect[1], 1.0f, 0.0f, 0.0f, 0.0f};
    HIP_TRY(hipMemcpyAsync(c->s_cstate, cst, sizeof cst, hipMemcpyHostToDevice, c->stream));
    c->carrier_shared = false;
    c->carrier_blocks = 0;
    c->carrier_scans = 0;
    c->carrier_pending = nullptr;
    if (c->prm.frame_size % 2 == 0 && (stream_scan_ok(c, true, true) || stream_block_ok(c))) {      /* (a kernel that uses it would take these streams) */
        KERNEL_TRY(launch_carrier_table(c->s_cstate, c->s_ctab, c->prm.frame_size, false, c->stream));
        c->carrier_shared = true;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    /* the one-launch kernel's wave counter: whatever an earlier, failed call left between the host's expectation and the device's count
     * (a polled call whose synchronisation failed had already advanced done_expect), the stream is idle now: back in step */
    if (c->h_done) c->done_expect = *(volatile unsigned *)c->h_done;
    (void)__atomic_exchange_n(c->h_status, 0, __ATOMIC_ACQ_REL);      /* a status word left by the call that poisoned the streams belongs to it */
    c->streams_poisoned = false;
    return QPSK_OK;
}

/* the shared carrier ends: every stream's mixer state = the phase the next block starts from (before any other reader of s_mixer) */
static int carrier_to_streams(qpsk_ctx *c)
{
    if (!c->carrier_shared) return QPSK_OK;
    KERNEL_TRY(launch_carrier_broadcast(c->s_cstate, c->s_mixer, c->nstreams, c->stream));
    c->carrier_shared = false;
    return QPSK_OK;
}

int qpsk_streams_set_loop_state(qpsk_ctx *c, const float *h_state)
{
    if (!c || !h_state) return fail(QPSK_ERR_ARG, "null argument");
    if (int ru = streams_usable(c, "qpsk_streams_set_loop_state")) return ru;
    if (bind(c)) return QPSK_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(c->s_loop, h_state, sizeof(float) * 2 * (size_t)c->nstreams, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QPSK_OK;
}

int qpsk_streams_get_loop_state(qpsk_ctx *c, float *h_state)
{
    if (!c || !h_state) return fail(QPSK_ERR_ARG, "null argument");
    if (int ru = streams_usable(c, "qpsk_streams_get_loop_state")) return ru;
    if (bind(c)) return QPSK_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(h_state, c->s_loop, sizeof(float) * 2 * (size_t)c->nstreams, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return check_status(c);
}

/* One launch per block (streamblock.hip) instead of the five-kernel composition: few, short streams -- the reference's call
 * pattern, where the launches are the cost.  Histogram or fixed timing (the FFT estimate keeps its own kernel). */
static bool stream_block_ok(const qpsk_ctx *c, bool pcm)
{
    const int t = tuned(c->tune.stream_block, -1);
    if (t == 0 || c->prm.timing_mode == QPSK_TIMING_FFT || c->prm.frame_size > stream_block_max_frame() ||
        c->nsym * c->cycles > c->prm.frame_size || tuned(c->tune.generic, 0) != 0 ||
        stream_block_lds_bytes(c->prm.frame_size, c->nsym) > (size_t)MAX_LDS_BYTES)
        return false;
    /* [measured, profiles/r04_streams_short_blocks.txt: PCM streams back to back] the kernels apart cost 70 us (512-sample blocks) to
     * 220 us (2048) whatever the count -- launches and serial chains -- and this kernel ~17 us + 13.5 ns per 512 samples of a stream:
     * it wins up to ~4 M samples per block of all streams (7800 x 512, 4400 x 1024, 2400 x 2048).  Complex input: round 4's first rule */
    const long long nl = (long long)c->nstreams * c->prm.frame_size;
    /* complex input has no carrier chain in front of it: at CYCLES = 8 the kernels apart (hand-scheduled filter, 8-lane scan) take 29 us
     * at 1024 x 512 and 55 us at 256 x 2048, where this kernel takes 27 and 73 -- it keeps the small batches (<= 0.4 M samples); at other
     * rates the kernels apart run the generic scan (72 us at 2048 x 512 against 40 here) and the PCM rule holds */
    return t == 1 || (pcm || c->cycles != 8 ? nl <= 3500000LL : nl <= 400000LL);
}

/* PCM streams that both stream_scan_kernel and the one-launch kernel would take: which of the two */
static bool prefer_stream_scan(const qpsk_ctx *c, bool scan_ok)
{
    return scan_ok && tuned(c->tune.stream_block, -1) != 1;
}

/* pcm / cplx: exactly one of them; every pointer device-visible (device memory or mapped pinned host memory) */
static int streams_block_launch(qpsk_ctx *c, const int16_t *pcm, const float *cplx, const float *loop_in, float *loop_out,
                                uint8_t *sym, float *costas, int32_t *index, const StreamBlockInline *inl = nullptr, bool count = false)
{
    if (int rg = use_context_gains(c)) return rg;
    StreamBlockArgs a{};
    if (pcm) {
        if (c->carrier_shared && tuned(c->tune.stream_carrier, 1) != 0 && c->prm.frame_size % 2 == 0) {
            /* the streams' one carrier (carrier.h): this block's phases from the table, the next block's by workgroup 0's third wave */
            const size_t L2 = 2 * (size_t)c->prm.frame_size;
            a.ctab = reinterpret_cast<const float2 *>(c->s_ctab + L2 * (c->carrier_blocks & 1u));
            a.ctab_next = reinterpret_cast<float2 *>(c->s_ctab + L2 * ((c->carrier_blocks + 1u) & 1u));
            a.cstate = c->s_cstate;
        } else if (int rb = carrier_to_streams(c)) {
            return rb;
        }
    }
    a.pcm = pcm;
    a.cplx = reinterpret_cast<const float2 *>(cplx);
    a.mixer = c->s_mixer;
    a.memory = reinterpret_cast<float2 *>(c->s_memory);
    a.dec = reinterpret_cast<float2 *>(c->s_dec);
    a.loop = c->s_loop;
    a.loop_in = loop_in;
    a.loop_out = loop_out;
    a.taps = c->d_taps;
    a.gains = c->d_gains;
    a.min_freq = c->min_freq;
    a.max_freq = c->max_freq;
    a.frame_size = c->prm.frame_size;
    a.cycles = c->cycles;
    a.nsym = c->nsym;
    a.hist_timing = c->prm.timing_mode == QPSK_TIMING_HIST;
    a.fixed_index = c->prm.fixed_index;
    a.sym = sym;
    a.costas = reinterpret_cast<float2 *>(costas);
    a.index = index;
    a.status = c->d_status;
    a.done = count ? c->d_done : nullptr;
    KERNEL_TRY(launch_stream_block(a, c->nstreams, c->stream, inl));
    if (a.ctab) c->carrier_blocks++;      /* the tables flip only once the launch that fills the next one is in the stream */
    c->last_kernel = "stream_block_kernel";
    return QPSK_OK;
}

static int streams_copy_loop(qpsk_ctx *c, float *d_freq, float *d_phase)
{
    const int n = c->nstreams;
    if (d_phase) HIP_TRY(hipMemcpy2DAsync(d_phase, sizeof(float), c->s_loop, 2 * sizeof(float), sizeof(float), n, hipMemcpyDeviceToDevice, c->stream));
    if (d_freq) HIP_TRY(hipMemcpy2DAsync(d_freq, sizeof(float), c->s_loop + 1, 2 * sizeof(float), sizeof(float), n, hipMemcpyDeviceToDevice, c->stream));
    return QPSK_OK;
}

/* A block of the running streams from the rrc_fir() call on (qpsk.c:125-212).  filtered: c->filtered already holds the filtered
 * block and the delay lines are updated (mix_fir_kernel did both); otherwise d_in is the complex block to filter. */
static int streams_from_filter(qpsk_ctx *c, const float *d_in, bool filtered, uint8_t *d_sym, float *d_freq, float *d_phase,
                               float *d_costas, int32_t *d_index, bool scanned = false)
{
    const int n = c->nstreams, L = c->prm.frame_size, N = c->nsym;
    int rc = ensure(c, c->filtered, sizeof(float) * 2 * (size_t)n * L);
    if (rc) return rc;
    rc = ensure(c, c->index, sizeof(int32_t) * (size_t)n);
    if (rc) return rc;
    float *filt = (float *)c->filtered.p;
    int32_t *idx = (int32_t *)c->index.p;
    if (!filtered) {
        /* qpsk.c:125 */
        KERNEL_TRY(fir_full_rate(c, d_in, c->s_memory, filt, n, L));
        KERNEL_TRY(launch_delay_line(d_in, c->s_memory, n, L, c->stream));
    }
    /* qpsk.c:127-180 (scanned: stream_scan_kernel has left the index, and the filtered block planar by decimation phase) */
    if (scanned) {
        /* fixed timing rides on the same kernel (its scan waves have slack; their index is simply replaced) */
        if (c->prm.timing_mode == QPSK_TIMING_FIXED) KERNEL_TRY(launch_fill_i32(idx, n, c->prm.fixed_index, c->stream));
    } else if (c->prm.timing_mode == QPSK_TIMING_HIST)
        KERNEL_TRY(launch_timing_hist(filt, n, L, c->cycles, idx, nullptr, tuned(c->tune.hist_generic, 0) == 2, c->stream));
    else if (c->prm.timing_mode == QPSK_TIMING_FIXED)
        KERNEL_TRY(launch_fill_i32(idx, n, c->prm.fixed_index, c->stream));
    else if (!d_in)
        return fail(QPSK_ERR_STATE, "internal: the FFT timing estimate needs the unfiltered block");
    else if (int rf = fft_timing_indices(c, d_in, n, idx))   /* stateless: it looks at the raw block from sample 2 on */
        return rf;
    /* qpsk.c:196-212 over decimated_frame[0..N) = the PREVIOUS block's picks, which s_dec holds; qpsk.c:186-191:
     * this block's picks replace them for the next call */
    if (int rg = use_context_gains(c)) return rg;
    rc = costas_over_symbols(c, c->s_dec, n, N, N, c->s_loop, d_sym, d_costas, filt, idx, scanned);
    if (rc) return rc;
    if (d_index) HIP_TRY(hipMemcpyAsync(d_index, idx, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
    c->last_kernel = scanned ? "stream_scan_kernel + costas_pipe_kernel"
                             : tuned(c->tune.generic, 0) ? "filter, timing, costas_kernel, decimate_kernel" : "filter, timing, costas_pipe_kernel";
    return streams_copy_loop(c, d_freq, d_phase);
}

/* Many streams, histogram timing: mixer, filter and scan as ONE kernel (streamscan.hip) -- the carrier recurrences run a tile ahead
 * of the filter waves of the same workgroup, PCM comes in at 2 bytes per sample, no mixed block goes through HBM and the scan
 * reads the filtered samples from LDS.  A workgroup takes 16 streams through the whole block (its time does not shrink with the
 * batch), so it pays from about 2500 streams on; below that the kernels apart are quicker. */
static bool stream_scan_ok(const qpsk_ctx *c, bool pcm, bool shared_carrier)
{
    /* [measured, profiles/r04_streams_blocks.txt] 2560 streams: PCM input 0.94 against 1.03 ms per block with the kernels apart, complex
     * input 0.79 against 0.70; 4096 streams: 1.00 against 1.27 and 0.92 against 1.01.  With the streams' one carrier from the table
     * (MODE 2: no mixer wave, no mixer kernel) PCM input pays from 1024 streams on: 0.73 against 0.79 ms there, 0.75 against 0.93 at
     * 2048, 0.76 against 1.00 at 2560 (profiles/r04_streams_carrier.txt) */
    const int from = pcm ? (shared_carrier ? 1024 : 2560) : 3584;
    bool wanted = c->nstreams >= from;
    /* short blocks (the one-launch kernel's range), shared carrier: a workgroup's time shrinks with the block, so the kernel pays from
     * ~1.4 M samples per block of all streams on -- 106.7 us against 151.9 (one launch per block) and 117.1 (kernels apart) at
     * 1024 x 2048, 61.8 against 95.9 and 80.6 at 2048 x 1024, 41.7 against 71.7 and 61.8 at 4096 x 512; below that the one-launch
     * kernel is ahead (profiles/r04_streams_short_blocks.txt) */
    if (pcm && shared_carrier && c->prm.frame_size <= stream_block_max_frame())
        wanted = (long long)c->nstreams * c->prm.frame_size >= 1400000LL && c->nstreams >= 256;
    return c->taps_symmetric && tuned(c->tune.fir_generic, 0) == 0 && tuned(c->tune.generic, 0) == 0 && (c->cycles == 8 || c->cycles == 4) &&
           (c->prm.timing_mode == QPSK_TIMING_HIST || c->prm.timing_mode == QPSK_TIMING_FIXED) && c->prm.frame_size % stream_scan_tile() == 0 &&
           tuned(c->tune.stream_scan, wanted ? 1 : 0) != 0;
}

static int streams_scanned(qpsk_ctx *c, const int16_t *d_pcm, const float *d_cplx, uint8_t *d_sym, float *d_freq, float *d_phase,
                           float *d_costas, int32_t *d_index)
{
    const int n = c->nstreams, L = c->prm.frame_size;
    int rf = ensure(c, c->filtered, sizeof(float) * 2 * (size_t)n * L);
    if (rf) return rf;
    rf = ensure(c, c->index, sizeof(int32_t) * (size_t)n);
    if (rf) return rf;
    const float *ctab = nullptr;
    float *ctab_next = nullptr;
    if (d_pcm) {
        if (c->carrier_shared && tuned(c->tune.stream_carrier, 1) != 0) {
            ctab = c->s_ctab + 2 * (size_t)L * (c->carrier_blocks & 1u);
            ctab_next = c->s_ctab + 2 * (size_t)L * ((c->carrier_blocks + 1u) & 1u);
        } else if (int rb = carrier_to_streams(c)) {
            return rb;
        }
    }
    KERNEL_TRY(launch_stream_scan(d_pcm, d_cplx, c->s_mixer, c->s_memory, (float *)c->filtered.p, c->d_taps, (int32_t *)c->index.p, n, L,
                                  c->d_status, c->stream, ctab, ctab_next, c->s_cstate, ctab ? c->carrier_scans : 0u, c->cycles));
    if (ctab) {      /* the bookkeeping describes launches that ARE in the stream: the tables flip, the relay's turn advances, the rest of the
                      * next table is owed (a failure between here and the launch that pays it poisons the streams: stream_call_done) */
        c->carrier_blocks++;
        c->carrier_scans++;
        c->carrier_pending = ctab_next;
    }
    const int rc = streams_from_filter(c, nullptr, true, d_sym, d_freq, d_phase, d_costas, d_index, true);
    if (c->carrier_pending) {      /* the loop kernel was never reached: the table is finished all the same */
        float *tab = c->carrier_pending;
        c->carrier_pending = nullptr;
        KERNEL_TRY(launch_carrier_table(c->s_cstate, tab, L, true, c->stream));
    }
    return rc;
}

/* A stream call enqueues several launches and keeps host-side books about them (which carrier table is current, the relay's turn, a
 * table half built).  If it fails between its launches -- or a kernel reports that it gave up (QPSK_ERR_HIP from the status word) -- the
 * carried state of the streams is undefined: the shared carrier is dropped and every later stream call is refused until
 * qpsk_streams_reset().  Argument errors and flagged NUMBERS (QPSK_ERR_RANGE: NaN samples, a phase beyond the bounded wrap) do not
 * poison: the kernels completed and the state is what the reference's would be or fenced as documented. */
static int stream_call_done(qpsk_ctx *c, int rc)
{
    if (rc == QPSK_ERR_HIP || rc == QPSK_ERR_ALLOC) {
        c->streams_poisoned = true;
        c->carrier_shared = false;
        c->carrier_pending = nullptr;
    }
    return rc;
}

static int streams_usable(qpsk_ctx *c, const char *who)
{
    if (c->nstreams <= 0) return fail(QPSK_ERR_STATE, "call qpsk_streams_reset() first");
    if (c->streams_poisoned)
        return fail(QPSK_ERR_STATE, "%s: an earlier stream call failed between its launches; the streams' carried state is undefined until qpsk_streams_reset()", who);
    c->stream_work_unchecked = true;      /* every caller goes on to enqueue stream work (or to read what such work left) */
    return QPSK_OK;
}

static int streams_rx_cplx_impl(qpsk_ctx *c, const float *d_in, uint8_t *d_sym, float *d_freq, float *d_phase,
                                float *d_costas, int32_t *d_index)
{
    const bool scan = stream_scan_ok(c, false, false) && ((uintptr_t)d_in % 16) == 0;
    if (d_sym && stream_block_ok(c, false) && !prefer_stream_scan(c, scan)) {
        if (int rb = streams_block_launch(c, nullptr, d_in, nullptr, nullptr, d_sym, d_costas, d_index)) return rb;
        return streams_copy_loop(c, d_freq, d_phase);
    }
    if (scan)
        return streams_scanned(c, nullptr, d_in, d_sym, d_freq, d_phase, d_costas, d_index);
    return streams_from_filter(c, d_in, false, d_sym, d_freq, d_phase, d_costas, d_index);
}

int qpsk_streams_rx_cplx(qpsk_ctx *c, const float *d_in, uint8_t *d_sym, float *d_freq, float *d_phase,
                         float *d_costas, int32_t *d_index)
{
    if (!c || !d_in) return fail(QPSK_ERR_ARG, "qpsk_streams_rx_cplx: null argument");
    if (int ru = streams_usable(c, "qpsk_streams_rx_cplx")) return ru;
    if (bind(c)) return QPSK_ERR_HIP;
    return stream_call_done(c, streams_rx_cplx_impl(c, d_in, d_sym, d_freq, d_phase, d_costas, d_index));
}

static int streams_rx_pcm_impl(qpsk_ctx *c, const int16_t *d_pcm, uint8_t *d_sym, float *d_freq, float *d_phase,
                               float *d_costas, int32_t *d_index)
{
    const bool scan = stream_scan_ok(c, true, c->carrier_shared && tuned(c->tune.stream_carrier, 1) != 0) && ((uintptr_t)d_pcm % 4) == 0;
    if (d_sym && stream_block_ok(c) && !prefer_stream_scan(c, scan)) {
        if (int rb = streams_block_launch(c, d_pcm, nullptr, nullptr, nullptr, d_sym, d_costas, d_index)) return rb;
        return streams_copy_loop(c, d_freq, d_phase);
    }
    const int n = c->nstreams, L = c->prm.frame_size;
    if (scan)
        return streams_scanned(c, d_pcm, nullptr, d_sym, d_freq, d_phase, d_costas, d_index);
    int rc = ensure(c, c->mixed, sizeof(float) * 2 * (size_t)n * L);
    if (rc) return rc;
    /* qpsk.c:114-120 */
    if (int rb = carrier_to_streams(c)) return rb;
    KERNEL_TRY(launch_mixer(d_pcm, (float *)c->mixed.p, c->s_mixer, n, L, c->stream));
    return streams_rx_cplx_impl(c, (const float *)c->mixed.p, d_sym, d_freq, d_phase, d_costas, d_index);
}

int qpsk_streams_rx_pcm(qpsk_ctx *c, const int16_t *d_pcm, uint8_t *d_sym, float *d_freq, float *d_phase,
                        float *d_costas, int32_t *d_index)
{
    if (!c || !d_pcm) return fail(QPSK_ERR_ARG, "qpsk_streams_rx_pcm: null argument");
    if (int ru = streams_usable(c, "qpsk_streams_rx_pcm")) return ru;
    if (bind(c)) return QPSK_ERR_HIP;
    return stream_call_done(c, streams_rx_pcm_impl(c, d_pcm, d_sym, d_freq, d_phase, d_costas, d_index));
}

/*
 * One block per stream with HOST buffers on both sides -- the call pattern of the reference's main loop
 * (qpsk.c:344-354: fread 512 int16, rx_frame()).  Everything a block needs goes up in ONE pinned copy (PCM, the
 * streams' loop state, the loop gains when they changed), everything it produces comes down in ONE (symbols,
 * costas_frame[], loop state, timing index), and the stream is synchronised ONCE, at the end; the per-call pieces
 * qpsk_dev_upload / qpsk_streams_rx_pcm / qpsk_streams_get_loop_state / qpsk_dev_download synchronise eight times
 * for the same work.
 *   h_pcm [nstreams][frame_size] int16          h_loop_io [nstreams][2] (phase, freq) in and out, may be NULL (carried state)
 *   h_sym [nstreams][nsym] uint8                h_costas [nstreams][nsym][2] float, may be NULL
 *   h_index [nstreams] int32, may be NULL
 */
static int streams_rx_pcm_host_impl(qpsk_ctx *c, const int16_t *h_pcm, float *h_loop_io, uint8_t *h_sym, float *h_costas, int32_t *h_index);
int qpsk_streams_rx_pcm_host(qpsk_ctx *c, const int16_t *h_pcm, float *h_loop_io, uint8_t *h_sym, float *h_costas, int32_t *h_index)
{
    if (!c || !h_pcm || !h_sym) return fail(QPSK_ERR_ARG, "qpsk_streams_rx_pcm_host: null argument");
    if (int ru = streams_usable(c, "qpsk_streams_rx_pcm_host")) return ru;
    if (bind(c)) return QPSK_ERR_HIP;
    return stream_call_done(c, streams_rx_pcm_host_impl(c, h_pcm, h_loop_io, h_sym, h_costas, h_index));
}

static inline void cpu_relax(void)
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}

static int streams_rx_pcm_host_impl(qpsk_ctx *c, const int16_t *h_pcm, float *h_loop_io, uint8_t *h_sym, float *h_costas, int32_t *h_index)
{
    const size_t n = (size_t)c->nstreams, L = (size_t)c->prm.frame_size, N = (size_t)c->nsym;
    /* arena layout, every part 16-byte aligned: in = [loop 8n | pcm 2nL], out = [sym nN | costas 8nN | loop 8n | index 4n] */
    auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t o_loop = 0, o_pcm = al(8 * n), in_bytes = o_pcm + al(2 * n * L);
    const size_t o_sym = in_bytes, o_cos = o_sym + al(n * N), o_lout = o_cos + al(8 * n * N), o_idx = o_lout + al(8 * n),
                 total = o_idx + al(4 * n);
    if (c->stage_cap < total) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->h_stage) hipHostFree(c->h_stage);
        hipFree(c->d_stage);
        c->h_stage = c->d_stage = nullptr;
        c->stage_cap = 0;
        if (hipHostMalloc((void **)&c->h_stage, total, hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&c->d_stage, total) != hipSuccess)
            return fail(QPSK_ERR_ALLOC, "staging buffers of %zu bytes", total);
        c->stage_cap = total;
    }
    if (h_loop_io) memcpy(c->h_stage + o_loop, h_loop_io, 8 * n);
    memcpy(c->h_stage + o_pcm, h_pcm, 2 * n * L);
    if (stream_block_ok(c) && !prefer_stream_scan(c, stream_scan_ok(c, true, c->carrier_shared && tuned(c->tune.stream_carrier, 1) != 0))) {
        /* one launch, no copy engine: the kernel reads the PCM (and the loop state) from the pinned staging buffer and leaves its
         * results there -- a kilobyte in, a few hundred bytes out per stream */
        unsigned char *m = nullptr;
        HIP_TRY(hipHostGetDevicePointer((void **)&m, c->h_stage, 0));
        /* a block of up to 2 KB rides in the kernel arguments (device memory the host writes into): the kernel then reads no host
         * memory at all -- a read of pinned host memory from the GPU costs 8-14 us on this pool, a posted write a fraction of that */
        /* few streams: the waves count themselves off in pinned memory (a system-scope atomic each: ~1 us apiece over PCIe, so only
         * for a handful -- 64 streams took 156 us that way against 47 with the stream's own completion signal) */
        const bool poll = tuned(c->tune.stream_poll, n <= (size_t)StreamBlockInline::MAX_STREAMS ? 1 : 0) != 0;
        StreamBlockInline inl;
        const bool use_inl = n <= (size_t)StreamBlockInline::MAX_STREAMS && n * L <= (size_t)StreamBlockInline::MAX_SAMPLES;
        if (use_inl) {
            if (h_loop_io) memcpy(inl.loop, h_loop_io, 8 * n);
            memcpy(inl.pcm, h_pcm, 2 * n * L);
        }
        if (int rb = streams_block_launch(c, (const int16_t *)(m + o_pcm), nullptr, h_loop_io ? (const float *)(m + o_loop) : nullptr,
                                          (float *)(m + o_lout), m + o_sym, h_costas ? (float *)(m + o_cos) : nullptr, (int32_t *)(m + o_idx),
                                          use_inl ? &inl : nullptr, poll))
            return rb;
        if (poll) {
            /* the kernel's two waves per stream count themselves off in pinned memory behind their last store: watching that word
             * is quicker than the stream's completion signal.  Bounded by the WALL CLOCK (2 s, looked at every 4096 spins): then the
             * ordinary synchronisation takes over, and a counter that is still short after THAT is an error -- the kernel did not run to
             * its end, its results are not in the staging buffer -- with the expectation put back in step with the counter. */
            c->done_expect += 2u * (unsigned)n;
            const volatile unsigned *dn = c->h_done;
            struct timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (unsigned spins = 0; (int)(*dn - c->done_expect) < 0; ) {
                cpu_relax();
                if ((++spins & 4095u) == 0) {
                    clock_gettime(CLOCK_MONOTONIC, &t1);
                    if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > 2.0) break;
                }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            if ((int)(*dn - c->done_expect) < 0) {
                HIP_TRY(hipStreamSynchronize(c->stream));
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                if ((int)(*dn - c->done_expect) < 0) {
                    const unsigned have = *dn, want = c->done_expect;
                    c->done_expect = have;
                    (void)check_status(c);      /* take the kernel's own word, if it left one, out of the status word */
                    return fail(QPSK_ERR_HIP, "stream_block_kernel: %u of %u waves reported completion; the block's results are invalid",
                                2u * (unsigned)n - (want - have), 2u * (unsigned)n);
                }
            }
        } else {
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        if (int st = check_status(c)) return st;
        memcpy(h_sym, c->h_stage + o_sym, n * N);
        if (h_costas) memcpy(h_costas, c->h_stage + o_cos, 8 * n * N);
        if (h_loop_io) memcpy(h_loop_io, c->h_stage + o_lout, 8 * n);
        if (h_index) memcpy(h_index, c->h_stage + o_idx, 4 * n);
        return QPSK_OK;
    }
    const size_t up0 = h_loop_io ? o_loop : o_pcm;
    HIP_TRY(hipMemcpyAsync(c->d_stage + up0, c->h_stage + up0, in_bytes - up0, hipMemcpyHostToDevice, c->stream));
    if (h_loop_io) HIP_TRY(hipMemcpyAsync(c->s_loop, c->d_stage + o_loop, 8 * n, hipMemcpyDeviceToDevice, c->stream));
    uint8_t *d_sym = c->d_stage + o_sym;
    float *d_cos = h_costas ? (float *)(c->d_stage + o_cos) : nullptr;
    int32_t *d_idx = (int32_t *)(c->d_stage + o_idx);
    int rc = streams_rx_pcm_impl(c, (const int16_t *)(c->d_stage + o_pcm), d_sym, nullptr, nullptr, d_cos, d_idx);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_stage + o_lout, c->s_loop, 8 * n, hipMemcpyDeviceToDevice, c->stream));
    /* one copy down: from the symbols to the index (costas_frame[] in between travels even when it is not wanted
     * only if it was computed: without it the two parts around it go separately) */
    if (h_costas) {
        HIP_TRY(hipMemcpyAsync(c->h_stage + o_sym, c->d_stage + o_sym, total - o_sym, hipMemcpyDeviceToHost, c->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(c->h_stage + o_sym, c->d_stage + o_sym, al(n * N), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(c->h_stage + o_lout, c->d_stage + o_lout, total - o_lout, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (int st = check_status(c)) return st;
    memcpy(h_sym, c->h_stage + o_sym, n * N);
    if (h_costas) memcpy(h_costas, c->h_stage + o_cos, 8 * n * N);
    if (h_loop_io) memcpy(h_loop_io, c->h_stage + o_lout, 8 * n);
    if (h_index) memcpy(h_index, c->h_stage + o_idx, 4 * n);
    return QPSK_OK;
}

/* ------------------------------------------------------------ transmit side (N2) */
int qpsk_tx_reset(qpsk_ctx *c, int nstreams, double tx_hz)
{
    if (!c || nstreams <= 0) return fail(QPSK_ERR_ARG, "qpsk_tx_reset: bad argument");
    if (bind(c)) return QPSK_ERR_HIP;
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t n = (size_t)nstreams;
    const size_t hist_bytes = (size_t)tx_history_symbols() * n;
    if (nstreams != c->ntx) {
        hipFree(c->t_hist); hipFree(c->t_mixer);
        c->t_hist = nullptr;
        c->t_mixer = nullptr;
        c->ntx = 0;
        if (hipMalloc((void **)&c->t_hist, hist_bytes) != hipSuccess ||
            hipMalloc((void **)&c->t_mixer, sizeof(float) * 4 * n) != hipSuccess)
            return fail(QPSK_ERR_ALLOC, "hipMalloc of transmitter state failed");
        c->ntx = nstreams;
    }
    /* memset(tx_filter, 0, ...): no symbol sent yet (code 4), i.e. a delay line of zeros */
    HIP_TRY(hipMemsetAsync(c->t_hist, 4, hist_bytes, c->stream));
    /* fbb_tx_phase = cmplx(0.0f); fbb_tx_rect = cmplx(TAU * hz / FS)  (qpsk.c:316,320) */
    float rect[2];
    qpsk_host_rect_tx(tx_hz, c->prm.fs, rect);
    std::vector<float> m(4 * n);
    for (size_t i = 0; i < n; i++) { m[4 * i] = 1.0f; m[4 * i + 1] = 0.0f; m[4 * i + 2] = rect[0]; m[4 * i + 3] = rect[1]; }
    HIP_TRY(hipMemcpyAsync(c->t_mixer, m.data(), sizeof(float) * m.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QPSK_OK;
}

int qpsk_tx_symbols(qpsk_ctx *c, const uint8_t *d_symbols, int nsym, int16_t *d_pcm, float *d_baseband)
{
    if (!c || !d_symbols || (!d_pcm && !d_baseband)) return fail(QPSK_ERR_ARG, "qpsk_tx_symbols: null argument");
    if (c->ntx <= 0) return fail(QPSK_ERR_STATE, "call qpsk_tx_reset() first");
    if (nsym <= 0) return fail(QPSK_ERR_ARG, "nsym = %d", nsym);
    if (bind(c)) return QPSK_ERR_HIP;
    const int n = c->ntx, len = nsym * c->cycles;
    const size_t bytes = sizeof(float) * 2 * (size_t)n * len;
    float *shaped = d_baseband;
    if (!shaped) {
        int rc = ensure(c, c->tx_b, bytes);
        if (rc) return rc;
        shaped = (float *)c->tx_b.p;
    }
    /* qpsk.c:273-282 + 232-238 + 243: Gray map, zero-stuffing and rrc_fir(tx_filter, ...) in one kernel;
     * :248-261 up-mix */
    KERNEL_TRY(launch_tx_shape(d_symbols, c->t_hist, c->d_taps, shaped, n, nsym, c->cycles, c->stream));
    if (d_pcm) KERNEL_TRY(launch_tx_upmix(shaped, d_pcm, c->t_mixer, n, len, c->stream));
    return QPSK_OK;
}

/* ------------------------------------------------------------ bit stages */
int qpsk_crc16_batch(qpsk_ctx *c, const uint8_t *d_data, int npackets, int nbytes, uint16_t *d_crc)
{
    if (!c || !d_data || !d_crc) return fail(QPSK_ERR_ARG, "qpsk_crc16_batch: null argument");
    if (npackets <= 0 || nbytes < 0) return fail(QPSK_ERR_ARG, "npackets %d nbytes %d", npackets, nbytes);
    if (bind(c)) return QPSK_ERR_HIP;
    KERNEL_TRY(launch_crc16(d_data, npackets, nbytes, d_crc, c->stream));
    return QPSK_OK;
}

int qpsk_interleave_batch(qpsk_ctx *c, uint8_t *d_data, int npackets, int nbytes, int dir)
{
    if (!c || !d_data) return fail(QPSK_ERR_ARG, "qpsk_interleave_batch: null argument");
    if (npackets <= 0 || nbytes <= 0 || nbytes * 8 >= 65536 || (dir != 0 && dir != 1))
        return fail(QPSK_ERR_ARG, "npackets %d, nbytes %d (1..8191), dir %d (0|1)", npackets, nbytes, dir);
    if (bind(c)) return QPSK_ERR_HIP;
    KERNEL_TRY(launch_interleave(d_data, npackets, nbytes, qpsk_host_interleave_prime((unsigned)nbytes * 8u), dir, c->stream));
    return QPSK_OK;
}

int qpsk_scramble_batch(qpsk_ctx *c, uint8_t *d_sym, int npackets, int nsym)
{
    if (!c || !d_sym) return fail(QPSK_ERR_ARG, "qpsk_scramble_batch: null argument");
    if (npackets <= 0 || nsym <= 0) return fail(QPSK_ERR_ARG, "npackets %d nsym %d", npackets, nsym);
    if (bind(c)) return QPSK_ERR_HIP;
    int rc = ensure(c, c->keystream, (size_t)nsym);
    if (rc) return rc;
    if (c->keystream_len != nsym) {
        std::vector<unsigned char> ks((size_t)nsym);
        qpsk_host_scramble_keystream(ks.data(), nsym);
        HIP_TRY(hipMemcpy(c->keystream.p, ks.data(), (size_t)nsym, hipMemcpyHostToDevice));
        c->keystream_len = nsym;
    }
    KERNEL_TRY(launch_scramble(d_sym, (const uint8_t *)c->keystream.p, npackets, nsym, c->stream));
    return QPSK_OK;
}

/* ---------------------------------------------------------------- memory */
int qpsk_dev_alloc(qpsk_ctx *c, void **d_ptr, size_t bytes)
{
    if (!c || !d_ptr) return fail(QPSK_ERR_ARG, "null argument");
    if (bind(c)) return QPSK_ERR_HIP;
    hipError_t e = hipMalloc(d_ptr, bytes);
    if (e != hipSuccess) return fail(QPSK_ERR_ALLOC, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return QPSK_OK;
}

int qpsk_dev_free(qpsk_ctx *c, void *d_ptr)
{
    if (!c) return fail(QPSK_ERR_ARG, "null argument");
    if (bind(c)) return QPSK_ERR_HIP;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(d_ptr));
    return QPSK_OK;
}

int qpsk_dev_upload(qpsk_ctx *c, void *d_dst, const void *h_src, size_t bytes)
{
    if (!c || !d_dst || !h_src) return fail(QPSK_ERR_ARG, "null argument");
    if (bind(c)) return QPSK_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QPSK_OK;
}

int qpsk_dev_download(qpsk_ctx *c, void *h_dst, const void *d_src, size_t bytes)
{
    if (!c || !h_dst || !d_src) return fail(QPSK_ERR_ARG, "null argument");
    if (bind(c)) return QPSK_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return check_status(c);   /* the bytes just handed over may come from a call whose kernel flagged its results */
}

/* test hook: order-independent hash of the device sin/cos over float bit patterns [first, first+count) in both signs */
int qpsk_selftest_sincos_hash(qpsk_ctx *c, uint32_t first, uint32_t count, unsigned long long *h_out)
{
    if (!c || !h_out) return fail(QPSK_ERR_ARG, "null argument");
    if (bind(c)) return QPSK_ERR_HIP;
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, sizeof *d));
    HIP_TRY(hipMemsetAsync(d, 0, sizeof *d, c->stream));
    KERNEL_TRY(launch_sincos_hash(first, count, d, c->stream));
    HIP_TRY(hipMemcpyAsync(h_out, d, sizeof *d, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(d));
    return QPSK_OK;
}

} /* extern "C" */
