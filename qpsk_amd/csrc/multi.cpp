/*
 * multi.cpp -- a job of independent frames sharded over the GPUs of one node (include/qpsk_hip.h "MULTI").
 *
 * What it replaces: the reference's `while (fread) rx_frame(frame);` loop (qpsk.c:344-354) with its process-global per-frame state
 * (qpsk.c:36-53, costas_loop.c:13-23) -- here every frame of a batch is its own modem, frames are independent, and a batch of F frames
 * splits into contiguous ranges [r F / N, (r + 1) F / N) (the rule of qpsk_amd/shard.py, which bench.py --gpus N and the gloo tests
 * use), one per device.  No collective, no traffic between devices: what comes back is 1 byte per symbol + 8 bytes per frame, copied
 * per device over PCIe and written at the shard's place of the caller's host arrays ("concatenated on the host").
 *
 * Per shard: one qpsk_ctx on a compute stream of its own, a second stream for the copy-back, TWO result slots (device buffers +
 * pinned host staging) and one host thread that issues the shard's HIP calls, so that N devices are driven in parallel and the
 * copy-back of step k runs while the kernel of step k + 1 does:
 *     begin(slot): compute stream waits until the slot's last copy-back has left it; qpsk_rx_batch -> slot; event;
 *                  copy stream waits for the event; three hipMemcpyAsync to the slot's pinned staging; event
 *     end(slot):   host waits for the slot's copy event, looks at the context's status word, copies staging -> the caller's arrays
 * The two result slots are what makes the overlap legal: the kernel of step k + 1 writes the OTHER slot.
 */
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/qpsk_hip.h"

extern "C" int qpsk_set_error(int code, const char *msg);      /* api.cpp: the thread's last error text */

namespace {

struct Shard {
    int device = 0;
    long long first = 0, count = 0;
    qpsk_ctx *ctx = nullptr;
    hipStream_t compute = nullptr, copy = nullptr;
    float *d_in = nullptr;            /* [count][frame_size][2]; owned unless `borrowed` */
    bool borrowed = false;
    uint8_t *d_sym[2] = {nullptr, nullptr};
    uint8_t *d_pack[2] = {nullptr, nullptr};      /* packed mode: four symbols per byte, what is copied back instead of d_sym */
    float *d_fp[2] = {nullptr, nullptr};          /* freq [count] then phase [count] */
    uint8_t *h_sym[2] = {nullptr, nullptr};       /* pinned */
    float *h_fp[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
    bool in_flight[2] = {false, false};
    /* direct mode: the slot's copy-back goes straight to the caller's (pinned) arrays, at this shard's place */
    uint8_t *dir_sym[2] = {nullptr, nullptr};
    float *dir_freq[2] = {nullptr, nullptr}, *dir_phase[2] = {nullptr, nullptr};
    /* the shard's host thread */
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    int cmd = 0;                      /* 0 idle, 1 begin, 2 end, 3 quit */
    int cmd_slot = 0;
    uint8_t *out_sym = nullptr;
    float *out_freq = nullptr, *out_phase = nullptr;
    int result = 0;
    bool busy = false;
    char err[512] = {0};
};

} // namespace

struct qpsk_multi {
    qpsk_params prm{};
    int nsym = 0;
    bool packed = false;      /* h_sym rows are ceil(nsym / 4) bytes (qpsk_multi_set_packed) */
    size_t row_bytes() const { return packed ? (size_t)(nsym + 3) / 4 : (size_t)nsym; }
    long long total = 0;
    std::vector<Shard *> shards;
};

namespace {

#define M_HIP(s, expr)                                                                                           \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            snprintf((s)->err, sizeof((s)->err), "device %d: %s: %s", (s)->device, #expr, hipGetErrorString(e_)); \
            return QPSK_ERR_HIP;                                                                                 \
        }                                                                                                        \
    } while (0)

int shard_begin(qpsk_multi *mj, Shard *s, int slot)
{
    if (s->count == 0) return QPSK_OK;
    M_HIP(s, hipSetDevice(s->device));
    if (s->in_flight[slot]) {
        snprintf(s->err, sizeof(s->err), "qpsk_multi_rx_begin: slot %d is still in flight (call qpsk_multi_rx_end for it first)", slot);
        return QPSK_ERR_STATE;
    }
    /* the kernel may overwrite the slot only once its previous copy-back has left it */
    M_HIP(s, hipStreamWaitEvent(s->compute, s->copied[slot], 0));
    const int rc = qpsk_rx_batch(s->ctx, s->d_in, (int)s->count, s->d_sym[slot], s->d_fp[slot], s->d_fp[slot] + s->count, nullptr,
                                 nullptr, nullptr);
    if (rc) {
        snprintf(s->err, sizeof(s->err), "device %d: %s", s->device, qpsk_last_error());
        return rc;
    }
    const uint8_t *d_out = s->d_sym[slot];
    const size_t rb = mj->row_bytes();
    if (mj->packed) {
        const int rp = qpsk_pack_symbols(s->ctx, s->d_sym[slot], s->count, mj->nsym, s->d_pack[slot]);
        if (rp) {
            snprintf(s->err, sizeof(s->err), "device %d: %s", s->device, qpsk_last_error());
            return rp;
        }
        d_out = s->d_pack[slot];
    }
    M_HIP(s, hipEventRecord(s->done[slot], s->compute));
    M_HIP(s, hipStreamWaitEvent(s->copy, s->done[slot], 0));
    if (s->dir_sym[slot] || s->dir_freq[slot] || s->dir_phase[slot]) {
        if (s->dir_sym[slot])
            M_HIP(s, hipMemcpyAsync(s->dir_sym[slot] + (size_t)s->first * rb, d_out, (size_t)s->count * rb, hipMemcpyDeviceToHost, s->copy));
        if (s->dir_freq[slot])
            M_HIP(s, hipMemcpyAsync(s->dir_freq[slot] + s->first, s->d_fp[slot], sizeof(float) * (size_t)s->count, hipMemcpyDeviceToHost, s->copy));
        if (s->dir_phase[slot])
            M_HIP(s, hipMemcpyAsync(s->dir_phase[slot] + s->first, s->d_fp[slot] + s->count, sizeof(float) * (size_t)s->count, hipMemcpyDeviceToHost, s->copy));
    } else {
        M_HIP(s, hipMemcpyAsync(s->h_sym[slot], d_out, (size_t)s->count * rb, hipMemcpyDeviceToHost, s->copy));
        M_HIP(s, hipMemcpyAsync(s->h_fp[slot], s->d_fp[slot], sizeof(float) * 2 * (size_t)s->count, hipMemcpyDeviceToHost, s->copy));
    }
    M_HIP(s, hipEventRecord(s->copied[slot], s->copy));
    s->in_flight[slot] = true;
    return QPSK_OK;
}

int shard_end(qpsk_multi *mj, Shard *s, int slot)
{
    if (s->count == 0) return QPSK_OK;
    M_HIP(s, hipSetDevice(s->device));
    if (!s->in_flight[slot]) {
        snprintf(s->err, sizeof(s->err), "qpsk_multi_rx_end: nothing in flight in slot %d", slot);
        return QPSK_ERR_STATE;
    }
    s->in_flight[slot] = false;
    M_HIP(s, hipEventSynchronize(s->copied[slot]));
    /* did a kernel flag its results?  (the status word, without waiting for the step that may already run behind this one) */
    const int rc = qpsk_ctx_check(s->ctx);
    if (rc) {
        snprintf(s->err, sizeof(s->err), "device %d: %s", s->device, qpsk_last_error());
        return rc;
    }
    if (s->dir_sym[slot] || s->dir_freq[slot] || s->dir_phase[slot]) return QPSK_OK;      /* already where the caller wants it */
    if (s->out_sym) memcpy(s->out_sym + (size_t)s->first * mj->row_bytes(), s->h_sym[slot], (size_t)s->count * mj->row_bytes());
    if (s->out_freq) memcpy(s->out_freq + s->first, s->h_fp[slot], sizeof(float) * (size_t)s->count);
    if (s->out_phase) memcpy(s->out_phase + s->first, s->h_fp[slot] + s->count, sizeof(float) * (size_t)s->count);
    return QPSK_OK;
}

void shard_thread(qpsk_multi *mj, Shard *s)
{
    for (;;) {
        int cmd, slot;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [&] { return s->cmd != 0; });
            cmd = s->cmd;
            slot = s->cmd_slot;
        }
        int rc = QPSK_OK;
        if (cmd == 1) rc = shard_begin(mj, s, slot);
        else if (cmd == 2) rc = shard_end(mj, s, slot);
        {
            std::lock_guard<std::mutex> lk(s->mu);
            s->result = rc;
            s->cmd = 0;
            s->busy = false;
        }
        s->cv.notify_all();
        if (cmd == 3) return;
    }
}

/* every shard's thread runs `cmd`; returns the first failure with that shard's text */
int run_all(qpsk_multi *mj, int cmd, int slot, uint8_t *h_sym, float *h_freq, float *h_phase)
{
    for (Shard *s : mj->shards) {
        std::lock_guard<std::mutex> lk(s->mu);
        s->cmd = cmd;
        s->cmd_slot = slot;
        s->out_sym = h_sym;
        s->out_freq = h_freq;
        s->out_phase = h_phase;
        s->busy = true;
        s->err[0] = 0;
        s->cv.notify_all();
    }
    int first_rc = QPSK_OK;
    for (Shard *s : mj->shards) {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return !s->busy; });
        if (s->result && !first_rc) first_rc = qpsk_set_error(s->result, s->err);
    }
    return first_rc;
}

void free_shard_buffers(Shard *s)
{
    hipSetDevice(s->device);
    if (s->compute) hipStreamSynchronize(s->compute);
    if (s->copy) hipStreamSynchronize(s->copy);
    if (s->d_in && !s->borrowed) hipFree(s->d_in);
    s->d_in = nullptr;
    s->borrowed = false;
    for (int k = 0; k < 2; k++) {
        if (s->d_sym[k]) hipFree(s->d_sym[k]);
        if (s->d_pack[k]) hipFree(s->d_pack[k]);
        s->d_pack[k] = nullptr;
        if (s->d_fp[k]) hipFree(s->d_fp[k]);
        if (s->h_sym[k]) hipHostFree(s->h_sym[k]);
        if (s->h_fp[k]) hipHostFree(s->h_fp[k]);
        s->d_sym[k] = nullptr; s->d_fp[k] = nullptr; s->h_sym[k] = nullptr; s->h_fp[k] = nullptr;
        s->in_flight[k] = false;
    }
}

} // namespace

extern "C" {

int qpsk_multi_create(qpsk_multi **out, const int *devices, int ndev, const qpsk_params *p)
{
    if (!out || !devices || ndev <= 0 || !p) return qpsk_set_error(QPSK_ERR_ARG, "qpsk_multi_create: null argument or no device");
    qpsk_multi *mj = new qpsk_multi;
    mj->prm = *p;
    for (int r = 0; r < ndev; r++) {
        Shard *s = new Shard;
        s->device = devices[r];
        mj->shards.push_back(s);
        bool ok = hipSetDevice(s->device) == hipSuccess &&
                  hipStreamCreateWithFlags(&s->compute, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking) == hipSuccess;
        for (int k = 0; k < 2 && ok; k++)
            ok = hipEventCreateWithFlags(&s->done[k], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&s->copied[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            qpsk_multi_destroy(mj);
            return qpsk_set_error(QPSK_ERR_HIP, "qpsk_multi_create: streams / events on a device could not be created");
        }
        const int rc = qpsk_ctx_create(&s->ctx, s->device, p, s->compute);
        if (rc) {
            qpsk_multi_destroy(mj);
            return rc;
        }
        mj->nsym = qpsk_ctx_nsym(s->ctx);
    }
    for (Shard *s : mj->shards) s->th = std::thread(shard_thread, mj, s);
    *out = mj;
    return QPSK_OK;
}

void qpsk_multi_destroy(qpsk_multi *mj)
{
    if (!mj) return;
    for (Shard *s : mj->shards) {
        if (s->th.joinable()) {
            {
                std::lock_guard<std::mutex> lk(s->mu);
                s->cmd = 3;
                s->busy = true;
            }
            s->cv.notify_all();
            s->th.join();
        }
        free_shard_buffers(s);
        hipSetDevice(s->device);
        for (int k = 0; k < 2; k++) {
            if (s->done[k]) hipEventDestroy(s->done[k]);
            if (s->copied[k]) hipEventDestroy(s->copied[k]);
        }
        if (s->ctx) qpsk_ctx_destroy(s->ctx);
        if (s->compute) hipStreamDestroy(s->compute);
        if (s->copy) hipStreamDestroy(s->copy);
        delete s;
    }
    delete mj;
}

int qpsk_multi_shards(const qpsk_multi *mj) { return mj ? (int)mj->shards.size() : 0; }

int qpsk_multi_load(qpsk_multi *mj, long long total_frames, const float *h_in)
{
    if (!mj || total_frames <= 0) return qpsk_set_error(QPSK_ERR_ARG, "qpsk_multi_load: null job or no frames");
    const long long N = (long long)mj->shards.size();
    const size_t frame_bytes = sizeof(float) * 2 * (size_t)mj->prm.frame_size;
    mj->total = 0;      /* no job until every shard stands: a failed load leaves nothing qpsk_multi_rx_begin would run on */
    for (Shard *s : mj->shards) {
        free_shard_buffers(s);
        s->first = s->count = 0;
    }
    if ((total_frames + N - 1) / N > 0x7fffffffLL) return qpsk_set_error(QPSK_ERR_ARG, "qpsk_multi_load: more than 2^31 frames in a shard");
    for (long long r = 0; r < N; r++) {
        Shard *s = mj->shards[(size_t)r];
        s->first = r * total_frames / N;
        s->count = (r + 1) * total_frames / N - s->first;
        if (s->count == 0) continue;
        const size_t n = (size_t)s->count;
        bool ok = hipSetDevice(s->device) == hipSuccess && hipMalloc((void **)&s->d_in, n * frame_bytes) == hipSuccess;
        for (int k = 0; k < 2 && ok; k++)
            ok = hipMalloc((void **)&s->d_sym[k], n * (size_t)mj->nsym) == hipSuccess &&
                 hipMalloc((void **)&s->d_pack[k], n * ((size_t)(mj->nsym + 3) / 4)) == hipSuccess &&
                 hipMalloc((void **)&s->d_fp[k], sizeof(float) * 2 * n) == hipSuccess &&
                 hipHostMalloc((void **)&s->h_sym[k], n * (size_t)mj->nsym, hipHostMallocDefault) == hipSuccess &&
                 hipHostMalloc((void **)&s->h_fp[k], sizeof(float) * 2 * n, hipHostMallocDefault) == hipSuccess &&
                 hipEventRecord(s->copied[k], s->copy) == hipSuccess;      /* "the slot is free" */
        if (ok && h_in)
            ok = hipMemcpy(s->d_in, h_in + (size_t)s->first * 2 * (size_t)mj->prm.frame_size, n * frame_bytes, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) {
            for (Shard *t : mj->shards) {
                free_shard_buffers(t);
                t->first = t->count = 0;
            }
            return qpsk_set_error(QPSK_ERR_ALLOC, "qpsk_multi_load: allocation or upload of a shard failed");
        }
    }
    mj->total = total_frames;
    return QPSK_OK;
}

int qpsk_multi_shard(qpsk_multi *mj, int r, int *device, long long *first, long long *count, qpsk_ctx **ctx, float **d_in)
{
    if (!mj || r < 0 || r >= (int)mj->shards.size()) return qpsk_set_error(QPSK_ERR_ARG, "qpsk_multi_shard: no such shard");
    Shard *s = mj->shards[(size_t)r];
    if (device) *device = s->device;
    if (first) *first = s->first;
    if (count) *count = s->count;
    if (ctx) *ctx = s->ctx;
    if (d_in) *d_in = s->d_in;
    return QPSK_OK;
}

int qpsk_multi_use_device_input(qpsk_multi *mj, int r, const float *d_in)
{
    if (!mj || r < 0 || r >= (int)mj->shards.size() || !d_in) return qpsk_set_error(QPSK_ERR_ARG, "qpsk_multi_use_device_input: bad argument");
    Shard *s = mj->shards[(size_t)r];
    if (s->count == 0) return qpsk_set_error(QPSK_ERR_STATE, "qpsk_multi_use_device_input: call qpsk_multi_load first");
    if (s->in_flight[0] || s->in_flight[1]) return qpsk_set_error(QPSK_ERR_STATE, "qpsk_multi_use_device_input: a slot is in flight");
    if (hipSetDevice(s->device) != hipSuccess || hipStreamSynchronize(s->compute) != hipSuccess)
        return qpsk_set_error(QPSK_ERR_HIP, "qpsk_multi_use_device_input: the shard's device or stream is not usable");
    if (s->d_in && !s->borrowed) hipFree(s->d_in);
    s->d_in = const_cast<float *>(d_in);
    s->borrowed = true;
    return QPSK_OK;
}

int qpsk_multi_set_direct_output(qpsk_multi *mj, int slot, uint8_t *h_sym, float *h_freq, float *h_phase)
{
    if (!mj || (slot != 0 && slot != 1)) return qpsk_set_error(QPSK_ERR_ARG, "qpsk_multi_set_direct_output: bad job or slot");
    for (Shard *s : mj->shards) {
        if (s->in_flight[slot]) return qpsk_set_error(QPSK_ERR_STATE, "qpsk_multi_set_direct_output: the slot is in flight");
        s->dir_sym[slot] = h_sym;
        s->dir_freq[slot] = h_freq;
        s->dir_phase[slot] = h_phase;
    }
    return QPSK_OK;
}

int qpsk_multi_set_packed(qpsk_multi *mj, int on)
{
    if (!mj) return qpsk_set_error(QPSK_ERR_ARG, "qpsk_multi_set_packed: null job");
    for (Shard *s : mj->shards)
        if (s->in_flight[0] || s->in_flight[1]) return qpsk_set_error(QPSK_ERR_STATE, "qpsk_multi_set_packed: a slot is in flight");
    mj->packed = on != 0;
    return QPSK_OK;
}

int qpsk_host_alloc(void **h_ptr, size_t bytes)
{
    if (!h_ptr) return qpsk_set_error(QPSK_ERR_ARG, "qpsk_host_alloc: null argument");
    if (hipHostMalloc(h_ptr, bytes, hipHostMallocPortable) != hipSuccess) return qpsk_set_error(QPSK_ERR_ALLOC, "qpsk_host_alloc: hipHostMalloc failed");
    return QPSK_OK;
}

int qpsk_host_free(void *h_ptr)
{
    return hipHostFree(h_ptr) == hipSuccess ? QPSK_OK : qpsk_set_error(QPSK_ERR_HIP, "qpsk_host_free: hipHostFree failed");
}

int qpsk_multi_rx_begin(qpsk_multi *mj, int slot)
{
    if (!mj || (slot != 0 && slot != 1) || mj->total <= 0) return qpsk_set_error(QPSK_ERR_ARG, "qpsk_multi_rx_begin: bad job or slot (load a job first; slots are 0 and 1)");
    return run_all(mj, 1, slot, nullptr, nullptr, nullptr);
}

int qpsk_multi_rx_end(qpsk_multi *mj, int slot, uint8_t *h_sym, float *h_freq, float *h_phase)
{
    if (!mj || (slot != 0 && slot != 1) || mj->total <= 0) return qpsk_set_error(QPSK_ERR_ARG, "qpsk_multi_rx_end: bad job or slot");
    return run_all(mj, 2, slot, h_sym, h_freq, h_phase);
}

} // extern "C"
