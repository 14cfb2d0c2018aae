// ubench_valu_power.hip -- what the chip sustains at its power limit: every SIMD of every CU issuing the receive filter's
// instruction pair (v_pk_mul_f32 with an SGPR operand + v_pk_add_f32, unfused) back to back, optionally beside an HBM stream.
// Runs for `seconds`, printing packed instructions per second; run it under tools/power_probe.py --cmd to get power and clock.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu_power.hip -o /tmp/ubench_valu && /tmp/ubench_valu 5 [waves_per_simd] [stream: 0 none, 1 with, 2 only] [lds reads: 0/1] [random data: 1/0] [workgroups per CU]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP4(X) X X X X
// 4 accumulators, 16 packed instructions per group, 64 per loop iteration
__global__ void __launch_bounds__(1024) valu_kernel(float *sink, const float4 *stream, size_t stream_n, int iters, int only_stream, int lds_reads, int random_data)
{
    __shared__ float4 win[2560];                    /* 40 KB (three workgroups fit a CU): the receive kernel's window reads (ds_read_b128, lanes 144 bytes apart) */
    for (int i = threadIdx.x; i < 2560; i += blockDim.x) {      /* pseudo-random contents (random_data) or constants: switching activity is power */
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return random_data ? (float)(int)(h & 0xffff) * (1.0f / 32768.0f) - 1.0f : 0.5f; };
        win[i] = make_float4(rnd(), rnd(), rnd(), rnd());
    }
    __syncthreads();
    const unsigned wbase = (unsigned)(__UINTPTR_TYPE__)(const __attribute__((address_space(3))) void *)win + (threadIdx.x & 63) * 144u + (threadIdx.x >> 6) * 1024u % 16384u;
    asm volatile("v_mov_b32 v10, 1.0\n\tv_mov_b32 v11, 0.5\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\t"
                 "v_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\ts_mov_b32 s20, 0x3f7fff00\n\ts_mov_b32 s21, 0x3f7fff00\n\t"
                 ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "s20", "s21");
    float4 acc = make_float4(0, 0, 0, 0);
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    size_t pos = tid;
    for (int i = 0; i < iters; i++) {
        if (stream) {      /* one 16-byte load per lane per 64 packed instructions: the receive kernel's ratio (8 B per 63.5 unfused flops) */
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(stream) + pos % stream_n);
            acc.x += v.x;
            pos += nth;
        }
        if (only_stream) continue;
        if (lds_reads) {      /* 9 ds_read_b128 per 64 packed instructions = the lean stream's 69 per 508 */
            asm volatile("ds_read_b128 v[28:31], %0\n\tds_read_b128 v[32:35], %0 offset:16\n\tds_read_b128 v[36:39], %0 offset:32\n\t"
                         "ds_read_b128 v[40:43], %0 offset:48\n\tds_read_b128 v[44:47], %0 offset:64\n\tds_read_b128 v[48:51], %0 offset:80\n\t"
                         "ds_read_b128 v[52:55], %0 offset:96\n\tds_read_b128 v[56:59], %0 offset:112\n\tds_read_b128 v[60:63], %0 offset:128\n\t"
                         :: "v"(wbase + (unsigned)(i & 7) * 1024u) : "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44",
                            "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "memory");
        }
        if (lds_reads) {      /* multiplicands = the window values just read (the previous iteration's: no wait in front of them) */
            asm volatile(REP4(
                "v_pk_mul_f32 v[20:21], s[20:21], v[28:29] op_sel_hi:[0,1]\n\tv_pk_mul_f32 v[22:23], s[20:21], v[30:31] op_sel:[1,0]\n\t"
                "v_pk_mul_f32 v[24:25], s[20:21], v[32:33] op_sel_hi:[0,1]\n\tv_pk_mul_f32 v[26:27], s[20:21], v[34:35] op_sel:[1,0]\n\t"
                "v_pk_add_f32 v[12:13], v[12:13], v[20:21]\n\tv_pk_add_f32 v[14:15], v[14:15], v[22:23]\n\t"
                "v_pk_add_f32 v[16:17], v[16:17], v[24:25]\n\tv_pk_add_f32 v[18:19], v[18:19], v[26:27]\n\t"
                "v_pk_mul_f32 v[20:21], s[20:21], v[36:37] op_sel_hi:[0,1]\n\tv_pk_mul_f32 v[22:23], s[20:21], v[40:41] op_sel:[1,0]\n\t"
                "v_pk_mul_f32 v[24:25], s[20:21], v[44:45] op_sel_hi:[0,1]\n\tv_pk_mul_f32 v[26:27], s[20:21], v[48:49] op_sel:[1,0]\n\t"
                "v_pk_add_f32 v[12:13], v[12:13], v[20:21]\n\tv_pk_add_f32 v[14:15], v[14:15], v[22:23]\n\t"
                "v_pk_add_f32 v[16:17], v[16:17], v[24:25]\n\tv_pk_add_f32 v[18:19], v[18:19], v[26:27]\n\t")
                ::: "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        } else
        asm volatile(REP4(
            "v_pk_mul_f32 v[20:21], s[20:21], v[10:11] op_sel_hi:[0,1]\n\tv_pk_mul_f32 v[22:23], s[20:21], v[10:11] op_sel:[1,0]\n\t"
            "v_pk_mul_f32 v[24:25], s[20:21], v[10:11] op_sel_hi:[0,1]\n\tv_pk_mul_f32 v[26:27], s[20:21], v[10:11] op_sel:[1,0]\n\t"
            "v_pk_add_f32 v[12:13], v[12:13], v[20:21]\n\tv_pk_add_f32 v[14:15], v[14:15], v[22:23]\n\t"
            "v_pk_add_f32 v[16:17], v[16:17], v[24:25]\n\tv_pk_add_f32 v[18:19], v[18:19], v[26:27]\n\t"
            "v_pk_mul_f32 v[20:21], s[20:21], v[10:11] op_sel_hi:[0,1]\n\tv_pk_mul_f32 v[22:23], s[20:21], v[10:11] op_sel:[1,0]\n\t"
            "v_pk_mul_f32 v[24:25], s[20:21], v[10:11] op_sel_hi:[0,1]\n\tv_pk_mul_f32 v[26:27], s[20:21], v[10:11] op_sel:[1,0]\n\t"
            "v_pk_add_f32 v[12:13], v[12:13], v[20:21]\n\tv_pk_add_f32 v[14:15], v[14:15], v[22:23]\n\t"
            "v_pk_add_f32 v[16:17], v[16:17], v[24:25]\n\tv_pk_add_f32 v[18:19], v[18:19], v[26:27]\n\t")
            ::: "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        if (lds_reads) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float r;
    asm volatile("v_add_f32 %0, v12, v14" : "=v"(r));
    if (r == 123.456f || acc.x == 7.0f) sink[0] = r;
}
__global__ void fill_random(unsigned *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + 12345u;
        h ^= h << 13; h ^= h >> 17; h ^= h << 5;
        p[i] = 0x3f000000u | (h & 0x007fffffu) | (h & 0x80000000u);      /* +-[0.5, 1): finite floats with random mantissas */
    }
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const int wps = argc > 2 ? atoi(argv[2]) : 2;                 // waves per SIMD
    const int rnd = argc > 5 ? atoi(argv[5]) : 1;                 // 0: constant operands (what a naive micro-benchmark multiplies)
    const bool stream = argc > 3 && atoi(argv[3]) != 0;
    const int only = argc > 3 && atoi(argv[3]) == 2;          // 2: the loads alone, no arithmetic
    const int lds = argc > 4 ? atoi(argv[4]) : 0;
    const int bpc = argc > 6 ? atoi(argv[6]) : 1;                 // workgroups per CU (each 4 x waves_per_simd waves)                 // 1: the window reads of the receive kernel beside the arithmetic
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 4000;
    float *sink; CHECK(hipMalloc(&sink, 4));
    float4 *buf = nullptr; const size_t n = stream ? ((size_t)1 << 30) / 16 : 0;
    if (stream) {
        CHECK(hipMalloc(&buf, n * 16));
        if (rnd) hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned *>(buf), n * 4);
        else CHECK(hipMemset(buf, 0, n * 16));
        CHECK(hipDeviceSynchronize());
    }
    const auto t0 = std::chrono::steady_clock::now();
    double inst = 0, last_rate = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        CHECK(hipEventRecord(a));
        for (int k = 0; k < 10; k++) hipLaunchKernelGGL(valu_kernel, dim3(cus * bpc), dim3(64 * 4 * wps), 0, 0, sink, buf, n, iters, only, lds, rnd);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        const double packed = 10.0 * cus * bpc * 4 * wps * (double)iters * 64.0;        // wave-level packed instructions
        last_rate = packed / (ms * 1e-3);
        inst += packed;
        CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    }
    printf("child: %d CUs x 4 SIMDs x %d waves x %d%s%s%s: %.3e packed wave-instructions per second = %.2f cycles per instruction and SIMD at 2.4 GHz, "
           "%.1f unfused TFLOP/s%s\n", cus, wps, bpc, stream ? " + HBM stream" : "", lds ? " + LDS window reads" : "", rnd ? ", random data" : ", constant data", last_rate, 2.4e9 * cus * 4 / last_rate, last_rate * 128 / 1e12,
           stream ? "" : "");
    if (stream) printf("child: stream: %.2f TB/s\n", last_rate / 64.0 * 64 * 16 / 1e12);
    return 0;
}
