// ubench_step.hip -- the serial wave's stream (costas_asm_run_ring of qpsk_amd/csrc/costas_asm.h, or a variant of that
// header made by tools/ubench_step.py) alone on a CU: cycles per Costas step for 16 / 32 / 64 enabled lanes.
// Not product code; built and named per variant by tools/ubench_step.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include COSTAS_HEADER
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using namespace qpsk;

__global__ void __launch_bounds__(256) k(unsigned long long *cyc, float *sink, int groups, int nl, int real, int target)
{
    __shared__ __attribute__((aligned(16))) float2 dring[64][130];
    __shared__ __attribute__((aligned(16))) float zring[64][132];
    __shared__ int flags[4];
    const int lane = threadIdx.x & 63;
    if ((int)(threadIdx.x >> 6) != target) return;      /* the stream runs in hardware wave `target` = SIMD `target` of the CU */
    for (int i = 0; i < 130; i++) {
        const float a = 0.1309f * i + 1.5707963f * ((i * 7 + lane) & 3) + 0.01f * lane;
        dring[lane][i] = make_float2(cosf(a), sinf(a));
    }
    if (lane == 0) { flags[0] = 1 << 30; flags[1] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    /* lanes >= real run idle loops: gains 0, phase and frequency 0 (no wrap ever) */
#ifdef PAIRED
    /* the paired-lane stream: lanes 2f, 2f + 1 carry loop f (costas_asm_run_ring_pair); nl enabled lanes = nl / 2 loops */
    const int loop = lane >> 1;
#else
    const int loop = lane;
#endif
    float ph = loop < real ? 0.3f : 0.0f, fr = loop < real ? 0.1f : 0.0f;
    const float al = loop < real ? 0x1.4d0d4ap-3f : 0.0f, be = loop < real ? 0x1.d981e8p-7f : 0.0f;
    unsigned long long t0 = 0, t1 = 0, fl = 0;
    unsigned kk = 4;
    if (lane < nl) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#ifdef PAIRED
        costas_asm_run_ring_pair(ph, fr, lds_addr(&dring[loop][0]), lds_addr(&zring[loop][0]), lds_addr(&flags[0]), lds_addr(&flags[1]), kk,
                                 4u + (unsigned)groups, al, be, -1.0f, 1.0f, (lane & 1) != 0, fl);
#else
        costas_asm_run_ring(ph, fr, lds_addr(&dring[lane][0]), lds_addr(&zring[lane][0]), lds_addr(&flags[0]), lds_addr(&flags[1]), kk,
                            4u + (unsigned)groups, al, be, -1.0f, 1.0f, fl);
#endif
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    }
    sink[lane] = ph + fr + zring[loop][5];
    /* the state the loops ended in, as a checksum over the real loops: every bit-exact variant of the stream prints the same one
     * (the rings wrap, so every loop sees the same 128 symbols again and again: a long run from a fixed start) */
    sink[64 + lane] = ph;
    sink[128 + lane] = fr;
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = kk; cyc[2] = fl; }
}

int main(int argc, char **argv)
{
    unsigned long long *cyc, h[3];
    float *sink;
    CHECK(hipMalloc(&cyc, 24));
    CHECK(hipMalloc(&sink, 4 * 192));
    const int groups = 4096;       // 65536 steps
    printf("%-46s cycles/step for enabled lanes / real loops", argc > 1 ? argv[1] : "stream");
#ifdef PAIRED
    const int cfg[8][3] = {{32, 16, 0}, {32, 16, 1}, {32, 16, 2}, {32, 16, 3}, {64, 16, 0}, {64, 32, 0}, {64, 32, 1}, {64, 32, 3}};      // enabled lanes, real loops, wave
#else
    const int cfg[8][3] = {{16, 16, 0}, {16, 16, 1}, {16, 16, 2}, {16, 16, 3}, {64, 16, 0}, {32, 32, 0}, {32, 32, 1}, {32, 32, 3}};      // enabled lanes, real loops, wave
#endif
    unsigned sum = 0;
    for (int c = 0; c < 8; c++) {
        const int nl = cfg[c][0], real = cfg[c][1], target = cfg[c][2];
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, cyc, sink, groups, nl, real, target);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, cyc, sink, groups, nl, real, target);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost));
        const double steps = 16.0 * (double)(h[1] - 4);
        printf("  %d/%d w%d: %5.1f%s", nl, real, target, (double)h[0] / steps, h[2] ? " (flag)" : "");
        if (c == 0) {      /* 16 real loops in every variant's first configuration */
            unsigned hs[192];
            CHECK(hipMemcpy(hs, sink, sizeof(hs), hipMemcpyDeviceToHost));
#ifdef PAIRED
            for (int l = 0; l < 16; l++) sum = sum * 0x01000193u ^ hs[64 + 2 * l] ^ (hs[128 + 2 * l] * 31u) ^ (hs[64 + 2 * l + 1] - hs[64 + 2 * l]);
#else
            for (int l = 0; l < 16; l++) sum = sum * 0x01000193u ^ hs[64 + l] ^ (hs[128 + l] * 31u);
#endif
        }
    }
    printf("  state %08x\n", sum);
    return 0;
}
