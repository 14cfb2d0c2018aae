// clock_probe.hip -- what the shader clock does while the FIR waves are busy.  Not product code.
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
//
// One workgroup per CU, 6 waves: wave 0 runs a dependent fp64 chain (the shape of the Costas recurrence) and reads
// s_memtime (shader clocks) and s_memrealtime (100 MHz) around it; the other waves either exit (idle run) or
// run packed fp32 multiply/add streams like the FIR waves (loaded run).  Prints the shader clock of both runs
// and the chain's cycles per instruction: if the recurrence slows down under load although it owns its SIMD, the
// clock shows whether that is frequency or contention.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long shader_clock()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__device__ __forceinline__ unsigned long long real_clock()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__global__ void __launch_bounds__(384) probe(double *sink, unsigned long long *out, int iters, int load, int load_iters)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        double a = 1.0 + lane * 1e-9;
        const double b = 1.0000001, c = 1e-9;
        const unsigned long long s0 = shader_clock(), r0 = real_clock();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 64; r++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
        }
        const unsigned long long s1 = shader_clock(), r1 = real_clock();
        if (lane == 0) {
            out[2 * blockIdx.x] = s1 - s0;
            out[2 * blockIdx.x + 1] = r1 - r0;
        }
        sink[blockIdx.x * 64 + lane] = a;
        return;
    }
    if (!load || wave == 4) return;
    float2 acc0 = make_float2(lane, 1.0f), acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const float2 x = make_float2(1.0000001f, 0.9999999f), t = make_float2(1e-7f, -1e-7f);
    for (int i = 0; i < load_iters; i++) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            float2 p0, p1, p2, p3;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(acc0), "v"(x));
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(acc1), "v"(x));
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p2) : "v"(acc2), "v"(x));
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p3) : "v"(acc3), "v"(x));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc0) : "v"(p0), "v"(t));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc1) : "v"(p1), "v"(t));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc2) : "v"(p2), "v"(t));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc3) : "v"(p3), "v"(t));
        }
    }
    sink[65536 + blockIdx.x * 384 + threadIdx.x] = acc0.x + acc1.y + acc2.x + acc3.y;
}

int main()
{
    int ncu = 0;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    double *sink;
    unsigned long long *out;
    CHECK(hipMalloc(&sink, sizeof(double) * (65536 + (size_t)ncu * 384)));
    CHECK(hipMalloc(&out, sizeof(unsigned long long) * 2 * ncu));
    const int iters = 6000;              // 384k dependent fp64 ops, about 1.3 ms
    std::vector<unsigned long long> h(2 * ncu);
    for (int load = 0; load < 2; load++) {
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(probe, dim3(ncu), dim3(384), 0, 0, sink, out, iters, load, 4000);   // outlasts the probe
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h.data(), out, sizeof(unsigned long long) * 2 * ncu, hipMemcpyDeviceToHost));
        double mhz = 0, cpi = 0;
        for (int i = 0; i < ncu; i++) {
            mhz += (double)h[2 * i] / ((double)h[2 * i + 1] / 100.0);
            cpi += (double)h[2 * i] / (64.0 * iters);
        }
        printf("%s: shader clock %.0f MHz, dependent fp64 fma every %.2f cycles (mean over %d CUs)\n",
               load ? "FIR-like load on the other SIMDs" : "other waves idle", mhz / ncu, cpi / ncu, ncu);
    }
    return 0;
}
