// ubench_rows.hip -- does the cost of a VALU instruction depend on how many 16-lane rows of the wave are enabled?
// One wave runs dependent chains and independent streams of fp64 / fp32 / packed instructions under exec masks of
// 16, 32, 48 and 64 lanes.  Not product code.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_rows.hip -o /tmp/ubench_rows && /tmp/ubench_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
template <int K> __global__ void bench(unsigned long long *cyc, unsigned long long mask, int iters)
{
    unsigned long long t0 = 0, t1 = 0, save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1" : "=&s"(save) : "s"(mask));
    asm volatile("v_mov_b32 v10, 1.0\n\tv_mov_b32 v11, 1.0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0x3ff00000\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0x3ff00000\n\t"
                 "v_mov_b32 v16, 0\n\tv_mov_b32 v17, 0x3ff00000\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0x3ff00000\n\t" ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19");
    t0 = now();
    for (int i = 0; i < iters; i++) {
        if (K == 0) asm volatile(REP64("v_fma_f64 v[12:13], v[12:13], v[14:15], v[14:15]\n\t") ::: "v12", "v13");                 // dependent fp64
        if (K == 1) asm volatile(REP64("v_fma_f32 v10, v10, v11, v11\n\t") ::: "v10");                                             // dependent fp32
        if (K == 2) asm volatile(REP64("v_pk_mul_f32 v[12:13], v[12:13], v[14:15]\n\t") ::: "v12", "v13");                         // dependent packed
        if (K == 3) asm volatile(REP8(REP8("v_fma_f64 v[16:17], v[12:13], v[14:15], v[14:15]\n\t")) ::: "v16", "v17");               // independent fp64
        if (K == 4) asm volatile(REP64("v_cvt_f64_f32 v[12:13], v12\n\t") ::: "v12", "v13");                                        // dependent conversion
        if (K == 5) asm volatile(REP8(REP8("v_pk_mul_f32 v[16:17], v[12:13], v[14:15]\n\t")) ::: "v16", "v17");                      // independent packed
    }
    t1 = now();
    asm volatile("s_mov_b64 exec, %0" ::"s"(save));
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int K> static void run(const char *name)
{
    unsigned long long *cyc, h;
    CHECK(hipMalloc(&cyc, 8));
    printf("%-28s", name);
    for (unsigned long long m : {0xffffull, 0xffffffffull, 0xffffffffffffull, ~0ull}) {
        hipLaunchKernelGGL(bench<K>, dim3(1), dim3(64), 0, 0, cyc, m, 64);
        hipLaunchKernelGGL(bench<K>, dim3(1), dim3(64), 0, 0, cyc, m, 64);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("  %5.2f", (double)h / (64.0 * 64.0));
    }
    printf("   cycles per instruction with 16 / 32 / 48 / 64 lanes enabled\n");
    CHECK(hipFree(cyc));
}
int main()
{
    run<0>("dependent v_fma_f64");
    run<1>("dependent v_fma_f32");
    run<2>("dependent v_pk_mul_f32");
    run<4>("dependent v_cvt_f64_f32");
    run<3>("independent v_fma_f64");
    run<5>("independent v_pk_mul_f32");
    return 0;
}
