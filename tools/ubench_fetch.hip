// ubench_fetch.hip -- is a lone wave's issue rate set by instruction BYTES (fetch) rather than by the instructions?
// One wave (alone on its CU) runs long straight-line blocks of the same operation in its 4-byte (e32), 8-byte (e64 / VOP3)
// and 12-byte (VOP3 + literal... via v_add_f32 with a literal: 8 bytes in e32 form) encodings, dependent and independent,
// then the same blocks with 1 or 2 partner waves on the same SIMD idle / busy.  Not product code.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o /tmp/ubench_fetch && /tmp/ubench_fetch      (tools/measure_all.sh does)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
#define REP256(X) REP64(X) REP64(X) REP64(X) REP64(X)
__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
// K selects the block; 256 instructions per block, `iters` blocks
template <int K> __global__ void bench(unsigned long long *cyc, int iters)
{
    if (threadIdx.x >= 64) return;          // partner waves (if launched) retire: "alone"
    asm volatile("v_mov_b32 v10, 1.0\n\tv_mov_b32 v11, 1.0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0x3ff00000\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0x3ff00000\n\t"
                 "v_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\t" ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21");
    unsigned long long t0 = now();
    for (int i = 0; i < iters; i++) {
        if (K == 0) asm volatile(REP256("v_add_f32_e32 v10, v10, v11\n\t") ::: "v10");                       // 4 B, dependent
        if (K == 1) asm volatile(REP256("v_add_f32_e64 v10, v10, v11\n\t") ::: "v10");                       // 8 B, dependent
        if (K == 2) asm volatile(REP256("v_add_f32_e32 v10, 0x3f800001, v10\n\t") ::: "v10");                // 4 + 4 literal, dependent
        if (K == 3) asm volatile(REP64("v_add_f32_e32 v16, v10, v11\n\tv_add_f32_e32 v17, v10, v11\n\tv_add_f32_e32 v18, v10, v11\n\tv_add_f32_e32 v19, v10, v11\n\t") ::: "v16", "v17", "v18", "v19");   // 4 B, independent
        if (K == 4) asm volatile(REP64("v_add_f32_e64 v16, v10, v11\n\tv_add_f32_e64 v17, v10, v11\n\tv_add_f32_e64 v18, v10, v11\n\tv_add_f32_e64 v19, v10, v11\n\t") ::: "v16", "v17", "v18", "v19");   // 8 B, independent
        if (K == 5) asm volatile(REP256("v_fma_f64 v[12:13], v[12:13], v[14:15], v[14:15]\n\t") ::: "v12", "v13");                    // 8 B fp64 dependent
        if (K == 6) asm volatile(REP256("v_pk_mul_f32 v[12:13], v[12:13], v[14:15]\n\t") ::: "v12", "v13");                            // 8 B packed dependent
        if (K == 7) asm volatile(REP256("v_cvt_f64_f32_e32 v[12:13], v12\n\t") ::: "v12", "v13");                                      // 4 B conversion dependent
        if (K == 8) asm volatile(REP256("v_cvt_f32_f64_e32 v12, v[12:13]\n\t") ::: "v12");                                             // 4 B conversion dependent
        if (K == 9) asm volatile(REP64("v_add_f32_e32 v10, v10, v11\n\tv_fma_f64 v[12:13], v[12:13], v[14:15], v[14:15]\n\tv_add_f32_e32 v16, v16, v11\n\tv_fma_f64 v[20:21], v[20:21], v[14:15], v[14:15]\n\t") ::: "v10", "v12", "v13", "v16", "v20", "v21");   // 4/8 B mixed, 2 chains
        if (K == 10) asm volatile(REP256("s_nop 0\n\t"));                                                                               // 4 B scalar nop
        if (K == 11) asm volatile(REP256("v_mul_f64 v[12:13], v[12:13], v[14:15]\n\t") ::: "v12", "v13");                               // 8 B fp64 mul dependent
        if (K == 12) asm volatile(REP256("v_add_f64 v[12:13], v[12:13], v[14:15]\n\t") ::: "v12", "v13");                               // 8 B fp64 add dependent
        if (K == 13) asm volatile(REP256("v_pk_add_f32 v[12:13], v[12:13], v[14:15]\n\t") ::: "v12", "v13");                            // 8 B packed add dependent
        if (K == 14) asm volatile(REP256("v_fma_f32 v10, v10, v11, v11\n\t") ::: "v10");                                                // 8 B fp32 fma dependent
        if (K == 15) asm volatile(REP256("v_med3_f32 v10, v10, v11, v11\n\t") ::: "v10");                                               // 8 B med3
        if (K == 16) asm volatile(REP256("v_xor_b32_e32 v10, v10, v11\n\t") ::: "v10");                                                 // 4 B int
        if (K == 17) asm volatile(REP256("v_bfi_b32 v10, v11, v10, v11\n\t") ::: "v10");                                                // 8 B int
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int K> static void run(const char *name, int bytes)
{
    unsigned long long *cyc, h;
    CHECK(hipMalloc(&cyc, 8));
    hipLaunchKernelGGL(bench<K>, dim3(1), dim3(64), 0, 0, cyc, 64);
    hipLaunchKernelGGL(bench<K>, dim3(1), dim3(64), 0, 0, cyc, 64);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    const double c = (double)h / (64.0 * 256.0);
    printf("%-44s %2d B  %6.2f cycles per instruction  %5.2f B per cycle\n", name, bytes, c, bytes / c);
    CHECK(hipFree(cyc));
}
int main()
{
    run<0>("dependent v_add_f32_e32", 4);
    run<1>("dependent v_add_f32_e64", 8);
    run<2>("dependent v_add_f32_e32 + literal", 8);
    run<3>("independent v_add_f32_e32 (4 targets)", 4);
    run<4>("independent v_add_f32_e64 (4 targets)", 8);
    run<14>("dependent v_fma_f32", 8);
    run<15>("dependent v_med3_f32", 8);
    run<16>("dependent v_xor_b32_e32", 4);
    run<17>("dependent v_bfi_b32", 8);
    run<5>("dependent v_fma_f64", 8);
    run<11>("dependent v_mul_f64", 8);
    run<12>("dependent v_add_f64", 8);
    run<6>("dependent v_pk_mul_f32", 8);
    run<13>("dependent v_pk_add_f32", 8);
    run<7>("dependent v_cvt_f64_f32_e32", 4);
    run<8>("dependent v_cvt_f32_f64_e32", 4);
    run<9>("two chains: add_e32 / fma_f64 alternating", 6);
    run<10>("s_nop 0", 4);
    return 0;
}
