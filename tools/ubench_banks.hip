// ubench_banks.hip -- (1) does the VGPR bank of a packed / fp64 instruction's operands change its issue cost on gfx950?
// (VGPR r lives in bank r % 4; a 64-bit operand covers two banks.)  (2) what one wave gets of its SIMD's packed / fp64
// rate when 2, 3 or 4 waves of the workgroup run the same stream on each SIMD.  Not product code.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_banks.hip -o /tmp/ubench_banks && /tmp/ubench_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP16(X) X X X X X X X X X X X X X X X X

__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// 8 independent destinations v[16:31]; sources fixed registers chosen per variant
#define BODY(S0, S1, MOD)                                              \
    "v_pk_mul_f32 v[16:17], " S0 ", " S1 " " MOD "\n\t"                \
    "v_pk_mul_f32 v[18:19], " S0 ", " S1 " " MOD "\n\t"                \
    "v_pk_mul_f32 v[20:21], " S0 ", " S1 " " MOD "\n\t"                \
    "v_pk_mul_f32 v[22:23], " S0 ", " S1 " " MOD "\n\t"                \
    "v_pk_mul_f32 v[24:25], " S0 ", " S1 " " MOD "\n\t"                \
    "v_pk_mul_f32 v[26:27], " S0 ", " S1 " " MOD "\n\t"                \
    "v_pk_mul_f32 v[28:29], " S0 ", " S1 " " MOD "\n\t"                \
    "v_pk_mul_f32 v[30:31], " S0 ", " S1 " " MOD "\n\t"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39"

template <int K>
__global__ void bench(unsigned long long *cyc, int iters)
{
    asm volatile("v_mov_b32 v0, 1.0\n\tv_mov_b32 v1, 1.0\n\tv_mov_b32 v2, 1.0\n\tv_mov_b32 v3, 1.0\n\tv_mov_b32 v4, 1.0\n\tv_mov_b32 v5, 1.0\n\t"
                 "v_mov_b32 v6, 1.0\n\tv_mov_b32 v7, 1.0\n\tv_mov_b32 v8, 1.0\n\tv_mov_b32 v9, 1.0\n\tv_mov_b32 v10, 1.0\n\tv_mov_b32 v11, 1.0\n\t"
                 "v_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\tv_mov_b32 v36, 0\n\tv_mov_b32 v37, 0\n\tv_mov_b32 v38, 0\n\tv_mov_b32 v39, 0\n\t" ::: CLOB);
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = now();
    for (int i = 0; i < iters; i++) {
        if (K == 0) asm volatile(REP16(BODY("v[0:1]", "v[4:5]", "op_sel_hi:[0,1]")) ::: CLOB);      // tap bank 0, window bank 0
        if (K == 1) asm volatile(REP16(BODY("v[0:1]", "v[6:7]", "op_sel_hi:[0,1]")) ::: CLOB);      // tap bank 0, window bank 2
        if (K == 2) asm volatile(REP16(BODY("v[0:1]", "v[4:5]", "")) ::: CLOB);                      // plain, same banks
        if (K == 3) asm volatile(REP16(BODY("v[0:1]", "v[6:7]", "")) ::: CLOB);                      // plain, different banks
        if (K == 4) asm volatile(REP16(BODY("v[0:1]", "v[4:5]", "op_sel:[1,0]")) ::: CLOB);         // tap = high dword, same banks
        if (K == 5) asm volatile(REP16(BODY("v[0:1]", "v[6:7]", "op_sel:[1,0]")) ::: CLOB);
        if (K == 6) asm volatile(REP16("v_pk_add_f32 v[32:33], v[32:33], v[16:17]\n\tv_pk_add_f32 v[34:35], v[34:35], v[18:19]\n\tv_pk_add_f32 v[36:37], v[36:37], v[20:21]\n\tv_pk_add_f32 v[38:39], v[38:39], v[22:23]\n\t") ::: CLOB);   // acc vs product: 32 vs 16 same bank, 34 vs 18 same ...
        if (K == 7) asm volatile(REP16("v_pk_add_f32 v[32:33], v[32:33], v[18:19]\n\tv_pk_add_f32 v[34:35], v[34:35], v[16:17]\n\tv_pk_add_f32 v[36:37], v[36:37], v[22:23]\n\tv_pk_add_f32 v[38:39], v[38:39], v[20:21]\n\t") ::: CLOB);   // different banks
        if (K == 8) asm volatile(REP16("v_fma_f64 v[16:17], v[0:1], v[4:5], v[8:9]\n\tv_fma_f64 v[18:19], v[0:1], v[4:5], v[8:9]\n\tv_fma_f64 v[20:21], v[0:1], v[4:5], v[8:9]\n\tv_fma_f64 v[22:23], v[0:1], v[4:5], v[8:9]\n\t") ::: CLOB);   // three operands, one bank pair
        if (K == 9) asm volatile(REP16("v_fma_f64 v[16:17], v[0:1], v[6:7], v[8:9]\n\tv_fma_f64 v[18:19], v[0:1], v[6:7], v[8:9]\n\tv_fma_f64 v[20:21], v[0:1], v[6:7], v[8:9]\n\tv_fma_f64 v[22:23], v[0:1], v[6:7], v[8:9]\n\t") ::: CLOB);   // two in one bank pair
        if (K == 10) asm volatile(REP16("v_fma_f64 v[16:17], v[0:1], v[6:7], s[4:5]\n\tv_fma_f64 v[18:19], v[0:1], v[6:7], s[4:5]\n\tv_fma_f64 v[20:21], v[0:1], v[6:7], s[4:5]\n\tv_fma_f64 v[22:23], v[0:1], v[6:7], s[4:5]\n\t") ::: CLOB);   // spread + SGPR
    }
    unsigned long long t1 = now();
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;      /* every wave: the oldest of a SIMD is served first */
}

template <int K>
static void run(const char *name, int per_iter, int waves)
{
    unsigned long long *cyc;
    CHECK(hipMalloc(&cyc, 8 * 16));
    const int iters = 64;
    hipLaunchKernelGGL(bench<K>, dim3(1), dim3(64 * waves), 0, 0, cyc, iters);
    hipLaunchKernelGGL(bench<K>, dim3(1), dim3(64 * waves), 0, 0, cyc, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[16];
    CHECK(hipMemcpy(h, cyc, 8 * waves, hipMemcpyDeviceToHost));
    unsigned long long mx = 0;
    for (int i = 0; i < waves; i++) mx = h[i] > mx ? h[i] : mx;
    /* SIMD rate: waves / 4 per SIMD run the stream; the slowest one bounds the time in which they all got through */
    printf("%-62s waves/CU=%2d  wave 0 %5.2f, slowest wave %5.2f cycles/instr -> %5.2f cycles per instruction and SIMD\n", name, waves,
           (double)h[0] / ((double)iters * per_iter), (double)mx / ((double)iters * per_iter),
           (double)mx / ((double)iters * per_iter) / (waves < 4 ? 1.0 : waves / 4.0));
    CHECK(hipFree(cyc));
}

// (3) what s_memtime counts and what the shader clock is under load: counter ticks of workgroup 0 / wave 0 against the
// kernel's HIP-event time, for one workgroup alone and for the whole chip running the packed stream
static void clock_test(int grid)
{
    unsigned long long *cyc;
    CHECK(hipMalloc(&cyc, 8 * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    hipLaunchKernelGGL(bench<0>, dim3(grid), dim3(1024), 0, 0, cyc, 200);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(bench<0>, dim3(grid), dim3(1024), 0, 0, cyc, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[16], mx = 0;
    CHECK(hipMemcpy(h, cyc, 8 * 16, hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; i++) mx = h[i] > mx ? h[i] : mx;
    printf("clock: %4d workgroups x 16 waves of pk_mul: %.3f ms by events, %llu s_memtime ticks in the slowest wave -> %.1f ticks/us; %.2f ticks per instruction and SIMD\n",
           grid, ms, mx, (double)mx / (ms * 1e3), (double)mx / ((double)iters * 128) / 4.0);
    CHECK(hipFree(cyc));
}

int main(int argc, char **argv)
{
    if (argc > 1) {
        for (int g : {1, 64, 256, 512}) clock_test(g);
        return 0;
    }
    for (int w : {1, 8, 12, 16}) {      // 8 waves = two per SIMD, 12 = three, 16 = four
        run<0>("pk_mul tap-broadcast(lo), tap and window in the SAME banks", 128, w);
        run<1>("pk_mul tap-broadcast(lo), tap and window in DIFFERENT banks", 128, w);
        run<2>("pk_mul plain, same banks", 128, w);
        run<3>("pk_mul plain, different banks", 128, w);
        run<4>("pk_mul tap-broadcast(hi), same banks", 128, w);
        run<5>("pk_mul tap-broadcast(hi), different banks", 128, w);
        run<6>("pk_add acc += product, same banks", 64, w);
        run<7>("pk_add acc += product, different banks", 64, w);
        run<8>("fma_f64 three VGPR pairs in ONE bank pair", 64, w);
        run<9>("fma_f64 two of three in one bank pair", 64, w);
        run<10>("fma_f64 two VGPR pairs in different banks + SGPR", 64, w);
        printf("\n");
    }
    return 0;
}
