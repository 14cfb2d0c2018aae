// ubench.hip -- instruction-cost microbenchmarks that size the Costas recurrence and the FIR inner loop
// on gfx950.  Not product code.  hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench
//
// Each kernel runs ONE workgroup of W waves on one CU and reports s_memtime cycles per instruction for
// wave 0 (100 MHz-independent: s_memtime counts shader clocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP 256

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <int KIND>
__global__ void chain(double *out, unsigned long long *cyc, int iters)
{
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
    float fa = (float)a, fb = 1.0000001f, fc = 1e-9f;
    float2 pa = make_float2(fa, fa + 1), pb = make_float2(fb, fb), pc = make_float2(fc, fc);
    double a2 = a + 1, a3 = a + 2, a4 = a + 3;
    float f2 = fa + 1, f3 = fa + 2, f4 = fa + 3;
    int ia = (int)a;
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = now();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));                 // dependent f64 fma
            if (KIND == 1) {                                                                                   // 4 independent f64 fma
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c));
            }
            if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc));              // dependent f32 fma
            if (KIND == 3) {                                                                                   // 4 independent f32
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f2) : "v"(fb), "v"(fc));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(fb), "v"(fc));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f4) : "v"(fb), "v"(fc));
            }
            if (KIND == 4) {                                                                                   // dependent pk_mul + pk_add pair
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa) : "v"(pb));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa) : "v"(pc));
            }
            if (KIND == 5) {                                                                                   // FIR-like: independent pk_mul feeding dependent pk_add
                float2 t;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(pb), "v"(pc));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa) : "v"(t));
            }
            if (KIND == 6) {                                                                                   // same with scalar ops: 2 mul + 2 add
                float t0_, t1_;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0_) : "v"(fb), "v"(fc));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1_) : "v"(fb), "v"(f2));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa) : "v"(t0_));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(f3) : "v"(t1_));
            }
            if (KIND == 7) {                                                                                   // cvt chain f32->f64->i32->f64->f32
                double d;
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(fa));
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ia) : "v"(d));
                asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d) : "v"(ia));
                asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(fa) : "v"(d));
            }
            if (KIND == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));                             // dependent f64 mul
            if (KIND == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));                           // dependent f32 add
            if (KIND == 10) {                                                                                  // 4 independent pk pairs (8 pk instr)
                float2 t;
                asm volatile("v_pk_mul_f32 %0, %1, %2\n\tv_pk_add_f32 %3, %3, %0" : "=&v"(t), "+v"(pb) : "v"(pc), "v"(pa));
            }
        }
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[threadIdx.x] = a + a2 + a3 + a4 + fa + f2 + f3 + f4 + pa.x + pa.y + ia + pb.x;
}

// LDS read throughput: each lane reads consecutive float2 (conflict-free) and accumulates
__global__ void ldsread(float *out, unsigned long long *cyc, int iters)
{
    __shared__ float2 buf[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = make_float2((float)i, 1.0f);
    __syncthreads();
    float2 acc = make_float2(0, 0);
    unsigned long long t0 = now();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 128; r++) {
            float2 v = buf[(threadIdx.x & 63) + r + (i & 7)];
            acc.x += v.x;
            acc.y += v.y;
        }
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[threadIdx.x] = acc.x + acc.y;
}

template <int KIND>
static void run(const char *name, int waves, int instr_per_rep, int lanes = 64)
{
    double *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, 1024 * sizeof(double)));
    CHECK(hipMemset(out, 0, 1024 * sizeof(double)));
    CHECK(hipMalloc(&cyc, 8 * 256));
    const int iters = 64;
    const int threads = lanes < 64 ? lanes : 64 * waves;
    hipLaunchKernelGGL(chain<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(chain<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long h;
    CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-46s waves/CU=%2d lanes=%2d %7.2f cycles/instr (wave 0)\n", name, waves, lanes, (double)h / ((double)iters * REP * instr_per_rep));
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main()
{
    for (int lanes : {32, 16}) {
        run<0>("dependent v_fma_f64", 1, 1, lanes);
        run<1>("4 independent v_fma_f64", 1, 4, lanes);
        run<2>("dependent v_fma_f32", 1, 1, lanes);
        run<3>("4 independent v_fma_f32", 1, 4, lanes);
        run<5>("pk_mul (indep) -> pk_add (dep chain)", 1, 2, lanes);
        run<7>("cvt chain f32>f64>i32>f64>f32", 1, 4, lanes);
        printf("\n");
    }
    for (int w : {1, 8}) {
        run<0>("dependent v_fma_f64", w, 1);
        run<1>("4 independent v_fma_f64", w, 4);
        run<8>("dependent v_mul_f64", w, 1);
        run<2>("dependent v_fma_f32", w, 1);
        run<3>("4 independent v_fma_f32", w, 4);
        run<9>("dependent v_add_f32", w, 1);
        run<4>("dependent v_pk_mul_f32 -> v_pk_add_f32", w, 2);
        run<5>("pk_mul (indep) -> pk_add (dep chain)", w, 2);
        run<10>("pk_mul -> pk_add, both chains", w, 2);
        run<6>("2 v_mul_f32 + 2 v_add_f32 (FIR scalar form)", w, 4);
        run<7>("cvt chain f32>f64>i32>f64>f32", w, 4);
        printf("\n");
    }
    for (int w : {1, 8}) {
        float *out; unsigned long long *cyc;
        CHECK(hipMalloc(&out, 1024 * 4)); CHECK(hipMalloc(&cyc, 64));
        hipLaunchKernelGGL(ldsread, dim3(1), dim3(64 * w), 0, 0, out, cyc, 64);
        hipLaunchKernelGGL(ldsread, dim3(1), dim3(64 * w), 0, 0, out, cyc, 64);
        CHECK(hipDeviceSynchronize());
        unsigned long long h; CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("ds_read_b64 + 2 v_add_f32 per read         waves/CU=%2d  %7.2f cycles/read (wave 0)\n", w, (double)h / (64.0 * 128));
    }
    return 0;
}
