// ubench_fir.hip -- the FIR step of rx_fused_pipe_kernel in isolation: how fast does a wave issue it, alone on a
// SIMD or with company, with and without the LDS reads?  Not product code.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 tools/ubench_fir.hip -o build_tools/ubench_fir
//
// One workgroup on one CU, W waves; every wave sweeps `chunks` chunks of the R = 4 sliding-window filter
// (151 window positions, 4 packed multiplies + 4 packed adds each, window values and taps from LDS one block of
// 8 positions ahead, order pinned as in the product).  Prints shader cycles per packed instruction of wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int NTAPS = 127, C = 8, R = 4, TSTEPS = NTAPS + C * (R - 1), NB = (TSTEPS + C - 1) / C, PAD = 32;

template <int I> struct IntC { static constexpr int value = I; };
template <int FIRST, int LAST, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (FIRST < LAST) { f(IntC<FIRST>{}); static_for<FIRST + 1, LAST>(f); }
}

__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <bool LDS_READS>
__global__ void __launch_bounds__(512) fir(float *sink, unsigned long long *cyc, int chunks)
{
    __shared__ __attribute__((aligned(16))) float taps[128];
    __shared__ float2 win[8][704];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane & 15;
    for (int i = threadIdx.x; i < 128; i += blockDim.x) taps[i] = 1e-3f * (i + 1);
    for (int i = threadIdx.x; i < 8 * 704; i += blockDim.x) (&win[0][0])[i] = make_float2(1.0f + i * 1e-6f, 1.0f - i * 1e-6f);
    __syncthreads();
    const float2 *rd = &win[wave & 7][0] + 33 * q + 170 * (lane >> 4) % 16;
    const float4 *taps4 = reinterpret_cast<const float4 *>(taps);
    v2f ac[R];
    for (int r = 0; r < R; r++) ac[r] = v2f{0.0f, 0.0f};
    float tg[R + 1][C];
    float2 wv[2][C];
    for (int tb = 0; tb < R + 1; tb++) for (int u = 0; u < C; u++) tg[tb][u] = 1e-3f * (u + 1);
    for (int u = 0; u < C; u++) wv[0][u] = wv[1][u] = make_float2(1.0f + u, 1.0f - u);
    auto fetch_block = [&](int tb) {
        if (!LDS_READS) return;
        if (tb * C < NTAPS) {
            const float4 ta = taps4[2 * tb], tb4 = taps4[2 * tb + 1];
            float *g_ = tg[tb % (R + 1)];
            g_[0] = ta.x; g_[1] = ta.y; g_[2] = ta.z; g_[3] = ta.w;
            g_[4] = tb4.x; g_[5] = tb4.y; g_[6] = tb4.z; g_[7] = tb4.w;
        }
#pragma unroll
        for (int u = 0; u < C; u++) {
            const int t = tb * C + u;
            if (t < TSTEPS) wv[tb & 1][u] = rd[t + t / PAD];
        }
    };
    const unsigned long long t0 = now();
    for (int c = 0; c < chunks; c++) {
        fetch_block(0);
        static_for<0, NB>([&](auto tbc) {
            constexpr int tb = decltype(tbc)::value;
            if (tb + 1 < NB) fetch_block(tb + 1);
            static_for<0, C>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                constexpr int t = tb * C + u;
                if constexpr (t < TSTEPS) {
                    const v2f v = v2f{wv[tb & 1][u].x, wv[tb & 1][u].y};
                    constexpr bool ok0 = t < NTAPS, ok1 = t >= C && t - C < NTAPS, ok2 = t >= 2 * C && t - 2 * C < NTAPS,
                                   ok3 = t >= 3 * C && t - 3 * C < NTAPS;
                    v2f p0, p1, p2, p3;
                    if constexpr (ok0) p0 = v * tg[(tb + (R + 1)) % (R + 1)][u];
                    if constexpr (ok1) p1 = v * tg[(tb - 1 + (R + 1)) % (R + 1)][u];
                    if constexpr (ok2) p2 = v * tg[(tb - 2 + (R + 1)) % (R + 1)][u];
                    if constexpr (ok3) p3 = v * tg[(tb - 3 + (R + 1)) % (R + 1)][u];
                    if constexpr (ok0 && ok1 && ok2 && ok3) {
                        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                    } else {
                        if constexpr (ok0) asm volatile("" : "+v"(p0));
                        if constexpr (ok1) asm volatile("" : "+v"(p1));
                        if constexpr (ok2) asm volatile("" : "+v"(p2));
                        if constexpr (ok3) asm volatile("" : "+v"(p3));
                    }
                    if constexpr (ok0) ac[0] = ac[0] + p0;
                    if constexpr (ok1) ac[1] = ac[1] + p1;
                    if constexpr (ok2) ac[2] = ac[2] + p2;
                    if constexpr (ok3) ac[3] = ac[3] + p3;
                    asm volatile("" : "+v"(ac[0]), "+v"(ac[1]), "+v"(ac[2]), "+v"(ac[3]));
                }
            });
        });
    }
    const unsigned long long t1 = now();
    if (lane == 0) cyc[wave] = t1 - t0;
    sink[threadIdx.x] = ac[0].x + ac[1].y + ac[2].x + ac[3].y;
}

int main()
{
    float *sink;
    unsigned long long *cyc;
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMalloc(&cyc, 64));
    const int chunks = 200;
    const double instr = 2.0 * 4 * NTAPS;      // packed multiplies + adds per chunk
    for (int lds = 1; lds >= 0; lds--) {
        for (int waves : {1, 2, 3, 4, 5, 8}) {
            unsigned long long h[8];
            for (int rep = 0; rep < 2; rep++) {
                if (lds) hipLaunchKernelGGL(fir<true>, dim3(1), dim3(64 * waves), 0, 0, sink, cyc, chunks);
                else hipLaunchKernelGGL(fir<false>, dim3(1), dim3(64 * waves), 0, 0, sink, cyc, chunks);
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
            printf("%s, %d wave(s) in the workgroup: wave 0 %.2f cycles per packed instruction (%.0f per chunk)\n",
                   lds ? "LDS reads" : "registers only", waves, (double)h[0] / chunks / instr, (double)h[0] / chunks);
        }
    }
    return 0;
}
