// ubench_costas.hip -- cycles per Costas step of the hand-scheduled stream (qpsk_amd/csrc/costas_asm.h)
// running alone in one wave, and the shader clock it runs at.  Not product code.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I qpsk_amd/csrc tools/ubench_costas.hip -o build_tools/ubench_costas
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "costas_asm.h"

using namespace qpsk;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k(const float2 *din, int nsym, float alpha, float beta, unsigned long long *out, float *state, int lanes, int usecpp)
{
    __shared__ __attribute__((aligned(16))) float2 d[64 * 66];      // [lane][64+2]
    __shared__ __attribute__((aligned(16))) float4 z[64 * 65];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 66; i += blockDim.x) d[i] = din[i % 4096];
    __syncthreads();
    float ph = 0.1f * lane, fr = 0.13f;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    if (lane < lanes) {
        for (int c = 0; c < nsym / 64; c++) {
            if (!usecpp) {
                unsigned da = lds_addr(d + lane * 66), za = lds_addr(z + lane * 65);
                unsigned long long fl;
                unsigned left = 64 / COSTAS_ASM_GROUP;
                while (left) {
                    left = costas_asm_run(ph, fr, da, za, __builtin_amdgcn_readfirstlane(left), alpha, beta, -1.0f, 1.0f, fl);
                    if (left) { da += 8 * COSTAS_ASM_GROUP; za += 16 * COSTAS_ASM_GROUP; left--; }   // skip a flagged group (timing only)
                }
            } else {
                for (int j = 0; j < 64; j++) {
                    float tx, ty; unsigned qq;
                    costas_step_t<true>(ph, fr, alpha, beta, -1.0f, 1.0f, d[lane * 66 + j], tx, ty, qq);
                    z[lane * 65 + j] = make_float4(tx, ty, __uint_as_float(qq), 0.0f);
                }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    if (lane == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    state[lane] = ph + fr + z[lane * 65].x;
}

int main()
{
    std::vector<float2> h(4096);
    for (int i = 0; i < 4096; i++) { float a = 0.7853981f + 1.5707963f * (rand() & 3) + 0.1f * ((rand() % 100) / 100.0f - 0.5f); h[i] = make_float2(cosf(a), sinf(a)); }
    float2 *d; unsigned long long *o; float *st;
    CHECK(hipMalloc(&d, 4096 * 8)); CHECK(hipMalloc(&o, 16)); CHECK(hipMalloc(&st, 256));
    CHECK(hipMemcpy(d, h.data(), 4096 * 8, hipMemcpyHostToDevice));
    const int nsym = 64 * 256;
    for (int usecpp = 0; usecpp < 2; usecpp++)
        for (int lanes : {16, 64}) {
            for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, nsym, 0.1626f, 0.01445f, o, st, lanes, usecpp);
            CHECK(hipDeviceSynchronize());
            unsigned long long ho[2]; CHECK(hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost));
            printf("%s lanes=%2d: %.1f cycles/step, shader clock %.0f MHz\n", usecpp ? "C++ step " : "asm stream", lanes, (double)ho[0] / nsym, (double)ho[0] / ((double)ho[1] / 100.0));
        }
    return 0;
}
