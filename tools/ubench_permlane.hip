// ubench_permlane.hip -- what v_permlane16_swap_b32 / v_permlane32_swap_b32 do on gfx950 when vdst and src are the SAME
// register (the serial wave's "wide" symbol fetch rotates the four 16-lane rows of a register through row 0 with them).
// Prints the row order after each step; not product code.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_permlane.hip -o /tmp/ubench_permlane && /tmp/ubench_permlane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    int v = threadIdx.x;            // lane id: row = v / 16
    int a = v, b = v, c = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %0" : "+v"(a));
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %0" : "+v"(b));
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %0" : "+v"(c));
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b; out[128 + threadIdx.x] = c;
}
int main()
{
    int *d, h[192];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *name[3] = {"permlane16_swap v,v", "permlane32_swap v,v", "16, 32, 16 in turn"};
    for (int t = 0; t < 3; t++) {
        printf("%-22s rows now hold old rows:", name[t]);
        for (int r = 0; r < 4; r++) {
            int ok = 1;
            for (int i = 0; i < 16; i++) ok &= (h[64 * t + 16 * r + i] % 16 == i) && (h[64 * t + 16 * r + i] / 16 == h[64 * t + 16 * r] / 16);
            printf(" %d%s", h[64 * t + 16 * r] / 16, ok ? "" : "(!)");
        }
        printf("\n");
    }
    return 0;
}
