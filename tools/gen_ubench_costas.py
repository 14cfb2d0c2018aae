#!/usr/bin/env python3
"""Generate build_tools/ubench_costas_var.hip: timing variants of the hand-scheduled Costas step stream
(qpsk_amd/csrc/costas_asm.h) to learn what each part costs on gfx950.  Measurement tooling, not product code;
the variants that drop parts compute wrong values on purpose (timing only).

Historical: this models the FIRST hand-scheduled stream (compare/select detector, 8-byte records).  The stream has
since been measured in place instead -- timing-only edits of costas_asm.h built as a second library and alternated
with the product build in one process by tools/ab_libs.py.

    python tools/gen_ubench_costas.py && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off \
        build_tools/ubench_costas_var.hip -o build_tools/ubench_costas_var
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def step(pin, fin, pout, fout, doff, zoff, qoff, i, v):
    L = []
    a = L.append
    a(f"v_cvt_f64_f32 v[200:201], {pin}")
    a("v_fma_f64 v[202:203], v[200:201], %[k2pi], %[magic]")
    a("v_add_f64 v[204:205], v[202:203], -%[magic]")
    a("v_fma_f64 v[206:207], -v[204:205], %[hpi], v[200:201]")
    a("v_mul_f64 v[208:209], v[206:207], v[206:207]")
    a("v_fma_f64 v[210:211], v[208:209], %[c4], %[c3]")
    a("v_fma_f64 v[212:213], v[208:209], %[s3], %[s2]")
    a("v_fma_f64 v[210:211], v[208:209], v[210:211], %[c2]")
    a("v_mul_f64 v[204:205], v[206:207], v[208:209]")
    a("v_fma_f64 v[210:211], v[208:209], v[210:211], %[c1]")
    a("v_fma_f64 v[212:213], v[208:209], v[212:213], %[s1]")
    a("v_fma_f64 v[210:211], v[208:209], v[210:211], 1.0")
    a("v_fma_f64 v[212:213], v[204:205], v[212:213], v[206:207]")
    a("v_cvt_f32_f64 v210, v[210:211]")
    a("v_cvt_f32_f64 v212, v[212:213]")
    if v.get("lds", True):
        a("s_waitcnt lgkmcnt(2)")
    a("v_pk_mul_f32 v[200:201], v[220:221], v[210:211] op_sel_hi:[1,0]")
    a("v_pk_mul_f32 v[204:205], v[220:221], v[212:213] op_sel:[1,0] op_sel_hi:[0,0]")
    if v.get("lds", True):
        a(f"ds_read_b64 v[220:221], %[da] offset:{doff}")
    a("v_pk_add_f32 v[214:215], v[200:201], v[204:205] neg_hi:[0,1]")
    if v.get("det", "sgpr") == "sgpr":
        a("v_cmp_lt_f32_e32 vcc, 0, v214")
        a("v_cmp_lt_f32_e64 %[tm], 0, v215")
        a("v_min3_f32 v226, v226, |v214|, |v215|")
        a("v_cndmask_b32_e64 v208, -v215, v215, vcc")
        a("v_cndmask_b32_e64 v209, -v214, v214, %[tm]")
    else:  # both compares through VCC
        a("v_cmp_lt_f32_e32 vcc, 0, v214")
        a("v_min3_f32 v226, v226, |v214|, |v215|")
        a("s_nop 0")
        a("v_cndmask_b32_e64 v208, -v215, v215, vcc")
        a("v_cmp_lt_f32_e32 vcc, 0, v215")
        a("s_nop 1")
        a("v_cndmask_b32_e64 v209, -v214, v214, vcc")
    a("v_sub_f32_e32 v206, v208, v209")
    a("v_pk_mul_f32 v[216:217], %[beal], v[206:207] op_sel_hi:[1,0]")
    a(f"v_add_f32_e32 v218, {fin}, v216")
    if v.get("lds", True):
        a(f"ds_write_b64 %[za], v[214:215] offset:{zoff}")
    a(f"v_add_f32_e32 v219, {pin}, v218")
    if v.get("lds", True):
        a(f"ds_write_b8 %[qa], v202 offset:{qoff}")
    a(f"v_add_f32_e32 {pout}, v219, v217")
    a(f"v_med3_f32 {fout}, v218, %[fmin], %[fmax]")
    w = v.get("wrap", "inline")
    wrapfix = [f"v_cvt_f64_f32 v[200:201], {pout}",
               f"v_bfi_b32 v229, %[absm], v227, {pout}",
               "v_add_f64 v[200:201], v[200:201], -v[228:229]",
               "v_cvt_f32_f64 v204, v[200:201]",
               f"v_cndmask_b32_e32 {pout}, {pout}, v204, vcc",
               f"v_cmp_ge_f32_e64 vcc, |{pout}|, %[tau]",
               "s_or_b64 %[fl], %[fl], vcc"]
    tail = []
    if w == "inline":
        a(f"v_cmp_ge_f32_e64 vcc, |{pout}|, %[tau]")
        a("s_cbranch_vccz 1f")
        L += wrapfix
        a("1:")
    elif w == "ool":
        a(f"v_cmp_ge_f32_e64 vcc, |{pout}|, %[tau]")
        a(f"s_cbranch_vccnz 1{i}f")
        a(f"2{i}:")
        tail = [f"1{i}:"] + wrapfix + [f"s_branch 2{i}b"]
    elif w == "branchless":
        a(f"v_cmp_ge_f32_e64 vcc, |{pout}|, %[tau]")
        L += wrapfix[:5]
    elif w == "none":
        pass
    return L, tail


def run_fn(name, v):
    regs = [("%[p]", "%[f]", "v224", "v225"), ("v224", "v225", "v222", "v223"), ("v222", "v223", "v224", "v225"),
            ("v224", "v225", "v222", "v223"), ("v222", "v223", "v224", "v225"), ("v224", "v225", "v222", "v223"),
            ("v222", "v223", "v224", "v225"), ("v224", "v225", "%[p]", "%[f]")]
    body, tails = [], []
    for i, (pi, fi, po, fo) in enumerate(regs):
        b, t = step(pi, fi, po, fo, 8 * (i + 1), 8 * i, i, i, v)
        body += b
        tails += t
    lines = ["v_mov_b32 v228, 0x54442d18", "v_mov_b32 v227, 0x401921fb", "ds_read_b64 v[220:221], %[da]",
             "s_mov_b64 %[fl], 0", "s_waitcnt lgkmcnt(0)", "2:", "v_mov_b32 v230, %[p]", "v_mov_b32 v231, %[f]",
             "v_mov_b32 v226, 0x7f800000"] + body + [
        "v_cmp_eq_f32_e64 %[tm], 0, v226", "s_or_b64 %[fl], %[fl], %[tm]",
        "v_add_u32_e32 %[da], 64, %[da]", "v_add_u32_e32 %[za], 64, %[za]", "v_add_u32_e32 %[qa], 8, %[qa]",
        "s_sub_u32 %[ng], %[ng], 1", "s_cmp_lg_u32 %[ng], 0", "s_cbranch_scc1 2b", "s_branch 4f"] + tails + [
        "4:", "s_waitcnt lgkmcnt(0)"]
    txt = "\n".join('        "%s\\n\\t"' % ln for ln in lines)
    return f"""
__device__ __forceinline__ void run_{name}(float &phase, float &freq, unsigned &d_addr, unsigned &z_addr, unsigned &q_addr,
                                           unsigned groups, float alpha, float beta, float min_freq, float max_freq)
{{
    unsigned long long flags, tmp;
    const double magic = 0x1.8p52, c3 = -0x1.6c087e89a359dp-10, s2 = 0x1.1107605230bc4p-7;
    double beal; {{ const float2 ba = make_float2(beta, alpha); __builtin_memcpy(&beal, &ba, 8); }}
    asm volatile(
{txt}
        : [p] "+v"(phase), [f] "+v"(freq), [da] "+v"(d_addr), [za] "+v"(z_addr), [qa] "+v"(q_addr),
          [ng] "+s"(groups), [fl] "=&s"(flags), [tm] "=&s"(tmp)
        : [magic] "v"(magic), [c3] "v"(c3), [s2] "v"(s2), [fmax] "v"(max_freq), [beal] "v"(beal),
          [k2pi] "s"(0x1.45F306DC9C883p-1), [hpi] "s"(0x1.921FB54442D18p0), [c4] "s"(0x1.99343027bf8c3p-16),
          [s3] "s"(-0x1.994eb3774cf24p-13), [c2] "s"(0x1.55553e1068f19p-5), [s1] "s"(-0x1.555545995a603p-3),
          [c1] "s"(-0x1.ffffffd0c621cp-2), [fmin] "s"(min_freq), [tau] "s"(0x1.921fb6p+2f), [absm] "s"(0x7fffffffu)
        : "vcc", "scc", "memory", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209",
          "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222",
          "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231");
}}
"""


VARIANTS = [("inline", dict(wrap="inline")), ("ool", dict(wrap="ool")), ("branchless", dict(wrap="branchless")),
            ("nowrap", dict(wrap="none")), ("ool_nolds", dict(wrap="ool", lds=False)),
            ("ool_vccdet", dict(wrap="ool", det="vcc")), ("nowrap_nolds", dict(wrap="none", lds=False))]

src = """// GENERATED by tools/gen_ubench_costas.py -- timing variants of the Costas step stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(__UINTPTR_TYPE__)(const __attribute__((address_space(3))) void *)p; }
"""
for n, v in VARIANTS:
    src += run_fn(n, v)
src += """
template <int V>
__global__ void k(const float2 *din, int nsym, float alpha, float beta, unsigned long long *out, float *state, int lanes, float f0)
{
    __shared__ float2 d[64 * 65];
    __shared__ float2 z[64 * 65];
    __shared__ unsigned char q[64 * 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 65; i += blockDim.x) d[i] = din[i % 4096];
    __syncthreads();
    float ph = 0.1f * lane, fr = f0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (lane < lanes) {
        for (int c = 0; c < nsym / 64; c++) {
            unsigned da = lds_addr(d + lane * 65), za = lds_addr(z + lane * 65), qa = lds_addr(q + lane * 64);
"""
for i, (n, v) in enumerate(VARIANTS):
    src += f"            if (V == {i}) run_{n}(ph, fr, da, za, qa, 8, alpha, beta, -1.0f, 1.0f);\n"
src += """        }
    }
    asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[0] = t1 - t0;
    state[lane] = ph + fr + z[lane * 65].x;
}

template <int V> void go(const char *name, const float2 *d, unsigned long long *o, float *st)
{
    const int nsym = 64 * 256;
    for (float f0 : {0.13f, 0.0f})
        for (int lanes : {16, 64}) {
            for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, d, nsym, 0.1626f, 0.01445f, o, st, lanes, f0);
            CHECK(hipDeviceSynchronize());
            unsigned long long ho; CHECK(hipMemcpy(&ho, o, 8, hipMemcpyDeviceToHost));
            printf("%-14s lanes=%2d freq0=%.2f: %.1f cycles/step\\n", name, lanes, f0, (double)ho / nsym);
        }
}

int main()
{
    std::vector<float2> h(4096);
    for (int i = 0; i < 4096; i++) { float a = 0.7853981f + 1.5707963f * (rand() & 3) + 0.1f * ((rand() % 100) / 100.0f - 0.5f); h[i] = make_float2(cosf(a), sinf(a)); }
    float2 *d; unsigned long long *o; float *st;
    CHECK(hipMalloc(&d, 4096 * 8)); CHECK(hipMalloc(&o, 16)); CHECK(hipMalloc(&st, 256));
    CHECK(hipMemcpy(d, h.data(), 4096 * 8, hipMemcpyHostToDevice));
"""
for i, (n, v) in enumerate(VARIANTS):
    src += f'    go<{i}>("{n}", d, o, st);\n'
src += "    return 0;\n}\n"
os.makedirs(os.path.join(ROOT, "build_tools"), exist_ok=True)
open(os.path.join(ROOT, "build_tools", "ubench_costas_var.hip"), "w").write(src)
print("wrote build_tools/ubench_costas_var.hip")
