// pattern_cost.hip -- development microbenchmark (not part of the product): the latency of the instruction PATTERNS the self-play
// kernel is made of, as dependent chains (one wave per workgroup; 1 and 2 waves per SIMD), 256 links per loop iteration so that the
// loop's taken branch disappears in the noise.  Reports cycles per link (a link = the instructions listed).
//   hipcc --offload-arch=gfx950 -O3 -o pattern_cost tools/pattern_cost.hip && ./pattern_cost
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP256(x) REP16(REP16(x))

#define KERNEL(name, decl, body, fin) \
__global__ void __launch_bounds__(64) name(unsigned *out, int iters) { \
    decl; \
    for (int i = 0; i < iters; i++) { REP256(body) } \
    out[blockIdx.x * 64 + threadIdx.x] = fin; }

KERNEL(k_vadd, unsigned v = threadIdx.x, asm volatile("v_add_u32 %0, 7, %0" : "+v"(v));, v)
KERNEL(k_vadd_ind, unsigned v = threadIdx.x; unsigned w = 3; unsigned x = 5; unsigned y = 9,
       asm volatile("v_add_u32 %0, 7, %0\n v_add_u32 %1, 5, %1\n v_add_u32 %2, 3, %2\n v_add_u32 %3, 1, %3" : "+v"(v), "+v"(w), "+v"(x), "+v"(y));, v + w + x + y)
// v_cmp writes an SGPR pair, v_cndmask consumes it (the select pattern of every `cond ? a : b` on half-uniform values)
KERNEL(k_cmp_cnd, unsigned v = threadIdx.x; unsigned long long s = 0, asm volatile("v_cmp_ne_u32_e64 %1, 3, %0\n v_cndmask_b32_e64 %0, %0, 5, %1" : "+v"(v), "=s"(s));, v)
// hb(): compare -> lane mask in VCC -> my half of it through a 64-bit shift by (lane & 32)
KERNEL(k_hb, unsigned v = threadIdx.x; unsigned sh = threadIdx.x & 32u; unsigned long long t = 0,
       asm volatile("v_cmp_ne_u32_e32 vcc, 0, %1\n v_lshrrev_b64 %0, %2, vcc" : "=v"(t) : "v"(v), "v"(sh) : "vcc"); v = (unsigned)t | 1u;, v)
KERNEL(k_shr64, unsigned long long t = threadIdx.x + 0x100000000ull; unsigned sh = 1, asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(t) : "v"(sh)); t |= 0x100000000ull;, (unsigned)t)
KERNEL(k_bcnt, unsigned v = threadIdx.x, asm volatile("v_bcnt_u32_b32 %0, %0, %0" : "+v"(v));, v)
KERNEL(k_ffbl, unsigned v = threadIdx.x | 256u, asm volatile("v_ffbl_b32 %0, %0\n v_or_b32 %0, 0x100, %0" : "+v"(v));, v)
KERNEL(k_mul24, unsigned v = threadIdx.x, asm volatile("v_mul_u32_u24 %0, 3, %0" : "+v"(v));, v)
KERNEL(k_mullo, unsigned v = threadIdx.x, asm volatile("v_mul_lo_u32 %0, %0, 3" : "+v"(v));, v)
KERNEL(k_mulhi, unsigned v = threadIdx.x | 0x80000000u; unsigned c = 0xfffffff0u, asm volatile("v_mul_hi_u32 %0, %0, %1\n v_or_b32 %0, 0x80000000, %0" : "+v"(v) : "v"(c));, v)
KERNEL(k_addf64, double d = threadIdx.x, asm volatile("v_add_f64 %0, %0, 1.0" : "+v"(d));, (unsigned)d)
KERNEL(k_mulf64, double d = 1.0 + threadIdx.x * 1e-9, asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d));, (unsigned)d)
KERNEL(k_cvt_f64, double d = 0; unsigned v = threadIdx.x, asm volatile("v_cvt_f64_u32 %1, %0\n v_cvt_u32_f64 %0, %1" : "+v"(v), "=v"(d));, v)
KERNEL(k_ldexp64, double d = 1.0; int e = 0, asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d) : "v"(e));, (unsigned)d)
KERNEL(k_bperm, unsigned v = threadIdx.x; unsigned a = (threadIdx.x ^ 1u) << 2, asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(a));, v)
KERNEL(k_bperm2, unsigned v = threadIdx.x; unsigned w = 1; unsigned a = (threadIdx.x ^ 1u) << 2,
       asm volatile("ds_bpermute_b32 %0, %2, %0\n ds_bpermute_b32 %1, %2, %1\n s_waitcnt lgkmcnt(0)" : "+v"(v), "+v"(w) : "v"(a));, v + w)
KERNEL(k_ldsread, unsigned v = (threadIdx.x & 15u) << 2, asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(v));, v)
KERNEL(k_dpp, unsigned v = threadIdx.x, asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));, v)
KERNEL(k_readlane, unsigned v = threadIdx.x; unsigned s = 0, asm volatile("v_readlane_b32 %1, %0, 5\n v_add_u32 %0, %1, %0" : "+v"(v), "=s"(s));, v)
KERNEL(k_swap16, unsigned v = threadIdx.x; unsigned w = 7, asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1\n v_add_u32 %0, %0, %1" : "+v"(v), "+v"(w));, v + w)
KERNEL(k_mad64, unsigned long long t = threadIdx.x; unsigned a = 3; unsigned b = 5, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(t) : "v"(a), "v"(b) : "vcc");, (unsigned)t)
KERNEL(k_bitop3, unsigned v = threadIdx.x; unsigned w = 0x55, asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0x80" : "+v"(v) : "v"(w));, v)
// a store per link (same address per half, like the record stores), and the packed mask-row store
KERNEL(k_store, unsigned v = threadIdx.x; unsigned *p = out + 4096 * 64 + blockIdx.x * 2 + (threadIdx.x >> 5), asm volatile("global_store_dword %1, %0, off\n v_add_u32 %0, 1, %0" : "+v"(v) : "v"(p) : "memory");, v)
KERNEL(k_store_x2, unsigned v = threadIdx.x; unsigned long long t = 5; unsigned long long *p = (unsigned long long *)(out + 8192 * 64) + blockIdx.x * 64 + threadIdx.x,
       asm volatile("global_store_dwordx2 %2, %1, off\n v_add_u32 %0, 1, %0" : "+v"(v) : "v"(t), "v"(p) : "memory");, v)
// the scalar side of a wave-uniform test: v_cmp -> vcc -> s_cbranch (not taken)
KERNEL(k_wave_any, unsigned v = threadIdx.x, asm volatile("v_cmp_eq_u32_e32 vcc, 0x7fffffff, %0\n s_cbranch_vccnz 1f\n v_add_u32 %0, 1, %0\n 1:" : "+v"(v) : : "vcc");, v)

template <typename F>
static double time_ms(F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    unsigned *out;
    hipMalloc(&out, (size_t)(8192 * 64 + 4096 * 64 * 2) * sizeof(unsigned) * 2);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const int iters = 200;
    printf("clock %d kHz (nominal); one-wave workgroups; cycles per LINK of a dependent chain, 1 and 2 waves per SIMD (wall time x clock / links)\n", clk_khz);
#define RUN(name, K) { double r[2]; int wi = 0; for (int w : {1, 2}) { \
        double ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(1024 * w), dim3(64), 0, 0, out, iters); }); \
        r[wi++] = ms * 1e-3 * clk_khz * 1e3 / ((double)iters * 256); } \
        printf("%-72s %8.2f %8.2f\n", name, r[0], r[1]); }
    printf("%-72s %8s %8s\n", "link", "1 w/SIMD", "2 w/SIMD");
    RUN("v_add_u32 (dependent)", k_vadd)
    RUN("4 x v_add_u32 (independent)", k_vadd_ind)
    RUN("v_cmp_e64 -> sgpr; v_cndmask(sgpr)", k_cmp_cnd)
    RUN("hb: v_cmp -> vcc; v_lshrrev_b64 v, lane&32, vcc; v_or", k_hb)
    RUN("v_lshrrev_b64; v_or", k_shr64)
    RUN("v_bcnt_u32_b32", k_bcnt)
    RUN("v_ffbl_b32; v_or", k_ffbl)
    RUN("v_mul_u32_u24", k_mul24)
    RUN("v_mul_lo_u32", k_mullo)
    RUN("v_mul_hi_u32; v_or", k_mulhi)
    RUN("v_mad_u64_u32", k_mad64)
    RUN("v_bitop3_b32", k_bitop3)
    RUN("v_add_f64", k_addf64)
    RUN("v_mul_f64", k_mulf64)
    RUN("v_cvt_f64_u32; v_cvt_u32_f64", k_cvt_f64)
    RUN("v_ldexp_f64", k_ldexp64)
    RUN("ds_bpermute_b32; s_waitcnt", k_bperm)
    RUN("2 x ds_bpermute_b32; s_waitcnt", k_bperm2)
    RUN("ds_read_b32; s_waitcnt", k_ldsread)
    RUN("s_nop 1; v_add_u32_dpp row_shr:1", k_dpp)
    RUN("v_readlane_b32 -> sgpr; v_add_u32(sgpr)", k_readlane)
    RUN("s_nop 1; v_permlane16_swap_b32; v_add_u32", k_swap16)
    RUN("global_store_dword (2 addresses per wave); v_add_u32", k_store)
    RUN("global_store_dwordx2 (8 B per lane); v_add_u32", k_store_x2)
    RUN("v_cmp -> vcc; s_cbranch_vccnz (not taken); v_add_u32", k_wave_any)
    hipFree(out);
    return 0;
}
