// issue_model.hip -- development microbenchmark (not part of the product): how fast do one-wave workgroups issue scalar
// and vector ALU instructions on gfx950, as a function of waves per SIMD and of the scalar : vector mix?  The self-play
// kernel is issue bound; this measures the exchange rate between its two pipes.
//   hipcc --offload-arch=gfx950 -O3 -o issue_model tools/issue_model.hip && ./issue_model
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

// body: S16 groups of 16 scalar adds (4 independent chains) and V16 groups of 16 vector adds, interleaved group by group
// HALF: only lanes 0..31 stay active (EXEC's upper half is zero): does a wave64 instruction on the 32-wide SIMD then skip its second pass?
template <int S16, int V16, bool DEP, bool HALF = false>
__global__ void __launch_bounds__(64) mix_kernel(unsigned *out, int iters)
{
    if (HALF && threadIdx.x >= 32) return;
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    unsigned v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int g = 0; g < (S16 > V16 ? S16 : V16); g++) {
            if (g < S16) {
                if (DEP) { REP16(asm volatile("s_add_u32 %0, %0, 7" : "+s"(s0) : : "scc");) }
                else { REP4(asm volatile("s_add_u32 %0, %0, 7\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");) }
            }
            if (g < V16) {
                if (DEP) { REP16(asm volatile("v_add_u32 %0, 7, %0" : "+v"(v0));) }
                else { REP4(asm volatile("v_add_u32 %0, 7, %0\n v_add_u32 %1, 5, %1\n v_add_u32 %2, 3, %2\n v_add_u32 %3, 1, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));) }
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = s0 + s1 + s2 + s3 + v0 + v1 + v2 + v3;
}

// fine-grained alternation: one scalar, one vector, ...
__global__ void __launch_bounds__(64) alt_kernel(unsigned *out, int iters)
{
    unsigned s0 = blockIdx.x, s1 = 1, v0 = threadIdx.x, v1 = 1;
    for (int i = 0; i < iters; i++) {
        REP16(asm volatile("s_add_u32 %0, %0, 7\n v_add_u32 %2, 7, %2\n s_add_u32 %1, %1, 5\n v_add_u32 %3, 5, %3" : "+s"(s0), "+s"(s1), "+v"(v0), "+v"(v1) : : "scc");)
    }
    out[blockIdx.x * 64 + threadIdx.x] = s0 + s1 + v0 + v1;
}

// taken branches: a chain of 16 always-taken scalar branches per iteration
__global__ void __launch_bounds__(64) branch_kernel(unsigned *out, int iters)
{
    unsigned s0 = blockIdx.x;
    for (int i = 0; i < iters; i++) {
        REP16(asm volatile("s_cmp_lg_u32 %0, 0xffffff\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n 1:\n s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");)
    }
    out[blockIdx.x * 64 + threadIdx.x] = s0;
}

// readlane / readfirstlane / ballot-style traffic between the pipes
__global__ void __launch_bounds__(64) xpipe_kernel(unsigned *out, int iters)
{
    unsigned v0 = threadIdx.x, s0 = 0;
    for (int i = 0; i < iters; i++) {
        REP16(asm volatile("v_readlane_b32 %0, %1, 3\n v_add_u32 %1, %0, %1" : "+s"(s0), "+v"(v0));)
    }
    out[blockIdx.x * 64 + threadIdx.x] = s0 + v0;
}

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
    hipMalloc(&out, 65536 * 64 * sizeof(unsigned));
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const int iters = 2000;
    printf("clock %d kHz (nominal); cycles below assume it; one-wave workgroups, waves/SIMD = grid / 1024\n", clk_khz);
    printf("%-38s %6s %12s %14s\n", "kernel", "w/SIMD", "ms", "cyc/instr/SIMD");
    const int waves[] = {1, 2, 4, 8};
#define RUN(name, K, ninstr) for (int w : waves) { \
        double ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(1024 * w), dim3(64), 0, 0, out, iters); }); \
        double cyc = ms * 1e-3 * clk_khz * 1e3; \
        printf("%-38s %6d %12.3f %14.3f\n", name, w, ms, cyc / ((double)iters * (ninstr) * w)); }
    RUN("salu x64 independent", (mix_kernel<4, 0, false>), 64)
    RUN("salu x64 dependent", (mix_kernel<4, 0, true>), 64)
    RUN("valu x64 independent", (mix_kernel<0, 4, false>), 64)
    RUN("valu x64 dependent", (mix_kernel<0, 4, true>), 64)
    RUN("valu x64 independent, EXEC = low half", (mix_kernel<0, 4, false, true>), 64)
    RUN("valu x64 dependent, EXEC = low half", (mix_kernel<0, 4, true, true>), 64)
    RUN("salu 64 + valu 64 (groups of 16)", (mix_kernel<4, 4, false>), 128)
    RUN("salu 32 + valu 96", (mix_kernel<2, 6, false>), 128)
    RUN("salu 96 + valu 32", (mix_kernel<6, 2, false>), 128)
    RUN("alternating s,v (64 total)", alt_kernel, 64)
    RUN("16 taken branches (+32 salu)", branch_kernel, 48)
    RUN("readlane+vadd pairs (32 total)", xpipe_kernel, 32)
    hipFree(out);
    return 0;
}
