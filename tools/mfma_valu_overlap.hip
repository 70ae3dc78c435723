// mfma_valu_overlap.hip -- development microbenchmark (not part of the product): can a SIMD of gfx950 issue ordinary vector ALU / LDS
// instructions of ONE wave while ANOTHER wave of the same SIMD streams v_mfma_f32_16x16x4_f32?  The policy rollout kernel puts the
// epilogue of a matrix phase (bias + relu + LDS stores) of wave w in the shadow of wave w + 4's matrix loop; the phase stamps say it
// does not fit there (profiles/round6_policy_rollout_phases.txt).  Workgroup of 8 waves: waves w and w + 4 share SIMD w.
//   mode 0: waves 0..3 stream MFMAs (2 independent accumulators), waves 4..7 idle            -> MFMA time alone
//   mode 1: waves 0..3 idle, waves 4..7 run a chain of independent v_add_f32 / v_max_f32      -> VALU time alone
//   mode 2: both at once                                                                      -> overlap or serialisation?
//   mode 3 / 4: the same with ds_write_b32 + ds_read_b32 instead of vector ALU work
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap tools/mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

__global__ void __launch_bounds__(512) overlap_kernel(unsigned long long *out, int mode, int iters)
{
    __shared__ float lds[8 * 64 * 2];
    const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const bool mfma = w < 4u && (mode == 0 || mode == 2 || mode == 4);
    const bool valu = w >= 4u && (mode == 1 || mode == 2);
    const bool ldsw = w >= 4u && (mode == 3 || mode == 4);
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = (float)lane, v0 = 1.f, v1 = 2.f, v2 = 3.f, v3 = 4.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mfma) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 1.0f, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.5f, a1, 0, 0, 0);
            }
        }
    } else if (valu) {
        for (int i = 0; i < iters; i++) {
            REP16(asm volatile("v_add_f32 %0, 1.0, %0\n v_max_f32 %1, %1, %0\n v_add_f32 %2, 1.0, %2\n v_max_f32 %3, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
        }
    } else if (ldsw) {
        float *p = lds + w * 128u + lane;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) { p[0] = v0; v0 += p[64]; }
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0u) out[blockIdx.x * 8u + w] = t1 - t0;
    if (a0[0] + a1[0] + v0 + v1 + v2 + v3 == 123.456f) out[1000] = 1;      // keep the results alive
}

int main()
{
    unsigned long long *d, h[8];
    hipMalloc(&d, 2048 * sizeof(*d));
    const int iters = 200;
    const char *names[] = {"MFMA alone (waves 0..3: 16 x v_mfma_f32_16x16x4_f32 per iteration, 2 accumulators)", "VALU alone (waves 4..7: 64 v_add / v_max per iteration)",
                           "MFMA (waves 0..3) + VALU (waves 4..7) together", "LDS alone (waves 4..7: 16 ds_write + 16 dependent ds_read per iteration)",
                           "MFMA (waves 0..3) + LDS (waves 4..7) together"};
    for (int mode = 0; mode < 5; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(overlap_kernel, dim3(1), dim3(512), 0, 0, d, mode, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-100s cycles per iteration: wave 0 (SIMD 0) %7.1f   wave 4 (SIMD 0) %7.1f\n", names[mode], (double)h[0] / iters, (double)h[4] / iters);
    }
    printf("(16 MFMAs x 32 cycles = 512 cycles of matrix pipe per iteration; 64 full-rate VALU instructions of one wave = 64 x 4 = 256+ cycles)\n");
    hipFree(d);
    return 0;
}
