// pf_profile.hip -- phase timing of azul_policy_forward_kernel (s_memtime deltas summed over workgroups) + launch time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAZ_PF_PROFILE -I include -I azul_deep_reinforcement_learning_amd/csrc tools/pf_profile.hip -o gpurun_out/pf_profile
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64; typedef int32_t i32;
#define AZUL_NUM_ACTIONS 180
#include "azul_policy.hpp"

__global__ void calib_kernel(u64 ticks, u64 *out)
{
    u64 t0 = __builtin_amdgcn_s_memtime(), t1 = t0;
    while (t1 - t0 < ticks) t1 = __builtin_amdgcn_s_memtime();
    out[0] = t1 - t0;
}

int main(int argc, char **argv)
{
    {
        u64 *o; hipMalloc(&o, 8);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(calib_kernel, dim3(1), dim3(64), 0, 0, 1000ull, o); hipDeviceSynchronize();
        hipEventRecord(a); hipLaunchKernelGGL(calib_kernel, dim3(1), dim3(64), 0, 0, 1000000ull, o); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("s_memtime: 1e6 ticks = %.1f us  => %.2f MHz\n", ms * 1e3, 1e6 / (ms * 1e3));
    }
  for (int ai = 1; ai < (argc > 1 ? argc : 2); ai++) {
    int n = argc > 1 ? atoi(argv[ai]) : 4096, reps = 200;
    std::vector<float> h(136 * 360 + 360 + 180 + 1 + 180 * 180 + 180);
    for (auto &x : h) x = (float)(rand() % 2001 - 1000) / 5000.f;
    float *w; hipMalloc(&w, h.size() * 4); hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    PolicyWeights W = {w, w + 136 * 360, w + 136 * 360 + 360, w + 136 * 360 + 540, w + 136 * 360 + 541 + 3 /*keep 8B alignment irrelevant*/, w + 136 * 360 + 544 + 180 * 180};
    std::vector<float> ho((size_t)n * 136); for (auto &x : ho) x = (float)(rand() % 5);
    std::vector<uint8_t> hm((size_t)n * 180); for (auto &x : hm) x = rand() % 6 == 0; for (int g = 0; g < n; g++) hm[(size_t)g * 180 + g % 180] = 1;
    float *obs, *value, *logp, *ent; uint8_t *mask; i32 *action; u64 *ctr;
    hipMalloc(&obs, ho.size() * 4); hipMemcpy(obs, ho.data(), ho.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&mask, hm.size()); hipMemcpy(mask, hm.data(), hm.size(), hipMemcpyHostToDevice);
    hipMalloc(&value, n * 4); hipMalloc(&logp, n * 4); hipMalloc(&ent, n * 4 + 64); hipMalloc(&action, n * 4); hipMalloc(&ctr, 16);
    hipMemset(ctr, 0, 16); hipMemset(ent, 0, n * 4 + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((n + 15) / 16), block(256);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(azul_policy_forward_kernel, grid, block, 0, 0, obs, mask, W, 1ull, 0ull, ctr, 1, (u32)n, value, action, logp, ent, (float *)nullptr);
    hipDeviceSynchronize();
    hipMemset((char *)ent + n * 4, 0, 64);
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(azul_policy_forward_kernel, grid, block, 0, 0, obs, mask, W, 1ull, 0ull, ctr, 1, (u32)n, value, action, logp, ent, (float *)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    u64 t[8]; hipMemcpy(t, (char *)ent + n * 4, 64, hipMemcpyDeviceToHost);
    printf("n=%d  %.2f us per launch (back-to-back)\n", n, ms * 1e3 / reps);
    const char *names[] = {"obs->LDS", "loads+layer1", "layer2+critic", "head"};
    for (int i = 0; i < 4; i++) printf("  %-14s %8.2f x100 shader cycles per workgroup\n", names[i], (double)t[i] / reps / grid.x / 100.0);
  }
    return 0;
}
