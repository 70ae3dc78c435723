// TEST-ONLY stand-in for <hip/hip_runtime.h> when the product's device headers are compiled by g++ for the lockstep wave
// emulation (tests/hostcheck/simt/simt.hpp): HIP's device qualifiers become host spellings, the gfx950 builtins the headers
// call become the emulator's functions.  Only what csrc/azul_common.hpp, azul_tables.hpp, azul_selfplay2.hpp, azul_env2.hpp,
// azul_rules_x.hpp, azul_ops2.hpp, azul_policy.hpp, azul_rollout2.hpp and azul_learner.hpp use.
#pragma once
#include <math.h>
#include <string.h>
#include "../simt.hpp"

#define __device__ static
#define __forceinline__ inline
#define __host__

struct double2 { double x, y; };
static inline double2 make_double2(double x, double y) { double2 r = {x, y}; return r; }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline long long __double_as_longlong(double d) { long long r; memcpy(&r, &d, 8); return r; }
static inline double __longlong_as_double(long long v) { double r; memcpy(&r, &v, 8); return r; }

#define __builtin_amdgcn_ballot_w64(p) simt::ballot(p)
#define __builtin_amdgcn_readlane(v, l) simt::readlane((int)(v), (int)(l))
#define __builtin_amdgcn_readfirstlane(v) simt::readfirstlane((int)(v))
#define __builtin_amdgcn_ds_bpermute(a, v) simt::ds_bpermute((int)(a), (int)(v))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) simt::update_dpp((int)(old), (int)(src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) simt::permlane16_swap((a), (b), (fi), (bc))
#define __builtin_amdgcn_mbcnt_lo(m, b) simt::mbcnt_lo((m), (b))
#define __builtin_amdgcn_mbcnt_hi(m, b) simt::mbcnt_hi((m), (b))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() simt::wave_barrier()
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_s_memtime() 0ull

// ---- additions for the matrix-core kernels (csrc/azul_policy.hpp, azul_rollout2.hpp): workgroups of several waves, MFMA, buffer loads ----
#undef __builtin_amdgcn_fence
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_barrier() simt::wg_barrier()
#define __syncthreads() simt::wg_barrier()
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_memrealtime() 0ull
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) simt::mfma_f32_16x16x4f32((a), (b), (c))
#define ext_vector_type(n) vector_size(4 * (n))          /* float __attribute__((ext_vector_type(4))): subscriptable 16-byte vector in g++ too */
typedef simt::Rsrc __amdgpu_buffer_rsrc_t;
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, bytes, flags) (simt::Rsrc{(const char *)(p), (uint32_t)(bytes)})
#define __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, aux) (simt::buffer_load<1>((r), (vo), (so)).v[0])
#define __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, aux) simt::buffer_load<2>((r), (vo), (so))
#define __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, aux) simt::buffer_load<4>((r), (vo), (so))
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline unsigned long long __ballot(bool p) { return simt::ballot(p); }
static inline float __shfl_xor(float v, int lane_mask, int width = 64)          // the value of lane (l ^ lane_mask)
{
    (void)width;
    unsigned u;
    memcpy(&u, &v, 4);
    const unsigned r = (unsigned)simt::ds_bpermute((int)(((simt::lane_id() ^ (unsigned)lane_mask) & 63u) << 2), (int)u);
    float f;
    memcpy(&f, &r, 4);
    return f;
}
static inline void __threadfence() {}
template <class T, class V>
static inline T atomicAdd(T *p, V v) { T o = *p; *p = (T)(o + (T)v); return o; }      // (the emulation runs one lane at a time)
#define __global__ static
// LDS arrays become function-local statics in a linker section of their own; run_wave / run_workgroup fill that section with 0xA5 before
// every emulated workgroup (simt::poison_lds), so that a kernel reading LDS before writing it sees garbage like on the hardware -- not the
// zeros, or the previous workgroup's data, a plain static would hand it -- and differs from the oracle.
#define __shared__ static __attribute__((section("simt_lds")))
#define __launch_bounds__(...)
#define __restrict__ __restrict
namespace simt {
struct Dim3 { unsigned x, y, z; };
struct ThreadIdx { struct X { operator unsigned() const { return thread_id(); } } x; };
static Dim3 g_block_idx = {0, 0, 0}, g_grid_dim = {1, 1, 1};
}
#define threadIdx (simt::ThreadIdx())
#define blockIdx simt::g_block_idx
#define gridDim simt::g_grid_dim
