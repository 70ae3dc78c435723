// TEST-ONLY stand-in for <hip/hip_runtime.h> when the product's device headers are compiled by g++ for the lockstep wave
// emulation (tests/hostcheck/simt/simt.hpp): HIP's device qualifiers become host spellings, the gfx950 builtins the headers
// call become the emulator's functions.  Only what csrc/azul_wave.hpp, azul_core.hpp, azul_core_np.hpp, azul_tables.hpp and
// azul_selfplay2.hpp use.
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
