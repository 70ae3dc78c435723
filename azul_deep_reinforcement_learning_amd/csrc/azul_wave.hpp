// azul_wave.hpp -- the wave64 programming model the Azul core is written against.
//
// The core (azul_core.hpp) maps ONE GAME TO ONE 64-LANE WAVEFRONT.  Game state lives in
//   * "uniform" values  (plain u32/u64/i32/double: identical in every lane; hipcc keeps them in
//     SGPRs / scalar ALU where it can), and
//   * "lane" values     (vu32: one record byte per lane, i.e. a 64-entry cell array in ONE VGPR),
// and the rules are expressed with wave primitives: ballot (cell-array -> bitboard), readlane /
// writelane (uniform <-> one cell), bpermute (cell gather), mbcnt (prefix popcount).
//
// This header lowers those primitives to gfx950 builtins: it is device code only, there is no CPU execution
// path in the product.  (tests/hostcheck/azul_wave_host.hpp is a TEST-ONLY lane-by-lane emulation of the same
// names; it defines AZ_WAVE_HPP so that this file is skipped when the core is compiled by g++ for the logic
// cross-check against the oracle.)
#ifndef AZ_WAVE_HPP
#define AZ_WAVE_HPP
#include <stdint.h>

typedef uint32_t u32;
typedef int32_t  i32;
typedef uint64_t u64;
typedef int64_t  i64;

#if !defined(__HIPCC__)
#error "azul_wave.hpp is gfx950 device code: compile with hipcc --offload-arch=gfx950"
#endif
#include <hip/hip_runtime.h>
#define AZ_FN __device__ __forceinline__

namespace wv {
typedef u32    vu32;
typedef bool   vbool;
typedef double vf64;

AZ_FN vu32 lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
AZ_FN u64  ballot(vbool p) { return __builtin_amdgcn_ballot_w64(p); }
AZ_FN u32  readlane(vu32 v, u32 l) { return __builtin_amdgcn_readlane(v, l); }
// (clang exposes no writelane builtin for this target; compare+select is one v_cmp + one v_cndmask)
AZ_FN vu32 writelane(vu32 v, u32 val, u32 l) { return lane() == l ? val : v; }
AZ_FN vu32 bperm(vu32 v, vu32 idx) { return __builtin_amdgcn_ds_bpermute((int)(idx << 2), (int)v); }
AZ_FN vu32 sel(vbool p, vu32 a, vu32 b) { return p ? a : b; }
AZ_FN vu32 splat(u32 x) { return x; }
// number of set bits of a uniform mask strictly below this lane
AZ_FN vu32 mbcnt(u64 m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }
// per-lane IEEE fp64 quotient num/den (den uniform); hipcc emits the correctly rounded
// v_div_scale/v_rcp/v_fma/v_div_fmas/v_div_fixup sequence (no fast-math in this build)
AZ_FN vf64 divlanes(vu32 num, double den) { return (double)num / den; }
// CPython random_random() from two tempered words per lane: (a>>5, b>>6) -> (a*2^26 + b) / 2^53
AZ_FN vf64 mkrandom(vu32 a, vu32 b) { return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0); }
AZ_FN double readlane_d(vf64 v, u32 l)
{
    u64 b = (u64)__double_as_longlong(v);
    u32 lo = __builtin_amdgcn_readlane((u32)b, l), hi = __builtin_amdgcn_readlane((u32)(b >> 32), l);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
AZ_FN u32 popc64(u64 x) { return (u32)__builtin_popcountll(x); }
AZ_FN u32 ctz64(u64 x) { return (u32)__builtin_ctzll(x); }
AZ_FN u32 ctz32(u32 x) { return (u32)__builtin_ctz(x); }
AZ_FN u32 clz32(u32 x) { return (u32)__builtin_clz(x); }
AZ_FN vu32 vmulhi(vu32 a, vu32 b) { return __umulhi(a, b); }
AZ_FN vu32 vctz(vu32 x) { return (u32)__builtin_ctz(x); }      // x != 0
AZ_FN vu32 vclz(vu32 x) { return (u32)__builtin_clz(x); }      // x != 0

// memory (addresses: uniform base + per-lane offset)
AZ_FN vu32 ld_u8(const uint8_t *base, vu32 off, vbool act) { return act ? (u32)base[off] : 0u; }
AZ_FN void st_u8(uint8_t *base, vu32 off, vu32 val, vbool act) { if (act) base[off] = (uint8_t)val; }
AZ_FN vu32 ld_u32(const u32 *base, vu32 off, vbool act) { return act ? base[off] : 0u; }
AZ_FN void st_u32(u32 *base, vu32 off, vu32 val, vbool act) { if (act) base[off] = val; }
AZ_FN void st_f32(float *base, vu32 off, vu32 ival, vbool act) { if (act) base[off] = (float)(i32)ival; }
AZ_FN vf64 ld_f64(const double *base, vu32 off, vbool act) { return act ? base[off] : 0.0; }
// LDS (one private region per wave; a wave's DS operations execute in program order)
AZ_FN vu32 lds_ld(const u32 *lds, vu32 idx, vbool act) { return act ? lds[idx] : 0u; }
AZ_FN void lds_st(u32 *lds, vu32 idx, vu32 val, vbool act) { if (act) lds[idx] = val; }
AZ_FN u32  lds_ldu(const u32 *lds, u32 idx) { return __builtin_amdgcn_readfirstlane(lds[idx]); }
AZ_FN void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
AZ_FN double lds_ldu_f64(const double *lds, u32 idx) { return lds[idx]; }
AZ_FN void lds_st_f64(double *lds, vu32 idx, vf64 v, vbool act) { if (act) lds[idx] = v; }
// run `stmt` on one lane only (uniform values -> memory)
#define AZ_LANE0(stmt) do { if (wv::lane() == 0) { stmt; } } while (0)
// store a uniform value from EVERY lane (same address, same data: one coalesced request, no exec masking, no branch)
#ifdef AZ_STU_LANE0
AZ_FN void stu_i32(i32 *p, i32 v) { if (lane() == 0) *p = v; }
AZ_FN void stu_u8(uint8_t *p, u32 v) { if (lane() == 0) *p = (uint8_t)v; }
AZ_FN void stu_u64(u64 *p, u64 v) { if (lane() == 0) *p = v; }
#else
AZ_FN void stu_i32(i32 *p, i32 v) { *p = v; }
AZ_FN void stu_u8(uint8_t *p, u32 v) { *p = (uint8_t)v; }
AZ_FN void stu_u64(u64 *p, u64 v) { *p = v; }
#endif
AZ_FN vf64 self64(bool c, vf64 a, vf64 b) { return c ? a : b; }
AZ_FN vu32 selu(bool c, vu32 a, vu32 b) { return c ? a : b; }     // select between lane values on a UNIFORM condition
AZ_FN vu32 vmin(vu32 a, u32 b) { return a < b ? a : b; }
// per-lane pointers (a VGPR pair on the device): trajectory streams advance with ONE vector add per step and are
// written by all lanes (lanes without a value of their own repeat a neighbour's address AND data)
typedef u64 vptr;
AZ_FN vptr vptr_splat(const void *p) { return (u64)p; }
AZ_FN vptr vptr_sel(vbool c, vptr a, vptr b) { return c ? a : b; }
AZ_FN vptr vptr_off(vptr p, vu32 bytes) { return p + bytes; }
AZ_FN vptr vptr_add(vptr p, u64 bytes) { return p + bytes; }
// (explicit global address space: a plain integer->pointer cast would compile to slower flat_store instructions)
#define AZ_GLOBAL __attribute__((address_space(1)))
AZ_FN void vst_u32(vptr p, vu32 v) { *(AZ_GLOBAL u32 *)p = v; }
AZ_FN void vst_u8(vptr p, vu32 v) { *(AZ_GLOBAL uint8_t *)p = (uint8_t)v; }
AZ_FN void vst_u8_at(vptr p, u32 imm, vu32 v) { ((AZ_GLOBAL uint8_t *)p)[imm] = (uint8_t)v; }
AZ_FN void vst_u64(vptr p, vu32 lo, vu32 hi) { *(AZ_GLOBAL u64 *)p = ((u64)hi << 32) | lo; }
#define AZ_UNLIKELY(x) __builtin_expect(!!(x), 0)
} // namespace wv

#endif // AZ_WAVE_HPP
