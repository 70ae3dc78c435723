// azul_wave_host.hpp -- TEST-ONLY 64-lane host emulation of csrc/azul_wave.hpp (same names, lane by lane).
// Included by tests/hostcheck/hostcheck.cpp BEFORE the core so that the device header is skipped (AZ_WAVE_HPP):
// the wave-level game logic can then be diffed against the oracle in the build container (no GPU).
// Not part of the product; never compiled into libazulhip.so.
#ifndef AZ_WAVE_HPP
#define AZ_WAVE_HPP
#include <stdint.h>

typedef uint32_t u32;
typedef int32_t  i32;
typedef uint64_t u64;
typedef int64_t  i64;

#include <string.h>
#define AZ_FN static inline

namespace wv {
struct vbool { bool v[64]; };
struct vu32 {
    u32 v[64];
};
struct vf64 { double v[64]; };

AZ_FN vu32 splat(u32 x) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = x; return r; }
AZ_FN vu32 lane() { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = (u32)i; return r; }

#define AZ_VOP(op) \
    AZ_FN vu32 operator op(const vu32 &a, const vu32 &b) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] op b.v[i]; return r; } \
    AZ_FN vu32 operator op(const vu32 &a, u32 b) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] op b; return r; } \
    AZ_FN vu32 operator op(u32 a, const vu32 &b) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = a op b.v[i]; return r; }
AZ_VOP(+) AZ_VOP(-) AZ_VOP(*) AZ_VOP(&) AZ_VOP(|) AZ_VOP(^) AZ_VOP(<<) AZ_VOP(>>) AZ_VOP(/) AZ_VOP(%)
#undef AZ_VOP
#define AZ_VCMP(op) \
    AZ_FN vbool operator op(const vu32 &a, const vu32 &b) { vbool r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] op b.v[i]; return r; } \
    AZ_FN vbool operator op(const vu32 &a, u32 b) { vbool r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] op b; return r; }
AZ_VCMP(==) AZ_VCMP(!=) AZ_VCMP(<) AZ_VCMP(<=) AZ_VCMP(>) AZ_VCMP(>=)
#undef AZ_VCMP
AZ_FN vbool operator&(const vbool &a, const vbool &b) { vbool r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] && b.v[i]; return r; }
AZ_FN vbool operator|(const vbool &a, const vbool &b) { vbool r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] || b.v[i]; return r; }
AZ_FN vbool operator!(const vbool &a) { vbool r; for (int i = 0; i < 64; i++) r.v[i] = !a.v[i]; return r; }
AZ_FN vbool operator&(const vbool &a, bool b) { vbool r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] && b; return r; }

AZ_FN u64  ballot(const vbool &p) { u64 m = 0; for (int i = 0; i < 64; i++) if (p.v[i]) m |= 1ull << i; return m; }
AZ_FN u32  readlane(const vu32 &v, u32 l) { return v.v[l & 63]; }
AZ_FN vu32 writelane(vu32 v, u32 val, u32 l) { v.v[l & 63] = val; return v; }
AZ_FN vu32 bperm(const vu32 &v, const vu32 &idx) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = v.v[idx.v[i] & 63]; return r; }
AZ_FN vu32 sel(const vbool &p, const vu32 &a, const vu32 &b) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = p.v[i] ? a.v[i] : b.v[i]; return r; }
AZ_FN vu32 sel(const vbool &p, u32 a, const vu32 &b) { return sel(p, splat(a), b); }
AZ_FN vu32 sel(const vbool &p, const vu32 &a, u32 b) { return sel(p, a, splat(b)); }
AZ_FN vu32 sel(const vbool &p, u32 a, u32 b) { return sel(p, splat(a), splat(b)); }
AZ_FN vu32 mbcnt(u64 m) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = (u32)__builtin_popcountll(m & ((1ull << i) - 1)); return r; }
AZ_FN vf64 divlanes(const vu32 &num, double den) { vf64 r; for (int i = 0; i < 64; i++) r.v[i] = (double)num.v[i] / den; return r; }
AZ_FN vf64 mkrandom(const vu32 &a, const vu32 &b)
{
    vf64 r;
    for (int i = 0; i < 64; i++) r.v[i] = ((double)(a.v[i] >> 5) * 67108864.0 + (double)(b.v[i] >> 6)) * (1.0 / 9007199254740992.0);
    return r;
}
AZ_FN double readlane_d(const vf64 &v, u32 l) { return v.v[l & 63]; }
AZ_FN u32 popc64(u64 x) { return (u32)__builtin_popcountll(x); }
AZ_FN u32 ctz64(u64 x) { return (u32)__builtin_ctzll(x); }
AZ_FN u32 ctz32(u32 x) { return (u32)__builtin_ctz(x); }
AZ_FN u32 clz32(u32 x) { return (u32)__builtin_clz(x); }
AZ_FN vu32 vmulhi(const vu32 &a, const vu32 &b) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = (u32)(((u64)a.v[i] * b.v[i]) >> 32); return r; }
AZ_FN vu32 vctz(const vu32 &x) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = x.v[i] ? (u32)__builtin_ctz(x.v[i]) : 32u; return r; }
AZ_FN vu32 vclz(const vu32 &x) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = x.v[i] ? (u32)__builtin_clz(x.v[i]) : 32u; return r; }
AZ_FN vu32 operator~(const vu32 &a) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = ~a.v[i]; return r; }

AZ_FN vu32 ld_u8(const uint8_t *base, const vu32 &off, const vbool &act) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = act.v[i] ? base[off.v[i]] : 0u; return r; }
AZ_FN void st_u8(uint8_t *base, const vu32 &off, const vu32 &val, const vbool &act) { for (int i = 0; i < 64; i++) if (act.v[i]) base[off.v[i]] = (uint8_t)val.v[i]; }
AZ_FN vu32 ld_u32(const u32 *base, const vu32 &off, const vbool &act) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = act.v[i] ? base[off.v[i]] : 0u; return r; }
AZ_FN void st_u32(u32 *base, const vu32 &off, const vu32 &val, const vbool &act) { for (int i = 0; i < 64; i++) if (act.v[i]) base[off.v[i]] = val.v[i]; }
AZ_FN void st_f32(float *base, const vu32 &off, const vu32 &ival, const vbool &act) { for (int i = 0; i < 64; i++) if (act.v[i]) base[off.v[i]] = (float)(i32)ival.v[i]; }
AZ_FN vf64 ld_f64(const double *base, const vu32 &off, const vbool &act) { vf64 r; for (int i = 0; i < 64; i++) r.v[i] = act.v[i] ? base[off.v[i]] : 0.0; return r; }
AZ_FN vu32 lds_ld(const u32 *lds, const vu32 &idx, const vbool &act) { return ld_u32(lds, idx, act); }
AZ_FN void lds_st(u32 *lds, const vu32 &idx, const vu32 &val, const vbool &act)
{
    // all lanes read their operands before any lane writes (SIMD semantics)
    for (int i = 0; i < 64; i++) if (act.v[i]) lds[idx.v[i]] = val.v[i];
}
AZ_FN u32  lds_ldu(const u32 *lds, u32 idx) { return lds[idx]; }
AZ_FN void lds_fence() {}
AZ_FN double lds_ldu_f64(const double *lds, u32 idx) { return lds[idx]; }
AZ_FN void lds_st_f64(double *lds, const vu32 &idx, const vf64 &v, const vbool &act) { for (int i = 0; i < 64; i++) if (act.v[i]) lds[idx.v[i]] = v.v[i]; }
#define AZ_LANE0(stmt) do { stmt; } while (0)
AZ_FN void stu_i32(i32 *p, i32 v) { *p = v; }
AZ_FN void stu_u8(uint8_t *p, u32 v) { *p = (uint8_t)v; }
AZ_FN void stu_u64(u64 *p, u64 v) { *p = v; }
AZ_FN vf64 self64(bool c, const vf64 &a, const vf64 &b) { return c ? a : b; }
AZ_FN vu32 selu(bool c, const vu32 &a, const vu32 &b) { return c ? a : b; }   // select between lane values on a UNIFORM condition
AZ_FN vu32 vmin(const vu32 &a, u32 b) { vu32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] < b ? a.v[i] : b; return r; }
struct vptr { uintptr_t v[64]; };
AZ_FN vptr vptr_splat(const void *p) { vptr r; for (int i = 0; i < 64; i++) r.v[i] = (uintptr_t)p; return r; }
AZ_FN vptr vptr_sel(const vbool &c, const vptr &a, const vptr &b) { vptr r; for (int i = 0; i < 64; i++) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r; }
AZ_FN vptr vptr_off(const vptr &p, const vu32 &bytes) { vptr r; for (int i = 0; i < 64; i++) r.v[i] = p.v[i] + bytes.v[i]; return r; }
AZ_FN vptr vptr_add(const vptr &p, u64 bytes) { vptr r; for (int i = 0; i < 64; i++) r.v[i] = p.v[i] + bytes; return r; }
AZ_FN void vst_u32(const vptr &p, const vu32 &v) { for (int i = 0; i < 64; i++) *(u32 *)p.v[i] = v.v[i]; }
AZ_FN void vst_u8(const vptr &p, const vu32 &v) { for (int i = 0; i < 64; i++) *(uint8_t *)p.v[i] = (uint8_t)v.v[i]; }
AZ_FN void vst_u8_at(const vptr &p, u32 imm, const vu32 &v) { for (int i = 0; i < 64; i++) ((uint8_t *)p.v[i])[imm] = (uint8_t)v.v[i]; }
AZ_FN void vst_u64(const vptr &p, const vu32 &lo, const vu32 &hi) { for (int i = 0; i < 64; i++) *(u64 *)p.v[i] = ((u64)hi.v[i] << 32) | lo.v[i]; }
#define AZ_UNLIKELY(x) __builtin_expect(!!(x), 0)
} // namespace wv

#endif // AZ_WAVE_HPP
