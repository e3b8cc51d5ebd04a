// rl_wave.h -- wavefront-wide scans and reductions on the DPP data path (gfx950, wave64).
//
// `__shfl_xor` / `__shfl_up` compile to ds_bpermute_b32: every step of a reduction is a round trip through the LDS pipe
// (~100-130 cycles, `s_waitcnt lgkmcnt(0)` after each), so one 6-step wave reduction costs ~0.35 us -- and the growth
// kernels (partition look-back, finish, bookkeeping) are chains of a dozen of them.  A DPP operand is read by the vector
// ALU itself: a step is one v_add / v_mov with two wait states.  The sequence is the GFX9 one (row_shr 1, 2, 4, 8 inside
// the 16-lane rows, row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3): an inclusive scan whose lane 63 holds
// the wave total; totals come back through v_readlane (an SGPR, uniform).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rl_device.h"

namespace rl {

constexpr int kDppShr1 = 0x111, kDppShr2 = 0x112, kDppShr4 = 0x114, kDppShr8 = 0x118, kDppBcast15 = 0x142, kDppBcast31 = 0x143;

// the DPP source lane's value, `idv` where there is none (out of the row / row masked out)
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v, uint32_t idv = 0u)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)idv, (int)v, CTRL, ROWMASK, 0xf, false);
}
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v, uint64_t idv = 0ull)
{
    const uint32_t lo = dpp_u32<CTRL, ROWMASK>((uint32_t)v, (uint32_t)idv), hi = dpp_u32<CTRL, ROWMASK>((uint32_t)(v >> 32), (uint32_t)(idv >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ i128 dpp_i128(i128 v)
{
    const uint64_t lo = dpp_u64<CTRL, ROWMASK>((uint64_t)(u128)v), hi = dpp_u64<CTRL, ROWMASK>((uint64_t)((u128)v >> 64));
    return (i128)(((u128)hi << 64) | (u128)lo);
}

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane)
{
    return ((uint64_t)readlane_u32((uint32_t)(v >> 32), lane) << 32) | readlane_u32((uint32_t)v, lane);
}
__device__ __forceinline__ i128 readlane_i128(i128 v, int lane)
{
    return (i128)(((u128)readlane_u64((uint64_t)((u128)v >> 64), lane) << 64) | (u128)readlane_u64((uint64_t)(u128)v, lane));
}
__device__ __forceinline__ double readlane_f64(double v, int lane) { return bits2d(readlane_u64(d2bits(v), lane)); }

// inclusive scans over the 64 lanes (integer addition: exact, any grouping)
#define RL_WAVE_SCAN_BODY(T, DPP)                                      \
    v += DPP<kDppShr1>(v); v += DPP<kDppShr2>(v); v += DPP<kDppShr4>(v); v += DPP<kDppShr8>(v); \
    v += DPP<kDppBcast15, 0xa>(v); v += DPP<kDppBcast31, 0xc>(v);      \
    return v;
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) { RL_WAVE_SCAN_BODY(uint32_t, dpp_u32) }
__device__ __forceinline__ uint64_t wave_scan_u64(uint64_t v) { RL_WAVE_SCAN_BODY(uint64_t, dpp_u64) }
__device__ __forceinline__ i128 wave_scan_i128(i128 v) { RL_WAVE_SCAN_BODY(i128, dpp_i128) }
#undef RL_WAVE_SCAN_BODY

// f64 inclusive scan (the grouping differs from a Kogge-Stone shuffle scan in the last two steps: only for values whose rounding is free to differ --
// the float chains' prefix GUESSES, rl_chain.inc; adding the +0.0 of a lane without a source is exact)
__device__ __forceinline__ double wave_scan_f64(double x)
{
#define RL_SF(CTRL, RM) x += bits2d(dpp_u64<CTRL, RM>(d2bits(x)));
    RL_SF(kDppShr1, 0xf) RL_SF(kDppShr2, 0xf) RL_SF(kDppShr4, 0xf) RL_SF(kDppShr8, 0xf) RL_SF(kDppBcast15, 0xa) RL_SF(kDppBcast31, 0xc)
#undef RL_SF
    return x;
}

// wave totals, uniform
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return readlane_u32(wave_scan_u32(v), 63); }
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) { return readlane_u64(wave_scan_u64(v), 63); }
__device__ __forceinline__ i128 wave_sum_i128(i128 v) { return readlane_i128(wave_scan_i128(v), 63); }

// maximum over the 64 lanes, uniform (identity 0)
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v)
{
#define RL_MX(CTRL, RM) { const uint64_t o_ = dpp_u64<CTRL, RM>(v); v = o_ > v ? o_ : v; }
    RL_MX(kDppShr1, 0xf) RL_MX(kDppShr2, 0xf) RL_MX(kDppShr4, 0xf) RL_MX(kDppShr8, 0xf) RL_MX(kDppBcast15, 0xa) RL_MX(kDppBcast31, 0xc)
#undef RL_MX
    return readlane_u64(v, 63);
}

// sums over the two 32-lane halves: lane 31 / lane 63 of the scan before the last step
__device__ __forceinline__ uint32_t half_scan_u32(uint32_t v)
{
    v += dpp_u32<kDppShr1>(v); v += dpp_u32<kDppShr2>(v); v += dpp_u32<kDppShr4>(v); v += dpp_u32<kDppShr8>(v);
    v += dpp_u32<kDppBcast15, 0xa>(v);
    return v;
}

// A candidate of a first-maximum search: larger S wins, on equal S the lower index (>= 0) -- the order in which the Java's loops meet them
// (FeatureHistogram.java:236-264).  `c` travels with it (the left count at the candidate).
struct WBest { double S; int t; int c; };
__device__ __forceinline__ WBest wbetter(WBest a, WBest b)
{
    if (b.S > a.S || (b.S == a.S && b.t >= 0 && (a.t < 0 || b.t < a.t))) return b;
    return a;
}
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ WBest dpp_wbest(WBest v)
{
    WBest r;
    r.S = bits2d(dpp_u64<CTRL, ROWMASK>(d2bits(v.S), 0xbff0000000000000ull /* -1.0: no candidate */));
    r.t = (int)dpp_u32<CTRL, ROWMASK>((uint32_t)v.t, 0xffffffffu);
    r.c = (int)dpp_u32<CTRL, ROWMASK>((uint32_t)v.c, 0u);
    return r;
}
// best candidate of the wavefront, uniform (S >= 0 for every real candidate; {-1.0, -1} = none)
__device__ __forceinline__ WBest wave_best(WBest v)
{
    v = wbetter(v, dpp_wbest<kDppShr1>(v)); v = wbetter(v, dpp_wbest<kDppShr2>(v)); v = wbetter(v, dpp_wbest<kDppShr4>(v));
    v = wbetter(v, dpp_wbest<kDppShr8>(v)); v = wbetter(v, dpp_wbest<kDppBcast15, 0xa>(v)); v = wbetter(v, dpp_wbest<kDppBcast31, 0xc>(v));
    WBest r;
    r.S = readlane_f64(v.S, 63); r.t = (int)readlane_u32((uint32_t)v.t, 63); r.c = (int)readlane_u32((uint32_t)v.c, 63);
    return r;
}

}  // namespace rl
