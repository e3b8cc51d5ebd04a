// rl_device.h -- device-side numeric helpers for the gfx950 LambdaMART kernels.
//
// Everything here is exact-by-construction arithmetic that the parity argument in DESIGN.md
// relies on.  Compile with -ffp-contract=off: the reference (Java) never fuses a*b+c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace rl {

typedef __int128 i128;
typedef unsigned __int128 u128;

__device__ __forceinline__ double bits2d(uint64_t u) { return __longlong_as_double((long long)u); }
__device__ __forceinline__ uint64_t d2bits(double d) { return (uint64_t)__double_as_longlong(d); }

// exp(x) for the rho of LambdaMART.java:383.  Java's StrictMath.exp is fdlibm's e_exp and Math.exp must
// stay within 1 ulp of it; this follows the published fdlibm scheme (ln2 hi/lo argument reduction,
// degree-5 rational remainder) so that host oracle and device agree bit for bit.
__device__ inline double exp_fdlibm(double x)
{
    const double LN2_HI = bits2d(0x3fe62e42fee00000ULL), LN2_LO = bits2d(0x3dea39ef35793c76ULL);
    const double INV_LN2 = bits2d(0x3ff71547652b82feULL);
    const double C1 = bits2d(0x3FC555555555553EULL), C2 = bits2d(0xBF66C16C16BEBD93ULL),
                 C3 = bits2d(0x3F11566AAF25DE2CULL), C4 = bits2d(0xBEBBBD41C5D26BF1ULL),
                 C5 = bits2d(0x3E66376972BEA4D0ULL);
    const uint64_t ux = d2bits(x);
    const uint32_t top = (uint32_t)(ux >> 32) & 0x7fffffffu;
    const bool neg = (ux >> 63) != 0;
    if (top >= 0x40862E42u) {
        if (top >= 0x7ff00000u) {
            if ((ux & 0x000fffffffffffffULL) != 0) return x + x;
            return neg ? 0.0 : x;
        }
        if (x > bits2d(0x40862E42FEFA39EFULL)) return bits2d(0x7ff0000000000000ULL);
        if (x < bits2d(0xc0874910D52D3051ULL)) return 0.0;
    }
    double hi = 0.0, lo = 0.0;
    int k = 0;
    if (top > 0x3fd62e42u) {
        if (top < 0x3FF0A2B2u) {
            hi = x - (neg ? -LN2_HI : LN2_HI);
            lo = neg ? -LN2_LO : LN2_LO;
            k = neg ? -1 : 1;
        } else {
            k = (int)(INV_LN2 * x + (neg ? -0.5 : 0.5));
            const double t = (double)k;
            hi = x - t * LN2_HI;
            lo = t * LN2_LO;
        }
        x = hi - lo;
    } else if (top < 0x3e300000u) {
        return 1.0 + x;
    }
    const double t = x * x;
    const double c = x - t * (C1 + t * (C2 + t * (C3 + t * (C4 + t * C5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    if (k >= -1021) return bits2d(d2bits(y) + ((uint64_t)(uint32_t)k << 52));
    y = bits2d(d2bits(y) + ((uint64_t)(uint32_t)(k + 1000) << 52));
    return y * bits2d(0x0170000000000000ULL);
}

// The same function without data-dependent branches on the common range (|x| < 700): the three argument-reduction cases
// of e_exp only differ in k (0, +-1, or round(x / ln2)); hi = x - k*ln2HI and lo = k*ln2LO are the same expressions in all
// of them (k = +-1 and k = 0 multiply exactly), and the two final formulas share one division with a selected
// denominator.  Lanes of a wavefront therefore run ONE instruction stream whatever their arguments are, and two calls can
// be interleaved by the compiler.  Rare arguments (overflow / underflow / subnormal results / NaN) take exp_fdlibm.
__device__ __forceinline__ double exp_fdlibm_bf(double x)
{
    const double LN2_HI = bits2d(0x3fe62e42fee00000ULL), LN2_LO = bits2d(0x3dea39ef35793c76ULL);
    const double INV_LN2 = bits2d(0x3ff71547652b82feULL);
    const double C1 = bits2d(0x3FC555555555553EULL), C2 = bits2d(0xBF66C16C16BEBD93ULL),
                 C3 = bits2d(0x3F11566AAF25DE2CULL), C4 = bits2d(0xBEBBBD41C5D26BF1ULL),
                 C5 = bits2d(0x3E66376972BEA4D0ULL);
    const uint64_t ux = d2bits(x);
    const uint32_t top = (uint32_t)(ux >> 32) & 0x7fffffffu;
    if (top >= 0x4085E000u) return exp_fdlibm(x);                  // |x| >= 700: everything unusual lives here
    const bool neg = (ux >> 63) != 0;
    const bool small = !(top > 0x3fd62e42u);                        // |x| <= 0.5 ln2: k = 0
    const bool mid = top < 0x3FF0A2B2u;                             // |x| < 1.5 ln2: k = +-1
    const int kg = (int)(INV_LN2 * x + (neg ? -0.5 : 0.5));
    const int k = small ? 0 : (mid ? (neg ? -1 : 1) : kg);
    const double t = (double)k;
    const double hi = x - t * LN2_HI;
    const double lo = t * LN2_LO;
    const double xr = hi - lo;                                      // == x when k == 0
    const double tt = xr * xr;
    const double c = xr - tt * (C1 + tt * (C2 + tt * (C3 + tt * (C4 + tt * C5))));
    const double q = (xr * c) / (small ? (c - 2.0) : (2.0 - c));
    const double y = small ? (1.0 - (q - xr)) : (1.0 - ((lo - q) - hi));
    const double r = bits2d(d2bits(y) + ((uint64_t)(uint32_t)k << 52));       // k >= -1021 here (|x| < 700)
    return (top < 0x3e300000u) ? (1.0 + x) : r;                     // |x| < 2^-28
}

// IEEE f64 division n / d as the compiler expands it for gfx9:
//     ds = v_div_scale(d), ns = v_div_scale(n);  r = v_rcp_f64(ds);  two Newton steps r = fma(r, fma(-ds, r, 1), r);
//     q = ns * r;  e = fma(-ds, q, ns);  v_div_fmas(e, r, q);  v_div_fixup
// v_div_scale returns its operand unchanged, v_div_fmas is a plain fma and v_div_fixup passes the quotient through unless an operand is
// zero / subnormal / infinite / NaN, the divisor lies outside [2^-1022, 2^1022), the numerator below 2^-969, or the quotient's exponent is within
// ~2^+-768 of the format's ends.  Where the CALLER rules those cases out, the two functions below are the same instructions on the same values
// -- the same correctly rounded quotient, bit for bit -- without the three instructions that do nothing, and the reciprocal part depends on the
// divisor alone: a divisor shared by many quotients (the ideal DCG of a list, NDCGScorer.java:154) pays it once.
// (A zero numerator gives a zero whose SIGN may differ from IEEE's; the callers' next operation absorbs it.)
__device__ __forceinline__ double rcp_newton2(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}
__device__ __forceinline__ double div_by_rcp(double n, double d, double r2)
{
    const double q = n * r2;
    return fma(fma(-d, q, n), r2, q);
}

// rho = 1.0 / (1 + Math.exp(x)) of LambdaMART.java:383, bit for bit: exp_fdlibm_bf with its division and the outer one through div_by_rcp.
//   inner: (xr * c) / +-(2 - c): |divisor| in [1.6, 2.4]; |xr * c| >= 2^-57 wherever the result is used (|x| >= 2^-28, else 1 + x is returned),
//          and a zero xr * c only changes the sign of a zero that `lo - q` / `q - xr` absorb
//   outer: 1 / (1 + e), 1 + e in [1, 2^1011) for |x| < 700
// rho_fast is STRAIGHT-LINE code (no branch at all), valid for |x| < 700 and harmless garbage beyond: two calls in one basic block are two
// independent dependency chains the scheduler interleaves (an f64 operation can be issued every 4 cycles but its result takes ~8: one chain alone
// leaves half of a wavefront's issue slots empty).  The data-dependent branch to the literal e_exp that exp_fdlibm_bf carries in its first lines
// cut every pair's chain into a basic block of its own.  |x| >= 700 (overflow, underflow, subnormal results, infinities, NaN) is patched
// AFTERWARDS (rho_rare / rho_slow: exp_fdlibm and the compiler's division).
__device__ __forceinline__ bool rho_rare(double x) { return ((uint32_t)(d2bits(x) >> 32) & 0x7fffffffu) >= 0x4085E000u; }
__device__ inline double rho_slow(double x) { return 1.0 / (1 + exp_fdlibm(x)); }
__device__ __forceinline__ double rho_fast(double x)
{
    const double LN2_HI = bits2d(0x3fe62e42fee00000ULL), LN2_LO = bits2d(0x3dea39ef35793c76ULL);
    const double INV_LN2 = bits2d(0x3ff71547652b82feULL);
    const double C1 = bits2d(0x3FC555555555553EULL), C2 = bits2d(0xBF66C16C16BEBD93ULL),
                 C3 = bits2d(0x3F11566AAF25DE2CULL), C4 = bits2d(0xBEBBBD41C5D26BF1ULL),
                 C5 = bits2d(0x3E66376972BEA4D0ULL);
    const uint64_t ux = d2bits(x);
    const uint32_t top = (uint32_t)(ux >> 32) & 0x7fffffffu;
    const bool neg = (ux >> 63) != 0;
    const bool small = !(top > 0x3fd62e42u);                        // |x| <= 0.5 ln2: k = 0
    const bool mid = top < 0x3FF0A2B2u;                             // |x| < 1.5 ln2: k = +-1
    const int kg = (int)(INV_LN2 * x + (neg ? -0.5 : 0.5));
    const int k = small ? 0 : (mid ? (neg ? -1 : 1) : kg);
    const double t = (double)k;
    const double hi = x - t * LN2_HI;
    const double lo = t * LN2_LO;
    const double xr = hi - lo;                                      // == x when k == 0
    const double tt = xr * xr;
    const double c = xr - tt * (C1 + tt * (C2 + tt * (C3 + tt * (C4 + tt * C5))));
    const double dn = small ? (c - 2.0) : (2.0 - c);
    const double q = div_by_rcp(xr * c, dn, rcp_newton2(dn));
    const double y = small ? (1.0 - (q - xr)) : (1.0 - ((lo - q) - hi));
    const double r = bits2d(d2bits(y) + ((uint64_t)(uint32_t)k << 52));       // k >= -1021 here (|x| < 700)
    const double e = (top < 0x3e300000u) ? (1.0 + x) : r;           // |x| < 2^-28
    const double de = 1 + e;
    const double rr = rcp_newton2(de);
    return fma(fma(-de, rr, 1.0), rr, rr);                          // == div_by_rcp(1.0, de, rr): 1.0 * rr is rr
}
// N at once: all chains in one basic block, one patch behind them
template <int N>
__device__ __forceinline__ void rho_fdlibm_n(const double (&x)[N], double (&r)[N])
{
#ifdef RL_LAMBDA_R04        // A/B builds: round 4's form -- one chain per basic block (the branch inside exp_fdlibm_bf), the compiler's divisions
#pragma unroll
    for (int u = 0; u < N; u++) r[u] = 1.0 / (1 + exp_fdlibm_bf(x[u]));
#else
    bool rare = false;
#pragma unroll
    for (int u = 0; u < N; u++) { r[u] = rho_fast(x[u]); rare = rare || rho_rare(x[u]); }
    if (rare) {
#pragma unroll
        for (int u = 0; u < N; u++) if (rho_rare(x[u])) r[u] = rho_slow(x[u]);
    }
#endif
}
__device__ __forceinline__ double rho_fdlibm(double x)
{
    const double xs[1] = {x}; double rs[1];
    rho_fdlibm_n<1>(xs, rs);
    return rs[0];
}
__device__ __forceinline__ void rho_fdlibm2(double xa, double xb, double &ra, double &rb)
{
    const double xs[2] = {xa, xb}; double rs[2];
    rho_fdlibm_n<2>(xs, rs);
    ra = rs[0]; rb = rs[1];
}

// value * 2^-e of a 128-bit fixed-point integer, rounded ONCE to nearest-even.
__device__ inline double fixed_to_double(i128 v, int e)
{
    if (v == 0) return 0.0;
    const bool neg = v < 0;
    u128 a = neg ? (u128)(-v) : (u128)v;
    const uint64_t ahi = (uint64_t)(a >> 64);
    int shift = 0;
    uint64_t top;
    if (ahi == 0) {
        top = (uint64_t)a;
    } else {
        shift = 64 - __clzll((long long)ahi);
        top = (uint64_t)(a >> shift);
        const u128 mask = (((u128)1) << shift) - 1;
        if ((a & mask) != 0) top |= 1ULL;          // sticky bit keeps the single rounding correct
    }
    double d = (double)top;                        // 64 -> 53 bits, round to nearest even
    d = ldexp(d, shift - e);
    return neg ? -d : d;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// total order key of a float (monotone over all non-NaN floats, -0 < +0)
__device__ __forceinline__ uint32_t float_key(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k)
{
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// Seeded stand-in for the unseeded `new Random()` of FeatureHistogram.java:283-287 (feature sampling of Random Forests);
// identical to oracle/rl_oracle.c ro_root_hash / ro_child_hash / ro_feature_key.
__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{   // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ unsigned long long root_hash(unsigned long long seed, int tree) { return mix64(seed ^ mix64((unsigned long long)(unsigned)tree)); }
__host__ __device__ __forceinline__ unsigned long long child_hash(unsigned long long parent, int side) { return mix64(parent + 1u + (unsigned long long)(side != 0)); }
__host__ __device__ __forceinline__ unsigned long long feature_key(unsigned long long h, int f) { return mix64(h ^ ((unsigned long long)(unsigned)(f + 1) * 0xA24BAED4963EE407ULL)); }

// one step of a Java `float s; s += double x;`  (LambdaMART.java:406-407)
__device__ __forceinline__ float float_chain_step(float s, double x) { return (float)((double)s + x); }

}  // namespace rl
