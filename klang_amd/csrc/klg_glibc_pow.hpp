// klang_amd/csrc/klg_glibc_pow.hpp — glibc 2.35's double `pow(x, y)` (x a positive normal number) and `exp2(x)`, restated: what `pow(10, 2 * (x - 1))` and
// `pow(2, (signal)osc)` of examples/Subtractive/Modular.k:24, 123 are in the pinned build of the reference — std::pow(int, float) promotes to the C library's DOUBLE pow,
// and clang turns pow(2.0, x) into exp2(x) (the reference binary imports `pow` and `exp2`, no `powf`) — the result rounded to float where it becomes a signal again.
//
// Source of the algorithm: sysdeps/ieee754/dbl-64/e_pow.c, e_exp2.c, e_exp_data.c, e_pow_log_data.c (glibc 2.35; Szabolcs Nagy's routines, also in ARM's
// optimized-routines): log(x) = k ln2 + log(c) + log1p(z/c - 1) from a 128-entry table in double-double, exp(y log x) = 2^(k/128) exp(r) from another.  What the
// C library EXECUTES on an x86-64 host with FMA is the ifunc variant `__pow_fma`: the same source compiled with -mfma -mavx2 under GCC's default -ffp-contract=fast, so
// WHICH products are fused into their sums is the compiler's choice.  The sequence below is the one in the pinned libm.so.6 (read off its disassembly: every
// vfmadd / vfmsub is a __builtin_fma here, every vmulsd / vaddsd a plain operation; the build forbids contraction).  `exp2` has no ifunc variant: plain operations only.
// tools/verify_glibc_pow.cpp compiles THIS header for the host and compares with the host's libm: pow(10, (double)f) and exp2((double)f) for all 2^32 floats f, and
// pow(x, y) on 10^9 random (x, y) — bit for bit as doubles.
//
// Host side (klg_graph.hpp): pow_log() of a CONSTANT base gives the (hi, lo) the generated code carries as literals; the device then runs pow_of_log().
#pragma once
#include "klg_glibc_tables.hpp"
#if defined(__HIPCC__) || defined(__HIPCC_RTC__) || defined(__HIP__)
#define KLG_GF __host__ __device__ inline
#else
#define KLG_GF static inline
#endif
namespace klg { namespace glibc {
typedef unsigned long long gu64;
KLG_GF double as_f64(gu64 u) { return __builtin_bit_cast(double, u); }
KLG_GF gu64 as_u64(double d) { return __builtin_bit_cast(gu64, d); }
KLG_GF gu64 exp_head(int i) { static const gu64 H[14] = KLG_GLIBC_EXP_HEAD; return H[i]; }
KLG_GF gu64 exp_tab(unsigned i) { static const gu64 T[256] = KLG_GLIBC_EXP_TAB; return T[i]; }
enum { EXP_INVLN2N = 0, EXP_SHIFT, EXP_NEGLN2HIN, EXP_NEGLN2LON, EXP_C2, EXP_C3, EXP_C4, EXP_C5, EXP2_SHIFT, EXP2_C1, EXP2_C2, EXP2_C3, EXP2_C4, EXP2_C5 };
#define KLG_EXPK(i) as_f64(exp_head(i))

// e_pow.c specialcase(): the result's exponent is outside what `scale` can carry (512 <= |y log x| < 1024)
KLG_GF double pow_exp_special(double tmp, gu64 sbits, gu64 ki) {
	if ((ki & 0x80000000ull) == 0) {                                          // k > 0
		sbits -= 1009ull << 52;
		const double scale = as_f64(sbits);
		return 0x1p1009 * __builtin_fma(scale, tmp, scale);
	}
	sbits += 1022ull << 52;                                                   // k < 0: care in the subnormal range
	const double scale = as_f64(sbits), st = tmp * scale;
	double y = scale + st;
	if (__builtin_fabs(y) < 1.0) {
		const double one = y < 0.0 ? -1.0 : 1.0;
		const double lo = (scale - y) + st, hi = y + one;
		const double lo2 = ((one - hi) + y) + lo;
		y = (lo2 + hi) - one;
		if (y == 0.0) y = as_f64(sbits & 0x8000000000000000ull);
	}
	return 0x1p-1022 * y;
}
// exp_inline(x, xtail) of e_pow.c: exp(x + xtail) (sign_bias 0: the base is positive)
KLG_GF double pow_exp(double x, double xtail) {
	unsigned abstop = (unsigned)(as_u64(x) >> 52) & 0x7ffu;
	if (abstop - 0x3c9u > 0x3eu) {
		if ((int)(abstop - 0x3c9u) < 0) return 1.0 + x;                        // |x| < 2^-54
		if (abstop >= 0x409u) return (as_u64(x) >> 63) ? 0.0 : __builtin_inf();   // |x| >= 1024: under- / overflow
		abstop = 0u;                                                           // 512 <= |x| < 1024: pow_exp_special below
	}
	const double kds = __builtin_fma(x, KLG_EXPK(EXP_INVLN2N), KLG_EXPK(EXP_SHIFT));
	const gu64 ki = as_u64(kds);
	const double kd = kds - KLG_EXPK(EXP_SHIFT);
	double r = __builtin_fma(kd, KLG_EXPK(EXP_NEGLN2HIN), x);
	r = __builtin_fma(kd, KLG_EXPK(EXP_NEGLN2LON), r);
	r = xtail + r;
	const unsigned idx = 2u * (unsigned)(ki & 127u);
	const gu64 sbits = exp_tab(idx + 1u) + (ki << 45);
	const double tail = as_f64(exp_tab(idx));
	const double p23 = __builtin_fma(r, KLG_EXPK(EXP_C3), KLG_EXPK(EXP_C2));
	const double rt = r + tail, r2 = r * r;
	const double p45 = __builtin_fma(r, KLG_EXPK(EXP_C5), KLG_EXPK(EXP_C4));
	const double t1 = __builtin_fma(p23, r2, rt), r4 = r2 * r2;
	const double tmp = __builtin_fma(p45, r4, t1);
	if (abstop == 0u) return pow_exp_special(tmp, sbits, ki);
	const double scale = as_f64(sbits);
	return __builtin_fma(tmp, scale, scale);
}
// x^y given log(x) = hi + lo (pow_log below): x positive and normal, x != 1 handled by the caller's `x_is_one`
KLG_GF double pow_of_log(double x, double y, double hi, double lo) {
	const gu64 ix = as_u64(x), iy = as_u64(y), one = 0x3ff0000000000000ull, inf = 0x7ff0000000000000ull;
	const unsigned topy = (unsigned)(iy >> 52) & 0x7ffu;
	if (topy - 0x3beu > 0x7fu) {                                               // |y| < 2^-65 or |y| >= 2^63 (or zero, inf, nan)
		if (2 * iy - 1 >= 2 * inf - 1) {
			if (2 * iy == 0) return 1.0;
			if (ix == one) return 1.0;
			if (2 * iy > 2 * inf) return x + y;
			if ((2 * ix < 2 * one) == !(iy >> 63)) return 0.0;                   // |x| < 1 and y = +inf, |x| > 1 and y = -inf
			return y * y;
		}
		if (ix == one) return 1.0;
		if (topy < 0x3beu) return ix > one ? 1.0 + y : 1.0 - y;
		return ((ix > one) == !(iy >> 63)) ? __builtin_inf() : 0.0;
	}
	const double ehi = y * hi;
	const double elo = __builtin_fma(y, lo, __builtin_fma(hi, y, -ehi));
	return pow_exp(ehi, elo);
}
// ---- exp2(x), e_exp2.c (no fused operation) ----
KLG_GF double exp2_special(double tmp, gu64 sbits, gu64 ki) {
	if ((ki & 0x80000000ull) == 0) {
		sbits -= 1ull << 52;
		const double scale = as_f64(sbits);
		const double t = tmp * scale + scale;
		return t + t;
	}
	sbits += 1022ull << 52;
	const double scale = as_f64(sbits), st = tmp * scale;
	double y = scale + st;
	if (y < 1.0) {
		const double lo = (scale - y) + st, hi = y + 1.0;
		const double lo2 = ((1.0 - hi) + y) + lo;
		y = (lo2 + hi) - 1.0;
		if (y == 0.0) y = 0.0;
	}
	return 0x1p-1022 * y;
}
KLG_GF double exp2(double x) {
	const gu64 ix = as_u64(x);
	unsigned abstop = (unsigned)(ix >> 52) & 0x7ffu;
	if (abstop - 0x3c9u > 0x3eu) {
		if ((int)(abstop - 0x3c9u) < 0) return 1.0 + x;                        // |x| < 2^-54
		if (abstop >= 0x409u) {
			if (ix == 0xfff0000000000000ull) return 0.0;
			if (abstop == 0x7ffu) return 1.0 + x;
			if (!(ix >> 63)) return __builtin_inf();
			if (ix >= 0xc090cc0000000000ull) return 0.0;                        // x <= -1075
		}
		if (2 * ix > 0x811a000000000000ull) abstop = 0u;                       // |x| > 928: exp2_special below
	}
	const double shift = KLG_EXPK(EXP2_SHIFT);
	const double kds = shift + x;
	const gu64 ki = as_u64(kds);
	const double kd = kds - shift;
	const double r = x - kd;
	const unsigned idx = 2u * (unsigned)(ki & 127u);
	const gu64 sbits = exp_tab(idx + 1u) + (ki << 45);
	const double tail = as_f64(exp_tab(idx));
	const double r2 = r * r;
	const double a = KLG_EXPK(EXP2_C3) * r + KLG_EXPK(EXP2_C2), b = KLG_EXPK(EXP2_C1) * r + tail, c = r * KLG_EXPK(EXP2_C5) + KLG_EXPK(EXP2_C4);
	const double a2 = a * r2, r4 = r2 * r2;
	const double s = a2 + b, c2 = c * r4;
	const double tmp = s + c2;
	if (abstop == 0u) return exp2_special(tmp, sbits, ki);
	const double scale = as_f64(sbits);
	return scale + tmp * scale;
}
#undef KLG_EXPK
// ---- host only (a plain host function: the device passes just parse it): log_inline() of e_pow.c — log(x) = hi + lo for a positive normal x, with the fused operations of __pow_fma ----
inline double pow_log(double x, double* lo_out) {
	static const gu64 H[9] = KLG_GLIBC_POWLOG_HEAD; static const gu64 T[512] = KLG_GLIBC_POWLOG_TAB;
	const double ln2hi = as_f64(H[0]), ln2lo = as_f64(H[1]); double A[7]; for (int i = 0; i < 7; i++) A[i] = as_f64(H[2 + i]);
	const gu64 ix = as_u64(x), tmp = ix - 0x3fe6955500000000ull;
	const int i = (int)((tmp >> 45) & 127u), k = (int)((long long)tmp >> 52);
	const gu64 iz = ix - (tmp & 0xfff0000000000000ull);
	const double z = as_f64(iz), kd = (double)k;
	const double invc = as_f64(T[4 * i]), logc = as_f64(T[4 * i + 2]), logctail = as_f64(T[4 * i + 3]);
	const double t1 = __builtin_fma(kd, ln2hi, logc);
	const double r = __builtin_fma(z, invc, -1.0);
	const double ar = r * A[0];
	const double lo1 = __builtin_fma(kd, ln2lo, logctail);
	const double p12 = __builtin_fma(r, A[2], A[1]), p34 = __builtin_fma(r, A[4], A[3]);
	const double t2 = r + t1, ar2 = r * ar;
	const double ar3 = r * ar2;
	const double lo3 = __builtin_fma(ar, r, -ar2);
	const double lo2 = (t1 - t2) + r;
	const double p56 = __builtin_fma(r, A[6], A[5]);
	const double hi = t2 + ar2;
	const double q = __builtin_fma(p56, ar2, p34);
	const double lo4 = (t2 - hi) + ar2;
	const double p = __builtin_fma(ar2, q, p12);
	double lo = lo1 + lo2; lo = lo + lo3; lo = lo + lo4;
	lo = __builtin_fma(ar3, p, lo);
	const double y = hi + lo;
	*lo_out = (hi - y) + lo;
	return y;
}
inline double pow_pos(double x, double y) { double lo; const double hi = pow_log(x, &lo); return pow_of_log(x, y, hi, lo); }   // (x positive and normal)
} }
