// klang_amd/csrc/klg_device.hpp — hand-written CDNA4 (gfx950) device primitives of the klang hot path.
//
// One GPU lane evaluates one voice; everything here is a __device__ function that keeps its state in
// registers (plain structs passed by reference, fully inlined).  Arithmetic follows the reference's
// fp32 evaluation order operation by operation (citations are file:line into the reference's klang.h,
// v0.7.8) and this translation unit is compiled with -ffp-contract=off: the reference path is only
// bit-stable without FMA contraction.  Divisions and sqrt are IEEE (hipcc default).
#pragma once
#ifndef __HIPCC_RTC__            // hiprtc (generated patches, klg_graph.hpp) has the HIP runtime and the fixed-width integers built in
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
#include "../../include/klang_mi355_records.h"
#include "klg_glibc_pow.hpp"                     // glibc 2.35's double pow (constant base) / exp2, restated: graph OP_FUNC 1, 2

#pragma clang fp contract(off)

namespace klg {

// ---- constants: `constant` klang.h:93-111, pi/root2 227-233, DENORMALISE 90 ----
#define KLG_PI_F      3.14159274101257324f            /* (float)3.14159265358979... */
#define KLG_PI_INV    0.318309873342514038f           /* (float)(1.0/pi) */
#define KLG_TWO_PI    6.28318548202514648f            /* 2 * pi.f */
#define KLG_HALF_PI   1.57079637050628662f            /* pi.f / 2.f */
#define KLG_3HALF_PI  4.71238899230957031f            /* 3.f / 2.f * pi.f */
#define KLG_DENORM    1.175494e-38f
#define KLG_FINTMAX   2147483648.0f

struct SampleRate { float f, w, timeInc; };           // SampleRate klang.h:1593-1604; timeInc = 1.0f / fs (3976)

__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
// (pos >> 9) | (EXP << 23) — the 23 top bits of a 32-bit phase under a float exponent — in ONE v_alignbit_b32: the low word of
// ({EXP, pos} >> 9).  EXP = 0x7F: a float in [1, 2); EXP = 0x80: in [2, 4).
template<uint32_t EXP> __device__ __forceinline__ uint32_t phase_mantissa(uint32_t pos) { return __builtin_amdgcn_alignbit(EXP, pos, 9u); }

// 2-vectors: fp32 multiplies / adds / fmas on them issue as ONE v_pk_* operation whose halves are ordinary IEEE operations (the packed
// two-voices-per-lane primitives of klg_device_x2.hpp, and osm_saw_pair below: two oscillators of one voice)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 as_f2(u2 v) { return __builtin_bit_cast(f2, v); }
__device__ __forceinline__ u2 as_u2(f2 v) { return __builtin_bit_cast(u2, v); }
__device__ __forceinline__ f2 splat(float x) { f2 r = { x, x }; return r; }
template<uint32_t EXP> __device__ __forceinline__ f2 phase_float2(u2 pos) { u2 m = { phase_mantissa<EXP>(pos.x), phase_mantissa<EXP>(pos.y) }; return as_f2(m); }   // see phase_mantissa

// float -> unsigned exactly as the pinned oracle build does it (clang, baseline x86-64: cvttss2si r64,
// truncate to 32 bits; "integer indefinite" = 0 in the low word when out of range / NaN).  Reproduces the
// wrap of negative FM phase offsets (F3, klang.h:4995-4996).
__device__ __forceinline__ uint32_t f2u_wrap(float x) {
	const bool in_range = fabsf(x) < 9223372036854775808.0f;
	return in_range ? (uint32_t)(int64_t)x : 0u;
}

// x / Y for a compile-time constant Y (bit pattern YBITS) in three VALU operations instead of the ~12 of the IEEE division
// expansion: with r = RN(1/Y), q = RN(x*r), e = x - Y*q (exact in an fma), q + e*r rounds to the IEEE quotient.  Only divisors for
// which this was checked against x / Y on ALL 2^32 float bit patterns (tools/verify_div_const.c: 0 mismatches) are accepted;
// zeros, infinities and NaN take q, which is the IEEE result for them.
template<uint32_t YBITS> struct DivConstChecked { static constexpr bool value = YBITS == 0x40e00000u /*7*/ || YBITS == 0x40400000u /*3*/ || YBITS == 0x40a00000u /*5*/
	|| YBITS == 0x41100000u /*9*/ || YBITS == 0x40200000u /*2.5*/; };
template<uint32_t YBITS> __device__ __forceinline__ float div_const(float x) {
	static_assert(DivConstChecked<YBITS>::value, "divisor not in the exhaustively verified set");
	const float y = u2f(YBITS), r = 1.0f / y;
	const float q = x * r;
	const float e = __builtin_fmaf(-y, q, x);
	const float q2 = __builtin_fmaf(e, r, q);
	return __builtin_amdgcn_class(x, 0x198) ? q2 : q;                      // -normal | -subnormal | +subnormal | +normal
}

// ---- Generators::Fast helpers ----
// Fast::Phase::operator=(float radians) klang.h:4993-4998: position = (uint32_t)(int64_t)(radians * FINTMAX / twoPi), FINTMAX = 2^31, with the float ->
// unsigned wrap of F3 (f2u_wrap).  Written here WITHOUT the IEEE division expansion (11 operations), the 64-bit conversion (10) and the scaling by FINTMAX
// (a power of two commutes with every rounding below): with Y = 2 twoPi and R = RN(1 / twoPi) / 2,
//   q = radians * R;  e = fma(-Y, q, radians);  q2 = fma(e, R, q)
// is the IEEE quotient radians / twoPi, halved, wherever the conversion can see it (anything much smaller truncates to 0, anything much larger — infinities, NaN —
// wraps to 0 either way); the position is the low word of trunc(|q2| * 2^32) = cvt_u32(fract(|q2|) * 2^32): the integer part of |q2| only contributes multiples of
// 2^32, v_fract_f32 of a non-negative number is exact, v_cvt_u32_f32 truncates and gives 0 for the NaN that an infinite q2 leaves; negated for a negative q2.
// tools/verify_fast_phase_fract.c compares this composition with f2u_wrap((radians * FINTMAX) / twoPi) on ALL 2^32 floats: 0 mismatches.
// 9 VALU operations (round 5's form through floor / fma: 10, the plain form: 23); as a pair 15 instead of 19.
__device__ __forceinline__ uint32_t cvt_u32_trunc(float x) { uint32_t u; asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(u) : "v"(x)); return u; }   // (the C cast is undefined for NaN; the instruction is not)
__device__ __forceinline__ uint32_t fast_phase(float radians) {
	constexpr float r = 0.5f * (1.0f / KLG_TWO_PI);
	const float q = radians * r;
	const float e = __builtin_fmaf(-2.f * KLG_TWO_PI, q, radians);
	const float q2 = __builtin_fmaf(e, r, q);
	const float f = __builtin_amdgcn_fractf(__builtin_fabsf(q2));
	const uint32_t u = cvt_u32_trunc(f * 4294967296.0f);
	const uint32_t s = (uint32_t)((int32_t)__float_as_uint(q2) >> 31);
	return (u ^ s) - s;
}
// ... of two values at once (packed multiplies / fmas)
__device__ __forceinline__ u2 fast_phase(f2 radians) {
	constexpr float r = 0.5f * (1.0f / KLG_TWO_PI);
	const f2 q = radians * r;
	const f2 e = __builtin_elementwise_fma(splat(-2.f * KLG_TWO_PI), q, radians);
	const f2 q2 = __builtin_elementwise_fma(e, splat(r), q);
	f2 f; f.x = __builtin_amdgcn_fractf(__builtin_fabsf(q2.x)); f.y = __builtin_amdgcn_fractf(__builtin_fabsf(q2.y));
	const f2 g = f * 4294967296.0f;
	u2 u = { cvt_u32_trunc(g.x), cvt_u32_trunc(g.y) };
	const u2 s = __builtin_bit_cast(u2, __builtin_bit_cast(i2, q2) >> 31);
	return (u ^ s) - s;
}
// pos + fast_phase(radians) (an oscillator's position plus a modulation's phase: OSCILLATOR::set(+in), klang.h:4165), associated as ((u ^ s) + pos) - s — integer
// arithmetic modulo 2^32, the same value — so that the exclusive-or and the first addition are ONE v_xad_u32
__device__ __forceinline__ u2 fast_phase_add(u2 pos, f2 radians) {
	constexpr float r = 0.5f * (1.0f / KLG_TWO_PI);
	const f2 q = radians * r;
	const f2 e = __builtin_elementwise_fma(splat(-2.f * KLG_TWO_PI), q, radians);
	const f2 q2 = __builtin_elementwise_fma(e, splat(r), q);
	f2 f; f.x = __builtin_amdgcn_fractf(__builtin_fabsf(q2.x)); f.y = __builtin_amdgcn_fractf(__builtin_fabsf(q2.y));
	const f2 g = f * 4294967296.0f;
	u2 u = { cvt_u32_trunc(g.x), cvt_u32_trunc(g.y) };
	const u2 s = __builtin_bit_cast(u2, __builtin_bit_cast(i2, q2) >> 31);
	u2 t;                                                                  // (written as the instruction: the compiler re-associates the C expression into xor, shift, three-way add)
	asm("v_xad_u32 %0, %1, %2, %3" : "=v"(t.x) : "v"(u.x), "v"(s.x), "v"(pos.x));
	asm("v_xad_u32 %0, %1, %2, %3" : "=v"(t.y) : "v"(u.y), "v"(s.y), "v"(pos.y));
	return t - s;
}
__device__ __forceinline__ float fast_phase_float(uint32_t pos) {        // Fast::Phase::operator float 5004-5007
	return u2f(phase_mantissa<0x7Fu>(pos)) - 1.f;
}
__device__ __forceinline__ float fast_increment_float(int32_t amount) {  // Fast::Increment::operator float 4979-4983
	return u2f((uint32_t)((amount >> 9) | 0x3f800000)) - 1.f;
}
__device__ __forceinline__ float polysin(float x) {                      // klang.h:5093-5096
	const float x2 = x * x;
	return (((-0.00018542f * x2 + 0.0083143f) * x2 - 0.16666f) * x2 + 1.0f) * x;
}
// fastsinp klang.h:5117-5132 (+ fast_modp 1424-1428).  The quadrant fold `if (x > 3pi/2) x -= 2pi; else if (x > pi/2) x = pi - x;` of an
// angle in [0, 2pi) is the MEDIAN of { x, pi - x, x - 2pi } — ordered (x-2pi, x, pi-x) in the first quadrant, (x-2pi, pi-x, x) in the middle
// two, (pi-x, x-2pi, x) in the last — : one v_med3_f32 instead of two compares and two selects (or exec-masked branches), bit for bit on
// all 2^23 phase mantissas (tools/verify_fastsinp_med3.c).
__device__ __forceinline__ float fold_quadrant(float x) { return __builtin_amdgcn_fmed3f(x, KLG_PI_F - x, x - KLG_TWO_PI); }
// (m - 1) * twoPi for the mantissa float m in [1, 2): m - 1 is exact, so the product is ONE rounding of the real number m * twoPi - twoPi — which is what
//  fma(m, twoPi, -twoPi) rounds: the same bits in one operation instead of two (all 2^23 mantissas: tools/verify_fastsinp_med3.c)
__device__ __forceinline__ float phase_radians(uint32_t p) { return __builtin_fmaf(u2f(phase_mantissa<0x7Fu>(p)), KLG_TWO_PI, -KLG_TWO_PI); }
__device__ __forceinline__ float fastsinp(uint32_t p) {
	const float x = phase_radians(p);
	return polysin(fold_quadrant(x));
}
// ... of two phases at once (two voices of a lane, or two consecutive samples of one voice): the polynomial as packed operations
__device__ __forceinline__ f2 polysin(f2 x) { const f2 x2 = x * x; return (((-0.00018542f * x2 + 0.0083143f) * x2 - 0.16666f) * x2 + 1.0f) * x; }
__device__ __forceinline__ f2 fastsinp(u2 p) {
	const f2 x = __builtin_elementwise_fma(phase_float2<0x7Fu>(p), splat(KLG_TWO_PI), splat(-KLG_TWO_PI));   // (see phase_radians)
	const f2 a = KLG_PI_F - x, b = x - KLG_TWO_PI;                     // (packed), then the median per half
	f2 r; r.x = __builtin_amdgcn_fmed3f(x.x, a.x, b.x); r.y = __builtin_amdgcn_fmed3f(x.y, a.y, b.y);
	return polysin(r);
}

// ---- Generators::Fast::Sine klang.h:5135-5172 (lane state: inc, pos; offset only lives inside a sample) ----
// Fast::Increment::set(f) klang.h:4966-4972 on the device (per-sample set(f): vibrato, FM of a recorded graph patch)
__device__ __forceinline__ int32_t fast_increment_set(float f, float fs) {
	const float FC4 = float(261.62556530059862);
	const float FC4_FINTMAX = float(261.62556530059862 * 2147483648.0);
	const float FBASE = FC4_FINTMAX / fs;
	return (int32_t)(2u * (uint32_t)(int32_t)(FBASE / FC4 * f));
}
struct FSine { int32_t inc; uint32_t pos; };
// Fast::Sine::set(frequency) klang.h:5142-5147: only when it differs from the cached Oscillator::frequency
__device__ __forceinline__ void fsine_set_f(FSine& o, float& cached, float f, float fs) { if (f != cached) { cached = f; o.inc = fast_increment_set(f, fs); } }
// set(frequency, phase) klang.h:5149-5153: position = phase (offset = 0: phase modulation is not a node feature), then set(frequency)
__device__ __forceinline__ void fsine_set_fp(FSine& o, float& cached, float f, float phase, float fs) { o.pos = fast_phase(phase); fsine_set_f(o, cached, f, fs); }
__device__ __forceinline__ float fsine_process(FSine& o, uint32_t off) {
	const float y = fastsinp(o.pos + off);
	o.pos += (uint32_t)o.inc;
	return y;
}
// Sine::set(relative phase) klang.h:5160-5162: offset = phase * twoPi -> Fast::Phase
__device__ __forceinline__ uint32_t fsine_rel_offset(float rel) { return fast_phase(rel * KLG_TWO_PI); }
// ... and the oscillator processed with that offset — an FM operator's `OSCILLATOR::set(+in); OSCILLATOR::process()` (klang.h:4165-4166) — with the sign
// of the offset folded into the addition (fast_phase_add: one operation fewer than fsine_process(o, fsine_rel_offset(rel)), the same bits)
__device__ __forceinline__ uint32_t fast_phase_add(uint32_t pos, float radians) {
	constexpr float r = 0.5f * (1.0f / KLG_TWO_PI);
	const float q = radians * r;
	const float e = __builtin_fmaf(-2.f * KLG_TWO_PI, q, radians);
	const float q2 = __builtin_fmaf(e, r, q);
	const float f = __builtin_amdgcn_fractf(__builtin_fabsf(q2));
	const uint32_t u = cvt_u32_trunc(f * 4294967296.0f);
	const uint32_t s = (uint32_t)((int32_t)__float_as_uint(q2) >> 31);
	uint32_t t; asm("v_xad_u32 %0, %1, %2, %3" : "=v"(t) : "v"(u), "v"(s), "v"(pos));
	return t - s;
}
__device__ __forceinline__ float fsine_process_rel(FSine& o, float rel) {
	const float y = fastsinp(fast_phase_add(o.pos, rel * KLG_TWO_PI));
	o.pos += (uint32_t)o.inc;
	return y;
}

// ---- Generic::Oscillator + Generators::Basic klang.h:2849-2880, 4899-4944 ----
struct BOsc { float increment, position, offset; };
__device__ __forceinline__ void phase_advance(float& value, float inc) {  // Phase::operator+=(float) 1518-1525
	// (as selects: the same three operations on the path that is taken — add, compare, subtract — without the two nested branches a lone wave walking a recorded
	//  effect's serial loop paid ~8 scalar instructions and two jumps a sample for)
	const float p1 = value + inc, p2 = (p1 > KLG_TWO_PI) ? p1 - KLG_TWO_PI : p1;
	value = (inc >= KLG_TWO_PI) ? value : p2;
}
// ::sin(double) for the arguments an oscillator has (klang.h:4902: `sin(position + offset)`, a float in [0, 2 pi] plus a phase offset).  The device
// library's sin carries the whole-range argument reduction and was ~3,000 cycles per sample on a chain nothing overlaps (PingPong's LFO with vibrato,
// every Basic::Sine LFO of a recorded effect).  Here: k = rint(x * 2/pi), r = x - k * pi/2 by two fused steps (pi/2 as hi + lo: the first product
// is exact in the fma, so r keeps a relative error of an ulp however close x is to a multiple of pi/2), then fdlibm's kernels for sin / cos on
// [-pi/4, pi/4] (Sun's k_sin.c / k_cos.c polynomials, with the quarter trick of k_cos.c) — under an ulp in double, like glibc's.  What counts is
// the value ROUNDED TO FLOAT: equal to glibc's for every float of [0, 2 pi] (exhaustive, 105 M arguments) and for 2e8 random arguments up to
// +-9,000 (tools/verify_sin_f64.c, on the host: IEEE double is IEEE double).  |x| >= 1e4, NaN: the library's.
__device__ __forceinline__ double sin_f64_core(double x) {                // |x| < 1e4 is the caller's business (straight-line code: several can be in flight side by side)
	const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
	const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
	const double fn = __builtin_rint(x * 6.36619772367581382433e-01);
	double r = __builtin_fma(-fn, 1.57079632679489655800e+00, x);
	r = __builtin_fma(-fn, 6.12323399573676603587e-17, r);
	const int n = (int)fn;
	// odd k: cos(r), even: sin(r).  Both kernels run the same four Horner steps on their coefficients 2..6 — taken per lane, once — and differ in their tails
	const bool odd = (n & 1) != 0;
	const double K2 = odd ? C2 : S2, K3 = odd ? C3 : S3, K4 = odd ? C4 : S4, K5 = odd ? C5 : S5, K6 = odd ? C6 : S6;
	const double z = r * r, v = z * r;
	const double h = K2 + z * (K3 + z * (K4 + z * (K5 + z * K6)));          // k_sin.c's r; k_cos.c's C2 + z * (...)
	const double s = r + v * (S1 + z * h);
	const double rc = z * (C1 + z * h);
	const uint32_t hi = (uint32_t)((unsigned long long)__double_as_longlong(__builtin_fabs(r)) >> 32);
	const double qx = hi < 0x3FD33333u ? 0.0 : hi > 0x3fe90000u ? 0.28125 : __longlong_as_double((long long)((unsigned long long)(hi - 0x00200000u) << 32));
	const double hz = 0.5 * z - qx, a = 1.0 - qx;
	const double c = a - (hz - z * rc);
	const double res = odd ? c : s;
	return (n & 2) ? -res : res;
}
__device__ __forceinline__ double sin_f64(double x) { return (__builtin_fabs(x) < 1.0e4) ? sin_f64_core(x) : sin(x); }
__device__ __forceinline__ float basic_sine(BOsc& o) {                    // double ::sin, rounded
	const float y = (float)sin_f64((double)(o.position + o.offset));
	phase_advance(o.position, o.increment);
	return y;
}
// the two halves of basic_sine for code that runs them in different places (the staged effect kernel of klg_graph_staged.hpp: the phase walk is the
// oscillator's state and stays in sample order, the sine of each sample's argument is taken with samples side by side): same operations, same values
__device__ __forceinline__ float basic_sine_arg(BOsc& o) { const float x = o.position + o.offset; phase_advance(o.position, o.increment); return x; }
__device__ __forceinline__ float basic_sine_of(float x) { return (float)sin_f64((double)x); }
__device__ __forceinline__ float basic_saw(BOsc& o) { const float y = o.position * KLG_PI_INV - 1.f; phase_advance(o.position, o.increment); return y; }
__device__ __forceinline__ float basic_triangle(BOsc& o) { const float y = fabsf(2.f * o.position * KLG_PI_INV - 2.f) - 1.f; phase_advance(o.position, o.increment); return y; }
__device__ __forceinline__ float basic_square(BOsc& o) { const float y = o.position > KLG_PI_F ? 1.f : -1.f; phase_advance(o.position, o.increment); return y; }
__device__ __forceinline__ float basic_pulse(BOsc& o, float duty) { const float y = o.position > (duty * KLG_PI_F) ? 1.f : -1.f; phase_advance(o.position, o.increment); return y; }

// ---- tanh of a float as a patch's plain C function computes it: the C library's DOUBLE tanh, rounded back to float ----
// `float softclip(float x, float c) { return tanh(c * x) / tanh(c); }` (examples/Distortion/Shaping.k:15): unqualified tanh of a float is ::tanh(double) (the pinned
// build imports `tanh`, not `tanhf`).  glibc 2.35 sysdeps/ieee754/dbl-64/s_tanh.c over s_expm1.c (fdlibm): restated here in fp64 without fma; for every float argument the result's
// float rounding equals the host library's (all 2^32 floats: tools/verify_tanh_f64.c) — the double itself is the same sequence of IEEE operations.
__device__ __forceinline__ uint32_t dbl_hi(double x) { return (uint32_t)((unsigned long long)__double_as_longlong(x) >> 32); }
__device__ __forceinline__ double dbl_with_hi(double x, uint32_t h) { return __longlong_as_double((long long)(((unsigned long long)__double_as_longlong(x) & 0xFFFFFFFFull) | ((unsigned long long)h << 32))); }
__device__ inline double glibc_expm1(double x) {
	const double one = 1.0, tiny = 1.0e-300, ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
		Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03, Q3 = -7.93650757867487942473e-05, Q4 = 4.00821782732936239552e-06, Q5 = -2.01099218183624371326e-07;
	double y, hi, lo, c = 0.0, t, e;
	int k;
	uint32_t hx = dbl_hi(x);
	const uint32_t xsb = hx & 0x80000000u;
	hx &= 0x7fffffffu;
	if (hx >= 0x4043687Au) {                                   // |x| >= 56 ln2 (tanh never asks for more than 44: the overflow side is not reached)
		if (hx >= 0x40862E42u) { if (hx >= 0x7ff00000u) return (xsb == 0 || x != x) ? x + x : -1.0; if (x > 7.09782712893383973096e+02) return 1.0e+300 * 1.0e+300; }
		if (xsb != 0) return tiny - one;
	}
	if (hx > 0x3fd62e42u) {                                    // |x| > 0.5 ln2
		if (hx < 0x3FF0A2B2u) { if (xsb == 0) { hi = x - ln2_hi; lo = ln2_lo; k = 1; } else { hi = x + ln2_hi; lo = -ln2_lo; k = -1; } }
		else { k = (int)(invln2 * x + (xsb == 0 ? 0.5 : -0.5)); t = (double)k; hi = x - t * ln2_hi; lo = t * ln2_lo; }
		x = hi - lo; c = (hi - x) - lo;
	}
	else if (hx < 0x3c900000u) return x;
	else k = 0;
	const double hfx = 0.5 * x, hxs = x * hfx;
	const double R1 = one + hxs * Q1, h2 = hxs * hxs, R2 = Q2 + hxs * Q3, h4 = h2 * h2, R3 = Q4 + hxs * Q5;
	const double r1 = R1 + h2 * R2 + h4 * R3;
	t = 3.0 - r1 * hfx;
	e = hxs * ((r1 - t) / (6.0 - x * t));
	if (k == 0) return x - (x * e - hxs);
	e = (x * (e - c) - c); e -= hxs;
	if (k == -1) return 0.5 * (x - e) - 0.5;
	if (k == 1) return (x < -0.25) ? -2.0 * (e - (x + 0.5)) : one + 2.0 * (x - e);
	if (k <= -2 || k > 56) { y = one - (e - x); y = dbl_with_hi(y, dbl_hi(y) + ((uint32_t)k << 20)); return y - one; }
	if (k < 20) { t = dbl_with_hi(one, 0x3ff00000u - (0x200000u >> k)); y = t - (e - x); return dbl_with_hi(y, dbl_hi(y) + ((uint32_t)k << 20)); }
	t = dbl_with_hi(one, (uint32_t)(0x3ff - k) << 20); y = x - (e + t); y += one;
	return dbl_with_hi(y, dbl_hi(y) + ((uint32_t)k << 20));
}
__device__ inline double glibc_tanh(double x) {
	const double one = 1.0, two = 2.0, tiny = 1.0e-300;
	const uint32_t jx = dbl_hi(x), ix = jx & 0x7fffffffu;
	double z;
	if (ix >= 0x7ff00000u) return ((int32_t)jx >= 0) ? one / x + one : one / x - one;
	if (ix < 0x40360000u) {                                    // |x| < 22
		if (x == 0.0) return x;
		if (ix < 0x3c800000u) return x * (one + x);
		if (ix >= 0x3ff00000u) { const double t = glibc_expm1(two * __builtin_fabs(x)); z = one - two / (t + two); }
		else { const double t = glibc_expm1(-two * __builtin_fabs(x)); z = -t / (t + two); }
	}
	else z = one - tiny;
	return ((int32_t)jx >= 0) ? z : -z;
}

// ---- Noise klang.h:4947-4951 (Basic), 5357-5366 (Fast) ----
// The reference draws from libc rand(), one global sequential stream shared by every voice (F5).  The device
// functions are the pure arithmetic applied to a rand() result; the stream itself is produced by klg_rand_fill (klg_rand_dev.hpp)
// (glibc) and injected, which keeps the reference's exact sequence.
__device__ __forceinline__ float basic_noise(int r) { return (float)r * 2.f / 2147483648.0f - 1.f; }   // RAND_MAX = 2^31 - 1 -> (float) 2^31
__device__ __forceinline__ float fast_noise(int r) { return u2f((((uint32_t)r & 0x7FFFu) << 1) | 0x43800000u) - 257.f; }

// ---- Fast::OSM klang.h:5175-5317 ----
// persistent: inc, offset, duty, delta, state(2 bits); the seven coefficients are re-derived per block with the
// reference's own fp32 operations (OSM::init 5206-5215) so they cost registers, not HBM bytes.
struct Osm {
	int32_t inc; uint32_t offset, duty; int state; float delta;
	float f, omf, rcpf, rcpf2, col, c1, c2;
};
__device__ __forceinline__ void osm_derive(Osm& o) {
	o.f = o.delta;
	o.omf = 1.f - o.f;
	o.rcpf = 1.f / o.f;
	o.rcpf2 = 2.f * o.rcpf;
	o.col = fast_phase_float(o.duty);
	o.c1 = 1.f / o.col;
	o.c2 = -1.f / (1.0f - o.col);
}
// OSM::set(frequency) klang.h:5217-5224: increment, delta and init() — which also re-derives the state from the phase
__device__ __forceinline__ void osm_set_f(Osm& o, float& cached, float f, float fs) {
	if (cached != f) {
		cached = f;
		o.inc = fast_increment_set(f, fs);
		o.delta = fast_increment_float(o.inc);
		o.state = ((uint32_t)(o.offset - (uint32_t)o.inc) < o.duty) ? 3 : 0;
		osm_derive(o);
	}
}
// OSM::set(frequency, phase) klang.h:5226-5234: increment and delta only when the frequency changes; the phase and init() always
__device__ __forceinline__ void osm_set_fp(Osm& o, float& cached, float f, float phase, float fs) {
	if (cached != f) { cached = f; o.inc = fast_increment_set(f, fs); o.delta = fast_increment_float(o.inc); }
	o.offset = fast_phase(phase);
	o.state = ((uint32_t)(o.offset - (uint32_t)o.inc) < o.duty) ? 3 : 0;
	osm_derive(o);
}
// OSM::setDuty klang.h:5246-5249: duty = duty * (2 pi) as a phase, then init() (the state from the phase, the coefficients from the duty).
// `set(f, phase, duty)` 5236-5244 is set(f, phase) without its init() followed by this: the second init() recomputes everything the first one did
__device__ __forceinline__ void osm_set_duty(Osm& o, float duty) {
	o.duty = fast_phase(duty * (2.f * KLG_PI_F));
	o.state = ((uint32_t)(o.offset - (uint32_t)o.inc) < o.duty) ? 3 : 0;
	osm_derive(o);
}
__device__ __forceinline__ int osm_tick(Osm& o) {                           // klang.h:5251-5263
	o.state = ((o.state << 1) | (o.offset < o.duty ? 1 : 0)) & 3;
	const int tr = o.state | (o.offset < (uint32_t)o.inc ? 4 : 0);
	o.offset += (uint32_t)o.inc;
	return tr;
}
__device__ __forceinline__ float sqrf(float x) { return x * x; }
// The six-case formula tables of saw()/pulse() (klang.h:5290-5316) are evaluated BRANCH-FREE: with 64 voices at
// 64 different phases some lane of the wave is in a transition state on most samples, so a per-case branch makes
// the wave walk every case anyway and pays the exec-mask bookkeeping on top.  Each case keeps the reference's
// exact fp32 operation order; the unused candidates (which may be inf/NaN when col == 0) are discarded by selects.
__device__ __forceinline__ float osm_saw(Osm& o) {                          // saw() 5290-5302: p evaluated before tick()
	const float p = fast_phase_float(o.offset) - o.col;
	const int tr = osm_tick(o);
	const float f = o.f, omf = o.omf, rcpf = o.rcpf, c1 = o.c1, c2 = o.c2;
	const bool n_up = (tr & 1) != 0, o_up = (tr & 2) != 0, carry = (tr & 4) != 0;
	const float cN = n_up ? c1 : c2;
	const float pp = p + p;
	const float y_lin = cN * (pp - f) + 1.f;                                    // Up (3) / Down (0)
	const float y_wrap = -rcpf * (1.f + cN * omf * (pp + omf)) + 1.f;           // UpDownUp (7) / DownUpDown (4)
	const float p2 = sqrf(p);
	const float y_ud = rcpf * (c2 * p2 - c1 * sqrf(p - f)) + 1.f;               // UpDown (2)
	const float y_du = -rcpf * (1.f + c2 * sqrf(p + omf) - c1 * p2) + 1.f;      // DownUp (5)
	const float y_same = carry ? y_wrap : y_lin;                                // old == new
	const float y_edge = carry ? y_du : y_ud;                                   // old != new: (carry, !o_up, n_up) or (!carry, o_up, !n_up)
	const bool valid_edge = carry ? (!o_up && n_up) : (o_up && !n_up);
	return (o_up == n_up) ? y_same : (valid_edge ? y_edge : 0.f);               // states 1 and 6 "should never happen" -> 0
}
// Saw() : Osm(&OSM::saw, 0.f) whose duty is never changed (patch invariant, e.g. config 2a): duty == 0 so
// col = 0, c2 = -1, `offset < duty` is never true, state stays Down and only Down (0) / DownUpDown (4) occur.
__device__ __forceinline__ float osm_saw_duty0(Osm& o) {
	// p = float(offset) - col with col == 0 is the 23-bit fraction exactly and only p + p is used: the same bits under exponent 2
	// are 2 + 2p, and taking 2 off is exact — one operation less, the same value
	const float pp = u2f(phase_mantissa<0x80u>(o.offset)) - 2.f;
	const bool carry = o.offset < (uint32_t)o.inc;
	o.offset += (uint32_t)o.inc;
	const float y_lin = o.c2 * (pp - o.f) + 1.f;
	const float y_wrap = -o.rcpf * (1.f + o.c2 * o.omf * (pp + o.omf)) + 1.f;
	return carry ? y_wrap : y_lin;
}
// What a generated patch uses for a saw-family oscillator whose duty is only known at run time: when no voice of the wave
// has a duty (and every state is Down) the short form above is exact, otherwise the general table.
__device__ __forceinline__ float osm_saw_auto(Osm& o) {
	if (__ballot(o.duty != 0u || o.state != 0) == 0ull) return osm_saw_duty0(o);
	return osm_saw(o);
}
__device__ __forceinline__ float osm_pulse(Osm& o) {                        // pulse() 5304-5316
	const float p = fast_phase_float(o.offset);
	const int tr = osm_tick(o);
	const float rcpf2 = o.rcpf2, col = o.col;
	const bool n_up = (tr & 1) != 0, o_up = (tr & 2) != 0, carry = (tr & 4) != 0;
	const float y_flat = n_up ? 1.f : -1.f;                                     // Up (3) / Down (0)
	const float y_wrap = n_up ? (rcpf2 * (col - 1.0f) + 1.f) : (rcpf2 * col - 1.f);   // UpDownUp (7) / DownUpDown (4)
	const float y_ud = rcpf2 * (col - p) + 1.f;                                 // UpDown (2)
	const float y_du = rcpf2 * p - 1.f;                                         // DownUp (5)
	const float y_same = carry ? y_wrap : y_flat;
	const float y_edge = carry ? y_du : y_ud;
	const bool valid_edge = carry ? (!o_up && n_up) : (o_up && !n_up);
	return (o_up == n_up) ? y_same : (valid_edge ? y_edge : 0.f);
}

// ---- Wavetable / Sample klang.h:3626-3720 and Table<float, SIZE>::operator[](float) 3365-3377 ----
// Tables live in HBM (klg_table_upload); a lane names its table by id, identical tables share one (so a bank of notes built
// from the same oscillator reads ONE 8 KB table, which stays in the CU's L1 / L2).  Id 0 is a two-sample table of zeros:
// what the dead lanes of a live wave (all-zero record) read.
struct TableDesc { const float* p; int size; int pad_; };
struct WTab { float inc, pos, off; const float* p; int size; };      // p / size: the lane's table, looked up once per block
__device__ __forceinline__ void wavetable_load(WTab& w, const TableDesc* tabs, uint32_t id) { const TableDesc t = tabs[id]; w.p = t.p; w.size = t.size; }
// Oscillator::set(f) of a Wavetable klang.h:3655-3658: increment = frequency * (size / fs)
__device__ __forceinline__ void wavetable_set_f(WTab& w, float& cached, float f, float fs) { cached = f; w.inc = f * ((float)w.size / fs); }
// Wavetable::process 3676-3679: position += { increment, size } (Phase::operator+=(increment) 1527-1534), then the linear read
// buffer::operator[](float) 2070-2078.  (An index past the last sample — position == size exactly, or a phase offset pushing
// it there — reads beyond the array in the reference; here it wraps to the start.)  The two neighbours are one 8-byte
// gather unless some lane of the wave sits on the last sample (whose neighbour is sample 0).
__device__ __forceinline__ float wavetable_process(WTab& w) {
	const float size = (float)w.size;
	if (!(w.inc >= size)) { w.pos += w.inc; if (w.pos > size) w.pos -= size; }
	const float o = w.pos + w.off;
	const float fl = floorf(o), frac = o - fl;
	int i = (int)o;
	i = (i >= w.size) ? i - w.size : i; i = (i < 0 || i >= w.size) ? 0 : i;
	float a, b;
	if (__ballot(i == w.size - 1) == 0ull) {
		typedef float f2u_t __attribute__((ext_vector_type(2), aligned(4)));
		const f2u_t ab = *reinterpret_cast<const f2u_t*>(w.p + i);
		a = ab.x; b = ab.y;
	}
	else { a = w.p[i]; b = w.p[(i == w.size - 1) ? 0 : i + 1]; }
	return a * (1.f - frac) + b * frac;
}
__device__ __forceinline__ float table_read(const TableDesc* tabs, uint32_t id, float index) {
	const TableDesc t = tabs[id];
	if (index < 0.f) return t.p[0];
	if (index >= (float)(t.size - 1)) return t.p[t.size - 1];
	const float x = floorf(index); const int i = (int)x;
	const float dx = index - x, dy = t.p[i + 1] - t.p[i];
	return t.p[i] + dx * dy;
}

// ---- Filters::Biquad klang.h:5550-5773 ----
struct Biquad { float b0, b1, b2, a1, a2, z0, z1; };
__device__ __forceinline__ float biquad_process(Biquad& q, float in) {      // TDF-II 5605-5612
	const float z0 = q.z0, z1 = q.z1;
	const float y = q.b0 * in + z0;
	q.z0 = q.b1 * in - q.a1 * y + z1;
	q.z1 = q.b2 * in - q.a2 * y;
	return y;
}
// ---- libm parity: sinf / cosf as the reference's host libm computes them ----
// The reference calls libm's cosf/sinf (klang.h:5593-5594); the library in question is glibc 2.35 (Ubuntu 22.04,
// this image), whose sinf/cosf are the ARM "optimized routines" implementation (sysdeps/ieee754/flt-32/s_sinf.c,
// s_cosf.c, sincosf.h, sincosf_data.c): argument reduction by pi/2 in double, two degree-7/8 double polynomials,
// one final rounding to float.  On x86-64 CPUs with FMA glibc's ifunc selects the FMA build, in which every
// a*b+c of those routines is a fused multiply-add.  This is a restatement of that published algorithm with the
// same double coefficients and explicit fma(); it was compared against the host libm on all 317,718,528 floats in
// [2^-31, 120): 0 mismatches for sinf and for cosf (without fma: 6 / 11).  Arguments >= 120 (never reached by a
// filter: w = 2*pi*f/fs < pi) take the OCML fp64 path.
struct SinCosTab { double c0, c1, c2, c3, c4, s1, s2, s3; };
__device__ __forceinline__ float glibc_sincos_poly(double x, double x2, bool neg_tab, int n) {
	const double sg = neg_tab ? -1.0 : 1.0;                     // __sincosf_table[1] = the cosine coefficients negated
	const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5, c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
	const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
	if ((n & 1) == 0) {
		const double x3 = x * x2, t1 = __builtin_fma(x2, s3, s2), x7 = x3 * x2, s = __builtin_fma(x3, s1, x);
		return (float)__builtin_fma(x7, t1, s);
	}
	const double x4 = x2 * x2, q2 = __builtin_fma(x2, c4, c3), q1 = __builtin_fma(x2, c1, c0), x6 = x4 * x2, c = __builtin_fma(x4, c2, q1);
	return (float)__builtin_fma(x6, q2, c);
}
template<bool COS> __device__ __forceinline__ float glibc_sincosf(float y) {
	const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;
	double x = (double)y;
	if (top < ((0x3f490fdbu >> 20) & 0x7ffu)) {                 // |y| < pi/4
		if (top < ((0x39800000u >> 20) & 0x7ffu)) return COS ? 1.0f : y;   // |y| < 2^-12
		return glibc_sincos_poly(x, x * x, false, COS ? 1 : 0);
	}
	if (top < ((0x42f00000u >> 20) & 0x7ffu)) {                 // |y| < 120: reduce_fast
		const double r = x * 0x1.45F306DC9C883p+23;
		const int n = ((int32_t)r + 0x800000) >> 24;
		x = __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
		const int m = COS ? n + 1 : n;
		const double sign = ((m & 3) == 1 || (m & 3) == 2) ? -1.0 : 1.0;   // sign[] = { 1, -1, -1, 1 }
		return glibc_sincos_poly(x * sign, x * x, (m & 2) != 0, COS ? (n ^ 1) : n);
	}
	return COS ? (float)cos((double)y) : (float)sin((double)y);
}
__device__ __forceinline__ float glibc_sinf(float y) { return glibc_sincosf<false>(y); }
__device__ __forceinline__ float glibc_cosf(float y) { return glibc_sincosf<true>(y); }

// Biquad::Filter::set(f, Q) 5584-5600 + LPF::init 5658-5665, evaluated on the device for per-sample swept
// cutoffs (shipped subtractive.k:29, F6) with the libm-exact cosf/sinf above.
struct BiquadSweep { float f, Q; };
__device__ __forceinline__ void biquad_lpf_set(Biquad& q, BiquadSweep& c, float f, float Q, float fs_w) {
	if (Q < 0) Q = f / -Q;
	if (c.f != f || c.Q != Q) {
		c.f = f; c.Q = Q;
		const float w = f * fs_w;
		const float cos0 = glibc_cosf(w);
		const float sin0 = glibc_sinf(w);
		if (Q < 0.5f) Q = 0.5f;
		const float a = sin0 / (2.f * Q);
		const double a0 = (double)(1.f + a);                               // constant a0 = { 1.f + a }  klang.h:97
		const float inv = (a0 == 0.0) ? 0.0f : (float)(1.0 / a0);
		q.a1 = inv * (-2.f * cos0);
		q.a2 = inv * (1.f - a);
		q.b2 = q.b0 = inv * (1.f - cos0) * 0.5f;
		q.b1 = inv * (1.f - cos0);
	}
}

// ---- Filters::OnePole klang.h:5470-5543 ----
// The same for every Filters::Biquad type but the APF (its init() uses a double cos): TYPE as in host_dsl.hpp
// (0 LPF, 1 HPF, 2 BPF constant peak, 3 BPF constant skirt, 4 BRF, 6 Butterworth::LPF<2>); klang.h:5658-5739, 5801-5811.
// set(f, Q) in two halves — what it computes from (f, Q) alone, and what it does to the filter — so that the staged effect kernel (klg_graph_staged.hpp)
// can take the cosine, the sine and the divisions of every sample of a chunk side by side and leave only the comparison with the cached (f, Q) and the five
// assignments in the filter's sample-ordered loop.  biquad_set is apply(coefs()): the same operations in the same order either way.
struct BiquadCoefs { float b0, b1, b2, a1, a2; };
__device__ __forceinline__ float biquad_q(float f, float Q) { return Q < 0 ? f / -Q : Q; }            // what set() caches beside f (klang.h:5585)
template<int TYPE> __device__ __forceinline__ BiquadCoefs biquad_coefs(float f, float Q /* biquad_q(f, Q) */, float fs_w) {
	BiquadCoefs q;
	const float w = f * fs_w;
	const float cos0 = glibc_cosf(w);
	const float sin0 = glibc_sinf(w);
	if (Q < 0.5f) Q = 0.5f;
	const float a = sin0 / (2.f * Q);
	const double a0 = (double)(1.f + a);
	const float inv = (a0 == 0.0) ? 0.0f : (float)(1.0 / a0);
	q.a1 = inv * (-2.f * cos0);
	q.a2 = inv * (1.f - a);
	if (TYPE == 0) { q.b2 = q.b0 = inv * (1.f - cos0) * 0.5f; q.b1 = inv * (1.f - cos0); }
	else if (TYPE == 1) { q.b2 = q.b0 = inv * (1.f + cos0) * 0.5f; q.b1 = inv * -(1.f + cos0); }
	else if (TYPE == 2) { q.b0 = inv * a; q.b1 = 0.f; q.b2 = inv * -a; }
	else if (TYPE == 3) { q.b0 = inv * sin0 * 0.5f; q.b1 = 0.f; q.b2 = -q.b0; }
	else if (TYPE == 4) { q.b1 = q.a1; q.b0 = q.b2 = inv; }
	else { q.b0 = inv * ((1.f - cos0) / 2.f); q.b1 = inv * (1.f - cos0); q.b2 = inv * ((1.f - cos0) / 2.f); }
	return q;
}
template<int TYPE, int K> __device__ __forceinline__ float biquad_coef(float f, float Q, float fs_w) {  // one of them (the calls of a sample stand side by side: the compiler shares the rest)
	const BiquadCoefs q = biquad_coefs<TYPE>(f, Q, fs_w);
	return K == 0 ? q.b0 : K == 1 ? q.b1 : K == 2 ? q.b2 : K == 3 ? q.a1 : q.a2;
}
__device__ __forceinline__ void biquad_apply(Biquad& q, BiquadSweep& c, float f, float Q /* biquad_q */, float b0, float b1, float b2, float a1, float a2) {
	if (c.f != f || c.Q != Q) { c.f = f; c.Q = Q; q.b0 = b0; q.b1 = b1; q.b2 = b2; q.a1 = a1; q.a2 = a2; }
}
template<int TYPE> __device__ __forceinline__ void biquad_set(Biquad& q, BiquadSweep& c, float f, float Q, float fs_w) {
	Q = biquad_q(f, Q);
	if (c.f != f || c.Q != Q) {
		const BiquadCoefs k = biquad_coefs<TYPE>(f, Q, fs_w);
		c.f = f; c.Q = Q;
		q.b0 = k.b0; q.b1 = k.b1; q.b2 = k.b2; q.a1 = k.a1; q.a2 = k.a2;
	}
}
struct OnePole { float b0, b1, a1, z, out; };
__device__ __forceinline__ float onepole_lpf_process(OnePole& q, float in) { q.out = q.b0 * in + q.a1 * q.out + KLG_DENORM; return q.out; }
__device__ __forceinline__ float onepole_process(OnePole& q, float in) { q.out = q.b0 * in + q.b1 * q.z + q.a1 * q.out + KLG_DENORM; q.z = in; return q.out; }

// ---- the remaining filters / modifiers with the hot loop's shape ----
struct Dcf { float r, z, out; };                                             // Filters::DCF klang.h:5386-5397 (r = 0.995 by default)
__device__ __forceinline__ float dcf_process(Dcf& q, float in) { q.out = in - q.z + q.r * q.out; q.z = in; return q.out; }
template<int ORDER> struct Iir { float a[ORDER], y[ORDER]; };                // Filters::IIR<ORDER> 5399-5432
template<int ORDER> __device__ __forceinline__ float iir_process(Iir<ORDER>& q, float in) {
	float out = in;
#pragma unroll
	for (int i = 0; i < ORDER; i++) out -= q.a[i] * q.y[i];
#pragma unroll
	for (int i = ORDER - 1; i > 0; --i) q.y[i] = q.y[i - 1];
	q.y[0] = out;
	return out;
}
struct Iir1 { float a, b, out; };                                            // Filters::IIR<1> 5434-5447
__device__ __forceinline__ float iir1_process(Iir1& q, float in) { q.out = in * q.a + q.out * q.b; return q.out; }
struct Butter1 { float b0, a1, z, out; };                                    // Filters::Butterworth::LPF<1> 5786-5799 (LPF<2> is a Biquad with its own init)
__device__ __forceinline__ float butter1_process(Butter1& q, float in) { q.out = q.b0 * (in + q.z) - q.a1 * q.out; q.z = in; return q.out; }
struct Modal { float a1, a2, y1, y2, gain; };                                // Modifiers::Modal 5815-5859
__device__ __forceinline__ float modal_process(Modal& q, float in) {
	in *= q.gain;                                                            // input()
	const float out = in + q.a1 * q.y1 + q.a2 * q.y2;
	q.y2 = q.y1; q.y1 = out;
	return out;
}
struct FollowerAR { float A, R, out; };                                      // Envelope::Follower::AR 5865-5885
__device__ __forceinline__ float follower_ar_process(FollowerAR& q, float in) { const float smoothing = in > q.out ? q.A : q.R; q.out = q.out + smoothing * (in - q.out); return q.out; }
__device__ __forceinline__ float follower_peak(FollowerAR& q, float in) { return follower_ar_process(q, fabsf(in)); }          // abs(in) >> ar >> out   5902
__device__ __forceinline__ float follower_rms(FollowerAR& q, float in) { return sqrtf(follower_ar_process(q, in * in)); } // (in * in) >> ar >> sqrt >> out   5903

// ---- Envelope klang.h:3722-4102 ----
// Lane state: the Linear ramp (out, target, rate, active 3731-3807), stage, point, time.  Breakpoints live in
// a small register array; they are only touched on the (rare) segment change.
struct Env { float r_out, r_target, r_rate, time; int stage, point; bool active; };

__device__ __forceinline__ void env_set_value(Env& e, float v) { e.r_out = v; e.r_target = v; e.active = false; }      // 3762-3766
__device__ __forceinline__ void env_set_target_time(Env& e, float px, float py, float time, float fs) {              // 4077-4081
	e.time = time;
	e.r_target = py; e.active = (e.r_out != py);
	e.r_rate = fabsf(py - e.r_out) / ((px - time) * fs);
}
__device__ __forceinline__ void env_release(Env& e, float time, float level, float fs) {                             // 3961-3966
	e.stage = ENV_RELEASE;
	env_set_target_time(e, time, level, 0.f, fs);
}
// Breakpoints are held in SCALAR registers and selected with compare/select chains: a register array indexed by
// the (run-time) point number would be demoted to scratch memory, dragging the whole lane state with it.
struct Pts2 { float x0, x1, y0, y1;
	__device__ __forceinline__ float x(int i) const { const float a = x0, b = x1; return i == 1 ? b : a; }
	__device__ __forceinline__ float y(int i) const { const float a = y0, b = y1; return i == 1 ? b : a; } };
struct Pts3 { float x0, x1, x2, y0, y1, y2;
	__device__ __forceinline__ float x(int i) const { const float a = x0, b = x1, c = x2; return i == 2 ? c : (i == 1 ? b : a); }
	__device__ __forceinline__ float y(int i) const { const float a = y0, b = y1, c = y2; return i == 2 ? c : (i == 1 ? b : a); } };

struct Pts4 { float x0, x1, x2, x3, y0, y1, y2, y3;
	__device__ __forceinline__ float x(int i) const { const float a = x0, b = x1, c = x2, d = x3; return i == 3 ? d : (i == 2 ? c : (i == 1 ? b : a)); }
	__device__ __forceinline__ float y(int i) const { const float a = y0, b = y1, c = y2, d = y3; return i == 3 ? d : (i == 2 ? c : (i == 1 ? b : a)); } };

// Envelope::process 4018-4051.  HOLD = the ADSR loop (setLoop(2,2), klang.h:4128): hold at the last point (NP-1).
// The ramp step (Linear::operator++ 3785-3806) is branch-free.  Segment changes / stage changes are rare per lane
// and sit behind ONE wave-uniform branch.  While an ADSR holds at its last point the reference re-applies
// setValue(S) every sample, which changes nothing: that state is treated as settled and skips the rare path.
template<int NP, bool HOLD, class PTS>
__device__ __forceinline__ void env_segment_end(Env& e, const PTS& p, int npoints, const SampleRate& fs) {
	if (e.stage == ENV_SUSTAIN) {
		if (HOLD && (e.point + 1) >= (NP - 1)) {                        // loop.isActive() && (point + 1) >= loop.end
			e.point = NP - 1;
			env_set_value(e, p.y(NP - 1));
		}
		else if ((e.point + 1) < npoints) {
			if (e.time >= p.x(e.point + 1)) {
				e.point++;
				env_set_value(e, p.y(e.point));
				if ((e.point + 1) < npoints)
					env_set_target_time(e, p.x(e.point + 1), p.y(e.point + 1), p.x(e.point), fs.f);
			}
		}
		else e.stage = ENV_OFF;
	}
	else if (e.stage == ENV_RELEASE) e.stage = ENV_OFF;
}
template<int NP, bool HOLD, class PTS>
__device__ __forceinline__ float env_process(Env& e, const PTS& p, int npoints, const SampleRate& fs) {
	const float out = e.r_out;                                          // out = (*ramp)++ : pre-step value
	// Linear::operator++ (3785-3806) without branches.  While a ramp is active r_out != r_target, so r_out is the
	// smallest (ramp up) or largest (ramp down) of { r_out, r_out +/- rate, target } and the reference's
	// "step, then clamp to target when reached or overshot" is exactly the median of the three: one v_med3_f32.
	// (rate = +inf, the zero-length segment of ADSR(0, ...), gives +/-inf and the median is the target.)
	const bool up = e.r_target > e.r_out;
	const float nxt = e.r_out + (up ? e.r_rate : -e.r_rate);
	const float stepped = __builtin_amdgcn_fmed3f(e.r_out, nxt, e.r_target);
	e.r_out = e.active ? stepped : e.r_out;
	e.active = e.active && (stepped != e.r_target);
	const bool sustain = (e.stage == ENV_SUSTAIN);
	e.time = sustain ? (e.time + fs.timeInc) : e.time;
	const bool settled = HOLD && (e.point == NP - 1);
	const bool rare = !e.active && ((sustain && !settled) || e.stage == ENV_RELEASE);
	if (__ballot(rare) != 0ull) {
		if (rare) env_segment_end<NP, HOLD>(e, p, npoints, fs);
	}
	return out;
}

// The same with everything known only at run time: what a recorded graph patch uses for its Envelope members.  Any number of breakpoints (the first
// four in registers — Pts4 —, the others read from the voice's record when a segment ends: PtsN), the loop (Envelope::setLoop, klang.h:3923-3926;
// Loop 3853-3864: ls / le = loop.start / loop.end, -1 = no loop) and the mode (setMode, klang.h:4064-4071): in Rate mode a point's x is the ramp's step
// per sample (setTargetRate 4083-4092) and a finished segment goes on at once (`mode() == Rate ||`, 4031).  `npm` = point count | Rate mode << 16.
struct PtsN { Pts4 head; const uint32_t* ext; size_t stride; int slots;      // ext: the record word of point 4's x; x of points 4 .. follow, then (slots words on) their y
	__device__ __forceinline__ float x(int i) const { const float far = u2f(ext[(size_t)(i > 4 ? i - 4 : 0) * stride]); return i < 4 ? head.x(i) : far; }
	__device__ __forceinline__ float y(int i) const { const float far = u2f(ext[(size_t)(slots + (i > 4 ? i - 4 : 0)) * stride]); return i < 4 ? head.y(i) : far; } };
enum { ENV_NPM_RATE = 1 << 16 };
__device__ __forceinline__ void env_set_target_rt(Env& e, float px, float py, float time, float fs, bool rate) {
	if (!rate) { env_set_target_time(e, px, py, time, fs); return; }
	e.time = 0.f;                                                       // setTargetRate 4083-4092
	if (px == 0.f) env_set_value(e, py);
	else { e.r_target = py; e.active = (e.r_out != py); e.r_rate = px; }
}
__device__ __forceinline__ void env_release_rt(Env& e, float time, float level, float fs, bool rate) { e.stage = ENV_RELEASE; env_set_target_rt(e, time, level, 0.f, fs, rate); }   // 3961-3966
template<class PTS>
__device__ __forceinline__ void env_segment_end_rt(Env& e, const PTS& p, int npm, int ls, int le, const SampleRate& fs) {
	const int npoints = npm & 0xFFFF; const bool rate = (npm & ENV_NPM_RATE) != 0;
	if (e.stage == ENV_SUSTAIN) {
		if (ls >= 0 && le >= 0 && (e.point + 1) >= le) {                // loop.isActive() && (point + 1) >= loop.end
			e.point = ls;
			env_set_value(e, p.y(ls));
			if (ls != le) env_set_target_rt(e, p.x(ls + 1), p.y(ls + 1), p.x(ls), fs.f, rate);
		}
		else if ((e.point + 1) < npoints) {
			if (rate || e.time >= p.x(e.point + 1)) {                      // mode() == Rate || time >= points[point + 1].x   4031
				e.point++;
				env_set_value(e, p.y(e.point));
				if ((e.point + 1) < npoints)
					env_set_target_rt(e, p.x(e.point + 1), p.y(e.point + 1), p.x(e.point), fs.f, rate);
			}
		}
		else e.stage = ENV_OFF;
	}
	else if (e.stage == ENV_RELEASE) e.stage = ENV_OFF;
}
// holding on a one-point loop re-applies setValue(points[start].y) every sample: nothing changes once it has been applied.  `hold_y` = points[loop.start].y,
// looked up once per block (env_hold_y): the per-sample test reads no breakpoint.
template<class PTS> __device__ __forceinline__ float env_hold_y(const PTS& p, int ls) { return p.y(ls < 0 ? 0 : ls); }
__device__ __forceinline__ bool env_settled(const Env& e, int ls, int le, float hold_y) { return ls >= 0 && ls == le && e.point == ls && e.r_out == hold_y; }
template<class PTS>
__device__ __forceinline__ float env_process_rt(Env& e, const PTS& p, int npm, int ls, int le, float hold_y, const SampleRate& fs) {
	const float out = e.r_out;
	const bool up = e.r_target > e.r_out;
	const float nxt = e.r_out + (up ? e.r_rate : -e.r_rate);
	const float stepped = __builtin_amdgcn_fmed3f(e.r_out, nxt, e.r_target);
	e.r_out = e.active ? stepped : e.r_out;
	e.active = e.active && (stepped != e.r_target);
	const bool sustain = (e.stage == ENV_SUSTAIN);
	e.time = sustain ? (e.time + fs.timeInc) : e.time;
	const bool rare = !e.active && ((sustain && !env_settled(e, ls, le, hold_y)) || e.stage == ENV_RELEASE);
	if (__ballot(rare) != 0ull) {
		if (rare) env_segment_end_rt(e, p, npm, ls, le, fs);
	}
	return out;
}

// Envelope state <-> 6 flag bits: stage(2) | point(3) | active(1)
__device__ __forceinline__ uint32_t env_pack(const Env& e) { return (uint32_t)e.stage | ((uint32_t)e.point << 2) | ((uint32_t)e.active << 5); }
__device__ __forceinline__ void env_unpack(Env& e, uint32_t b) { e.stage = (int)(b & 3u); e.point = (int)((b >> 2) & 7u); e.active = ((b >> 5) & 1u) != 0; }
// ... and the bits word of a graph record (include/klang_mi355_graph.h env_bits): the same six, Rate mode in bit 6 (host-owned: kept as it came), the point's higher bits from bit 7
__device__ __forceinline__ uint32_t env_pack_rt(const Env& e, int npm) { return (uint32_t)e.stage | ((uint32_t)(e.point & 7) << 2) | ((uint32_t)e.active << 5) | ((npm & ENV_NPM_RATE) ? 64u : 0u) | ((uint32_t)(e.point >> 3) << 7); }
__device__ __forceinline__ void env_unpack_rt(Env& e, int& npm, uint32_t b, uint32_t npoints) { e.stage = (int)(b & 3u); e.point = (int)(((b >> 2) & 7u) | ((b >> 7) << 3)); e.active = ((b >> 5) & 1u) != 0; npm = (int)(npoints & 0xFFFFu) | ((b & 64u) ? (int)ENV_NPM_RATE : 0); }

} // namespace klg
