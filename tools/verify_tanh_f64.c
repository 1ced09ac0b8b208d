// tools/verify_tanh_f64.c — the check behind klg::tanh_f64 (klg_device.hpp): `tanh(x)` of a float inside a patch's plain C function
// (examples/Distortion/Shaping.k:15 `tanh(c * x) / tanh(c)`) is the C library's DOUBLE tanh — the float converts to double, the result is rounded
// back to float (verified on the pinned build: the reference binary imports `tanh`, not `tanhf`).  glibc 2.35: sysdeps/ieee754/dbl-64/s_tanh.c over
// s_expm1.c (fdlibm).  This program restates both in plain C (no fma: -ffp-contract=off) and compares (float)restated((double)x) with (float)tanh((double)x)
// of the host's libm for ALL 2^32 float bit patterns.
// Build: gcc -O2 -ffp-contract=off -fopenmp tools/verify_tanh_f64.c -o /tmp/verify_tanh_f64 -lm ; run: /tmp/verify_tanh_f64
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline uint32_t hi_word(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
static inline uint32_t lo_word(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
static inline double with_hi(double x, uint32_t h) { uint64_t u; memcpy(&u, &x, 8); u = (u & 0xFFFFFFFFull) | ((uint64_t)h << 32); memcpy(&x, &u, 8); return x; }
static double k_expm1(double x) {
	static const double one = 1.0, tiny = 1.0e-300, o_threshold = 7.09782712893383973096e+02, ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
		invln2 = 1.44269504088896338700e+00, Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03, Q3 = -7.93650757867487942473e-05,
		Q4 = 4.00821782732936239552e-06, Q5 = -2.01099218183624371326e-07;
	double y, hi, lo, c = 0.0, t, e, hxs, hfx, r1, h2, h4, R1, R2, R3;
	int32_t k;
	uint32_t hx = hi_word(x);
	const uint32_t xsb = hx & 0x80000000u;
	hx &= 0x7fffffffu;
	if (hx >= 0x4043687Au) {                                   /* |x| >= 56 ln2 */
		if (hx >= 0x40862E42u) {
			if (hx >= 0x7ff00000u) { if (((hx & 0xfffffu) | lo_word(x)) != 0) return x + x; return xsb == 0 ? x : -1.0; }
			if (x > o_threshold) return 1.0e+300 * 1.0e+300;
		}
		if (xsb != 0) return tiny - one;
	}
	if (hx > 0x3fd62e42u) {                                    /* |x| > 0.5 ln2 */
		if (hx < 0x3FF0A2B2u) { if (xsb == 0) { hi = x - ln2_hi; lo = ln2_lo; k = 1; } else { hi = x + ln2_hi; lo = -ln2_lo; k = -1; } }
		else { k = (int32_t)(invln2 * x + (xsb == 0 ? 0.5 : -0.5)); t = k; hi = x - t * ln2_hi; lo = t * ln2_lo; }
		x = hi - lo; c = (hi - x) - lo;
	}
	else if (hx < 0x3c900000u) return x;                       /* |x| < 2^-54 */
	else k = 0;
	hfx = 0.5 * x; hxs = x * hfx;
	R1 = one + hxs * Q1; h2 = hxs * hxs;
	R2 = Q2 + hxs * Q3; h4 = h2 * h2;
	R3 = Q4 + hxs * Q5;
	r1 = R1 + h2 * R2 + h4 * R3;
	t = 3.0 - r1 * hfx;
	e = hxs * ((r1 - t) / (6.0 - x * t));
	if (k == 0) return x - (x * e - hxs);
	e = (x * (e - c) - c); e -= hxs;
	if (k == -1) return 0.5 * (x - e) - 0.5;
	if (k == 1) { if (x < -0.25) return -2.0 * (e - (x + 0.5)); return one + 2.0 * (x - e); }
	if (k <= -2 || k > 56) { y = one - (e - x); y = with_hi(y, hi_word(y) + ((uint32_t)k << 20)); return y - one; }
	t = one;
	if (k < 20) { t = with_hi(t, 0x3ff00000u - (0x200000u >> k)); y = t - (e - x); y = with_hi(y, hi_word(y) + ((uint32_t)k << 20)); }
	else { t = with_hi(t, (uint32_t)(0x3ff - k) << 20); y = x - (e + t); y += one; y = with_hi(y, hi_word(y) + ((uint32_t)k << 20)); }
	return y;
}
static double k_tanh(double x) {
	static const double one = 1.0, two = 2.0, tiny = 1.0e-300;
	double t, z;
	const uint32_t jx = hi_word(x), lx = lo_word(x), ix = jx & 0x7fffffffu;
	if (ix >= 0x7ff00000u) { if ((int32_t)jx >= 0) return one / x + one; return one / x - one; }
	if (ix < 0x40360000u) {                                     /* |x| < 22 */
		if ((ix | lx) == 0) return x;
		if (ix < 0x3c800000u) return x * (one + x);             /* |x| < 2^-55 */
		if (ix >= 0x3ff00000u) { t = k_expm1(two * fabs(x)); z = one - two / (t + two); }
		else { t = k_expm1(-two * fabs(x)); z = -t / (t + two); }
	}
	else z = one - tiny;
	return ((int32_t)jx >= 0) ? z : -z;
}
int main(void) {
	uint64_t bad = 0, shown = 0;
	#pragma omp parallel for reduction(+:bad) schedule(static)
	for (int64_t i = 0; i < (1ll << 32); i++) {
		uint32_t u = (uint32_t)i; float x; memcpy(&x, &u, 4);
		const float want = (float)tanh((double)x), got = (float)k_tanh((double)x);
		uint32_t a, b; memcpy(&a, &want, 4); memcpy(&b, &got, 4);
		if (a != b && !(want != want && got != got)) { bad++; if (shown < 5) { shown++; printf("x = %a: %a != %a\n", x, got, want); } }
	}
	printf("(float)tanh((double)x): %llu mismatches over 2^32 floats\n", (unsigned long long)bad);
	return bad != 0;
}
