// Exhaustive check behind the branch-free quadrant fold of klg::fastsinp (klg_device.hpp).  klang.h:5117-5132 folds the phase angle
//     x in [0, 2pi):   if (x > 3pi/2) x -= 2pi;  else if (x > pi/2) x = pi - x;
// and the same value is the MEDIAN of { x, pi - x, x - 2pi } (one v_med3_f32 instead of two compares and two selects): the three are
// ordered x-2pi <= x <= pi-x in the first quadrant, x-2pi <= pi-x < x in the middle two, pi-x < x-2pi < x in the last.  The fold only
// sees the 2^23 values (mantissa of the phase) * 2pi, rounded — all of them are tried here, in float arithmetic as the device does it.
// Build: gcc -O2 -mfma -ffp-contract=off tools/verify_fastsinp_med3.c -o /tmp/verify_fastsinp_med3 -lm ; run: /tmp/verify_fastsinp_med3
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
static float med3(float a, float b, float c) { const float lo = fminf(a, b), hi = fmaxf(a, b); return fmaxf(lo, fminf(hi, c)); }
int main(void) {
	const float PI_F = 3.14159274101257324f, TWO_PI = 6.28318548202514648f, HALF_PI = 1.57079637050628662f, THREE_HALF_PI = 4.71238899230957031f;
	uint64_t bad = 0, bad_fma = 0;
	for (uint32_t k = 0; k < (1u << 23); k++) {
		const uint32_t bits = 0x3f800000u | k; float m; memcpy(&m, &bits, 4);
		const float x = (m - 1.f) * TWO_PI;
		{ const float xf = fmaf(m, TWO_PI, -TWO_PI); uint32_t a, b; memcpy(&a, &x, 4); memcpy(&b, &xf, 4); if (a != b) bad_fma++; }   // (round 6) klg::phase_radians: the same value in one fma
		float want = x;
		if (want > THREE_HALF_PI) want -= TWO_PI; else if (want > HALF_PI) want = PI_F - want;
		const float got = med3(x, PI_F - x, x - TWO_PI);
		uint32_t a, b; memcpy(&a, &want, 4); memcpy(&b, &got, 4);
		if (a != b) { if (bad < 5) printf("k=%u x=%.9g want=%.9g got=%.9g\n", k, x, want, got); bad++; }
	}
	printf("fastsinp quadrant fold as a median: %llu mismatches over the 2^23 phase mantissas\n", (unsigned long long)bad);
	printf("(m - 1) * twoPi as fma(m, twoPi, -twoPi): %llu mismatches over the 2^23 phase mantissas\n", (unsigned long long)bad_fma);
	return bad != 0 || bad_fma != 0;
}
