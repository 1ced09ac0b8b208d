// Exhaustive check behind klg::fast_phase (klg_device.hpp, round 6 form): the Fast::Phase of a float (klang.h:4993-4998,
//     position = (uint32_t)(int64_t)(radians * FINTMAX / twoPi)          [x86: cvttss2si; out of range -> low word 0; FINTMAX = 2^31]
// computed at the scale of CYCLE PAIRS and through the FRACTION of the quotient:
//     q = radians * r;  e = fma(-y, q, radians);  q2 = fma(e, r, q)                  (y = 2 * twoPi, r = RN(1 / twoPi) / 2: the 3-operation quotient radians / 4 pi)
//     f = fract(|q2|)  [v_fract_f32: x - floor(x), exact for x >= 0];  u = cvt_u32(f * 2^32)  [NaN -> 0];  result = q2 < 0 ? -u : u
// Scaling by powers of two commutes with every rounding above (no overflow / underflow in the range the conversion looks at): q2 * 2^32 IS radians * 2^31 / twoPi, and
// trunc(|q| * 2^32) mod 2^32 = trunc(fract(|q|) * 2^32): the integer part of |q| only contributes multiples of 2^32.  This program checks the
// COMPOSITION against the reference expression on all 2^32 floats `radians`.
// Build: gcc -O2 -mfma -ffp-contract=off -fopenmp tools/verify_fast_phase_fract.c -o /tmp/verify_fast_phase_fract -lm ; run it (~1 min on 16 cores)
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
static uint32_t ref_wrap(float x) { return fabsf(x) < 9223372036854775808.0f ? (uint32_t)(int64_t)x : 0u; }   // klg_device.hpp f2u_wrap
static uint32_t cvt_u32(float x) { if (!(x == x)) return 0u; if (x <= 0.f) return 0u; if (x >= 4294967296.0f) return 0xFFFFFFFFu; return (uint32_t)x; }   // v_cvt_u32_f32
static float fract_hw(float x) { if (!(x == x) || isinf(x)) return NAN; const float f = x - floorf(x); return f >= 1.0f ? 0.99999994f : f; }   // v_fract_f32 (x >= 0 here: the subtraction is exact)
int main(void) {
	const float y = (float)(2.0 * 3.14159265358979323846), r = 1.0f / y, y2 = 2.f * y, r2 = 0.5f * r;
	const float FINTMAX = 2147483648.0f;
	uint64_t bad = 0;
	#pragma omp parallel for reduction(+:bad) schedule(static)
	for (int64_t i = 0; i < (1ll << 32); i++) {
		uint32_t u = (uint32_t)i; float t; memcpy(&t, &u, 4);
		const uint32_t want = ref_wrap((t * FINTMAX) / y);
		const float q = t * r2, e = fmaf(-y2, q, t), q2 = fmaf(e, r2, q);
		const float f = fract_hw(fabsf(q2)), g = f * 4294967296.0f;
		uint32_t s; memcpy(&s, &q2, 4); s = (uint32_t)((int32_t)s >> 31);
		const uint32_t got = (cvt_u32(g) ^ s) - s;
		if (got != want) { bad++; if (bad < 5) fprintf(stderr, "t %a want %08x got %08x\n", t, want, got); }
	}
	printf("fast_phase through fract: %llu mismatches over 2^32 inputs\n", (unsigned long long)bad);
	return bad != 0;
}
