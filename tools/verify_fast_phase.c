// Exhaustive check behind klg::fast_phase_of_product (klg_device.hpp): the Fast::Phase of a float (klang.h:4993-4998,
//     position = (uint32_t)(int64_t)(radians * FINTMAX / twoPi)          [x86: cvttss2si; out of range -> low word 0]
// computed WITHOUT the IEEE division expansion and without a 64-bit conversion:
//     n = radians * FINTMAX;  q = n * r;  e = fma(-y, q, n);  q2 = fma(e, r, q)        (y = twoPi, r = RN(1 / y): 3 operations)
//     hi = floor(|q2| * 2^-32);  lo = fma(hi, -2^32, |q2|);  u = cvt_u32(lo) [NaN -> 0];  result = q2 < 0 ? -u : u
// q2 is NOT the IEEE quotient for every n (it is for |n| in [2^-100, 2^100]) — but the conversion only looks at quotients of
// magnitude [1, 2^63), and everything outside that gives 0 either way.  This program checks the COMPOSITION on all 2^32 floats n.
// Build: gcc -O2 -mfma -ffp-contract=off -fopenmp tools/verify_fast_phase.c -o /tmp/verify_fast_phase -lm ; run: /tmp/verify_fast_phase
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
static uint32_t ref_wrap(float x) { return fabsf(x) < 9223372036854775808.0f ? (uint32_t)(int64_t)x : 0u; }   // klg_device.hpp f2u_wrap
static uint32_t cvt_u32(float x) { if (!(x == x)) return 0u; if (x <= 0.f) return 0u; if (x >= 4294967296.0f) return 0xFFFFFFFFu; return (uint32_t)x; }   // v_cvt_u32_f32
int main(void) {
	const float y = (float)(2.0 * 3.14159265358979323846), r = 1.0f / y;
	uint64_t bad = 0, bad_div = 0;
	#pragma omp parallel for reduction(+:bad,bad_div) schedule(static)
	for (int64_t i = 0; i < (1ll << 32); i++) {
		uint32_t u = (uint32_t)i; float n; memcpy(&n, &u, 4);
		const uint32_t want = ref_wrap(n / y);
		const float q = n * r, e = fmaf(-y, q, n), q2 = fmaf(e, r, q);
		const float a = fabsf(q2), hi = floorf(a * 2.3283064365386963e-10f), lo = fmaf(hi, -4294967296.0f, a);
		uint32_t s; memcpy(&s, &q2, 4); s = (uint32_t)((int32_t)s >> 31);
		const uint32_t got = (cvt_u32(lo) ^ s) - s;
		if (got != want) bad++;
		const float ieee = n / y; uint32_t b1, b2; memcpy(&b1, &ieee, 4); memcpy(&b2, &q2, 4);
		if (b1 != b2 && !(ieee != ieee && q2 != q2)) bad_div++;
	}
	printf("fast_phase composition: %llu mismatches over 2^32 inputs (the 3-operation quotient alone differs from IEEE on %llu of them)\n", (unsigned long long)bad, (unsigned long long)bad_div);
	return bad != 0;
}
