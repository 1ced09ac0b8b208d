// Exhaustive check behind klg::div_const (klg_device.hpp): for a constant divisor y and r = RN(1/y),
//   q = RN(x*r); e = fma(-y, q, x); q2 = fma(e, r, q)
// equals the IEEE quotient x / y for every finite non-zero float x (zeros, infinities and NaN take q).
// Usage: verify_div_const [stride] y1 y2 ...   (stride 1 = all 2^32 bit patterns, ~20 s per divisor on 2 cores)
// Build: gcc -O2 -mfma -ffp-contract=off -fopenmp tools/verify_div_const.c -o /tmp/verify_div_const -lm
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
static uint64_t check(float y, int64_t stride) {
	const float r = 1.0f / y;
	uint64_t bad = 0;
	#pragma omp parallel for reduction(+:bad) schedule(static)
	for (int64_t i = 0; i < (1ll << 32); i += stride) {
		uint32_t u = (uint32_t)i; float x; memcpy(&x, &u, 4);
		const float ref = x / y;
		const float q = x * r;
		const float e = fmaf(-y, q, x);
		const float q2 = fmaf(e, r, q);
		const int special = !(fabsf(x) < INFINITY) || x == 0.0f;
		const float got = special ? q : q2;
		uint32_t a, b; memcpy(&a, &ref, 4); memcpy(&b, &got, 4);
		if (a != b && !(isnan(ref) && isnan(got))) bad++;
	}
	return bad;
}
int main(int argc, char** argv) {
	if (argc < 3) { fprintf(stderr, "usage: %s stride y...\n", argv[0]); return 2; }
	const int64_t stride = atoll(argv[1]);
	int rc = 0;
	for (int k = 2; k < argc; k++) {
		const float y = strtof(argv[k], NULL);
		const uint64_t bad = check(y, stride);
		printf("%g %llu\n", y, (unsigned long long)bad);
		if (bad) rc = 1;
	}
	return rc;
}
