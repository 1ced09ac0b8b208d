// tools/verify_glibc_pow.cpp — the check behind klang_amd/csrc/klg_glibc_pow.hpp: compiles that header for the HOST and compares, bit for bit as doubles, with the
// C library this container pins (glibc 2.35, the __pow_fma variant on a host with FMA):
//   pow(10.0, (double)f) and exp2((double)f) for ALL 2^32 floats f (what Modular.k feeds them), pow(2.0, (double)f) likewise, and pow(x, y) for random positive normal x
//   and random y over every exponent range (including the under- / overflow edges).
// Build: g++ -O2 -std=c++17 -ffp-contract=off -mfma -fopenmp tools/verify_glibc_pow.cpp -o /tmp/verify_glibc_pow -lm ; run: /tmp/verify_glibc_pow [random pairs, default 1e9] [float stride, default 1]
// (the full run: all 2^32 floats 0 differ, 10^9 pairs 0 differ — 8 threads, 15 minutes; tests/test_glibc_pow_cpu.py runs every 251st float and 2 * 10^6 pairs)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../klang_amd/csrc/klg_glibc_pow.hpp"
static inline uint64_t bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline bool same(double a, double b) { return bits(a) == bits(b) || (a != a && b != b); }
static inline uint64_t mix(uint64_t x) { x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31); }
int main(int argc, char** argv) {
	const long long pairs = argc > 1 ? atoll(argv[1]) : 1000000000ll, stride = argc > 2 ? atoll(argv[2]) : 1;
	double lo10, lo2; const double hi10 = klg::glibc::pow_log(10.0, &lo10), hi2 = klg::glibc::pow_log(2.0, &lo2);
	long long bad10 = 0, bad2 = 0, bade = 0, badr = 0;
	volatile double ten = 10.0, two = 2.0;                       // (volatile: the compiler must call the library, not fold or rewrite pow(2, x) into exp2)
#pragma omp parallel for reduction(+ : bad10, bad2, bade) schedule(static)
	for (long long u = 0; u < (1ll << 32); u += stride) {
		const uint32_t w = (uint32_t)u; float f; memcpy(&f, &w, 4);
		const double y = (double)f;
		const double a = pow(ten, y), b = klg::glibc::pow_of_log(10.0, y, hi10, lo10);
		if (!same(a, b)) { if (bad10++ < 3) fprintf(stderr, "pow(10, %a): libm %a, restated %a\n", y, a, b); }
		const double c = pow(two, y), d = klg::glibc::pow_of_log(2.0, y, hi2, lo2);
		if (!same(c, d)) { if (bad2++ < 3) fprintf(stderr, "pow(2, %a): libm %a, restated %a\n", y, c, d); }
		const double e = exp2(y), g = klg::glibc::exp2(y);
		if (!same(e, g)) { if (bade++ < 3) fprintf(stderr, "exp2(%a): libm %a, restated %a\n", y, e, g); }
	}
	printf("floats (every %lld-th of 2^32): pow(10, f) %lld differ, pow(2, f) %lld differ, exp2(f) %lld differ\n", stride, bad10, bad2, bade);
#pragma omp parallel for reduction(+ : badr) schedule(static)
	for (long long i = 0; i < pairs; i++) {
		uint64_t a = mix(2 * (uint64_t)i), b = mix(2 * (uint64_t)i + 1);
		uint64_t ex = 1 + (mix(a) % 2046);                                             // x: positive, normal, any exponent; one in four near 1
		if ((a & 3) == 0) ex = 1022 + (a >> 2) % 2;
		const uint64_t xb = (ex << 52) | (a >> 12);
		double x, y; memcpy(&x, &xb, 8);
		uint64_t ey = 1023 - 70 + (mix(b) % 140);                                      // y: 2^-70 .. 2^70 either sign; now and then any bit pattern (zero, inf, nan, subnormal)
		uint64_t yb = (b & 0x8000000000000000ull) | (ey << 52) | ((b >> 11) & 0xfffffffffffffull);
		if ((b & 0xff) == 0) yb = mix(b);
		memcpy(&y, &yb, 8);
		const double r = pow(x, y), s = klg::glibc::pow_pos(x, y);
		if (!same(r, s)) { if (badr++ < 5) fprintf(stderr, "pow(%a, %a): libm %a, restated %a\n", x, y, r, s); }
	}
	printf("%lld random (x, y): %lld differ\n", pairs, badr);
	return (bad10 || bad2 || bade || badr) ? 1 : 0;
}
