/* tools/verify_sin_f64.c — klang_amd/csrc/klg_device.hpp sin_f64 (the device's ::sin(double) for oscillator arguments) against glibc's sin, AFTER
 * ROUNDING TO FLOAT (what Basic::Sine returns, klang.h:4902): every float of [0, 2 pi] and N random arguments (oscillator range, phase offsets up to
 * +-1000, arbitrary bit patterns below 9000).  The same IEEE double operations as the device function (fma, rint, no contraction).
 *   gcc -O2 -ffp-contract=off -mfma tools/verify_sin_f64.c -o /tmp/verify_sin -lm && /tmp/verify_sin 200000000
 *   -> N=200000000 float mismatches=0 ... exhaustive [0,2pi]: 105364991 floats, 0 mismatches */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static double klg_sin_f64(double x) {
	if (!(fabs(x) < 1.0e4)) return sin(x);
	const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
	const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
	const double fn = rint(x * 6.36619772367581382433e-01);
	double r = fma(-fn, 1.57079632679489655800e+00, x);
	r = fma(-fn, 6.12323399573676603587e-17, r);
	const int n = (int)fn;
	const int odd = (n & 1) != 0;
	const double K2 = odd ? C2 : S2, K3 = odd ? C3 : S3, K4 = odd ? C4 : S4, K5 = odd ? C5 : S5, K6 = odd ? C6 : S6;
	const double z = r * r, v = z * r;
	const double h = K2 + z * (K3 + z * (K4 + z * (K5 + z * K6)));
	const double s = r + v * (S1 + z * h);
	const double rc = z * (C1 + z * h);
	uint64_t b; const double ax = fabs(r); memcpy(&b, &ax, 8);
	const uint32_t hi = (uint32_t)(b >> 32);
	double qx;
	if (hi < 0x3FD33333u) qx = 0.0; else if (hi > 0x3fe90000u) qx = 0.28125; else { const uint64_t q = (uint64_t)(hi - 0x00200000u) << 32; memcpy(&qx, &q, 8); }
	const double hz = 0.5 * z - qx, a = 1.0 - qx;
	const double c = a - (hz - z * rc);
	double res = odd ? c : s;
	return (n & 2) ? -res : res;
}
int main(int argc, char** argv) {
	long N = argc > 1 ? atol(argv[1]) : 100000000L; long bad = 0, ulp1 = 0; double maxrel = 0;
	uint64_t st = 88172645463325252ull;
	for (long i = 0; i < N; i++) {
		st ^= st << 13; st ^= st >> 7; st ^= st << 17;
		float xf;
		const int mode = (int)(st >> 61);
		const double u = (double)(st & 0xFFFFFFFFFFFFull) / (double)0x1000000000000ull;
		if (mode < 5) xf = (float)(u * 6.2831853 * 1.0001);            /* the oscillator's own range */
		else if (mode == 5) xf = (float)(u * 40.0 - 20.0);             /* phase offsets */
		else if (mode == 6) xf = (float)((u - 0.5) * 2000.0);
		else { uint32_t w = (uint32_t)(st >> 16); memcpy(&xf, &w, 4); if (!(fabsf(xf) < 9000.f)) xf = (float)u; }   /* any float pattern */
		const double x = (double)xf, a = klg_sin_f64(x), b = sin(x);
		if ((float)a != (float)b) { if (bad < 5) printf("mismatch x=%.9g mine=%.17g glibc=%.17g\n", xf, a, b); bad++; }
		if (a != b) ulp1++;
		if (b != 0) { double rel = fabs((a - b) / b); if (rel > maxrel) maxrel = rel; }
	}
	printf("N=%ld float mismatches=%ld double differs=%ld (%.3g%%) max rel err=%.3g\n", N, bad, ulp1, 100.0 * ulp1 / N, maxrel);
	/* exhaustive over the float grid of [0, 2pi] */
	long ex = 0, exbad = 0; for (float xf = 0.f; xf <= 6.2831860f; xf = nextafterf(xf, 10.f)) { ex++; if ((float)klg_sin_f64((double)xf) != (float)sin((double)xf)) { if (exbad < 5) printf("exh mismatch %.9g\n", xf); exbad++; } if (xf < 1e-3f) xf = xf * 1.0001f + 1e-12f; }
	printf("exhaustive [0,2pi]: %ld floats, %ld mismatches\n", ex, exbad);
	return bad != 0 || exbad != 0;
}
