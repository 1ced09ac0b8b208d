// klg_rand.hpp — libc rand() as klang's Noise generators see it, restated so that it runs ON THE DEVICE.
//
// Replaces: the draws of Generators::Basic::Noise::process() (klang.h:4947-4951: `rand() * 2.f / (float)RAND_MAX - 1.f`) and
// Generators::Fast::Noise::process() (klang.h:5357-5366), and klang::random(seed) = srand(seed) (klang.h:236-240).  The reference's
// rand() is the C library's: on the Linux hosts klang is built for, glibc 2.35 stdlib/random_r.c, TYPE_3 — an additive feedback
// generator over 31 words of 32 bits:
//       o[i] = o[i-31] + o[i-3]   (mod 2^32),        rand() = o[i] >> 1
// seeded by srand(seed) with a Lehmer fill (r[i] = 16807 * r[i-1] mod 2^31 - 1 by Schrage's method, r[0] = seed or 1) and 310
// discarded draws.  (A third-party algorithm, absent from /root/reference: restated here from its published description;
// tools/verify_rand.c runs this file against the C library's own srand() / rand() — 10^8 draws for several seeds, the jump-ahead, and
// the round trip through the library's state — and tests/test_gpu_rand.py the device kernels against the library's draws.)
//
// Every Noise object of a process draws from the ONE sequence, in the order Synth::process walks the notes (klang.h:4842-4848: synth by
// synth, note slot by note slot, each sounding note through the whole block).  So the value a voice hears is a position in the stream:
// rank r among the block's sounding voices, sample i, generator g -> position G + (r * n + i) * draws + g.  The recurrence is LINEAR:
// with p(t) = t^31 - t^28 - 1, o[q + k + j] = sum_i c_i o[q + i + j] where sum c_i t^i = t^k mod p(t) (coefficients mod 2^32).  A lane
// takes the stream's state at the block's start, jumps to its rank's first draw with four such polynomials (the digits of the rank in
// base 64, tables made on the host per `n * draws`), and then runs the recurrence itself: one add per draw, 31 words in registers.
#pragma once
#include <cstdint>
#include <cstring>
#ifndef __HIPCC_RTC__
#include <cstdlib>
#include <vector>
#endif

#if defined(__HIPCC__) || defined(__HIP__)
#define KLG_RAND_HD __host__ __device__
#else
#define KLG_RAND_HD
#endif

namespace klg_rand {

constexpr int DEG = 31;        // words of state (TYPE_3)
constexpr int SEP = 3;         // the second lag
constexpr int LEVELS = 5;      // base-64 digits of a rank: 64^5 = 2^30 ranks
constexpr int DIGITS = 64;

// The stream's state in TIME ORDER: x[0] = o[i-31] (the oldest) ... x[30] = o[i-1].  The next value is x[0] + x[28].
struct State { uint32_t x[DEG]; };

KLG_RAND_HD inline uint32_t step(State& s) {           // one draw: advances the state, returns o[i] (rand() = >> 1)
	const uint32_t v = s.x[0] + s.x[DEG - SEP];
	for (int j = 0; j < DEG - 1; j++) s.x[j] = s.x[j + 1];
	s.x[DEG - 1] = v;
	return v;
}

#ifndef __HIPCC_RTC__
// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
// srand(seed): glibc __srandom_r for TYPE_3
inline State seeded(unsigned seed) {
	int32_t r[DEG];
	if (seed == 0) seed = 1;
	r[0] = (int32_t)seed;
	int32_t word = (int32_t)seed;
	for (int i = 1; i < DEG; i++) {
		const long hi = word / 127773, lo = word % 127773;
		long w = 16807 * lo - 2836 * hi;
		if (w < 0) w += 2147483647;
		word = (int32_t)w;
		r[i] = word;
	}
	// fptr = &state[3], rptr = &state[0]: the slot fptr points at holds the oldest value
	State s;
	for (int j = 0; j < DEG; j++) s.x[j] = (uint32_t)r[(SEP + j) % DEG];
	for (int i = 0; i < 10 * DEG; i++) (void)step(s);
	return s;
}

// t^k mod p(t) over Z / 2^32
struct Poly { uint32_t c[DEG]; };
inline Poly poly_one() { Poly p; std::memset(&p, 0, sizeof p); p.c[0] = 1; return p; }
inline Poly poly_t() { Poly p; std::memset(&p, 0, sizeof p); p.c[1] = 1; return p; }
inline Poly poly_mul(const Poly& a, const Poly& b) {
	uint32_t w[2 * DEG - 1] = {};
	for (int i = 0; i < DEG; i++) if (a.c[i]) for (int j = 0; j < DEG; j++) w[i + j] += a.c[i] * b.c[j];
	for (int d = 2 * DEG - 2; d >= DEG; d--) { w[d - SEP] += w[d]; w[d - DEG] += w[d]; }      // t^31 = t^28 + 1
	Poly p; std::memcpy(p.c, w, sizeof p.c); return p;
}
inline Poly poly_pow_t(unsigned long long k) {
	Poly r = poly_one(), b = poly_t();
	while (k) { if (k & 1ull) r = poly_mul(r, b); k >>= 1; if (k) b = poly_mul(b, b); }
	return r;
}
// the state k draws later
inline State jumped(const State& s, const Poly& p) {
	uint32_t y[2 * DEG - 1];
	for (int j = 0; j < DEG; j++) y[j] = s.x[j];
	for (int j = DEG; j < 2 * DEG - 1; j++) y[j] = y[j - DEG] + y[j - SEP];
	State o;
	for (int j = 0; j < DEG; j++) { uint32_t a = 0; for (int i = 0; i < DEG; i++) a += p.c[i] * y[i + j]; o.x[j] = a; }
	return o;
}
inline State jumped(const State& s, unsigned long long k) { return jumped(s, poly_pow_t(k)); }

// The jump table of one `per` (draws of one rank): table[level][i][d] = coefficient i of t^(per * 64^level * d) — the digit in the fastest
// index: a wave whose lanes hold consecutive ranks loads a coefficient of its lowest digit as one 256-byte piece.
inline std::vector<uint32_t> jump_table(unsigned long long per) {
	std::vector<uint32_t> t((size_t)LEVELS * DEG * DIGITS);
	Poly unit = poly_pow_t(per);
	for (int lv = 0; lv < LEVELS; lv++) {
		Poly p = poly_one();
		for (int d = 0; d < DIGITS; d++) {
			for (int i = 0; i < DEG; i++) t[((size_t)lv * DEG + i) * DIGITS + d] = p.c[i];
			p = poly_mul(p, unit);
		}
		unit = p;                                                           // unit^64
	}
	return t;
}

// ---- the C library's own generator state: read it, and put one back --------------------------------------------------------------
// setstate() returns the state array in use with its first word = 5 * (rptr - state) + type (glibc random_r.c __setstate_r); the 31 words behind
// it are the table.  libc_state() leaves the library's generator exactly as it was.  Returns false when the library is not running TYPE_3 (a
// host that called initstate() with a small buffer): the caller then has no device stream to offer and says so.
inline bool libc_state(State& out) {
	static int32_t parked[34];
	static bool parked_init = false;
	if (!parked_init) { char* prev = initstate(1u, (char*)parked, 128); setstate(prev); parked_init = true; }
	char* cur = setstate((char*)parked);              // the library now runs on `parked`; `cur` = the caller's state array, header word filled in
	if (!cur) return false;
	const int32_t* w = (const int32_t*)cur;
	const int type = w[0] % 5, rear = w[0] / 5;
	bool ok = type == 3 && rear >= 0 && rear < DEG;
	if (ok) { const int f = (rear + SEP) % DEG; for (int j = 0; j < DEG; j++) out.x[j] = (uint32_t)w[1 + (f + j) % DEG]; }
	setstate(cur);
	return ok;
}
// make the library continue from `s` (its state array becomes a buffer of ours that lives as long as the process)
inline void libc_set_state(const State& s) {
	static int32_t ours[34], scratch[34];
	static bool scratch_init = false;
	if (!scratch_init) { char* prev = initstate(1u, (char*)scratch, 128); setstate(prev); scratch_init = true; }
	setstate((char*)scratch);                          // (the library may be running on `ours` from an earlier call: move it off before filling it)
	ours[0] = 3;                                       // TYPE_3, rptr = &state[0] -> fptr = &state[3] holds the oldest word
	for (int j = 0; j < DEG; j++) ours[1 + (SEP + j) % DEG] = (int32_t)s.x[j];
	setstate((char*)ours);
}
#endif  // !__HIPCC_RTC__

}  // namespace klg_rand
