// tools/verify_rand.cpp — klang_amd/csrc/klg_rand.hpp against the C library's own srand() / rand() (what klang::random(seed) and the Noise
// generators call: klang.h:236-240, 4949, 5363).  Checks, for several seeds:
//   1. seeded(seed) + step() >> 1  ==  srand(seed); rand()           over N draws (default 10^8)
//   2. jumped(state, k) == the state reached by k steps               for k up to 10^7 and random k, and through jump_table() digit by digit
//   3. libc_state() reads the library's generator without disturbing it, libc_set_state() makes the library continue from a given state
// Build: g++ -O2 -std=c++17 tools/verify_rand.cpp -o /tmp/verify_rand ; run: /tmp/verify_rand [draws]
#include <cstdio>
#include <cstdlib>
#include "../klang_amd/csrc/klg_rand.hpp"
using namespace klg_rand;
static bool same(const State& a, const State& b) { return std::memcmp(a.x, b.x, sizeof a.x) == 0; }
int main(int argc, char** argv) {
	const unsigned long long N = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
	const unsigned seeds[] = { 1u, 0u, 272839u, 42u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFFu, 2463534242u };
	unsigned long long bad = 0, checked = 0;
	for (unsigned seed : seeds) {
		srand(seed);
		State s = seeded(seed);
		for (unsigned long long i = 0; i < N; i++) { const int want = rand(); const int got = (int)(step(s) >> 1); if (got != want) { if (bad < 5) printf("seed %u draw %llu: %d != %d\n", seed, i, got, want); bad++; } }
		checked += N;
		// 3a. reading the library's state mid-stream: equal to ours, and the library is left untouched
		State lib; if (!libc_state(lib) || !same(lib, s)) { printf("seed %u: libc_state() differs from the restated state after %llu draws\n", seed, N); bad++; }
		for (int i = 0; i < 1000; i++) if ((int)(step(s) >> 1) != rand()) { bad++; break; }
	}
	printf("draws: %llu checked over %zu seeds, %llu mismatches\n", checked, sizeof seeds / sizeof seeds[0], bad);
	// 2. jump-ahead
	unsigned long long jbad = 0, jn = 0;
	{
		State s = seeded(272839u), walk = s;
		unsigned long long at = 0;
		const unsigned long long ks[] = { 0, 1, 2, 3, 30, 31, 32, 61, 62, 256, 257, 1000, 4096, 65536, 1000003, 10000000 };
		for (unsigned long long k : ks) { while (at < k) { step(walk); at++; } if (!same(jumped(s, k), walk)) { printf("jump by %llu differs\n", k); jbad++; } jn++; }
		// composition of big jumps: jump(a) then jump(b) == jump(a + b), for distances no walk reaches
		unsigned long long r = 88172645463325252ull;
		for (int i = 0; i < 200; i++) {
			r ^= r << 13; r ^= r >> 7; r ^= r << 17; const unsigned long long a = r >> 8; r ^= r << 13; r ^= r >> 7; r ^= r << 17; const unsigned long long b = r >> 8;
			if (!same(jumped(jumped(s, a), b), jumped(s, a + b))) { jbad++; } jn++;
		}
		// the table: rank = digits base 64, per = 256 * 3
		const unsigned long long per = 768;
		const std::vector<uint32_t> T = jump_table(per);
		const unsigned long long ranks[] = { 0, 1, 63, 64, 65, 4095, 4096, 100000, 16777215, 16777216, 1073741823ull };
		for (unsigned long long rank : ranks) {
			State t = s;
			for (int lv = LEVELS - 1; lv >= 0; lv--) { const int d = (int)((rank >> (6 * lv)) & 63); Poly p; for (int i = 0; i < DEG; i++) p.c[i] = T[((size_t)lv * DEG + i) * DIGITS + d]; t = jumped(t, p); }
			if (!same(t, jumped(s, rank * per))) { printf("table jump to rank %llu differs\n", rank); jbad++; } jn++;
		}
	}
	printf("jumps: %llu checked, %llu mismatches\n", jn, jbad);
	// 3b. libc_set_state(): the library continues from a state of ours
	unsigned long long sbad = 0;
	{
		State s = jumped(seeded(7u), 123456789ull);
		libc_set_state(s);
		for (int i = 0; i < 100000; i++) if ((int)(step(s) >> 1) != rand()) sbad++;
		State lib; if (!libc_state(lib) || !same(lib, s)) sbad++;
		libc_set_state(s);                                                        // a second time (the library is running on our buffer)
		for (int i = 0; i < 1000; i++) if ((int)(step(s) >> 1) != rand()) sbad++;
		srand(5u); State t = seeded(5u);                                         // and srand() afterwards behaves
		for (int i = 0; i < 1000; i++) if ((int)(step(t) >> 1) != rand()) sbad++;
	}
	printf("state round trip: %llu mismatches\n", sbad);
	return bad || jbad || sbad ? 1 : 0;
}
