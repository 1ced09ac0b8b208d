// klg_rand_dev.hpp — the Noise generators' rand() stream produced ON THE DEVICE (kernels + the host-side bookkeeping of where the stream lives).
//
// Replaces: Generators::Basic::Noise::process() / Fast::Noise::process() calling libc rand() once per sample (klang.h:4947-4951, 5357-5366), all
// notes of all Synths of a process drawing from the one sequence in the order Synth::process walks them (klang.h:4842-4848).  klg_rand.hpp restates
// the generator and its jump-ahead; here:
//   klg_rand_rank   a synth bank's sounding voices -> their rank in that walk (prefix count over the note stages) and the number that draw
//   klg_rand_fill   rank r's `per` = n * draws values: a lane jumps from the block's start state to position r * per (one 31 x 31 product per non-zero
//                   base-64 digit of r, the coefficients from a table made on the host per `per`) and runs the recurrence from there.  Values are
//                   stored [index within the rank][rank]: a wave's 64 ranks write — and 64 voices later read — one 256-byte piece per access.
//   klg_rand_advance the stream's state moved past the block (count * per draws), in place, for the next block
//   klg_note_smooth controls[i].smooth() of a Note (klang.h:1715): see below
// and RngChain: the C library's generator and the device copy are ONE stream.  The first Noise block takes the library's state to the device; whoever
// draws on the host afterwards (klang::random in a patch's on(), Reverb.k's prepare(), the caller through klg_rand_sync) gets it back first.
#pragma once
#include "klg_rand.hpp"

namespace klg {

struct RandStateArg { uint32_t x[klg_rand::DEG]; };
__global__ void klg_rand_set(uint32_t* state, const RandStateArg s) { if (threadIdx.x < klg_rand::DEG) state[threadIdx.x] = s.x[threadIdx.x]; }

// ---- ranks --------------------------------------------------------------------------------------------------------------------------
// A voice draws when its note stage is not Off at the start of the block (after the block's events): Note::process(buffer) runs all n samples of a
// note that stop()s half way through (klang.h:4295-4303).
enum { RANK_WG = 1024 };
struct RankArgs {
	const uint32_t* flags; int voices;
	int* rank;                       // [voices]: a sounding voice's rank, 0 for the others
	unsigned* block_counts;          // [workgroups]: sounding voices per 1,024 (pass 1), their exclusive prefix (pass 2)
	unsigned* count;                 // the number of voices that draw
	unsigned long long* feedback;    // pinned host word: (seq << 32) | count — the host's capacity planning reads it a block or more later, never waits for it
	unsigned seq;
};
__device__ __forceinline__ unsigned rank_wg_scan(bool on, unsigned& total) {        // exclusive rank of this thread among the workgroup's `on` threads
	__shared__ unsigned wave_count[RANK_WG / 64];
	const unsigned long long b = __ballot(on);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) wave_count[wave] = (unsigned)__popcll(b);
	__syncthreads();
	unsigned before = 0, all = 0;
	for (int w = 0; w < RANK_WG / 64; w++) { const unsigned c = wave_count[w]; if (w < wave) before += c; all += c; }
	total = all;
	return before + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
}
// pass 1 (banks of more than one workgroup): sounding voices per workgroup
__global__ __launch_bounds__(RANK_WG) void klg_rand_count(const RankArgs a) {
	const int v = blockIdx.x * RANK_WG + threadIdx.x;
	unsigned total = 0;
	(void)rank_wg_scan(v < a.voices && (a.flags[v] & 3u) != (uint32_t)ST_OFF, total);
	if (threadIdx.x == 0) a.block_counts[blockIdx.x] = total;
}
// pass 2: exclusive prefix over the workgroups' counts, in place (one workgroup)
__global__ __launch_bounds__(RANK_WG) void klg_rand_scan(const RankArgs a, int groups) {
	__shared__ unsigned carry;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (int g0 = 0; g0 < groups; g0 += RANK_WG) {
		const int g = g0 + threadIdx.x;
		const unsigned mine = g < groups ? a.block_counts[g] : 0u;
		// exclusive scan of `mine` over the workgroup: by waves, then across the 16 wave sums
		__shared__ unsigned wsum[RANK_WG / 64];
		unsigned x = mine;
		for (int d = 1; d < 64; d <<= 1) { const unsigned y = __shfl_up(x, d); if ((int)(threadIdx.x & 63) >= d) x += y; }
		if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
		__syncthreads();
		unsigned before = carry;
		for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += wsum[w];
		if (g < groups) a.block_counts[g] = before + x - mine;
		__syncthreads();
		if (threadIdx.x == RANK_WG - 1) carry = before + x;
		__syncthreads();
	}
	if (threadIdx.x == 0) { *a.count = carry; if (a.feedback) *a.feedback = ((unsigned long long)a.seq << 32) | (unsigned long long)carry; }
}
// pass 3 (the only one for a bank of <= 1,024 voices: `single`): every voice's rank
__global__ __launch_bounds__(RANK_WG) void klg_rand_rank(const RankArgs a, int single) {
	const int v = blockIdx.x * RANK_WG + threadIdx.x;
	const bool on = v < a.voices && (a.flags[v] & 3u) != (uint32_t)ST_OFF;
	unsigned total = 0;
	const unsigned r = rank_wg_scan(on, total);
	if (v < a.voices) a.rank[v] = on ? (int)((single ? 0u : a.block_counts[blockIdx.x]) + r) : 0;
	if (single && threadIdx.x == 0) { *a.count = total; if (a.feedback) *a.feedback = ((unsigned long long)a.seq << 32) | (unsigned long long)total; }
}

// ---- the draws ------------------------------------------------------------------------------------------------------------------------
struct RandFillArgs {
	const uint32_t* state;           // [31] the stream where the block starts (time order, klg_rand::State)
	const uint32_t* table;           // [LEVELS][31][64] klg_rand::jump_table(per)
	const unsigned* count;           // device: the ranks that draw (null: count_imm)
	unsigned count_imm;
	int per;                         // draws per rank = n * Noise generators per sample
	int* out; size_t rstride;        // out[index * rstride + rank] = rand()
};
// `x` (time order) -> the state d * 64^lv * per draws later, d this lane's digit: x[j] <- sum_i c_i y[i + j], y = x continued by 30 steps
__device__ __forceinline__ void rand_jump(uint32_t (&x)[klg_rand::DEG], const uint32_t* __restrict__ coef /* + digit, stride 64 */) {
	constexpr int D = klg_rand::DEG, S = klg_rand::SEP;
	uint32_t c[D], y[2 * D - 1];
#pragma unroll
	for (int i = 0; i < D; i++) c[i] = coef[i * klg_rand::DIGITS];
#pragma unroll
	for (int j = 0; j < D; j++) y[j] = x[j];
#pragma unroll
	for (int j = D; j < 2 * D - 1; j++) y[j] = y[j - D] + y[j - S];
#pragma unroll
	for (int j = 0; j < D; j++) {
		uint32_t acc = 0;
#pragma unroll
		for (int i = 0; i < D; i++) acc += c[i] * y[i + j];
		x[j] = acc;
	}
}
__global__ __launch_bounds__(256) void klg_rand_fill(const RandFillArgs a) {
	constexpr int D = klg_rand::DEG, S = klg_rand::SEP;
	const unsigned count = a.count ? *a.count : a.count_imm;
	const unsigned r = blockIdx.x * 256u + threadIdx.x;
	if ((r & ~63u) >= count) return;                                   // (the whole wave)
	uint32_t x[D];
#pragma unroll
	for (int j = 0; j < D; j++) x[j] = a.state[j];
	for (int lv = klg_rand::LEVELS - 1; lv >= 0; lv--) {
		const unsigned d = (r >> (6 * lv)) & 63u;
		if (__ballot(d != 0u) == 0ull) continue;                       // (digit 0 = the polynomial 1: a wave whose lanes all hold it skips the product)
		rand_jump(x, a.table + (size_t)lv * D * klg_rand::DIGITS + d);
	}
	if (r >= count) return;
	int* o = a.out + r;
	for (int i0 = 0; i0 < a.per; i0 += D) {
#pragma unroll
		for (int j = 0; j < D; j++) {                                  // one turn of the ring: x[j] is the oldest word when step j comes
			x[j] += x[(j + D - S) % D];
			if (i0 + j < a.per) o[(size_t)(i0 + j) * a.rstride] = (int)(x[j] >> 1);
		}
	}
}
// the stream's state past the block: count * per draws on (one wave: lane j computes word j)
__global__ __launch_bounds__(64) void klg_rand_advance(uint32_t* state, const uint32_t* table, const unsigned* count, unsigned count_imm) {
	constexpr int D = klg_rand::DEG, S = klg_rand::SEP;
	__shared__ uint32_t y[2 * D - 1];
	const unsigned r = count ? *count : count_imm;
	const int j = threadIdx.x;
	uint32_t xj = j < D ? state[j] : 0u;
	for (int lv = klg_rand::LEVELS - 1; lv >= 0; lv--) {
		const unsigned d = (r >> (6 * lv)) & 63u;
		if (d == 0u) continue;
		if (j < D) y[j] = xj;
		__syncthreads();
		if (j == 0) for (int k = D; k < 2 * D - 1; k++) y[k] = y[k - D] + y[k - S];
		__syncthreads();
		if (j < D) { uint32_t acc = 0; for (int i = 0; i < D; i++) acc += table[((size_t)lv * D + i) * klg_rand::DIGITS + d] * y[i + j]; xj = acc; }
		__syncthreads();
	}
	if (j < D) state[j] = xj;
}

// ---- controls[i].smooth() inside a Note (klang.h:1715) -----------------------------------------------------------------------------------------
// The control is the Synth's: every sounding note advances it, one note after the other, each through its whole block (klang.h:4842-4848).  The chain
// smoothed = smoothed * 0.999f + (1.f - 0.999f) * value therefore runs THROUGH the sounding notes of an instance: a lane per instance walks them,
// gives each voice's record the value ITS block starts from (the voice's lane repeats the same operations per sample), and stops early at the chain's
// fp32 fixed point, where a step changes nothing.
struct SmoothArgs {
	uint32_t* state; size_t stride; int synths, notes_per_synth, n, nctl;
	const float* controls;           // [synths][KLG_MAX_CTL]
	float* smoothed;                 // [synths][nctl] Control::smoothed
	int count; struct { int word, ctl, calls; } sm[KLG_MAX_CTL];
};
__global__ __launch_bounds__(64) void klg_note_smooth(const SmoothArgs a) {
	const int i = blockIdx.x * 64 + threadIdx.x;
	if (i >= a.synths) return;
	for (int q = 0; q < a.count; q++) {
		float x = a.smoothed[(size_t)i * a.nctl + a.sm[q].ctl];
		const float k = (1.f - 0.999f) * a.controls[(size_t)i * KLG_MAX_CTL + a.sm[q].ctl];
		uint32_t* const word = a.state + (size_t)a.sm[q].word * a.stride;
		const long long steps = (long long)a.n * a.sm[q].calls;
		for (int v = i * a.notes_per_synth; v < (i + 1) * a.notes_per_synth; v++) {
			word[v] = __float_as_uint(x);
			if ((a.state[v] & 3u) == (uint32_t)ST_OFF) continue;
			for (long long t = steps; t > 0; t--) { const float y = x * 0.999f + k; if (y == x) break; x = y; }
		}
		a.smoothed[(size_t)i * a.nctl + a.sm[q].ctl] = x;
	}
}

}  // namespace klg
