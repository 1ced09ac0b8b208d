// klang_amd/csrc/klg_kernels.hpp — the synth-bank kernels (gfx950, wave64).
//
//   klg_apply_events<P>  one lane per voice that received events this block: note-on = overwrite the lane's
//                        record with the host-computed one; note-off = the patch's off() on the resident state.
//   klg_render<P,PV>     THE hot kernel.  One lane per voice, 256 voices per workgroup (4 waves).  Each lane loads
//                        its record (coalesced SoA planes), runs the patch body for n samples with all state in
//                        registers, and writes back only the words that change.  Voice mix: every 32 samples a wave
//                        transposes its 64x32 outputs through a padded LDS tile (stride 65 dwords: conflict-free
//                        ds_write_b32 / ds_read_b32), each lane sums 32 voices for one sample, the two half-sums are
//                        combined with one cross-lane exchange and added to the WAVE's own [n] row in LDS (dynamic LDS:
//                        WAVES x n floats, so only the wave itself ever touches its row: no atomics, a fixed order).  The
//                        workgroup then adds its four rows in wave order and writes its partial [n] row: the mix is
//                        bit-reproducible from run to run.
//                        A patch whose notes have a STEREO `out` (P::kStereo, Stereo::Note klang.h:4721-4733) renders the same way with
//                        two samples per voice and sample: the tile's 32 rows are 16 samples x 2 channels, every wave owns two mix
//                        rows, a workgroup writes two partial rows, per-voice output is [voice][2][n].
//   klg_reduce           partial rows -> the stereo block (ADDED to the destination, like the reference's `+=`); klg_reduce_stereo
//                        for banks of stereo notes (left rows to the left channel, right rows to the right).
//
// Workgroups are independent (no inter-workgroup communication inside a launch); groups of 256 voices are dealt
// round-robin to workgroups (group g -> block g % gridDim.x), which with the observed block->XCD mapping spreads
// consecutive voice groups over the 8 XCDs; nothing depends on that placement.
#pragma once
#include "klg_patches.hpp"

#pragma clang fp contract(off)

namespace klg {

enum { WG = 256, WAVES = 4, CHUNK = 32, TILE_LD = 65, MAX_BLOCK = 1024 };
static_assert(CHUNK <= KLG_CHUNK_MAX, "a patch's quiet() looks KLG_CHUNK_MAX samples ahead");

// Events: runs[r] = {voice, first, count}; ev_type[e] (0 = note-on with payload ev_payload[e], 1 = note-off);
// payload records are AoS [k][W] words as produced by the host-side on() code.
struct EventArgs {
	uint32_t* state; size_t stride;
	const int* run_voice; const int* run_first; const int* run_count; int runs;
	const int* ev_type; const int* ev_payload; const uint32_t* payload;
	float fs;
};

struct RenderArgs {
	uint32_t* state;          // [W][stride]
	size_t stride;            // voices rounded up to WG
	int voices, notes_per_synth, n;
	const float* controls;    // [synths][KLG_MAX_CTL]
	SampleRate fs;
	float* partials;          // [gridDim.x][n]
	float* per_voice;         // [voices][n] or null
	const TableDesc* tables;  // [tables] or null (klg_table_upload)
	float* rings;             // note delays: [stride + 1][ring_rows] (the last line is the dead lanes' scratch) — each voice's lines contiguous (a generated patch's Delay members), or null
	size_t ring_rows;         // ring positions per voice = the sum of the patch's Delay SIZEs
	const int* solo;          // KLG_MIX_LAST_ACTIVE: [synths] the one voice of each instance that is heard this block (-1: none), else null
	const int* rand;          // Noise generators of a generated patch: this block's rand() values [n * draws][rstride ranks], produced on the device in the reference's call order (klg_rand_dev.hpp)
	const int* rand_base;     // ... [voices] a sounding voice's rank = its column (lanes without a sounding voice read column 0), else null
	size_t rstride;
	// ONE LAUNCH PER BLOCK for banks of a few workgroups (a plugin's own synth: <= 128 notes, klang.h:4311), where three more launches cost as much as
	// the render itself: (1) ev.runs > 0 — every workgroup first applies the block's event runs of ITS OWN voices (what klg_apply_events does as
	// a launch); (2) ticket != null — the workgroup that finishes last adds all partial rows to `mix` in row order (what klg_reduce does as a
	// launch; the order is fixed, so the mix stays bit-reproducible).  Both null / 0: the separate launches.
	EventArgs ev;
	float* mix; int mix_channels; unsigned* ticket;
};

// record <-> word planes.  Words are moved with static indices only (fully unrolled) and converted with
// memcpy, which SROA turns into plain register moves: the record never touches scratch.
template<class REC> struct RecWords {
	static constexpr int W = sizeof(REC) / 4;
	uint32_t w[W];
	__device__ __forceinline__ void to(REC& r) const { __builtin_memcpy(&r, w, sizeof(REC)); }
	__device__ __forceinline__ void from(const REC& r) { __builtin_memcpy(w, &r, sizeof(REC)); }
};

// all lanes of the wave have written / may read the wave's LDS tile
__device__ __forceinline__ void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Dynamic LDS of the render kernels: [WAVES][n] floats, one mix row per wave (render_lds_bytes(n) at launch); [WAVES][2][n] for stereo notes.
// `keep = (this lane's sample == i) ? v : keep` for a recurrence every lane of a voice walks, called for i = 0 .. SLOTS - 1 in order (a whole tile): the lane's sample is
// lane % SLOTS (klg_render_gsp).  On a wave that has its SIMD to itself every instruction — scalar ones too — is an issue slot, so the select is not written as a
// compare and a select: with 64 or 16 samples per voice the values are PUSHED through the voice's lanes like a shift register, one DPP move a sample (wave_shl:1 /
// row_shl:1 — lane k takes lane k + 1's, the voice's last lane the new value: after SLOTS pushes lane k holds the k-th; klg_render_sub2a_sp.hpp's wave_push); with 32
// or 8 — no DPP shift of that width — WHICH lanes take the value is a constant of the unrolled loop, a 64-bit scalar mask.  All lanes of the wave must be active.
template<int SLOTS> __device__ __forceinline__ float sp_keep(float keep, float v, int i) {
	static_assert(SLOTS == 64 || SLOTS == 32 || SLOTS == 16 || SLOTS == 8, "samples per voice and tile");
	if constexpr (SLOTS == 64) return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(keep), 0x130, 0xF, 0xF, false));
	else if constexpr (SLOTS == 16) return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(keep), 0x101, 0xF, 0xF, false));
	else {
		constexpr unsigned long long rep = SLOTS == 32 ? 0x0000000100000001ull : 0x0101010101010101ull;
		const unsigned long long m = rep << i;
		float r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(keep), "v"(v), "s"(m));
		return r;
	}
}

extern __shared__ float klg_mix_rows[];
__device__ __forceinline__ float mix_rows_sum(int i, int n) { return ((klg_mix_rows[i] + klg_mix_rows[n + i]) + klg_mix_rows[2 * n + i]) + klg_mix_rows[3 * n + i]; }   // fixed order
__host__ __device__ inline unsigned render_lds_bytes(int n, int note_channels = 1) { return (unsigned)(WAVES * n * note_channels * sizeof(float)); }

// records longer than 64 words (generated graph patches, klg_graph.hpp) name their second store mask kStoreMask2
template<class...> using klg_void_t = void;
template<class P, class = void> struct StoreMask2 { static constexpr uint64_t value = 0; };
template<class P> struct StoreMask2<P, klg_void_t<decltype(P::kStoreMask2)>> { static constexpr uint64_t value = P::kStoreMask2; };
// ... and records longer than 128 words a function over one mask per 64 words (stores_word)
template<class P, class = void> struct StoresWord { static __device__ __forceinline__ constexpr bool at(int w) { return w < 64 ? ((P::kStoreMask >> (w & 63)) & 1ull) != 0 : ((StoreMask2<P>::value >> (w & 63)) & 1ull) != 0; } };
template<class P> struct StoresWord<P, klg_void_t<decltype(P::stores_word(0))>> { static __device__ __forceinline__ constexpr bool at(int w) { return P::stores_word(w); } };
template<class P> __device__ __forceinline__ constexpr bool patch_stores(int w) { return StoresWord<P>::at(w); }

// a patch may pin its occupancy with `static constexpr int kWavesPerEu` (measured per patch: SuperSaw renders 8 % faster at 4
// waves/SIMD with ~10 spilled registers than at the 3 the allocator picks on its own; FM and sub2 do not)
template<class P, class = void> struct WavesPerEu { static constexpr int lo = 1, hi = 8; };
template<class P> struct WavesPerEu<P, klg_void_t<decltype(P::kWavesPerEu)>> { static constexpr int lo = P::kWavesPerEu, hi = P::kWavesPerEu; };

// a patch may offer a second body for chunks in which every envelope of the wave merely holds: `static constexpr bool kHasQuiet`,
// `quiet(L)` (wave-uniform) and `sample_quiet(L, ctx)` (generated graph patches do, klg_graph.hpp)
template<class P, class = void> struct HasQuiet { static constexpr bool value = false; };
template<class P> struct HasQuiet<P, klg_void_t<decltype(P::kHasQuiet)>> { static constexpr bool value = P::kHasQuiet; };

// a patch may render TWO consecutive samples of an event-free chunk at once (`static constexpr bool kHasFast2`, `f2 sample_fast2(L, ctx)`:
// packed operations across the sample pair, PatchFM)
template<class P, class = void> struct HasFast2 { static constexpr bool value = false; };
template<class P> struct HasFast2<P, klg_void_t<decltype(P::kHasFast2)>> { static constexpr bool value = P::kHasFast2; };

// a patch with Noise generators: `static constexpr int kNoiseDraws` = rand() draws per sample (generated patches)
template<class P, class = void> struct NoiseDraws { static constexpr int value = 0; };
template<class P> struct NoiseDraws<P, klg_void_t<decltype(P::kNoiseDraws)>> { static constexpr int value = P::kNoiseDraws; };

// a patch whose sample() returns both channels of a Stereo::Note's `out` (Out2): `static constexpr bool kStereo = true`
template<class P, class = void> struct IsStereo { static constexpr bool value = false; };
template<class P> struct IsStereo<P, klg_void_t<decltype(P::kStereo)>> { static constexpr bool value = P::kStereo; };

// ---- the fused launch (RenderArgs::ev / ticket) ----
// (1) events: group g of `voices_per_group` voices is rendered by workgroup g % gridDim.x; that workgroup applies the runs of those voices first.
//     A workgroup's stores are seen by its own later loads (one CU, one L1, __syncthreads() between them); no other workgroup touches these records.
template<class P> __device__ __forceinline__ void apply_event_run(const EventArgs& a, int r);
template<class P> __device__ __forceinline__ void fused_events(const RenderArgs& a, int voices_per_group) {
	if (a.ev.runs == 0) return;
	for (int r = threadIdx.x; r < a.ev.runs; r += blockDim.x) {
		const int v = a.ev.run_voice[r];
		if ((v / voices_per_group) % (int)gridDim.x == (int)blockIdx.x) apply_event_run<P>(a.ev, r);
	}
	__syncthreads();
}
// (2) the combine: every workgroup has written its partial row ([n * nc] floats); the one that draws the last ticket adds the rows to the mix.
//     The hand-off is the guide's counter form (cdna_hip_programming.md §6 G16): all stores drained, ONE agent-scope release, then the relaxed
//     ticket; the last arriver takes ONE agent-scope acquire and reads the rows with plain loads — correct wherever the workgroups ran (other CUs
//     do not see this CU's L1, XCDs do not share an L2).  Rows are added in row order whoever is last: the mix is bit-reproducible.
__device__ __forceinline__ void fused_combine(const RenderArgs& a, int n, int nc, int* lds_flag) {
	if (!a.ticket) return;
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		const unsigned prev = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const int last = prev == gridDim.x - 1u;
		if (last) { __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }   // (the next block's launch finds the counter at 0)
		*lds_flag = last;
	}
	__syncthreads();
	if (!*lds_flag) return;
	const int rows = (int)gridDim.x, len = n * nc;
	for (int i = threadIdx.x; i < len; i += blockDim.x) {
		// (the rows were written on other XCDs: behind the acquire every load is a trip to memory.  Thirty-two under way at a time, added in row order — one at a
		//  time it was ~0.23 us a row: a SuperSaw bank of 4,096 voices, 128 rows, 51 us per block against 20 for its render alone)
		float t = 0.f;
		int r = 0;
		for (; r + 32 <= rows; r += 32) {
			float x[32];
#pragma unroll
			for (int q = 0; q < 32; q++) x[q] = a.partials[(size_t)(r + q) * len + i];
#pragma unroll
			for (int q = 0; q < 32; q++) t += x[q];
		}
		for (; r + 8 <= rows; r += 8) {
			float x[8];
#pragma unroll
			for (int q = 0; q < 8; q++) x[q] = a.partials[(size_t)(r + q) * len + i];
#pragma unroll
			for (int q = 0; q < 8; q++) t += x[q];
		}
		for (; r < rows; r++) t += a.partials[(size_t)r * len + i];
		if (nc == 2) { if (i / n < a.mix_channels) a.mix[i] += t; }                                      // stereo notes: rows are [2][n], left to left, right to right
		else for (int c = 0; c < a.mix_channels; c++) a.mix[(size_t)c * n + i] += t;              // a mono `out` goes to every channel (klg_reduce)
	}
}

template<class P, bool PER_VOICE>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(WavesPerEu<P>::lo, WavesPerEu<P>::hi))) void klg_render(const RenderArgs a) {
	using Rec = typename P::Rec;
	constexpr bool ST = IsStereo<P>::value;                  // stereo notes: tile rows = (channel, sample of a 16-sample chunk)
	constexpr int NC = ST ? 2 : 1, CH = ST ? CHUNK / 2 : CHUNK;
	constexpr int W = sizeof(Rec) / 4;
	__shared__ float lds[WAVES * CHUNK * TILE_LD];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	float* tile = lds + wave * CHUNK * TILE_LD;
	const int n = a.n;
	float* acc = klg_mix_rows + wave * n * NC;               // this wave's own mix row(s): [channel][n]
	// Noise generators: the draws of KLG_NZ_GROUP samples at a time, from the block's [index][rank] array into the wave's [index][lane] LDS copy — the group's
	// loads are all under way together (one memory round trip per group; read one by one where they are used, each sample waits out its own)
	constexpr int ND = NoiseDraws<P>::value;
	__shared__ int nz_lds[ND > 0 ? WAVES * KLG_NZ_GROUP * ND * 64 : 1];
	static_assert(CH % KLG_NZ_GROUP == 0, "a chunk is a whole number of Noise groups");

	for (int i = lane; i < n * NC; i += 64) acc[i] = 0.f;
	wave_sync();
	fused_events<P>(a, WG);

	const int groups = (int)(a.stride / WG);
	for (int g = blockIdx.x; g < groups; g += gridDim.x) {
		const int v = g * WG + tid;
		uint32_t flags = (v < a.voices) ? a.state[v] : (uint32_t)ST_OFF;
		const bool live = (flags & 3u) != (uint32_t)ST_OFF;
		// the mono Synth of the reference lets each sounding note overwrite the block in turn (klang.h:4299, 4450-4457): only the last one is heard
		const bool audible = live && (!a.solo || a.solo[v / a.notes_per_synth] == v);
		const bool in_tile = PER_VOICE ? live : audible;         // the per-voice dump wants every voice's own samples: there the selection happens in the sum
		const unsigned long long heard = __ballot(audible);
		if (__ballot(live) == 0ull) {                        // whole wave silent: nothing to add
			if (PER_VOICE) {
				const int v0 = g * WG + wave * 64;
				for (int j = 0; j < 64 && v0 + j < a.voices; j++)
					for (int i = lane; i < n * NC; i += 64) a.per_voice[(size_t)(v0 + j) * n * NC + i] = 0.f;
			}
			continue;
		}
		RecWords<Rec> rw;
		Rec rec;
		typename P::Live L;
		BlockCtx ctx;
		ctx.fs = a.fs;
		ctx.tables = a.tables;
		// this voice's lines, contiguous.  Dead lanes of a live wave run the body too (on an all-zero record): their ring accesses go to the
		// one scratch line behind the last voice's (line index `stride`), so an Off voice's own line keeps its contents like the reference's
		ctx.ring = a.rings ? a.rings + (size_t)(live ? (size_t)v : a.stride) * a.ring_rows : nullptr;
		ctx.ctl = a.controls + (size_t)((v < a.voices ? v : 0) / a.notes_per_synth) * KLG_MAX_CTL;
		ctx.rand = a.rand ? a.rand + (live ? a.rand_base[v] : 0) : nullptr; ctx.rstride = a.rstride;
		ctx.nz = nz_lds + wave * (KLG_NZ_GROUP * (ND > 0 ? ND : 1) * 64) + lane;
		int* const nz_mine = nz_lds + wave * (KLG_NZ_GROUP * (ND > 0 ? ND : 1) * 64) + lane;
		// (only this lane ever touches its column, LDS operations of a wave complete in order: no barrier.  The last group of a block reads up to
		//  KLG_NZ_GROUP - 1 samples' rows beyond the block: the array has that many spare rows, klg_api.hip note_prepass)
		auto nz_stage = [&](int sample) {
			if constexpr (ND > 0) {
				if ((sample & (KLG_NZ_GROUP - 1)) != 0) return;
				const int* src = ctx.rand + (size_t)sample * ND * a.rstride;
				int t[KLG_NZ_GROUP * ND];
#pragma unroll
				for (int i = 0; i < KLG_NZ_GROUP * ND; i++) t[i] = src[(size_t)i * a.rstride];
#pragma unroll
				for (int i = 0; i < KLG_NZ_GROUP * ND; i++) nz_mine[i * 64] = t[i];
			}
		};
		ctx.rec = a.state + v; ctx.stride = a.stride;             // (v < stride always: a lane without a voice points at padding or at an Off voice's words, and nothing it computes is kept)
		// Dead lanes of a live wave run the same instruction stream on an all-zero record (no per-sample exec
		// masking); their output is forced to 0 at the tile write and their record is never stored.
		rw.w[0] = live ? flags : (uint32_t)ST_OFF;                  // (a lane without a voice: an all-zero record whose note stage says Off)
#pragma unroll
		for (int w = 1; w < W; w++) rw.w[w] = live ? a.state[(size_t)w * a.stride + v] : 0u;
		rw.to(rec);
		P::begin(L, rec, ctx);
		for (int c0 = 0; c0 < n; c0 += CH) {
			const int cl = (n - c0 < CH) ? (n - c0) : CH;
			int quiet = 0;                                          // 0: full body, 1: envelopes holding, 2: ... and duty-0 saws (generated patches)
			if constexpr (HasQuiet<P>::value) quiet = P::quiet(L);
			// one sample into the tile: row s (mono) / rows s and CH + s (stereo: left, right)
			auto put = [&](int s, const auto y) {
				if constexpr (ST) { tile[s * TILE_LD + lane] = in_tile ? y.l : 0.f; tile[(CH + s) * TILE_LD + lane] = in_tile ? y.r : 0.f; }
				else tile[s * TILE_LD + lane] = in_tile ? y : 0.f;
			};
			if (quiet == 2) {
				if constexpr (HasFast2<P>::value) {
					int s = 0;
					if constexpr (P::kFastN > 2) {
						constexpr int PAIRS = P::kFastN / 2;
						for (; s + 2 * PAIRS <= cl; s += 2 * PAIRS) {
							f2 y[PAIRS]; P::template sample_fast_pairs<PAIRS>(L, y);
#pragma unroll
							for (int j = 0; j < PAIRS; j++) { put(s + 2 * j, y[j].x); put(s + 2 * j + 1, y[j].y); }
						}
					}
					for (; s + 1 < cl; s += 2) { const f2 y = P::sample_fast2(L, ctx); put(s, y.x); put(s + 1, y.y); }
					if (s < cl) put(s, P::sample_fast(L, ctx));
				}
				else if constexpr (HasQuiet<P>::value)
					for (int s = 0; s < cl; s++) { nz_stage(c0 + s); put(s, P::sample_fast(L, ctx)); }
			}
			else if (quiet == 1) {
				if constexpr (HasQuiet<P>::value)
					for (int s = 0; s < cl; s++) { nz_stage(c0 + s); put(s, P::sample_quiet(L, ctx)); }
			}
			else for (int s = 0; s < cl; s++) { nz_stage(c0 + s); put(s, P::sample(L, ctx)); }
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			// lane -> (tile row r, half h of the 64 voices); row r holds sample s of channel c
			const int r = lane & 31, h = lane >> 5, s = ST ? (r & (CH - 1)) : r, c = ST ? (r >> 4) : 0;
			static_assert(!ST || CH == 16, "stereo rows are addressed as channel = row >> 4");
			if (PER_VOICE) {
				const int v0 = g * WG + wave * 64;
				for (int j = 0; j < 32; j++) {
					const int jj = 2 * j + h;
					if (s < cl && v0 + jj < a.voices) a.per_voice[((size_t)(v0 + jj) * NC + c) * n + c0 + s] = tile[r * TILE_LD + jj];
				}
			}
			{
				float sum = 0.f;
				if (s < cl) {
					const float* row = tile + r * TILE_LD + h * 32;
					if (PER_VOICE && a.solo) { for (int j = 0; j < 32; j++) sum += ((heard >> (h * 32 + j)) & 1ull) ? row[j] : 0.f; }
					else {
#pragma unroll
						for (int j = 0; j < 32; j++) sum += row[j];
					}
				}
				sum += __shfl_xor(sum, 32);
				if (lane < 32 && s < cl) acc[c * n + c0 + s] += sum;    // the wave's own row(s): program order, no atomics
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
		if (live) {
			P::end(L, rec);
			rw.from(rec);
#pragma unroll
			for (int w = 0; w < W; w++)
				if (patch_stores<P>(w)) a.state[(size_t)w * a.stride + v] = rw.w[w];
		}
	}
	__syncthreads();
	// one partial row per workgroup ([2][n] for stereo notes: the waves' rows are [channel][n], so element i of a wave's block is the same (channel, sample) in all four)
	for (int i = tid; i < n * NC; i += WG) a.partials[(size_t)blockIdx.x * n * NC + i] = mix_rows_sum(i, n * NC);
	fused_combine(a, n, NC, reinterpret_cast<int*>(lds));
}

// KLG_MIX_LAST_ACTIVE: the highest-numbered sounding note of every synth instance (the one whose block survives in the reference's
// mono Synth::process, klang.h:4450-4457), after this block's events have been applied
__global__ void klg_select_last_active(const uint32_t* __restrict__ flags, int synths, int notes_per_synth, int* __restrict__ solo) {
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= synths) return;
	int sel = -1;
	for (int p = 0; p < notes_per_synth; p++) if ((flags[(size_t)s * notes_per_synth + p] & 3u) != (uint32_t)ST_OFF) sel = s * notes_per_synth + p;
	solo[s] = sel;
}

// partial rows [rows][n] -> mix[c][i] += sum  (Stereo::Mono::Note: L += out; R += out, klang.h:4751-4752).
// 1024 threads = 32 samples x 32 row partitions; fixed summation order (deterministic, no float atomics).
__global__ __launch_bounds__(1024) void klg_reduce(const float* __restrict__ partials, int rows, int n, float* mix, int channels) {
	__shared__ float part[32][33];
	const int sl = threadIdx.x & 31, p = threadIdx.x >> 5;
	const int s = blockIdx.x * 32 + sl;
	float sum = 0.f;
	if (s < n) {
#pragma unroll 8
		for (int r = p; r < rows; r += 32) sum += partials[(size_t)r * n + s];       // 8 independent loads in flight
	}
	part[p][sl] = sum;
	__syncthreads();
	if (p == 0 && s < n) {
		float t = 0.f;
#pragma unroll
		for (int k = 0; k < 32; k++) t += part[k][sl];
		for (int c = 0; c < channels; c++) mix[(size_t)c * n + s] += t;
	}
}

// banks of stereo notes: partial rows [rows][2][n] -> mix[c][i] += sum of channel c's rows (`buffer++ += out` on an {l, r} frame, klang.h:4731);
// with channels == 1 (a mono destination) only the left rows are taken
__global__ __launch_bounds__(1024) void klg_reduce_stereo(const float* __restrict__ partials, int rows, int n, float* mix, int channels) {
	__shared__ float part[32][33];
	const int sl = threadIdx.x & 31, p = threadIdx.x >> 5;
	const int s = blockIdx.x * 32 + sl, c = blockIdx.y;
	float sum = 0.f;
	if (s < n) {
#pragma unroll 8
		for (int r = p; r < rows; r += 32) sum += partials[((size_t)r * 2 + c) * n + s];
	}
	part[p][sl] = sum;
	__syncthreads();
	if (p == 0 && s < n && c < channels) {
		float t = 0.f;
#pragma unroll
		for (int k = 0; k < 32; k++) t += part[k][sl];
		mix[(size_t)c * n + s] += t;
	}
}

template<class P> __device__ __forceinline__ void apply_event_run(const EventArgs& a, int r);
template<class P>
__global__ __launch_bounds__(64) void klg_apply_events(const EventArgs a) {
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= a.runs) return;
	apply_event_run<P>(a, r);
}
// one run = the events one voice received this block, in order
template<class P> __device__ __forceinline__ void apply_event_run(const EventArgs& a, int r) {
	using Rec = typename P::Rec;
	constexpr int W = sizeof(Rec) / 4;
	const int v = a.run_voice[r];
	RecWords<Rec> rw;
	bool loaded = false;
	for (int e = a.run_first[r]; e < a.run_first[r] + a.run_count[r]; e++) {
		if (a.ev_type[e] == 0) {
			const uint32_t* src = a.payload + (size_t)a.ev_payload[e] * W;
#pragma unroll
			for (int w = 0; w < W; w++) rw.w[w] = src[w];
			loaded = true;
		}
		else {
			if (!loaded) {
#pragma unroll
				for (int w = 0; w < W; w++) rw.w[w] = a.state[(size_t)w * a.stride + v];
				loaded = true;
			}
			if ((rw.w[0] & 3u) == (uint32_t)ST_SUSTAIN) {               // Synth::noteOff only releases Sustain notes (klang.h:4432)
				Rec rec; rw.to(rec); P::release(rec, a.fs); rw.from(rec);
			}
		}
	}
	if (loaded) {
#pragma unroll
		for (int w = 0; w < W; w++) a.state[(size_t)w * a.stride + v] = rw.w[w];
	}
}

// Record uploads of patches without a device-side off() (graph patches): the last record queued for a voice wins.
__global__ __launch_bounds__(64) void klg_apply_records(const EventArgs a, int W) {
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= a.runs) return;
	const int v = a.run_voice[r];
	int last = -1;
	for (int e = a.run_first[r]; e < a.run_first[r] + a.run_count[r]; e++) if (a.ev_type[e] == 0) last = a.ev_payload[e];
	if (last < 0) return;
	const uint32_t* src = a.payload + (size_t)last * W;
	for (int w = 0; w < W; w++) a.state[(size_t)w * a.stride + v] = src[w];
}

// single-voice record copy (klg_voice_download / klg_voice_upload)
__global__ void klg_copy_record(uint32_t* state, size_t stride, int v, uint32_t* rec, int W, int to_state) {
	for (int w = threadIdx.x; w < W; w += blockDim.x) { if (to_state) state[(size_t)w * stride + v] = rec[w]; else rec[w] = state[(size_t)w * stride + v]; }
}

} // namespace klg
