// klang_amd/csrc/klg_render_lanes.hpp — SuperSaw.k with a voice spread over several lanes, for banks too small to fill the chip.
//
// klg_render<PatchSuperSaw> gives a lane a whole voice: seven six-case OSM tables, the `/ 7`s and the ADSR are ~380 instructions per
// sample, strictly in sequence — config 3's 16,384 voices are 256 waves, one per SIMD on a quarter of the chip, and a block takes the
// 0.19 ms that one wave needs for 256 x 380 instructions whatever else is idle.  Two kernels here spread a voice over lanes instead (A/B references since
// round 5: SuperSaw banks of every size run klg_render_supersaw_sp.hpp — samples side by side, the six-case table only where it is needed):
//   klg_render_supersaw_pairs<P>  (below; KLG_SUPERSAW_LANES=2) — an oscillator PAIR per lane, P samples
//                                 of a voice side by side: 37 us per block at 16,384 voices (voice per lane: 181)
//   klg_render_supersaw_lanes     (KLG_SUPERSAW_LANES=1; the first form, kept for A/B runs) — ONE oscillator per lane: 75 / 158 us
// klg_render_supersaw_lanes: lane = (voice vi of 8) x (slot k of 8): slot
// k < 7 runs oscillator k, all eight lanes of a voice carry a copy of its ADSR (a copy costs an issue slot nobody else wants), and
// `for s < 7: out += osc[s] / 7` (SuperSaw.k:28-29) — a sum in that order — is a running sum through the lanes: six v_add_f32 with a
// DPP row_shr:1 source, lane k taking lane k - 1's partial sum, ending in slot 6.  ~100 instructions per sample for 8 voices: the bank
// is eight times as many waves, each a quarter as long.  Same arithmetic, same order, same record layout as klg_render<PatchSuperSaw>,
// (KLG_SUPERSAW_LANES=0):
// tests/test_gpu_parity.py runs all of them against the golden vectors (KLG_SUPERSAW_LANES=0 / 1 / 2 forces the choice).
#pragma once
#include "klg_kernels.hpp"

namespace klg {

enum { KLG_LANES_VOICES_PER_WAVE = 8, KLG_LANES_VOICES_PER_WG = KLG_LANES_VOICES_PER_WAVE * WAVES };

template<bool PER_VOICE>
__global__ __launch_bounds__(WG) void klg_render_supersaw_lanes(const RenderArgs a) {
	using Rec = rec::SuperSaw;
	__shared__ float lds[WAVES * CHUNK * TILE_LD];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int vi = lane >> 3, k = lane & 7, ko = k < 7 ? k : 6;                 // (slot 7 runs a copy of oscillator 6; nobody takes its sum)
	float* tile = lds + wave * CHUNK * TILE_LD;
	const int n = a.n;
	float* acc = klg_mix_rows + wave * n;                                       // this wave's own mix row
	for (int i = lane; i < n; i += 64) acc[i] = 0.f;
	wave_sync();

	const int groups = (a.voices + KLG_LANES_VOICES_PER_WG - 1) / KLG_LANES_VOICES_PER_WG;
	for (int g = blockIdx.x; g < groups; g += gridDim.x) {
		const int v0 = g * KLG_LANES_VOICES_PER_WG + wave * KLG_LANES_VOICES_PER_WAVE, v = v0 + vi;
		const uint32_t flags = (v < a.voices) ? a.state[v] : (uint32_t)ST_OFF;
		const bool live = (flags & 3u) != (uint32_t)ST_OFF;
		const bool audible = live && (!a.solo || a.solo[v / a.notes_per_synth] == v);   // KLG_MIX_LAST_ACTIVE (see klg_render)
		const bool heard = (PER_VOICE ? live : audible) && k == 6;                  // the lane whose value is the voice's sample
		if (__ballot(live) == 0ull) {
			if (PER_VOICE) for (int j = 0; j < KLG_LANES_VOICES_PER_WAVE && v0 + j < a.voices; j++) for (int i = lane; i < n; i += 64) a.per_voice[(size_t)(v0 + j) * n + i] = 0.f;
			continue;
		}
		// ---- this lane's share of the record: its oscillator, the voice's ADSR ----
		auto word = [&](int w) { return live ? a.state[(size_t)w * a.stride + v] : 0u; };
		Osm o;
		{
			constexpr int O0 = offsetof(Rec, osc) / 4;
			OsmRec r; r.inc = (int32_t)word(O0 + 4 * ko); r.offset = word(O0 + 4 * ko + 1); r.duty = word(O0 + 4 * ko + 2); r.delta = u2f(word(O0 + 4 * ko + 3));
			osm_load(o, r, live ? KLG_FLAG_GET(flags, 8 + 2 * ko, 2) : 0u);
		}
		Adsr adsr;
		{
			constexpr int A0 = offsetof(Rec, adsr) / 4;
			AdsrRec r; r.r_out = u2f(word(A0)); r.r_target = u2f(word(A0 + 1)); r.r_rate = u2f(word(A0 + 2)); r.time = u2f(word(A0 + 3));
			r.A = u2f(word(A0 + 4)); r.AD = u2f(word(A0 + 5)); r.S = u2f(word(A0 + 6)); r.R = u2f(word(A0 + 7));
			adsr_load(adsr, r, live ? KLG_FLAG_GET(flags, 2, 6) : 0u);
		}
		int stage = live ? (int)(flags & 3u) : (int)ST_OFF;
		const float tinc = a.fs.timeInc;

		for (int c0 = 0; c0 < n; c0 += CHUNK) {
			const int cl = (n - c0 < CHUNK) ? (n - c0) : CHUNK;
			// the chunk's envelope work: none (holding), a glide (no event inside the chunk: env_safe), or the full Envelope::process
			float step, tstep;
			const bool safe = env_safe(adsr.e, adsr.e.point == 2, step, tstep, tinc);
			const bool glide = __ballot(stage != (int)ST_OFF && !safe) == 0ull;
			for (int s = 0; s < cl; s++) {
				const float x = div_const<0x40e00000u>(osm_saw(o));                  // osc[k] / 7   SuperSaw.k:29
				// out = 0; out += osc[0] / 7; out += osc[1] / 7; ...  — lane k's partial sum is lane k - 1's plus its own term
				float sum = 0.f + x;
#pragma unroll
				for (int j = 1; j < 7; j++) sum = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x111, 0xF, 0xF, true)) + x;   // row_shr:1
				float env;
				if (glide) env = env_glide(adsr.e, step, tstep);
				else { env = adsr_process(adsr, a.fs); stage = (adsr.e.stage == ENV_OFF) ? (int)ST_OFF : stage; }
				tile[s * TILE_LD + lane] = heard ? sum * env : 0.f;                  // out *= adsr++   SuperSaw.k:31
			}
			wave_sync();
			if (PER_VOICE) {
				const int s = lane & 31, h = lane >> 5;
				for (int j = h; j < KLG_LANES_VOICES_PER_WAVE; j += 2)
					if (s < cl && v0 + j < a.voices) a.per_voice[(size_t)(v0 + j) * n + c0 + s] = tile[s * TILE_LD + 8 * j + 6];
			}
			{
				const int s = lane & 31, h = lane >> 5;
				float part = 0.f;
				if (s < cl) {
					const float* row = tile + s * TILE_LD + h * 32;
					if (PER_VOICE && a.solo) { for (int j = 0; j < 4; j++) { const int vv = v0 + 4 * h + j; part += (vv < a.voices && a.solo[vv / a.notes_per_synth] == vv) ? row[8 * j + 6] : 0.f; } }
					else {
#pragma unroll
						for (int j = 0; j < 4; j++) part += row[8 * j + 6];                  // the four voices of this half of the wave, in order
					}
				}
				part += __shfl_xor(part, 32);
				if (lane < cl) acc[c0 + lane] += part;                                  // the wave's own row: program order, no atomics
			}
			wave_sync();
		}
		// ---- write back: every oscillator lane its phase, slot 6 the envelope and the flags (the state bits of all seven oscillators) ----
		uint32_t bits = (k < 7) ? ((uint32_t)o.state << (8 + 2 * k)) : 0u;
		bits |= (uint32_t)__shfl_xor((int)bits, 1); bits |= (uint32_t)__shfl_xor((int)bits, 2); bits |= (uint32_t)__shfl_xor((int)bits, 4);
		if (live) {
			constexpr int O0 = offsetof(Rec, osc) / 4, A0 = offsetof(Rec, adsr) / 4;
			if (k < 7) a.state[(size_t)(O0 + 4 * k + 1) * a.stride + v] = o.offset;
			if (k == 6) {
				a.state[(size_t)(A0) * a.stride + v] = __float_as_uint(adsr.e.r_out); a.state[(size_t)(A0 + 1) * a.stride + v] = __float_as_uint(adsr.e.r_target);
				a.state[(size_t)(A0 + 2) * a.stride + v] = __float_as_uint(adsr.e.r_rate); a.state[(size_t)(A0 + 3) * a.stride + v] = __float_as_uint(adsr.e.time);
				a.state[v] = (uint32_t)stage | (env_pack(adsr.e) << 2) | bits;
			}
		}
	}
	__syncthreads();
	for (int i = tid; i < n; i += WG) a.partials[(size_t)blockIdx.x * n + i] = mix_rows_sum(i, n);
}

// -------------------------------------------------------------------------------------------------------------------------------------
// ... and with an oscillator PAIR per lane, P consecutive SAMPLES side by side: lane = (voice vi of 16 / P) x (sample slot par of P) x
// (slot q of 4).  Slot q runs oscillators 2q and 2q + 1 in the halves of 2-vectors (osm_saw_pair, klg_device.hpp: packed fp32, the case
// logic on wave masks), slot 3 oscillator 6 and nothing.  The running sum goes through the four slots — slot q adds its two terms to slot
// q - 1's partial sum (three DPP row_shr:1 steps), in the reference's order.  An oscillator's phase is closed-form (offset + s * inc, and
// "was it below the duty one sample ago" is the same test one increment back), so the P sample slots of a voice are independent but for the
// ADSR, whose P steps every lane takes in sequence, keeping its own.
// Why: 73 VALU + 24 SALU operations per iteration serve 16 voice·samples, a third of the instructions per voice of the kernel above —
// but a bank that cannot fill the chip is bound by how fast ONE wave issues (a wave alone on its SIMD: 5-7 cycles per instruction,
// scalar ones included; SQ counters).  P = 4 makes a 16,384-voice bank 4096 waves of a quarter the iterations: four waves
// per SIMD overlap each other's scalar work and latencies (58 -> 49 us); a bank that has the waves anyway keeps P = 1.
// The pair form wants finite coefficients, one duty per lane (osm_pair_ok) and state bits that agree with the phase — what note_on makes.  A wave with a record that is not
// like that (hand-made: klg_voice_upload) rebuilds each oscillator as an Osm per sample and takes the scalar table: slow and exact.
// ---- Two saws of ONE voice side by side (SuperSaw.k's Saw oscillators: a lane per oscillator PAIR, below) ----
// Of the ~47 VALU operations of the six-case table ~33 are fp32 multiplies / adds.  Two oscillators of the same voice in the halves of
// 2-vectors run them as v_pk_* (one issue for both).  On top: (1) the three non-linear cases all end in `(+-rcpf) * X + 1`, so the
// case is selected on X (a negation is exact, (-a)*b == a*(-b)) and the tail is computed once; (2) the state machine's bits live as
// wave masks in SGPRs (`up` = the previous sample's "offset < duty" of all 64 lanes; the one before that is only needed when the state
// is written back and follows from the offset), so case selection is scalar mask logic and the VALU only sees the v_cndmasks; (3)
// `/ 7` drops div_const's class test, which only matters for inf / NaN inputs (excluded by osm_pair_ok) and for the sign of a zero
// quotient (which the sum `0 + a + b ...` cannot see: it never is -0).  Same operations in the same order per oscillator: bit-identical
// to osm_saw (klg_device.hpp) — 181 VALU operations for seven oscillators where the scalar listing has 331.  (For a whole voice in one lane this bought
// 2.5 %, DESIGN.md §3: there the compiler's exec-masked layout of the scalar table already skips the cases no lane needs.  Where a wave
// is short of instructions to overlap — small banks — the shorter sequence is what counts.)
struct SawPair { u2 off, inc; f2 f, omf, rcpf; unsigned long long upx, upy; };
struct SawShared { uint32_t duty; float col, c1, c2; };                             // duty and what OSM::init derives from it: shared by the pair
// what the pair form assumes of an oscillator: a positive increment of at least 2^-23 of a cycle (delta is a normal number: rcpf finite)
// and a duty that is 0 or at least 2^-23 (col = 0 or c1 finite); everything it computes is then finite
__device__ __forceinline__ bool osm_pair_ok(const Osm& o) { return o.inc > 511 && (o.duty == 0u || o.duty > 511u); }
__device__ __forceinline__ f2 div7_finite(f2 x) {                               // div_const<7> on finite inputs (see above)
	const float y = 7.f, r = 1.0f / y;
	const f2 q = x * r;
	const f2 e = __builtin_elementwise_fma(splat(-y), q, x);
	return __builtin_elementwise_fma(e, splat(r), q);
}
struct SawCase { bool carry, same, notlin, bad; };
// OSM::tick 5251-5263 on wave masks: om / nm = "offset < duty" of the previous / this sample in all 64 lanes, cm = this sample's carry
__device__ __forceinline__ SawCase saw_case(unsigned long long om, unsigned long long nm, unsigned long long cm) {
	const unsigned long long same = ~(om ^ nm), valid = (cm & ~om & nm) | (~cm & om & ~nm);      // old == new; DownUp (5) / UpDown (2)
	SawCase s;
	s.carry = __builtin_amdgcn_inverse_ballot_w64(cm);
	s.same = __builtin_amdgcn_inverse_ballot_w64(same);
	s.notlin = __builtin_amdgcn_inverse_ballot_w64(~same | cm);                                // anything but Up (3) / Down (0)
	s.bad = __builtin_amdgcn_inverse_ballot_w64(~same & ~valid);                               // states 1 and 6 "should never happen" -> 0
	return s;
}
__device__ __forceinline__ float saw_x(const SawCase& s, float ud, float du, float wrap) { return s.carry ? (s.same ? -wrap : -du) : ud; }
__device__ __forceinline__ float saw_y(const SawCase& s, float lin, float y) { return s.notlin ? (s.bad ? 0.f : y) : lin; }
// STRIDE: the lane's next sample is STRIDE samples on (klg_render_supersaw_pairs: consecutive samples side by side in lanes).  The
// previous sample's "offset < duty" is the mask the last call left (STRIDE == 1) or, the phase being what it is, the same test one
// increment back (STRIDE > 1: the caller has checked that the record's state bits say the same of the block's first sample).
template<int STRIDE> __device__ __forceinline__ f2 osm_saw_pair(SawPair& o, const SawShared& d) {
	const f2 p = (phase_float2<0x7Fu>(o.off) - 1.f) - d.col;
	const bool nx = o.off.x < d.duty, ny = o.off.y < d.duty;
	const unsigned long long nmx = __ballot(nx), nmy = __ballot(ny);
	const unsigned long long omx = STRIDE == 1 ? o.upx : __ballot(o.off.x - o.inc.x < d.duty), omy = STRIDE == 1 ? o.upy : __ballot(o.off.y - o.inc.y < d.duty);
	const SawCase sx = saw_case(omx, nmx, __ballot(o.off.x < o.inc.x)), sy = saw_case(omy, nmy, __ballot(o.off.y < o.inc.y));
	if (STRIDE == 1) { o.upx = nmx; o.upy = nmy; }
	o.off += o.inc * (unsigned)STRIDE;
	f2 cN;
	cN.x = nx ? d.c1 : d.c2; cN.y = ny ? d.c1 : d.c2;
	const f2 pp = p + p;
	const f2 y_lin = cN * (pp - o.f) + 1.f;                                       // Up (3) / Down (0)
	const f2 x_wrap = 1.f + cN * o.omf * (pp + o.omf);                            // UpDownUp (7) / DownUpDown (4): -rcpf * x + 1
	const f2 p2 = p * p;
	const f2 x_ud = d.c2 * p2 - d.c1 * ((p - o.f) * (p - o.f));                   // UpDown (2): rcpf * x + 1
	const f2 x_du = 1.f + d.c2 * ((p + o.omf) * (p + o.omf)) - d.c1 * p2;         // DownUp (5): -rcpf * x + 1
	f2 x;
	x.x = saw_x(sx, x_ud.x, x_du.x, x_wrap.x); x.y = saw_x(sy, x_ud.y, x_du.y, x_wrap.y);
	const f2 y = o.rcpf * x + 1.f;
	f2 r;
	r.x = saw_y(sx, y_lin.x, y.x); r.y = saw_y(sy, y_lin.y, y.y);
	return r;
}
// (here and not in klg_device.hpp: hipRTC's compiler, which builds the generated patches from that header, has no inverse-ballot builtin)
template<bool B> struct LanesFlag { static constexpr bool value = B; };
template<int P> struct PairsShape { static constexpr int VPW = 16 / P, VPWG = VPW * WAVES; };

template<int P, bool PER_VOICE>
__global__ __launch_bounds__(WG) void klg_render_supersaw_pairs(const RenderArgs a) {
	using Rec = rec::SuperSaw;
	static_assert(P == 1 || P == 2 || P == 4, "sample slots per voice"); static_assert(CHUNK % P == 0, "chunks hold whole iterations");
	constexpr int VPW = PairsShape<P>::VPW, VPWG = PairsShape<P>::VPWG, LPV = 4 * P;      // voices per wave / workgroup; lanes per voice
	__shared__ float lds[WAVES * CHUNK * TILE_LD];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int vi = lane / LPV, par = (lane >> 2) & (P - 1), q = lane & 3;
	const int kx = 2 * q, ky = q < 3 ? 2 * q + 1 : 6;                            // (slot 3's second half runs a copy of oscillator 6; nobody takes its term)
	float* tile = lds + wave * CHUNK * TILE_LD;
	const int n = a.n;
	float* acc = klg_mix_rows + wave * n;                                       // this wave's own mix row
	for (int i = lane; i < n; i += 64) acc[i] = 0.f;
	wave_sync();
	fused_events<PatchSuperSaw>(a, VPWG);                                       // (small banks: this block's note events in the same launch, klg_kernels.hpp)

	const int groups = (a.voices + VPWG - 1) / VPWG;
	for (int g = blockIdx.x; g < groups; g += gridDim.x) {
		const int v0 = g * VPWG + wave * VPW, v = v0 + vi;
		const uint32_t flags = (v < a.voices) ? a.state[v] : (uint32_t)ST_OFF;
		const bool live = (flags & 3u) != (uint32_t)ST_OFF;
		const bool audible = live && (!a.solo || a.solo[v / a.notes_per_synth] == v);   // KLG_MIX_LAST_ACTIVE (see klg_render)
		const bool heard = (PER_VOICE ? live : audible) && q == 3;                  // the lanes whose value is the voice's sample
		if (__ballot(live) == 0ull) {
			if (PER_VOICE) for (int j = 0; j < VPW && v0 + j < a.voices; j++) for (int i = lane; i < n; i += 64) a.per_voice[(size_t)(v0 + j) * n + i] = 0.f;
			continue;
		}
		// ---- this lane's share of the record: its two oscillators (at its sample slot), the voice's ADSR ----
		auto word = [&](int w) { return live ? a.state[(size_t)w * a.stride + v] : 0u; };
		constexpr int O0 = offsetof(Rec, osc) / 4;
		SawPair o; SawShared d;
		uint32_t duty_y;
		const uint32_t stx = live ? KLG_FLAG_GET(flags, 8 + 2 * kx, 2) : 0u, sty = live ? KLG_FLAG_GET(flags, 8 + 2 * ky, 2) : 0u;
		o.inc.x = word(O0 + 4 * kx); o.off.x = word(O0 + 4 * kx + 1); d.duty = word(O0 + 4 * kx + 2); o.f.x = u2f(word(O0 + 4 * kx + 3));
		o.inc.y = word(O0 + 4 * ky); o.off.y = word(O0 + 4 * ky + 1); duty_y = word(O0 + 4 * ky + 2); o.f.y = u2f(word(O0 + 4 * ky + 3));
		// (P > 1: "below the duty one sample ago" is computed from the phase; the record's state bits must agree about the block's first sample)
		const bool bits_agree = P == 1 || ((stx & 1u) == (uint32_t)(o.off.x - o.inc.x < d.duty) && (sty & 1u) == (uint32_t)(o.off.y - o.inc.y < duty_y));
		o.off += o.inc * (unsigned)par;                                             // sample slot par starts at sample par
		o.omf = 1.f - o.f; o.rcpf = 1.f / o.f;                                      // OSM::init 5206-5215
		d.col = fast_phase_float(d.duty); d.c1 = 1.f / d.col; d.c2 = -1.f / (1.0f - d.col);
		o.upx = __ballot(stx & 1u); o.upy = __ballot(sty & 1u);
		auto ok = [](uint32_t inc, uint32_t duty) { return (int32_t)inc > 511 && (duty == 0u || duty > 511u); };
		const bool pairs = __ballot(live && !(bits_agree && d.duty == duty_y && ok(o.inc.x, d.duty) && ok(o.inc.y, duty_y))) == 0ull;
		Adsr adsr;
		{
			constexpr int A0 = offsetof(Rec, adsr) / 4;
			AdsrRec r; r.r_out = u2f(word(A0)); r.r_target = u2f(word(A0 + 1)); r.r_rate = u2f(word(A0 + 2)); r.time = u2f(word(A0 + 3));
			r.A = u2f(word(A0 + 4)); r.AD = u2f(word(A0 + 5)); r.S = u2f(word(A0 + 6)); r.R = u2f(word(A0 + 7));
			adsr_load(adsr, r, live ? KLG_FLAG_GET(flags, 2, 6) : 0u);
		}
		int stage = live ? (int)(flags & 3u) : (int)ST_OFF;
		const float tinc = a.fs.timeInc;

		for (int c0 = 0; c0 < n; c0 += CHUNK) {
			const int cl = (n - c0 < CHUNK) ? (n - c0) : CHUNK;
			float step, tstep;
			const bool safe = env_safe(adsr.e, adsr.e.point == 2, step, tstep, tinc);
			const bool glide = __ballot(stage != (int)ST_OFF && !safe) == 0ull;
			// one iteration = the voice's next P samples.  GLIDE / FULL are compile-time so that the loop a chunk runs has no joins inside it
			// (as a run-time choice each envelope step ends in eight register copies); PAIRS: see `pairs`; FULL = all P samples lie inside the chunk
			auto iteration = [&](auto glide_c, auto pairs_c, auto full_c, const int s) {
				constexpr bool GLIDE = decltype(glide_c)::value, PAIRS = decltype(pairs_c)::value, FULL = decltype(full_c)::value;
				const bool first = P > 1 && c0 + s + par == 0;                          // sample 0 of the block: "one sample ago" is the record's state bit
				f2 x;                                                                 // osc[2q] / 7, osc[2q + 1] / 7   SuperSaw.k:29
				if (PAIRS) x = div7_finite(osm_saw_pair<P>(o, d));
				else {                                                                // (see above: an Osm per oscillator and sample)
					Osm t;
					const bool oux = (P == 1 || first) ? (bool)((o.upx >> lane) & 1ull) : (o.off.x - o.inc.x < d.duty);
					const bool ouy = (P == 1 || first) ? (bool)((o.upy >> lane) & 1ull) : (o.off.y - o.inc.y < duty_y);
					t.inc = (int32_t)o.inc.x; t.offset = o.off.x; t.duty = d.duty; t.delta = o.f.x; t.state = oux ? 1 : 0; osm_derive(t);
					x.x = div_const<0x40e00000u>(osm_saw(t)); o.off.x = t.offset + o.inc.x * (unsigned)(P - 1); o.upx = __ballot(t.state & 1);
					t.inc = (int32_t)o.inc.y; t.offset = o.off.y; t.duty = duty_y; t.delta = o.f.y; t.state = ouy ? 1 : 0; osm_derive(t);
					x.y = div_const<0x40e00000u>(osm_saw(t)); o.off.y = t.offset + o.inc.y * (unsigned)(P - 1); o.upy = __ballot(t.state & 1);
				}
				// out = 0; out += osc[0] / 7; out += osc[1] / 7; ...  — slot q's partial sum is slot q - 1's plus its own two terms
				auto from_left = [](float y) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y), 0x111, 0xF, 0xF, true)); };   // row_shr:1
				float sum = (0.f + x.x) + x.y;
				sum = (from_left(sum) + x.x) + x.y;
				sum = (from_left(sum) + x.x) + x.y;
				sum = from_left(sum) + x.x;                                           // slot 3: oscillator 6 closes the sum
				// adsr++ for the iteration's samples, in order; a lane keeps the value of its own
				float env = 0.f;
#pragma unroll
				for (int k = 0; k < P; k++) if (FULL || s + k < cl) {
					float e;
					if (GLIDE) e = env_glide(adsr.e, step, tstep);
					else { e = adsr_process(adsr, a.fs); stage = (adsr.e.stage == ENV_OFF) ? (int)ST_OFF : stage; }
					env = (par == k) ? e : env;
				}
				if (FULL || s + par < cl) tile[(s + par) * TILE_LD + lane] = heard ? sum * env : 0.f;   // out *= adsr++   SuperSaw.k:31
			};
			auto chunk = [&](auto glide_c, auto pairs_c) {
				int s = 0;
				for (; s + P <= cl; s += P) iteration(glide_c, pairs_c, LanesFlag<true>{}, s);
				if (P > 1 && s < cl) iteration(glide_c, pairs_c, LanesFlag<false>{}, s);
			};
			if (pairs) { if (glide) chunk(LanesFlag<true>{}, LanesFlag<true>{}); else chunk(LanesFlag<false>{}, LanesFlag<true>{}); }
			else { if (glide) chunk(LanesFlag<true>{}, LanesFlag<false>{}); else chunk(LanesFlag<false>{}, LanesFlag<false>{}); }
			wave_sync();
			// sample s of voice j: the slot-3 lane of sample slot s % P
			if (PER_VOICE) {
				const int s = lane & 31, h = lane >> 5;
				for (int j = h; j < VPW; j += 2)
					if (s < cl && v0 + j < a.voices) a.per_voice[(size_t)(v0 + j) * n + c0 + s] = tile[s * TILE_LD + LPV * j + 4 * (s & (P - 1)) + 3];
			}
			{
				const int s = lane & 31, h = lane >> 5;
				float part = 0.f;
				if (s < cl) {
					const float* row = tile + s * TILE_LD + h * 32 + 4 * (s & (P - 1)) + 3;
					if (PER_VOICE && a.solo) { for (int j = 0; j < VPW / 2; j++) { const int vv = v0 + (VPW / 2) * h + j; part += (vv < a.voices && a.solo[vv / a.notes_per_synth] == vv) ? row[LPV * j] : 0.f; } }
					else {
#pragma unroll
						for (int j = 0; j < VPW / 2; j++) part += row[LPV * j];              // the voices of this half of the wave, in order
					}
				}
				part += __shfl_xor(part, 32);
				if (lane < cl) acc[c0 + lane] += part;                                  // the wave's own row: program order, no atomics
			}
			wave_sync();
		}
		// ---- write back (sample slot 0): every slot its phases, slot 3 the envelope and the flags (the state bits of all seven oscillators) ----
		// the phase after n samples; OSM state = the last two samples' "offset < duty" (tick 5251-5263): one and two increments behind it
		// (after a single sample the older one is what the block started with)
		const unsigned back = (unsigned)(par + (n + P - 1) / P * P - n);            // samples this lane's phase is ahead of the block's end
		const uint32_t fx = o.off.x - o.inc.x * back, fy = o.off.y - o.inc.y * back;
		auto state_of = [&](uint32_t off, uint32_t inc, uint32_t duty, uint32_t st0) {
			const uint32_t newer = (uint32_t)((off - inc) < duty), older = n >= 2 ? (uint32_t)((off - 2u * inc) < duty) : (st0 & 1u);
			return newer | (older << 1);
		};
		uint32_t bits = state_of(fx, o.inc.x, d.duty, stx) << (8 + 2 * kx);
		if (q < 3) bits |= state_of(fy, o.inc.y, duty_y, sty) << (8 + 2 * ky);
		bits |= (uint32_t)__shfl_xor((int)bits, 1); bits |= (uint32_t)__shfl_xor((int)bits, 2);
		if (live && par == 0) {
			constexpr int A0 = offsetof(Rec, adsr) / 4;
			a.state[(size_t)(O0 + 4 * kx + 1) * a.stride + v] = fx;
			if (q < 3) a.state[(size_t)(O0 + 4 * ky + 1) * a.stride + v] = fy;
			if (q == 3) {
				a.state[(size_t)(A0) * a.stride + v] = __float_as_uint(adsr.e.r_out); a.state[(size_t)(A0 + 1) * a.stride + v] = __float_as_uint(adsr.e.r_target);
				a.state[(size_t)(A0 + 2) * a.stride + v] = __float_as_uint(adsr.e.r_rate); a.state[(size_t)(A0 + 3) * a.stride + v] = __float_as_uint(adsr.e.time);
				a.state[v] = (uint32_t)stage | (env_pack(adsr.e) << 2) | bits;
			}
		}
	}
	__syncthreads();
	for (int i = tid; i < n; i += WG) a.partials[(size_t)blockIdx.x * n + i] = mix_rows_sum(i, n);
	fused_combine(a, n, 1, reinterpret_cast<int*>(lds));
}

}  // namespace klg
