// klang_amd/csrc/klg_render_lanes.hpp — SuperSaw.k with ONE OSCILLATOR PER LANE, for banks too small to fill the chip.
//
// klg_render<PatchSuperSaw> gives a lane a whole voice: seven six-case OSM tables, the `/ 7`s and the ADSR are ~380 instructions per
// sample, strictly in sequence — config 3's 16,384 voices are 256 waves, one per SIMD on a quarter of the chip, and a block takes the
// 0.19 ms that one wave needs for 256 x 380 instructions whatever else is idle.  Here lane = (voice vi of 8) x (slot k of 8): slot
// k < 7 runs oscillator k, all eight lanes of a voice carry a copy of its ADSR (a copy costs an issue slot nobody else wants), and
// `for s < 7: out += osc[s] / 7` (SuperSaw.k:28-29) — a sum in that order — is a running sum through the lanes: six v_add_f32 with a
// DPP row_shr:1 source, lane k taking lane k - 1's partial sum, ending in slot 6.  ~100 instructions per sample for 8 voices: the bank
// is eight times as many waves, each a quarter as long.  Per voice·sample that is twice the instructions of the voice-per-lane kernel,
// so it only serves banks of up to KLG_LANES_MAX_VOICES voices (2048 workgroups: what the chip holds at once); larger banks keep
// klg_render<PatchSuperSaw>.  Same arithmetic, same order, same record layout: tests/test_gpu_parity.py runs both kernels against the
// golden vectors (KLG_SUPERSAW_LANES=0 / 1 forces the choice).
#pragma once
#include "klg_kernels.hpp"

namespace klg {

enum { KLG_LANES_VOICES_PER_WAVE = 8, KLG_LANES_VOICES_PER_WG = KLG_LANES_VOICES_PER_WAVE * WAVES, KLG_LANES_MAX_VOICES = 65536 };

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

}  // namespace klg
