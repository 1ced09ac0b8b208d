// klang_amd/csrc/klg_render_sub2a_sp.hpp — config 2's patch (Saw >> LPF >> ADSR) for SMALL banks: one voice per wave, its samples side by side.
//
// Replaces: the same per-note loop as klg_render<PatchSub2a> / klg_render_sub2a_x2 (`osc >> lpf >> out; out *= adsr++; if (adsr.finished()) stop();`, SURVEY §8d
// patch 2a; Fast::OSM::saw klang.h:5290-5302, Biquad::process 5605-5612, ADSR 3722-4137) — for banks that cannot fill the chip a lane (or half a lane) per voice:
// config 2 at its literal 1,024 voices is 8 waves of the packed kernel on 1,024 SIMDs, each walking 256 samples of ~32 dependent instructions: 20.7 us a block
// whatever else is idle.  What is a chain through the samples is only the filter's recurrence (four dependent operations per sample) and the envelope (one):
//   * the saw's phase is closed-form (offset + s * increment): a lane per SAMPLE computes osc[s] and the three products b0 * osc, b1 * osc, b2 * osc of 64 samples
//     at once (into the wave's LDS);
//   * then the wave — every lane the same values — runs y = b0x + z0; z0 = b1x - a1 * y + z1; z1 = b2x - a2 * y and adsr++ through those 64 samples, each lane
//     keeping the y and the envelope value of ITS sample: ~10 instructions per sample instead of ~32.
// Same operations in the same order per value as PatchSub2a::sample (osm_saw_duty0, biquad_process, adsr_process / env_glide): bit-identical per voice;
// tests/test_gpu_parity.py runs this kernel against the golden vectors and against the packed kernel (KLG_SUB2A_SP = 0 / 1 forces the choice).
#pragma once
#include "klg_render_lanes.hpp"

namespace klg {

// Every lane of the wave walks the same recurrence; lane k is to keep the value of sample k.  As `keep = (lane == s) ? v : keep` that is a scalar operation, a compare
// and a select per sample and value — and on a wave that has its SIMD to itself EVERY instruction, scalar ones included, is one issue slot of four cycles (measured:
// the loop ran 81 cycles a sample for ten vector operations).  Instead the values are pushed through the wave like a shift register: one DPP move (wave_shl:1 — lane
// k takes lane k + 1's, lane 63 the new value) per sample and value; after 64 pushes lane k holds the k-th.  All 64 lanes must be active.
__device__ __forceinline__ float wave_push(float hist, float v) {
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(hist), 0x130, 0xF, 0xF, false));
}

enum { S2_TILE = 64, KLG_SUB2A_SP_MAX_VOICES = 2048 };                      // samples side by side; banks up to this many voices take this kernel (one wave per voice: more than ~2 waves per SIMD and the packed kernel's 128 voices per wave win)

template<bool PER_VOICE>
__global__ __launch_bounds__(WG) void klg_render_sub2a_sp(const RenderArgs a) {
	using Rec = rec::Sub2a;
	constexpr int O0 = offsetof(Rec, osc) / 4, B0 = offsetof(Rec, lpf) / 4, A0 = offsetof(Rec, adsr) / 4;
	typedef float f4 __attribute__((ext_vector_type(4)));
	__shared__ __attribute__((aligned(16))) float p_all[WAVES][3][S2_TILE];             // per wave: [0]: b0 x of the tile's 64 samples; [1 .. 2]: { b1 x, b2 x } sample by sample
	__shared__ int lds_flag;
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	float (* const P)[S2_TILE] = p_all[wave];
	const int n = a.n;
	float* acc = klg_mix_rows + wave * n;                                       // this wave's own mix row
	for (int i = lane; i < n; i += 64) acc[i] = 0.f;
	wave_sync();
	fused_events<PatchSub2a>(a, WAVES);

	const int groups = (a.voices + WAVES - 1) / WAVES;
	for (int g = blockIdx.x; g < groups; g += gridDim.x) {
		const int v = g * WAVES + wave;                                           // the wave's voice (the same in every lane)
		const uint32_t flags = (v < a.voices) ? a.state[v] : (uint32_t)ST_OFF;
		const bool live = (flags & 3u) != (uint32_t)ST_OFF;
		if (!live) {
			if (PER_VOICE && v < a.voices) for (int i = lane; i < n; i += 64) a.per_voice[(size_t)v * n + i] = 0.f;
			continue;
		}
		const bool audible = !a.solo || a.solo[v / a.notes_per_synth] == v;        // KLG_MIX_LAST_ACTIVE (see klg_render)
		auto word = [&](int w) { return a.state[(size_t)w * a.stride + v]; };
		Osm o;
		{ OsmRec r; r.inc = (int32_t)word(O0); r.offset = word(O0 + 1); r.duty = word(O0 + 2); r.delta = u2f(word(O0 + 3)); osm_load(o, r, KLG_FLAG_GET(flags, 8, 2)); }
		Biquad q;
		q.b0 = u2f(word(B0)); q.b1 = u2f(word(B0 + 1)); q.b2 = u2f(word(B0 + 2)); q.a1 = u2f(word(B0 + 3)); q.a2 = u2f(word(B0 + 4)); q.z0 = u2f(word(B0 + 5)); q.z1 = u2f(word(B0 + 6));
		Adsr adsr;
		{
			AdsrRec r; r.r_out = u2f(word(A0)); r.r_target = u2f(word(A0 + 1)); r.r_rate = u2f(word(A0 + 2)); r.time = u2f(word(A0 + 3));
			r.A = u2f(word(A0 + 4)); r.AD = u2f(word(A0 + 5)); r.S = u2f(word(A0 + 6)); r.R = u2f(word(A0 + 7));
			adsr_load(adsr, r, KLG_FLAG_GET(flags, 2, 6));
		}
		int stage = (int)(flags & 3u);
		const float tinc = a.fs.timeInc;
		const uint32_t off0 = o.offset, inc = (uint32_t)o.inc;

		for (int t0 = 0; t0 < n; t0 += S2_TILE) {
			const int tl = (n - t0 < S2_TILE) ? (n - t0) : S2_TILE;
			// ---- the tile's oscillator samples and their products with the filter's feed-forward coefficients: a lane per sample ----
			{
				Osm t = o;
				t.offset = off0 + inc * (uint32_t)(t0 + lane);
				const float x = osm_saw_duty0(t);                                     // `Saw osc` never gets a duty: duty == 0 (patch invariant, PatchSub2a)
				P[0][lane] = q.b0 * x; { f2 p12 = { q.b1 * x, q.b2 * x }; reinterpret_cast<f2*>(&P[1][0])[lane] = p12; }
			}
			wave_sync();
			// ---- the recurrences, 32 samples at a time (what env_safe looks ahead): every lane the same values, each keeps its own sample's ----
			float y_own = 0.f, e_own = 0.f;
			for (int h = 0; h < tl; h += KLG_CHUNK_MAX) {
				const int hl = (tl - h < KLG_CHUNK_MAX) ? (tl - h) : KLG_CHUNK_MAX;
				float step, tstep;
				const bool safe = env_safe(adsr.e, adsr.e.point == 2, step, tstep, tinc);
				const bool glide = __builtin_amdgcn_readfirstlane((int)(stage == (int)ST_OFF || safe)) != 0;
				auto run = [&](auto glide_c, auto full_c) {                            // (compile-time forms: the loop a whole chunk runs has no test inside it)
					constexpr bool GLIDE = decltype(glide_c)::value, FULL = decltype(full_c)::value;
					// (gliding: the envelope's value and its Sustain clock advance together — ONE packed addition a sample: the pair is a chain of its own, nothing of the
					//  filter's waits for it, and an instruction fewer is an issue slot fewer)
					f2 et = { adsr.e.r_out, adsr.e.time }; const f2 es = { step, tstep }; (void)et; (void)es;
					const f2 a12 = { q.a1, q.a2 };
#pragma unroll
					for (int s4 = 0; s4 < KLG_CHUNK_MAX; s4 += 4) {
						if (!FULL && s4 >= hl) break;
						const f4 p0 = *reinterpret_cast<const f4*>(&P[0][h + s4]);
						const f4 pa = reinterpret_cast<const f4*>(&P[1][0])[(h + s4) / 2], pb = reinterpret_cast<const f4*>(&P[1][0])[(h + s4) / 2 + 1];   // { b1 x, b2 x } of samples s4, s4 + 1 | s4 + 2, s4 + 3
#pragma unroll
						for (int k = 0; k < 4; k++) if (FULL || s4 + k < hl) {
							// Biquad::process 5605-5612 (TDF-II): y = b0 in + z0; z0 = b1 in - a1 y + z1; z1 = b2 in - a2 y — the two products with y and the two subtractions as ONE
							// packed operation each (the same roundings per value; two issue slots fewer a sample on a wave that pays one for every instruction)
							const float y = p0[k] + q.z0;
							const f2 p12 = k == 0 ? f2{ pa.x, pa.y } : k == 1 ? f2{ pa.z, pa.w } : k == 2 ? f2{ pb.x, pb.y } : f2{ pb.z, pb.w };
							const f2 d = p12 - a12 * y;
							q.z0 = d.x + q.z1;
							q.z1 = d.y;
							float e;
							if (GLIDE) { e = et.x; et = et + es; }                            // env_glide: out, then out + step and time + tstep
							else { e = adsr_process(adsr, a.fs); stage = (adsr.e.stage == ENV_OFF) ? (int)ST_OFF : stage; }
							y_own = wave_push(y_own, y); e_own = wave_push(e_own, e);       // (see wave_push: after the tile's 64 samples lane k holds sample k's)
						}
					}
					if (GLIDE) { adsr.e.r_out = et.x; adsr.e.time = et.y; }
				};
				if (hl == KLG_CHUNK_MAX) { if (glide) run(LanesFlag<true>{}, LanesFlag<true>{}); else run(LanesFlag<false>{}, LanesFlag<true>{}); }
				else { if (glide) run(LanesFlag<true>{}, LanesFlag<false>{}); else run(LanesFlag<false>{}, LanesFlag<false>{}); }
			}
			for (int r = tl; r < S2_TILE; r++) { y_own = wave_push(y_own, 0.f); e_own = wave_push(e_own, 0.f); }   // (a ragged last tile: moved on to where a whole one ends)
			const float out = lane < tl ? y_own * e_own : 0.f;                        // out *= adsr++
			if (lane < tl) {
				if (PER_VOICE) a.per_voice[(size_t)v * n + t0 + lane] = out;
				acc[t0 + lane] += audible ? out : 0.f;                                  // the wave's own row: program order, no atomics
			}
			wave_sync();
		}
		if (lane == 0) {                                                            // the words PatchSub2a::kStoreMask names
			a.state[(size_t)(O0 + 1) * a.stride + v] = off0 + inc * (uint32_t)n;
			a.state[(size_t)(B0 + 5) * a.stride + v] = __float_as_uint(q.z0); a.state[(size_t)(B0 + 6) * a.stride + v] = __float_as_uint(q.z1);
			a.state[(size_t)(A0) * a.stride + v] = __float_as_uint(adsr.e.r_out); a.state[(size_t)(A0 + 1) * a.stride + v] = __float_as_uint(adsr.e.r_target);
			a.state[(size_t)(A0 + 2) * a.stride + v] = __float_as_uint(adsr.e.r_rate); a.state[(size_t)(A0 + 3) * a.stride + v] = __float_as_uint(adsr.e.time);
			a.state[v] = (uint32_t)stage | (env_pack(adsr.e) << 2) | ((uint32_t)o.state << 8);
		}
	}
	__syncthreads();
	for (int i = tid; i < n; i += WG) a.partials[(size_t)blockIdx.x * n + i] = mix_rows_sum(i, n);
	fused_combine(a, n, 1, &lds_flag);
}

}  // namespace klg
