// klang_amd/csrc/klg_render_supersaw_sp.hpp — SuperSaw.k (config 3) with its SAMPLES side by side and the rare cases apart.
//
// Replaces: SuperSaw.k:21-33 `out = 0; for s < 7: out += osc[s] / 7; out *= adsr++;` per note and sample (Fast::OSM::saw klang.h:5290-5302 over
// OSM::tick 5251-5263), for SuperSaw banks of every size.
//
// klg_render_supersaw_pairs issues 343 lane-instructions per voice·sample: every lane evaluates all six cases of the saw's table for both of its
// oscillators on every sample, and the four lanes of a voice·sample each repeat the ADSR steps and carry a quarter of the sum.  But:
//   * an oscillator's phase is closed-form (offset + s * increment), so are "did the phase wrap" and "is it below the duty" — and between those events,
//     ~98 samples in 100, the saw is ONE line: y = cN * (2p - f) + 1, cN by which side of the duty the phase is on (states Up / Down);
//   * only the ADSR is a chain through the samples.
// So a wave takes 8 voices x a 32-sample chunk through three passes over LDS (the wave's own: no workgroup barrier anywhere):
//   1. lane = (voice, sample slot), 4 iterations: the seven LINEAR values y / 7 of its voice·sample, written to X[voice][sample][oscillator].  ~75 instructions.
//   2. lane = (voice, oscillator), 56 lanes: walks the chunk's SPECIAL samples of its oscillator — the phase wraps, or crosses the duty: from a linear sample the
//      next one is ceil(distance / increment) samples on (a float estimate made exact) — and overwrites their X entries with the six-case formula itself (osm_saw on
//      the reconstructed state: bit-identical by construction).  A voice whose record does not fit pass 1's assumptions (seven duties not one, an increment or
//      duty below 2^-23 of a cycle: hand-made records) has ALL its samples redone here.
//   3. lane = (voice, sample slot): out = (((0 + x0) + x1) ... + x6) * adsr++ — the reference's order —, the per-voice dump, and the voice sum into the wave's mix
//      row.  Every lane of a voice steps its own copy of the voice's ADSR through the chunk and keeps the values of its own samples (two additions per step while no
//      envelope event falls into the chunk: env_safe).
// Same arithmetic per value, same record layout, same write-back as klg_render<PatchSuperSaw> / klg_render_supersaw_pairs: tests/test_gpu_parity.py runs all of
// them against the golden vectors (KLG_SUPERSAW_LANES = 0 / 1 / 2 / 3 forces the choice).
#pragma once
#include "klg_render_lanes.hpp"

namespace klg {

enum { SP_VPW = 8, SP_VPWG = SP_VPW * WAVES, SP_ROW = 40 };                  // voices per wave / workgroup (the default form; VPW = 4 / 2: banks that would otherwise leave SIMDs with one or two waves); float4 slots per voice row of X (32 + 8: the two voices of a quarter-wave land on different banks)

#ifndef KLG_SP_VPW4_MAX_VOICES
// Voices per wave: 8 is the default at every size.  4 / 2 (KLG_SUPERSAW_VPW; bit-identical) give a small bank more, shorter waves — but pass 2's walk costs a wave the same whatever
// it carries, so the work per voice doubles with each halving: 16,384 voices 26.3 / 40.2 / 60.5 us per block with 8 / 4 / 2, 262,144: 0.27 / 0.42 / 0.74 ms; below ~8,192 voices
// the kernel's time on a mostly idle device moves more from run to run than between the forms (2,048 voices: 37 / 49 / 19 us, 4,096: 51 / 20 / 22): no size is given to them.
#define KLG_SP_VPW4_MAX_VOICES 0          // banks up to this size: four voices per wave; up to KLG_SP_VPW2_MAX_VOICES: two
#define KLG_SP_VPW2_MAX_VOICES 0
#endif
template<bool PER_VOICE, int VPW = SP_VPW>
__global__ __launch_bounds__(WG) void klg_render_supersaw_sp(const RenderArgs a) {
	static_assert(VPW == 8 || VPW == 4 || VPW == 2, "voices per wave");
	constexpr int SLOTS = 64 / VPW, ITERS = CHUNK / SLOTS, VPWG = VPW * WAVES, OLANES = 7 * VPW;   // sample slots side by side per voice; iterations per chunk; voices per workgroup; oscillator lanes of pass 2
	using Rec = rec::SuperSaw;
	constexpr int O0 = offsetof(Rec, osc) / 4, A0 = offsetof(Rec, adsr) / 4;
	typedef float f4 __attribute__((ext_vector_type(4)));
	__shared__ f4 xa_all[WAVES][VPW * SP_ROW], xb_all[WAVES][VPW * SP_ROW];     // X[voice][sample]: oscillators 0-3, 4-6
	__shared__ uint32_t bits_all[WAVES][VPW];
	__shared__ int lds_flag;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	f4* const XA = xa_all[wave]; f4* const XB = xb_all[wave]; uint32_t* const SB = bits_all[wave];
	const int vi = lane / SLOTS, j = lane % SLOTS;                              // passes 1 and 3
	const int ev = lane < OLANES ? lane / 7 : 0, ek = lane < OLANES ? lane % 7 : 0;   // pass 2: the oscillator lanes
	const int n = a.n;
	float* acc = klg_mix_rows + wave * n;                                       // this wave's own mix row
	for (int i = lane; i < n; i += 64) acc[i] = 0.f;
	wave_sync();
	fused_events<PatchSuperSaw>(a, VPWG);

	const int groups = (a.voices + VPWG - 1) / VPWG;
	for (int g = blockIdx.x; g < groups; g += gridDim.x) {
		const int v0 = g * VPWG + wave * VPW, v = v0 + vi;
		const uint32_t flags = (v < a.voices) ? a.state[v] : (uint32_t)ST_OFF;
		const bool live = (flags & 3u) != (uint32_t)ST_OFF;
		const bool audible = live && (!a.solo || a.solo[v / a.notes_per_synth] == v);   // KLG_MIX_LAST_ACTIVE (see klg_render)
		const bool heard = PER_VOICE ? live : audible;
		if (__ballot(live) == 0ull) {
			if (PER_VOICE) for (int q = 0; q < VPW && v0 + q < a.voices; q++) for (int i = lane; i < n; i += 64) a.per_voice[(size_t)(v0 + q) * n + i] = 0.f;
			continue;
		}
		// ---- pass 1's view: the seven oscillators of voice vi, at sample slot j ----
		auto word = [&](int vv, bool on, int w) { return on ? a.state[(size_t)w * a.stride + vv] : 0u; };
		uint32_t off[7], inc[7]; float fq[7];
		bool fits = true;                                                         // one duty for all seven, every coefficient finite (what note_on makes)
		const uint32_t duty = word(v, live, O0 + 2);
#pragma unroll
		for (int k = 0; k < 7; k++) {
			inc[k] = word(v, live, O0 + 4 * k); off[k] = word(v, live, O0 + 4 * k + 1); fq[k] = u2f(word(v, live, O0 + 4 * k + 3));
			const uint32_t dk = word(v, live, O0 + 4 * k + 2);
			fits = fits && dk == duty && (int32_t)inc[k] > 511 && (dk == 0u || dk > 511u);
			off[k] += inc[k] * (uint32_t)j;
		}
		const float col = fast_phase_float(duty), c1 = 1.f / col, c2 = -1.f / (1.0f - col);   // OSM::init 5206-5215
		// ---- pass 2's view: oscillator ek of voice ev ----
		const int evv = v0 + ev;
		const uint32_t eflags = __shfl(flags, ev * SLOTS);
		const bool elive = lane < OLANES && (eflags & 3u) != (uint32_t)ST_OFF;
		const int efits = __shfl((int)fits, ev * SLOTS);                                // (every lane takes part in the exchange: not under `elive &&`)
		const bool eall = elive && !efits;                                         // every sample of this oscillator is redone with its own coefficients
		Osm eo;
		{
			OsmRec r; r.inc = (int32_t)word(evv, elive, O0 + 4 * ek); r.offset = word(evv, elive, O0 + 4 * ek + 1); r.duty = word(evv, elive, O0 + 4 * ek + 2); r.delta = u2f(word(evv, elive, O0 + 4 * ek + 3));
			osm_load(eo, r, elive ? KLG_FLAG_GET(eflags, 8 + 2 * ek, 2) : 0u);
		}
		const uint32_t eoff0 = eo.offset, einc = (uint32_t)eo.inc, est = (uint32_t)eo.state;
		const float erinc = 1.0f / (float)einc;
		int cur = 0; uint32_t eoff = eoff0; bool upv = (est & 1u) != 0u;             // the next sample not yet looked at, its phase, "below the duty one sample before it" (sample 0: the record's state bit, OSM::tick 5251-5263)
		// ---- the voice's ADSR (a copy in every lane of the voice) ----
		Adsr adsr;
		{
			AdsrRec r; r.r_out = u2f(word(v, live, A0)); r.r_target = u2f(word(v, live, A0 + 1)); r.r_rate = u2f(word(v, live, A0 + 2)); r.time = u2f(word(v, live, A0 + 3));
			r.A = u2f(word(v, live, A0 + 4)); r.AD = u2f(word(v, live, A0 + 5)); r.S = u2f(word(v, live, A0 + 6)); r.R = u2f(word(v, live, A0 + 7));
			adsr_load(adsr, r, live ? KLG_FLAG_GET(flags, 2, 6) : 0u);
		}
		int stage = live ? (int)(flags & 3u) : (int)ST_OFF;
		const float tinc = a.fs.timeInc;

#ifdef KLG_SP_STAMP
		long long tacc[4] = { 0, 0, 0, 0 }, tprev = __builtin_readcyclecounter(); int iters = 0;
#define SP_STAMP(i) { const long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tprev; tprev = now_; }
#else
#define SP_STAMP(i)
#endif
		SP_STAMP(0)
		for (int c0 = 0; c0 < n; c0 += CHUNK) {
			const int cl = (n - c0 < CHUNK) ? (n - c0) : CHUNK;
			// ---- pass 1: the linear values ----
#pragma unroll
			for (int it = 0; it < ITERS; it++) {
				float x[8];
#pragma unroll
				for (int k = 0; k < 8; k += 2) {                                       // oscillator pairs (0,1) (2,3) (4,5) and 6 with a copy of itself
					const int k1 = k + 1 < 7 ? k + 1 : 6;
					const u2 o2 = { off[k], off[k1] };
					const f2 f2q = { fq[k], fq[k1] };
					const f2 p = (phase_float2<0x7Fu>(o2) - 1.f) - col;                  // saw() 5290: p before tick()
					f2 cN; cN.x = off[k] < duty ? c1 : c2; cN.y = off[k1] < duty ? c1 : c2;
					const f2 y = cN * __builtin_elementwise_fma(p, splat(2.f), -f2q) + 1.f;   // Up (3) / Down (0): cN * ((p + p) - f) + 1 — p + p is exact, so the fma rounds what the subtraction rounds
					const f2 q = div7_finite(y);                                         // `/ 7` SuperSaw.k:29
					x[k] = q.x; x[k + 1] = q.y;
				}
#pragma unroll
				for (int k = 0; k < 7; k++) off[k] += inc[k] * (uint32_t)SLOTS;
				const int s = it * SLOTS + j;
				f4 qa = { x[0], x[1], x[2], x[3] }, qb = { x[4], x[5], x[6], 0.f };
				XA[vi * SP_ROW + s] = qa; XB[vi * SP_ROW + s] = qb;
			}
			wave_sync();
			SP_STAMP(1)
			// ---- pass 2: the special samples ----
			// A sample is special when the phase wrapped on its way to it (carry) or "below the duty" changed.  From a linear sample (phase p, no carry, status u) nothing
			// happens until the phase reaches the duty (u: it is below) or 2^32 (it is not): the next special sample is ceil(D / inc) samples on, D that distance.
			{
				const int cend = c0 + cl;
				if (cur < c0) { eoff += einc * (uint32_t)(c0 - cur); cur = c0; upv = (uint32_t)(eoff - einc) < eo.duty; }   // (the samples skipped were all linear; c0 > 0 here)
				for (;;) {
					// (written without branches up to the one test that ends the walk: the lanes of a wave are at different points of it)
					const bool carry = eoff < einc, nup = eoff < eo.duty;
					const bool here = eall || carry || nup != upv;                       // sample `cur` itself is special
					const uint32_t A = (nup ? eo.duty : 0u) - eoff - 1u;                 // D - 1 (mod 2^32: D = 2^32 - phase when the phase is not below the duty)
					const float qf = (float)A * erinc;
					// floor(A / inc) from the estimate: A < 2^32 and only quotients < 40 matter, so the float product is off by far less than one — its integer part
					// is the quotient or one beside it; the 64-bit remainder says which
					const uint32_t q0 = (uint32_t)fminf(qf, 64.f);
					const long long r = (long long)A - (long long)((unsigned long long)q0 * (unsigned long long)einc);
					const int d = here ? 0 : (qf < (float)(CHUNK + 8) ? (int)q0 + 1 + (r >= (long long)einc ? 1 : 0) - (r < 0 ? 1 : 0) : 2 * CHUNK);   // ceil(D / inc) samples on
					const int s = cur + d;
					const bool hit = elive && s < cend;
					if (__ballot(hit) == 0ull) break;
#ifdef KLG_SP_STAMP
					iters++;
#endif
					if (hit) {
						Osm t = eo;
						t.offset = eoff + einc * (uint32_t)d; t.state = (here ? upv : nup) ? 1 : 0;
						upv = t.offset < eo.duty; eoff = t.offset + einc; cur = s + 1;
						const float xv = div_const<0x40e00000u>(osm_saw(t));
						float* const row = reinterpret_cast<float*>((ek < 4 ? XA : XB) + ev * SP_ROW + (s - c0));
						row[ek & 3] = xv;
					}
				}
			}
			wave_sync();
			SP_STAMP(2)
			// ---- pass 3: adsr++, the voice's samples, the voice sum ----
			float step, tstep;
			const bool safe = env_safe(adsr.e, adsr.e.point == 2, step, tstep, tinc);
			const bool glide = __ballot(stage != (int)ST_OFF && !safe) == 0ull;
			auto pass3 = [&](auto glide_c, auto full_c) {                            // (compile-time forms: the loop a whole chunk runs has no test inside it)
				constexpr bool GLIDE = decltype(glide_c)::value, FULL = decltype(full_c)::value;
#pragma unroll
				for (int it = 0; it < ITERS; it++) {
					float env = 0.f;
#pragma unroll
					for (int q = 0; q < SLOTS; q++) if (FULL || it * SLOTS + q < cl) {            // (cl is the same for the whole wave)
						float e;
						if (GLIDE) e = env_glide(adsr.e, step, tstep);
						else { e = adsr_process(adsr, a.fs); stage = (adsr.e.stage == ENV_OFF) ? (int)ST_OFF : stage; }
						env = (j == q) ? e : env;
					}
					const int s = it * SLOTS + j;
					const f4 qa = XA[vi * SP_ROW + s], qb = XB[vi * SP_ROW + s];
					float sum = 0.f + qa.x; sum += qa.y; sum += qa.z; sum += qa.w; sum += qb.x; sum += qb.y; sum += qb.z;   // out = 0; out += osc[s] / 7 ...   SuperSaw.k:27-29
					const float y = (heard && (FULL || s < cl)) ? sum * env : 0.f;                    // out *= adsr++   SuperSaw.k:31
					if (PER_VOICE) { if ((FULL || s < cl) && v < a.voices) a.per_voice[(size_t)v * n + c0 + s] = y; }
					reinterpret_cast<float*>(XA + vi * SP_ROW + s)[0] = (PER_VOICE && !audible) ? 0.f : y;   // (X is done with: the voice's sample goes where its first term was)
				}
			};
			if (cl == CHUNK) { if (glide) pass3(LanesFlag<true>{}, LanesFlag<true>{}); else pass3(LanesFlag<false>{}, LanesFlag<true>{}); }
			else { if (glide) pass3(LanesFlag<true>{}, LanesFlag<false>{}); else pass3(LanesFlag<false>{}, LanesFlag<false>{}); }
			wave_sync();
			if (lane < cl) {                                                            // the wave's eight voices, a fixed order, into the wave's own mix row (program order, no atomics)
				float t = 0.f;
#pragma unroll
				for (int q = 0; q < VPW; q++) t += reinterpret_cast<const float*>(XA + q * SP_ROW + lane)[0];
				acc[c0 + lane] += t;
			}
			wave_sync();
			SP_STAMP(3)
		}
#ifdef KLG_SP_STAMP
		{ const unsigned long long bf = __ballot(fits), bl = __ballot(live), be = __ballot(eall), bel = __ballot(elive); if (blockIdx.x == 0 && tid == 0) printf("fits %llx live %llx eall %llx elive %llx duty %u inc0 %u inc6 %u\n", bf, bl, be, bel, duty, inc[0], inc[6]); }
		if (blockIdx.x == 0 && tid == 0) printf("sp stamps (cycles): setup %lld | pass1 %lld | pass2 %lld (%d event iterations) | pass3 %lld\n", tacc[0], tacc[1], tacc[2], iters, tacc[3]);
#endif
		// ---- write back: every oscillator lane its phase and state bits, slot 0 of a voice the envelope and the flags ----
		// the phase after n samples; OSM state = the last two samples' "offset < duty" (tick 5251-5263): one and two increments behind it
		// (after a single sample the older one is what the block started with)
		if (lane < VPW) SB[lane] = 0u;
		wave_sync();
		if (elive) {
			const uint32_t fin = eoff0 + einc * (uint32_t)n;
			const uint32_t newer = (uint32_t)((fin - einc) < eo.duty), older = n >= 2 ? (uint32_t)((fin - 2u * einc) < eo.duty) : (est & 1u);
			a.state[(size_t)(O0 + 4 * ek + 1) * a.stride + evv] = fin;
			atomicOr(&SB[ev], (newer | (older << 1)) << (8 + 2 * ek));
		}
		wave_sync();
		if (live && j == 0) {
			a.state[(size_t)(A0) * a.stride + v] = __float_as_uint(adsr.e.r_out); a.state[(size_t)(A0 + 1) * a.stride + v] = __float_as_uint(adsr.e.r_target);
			a.state[(size_t)(A0 + 2) * a.stride + v] = __float_as_uint(adsr.e.r_rate); a.state[(size_t)(A0 + 3) * a.stride + v] = __float_as_uint(adsr.e.time);
			a.state[v] = (uint32_t)stage | (env_pack(adsr.e) << 2) | SB[vi];
		}
		wave_sync();
	}
	__syncthreads();
	for (int i = tid; i < n; i += WG) a.partials[(size_t)blockIdx.x * n + i] = mix_rows_sum(i, n);
	fused_combine(a, n, 1, &lds_flag);
}

}  // namespace klg
