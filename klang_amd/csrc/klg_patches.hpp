// klang_amd/csrc/klg_patches.hpp — per-patch voice records (the HBM-resident lane state) and the hand-written
// per-sample bodies of the shipped patch kernels.
//
// Data layout in HBM: struct-of-arrays, one 32-bit word plane per record field, `state[w * stride + v]`, so a
// wavefront (64 consecutive voices) loads/stores each field as one coalesced 256-byte access.  A record is read
// once at block start, lives in registers for the n samples of the block, and is written back once: algorithmic
// HBM traffic per voice-block = 2 * sizeof(Rec) (+ the event records of voices that were started).
//
// Word 0 of every record is `flags`: bits 0-1 NoteBase::stage (klang.h:4286), the rest patch-specific small
// integers (envelope stage/point/active, OSM state) packed so they cost one word instead of a dozen.
#pragma once
#include "klg_device.hpp"
#include "../../include/klang_mi355_records.h"

#pragma clang fp contract(off)

namespace klg {

// record layouts: include/klang_mi355_records.h (shared with the host side and the DSL header)

// Per-block view every patch body gets.
// a Stereo::Note's sample (klang.h:4721-4733: `out` is a stereo signal, `buffer++ += out`): what sample() of a patch with kStereo returns
struct Out2 { float l, r; };

enum { KLG_NZ_GROUP = 8 };   // Noise draws are fetched this many samples at a time (one dependent memory round trip per group instead of one per sample)

struct BlockCtx {
	SampleRate fs;
	const float* ctl;        // this voice's synth instance controls [KLG_MAX_CTL]
	const TableDesc* tables; // klg_table_upload()ed sample tables (graph patches with Wavetable / Table reads), else null
	float* ring;             // this voice's note-delay lines (contiguous), else null
	const int* rand; size_t rstride;   // this voice's column of the block's rand() draws: rand[(sample * draws per sample + generator) * rstride] (a generated patch with Noise generators), else null
	const int* nz;           // ... and the lane's column of the wave's LDS copy of the current KLG_NZ_GROUP samples' draws: nz[((sample % KLG_NZ_GROUP) * draws per sample + generator) * 64] (klg_render stages them)
	const uint32_t* rec;     // this voice's record in HBM: word w at rec[w * stride] (what a patch reads only when a segment ends need not sit in registers: PtsLazy*)
	size_t stride;
};

// Breakpoints read from the voice's record when (and only when) a segment ends: envelope points are constants of a note and
// env_segment_end is rare, so they are L2 reads there instead of registers held through every sample.  x(i) / y(i) as Pts2 / Pts3.
struct PtsLazy2 { const uint32_t* px; size_t stride;                              // px: the record word of px[0]; px[1], py[0], py[1] follow
	__device__ __forceinline__ float x(int i) const { return u2f(px[(size_t)(i == 1 ? 1 : 0) * stride]); }
	__device__ __forceinline__ float y(int i) const { return u2f(px[(size_t)(i == 1 ? 3 : 2) * stride]); } };
struct PtsLazyAdsr { const uint32_t* a; size_t stride;                            // a: the record word of AdsrRec::A; AD, S follow — points (0,0) (A,1) (AD,S)
	__device__ __forceinline__ float x(int i) const { return i == 0 ? 0.f : u2f(a[(size_t)(i == 2 ? 1 : 0) * stride]); }
	__device__ __forceinline__ float y(int i) const { return i == 0 ? 0.f : (i == 1 ? 1.f : u2f(a[(size_t)2 * stride])); } };

// ---------------------------------------------------------------------------------------------
// helpers shared by the patch bodies
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void osm_load(Osm& o, const OsmRec& r, uint32_t state_bits) {
	o.inc = r.inc; o.offset = r.offset; o.duty = r.duty; o.delta = r.delta; o.state = (int)(state_bits & 3u);
	osm_derive(o);
}
__device__ __forceinline__ void osm_store(const Osm& o, OsmRec& r) { r.inc = o.inc; r.offset = o.offset; r.duty = o.duty; r.delta = o.delta; }

struct Adsr { Env e; Pts3 p; float R; };
__device__ __forceinline__ void adsr_set_points(Adsr& a, float A, float AD, float S, float R) { a.p.x0 = 0.f; a.p.x1 = A; a.p.x2 = AD; a.p.y0 = 0.f; a.p.y1 = 1.f; a.p.y2 = S; a.R = R; }
__device__ __forceinline__ void adsr_load(Adsr& a, const AdsrRec& r, uint32_t bits) {
	a.e.r_out = r.r_out; a.e.r_target = r.r_target; a.e.r_rate = r.r_rate; a.e.time = r.time;
	env_unpack(a.e, bits);
	a.p.x0 = 0.f; a.p.x1 = r.A; a.p.x2 = r.AD;
	a.p.y0 = 0.f; a.p.y1 = 1.f; a.p.y2 = r.S;
	a.R = r.R;
}
__device__ __forceinline__ void adsr_store(const Adsr& a, AdsrRec& r) {
	r.r_out = a.e.r_out; r.r_target = a.e.r_target; r.r_rate = a.e.r_rate; r.time = a.e.time;
}
__device__ __forceinline__ float adsr_process(Adsr& a, const SampleRate& fs) { return env_process<3, true>(a.e, a.p, 3, fs); }
// QUIET: the ramp is idle and no segment end is pending (holding at the sustain point, or Off) — only the Sustain clock runs,
// and only host events between blocks end that.  adsr_hold is adsr_process for such a sample.
__device__ __forceinline__ bool adsr_quiet(const Adsr& a) { return !a.e.active && ((a.e.stage == ENV_SUSTAIN && a.e.point == 2) || a.e.stage == ENV_OFF); }
__device__ __forceinline__ float adsr_hold(Adsr& a, const SampleRate& fs) { a.e.time = (a.e.stage == ENV_SUSTAIN) ? (a.e.time + fs.timeInc) : a.e.time; return a.e.r_out; }
// ADSR::release(time = 0, level = 0) klang.h:4131-4133 on the packed record (event kernel)
__device__ __forceinline__ void adsr_release_rec(AdsrRec& r, uint32_t& bits, float fs) {
	Env e; e.r_out = r.r_out; e.r_target = r.r_target; e.r_rate = r.r_rate; e.time = r.time; env_unpack(e, bits);
	env_release(e, r.R, 0.f, fs);
	r.r_out = e.r_out; r.r_target = e.r_target; r.r_rate = e.r_rate; r.time = e.time; bits = env_pack(e);
}

#define KLG_FLAG_GET(flags, shift, width) (((flags) >> (shift)) & ((1u << (width)) - 1u))

// Store masks: bit w set = record word w changes during a block and is written back (coefficients,
// breakpoints and increments are read-only for the render kernel, so they cost a read but no write).
constexpr uint64_t words(size_t byte_off, int n) { return ((1ull << n) - 1ull) << (byte_off / 4); }
#define KLG_W(REC, member, n) words(__builtin_offsetof(REC, member), n)

// ---------------------------------------------------------------------------------------------
// config 1: one Generators::Fast::Sine per note  (oracle/ref/ref_sine.cpp: `osc >> out`)
// ---------------------------------------------------------------------------------------------
struct PatchSine {
	using Rec = rec::Sine;
	static constexpr uint64_t kStoreMask = KLG_W(Rec, flags, 1) | KLG_W(Rec, pos, 1);
	struct Live { FSine osc; int stage; };
	static __device__ __forceinline__ void begin(Live& L, const Rec& r, const BlockCtx&) { L.osc.inc = r.inc; L.osc.pos = r.pos; L.stage = (int)(r.flags & 3u); }
	static __device__ __forceinline__ float sample(Live& L, const BlockCtx&) { return fsine_process(L.osc, 0u); }
	static __device__ __forceinline__ void end(const Live& L, Rec& r) { r.inc = L.osc.inc; r.pos = L.osc.pos; r.flags = (uint32_t)L.stage; }
	static __device__ __forceinline__ void release(Rec& r, float) { r.flags = (r.flags & ~3u) | ST_OFF; }   // off() { stop(); }
};

struct PatchBSine {
	using Rec = rec::BSine;
	static constexpr uint64_t kStoreMask = KLG_W(Rec, flags, 1) | KLG_W(Rec, position, 1);
	struct Live { BOsc osc; int stage; };
	static __device__ __forceinline__ void begin(Live& L, const Rec& r, const BlockCtx&) { L.osc.increment = r.increment; L.osc.position = r.position; L.osc.offset = r.offset; L.stage = (int)(r.flags & 3u); }
	static __device__ __forceinline__ float sample(Live& L, const BlockCtx&) { return basic_sine(L.osc); }
	static __device__ __forceinline__ void end(const Live& L, Rec& r) { r.position = L.osc.position; r.flags = (uint32_t)L.stage; }
	static __device__ __forceinline__ void release(Rec& r, float) { r.flags = (r.flags & ~3u) | ST_OFF; }
};

// ---------------------------------------------------------------------------------------------
// config 2a (north_star "Saw >> LPF >> ADSR"): `osc >> lpf >> out; out *= adsr++; if (adsr.finished()) stop();`
// flags: [0:2) note stage | [2:8) adsr | [8:10) osm state
// ---------------------------------------------------------------------------------------------
struct PatchSub2a {
	using Rec = rec::Sub2a;                                                                  // 20 words = 80 B read, 8 words = 32 B written
	static constexpr uint64_t kStoreMask = KLG_W(Rec, flags, 1) | KLG_W(Rec, osc.offset, 1) | KLG_W(Rec, lpf.z0, 2) | KLG_W(Rec, adsr.r_out, 4);
	struct Live { Osm osc; Biquad lpf; Adsr adsr; int stage; };
	static __device__ __forceinline__ void begin(Live& L, const Rec& r, const BlockCtx&) {
		L.stage = (int)(r.flags & 3u);
		adsr_load(L.adsr, r.adsr, KLG_FLAG_GET(r.flags, 2, 6));
		osm_load(L.osc, r.osc, KLG_FLAG_GET(r.flags, 8, 2));
		L.lpf.b0 = r.lpf.b0; L.lpf.b1 = r.lpf.b1; L.lpf.b2 = r.lpf.b2; L.lpf.a1 = r.lpf.a1; L.lpf.a2 = r.lpf.a2; L.lpf.z0 = r.lpf.z0; L.lpf.z1 = r.lpf.z1;
	}
	static __device__ __forceinline__ float sample(Live& L, const BlockCtx& c) {
		float out = biquad_process(L.lpf, osm_saw_duty0(L.osc));     // `Saw osc` never gets a duty: duty == 0 (patch invariant)
		out *= adsr_process(L.adsr, c.fs);
		L.stage = (L.adsr.e.stage == ENV_OFF) ? (int)ST_OFF : L.stage;
		return out;
	}
	static __device__ __forceinline__ void end(const Live& L, Rec& r) {
		osm_store(L.osc, r.osc);
		r.lpf.z0 = L.lpf.z0; r.lpf.z1 = L.lpf.z1;
		adsr_store(L.adsr, r.adsr);
		r.flags = (uint32_t)L.stage | (env_pack(L.adsr.e) << 2) | ((uint32_t)L.osc.state << 8);
	}
	static __device__ __forceinline__ void release(Rec& r, float fs) {                        // off(): adsr.release()
		uint32_t bits = KLG_FLAG_GET(r.flags, 2, 6);
		adsr_release_rec(r.adsr, bits, fs);
		r.flags = (r.flags & ~(0x3Fu << 2) & ~3u) | (bits << 2) | ST_RELEASE;
	}
};

// ---------------------------------------------------------------------------------------------
// Event-free chunks.  Most chunks of a block contain no envelope EVENT: a ramp still further from its target than (chunk + 2) steps —
// plus what a chunk of roundings can add up to — neither clamps nor goes idle inside the chunk, and an idle envelope whose idleness
// means nothing (Off, or an ADSR holding at its sustain point) stays as it is.  When that holds for every envelope of every sounding
// voice of a wave (a patch's quiet() = 2), the chunk runs `out = value; value += step; time += tstep` per envelope (env_glide) — two
// additions instead of the ramp's median, the activity bookkeeping and the wave-wide segment-end test of env_process.  Same bits: the
// median of (out, out + rate, target) IS out + rate for a ramp that does not arrive, and an idle ramp steps by -0.0 (x + -0.0 == x).
// ---------------------------------------------------------------------------------------------
enum { KLG_CHUNK_MAX = 32 };   // the longest chunk klg_render runs between two quiet() decisions (checked there)
__device__ __forceinline__ bool env_safe(const Env& e, bool settled, float& step, float& tstep, float tinc) {
	const bool sustain = e.stage == ENV_SUSTAIN;
	const float d = fabsf(e.r_target - e.r_out);
	const float mag = fmaxf(1.f, fmaxf(fabsf(e.r_target), fabsf(e.r_out)));
	const float lim = fabsf(e.r_rate) * (float)(KLG_CHUNK_MAX + 2) + mag * ((float)KLG_CHUNK_MAX * 2.4e-7f);
	step = e.active ? ((e.r_target > e.r_out) ? e.r_rate : -e.r_rate) : -0.f;
	tstep = sustain ? tinc : 0.f;
	return e.active ? (d > lim) : !((sustain && !settled) || e.stage == ENV_RELEASE);      // (`d > lim` is false for a NaN / infinite rate: not safe)
}
__device__ __forceinline__ float env_glide(Env& e, float step, float tstep) { const float out = e.r_out; e.r_out = out + step; e.time += tstep; return out; }

// ---------------------------------------------------------------------------------------------
// config 2b: the shipped subtractive.k (Square >> filter(env++, 10) >> out; out *= adsr)
// flags: [0:2) note | [2:8) adsr | [8:14) env | [14:16) osm state
// ---------------------------------------------------------------------------------------------
struct PatchSub2b {
	using Rec = rec::Sub2b;                                                                  // 1+4+8+10+9 = 32 words
	static constexpr uint64_t kStoreMask = KLG_W(Rec, flags, 1) | KLG_W(Rec, osc.offset, 1) | KLG_W(Rec, adsr.r_out, 4) | KLG_W(Rec, env.r_out, 4) | KLG_W(Rec, filter, 9);
	struct Live { Osm osc; Adsr adsr; Env env; Pts3 p; Biquad lpf; BiquadSweep sw; int stage; float step[2], tstep[2], tinc; };
	static __device__ __forceinline__ void begin(Live& L, const Rec& r, const BlockCtx& c) {
		L.tinc = c.fs.timeInc;
		L.stage = (int)(r.flags & 3u);
		adsr_load(L.adsr, r.adsr, KLG_FLAG_GET(r.flags, 2, 6));
		L.env.r_out = r.env.r_out; L.env.r_target = r.env.r_target; L.env.r_rate = r.env.r_rate; L.env.time = r.env.time;
		env_unpack(L.env, KLG_FLAG_GET(r.flags, 8, 6));
		L.p.x0 = r.env.px[0]; L.p.x1 = r.env.px[1]; L.p.x2 = r.env.px[2];
		L.p.y0 = r.env.py[0]; L.p.y1 = r.env.py[1]; L.p.y2 = r.env.py[2];
		osm_load(L.osc, r.osc, KLG_FLAG_GET(r.flags, 14, 2));
		L.sw.f = r.filter.f; L.sw.Q = r.filter.Q;
		L.lpf.b0 = r.filter.c.b0; L.lpf.b1 = r.filter.c.b1; L.lpf.b2 = r.filter.c.b2; L.lpf.a1 = r.filter.c.a1; L.lpf.a2 = r.filter.c.a2;
		L.lpf.z0 = r.filter.c.z0; L.lpf.z1 = r.filter.c.z1;
	}
	static __device__ __forceinline__ float sample(Live& L, const BlockCtx& c) {
		const float fc = env_process<3, false>(L.env, L.p, 3, c.fs);   // filter(env++, 10) is evaluated first (C++17 order)
		biquad_lpf_set(L.lpf, L.sw, fc, 10.f, c.fs.w);
		float out = biquad_process(L.lpf, osm_pulse(L.osc));
		out *= adsr_process(L.adsr, c.fs);
		L.stage = (L.adsr.e.stage == ENV_OFF) ? (int)ST_OFF : L.stage;
		return out;
	}
	// event-free chunks (see env_safe above): the filter envelope and the ADSR glide
	static constexpr bool kHasQuiet = true;
	static __device__ __forceinline__ int quiet(Live& L) {
		bool safe = env_safe(L.env, false, L.step[0], L.tstep[0], L.tinc);
		safe = env_safe(L.adsr.e, L.adsr.e.point == 2, L.step[1], L.tstep[1], L.tinc) && safe;
		return __ballot(L.stage != (int)ST_OFF && !safe) == 0ull ? 2 : 0;
	}
	static __device__ __forceinline__ float sample_fast(Live& L, const BlockCtx& c) {
		const float fc = env_glide(L.env, L.step[0], L.tstep[0]);
		biquad_lpf_set(L.lpf, L.sw, fc, 10.f, c.fs.w);
		float out = biquad_process(L.lpf, osm_pulse(L.osc));
		out *= env_glide(L.adsr.e, L.step[1], L.tstep[1]);
		return out;
	}
	static __device__ __forceinline__ float sample_quiet(Live& L, const BlockCtx& c) { return sample(L, c); }   // (level 1 is not used by this patch)
	static __device__ __forceinline__ void end(const Live& L, Rec& r) {
		osm_store(L.osc, r.osc);
		adsr_store(L.adsr, r.adsr);
		r.env.r_out = L.env.r_out; r.env.r_target = L.env.r_target; r.env.r_rate = L.env.r_rate; r.env.time = L.env.time;
		r.filter.f = L.sw.f; r.filter.Q = L.sw.Q;
		r.filter.c.b0 = L.lpf.b0; r.filter.c.b1 = L.lpf.b1; r.filter.c.b2 = L.lpf.b2; r.filter.c.a1 = L.lpf.a1; r.filter.c.a2 = L.lpf.a2;
		r.filter.c.z0 = L.lpf.z0; r.filter.c.z1 = L.lpf.z1;
		r.flags = (uint32_t)L.stage | (env_pack(L.adsr.e) << 2) | (env_pack(L.env) << 8) | ((uint32_t)L.osc.state << 14);
	}
	static __device__ __forceinline__ void release(Rec& r, float fs) {
		uint32_t bits = KLG_FLAG_GET(r.flags, 2, 6);
		adsr_release_rec(r.adsr, bits, fs);
		r.flags = (r.flags & ~(0x3Fu << 2) & ~3u) | (bits << 2) | ST_RELEASE;
	}
};

// ---------------------------------------------------------------------------------------------
// config 3: the shipped SuperSaw.k: `for s<7: out += osc[s] / 7; out *= adsr++;`
// flags: [0:2) note | [2:8) adsr | [8:22) 7 x osm state
// ---------------------------------------------------------------------------------------------
struct PatchSuperSaw {
	using Rec = rec::SuperSaw;                                                               // 37 words = 148 B read, 12 words written
	static constexpr uint64_t kStoreMask = KLG_W(Rec, flags, 1) | KLG_W(Rec, osc[0].offset, 1) | KLG_W(Rec, osc[1].offset, 1) | KLG_W(Rec, osc[2].offset, 1)
		| KLG_W(Rec, osc[3].offset, 1) | KLG_W(Rec, osc[4].offset, 1) | KLG_W(Rec, osc[5].offset, 1) | KLG_W(Rec, osc[6].offset, 1) | KLG_W(Rec, adsr.r_out, 4);
	static constexpr int kWavesPerEu = 3;                                                   // (no spills.  Four waves per SIMD with 25 spilled registers rendered 8 % faster while this kernel served the large banks; klg_render_supersaw_sp does now, this one is the A/B reference: KLG_SUPERSAW_LANES=0)
	struct Live { Osm osc[7]; Adsr adsr; int stage; bool duty0; float step, tstep, tinc; };
	static __device__ __forceinline__ void begin(Live& L, const Rec& r, const BlockCtx& c) {
		L.tinc = c.fs.timeInc;
		L.stage = (int)(r.flags & 3u);
		adsr_load(L.adsr, r.adsr, KLG_FLAG_GET(r.flags, 2, 6));
		bool general = false;
#pragma unroll
		for (int k = 0; k < 7; k++) { osm_load(L.osc[k], r.osc[k], KLG_FLAG_GET(r.flags, 8 + 2 * k, 2)); general = general || L.osc[k].duty != 0u || L.osc[k].state != 0; }
		// SuperSaw.k's oscillators are Saw() — Osm(&OSM::saw, 0.f), a duty nobody sets: while no voice of the wave has one (and every
		// state machine rests in Down) the seven saws take the two-case form of osm_saw_duty0 instead of the general table (wave-uniform,
		// decided once per block: nothing inside a block gives a saw a duty)
		L.duty0 = __ballot(general) == 0ull;
	}
	static __device__ __forceinline__ float saws(Live& L) {
		float out = 0.f;
		if (L.duty0) {
#pragma unroll
			for (int k = 0; k < 7; k++) out += div_const<0x40e00000u>(osm_saw_duty0(L.osc[k]));   // `/ 7` SuperSaw.k:29
		}
		else {
#pragma unroll
			for (int k = 0; k < 7; k++) out += div_const<0x40e00000u>(osm_saw(L.osc[k]));
		}
		return out;
	}
	static __device__ __forceinline__ float sample(Live& L, const BlockCtx& c) {
		float out = saws(L);
		out *= adsr_process(L.adsr, c.fs);
		L.stage = (L.adsr.e.stage == ENV_OFF) ? (int)ST_OFF : L.stage;
		return out;
	}
	// chunks in which every ADSR of the wave merely holds (klg_render: HasQuiet): the envelope is its value, only the Sustain clock runs
	static constexpr bool kHasQuiet = true;
	// ... and chunks without an envelope event (env_safe above): the ADSR glides (quiet() = 2)
	static __device__ __forceinline__ int quiet(Live& L) {
		if (__ballot(!adsr_quiet(L.adsr)) == 0ull) return 1;
		const bool safe = env_safe(L.adsr.e, L.adsr.e.point == 2, L.step, L.tstep, L.tinc);
		return __ballot(L.stage != (int)ST_OFF && !safe) == 0ull ? 2 : 0;
	}
	static __device__ __forceinline__ float sample_quiet(Live& L, const BlockCtx& c) {
		float out = saws(L);
		out *= adsr_hold(L.adsr, c.fs);
		return out;
	}
	static __device__ __forceinline__ float sample_fast(Live& L, const BlockCtx&) {
		float out = saws(L);
		out *= env_glide(L.adsr.e, L.step, L.tstep);
		return out;
	}
	static __device__ __forceinline__ void end(const Live& L, Rec& r) {
		uint32_t f = (uint32_t)L.stage | (env_pack(L.adsr.e) << 2);
#pragma unroll
		for (int k = 0; k < 7; k++) { osm_store(L.osc[k], r.osc[k]); f |= (uint32_t)L.osc[k].state << (8 + 2 * k); }
		adsr_store(L.adsr, r.adsr);
		r.flags = f;
	}
	static __device__ __forceinline__ void release(Rec& r, float fs) {
		uint32_t bits = KLG_FLAG_GET(r.flags, 2, 6);
		adsr_release_rec(r.adsr, bits, fs);
		r.flags = (r.flags & ~(0x3Fu << 2) & ~3u) | (bits << 2) | ST_RELEASE;
	}
};

// ---------------------------------------------------------------------------------------------
// FM.k (3 operators) and the 4-operator chain of BASELINE config 5:
//   op1 * I1 >> op2 * I2 >> [op3 * I3 >>] opN >> out;  out *= adsr++ * 0.1f;
// flags: [0:2) note | [2:8) adsr | [8+6k : 14+6k) operator k envelope ; meta: 2 bits npoints per operator
// ---------------------------------------------------------------------------------------------
template<int NOPS>
struct PatchFM {
	using OpRec = klg::OpRec;                                                                // 10 words
	using Rec = rec::FM<NOPS>;
	static constexpr uint64_t op_mask(int k) { return words(8 + 40 * (size_t)k + 4, 5); }   // pos, r_out, r_target, r_rate, time of operator k
	static constexpr uint64_t kStoreMask = 1ull | op_mask(0) | op_mask(1) | op_mask(2) | (NOPS > 3 ? op_mask(3) : 0ull) | words(8 + 40 * (size_t)NOPS, 4);
	// (breakpoints, point counts and the ADSR's times are read from the record when a segment ends — PtsLazy2 / PtsLazyAdsr —: 23 registers
	//  fewer through the sample loops, 4 waves per SIMD instead of 3)
#ifdef KLG_FM_WAVES
	static constexpr int kWavesPerEu = KLG_FM_WAVES;
#endif
	struct Op { FSine osc; Env env; float amp; };
	struct Live { Op op[NOPS]; Env adsr; int stage; uint32_t meta; float step[NOPS + 1], tstep[NOPS + 1], tinc; };   // step / tstep: see quiet()
	static constexpr int kOpWord0 = 2, kOpWords = 10, kAdsrWord0 = 2 + 10 * NOPS;      // rec::FM<NOPS> in words: flags, meta, op[NOPS], adsr
	static __device__ __forceinline__ PtsLazy2 op_pts(const BlockCtx& c, int k) { return PtsLazy2{ c.rec + (size_t)(kOpWord0 + kOpWords * k + 6) * c.stride, c.stride }; }
	static __device__ __forceinline__ PtsLazyAdsr adsr_pts(const BlockCtx& c) { return PtsLazyAdsr{ c.rec + (size_t)(kAdsrWord0 + 4) * c.stride, c.stride }; }
	static __device__ __forceinline__ void begin(Live& L, const Rec& r, const BlockCtx& c) {
		L.tinc = c.fs.timeInc;
		L.stage = (int)(r.flags & 3u);
		L.meta = r.meta;
		L.adsr.r_out = r.adsr.r_out; L.adsr.r_target = r.adsr.r_target; L.adsr.r_rate = r.adsr.r_rate; L.adsr.time = r.adsr.time;
		env_unpack(L.adsr, KLG_FLAG_GET(r.flags, 2, 6));
#pragma unroll
		for (int k = 0; k < NOPS; k++) {
			Op& o = L.op[k]; const OpRec& q = r.op[k];
			o.osc.inc = q.inc; o.osc.pos = q.pos;
			o.env.r_out = q.r_out; o.env.r_target = q.r_target; o.env.r_rate = q.r_rate; o.env.time = q.time;
			env_unpack(o.env, KLG_FLAG_GET(r.flags, 8 + 6 * k, 6));
			// `op * I` sets amp every sample from controls[1 + k] (FM.k:64-68); the last operator keeps amp = 1
			o.amp = (k < NOPS - 1) ? c.ctl[1 + k] : 1.f;
		}
	}
	// Operator::process klang.h:4164-4168
	static __device__ __forceinline__ float op_process(Op& o, int k, uint32_t meta, float in, const BlockCtx& c) {
		float y = fsine_process_rel(o.osc, in);                    // OSCILLATOR::set(+in); OSCILLATOR::process()
		y *= env_process<2, false>(o.env, op_pts(c, k), (int)KLG_FLAG_GET(meta, 2 * k, 2), c.fs) * o.amp;
		return y;
	}
	static __device__ __forceinline__ float sample(Live& L, const BlockCtx& c) {
		float m = 0.f;
#pragma unroll
		for (int k = 0; k < NOPS; k++) m = op_process(L.op[k], k, L.meta, m, c);
		float out = m;
		out *= env_process<3, true>(L.adsr, adsr_pts(c), 3, c.fs) * 0.1f;      // adsr_process
		L.stage = (L.adsr.stage == ENV_OFF) ? (int)ST_OFF : L.stage;
		return out;
	}
	// event-free chunks (see env_safe above): the five envelopes glide
	static constexpr bool kHasQuiet = true;
	static __device__ __forceinline__ int quiet(Live& L) {
		bool safe = true;
#pragma unroll
		for (int k = 0; k < NOPS; k++) safe = env_safe(L.op[k].env, false, L.step[k], L.tstep[k], L.tinc) && safe;
		safe = env_safe(L.adsr, L.adsr.point == 2, L.step[NOPS], L.tstep[NOPS], L.tinc) && safe;
		const bool sounding = L.stage != (int)ST_OFF;                      // (a lane without a voice, or whose note has ended, is heard by nobody: it does not veto)
		return __ballot(sounding && !safe) == 0ull ? 2 : 0;
	}
	static __device__ __forceinline__ float sample_fast(Live& L, const BlockCtx& c) {
		float m = 0.f;
#pragma unroll
		for (int k = 0; k < NOPS; k++) {
			Op& o = L.op[k];
			float y = fsine_process_rel(o.osc, m);
			y *= env_glide(o.env, L.step[k], L.tstep[k]) * o.amp;
			m = y;
		}
		float out = m;
		out *= env_glide(L.adsr, L.step[NOPS], L.tstep[NOPS]) * 0.1f;
		return out;
	}
	static __device__ __forceinline__ float sample_quiet(Live& L, const BlockCtx& c) { return sample(L, c); }   // (level 1 is not used by this patch)
	// TWO consecutive samples of an event-free chunk, side by side in the halves of 2-vectors (klg_render: HasFast2).  Nothing of a sample
	// depends on the sample before it except the oscillator positions (closed form: pos, pos + inc) and the gliding envelopes (two additions
	// each, taken in order), so the operator chain of sample s + 1 runs beside that of sample s: sine polynomial, envelope and amp products,
	// the modulation's Fast::Phase — as v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32.  Same operations per sample, same bits.
	// kFastN samples per call: kFastN / 2 INDEPENDENT pair chains — a bank of config 5's size gives a SIMD two waves, and one chain is a
	// serial dependence from the first operator's phase to the output (every packed operation waiting for the one before it); a second
	// chain fills those slots.
	static constexpr bool kHasFast2 = true;
#ifndef KLG_FM_FASTN
#define KLG_FM_FASTN 4
#endif
	static constexpr int kFastN = KLG_FM_FASTN;
	template<int PAIRS>
	static __device__ __forceinline__ void sample_fast_pairs(Live& L, f2 (&out)[PAIRS]) {
		f2 m[PAIRS];
#pragma unroll
		for (int j = 0; j < PAIRS; j++) m[j] = splat(0.f);
		// the Sustain clocks of the NOPS + 1 envelopes: nothing in an event-free chunk reads them — they are only carried — so two envelopes' clocks advance in ONE
		// packed addition (round 6: 2 x ceil((NOPS + 1) / 2) operations per sample pair instead of 2 x (NOPS + 1))
		constexpr int NT = (NOPS + 2) / 2;
		f2 tm[NT], tsp[NT];
#pragma unroll
		for (int i = 0; i < NT; i++) {
			const int a = 2 * i, b = 2 * i + 1;
			tm[i].x = a < NOPS ? L.op[a < NOPS ? a : 0].env.time : L.adsr.time;
			tm[i].y = b < NOPS ? L.op[b < NOPS ? b : 0].env.time : (b == NOPS ? L.adsr.time : 0.f);
			tsp[i].x = L.tstep[a]; tsp[i].y = b <= NOPS ? L.tstep[b <= NOPS ? b : 0] : 0.f;
		}
#pragma unroll
		for (int j = 0; j < PAIRS; j++) {
#pragma unroll
			for (int i = 0; i < NT; i++) tm[i] = (tm[i] + tsp[i]) + tsp[i];
		}
#pragma unroll
		for (int i = 0; i < NT; i++) {
			const int a = 2 * i, b = 2 * i + 1;
			if (a < NOPS) L.op[a < NOPS ? a : 0].env.time = tm[i].x; else L.adsr.time = tm[i].x;
			if (b < NOPS) L.op[b < NOPS ? b : 0].env.time = tm[i].y; else if (b == NOPS) L.adsr.time = tm[i].y;
		}
#pragma unroll
		for (int k = 0; k < NOPS; k++) {
			Op& o = L.op[k];
			const uint32_t inc = (uint32_t)o.osc.inc;
			const float st = L.step[k];
#pragma unroll
			for (int j = 0; j < PAIRS; j++) {
				u2 pos = { o.osc.pos, o.osc.pos + inc };
				o.osc.pos += 2u * inc;
				if (k > 0) pos = fast_phase_add(pos, m[j] * KLG_TWO_PI);       // OSCILLATOR::set(+in): fsine_rel_offset (operator 0 is not modulated: offset 0)
				f2 y = fastsinp(pos);
				f2 ev; ev.x = o.env.r_out; ev.y = ev.x + st; o.env.r_out = ev.y + st;      // two steps of env_glide
				y *= ev * o.amp;
				m[j] = y;
			}
		}
		const float st = L.step[NOPS];
#pragma unroll
		for (int j = 0; j < PAIRS; j++) {
			f2 av; av.x = L.adsr.r_out; av.y = av.x + st; L.adsr.r_out = av.y + st;
			out[j] = m[j] * (av * 0.1f);
		}
	}
	static __device__ __forceinline__ f2 sample_fast2(Live& L, const BlockCtx&) { f2 y[1]; sample_fast_pairs<1>(L, y); return y[0]; }
	static __device__ __forceinline__ void end(const Live& L, Rec& r) {
		uint32_t f = (uint32_t)L.stage | (env_pack(L.adsr) << 2);
#pragma unroll
		for (int k = 0; k < NOPS; k++) {
			const Op& o = L.op[k]; OpRec& q = r.op[k];
			q.pos = o.osc.pos;
			q.r_out = o.env.r_out; q.r_target = o.env.r_target; q.r_rate = o.env.r_rate; q.time = o.env.time;
			f |= env_pack(o.env) << (8 + 6 * k);
		}
		r.adsr.r_out = L.adsr.r_out; r.adsr.r_target = L.adsr.r_target; r.adsr.r_rate = L.adsr.r_rate; r.adsr.time = L.adsr.time;
		r.flags = f;
	}
	static __device__ __forceinline__ void release(Rec& r, float fs) {
		uint32_t bits = KLG_FLAG_GET(r.flags, 2, 6);
		adsr_release_rec(r.adsr, bits, fs);
		r.flags = (r.flags & ~(0x3Fu << 2) & ~3u) | (bits << 2) | ST_RELEASE;
	}
};

} // namespace klg
