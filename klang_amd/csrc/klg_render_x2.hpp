// klang_amd/csrc/klg_render_x2.hpp — the Subtractive (config 2a) voice kernel with TWO voices per lane.
//
// Why: the synth patches are bound by fp32 VALU issue, -ffp-contract=off forbids FMA, and a plain
// v_mul/v_add retires one fp32 op per lane.  gfx950's packed fp32 pipe (v_pk_mul_f32 / v_pk_add_f32) retires two.
// Giving every lane two voices (a float2 of independent state) turns ~60 % of the per-sample instruction stream
// into packed ops and doubles the independent work between dependent instructions.  Each half of every packed op
// is an ordinary IEEE fp32 mul/add, so results stay bit-identical to the one-voice-per-lane kernel (and to the
// reference); tests/test_gpu_parity.py runs both.
//
// Layout: lane l of wave w of workgroup g owns voices v = g*512 + w*128 + 2*l + {0,1}: state planes are read as
// 8-byte (dwordx2) coalesced accesses.  Mix: 16-sample chunks, tile[s][lane] of float2 with a 65-slot row stride
// (conflict-free ds_write_b64 / ds_read_b64), lane (s = l & 15, q = l >> 4) sums a quarter row with packed adds.
#pragma once
#include "klg_kernels.hpp"
#include "klg_device_x2.hpp"

#pragma clang fp contract(off)

namespace klg {

enum { X2_CHUNK = 16, X2_LD = 65, X2_VOICES_PER_WG = 2 * WG };

struct Sub2aX2 {
	using Rec = PatchSub2a::Rec;
	// OSM (Saw, duty == 0: only Down / DownUpDown occur, see osm_saw_duty0)
	u2 offset, inc; f2 f, omf, c2, k2, nrcpf;
	// Biquad
	f2 b0, b1, b2, a1, a2, z0, z1;
	// ADSR
	f2 r_out, r_target, r_rate, time, A, AD, S;
	i2 estage, point, active;      // active: 0 / -1 mask
	i2 stage;                      // NoteBase::stage
	// derived from the envelope's stage / point / ramp, which only change on the rare path (sub2a_x2_derive refreshes them there)
	f2 tinc, srate;                // time increment of this sample (timeInc at Sustain, else +0); signed ramp step
	i2 special;                    // an idle ramp means work: Sustain before its last point, or Release
};
// Envelope::process klang.h:4018-4051 per sample: `time += timeInc` at Sustain; the ramp steps by +-rate towards its target (the
// sign is fixed while a ramp runs: out only moves towards target); an idle ramp at a segment end / in Release takes the rare path.
__device__ __forceinline__ void sub2a_x2_derive(Sub2aX2& L, const SampleRate& fs) {
	const i2 sustain = (L.estage == (int)ENV_SUSTAIN);
	L.tinc = sustain ? splat(fs.timeInc) : splat(0.f);                 // time >= 0: adding +0 leaves it bit for bit
	L.special = (sustain & (L.point != 2)) | (L.estage == (int)ENV_RELEASE);
	L.srate = (L.r_target > L.r_out) ? L.r_rate : -L.r_rate;
	L.stage = (L.estage == (int)ENV_OFF) ? (i2)(int)ST_OFF : L.stage;  // if (adsr.finished()) stop();
}

__device__ __forceinline__ void sub2a_x2_begin(Sub2aX2& L, const u2 (&w)[sizeof(PatchSub2a::Rec) / 4]) {
	// word order of PatchSub2a::Rec: flags | inc offset duty delta | b0 b1 b2 a1 a2 z0 z1 | r_out r_target r_rate time A AD S R
	const u2 flags = w[0];
	L.stage = __builtin_convertvector(flags & 3u, i2);
	const u2 eb = (flags >> 2) & 0x3Fu;
	L.estage = __builtin_convertvector(eb & 3u, i2);
	L.point = __builtin_convertvector((eb >> 2) & 7u, i2);
	L.active = -__builtin_convertvector((eb >> 5) & 1u, i2);
	L.inc = w[1]; L.offset = w[2];
	const f2 delta = as_f2(w[4]);
	L.f = delta;                                   // OSM::init klang.h:5206-5215 with col = 0
	L.omf = 1.f - L.f;
	const f2 rcpf = 1.f / L.f;
	L.nrcpf = -rcpf;
	const f2 col = phase_float2<0x7Fu>(w[3]) - 1.f;
	L.c2 = -1.f / (1.0f - col);
	L.k2 = L.c2 * L.omf;                           // `c2 * omf * (...)` associates left: (c2 * omf) is loop invariant
	L.b0 = as_f2(w[5]); L.b1 = as_f2(w[6]); L.b2 = as_f2(w[7]); L.a1 = as_f2(w[8]); L.a2 = as_f2(w[9]); L.z0 = as_f2(w[10]); L.z1 = as_f2(w[11]);
	L.r_out = as_f2(w[12]); L.r_target = as_f2(w[13]); L.r_rate = as_f2(w[14]); L.time = as_f2(w[15]);
	L.A = as_f2(w[16]); L.AD = as_f2(w[17]); L.S = as_f2(w[18]);
}


// rare path of one voice (element c): segment end / stage change, the scalar code of klg_device.hpp
template<int c>
__device__ __forceinline__ void sub2a_x2_rare(Sub2aX2& L, const SampleRate& fs) {
	Env e; e.r_out = L.r_out[c]; e.r_target = L.r_target[c]; e.r_rate = L.r_rate[c]; e.time = L.time[c];
	e.stage = L.estage[c]; e.point = L.point[c]; e.active = L.active[c] != 0;
	Pts3 p; p.x0 = 0.f; p.x1 = L.A[c]; p.x2 = L.AD[c]; p.y0 = 0.f; p.y1 = 1.f; p.y2 = L.S[c];
	env_segment_end<3, true>(e, p, 3, fs);
	L.r_out[c] = e.r_out; L.r_target[c] = e.r_target; L.r_rate[c] = e.r_rate; L.time[c] = e.time;
	L.estage[c] = e.stage; L.point[c] = e.point; L.active[c] = e.active ? -1 : 0;
}

__device__ __forceinline__ f2 sub2a_x2_osc_filter(Sub2aX2& L) {
	// ---- Fast::Saw (OSM, duty 0)  klang.h:5251-5302 ----
	// p = float(offset) - col with col == 0 is the 23-bit fraction m / 2^23 exactly, and only p + p is used below: the same
	// bits under exponent 2 are 2 + 2p, and taking 2 off is exact — one operation instead of two, same value
	const f2 pp = phase_float2<0x80u>(L.offset) - 2.f;
	const i2 carry = L.offset < L.inc;
	L.offset += L.inc;
	const f2 y_lin = L.c2 * (pp - L.f) + 1.f;
	const f2 y_wrap = L.nrcpf * (1.f + L.k2 * (pp + L.omf)) + 1.f;
	const f2 osc = carry ? y_wrap : y_lin;
	// ---- Biquad LPF, TDF-II  klang.h:5605-5612 ----
	const f2 y = L.b0 * osc + L.z0;
	L.z0 = L.b1 * osc - L.a1 * y + L.z1;
	L.z1 = L.b2 * osc - L.a2 * y;
	return y;
}

// A voice is QUIET when its envelope needs no work this block beyond the Sustain time counter: the ramp is idle
// and the voice either holds at the ADSR sustain point or is already Off.  Nothing inside a block can end that
// state (only host events between blocks do), so a wave whose voices are all quiet runs the short loop below.
__device__ __forceinline__ i2 sub2a_x2_quiet(const Sub2aX2& L) {
	return (L.r_out == L.r_target) & (((L.estage == (int)ENV_SUSTAIN) & (L.point == 2)) | (L.estage == (int)ENV_OFF));
}
__device__ __forceinline__ f2 sub2a_x2_sample_quiet(Sub2aX2& L) {
	const f2 y = sub2a_x2_osc_filter(L);
	L.time += L.tinc;                                                 // Envelope::process, case Sustain: time += timeInc
	return y * L.r_out;                                               // out *= adsr++ (ramp idle: value unchanged)
}

// one sample of a ramping voice WITHOUT the segment-end test (see the render loop)
__device__ __forceinline__ f2 sub2a_x2_step(Sub2aX2& L) {
	const f2 y = sub2a_x2_osc_filter(L);
	const f2 env = L.r_out;
	const f2 nxt = L.r_out + L.srate;
	L.r_out.x = __builtin_amdgcn_fmed3f(env.x, nxt.x, L.r_target.x);
	L.r_out.y = __builtin_amdgcn_fmed3f(env.y, nxt.y, L.r_target.y);
	L.time += L.tinc;
	return y * env;                                                    // out *= adsr++
}
__device__ __forceinline__ f2 sub2a_x2_sample(Sub2aX2& L, const SampleRate& fs) {
	const f2 y = sub2a_x2_osc_filter(L);
	// ---- ADSR: Envelope::process fast path  klang.h:4018-4051 (see env_process) ----
	// Linear's `active` is exactly (out != target): setTarget sets it so (klang.h:3756), the ramp clears it when it clamps to the target
	// (3795-3805), setValue makes both equal.  So the flag needs no register: the step is the median of (out, out +- rate, target) for
	// every voice — an idle ramp has out == target and the median of (x, anything, x) is x — and "idle" is one compare.
	const f2 env = L.r_out;
	const f2 nxt = L.r_out + L.srate;
	L.r_out.x = __builtin_amdgcn_fmed3f(env.x, nxt.x, L.r_target.x);
	L.r_out.y = __builtin_amdgcn_fmed3f(env.y, nxt.y, L.r_target.y);
	L.time += L.tinc;
	const i2 rare = (L.r_out == L.r_target) & L.special;              // an idle ramp that means work: a segment end, or the end of the release
	if (__ballot((rare.x | rare.y) != 0) != 0ull) {
		L.active = L.r_out != L.r_target;
		if (rare.x) sub2a_x2_rare<0>(L, fs);
		if (rare.y) sub2a_x2_rare<1>(L, fs);
		sub2a_x2_derive(L, fs);
	}
	return y * env;                                                    // out *= adsr++
}

__device__ __forceinline__ void sub2a_x2_end(const Sub2aX2& L, u2& flags, u2& offset, u2& z0, u2& z1, u2& r_out, u2& r_target, u2& r_rate, u2& time) {
	const i2 active = L.r_out != L.r_target;                                 // Linear::active (see sub2a_x2_sample)
	const u2 eb = __builtin_convertvector(L.estage, u2) | (__builtin_convertvector(L.point, u2) << 2) | ((__builtin_convertvector(active, u2) & 1u) << 5);
	flags = __builtin_convertvector(L.stage, u2) | (eb << 2);                 // osm state bits stay 0 (Down)
	offset = L.offset; z0 = as_u2(L.z0); z1 = as_u2(L.z1);
	r_out = as_u2(L.r_out); r_target = as_u2(L.r_target); r_rate = as_u2(L.r_rate); time = as_u2(L.time);
}

template<bool PER_VOICE>
__global__ __launch_bounds__(WG) void klg_render_sub2a_x2(const RenderArgs a) {
	constexpr int W = sizeof(PatchSub2a::Rec) / 4;
	__shared__ __attribute__((aligned(16))) float lds[WAVES * X2_CHUNK * X2_LD * 2];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	f2* tile = reinterpret_cast<f2*>(lds) + wave * X2_CHUNK * X2_LD;
	const int n = a.n;
	float* acc = klg_mix_rows + wave * n;                    // this wave's own mix row (klg_kernels.hpp)
	for (int i = lane; i < n; i += 64) acc[i] = 0.f;
	wave_sync();
	fused_events<PatchSub2a>(a, X2_VOICES_PER_WG);            // (small banks: this block's note events in the same launch, klg_kernels.hpp)

	const int groups = (int)((a.stride + X2_VOICES_PER_WG - 1) / X2_VOICES_PER_WG);
	for (int g = blockIdx.x; g < groups; g += gridDim.x) {
		const int v0 = g * X2_VOICES_PER_WG + wave * 128;         // this wave's first voice
		const int v = v0 + 2 * lane;                              // this lane's first voice (v, v + 1)
		const bool in_range = (size_t)v < a.stride;              // stride is a multiple of 256 and v is even: v + 1 is in range too
		u2 w[W];
		w[0] = in_range ? *reinterpret_cast<const u2*>(a.state + v) : (u2)(unsigned)ST_OFF;
		i2 live = ((w[0] & 3u) != (unsigned)ST_OFF);
		live.x = (v < a.voices) ? live.x : 0; live.y = (v + 1 < a.voices) ? live.y : 0;
		const bool any_live = (live.x | live.y) != 0;
		if (__ballot(any_live) == 0ull) {
			if (PER_VOICE)
				for (int j = 0; j < 128 && v0 + j < a.voices; j++)
					for (int i = lane; i < n; i += 64) a.per_voice[(size_t)(v0 + j) * n + i] = 0.f;
			continue;
		}
		w[0] = live ? w[0] : (u2)0u;
#pragma unroll
		for (int k = 1; k < W; k++) {
			const u2 t = any_live ? *reinterpret_cast<const u2*>(a.state + (size_t)k * a.stride + v) : (u2)0u;
			w[k] = live ? t : (u2)0u;
		}
		Sub2aX2 L;
		sub2a_x2_begin(L, w);
		sub2a_x2_derive(L, a.fs);
		for (int c0 = 0; c0 < n; c0 += X2_CHUNK) {
			const int cl = (n - c0 < X2_CHUNK) ? (n - c0) : X2_CHUNK;
			const i2 quiet = sub2a_x2_quiet(L);
			// (no `live ? y : 0` at the tile write: a dead lane runs on an all-zero record — b0 = 0 and r_out = 0 — whose
			//  output is +0 by itself: osc = y_lin = 1, y = 0 * 1 + 0, 0 * 0)
			if (__ballot((quiet.x & quiet.y) == 0) == 0ull) {         // every voice of the wave is quiet
				if (cl == X2_CHUNK) {
#pragma unroll 4
					for (int s = 0; s < X2_CHUNK; s++) tile[s * X2_LD + lane] = sub2a_x2_sample_quiet(L);
				}
				else for (int s = 0; s < cl; s++) tile[s * X2_LD + lane] = sub2a_x2_sample_quiet(L);
			}
			else {
				// Some voice of the wave is ramping.  The ramp step itself is three operations for everybody (sub2a_x2_step); what costs is the
				// segment-end code: kept INSIDE the sample loop it makes every envelope register a phi of "came round the loop" and "came out of
				// the rare path" — some twenty register moves per sample.  So the sample loop only DETECTS a segment end and leaves; the rare
				// code runs between two runs of the loop.
				int s = 0;
				// Most chunks of a ramp cannot reach its end: a voice still further from its target than (chunk + 2) steps (plus what sixteen
				// roundings of a value <= 2 can add up to) neither clamps nor goes idle inside the chunk, so the chunk runs the bare step with no
				// per-sample test at all (a wave-uniform decision; voices whose idle ramp means nothing — holding at sustain, Off — do not count).
				{
					const f2 d = __builtin_elementwise_abs(L.r_target - L.r_out);
					const f2 lim = __builtin_elementwise_abs(L.srate) * (float)(X2_CHUNK + 2) + 4e-6f;
					const i2 may = ~(d > lim) & L.special;                            // (not `d <= lim`: a NaN — the rate of a zero-length segment — must count as "may")
					if (cl == X2_CHUNK && __ballot((may.x | may.y) != 0) == 0ull) {
						// ... and without the clamp: a ramp that cannot arrive steps by exactly its rate (the median of (out, out + rate, target) IS
						// out + rate), an idle one by -0.0 (x + -0.0 == x for every x, either zero included)
						const f2 step = (L.r_out != L.r_target) ? L.srate : splat(-0.f);
#pragma unroll 4
						for (; s < X2_CHUNK; s++) {
							const f2 y = sub2a_x2_osc_filter(L);
							const f2 env = L.r_out;
							L.r_out = env + step;
							L.time += L.tinc;
							tile[s * X2_LD + lane] = y * env;                          // out *= adsr++
						}
					}
				}
				while (s < cl) {
					i2 rare = (i2)0;
#pragma unroll 4
					for (; s < cl;) {
						tile[s * X2_LD + lane] = sub2a_x2_step(L);
						s++;
						rare = (L.r_out == L.r_target) & L.special;               // an idle ramp that means work: a segment end, or the end of the release
						if (__ballot((rare.x | rare.y) != 0) != 0ull) break;
					}
					if (__ballot((rare.x | rare.y) != 0) != 0ull) {
						L.active = L.r_out != L.r_target;
						if (rare.x) sub2a_x2_rare<0>(L, a.fs);
						if (rare.y) sub2a_x2_rare<1>(L, a.fs);
						sub2a_x2_derive(L, a.fs);
					}
				}
			}
			wave_sync();
			if (PER_VOICE) {
				constexpr int Q = 64 / X2_CHUNK;                          // lane groups per sample row
				const int s = lane & (X2_CHUNK - 1), q = lane / X2_CHUNK;
				const float* tf = reinterpret_cast<const float*>(tile);
				for (int j = 0; j < 128 / Q; j++) {
					const int vv = Q * j + q;                             // voice within the wave's 128
					if (s < cl && v0 + vv < a.voices) a.per_voice[(size_t)(v0 + vv) * n + c0 + s] = tf[(s * X2_LD) * 2 + vv];
				}
			}
			{
				constexpr int Q = 64 / X2_CHUNK;                          // lane groups per sample row, each sums 64 / Q pairs
				const int s = lane & (X2_CHUNK - 1), q = lane / X2_CHUNK;
				f2 sum2 = splat(0.f);
				if (s < cl) {
					const f2* row = tile + s * X2_LD + q * (64 / Q);
#pragma unroll
					for (int j = 0; j < 64 / Q; j++) sum2 += row[j];
				}
				float sum = sum2.x + sum2.y;
#pragma unroll
				for (int m = X2_CHUNK; m < 64; m <<= 1) sum += __shfl_xor(sum, m);
				if (lane < cl) acc[c0 + lane] += sum;
			}
			wave_sync();
		}
		if (any_live) {
			u2 flags, offset, z0, z1, r_out, r_target, r_rate, time;
			sub2a_x2_end(L, flags, offset, z0, z1, r_out, r_target, r_rate, time);
			// only the words PatchSub2a::kStoreMask names: flags(0) offset(2) z0 z1 (10,11) r_out..time (12..15)
			auto st = [&](int k, u2 val) {
				u2* dst = reinterpret_cast<u2*>(a.state + (size_t)k * a.stride + v);
				if (live.x && live.y) *dst = val;
				else if (live.x) a.state[(size_t)k * a.stride + v] = val.x;
				else if (live.y) a.state[(size_t)k * a.stride + v + 1] = val.y;
			};
			st(0, flags); st(2, offset); st(10, z0); st(11, z1); st(12, r_out); st(13, r_target); st(14, r_rate); st(15, time);
		}
	}
	__syncthreads();
	for (int i = tid; i < n; i += WG) a.partials[(size_t)blockIdx.x * n + i] = mix_rows_sum(i, n);
	fused_combine(a, n, 1, reinterpret_cast<int*>(lds));
}

// ---------------------------------------------------------------------------------------------
// The same shape for ANY patch written over the packed primitives of klg_device_x2.hpp: what a generated graph patch
// (klg_graph.hpp) uses when its nodes all have packed forms.  P: Rec { u2 w[kWords]; }, Live, begin / sample / end over
// BlockCtx2, kStoreMask / kStoreMask2 as for klg_render<P>.
// ---------------------------------------------------------------------------------------------
struct BlockCtx2 {
	SampleRate fs;
	const float* ctl0; const float* ctl1;     // the two voices' synth instances (they differ when notes_per_synth is odd)
	const TableDesc* tables;
	const uint32_t* rec; size_t stride;       // the pair's records in HBM: word w of voice 0 at rec[w * stride], of voice 1 one further (what is read only when a segment ends: PtsNx2)
};
__device__ __forceinline__ float ctl_read(const BlockCtx& c, unsigned i) { return c.ctl[i]; }
__device__ __forceinline__ f2 ctl_read(const BlockCtx2& c, unsigned i) { f2 r = { c.ctl0[i], c.ctl1[i] }; return r; }

template<class P, bool PER_VOICE>
__global__ __launch_bounds__(WG) void klg_render_x2(const RenderArgs a) {
	constexpr int W = P::kWords;
	__shared__ __attribute__((aligned(16))) float lds[WAVES * X2_CHUNK * X2_LD * 2];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	f2* tile = reinterpret_cast<f2*>(lds) + wave * X2_CHUNK * X2_LD;
	const int n = a.n;
	float* acc = klg_mix_rows + wave * n;                    // this wave's own mix row (klg_kernels.hpp)
	for (int i = lane; i < n; i += 64) acc[i] = 0.f;
	wave_sync();

	const int groups = (int)((a.stride + X2_VOICES_PER_WG - 1) / X2_VOICES_PER_WG);
	for (int g = blockIdx.x; g < groups; g += gridDim.x) {
		const int v0 = g * X2_VOICES_PER_WG + wave * 128;
		const int v = v0 + 2 * lane;
		const bool in_range = (size_t)v < a.stride;
		typename P::Rec rec;
		rec.w[0] = in_range ? *reinterpret_cast<const u2*>(a.state + v) : (u2)(unsigned)ST_OFF;
		i2 live = ((rec.w[0] & 3u) != (unsigned)ST_OFF);
		live.x = (v < a.voices) ? live.x : 0; live.y = (v + 1 < a.voices) ? live.y : 0;
		const bool any_live = (live.x | live.y) != 0;
		if (__ballot(any_live) == 0ull) {
			if (PER_VOICE)
				for (int j = 0; j < 128 && v0 + j < a.voices; j++)
					for (int i = lane; i < n; i += 64) a.per_voice[(size_t)(v0 + j) * n + i] = 0.f;
			continue;
		}
		rec.w[0] = live ? rec.w[0] : (u2)0u;
#pragma unroll
		for (int k = 1; k < W; k++) {
			const u2 t = any_live ? *reinterpret_cast<const u2*>(a.state + (size_t)k * a.stride + v) : (u2)0u;
			rec.w[k] = live ? t : (u2)0u;
		}
		// KLG_MIX_LAST_ACTIVE (klg_kernels.hpp): only one voice of each synth instance is heard; the per-voice dump keeps every voice
		i2 heard = live;
		if (a.solo) {
			heard.x = (live.x && a.solo[v / a.notes_per_synth] == v) ? -1 : 0;
			heard.y = (live.y && a.solo[(v + 1) / a.notes_per_synth] == v + 1) ? -1 : 0;
		}
		const unsigned long long heard_x = __ballot(heard.x != 0), heard_y = __ballot(heard.y != 0);
		if (PER_VOICE) heard = live;                              // the tile then holds every voice; the selection happens in the sum
		typename P::Live L;
		BlockCtx2 ctx;
		ctx.fs = a.fs; ctx.tables = a.tables;
		ctx.ctl0 = a.controls + (size_t)((v < a.voices ? v : 0) / a.notes_per_synth) * KLG_MAX_CTL;
		ctx.ctl1 = a.controls + (size_t)((v + 1 < a.voices ? v + 1 : 0) / a.notes_per_synth) * KLG_MAX_CTL;
		ctx.rec = a.state + (in_range ? v : 0); ctx.stride = a.stride;       // (a pair outside the planes computes on zeros and keeps nothing: it may read any record)
		P::begin(L, rec, ctx);
		for (int c0 = 0; c0 < n; c0 += X2_CHUNK) {
			const int cl = (n - c0 < X2_CHUNK) ? (n - c0) : X2_CHUNK;
			const int quiet = P::kHasQuiet ? P::quiet(L) : 0;         // 0: full body, 1: envelopes holding, 2: ... and duty-0 saws
			if (quiet == 2) {
				if (cl == X2_CHUNK) {
#pragma unroll 4
					for (int s = 0; s < X2_CHUNK; s++) { const f2 y = P::sample_fast(L, ctx); tile[s * X2_LD + lane] = heard ? y : splat(0.f); }
				}
				else for (int s = 0; s < cl; s++) { const f2 y = P::sample_fast(L, ctx); tile[s * X2_LD + lane] = heard ? y : splat(0.f); }
			}
			else if (quiet == 1) {
				for (int s = 0; s < cl; s++) { const f2 y = P::sample_quiet(L, ctx); tile[s * X2_LD + lane] = heard ? y : splat(0.f); }
			}
			else {
				for (int s = 0; s < cl; s++) {
					const f2 y = P::sample(L, ctx);
					tile[s * X2_LD + lane] = heard ? y : splat(0.f);
				}
			}
			wave_sync();
			if (PER_VOICE) {
				constexpr int Q = 64 / X2_CHUNK;
				const int s = lane & (X2_CHUNK - 1), q = lane / X2_CHUNK;
				const float* tf = reinterpret_cast<const float*>(tile);
				for (int j = 0; j < 128 / Q; j++) {
					const int vv = Q * j + q;
					if (s < cl && v0 + vv < a.voices) a.per_voice[(size_t)(v0 + vv) * n + c0 + s] = tf[(s * X2_LD) * 2 + vv];
				}
			}
			{
				constexpr int Q = 64 / X2_CHUNK;
				const int s = lane & (X2_CHUNK - 1), q = lane / X2_CHUNK;
				f2 sum2 = splat(0.f);
				if (s < cl) {
					const f2* row = tile + s * X2_LD + q * (64 / Q);
					if (PER_VOICE && a.solo) {
						for (int j = 0; j < 64 / Q; j++) { const int src = q * (64 / Q) + j; f2 t = row[j]; t.x = ((heard_x >> src) & 1ull) ? t.x : 0.f; t.y = ((heard_y >> src) & 1ull) ? t.y : 0.f; sum2 += t; }
					}
					else {
#pragma unroll
						for (int j = 0; j < 64 / Q; j++) sum2 += row[j];
					}
				}
				float sum = sum2.x + sum2.y;
#pragma unroll
				for (int m = X2_CHUNK; m < 64; m <<= 1) sum += __shfl_xor(sum, m);
				if (lane < cl) acc[c0 + lane] += sum;
			}
			wave_sync();
		}
		if (any_live) {
			P::end(L, rec);
#pragma unroll
			for (int k = 0; k < W; k++) if (patch_stores<P>(k)) {
				u2* dst = reinterpret_cast<u2*>(a.state + (size_t)k * a.stride + v);
				if (live.x && live.y) *dst = rec.w[k];
				else if (live.x) a.state[(size_t)k * a.stride + v] = rec.w[k].x;
				else if (live.y) a.state[(size_t)k * a.stride + v + 1] = rec.w[k].y;
			}
		}
	}
	__syncthreads();
	for (int i = tid; i < n; i += WG) a.partials[(size_t)blockIdx.x * n + i] = mix_rows_sum(i, n);
}

} // namespace klg
