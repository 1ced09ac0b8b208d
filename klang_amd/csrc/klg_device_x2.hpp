// klang_amd/csrc/klg_device_x2.hpp — the device primitives of klg_device.hpp for TWO voices per lane.
//
// Same arithmetic, same operation order, packed: every value is a 2-vector (one component per voice), so the fp32 multiplies
// and adds issue as v_pk_mul_f32 / v_pk_add_f32 and each half is an ordinary IEEE operation (bit-identical to the scalar
// primitives).  Integer state machines and selects run per component; the rare, branchy paths (an envelope reaching a
// segment end) extract the component and call the SCALAR code of klg_device.hpp.  The functions overload the scalar names, so
// a generated patch body (klg_graph.hpp) is the same text for one or two voices per lane — only its types differ.
#pragma once
#include "klg_device.hpp"

#pragma clang fp contract(off)

namespace klg {

// (f2 / i2 / u2, as_f2, as_u2, splat, phase_float2: klg_device.hpp)

// ---- helpers a generated body uses for both widths ----
__device__ __forceinline__ f2 u2f(u2 v) { return as_f2(v); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ u2 f2u(f2 v) { return as_u2(v); }
__device__ __forceinline__ int to_i(uint32_t u) { return (int)u; }
__device__ __forceinline__ i2 to_i(u2 u) { return __builtin_convertvector(u, i2); }
__device__ __forceinline__ uint32_t to_u(int i) { return (uint32_t)i; }
__device__ __forceinline__ u2 to_u(i2 i) { return __builtin_convertvector(i, u2); }
template<class F> __device__ __forceinline__ F kf(uint32_t bits);                                  // a literal
template<> __device__ __forceinline__ float kf<float>(uint32_t bits) { return u2f(bits); }
template<> __device__ __forceinline__ f2 kf<f2>(uint32_t bits) { return splat(u2f(bits)); }
__device__ __forceinline__ float cmp_sel(bool c) { return c ? 1.f : 0.f; }
__device__ __forceinline__ f2 cmp_sel(i2 c) { return c ? splat(1.f) : splat(0.f); }
__device__ __forceinline__ int stage_off_if(bool off, int stage) { return off ? (int)ST_OFF : stage; }
__device__ __forceinline__ i2 stage_off_if(i2 off, i2 stage) { return off ? (i2)(int)ST_OFF : stage; }
__device__ __forceinline__ int stage_all_off(int) { return (int)ST_OFF; }
__device__ __forceinline__ i2 stage_all_off(i2) { return (i2)(int)ST_OFF; }
__device__ __forceinline__ int loop_index(uint32_t v) { return v == 255u ? -1 : (int)v; }              // record byte of Envelope::Loop start / end: 255 = none
__device__ __forceinline__ i2 loop_index(u2 v) { return (v == 255u) ? (i2)(-1) : to_i(v); }
__device__ __forceinline__ bool env_is_off(int stage) { return stage == ENV_OFF; }
__device__ __forceinline__ bool env_is_sustain(int stage) { return stage == ENV_SUSTAIN; }
__device__ __forceinline__ bool env_is_release(int stage) { return stage == ENV_RELEASE; }
__device__ __forceinline__ i2 env_is_off(i2 stage) { return stage == (int)ENV_OFF; }

// ---- Generators::Fast::Sine klang.h:5093-5172 ----
struct FSine2 { i2 inc; u2 pos; };
// (polysin(f2), fastsinp(u2): klg_device.hpp)
__device__ __forceinline__ f2 fsine_process(FSine2& o, u2 off) { const f2 y = fastsinp(o.pos + off); o.pos += to_u(o.inc); return y; }
__device__ __forceinline__ f2 fsine_process(FSine2& o, uint32_t off) { u2 v = { off, off }; return fsine_process(o, v); }

// ---- Fast::OSM klang.h:5175-5317 ----
struct Osm2 { i2 inc; u2 offset, duty; i2 state; f2 delta, f, omf, rcpf, rcpf2, col, c1, c2; };
__device__ __forceinline__ void osm_derive(Osm2& o) {                           // OSM::init 5206-5215
	o.f = o.delta;
	o.omf = 1.f - o.f;
	o.rcpf = 1.f / o.f;
	o.rcpf2 = 2.f * o.rcpf;
	o.col = phase_float2<0x7Fu>(o.duty) - 1.f;
	o.c1 = 1.f / o.col;
	o.c2 = -1.f / (1.0f - o.col);
}
__device__ __forceinline__ i2 osm_tick(Osm2& o) {                               // klang.h:5251-5263
	const u2 uinc = to_u(o.inc);
	o.state = ((o.state << 1) | ((o.offset < o.duty) & 1)) & 3;
	const i2 tr = o.state | ((o.offset < uinc) & 4);
	o.offset += uinc;
	return tr;
}
__device__ __forceinline__ f2 sqrf(f2 x) { return x * x; }
__device__ __forceinline__ f2 osm_saw(Osm2& o) {                                // saw() 5290-5302 (see the scalar osm_saw)
	const f2 p = (phase_float2<0x7Fu>(o.offset) - 1.f) - o.col;
	const i2 tr = osm_tick(o);
	const f2 f = o.f, omf = o.omf, rcpf = o.rcpf, c1 = o.c1, c2 = o.c2;
	const i2 n_up = (tr & 1) != 0, o_up = (tr & 2) != 0, carry = (tr & 4) != 0;
	const f2 cN = n_up ? c1 : c2;
	const f2 pp = p + p;
	const f2 y_lin = cN * (pp - f) + 1.f;
	const f2 y_wrap = -rcpf * (1.f + cN * omf * (pp + omf)) + 1.f;
	const f2 p2 = sqrf(p);
	const f2 y_ud = rcpf * (c2 * p2 - c1 * sqrf(p - f)) + 1.f;
	const f2 y_du = -rcpf * (1.f + c2 * sqrf(p + omf) - c1 * p2) + 1.f;
	const f2 y_same = carry ? y_wrap : y_lin;
	const f2 y_edge = carry ? y_du : y_ud;
	const i2 valid_edge = carry ? (~o_up & n_up) : (o_up & ~n_up);
	return (o_up == n_up) ? y_same : (valid_edge ? y_edge : splat(0.f));
}
__device__ __forceinline__ f2 osm_saw_duty0(Osm2& o) {
	const f2 pp = phase_float2<0x80u>(o.offset) - 2.f;                   // 2p in one operation (see the scalar osm_saw_duty0)
	const u2 uinc = to_u(o.inc);
	const i2 carry = o.offset < uinc;
	o.offset += uinc;
	const f2 y_lin = o.c2 * (pp - o.f) + 1.f;
	const f2 y_wrap = -o.rcpf * (1.f + o.c2 * o.omf * (pp + o.omf)) + 1.f;
	return carry ? y_wrap : y_lin;
}
// no voice of the wave has a duty (and every state is Down): the short form is exact.  Block-invariant unless set() runs per sample.
__device__ __forceinline__ bool osm_is_duty0(const Osm& o) { return __ballot(o.duty != 0u || o.state != 0) == 0ull; }
__device__ __forceinline__ bool osm_is_duty0(const Osm2& o) { const i2 general = (o.duty != 0u) | (o.state != 0); return __ballot((general.x | general.y) != 0) == 0ull; }
__device__ __forceinline__ f2 osm_saw_auto(Osm2& o) {
	const i2 general = (o.duty != 0u) | (o.state != 0);
	if (__ballot((general.x | general.y) != 0) == 0ull) return osm_saw_duty0(o);
	return osm_saw(o);
}
__device__ __forceinline__ f2 osm_pulse(Osm2& o) {                              // pulse() 5304-5316
	const f2 p = phase_float2<0x7Fu>(o.offset) - 1.f;
	const i2 tr = osm_tick(o);
	const f2 rcpf2 = o.rcpf2, col = o.col;
	const i2 n_up = (tr & 1) != 0, o_up = (tr & 2) != 0, carry = (tr & 4) != 0;
	const f2 y_flat = n_up ? splat(1.f) : splat(-1.f);
	const f2 y_wrap = n_up ? (rcpf2 * (col - 1.0f) + 1.f) : (rcpf2 * col - 1.f);
	const f2 y_ud = rcpf2 * (col - p) + 1.f;
	const f2 y_du = rcpf2 * p - 1.f;
	const f2 y_same = carry ? y_wrap : y_flat;
	const f2 y_edge = carry ? y_du : y_ud;
	const i2 valid_edge = carry ? (~o_up & n_up) : (o_up & ~n_up);
	return (o_up == n_up) ? y_same : (valid_edge ? y_edge : splat(0.f));
}

// ---- Filters::Biquad process, TDF-II klang.h:5605-5612 ----
struct Biquad2 { f2 b0, b1, b2, a1, a2, z0, z1; };
struct BiquadSweep2 { f2 f, Q; };
__device__ __forceinline__ f2 biquad_process(Biquad2& q, f2 in) {
	const f2 z0 = q.z0, z1 = q.z1;
	const f2 y = q.b0 * in + z0;
	q.z0 = q.b1 * in - q.a1 * y + z1;
	q.z1 = q.b2 * in - q.a2 * y;
	return y;
}

// ---- Envelope / ADSR klang.h:3867-4137: the ramp step is packed, a segment end runs the scalar code on that component ----
struct Env2 { f2 r_out, r_target, r_rate, time; i2 stage, point, active; };   // active: 0 / -1
struct Pts4x2 { f2 x0, x1, x2, x3, y0, y1, y2, y3; };
__device__ __forceinline__ void env_unpack(Env2& e, u2 b) { e.stage = to_i(b & 3u); e.point = to_i((b >> 2) & 7u); e.active = -to_i((b >> 5) & 1u); }
__device__ __forceinline__ u2 env_pack(const Env2& e) { return to_u(e.stage) | (to_u(e.point) << 2) | ((to_u(e.active) & 1u) << 5); }
template<int c> __device__ __forceinline__ Env env_get(const Env2& e) { Env s; s.r_out = e.r_out[c]; s.r_target = e.r_target[c]; s.r_rate = e.r_rate[c]; s.time = e.time[c]; s.stage = e.stage[c]; s.point = e.point[c]; s.active = e.active[c] != 0; return s; }
template<int c> __device__ __forceinline__ void env_put(Env2& e, const Env& s) { e.r_out[c] = s.r_out; e.r_target[c] = s.r_target; e.r_rate[c] = s.r_rate; e.time[c] = s.time; e.stage[c] = s.stage; e.point[c] = s.point; e.active[c] = s.active ? -1 : 0; }
template<int c> __device__ __forceinline__ Pts4 pts_get(const Pts4x2& p) { Pts4 s; s.x0 = p.x0[c]; s.x1 = p.x1[c]; s.x2 = p.x2[c]; s.x3 = p.x3[c]; s.y0 = p.y0[c]; s.y1 = p.y1[c]; s.y2 = p.y2[c]; s.y3 = p.y3[c]; return s; }
// the packed ramp step shared by Envelope and ADSR: returns the pre-step value, leaves `sustain` for the caller's rare test
__device__ __forceinline__ f2 env_step(Env2& e, const SampleRate& fs, i2& sustain) {
	const f2 out = e.r_out;
	const i2 up = e.r_target > e.r_out;
	const f2 nxt = e.r_out + (up ? e.r_rate : -e.r_rate);
	f2 stepped;
	stepped.x = __builtin_amdgcn_fmed3f(e.r_out.x, nxt.x, e.r_target.x);
	stepped.y = __builtin_amdgcn_fmed3f(e.r_out.y, nxt.y, e.r_target.y);
	e.r_out = e.active ? stepped : e.r_out;
	e.active = e.active & (stepped != e.r_target);
	sustain = (e.stage == (int)ENV_SUSTAIN);
	e.time = sustain ? (e.time + fs.timeInc) : e.time;
	return out;
}
__device__ __forceinline__ f2 y_at(const Pts4x2& p, i2 i) { return (i == 3) ? p.y3 : ((i == 2) ? p.y2 : ((i == 1) ? p.y1 : p.y0)); }
// more than four point slots: the pair's record words in HBM behind the four in registers (PtsN, klg_device.hpp); voice 1's words are one further
struct PtsNx2 { Pts4x2 head; const uint32_t* ext; size_t stride; int slots; };
template<int c> __device__ __forceinline__ PtsN pts_get(const PtsNx2& p) { PtsN s; s.head = pts_get<c>(p.head); s.ext = p.ext + c; s.stride = p.stride; s.slots = p.slots; return s; }
__device__ __forceinline__ f2 env_hold_y(const Pts4x2& p, i2 ls) { return y_at(p, ls); }
__device__ __forceinline__ f2 env_hold_y(const PtsNx2& p, i2 ls) { f2 r = { env_hold_y(pts_get<0>(p), ls.x), env_hold_y(pts_get<1>(p), ls.y) }; return r; }
// npm: point count | Rate mode << 16 per voice (env_process_rt, klg_device.hpp)
__device__ __forceinline__ void env_unpack_rt(Env2& e, i2& npm, u2 b, u2 npoints) {
	e.stage = to_i(b & 3u); e.point = to_i(((b >> 2) & 7u) | ((b >> 7) << 3)); e.active = -to_i((b >> 5) & 1u);
	npm = to_i(npoints & 0xFFFFu) | (to_i((b >> 6) & 1u) << 16);
}
__device__ __forceinline__ u2 env_pack_rt(const Env2& e, i2 npm) { return to_u(e.stage) | ((to_u(e.point) & 7u) << 2) | ((to_u(e.active) & 1u) << 5) | (((to_u(npm) >> 16) & 1u) << 6) | ((to_u(e.point) >> 3) << 7); }
template<class PTS2>
__device__ __forceinline__ f2 env_process_rt(Env2& e, const PTS2& p, i2 npm, i2 ls, i2 le, f2 hold_y, const SampleRate& fs) {
	i2 sustain;
	const f2 out = env_step(e, fs, sustain);
	const i2 settled = (ls >= 0) & (ls == le) & (e.point == ls) & (e.r_out == hold_y);
	const i2 rare = ~e.active & ((sustain & ~settled) | (e.stage == (int)ENV_RELEASE));
	if (__ballot((rare.x | rare.y) != 0) != 0ull) {
		if (rare.x) { Env s = env_get<0>(e); env_segment_end_rt(s, pts_get<0>(p), npm.x, ls.x, le.x, fs); env_put<0>(e, s); }
		if (rare.y) { Env s = env_get<1>(e); env_segment_end_rt(s, pts_get<1>(p), npm.y, ls.y, le.y, fs); env_put<1>(e, s); }
	}
	return out;
}
// ADSR: what only changes when a segment ends — the Sustain time increment, the signed ramp step, whether an idle ramp means
// work (Sustain before the hold point, or Release) — is kept beside the envelope and refreshed on the rare path (adsr_derive).
struct Adsr2 { Env2 e; f2 A, AD, S, R; f2 tinc, srate; i2 special; };
template<int c> __device__ __forceinline__ Pts3 adsr_pts(const Adsr2& a) { Pts3 p; p.x0 = 0.f; p.x1 = a.A[c]; p.x2 = a.AD[c]; p.y0 = 0.f; p.y1 = 1.f; p.y2 = a.S[c]; return p; }
__device__ __forceinline__ void adsr_derive(Adsr2& a, const SampleRate& fs) {
	const i2 sustain = (a.e.stage == (int)ENV_SUSTAIN);
	a.tinc = sustain ? splat(fs.timeInc) : splat(0.f);                   // time >= 0: adding +0 leaves it bit for bit
	a.special = (sustain & (a.e.point != 2)) | (a.e.stage == (int)ENV_RELEASE);
	a.srate = (a.e.r_target > a.e.r_out) ? a.e.r_rate : -a.e.r_rate;      // the sign is fixed while a ramp runs
}
__device__ __forceinline__ void adsr_derive(Adsr&, const SampleRate&) {}
__device__ __forceinline__ f2 adsr_process(Adsr2& a, const SampleRate& fs) {      // env_process<3, true>, see sub2a_x2_sample
	Env2& e = a.e;
	const f2 out = e.r_out;
	const f2 nxt = e.r_out + a.srate;
	f2 stepped;
	stepped.x = __builtin_amdgcn_fmed3f(e.r_out.x, nxt.x, e.r_target.x);
	stepped.y = __builtin_amdgcn_fmed3f(e.r_out.y, nxt.y, e.r_target.y);
	e.r_out = e.active ? stepped : e.r_out;
	e.active = e.active & (stepped != e.r_target);
	e.time += a.tinc;
	const i2 rare = ~e.active & a.special;
	if (__ballot((rare.x | rare.y) != 0) != 0ull) {
		if (rare.x) { Env s = env_get<0>(e); env_segment_end<3, true>(s, adsr_pts<0>(a), 3, fs); env_put<0>(e, s); }
		if (rare.y) { Env s = env_get<1>(e); env_segment_end<3, true>(s, adsr_pts<1>(a), 3, fs); env_put<1>(e, s); }
		adsr_derive(a, fs);
	}
	return out;
}

// An ADSR is QUIET when nothing but the Sustain clock runs: the ramp is idle and no segment end is pending (holding at the
// sustain point, or Off).  Only host events between blocks end that, so a wave whose envelopes are all quiet can run a
// chunk with adsr_hold in place of adsr_process (klg_render_x2<P>, P::sample_quiet).
__device__ __forceinline__ i2 adsr_quiet(const Adsr2& a) { return ~a.e.active & ~a.special; }
__device__ __forceinline__ f2 adsr_hold(Adsr2& a, const SampleRate&) { a.e.time += a.tinc; return a.e.r_out; }
__device__ __forceinline__ void adsr_set_points(Adsr2& a, f2 A, f2 AD, f2 S, f2 R) { a.A = A; a.AD = AD; a.S = S; a.R = R; }

// x / Y for a constant Y of the verified set (div_const of klg_device.hpp), both voices
template<uint32_t YBITS> __device__ __forceinline__ f2 div_const(f2 x) { f2 r; r.x = div_const<YBITS>(x.x); r.y = div_const<YBITS>(x.y); return r; }

} // namespace klg
