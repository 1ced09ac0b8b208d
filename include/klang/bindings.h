// include/klang/bindings.h — which GPU kernel renders which Note type, and how a host-side Note object maps onto the
// lane record of that kernel (include/klang_mi355_records.h).  A binder names the members a patch must have; bind a
// concrete Note type with KLANG_GPU_BIND(NoteType, KLG_PATCH_x, binder<NoteType>) after the patch's definition.
//
//   shipped templates/juce/synth/Source/subtractive.k : KLANG_GPU_BIND(Subtractive::SubtractiveNote, KLG_PATCH_SUB2B, klang::gpu::SubtractiveK<Subtractive::SubtractiveNote>)
//   shipped examples/SuperSaw.k                        : KLANG_GPU_BIND(SuperSaw::MyNote, KLG_PATCH_SUPERSAW, klang::gpu::SuperSawK<SuperSaw::MyNote>)
//   config 2a (Saw >> LPF >> ADSR)                     : KLANG_GPU_BIND(X::MyNote, KLG_PATCH_SUB2A, klang::gpu::SawLpfAdsr<X::MyNote>)
#pragma once
#include "klang.h"

namespace klang { namespace gpu {

// only the words the kernels change are copied back into the host mirror (coefficients, increments and breakpoints
// are host-owned and already there)
inline void env_to_rec(const klg::host::EnvH& e, float& r_out, float& r_target, float& r_rate, float& time) { r_out = e.r_out; r_target = e.r_target; r_rate = e.r_rate; time = e.time; }
inline void env_from_rec(klg::host::EnvH& e, float r_out, float r_target, float r_rate, float time, uint32_t bits) {
	e.r_out = r_out; e.r_target = r_target; e.r_rate = r_rate; e.time = time;
	e.stage = (int)(bits & 3u); e.point = (int)((bits >> 2) & 7u); e.active = ((bits >> 5) & 1u) != 0;
}
// the hand-written kernels' records hold Time-mode envelopes of at most three points: anything else stops here (never a truncated or re-interpreted envelope)
inline void needs_hand_shape(const klg::host::EnvH& e, int max_points, const char* what) {
	if (e.fits_hand_kernel(max_points)) return;
	std::fprintf(stderr, "klang-mi355: %s holds %d points%s, but the hand-written kernel this Note type is bound to (KLANG_GPU_BIND) keeps %d Time-mode points: remove the binding — the recorded form takes any envelope\n",
		what, e.npoints, e.rate_mode ? " in Rate mode" : "", max_points);
	std::abort();
}
inline void adsr_pack(const ADSR& a, klg::AdsrRec& r) { needs_hand_shape(a.h, 3, "the ADSR"); env_to_rec(a.h, r.r_out, r.r_target, r.r_rate, r.time); r.A = a.a.A; r.AD = a.h.px[2]; r.S = a.a.S; r.R = a.a.R; }

template<class NOTE> struct SawLpfAdsr {            // members: osc (Fast::Saw), lpf (Biquad::LPF), adsr (ADSR)
	static void pack(const NOTE& n, uint32_t* w) {
		klg::rec::Sub2a r;
		n.osc.h.pack(r.osc); n.lpf.h.pack(r.lpf); adsr_pack(n.adsr, r.adsr);
		r.flags = (n.adsr.h.bits() << 2) | ((uint32_t)n.osc.h.state << 8);
		std::memcpy(w, &r, sizeof r);
	}
	static void unpack(NOTE& n, const uint32_t* w) {
		klg::rec::Sub2a r; std::memcpy(&r, w, sizeof r);
		n.osc.h.offset = r.osc.offset; n.osc.h.state = (int)((r.flags >> 8) & 3u);
		n.lpf.h.z0 = r.lpf.z0; n.lpf.h.z1 = r.lpf.z1;
		env_from_rec(n.adsr.h, r.adsr.r_out, r.adsr.r_target, r.adsr.r_rate, r.adsr.time, (r.flags >> 2) & 0x3Fu);
	}
};

template<class NOTE> struct SubtractiveK {          // members: osc (Fast::Square), adsr (ADSR), env (Envelope), filter (Biquad::LPF)
	static void pack(const NOTE& n, uint32_t* w) {
		klg::rec::Sub2b r;
		n.osc.h.pack(r.osc); adsr_pack(n.adsr, r.adsr);
		needs_hand_shape(n.env.h, 3, "the filter Envelope");
		env_to_rec(n.env.h, r.env.r_out, r.env.r_target, r.env.r_rate, r.env.time);
		for (int k = 0; k < 3; k++) { r.env.px[k] = n.env.h.px[k]; r.env.py[k] = n.env.h.py[k]; }
		r.filter.f = n.filter.h.f; r.filter.Q = n.filter.h.Q; n.filter.h.pack(r.filter.c);
		r.flags = (n.adsr.h.bits() << 2) | (n.env.h.bits() << 8) | ((uint32_t)n.osc.h.state << 14);
		std::memcpy(w, &r, sizeof r);
	}
	static void unpack(NOTE& n, const uint32_t* w) {
		klg::rec::Sub2b r; std::memcpy(&r, w, sizeof r);
		n.osc.h.offset = r.osc.offset; n.osc.h.state = (int)((r.flags >> 14) & 3u);
		env_from_rec(n.adsr.h, r.adsr.r_out, r.adsr.r_target, r.adsr.r_rate, r.adsr.time, (r.flags >> 2) & 0x3Fu);
		env_from_rec(n.env.h, r.env.r_out, r.env.r_target, r.env.r_rate, r.env.time, (r.flags >> 8) & 0x3Fu);
		klg::host::BiquadLpfH& f = n.filter.h;
		f.f = r.filter.f; f.Q = r.filter.Q; f.b0 = r.filter.c.b0; f.b1 = r.filter.c.b1; f.b2 = r.filter.c.b2; f.a1 = r.filter.c.a1; f.a2 = r.filter.c.a2; f.z0 = r.filter.c.z0; f.z1 = r.filter.c.z1;
	}
};

template<class NOTE> struct SuperSawK {             // members: osc[7] (Fast::Saw), adsr (ADSR)
	static void pack(const NOTE& n, uint32_t* w) {
		klg::rec::SuperSaw r; uint32_t flags = n.adsr.h.bits() << 2;
		for (int k = 0; k < 7; k++) { n.osc[k].h.pack(r.osc[k]); flags |= (uint32_t)n.osc[k].h.state << (8 + 2 * k); }
		adsr_pack(n.adsr, r.adsr); r.flags = flags;
		std::memcpy(w, &r, sizeof r);
	}
	static void unpack(NOTE& n, const uint32_t* w) {
		klg::rec::SuperSaw r; std::memcpy(&r, w, sizeof r);
		for (int k = 0; k < 7; k++) { n.osc[k].h.offset = r.osc[k].offset; n.osc[k].h.state = (int)((r.flags >> (8 + 2 * k)) & 3u); }
		env_from_rec(n.adsr.h, r.adsr.r_out, r.adsr.r_target, r.adsr.r_rate, r.adsr.time, (r.flags >> 2) & 0x3Fu);
	}
};

template<class NOTE> struct FastSineK {             // member: osc (Fast::Sine)
	static void pack(const NOTE& n, uint32_t* w) { klg::rec::Sine r; r.flags = 0; r.inc = n.osc.h.inc; r.pos = n.osc.h.pos; std::memcpy(w, &r, sizeof r); }
	static void unpack(NOTE& n, const uint32_t* w) { klg::rec::Sine r; std::memcpy(&r, w, sizeof r); n.osc.h.pos = r.pos; }
};

} }  // namespace klang::gpu
