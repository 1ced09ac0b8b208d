// include/klang/host_dsl.hpp — HOST side of the drop-in: the parts of klang's DSL that stay on the CPU
// (north_star: "host-side C++ keeps the DSL, voice allocation and event dispatch").
//
// These are the set()/initialise()/release() halves of the reference's primitives — the code a patch's
// on()/off() runs when a note event arrives — producing the packed lane records the GPU kernels consume.
// The per-sample process() halves exist ONLY as device code (klang_amd/csrc/klg_device.hpp); there is no CPU rendering
// path in this library.  Citations are file:line into the reference's klang.h (v0.7.8).
#pragma once
#include <cmath>
#include <vector>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../klang_mi355_records.h"

// Build every translation unit that includes this with -ffp-contract=off (bit parity with the reference).
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace klg { namespace host {

constexpr float PI_F = 3.14159274101257324f;
constexpr float TWO_PI = 2.f * PI_F;
constexpr float ROOT2_INV = 0.707106769084930420f;      // root2.inv = (float)(1.0 / 1.41421356...)

struct Fs {                                              // struct SampleRate klang.h:1593-1604
	float f, inv, w;
	explicit Fs(float sr = 44100.f) : f(sr), inv(1.f / sr), w(2.0f * PI_F * inv) {}
};

inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// Pitch::operator-> klang.h:1568-1571 with power(float, float) 191-217
inline float pitch_to_frequency(float pitch) {
	const float e = (pitch - 69.f) / 12.f;
	float p;
	if (e == 0.f) p = 1.f;
	else if (e == 1.f) p = 2.f;
	else if (e == 2.f) p = 2.f * 2.f;
	else if (e == 3.f) p = 2.f * 2.f * 2.f;
	else if (e == 4.f) p = 2.f * 2.f * 2.f * 2.f;
	else if (e == -1.f) p = 1.f / 2.f;
	else if (e == -2.f) p = 1.f / (2.f * 2.f);
	else if (e == -3.f) p = 1.f / (2.f * 2.f * 2.f);
	else if (e == -4.f) p = 1.f / (2.f * 2.f * 2.f * 2.f);
	else p = std::pow(2.f, e);
	return 440.f * p;
}

// klang::random<double>(min, max) klang.h:236
inline double random_d(double mn, double mx) { return std::rand() * ((mx - mn) / (double)RAND_MAX) + mn; }

// ---- Generators::Fast::Increment / Phase (klang.h:4960-5007) ----
inline int32_t fast_increment(float f, const Fs& fs) {
	constexpr float FC4 = float(261.62556530059862);
	constexpr float FC4_FINTMAX = float(261.62556530059862 * 2147483648.0);
	const float FBASE = FC4_FINTMAX / fs.f;
	return (int32_t)(2u * (uint32_t)(int32_t)(FBASE / FC4 * f));
}
inline float fast_increment_float(int32_t amount) { return u2f((uint32_t)((amount >> 9) | 0x3f800000)) - 1.f; }
inline uint32_t fast_phase(float radians) { return (uint32_t)(int64_t)(radians * 2147483648.0f / (2.f * PI_F)); }
inline float fast_phase_float(uint32_t pos) { return u2f((pos >> 9) | 0x3f800000u) - 1.f; }

// ---- Fast::Sine set() side (klang.h:5142-5153); frequency cache reproduces the `if (frequency != cached)` guard ----
struct FSineH {
	float frequency = 1000.f; int32_t inc = 0; uint32_t pos = 0;
	void set(float f, float phase, const Fs& fs) {
		pos = fast_phase(phase);
		if (f != frequency) { frequency = f; inc = fast_increment(f, fs); }
	}
};

// Fast::Sine::process on the host (klang.h:5093-5132, 5165-5171) — only to render a cycle into a Wavetable (klang.h:3646-3651);
// audio-rate processing is the device's
inline float fastsinp_host(uint32_t p) {
	float x = (u2f((p >> 9) | 0x3f800000u) - 1.f) * (2.f * PI_F);
	if (x > 4.71238899230957031f) x -= 2.f * PI_F;                 // 3.f / 2.f * pi.f
	else if (x > 1.57079637050628662f) x = PI_F - x;              // pi.f / 2.f
	const float x2 = x * x;
	return (((-0.00018542f * x2 + 0.0083143f) * x2 - 0.16666f) * x2 + 1.0f) * x;
}

// ---- Generic::Oscillator set() side (klang.h:2862-2870) ----
struct BOscH {
	float frequency = 1000.f, increment = 0.f, position = 0.f, offset = 0.f;
	void set(float f, float phase, const Fs& fs) { position = phase; frequency = f; increment = f * 2.f * PI_F / fs.f; }
};

// ---- Fast::OSM set() side (klang.h:5206-5249) ----
struct OsmH {
	int32_t inc = 0; uint32_t offset = 0, duty = 0; int state = 0; float delta = 0.f, frequency = 0.f;
	explicit OsmH(float duty_ = 0.f) { set_duty(duty_); }                      // Osm ctor klang.h:5323
	void init() { state = ((uint32_t)(offset - (uint32_t)inc) < duty) ? 3 : 0; }  // the coefficient half of init() is re-derived on the GPU
	void refresh(float f, const Fs& fs) { if (frequency != f) { frequency = f; inc = fast_increment(f, fs); delta = fast_increment_float(inc); } }
	void set_duty(float d) { duty = fast_phase(d * (2.f * PI_F)); init(); }
	void set(float f, float phase, const Fs& fs) { refresh(f, fs); offset = fast_phase(phase); init(); }
	void set(float f, float phase, float d, const Fs& fs) { refresh(f, fs); offset = fast_phase(phase); set_duty(d); }
	void pack(OsmRec& r) const { r.inc = inc; r.offset = offset; r.duty = duty; r.delta = delta; }
};

// ---- Filters::Biquad::LPF set()/reset() side (klang.h:5565-5600, 5658-5665); libm: this host's glibc cosf/sinf ----
enum { BQ_LPF = 0, BQ_HPF, BQ_BPF_PEAK, BQ_BPF_SKIRT, BQ_BRF, BQ_APF, BQ_BUTTER2 };
struct BiquadLpfH {                                      // (name kept: the LPF was first) — every Filters::Biquad type by `type`
	int type = BQ_LPF;
	float f = 0, Q = 0, a1 = 0, a2 = 0, b0 = 1, b1 = 0, b2 = 0, a = 0, cos0 = 1, sin0 = 0, z0 = 0, z1 = 0;
	void reset() { f = 0; Q = 0; b0 = 1; a1 = a2 = b1 = b2 = 0; a = 0; z0 = z1 = 0; }
	void init(const Fs& fs) {                            // the per-type init() klang.h:5658-5665, 5675-5682, 5708-5729, 5734-5739, 5765-5772, 5801-5811
		if (type == BQ_APF) {
			const float omega = 2.0f * PI_F * f / fs.f;
			const float c = (float)cos((double)omega);
			b0 = a2 = a * a; b1 = a1 = (-2.f * a * c); b2 = 1.f;
			return;
		}
		const double a0 = (double)(1.f + a);
		const float inv = (a0 == 0.0f) ? 0.0f : (float)(1.0 / a0);
		a1 = inv * (-2.f * cos0);
		a2 = inv * (1.f - a);
		switch (type) {
		case BQ_LPF: b2 = b0 = inv * (1.f - cos0) * 0.5f; b1 = inv * (1.f - cos0); break;
		case BQ_HPF: b2 = b0 = inv * (1.f + cos0) * 0.5f; b1 = inv * -(1.f + cos0); break;
		case BQ_BPF_PEAK: b0 = inv * a; b1 = 0; b2 = inv * -a; break;
		case BQ_BPF_SKIRT: b0 = inv * sin0 * 0.5f; b1 = 0; b2 = -b0; break;
		case BQ_BRF: b1 = a1; b0 = b2 = inv; break;
		case BQ_BUTTER2: b0 = inv * ((1.f - cos0) / 2.f); b1 = inv * (1.f - cos0); b2 = inv * ((1.f - cos0) / 2.f); break;
		}
	}
	void set(float f_, float Q_, const Fs& fs) {
		if (type == BQ_APF) {                               // APF::set(f, r) klang.h:5752-5763
			if (f != f_ || a != Q_) { f = f_; a = Q_; const float w = f_ * fs.w; cos0 = cosf(w); sin0 = sinf(w); init(fs); }
			return;
		}
		if (Q_ < 0) Q_ = f_ / -Q_;
		if (f != f_ || Q != Q_) {
			f = f_; Q = Q_;
			const float w = f_ * fs.w;
			cos0 = cosf(w); sin0 = sinf(w);
			if (Q_ < 0.5) Q_ = 0.5f;
			a = sin0 / (2.f * Q_);
			init(fs);
		}
	}
	void pack(BiquadRec& r) const { r.b0 = b0; r.b1 = b1; r.b2 = b2; r.a1 = a1; r.a2 = a2; r.z0 = z0; r.z1 = z1; }
};

// ---- Filters::OnePole::LPF / HPF set() side (klang.h:5489-5494, 5510-5514, 5537-5542) ----
struct OnePoleH {
	bool hpf = false; float f = 0, a1 = 0, b0 = 1, b1 = 0, z = 0;
	void reset() { a1 = 0; b0 = 1; b1 = 0; f = 0; z = 0; }
	void set(float f_, const Fs& fs) {
		if (f != f_) {
			f = f_;
			const float e = expf(-f * fs.w);
			if (hpf) { b0 = 0.5f * (1.f + e); b1 = -b0; a1 = e; } else { b0 = 1 - e; a1 = e; }
		}
	}
};

// ---- row f2 set() sides: Butterworth::LPF<1>/<2> (klang.h:5786-5811), Modal (5822-5846), Envelope::Follower::AR (5872-5879) ----
inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (hi < x ? hi : x); }
struct Butter1H {
	float f = 0, a1 = 0, b0 = 1, z = 0, out = 0;
	void set(float f_, const Fs& fs) {
		if (f != f_) {
			f = f_;
			const float c = 1.f / tanf(PI_F * f * fs.inv);
			const double a0 = (double)(1.f + c);
			const float inv = (a0 == 0.0f) ? 0.0f : (float)(1.0 / a0);
			b0 = inv; a1 = (1.f - c) * inv;
		}
	}
};
inline void butter2_design(float f, const Fs& fs, float c[5]) {             // Biquad::Filter::set(f) with Butterworth::LPF<2>::init
	const float Q = ROOT2_INV, w = f * fs.w, cos0 = cosf(w), sin0 = sinf(w);
	const float a = sin0 / (2.f * Q);
	const double a0 = (double)(1.f + a);
	const float inv = (a0 == 0.0f) ? 0.0f : (float)(1.0 / a0);
	c[0] = inv * ((1.f - cos0) / 2.f); c[1] = inv * (1.f - cos0); c[2] = inv * ((1.f - cos0) / 2.f); c[3] = inv * (-2.f * cos0); c[4] = inv * (1.f - a);
}
struct ModalH {
	float a1 = 0, a2 = 0, y1 = 0, y2 = 0, gain = 0.05f;
	void set(float f, float decay, const Fs& fs) {
		gain = 0.05f;
		const float w = f * fs.w;
		const float d = clampf(expf(-PI_F / (decay * fs.f)), 1e-6f, 0.9999f);
		a1 = 2.f * d * clampf(cosf(w), -0.9999f, 0.9999f);
		a2 = -d * d;
		y2 = 0; y1 = 0;
	}
	void set(float f, float decay, float g, const Fs& fs) { set(f, decay, fs); gain = clampf(g * 0.05f, -0.05f, 0.05f); }
};
struct FollowerArH {
	float attack = 0, release = 0, A = 1, R = 1;
	void set(float attack_, float release_, const Fs& fs) {
		if (attack != attack_ || release != release_) {
			attack = attack_; release = release_;
			A = 1.f - (attack == 0.f ? 0.f : expf(-1.0f / (fs.f * attack)));
			R = 1.f - (release == 0.f ? 0.f : expf(-1.0f / (fs.f * release)));
		}
	}
};

// ---- Envelope set()/initialise() side (klang.h:3893-3989, 4064-4092) ----
// Any number of points (the reference keeps a std::vector<Point>, klang.h:4093); Time or Rate mode (setMode 4064-4071).
struct EnvH {
	float r_out = 1.f, r_target = 1.f, r_rate = 0.f; bool active = false;
	int npoints = 0; std::vector<float> px = std::vector<float>(4, 0.f), py = std::vector<float>(4, 0.f);   // (at least four slots, zeros behind the points: what the four register slots of a lane read)
	int point = 0; float time = 0.f; int stage = ENV_SUSTAIN;
	int loop_start = -1, loop_end = -1;                                        // Envelope::Loop klang.h:3853-3864
	bool rate_mode = false;                                                    // setMode(Rate): a point's x is the ramp's step per sample, not a time (setTargetRate 4083-4092)
	void set_loop(int a, int b) { if (a >= 0 && b < npoints) { loop_start = a; loop_end = b; } }   // setLoop klang.h:3923-3926
	void set_value(float v) { r_out = v; r_target = v; active = false; }
	void set_target(float x, float y, float t, const Fs& fs) {
		if (!rate_mode) {                                                      // setTargetTime 4077-4081
			time = t; r_target = y; active = (r_out != y);
			r_rate = std::fabs(y - r_out) / ((x - t) * fs.f);
		}
		else {                                                                 // setTargetRate 4083-4092
			time = 0.f;
			if (x == 0.f) set_value(y);
			else { r_target = y; active = (r_out != y); r_rate = x; }
		}
	}
	void initialise(const Fs& fs) {                                            // klang.h:3974-3989
		point = 0; stage = ENV_SUSTAIN; loop_start = loop_end = -1;            // (initialise() resets the loop, klang.h:3977)
		if (npoints) { set_value(py[0]); if (npoints > 1) set_target(px[1], py[1], px[0], fs); }
		else set_value(1.f);
	}
	void set_points(int n, const float* xy, const Fs& fs) {
		npoints = n;
		px.assign((size_t)(n > 4 ? n : 4), 0.f); py.assign((size_t)(n > 4 ? n : 4), 0.f);
		for (int i = 0; i < n; i++) { px[(size_t)i] = xy[2 * i]; py[(size_t)i] = xy[2 * i + 1]; }
		initialise(fs);
	}
	// stage(2) | point(3) | active(1): the six flag bits of the hand-written kernels' records (three-point Time-mode envelopes; fits_hand_kernel() below)
	uint32_t bits() const { return (uint32_t)stage | ((uint32_t)point << 2) | ((uint32_t)active << 5); }
	bool fits_hand_kernel(int max_points) const { return !rate_mode && npoints <= max_points && point < 8; }
};

// ---- ADSR::set (klang.h:4116-4129) ----
struct AdsrH {
	EnvH env; float A = 0, D = 0, S = 0, R = 0;
	void set(float attack, float decay, float sustain, float release, const Fs& fs) {
		A = attack; D = decay + 0.005f; S = sustain; R = release + 0.005f;
		const float xy[6] = { 0.f, 0.f, A, 1.f, A + D, S };
		env.set_points(3, xy, fs);
		env.set_loop(2, 2);
	}
	void pack(AdsrRec& r) const { r.r_out = env.r_out; r.r_target = env.r_target; r.r_rate = env.r_rate; r.time = env.time; r.A = A; r.AD = env.px[2]; r.S = S; r.R = R; }
};

// Control::set (klang.h:1725-1728) / Dial() (1797-1800)
struct ControlH { float min, max, value; void set(float x) { value = (x < min) ? min : (max < x) ? max : x; } };

} } // namespace klg::host
