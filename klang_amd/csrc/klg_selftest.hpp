// klang_amd/csrc/klg_selftest.hpp — klg_selftest(): runs ONE device primitive of klg_device.hpp / klg_fx.hpp for n
// samples on the GPU and returns its output, so every row of the hot-path table (a7-a22) can be
// checked against the reference's known-answer vectors individually (tests/test_gpu_primitives.py), not only
// through the patch kernels.  Host halves (set(): increments, coefficients, breakpoints) come from klg_host_dsl.hpp
// exactly as they do for the patches; the per-sample halves run in a single lane.
#pragma once

namespace klg {

enum {
	ST_BASIC_SINE = 0, ST_BASIC_SAW, ST_BASIC_TRIANGLE, ST_BASIC_SQUARE, ST_BASIC_PULSE,   // p: increment, position, offset, duty
	ST_FAST_SINE,            // p: inc(bits), pos(bits) ; in (optional): relative phase offset per sample (Operator-style PM)
	ST_OSM_SAW, ST_OSM_PULSE,   // p: inc, offset, duty, state, delta (bits / float)
	ST_ONEPOLE_LPF, ST_ONEPOLE_HPF,   // p: b0, b1, a1 ; in: signal
	ST_BIQUAD,               // p: b0 b1 b2 a1 a2 ; in: signal
	ST_BIQUAD_LPF_SWEEP,     // p: Q, fs.w ; in: signal ; in2 (second half of `in`): cutoff per sample
	ST_ADSR,                 // p: r_out r_target r_rate time A AD S R bits release_at rel_time rel_level fs ; out[n] values, out[n..2n) stage
	ST_ENV3,                 // p: r_out r_target r_rate time bits npoints x0 x1 x2 y0 y1 y2 fs ; out values + stage
	ST_OPERATOR3,            // p: 3 x (inc pos r_out r_target r_rate time bits np x0 x1 y0 y1 amp) fs
	ST_DELAY,                // p: size, mode(0 set+process, 1 tap int, 2 tap float, 3 lagrange), arg ; in: signal (+ per-sample set() argument in second half for mode 4)
	ST_STEREO_DELAY_TAP,     // p: size, delay ; in: L then R ; out interleaved l r
	ST_MATRIX,               // in: 4 per row ; out 4 per row
	ST_CONTROL_SMOOTH,       // p: value ; out smoothed
	ST_NOISE_BASIC, ST_NOISE_FAST,  // in: rand() results as float-encoded ints (bit pattern)
	// SURVEY §8 row f2
	ST_DCF,                  // p: r ; in: signal
	ST_IIR2, ST_IIR4,        // p: a[ORDER] ; in: signal
	ST_IIR1,                 // p: a, b ; in: signal
	ST_BUTTER1,              // p: b0, a1 ; in: signal
	ST_MODAL,                // p: a1, a2, gain ; in: signal
	ST_FOLLOWER_AR, ST_FOLLOWER_PEAK, ST_FOLLOWER_RMS,  // p: A, R ; in: signal
	// the run-time envelope of recorded graph patches — any number of points, loop, Time / Rate mode (env_process_rt over PtsN)
	ST_ENVN                  // p: r_out r_target r_rate time bits(graph record form) npoints loop_start loop_end release_at rel_time rel_level fs capacity ;
	                         // in: the record's point words — x0..x3 y0..y3, then x of points 4.. and their y (capacity - 4 each) ; out values + stage
};

struct SelfTestArgs { int prim, n, n_in; const float* p; const float* in; float* out; float* scratch; };

__global__ void klg_selftest_kernel(const SelfTestArgs a) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	const float* p = a.p; const float* in = a.in; float* out = a.out; const int n = a.n;
	auto bits = [&](int i) { return __float_as_uint(p[i]); };
	switch (a.prim) {
	case ST_BASIC_SINE: case ST_BASIC_SAW: case ST_BASIC_TRIANGLE: case ST_BASIC_SQUARE: case ST_BASIC_PULSE: {
		BOsc o; o.increment = p[0]; o.position = p[1]; o.offset = p[2];
		for (int i = 0; i < n; i++)
			out[i] = a.prim == ST_BASIC_SINE ? basic_sine(o) : a.prim == ST_BASIC_SAW ? basic_saw(o) : a.prim == ST_BASIC_TRIANGLE ? basic_triangle(o)
			       : a.prim == ST_BASIC_SQUARE ? basic_square(o) : basic_pulse(o, p[3]);
	} break;
	case ST_FAST_SINE: {
		FSine o; o.inc = (int32_t)bits(0); o.pos = bits(1);
		for (int i = 0; i < n; i++) out[i] = fsine_process(o, a.n_in ? fsine_rel_offset(in[i]) : 0u);
	} break;
	case ST_OSM_SAW: case ST_OSM_PULSE: {
		Osm o; o.inc = (int32_t)bits(0); o.offset = bits(1); o.duty = bits(2); o.state = (int)bits(3); o.delta = p[4];
		osm_derive(o);
		for (int i = 0; i < n; i++) out[i] = a.prim == ST_OSM_SAW ? osm_saw(o) : osm_pulse(o);
	} break;
	case ST_ONEPOLE_LPF: case ST_ONEPOLE_HPF: {
		OnePole q; q.b0 = p[0]; q.b1 = p[1]; q.a1 = p[2]; q.z = 0.f; q.out = 0.f;
		for (int i = 0; i < n; i++) out[i] = a.prim == ST_ONEPOLE_LPF ? onepole_lpf_process(q, in[i]) : onepole_process(q, in[i]);
	} break;
	case ST_BIQUAD: {
		Biquad q = { p[0], p[1], p[2], p[3], p[4], 0.f, 0.f };
		for (int i = 0; i < n; i++) out[i] = biquad_process(q, in[i]);
	} break;
	case ST_BIQUAD_LPF_SWEEP: {
		Biquad q = { 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }; BiquadSweep c = { 0.f, 0.f };
		for (int i = 0; i < n; i++) { biquad_lpf_set(q, c, in[n + i], p[0], p[1]); out[i] = biquad_process(q, in[i]); }
	} break;
	case ST_ADSR: {
		Env e; e.r_out = p[0]; e.r_target = p[1]; e.r_rate = p[2]; e.time = p[3]; env_unpack(e, bits(8));
		Pts3 pt; pt.x0 = 0.f; pt.x1 = p[4]; pt.x2 = p[5]; pt.y0 = 0.f; pt.y1 = 1.f; pt.y2 = p[6];
		const int release_at = (int)p[9]; SampleRate fs; fs.f = p[12]; fs.timeInc = 1.0f / fs.f; fs.w = 0.f;
		for (int i = 0; i < n; i++) {
			if (i == release_at) env_release(e, p[10] ? p[10] : p[7], p[11], fs.f);      // ADSR::release klang.h:4131-4133
			out[i] = env_process<3, true>(e, pt, 3, fs); out[n + i] = (float)e.stage;
		}
	} break;
	case ST_ENV3: {
		Env e; e.r_out = p[0]; e.r_target = p[1]; e.r_rate = p[2]; e.time = p[3]; env_unpack(e, bits(4));
		Pts3 pt; pt.x0 = p[6]; pt.x1 = p[7]; pt.x2 = p[8]; pt.y0 = p[9]; pt.y1 = p[10]; pt.y2 = p[11];
		SampleRate fs; fs.f = p[12]; fs.timeInc = 1.0f / fs.f; fs.w = 0.f;
		for (int i = 0; i < n; i++) { out[i] = env_process<3, false>(e, pt, (int)p[5], fs); out[n + i] = (float)e.stage; }
	} break;
	case ST_ENVN: {
		Env e; int npm; e.r_out = p[0]; e.r_target = p[1]; e.r_rate = p[2]; e.time = p[3]; env_unpack_rt(e, npm, bits(4), (uint32_t)p[5]);
		const int ls = (int)p[6], le = (int)p[7], release_at = (int)p[8], cap = (int)p[12];
		SampleRate fs; fs.f = p[11]; fs.timeInc = 1.0f / fs.f; fs.w = 0.f;
		PtsN pt; pt.head.x0 = in[0]; pt.head.x1 = in[1]; pt.head.x2 = in[2]; pt.head.x3 = in[3]; pt.head.y0 = in[4]; pt.head.y1 = in[5]; pt.head.y2 = in[6]; pt.head.y3 = in[7];
		pt.ext = reinterpret_cast<const uint32_t*>(in + 8); pt.stride = 1; pt.slots = cap - 4;
		const float hy = env_hold_y(pt, ls);
		for (int i = 0; i < n; i++) {
			if (i == release_at) env_release_rt(e, p[9], p[10], fs.f, (npm & ENV_NPM_RATE) != 0);      // Envelope::release klang.h:3961-3966
			out[i] = env_process_rt(e, pt, npm, ls, le, hy, fs); out[n + i] = (float)e.stage;
		}
	} break;
	case ST_OPERATOR3: {
		FSine osc[3]; Env env[3]; Pts2 pt[3]; int np[3]; float amp[3];
		for (int k = 0; k < 3; k++) {
			const float* q = p + 13 * k;
			osc[k].inc = (int32_t)__float_as_uint(q[0]); osc[k].pos = __float_as_uint(q[1]);
			env[k].r_out = q[2]; env[k].r_target = q[3]; env[k].r_rate = q[4]; env[k].time = q[5]; env_unpack(env[k], __float_as_uint(q[6]));
			np[k] = (int)q[7]; pt[k].x0 = q[8]; pt[k].x1 = q[9]; pt[k].y0 = q[10]; pt[k].y1 = q[11]; amp[k] = q[12];
		}
		SampleRate fs; fs.f = p[39]; fs.timeInc = 1.0f / fs.f; fs.w = 0.f;
		for (int i = 0; i < n; i++) {
			float m = 0.f;
			for (int k = 0; k < 3; k++) {                                             // Operator::process klang.h:4164-4168
				float y = fsine_process(osc[k], fsine_rel_offset(m));
				y *= env_process<2, false>(env[k], pt[k], np[k], fs) * amp[k];
				m = y;
			}
			out[i] = m;
		}
	} break;
	case ST_DELAY: {
		const int size = (int)p[0], mode = (int)p[1];
		Ring r = { a.scratch, 1, size };
		for (int i = 0; i < size; i++) r.wr(i, 0.f);
		int position = 0; Tap last = { 0, 0.f };
		if (mode == 0) last = delay_set(position, size, p[2]);
		for (int i = 0; i < n; i++) {
			if (mode == 4) last = delay_set(position, size, in[n + i]);               // set() every sample, then input, then process
			r.wr(position, in[i]); position = (position + 1 == size) ? 0 : position + 1;   // Delay::input 3396-3403
			if (mode == 0 || mode == 4) out[i] = delay_process(r, last);
			else if (mode == 1) out[i] = delay_tap_int(r, position, (int)p[2]);
			else if (mode == 2) out[i] = delay_tap_float(r, position, p[2]);
			else out[i] = delay_lagrange(r, position, p[2]);
		}
	} break;
	case ST_STEREO_DELAY_TAP: {
		const int size = (int)p[0];
		Ring l = { a.scratch, 1, size }, r = { a.scratch + size, 1, size };
		for (int i = 0; i < size; i++) { l.wr(i, 0.f); r.wr(i, 0.f); }
		int position = 0;
		for (int i = 0; i < n; i++) {
			l.wr(position, in[i]); r.wr(position, in[n + i]); position = (position + 1 == size) ? 0 : position + 1;
			stereo_delay_tap(l, r, position, p[1], out[2 * i], out[2 * i + 1]);
		}
	} break;
	case ST_MATRIX:
		for (int i = 0; i < n; i++) {
			const float* d = in + 4 * i; float* o = out + 4 * i;                        // signals<4> >> Matrix, FDN matrix of Reverb.k:158-161
			o[0] = 0.f * d[0] + 1.f * d[1] + 1.f * d[2] + -1.f * d[3];
			o[1] = -1.f * d[0] + 0.f * d[1] + -1.f * d[2] + 1.f * d[3];
			o[2] = -1.f * d[0] + 1.f * d[1] + 0.f * d[2] + -1.f * d[3];
			o[3] = 1.f * d[0] + -1.f * d[1] + 1.f * d[2] + 0.f * d[3];
		}
		break;
	case ST_CONTROL_SMOOTH: {
		float sm = 0.f;
		for (int i = 0; i < n; i++) { sm = sm * 0.999f + (1.f - 0.999f) * p[0]; out[i] = sm; }   // Control::smooth klang.h:1715
	} break;
	case ST_NOISE_BASIC: for (int i = 0; i < n; i++) out[i] = basic_noise((int)__float_as_uint(in[i])); break;
	case ST_NOISE_FAST: for (int i = 0; i < n; i++) out[i] = fast_noise((int)__float_as_uint(in[i])); break;
	case ST_DCF: { Dcf q = { p[0], 0.f, 0.f }; for (int i = 0; i < n; i++) out[i] = dcf_process(q, in[i]); } break;
	case ST_IIR2: { Iir<2> q = { { p[0], p[1] }, { 0.f, 0.f } }; for (int i = 0; i < n; i++) out[i] = iir_process(q, in[i]); } break;
	case ST_IIR4: { Iir<4> q = { { p[0], p[1], p[2], p[3] }, { 0.f, 0.f, 0.f, 0.f } }; for (int i = 0; i < n; i++) out[i] = iir_process(q, in[i]); } break;
	case ST_IIR1: { Iir1 q = { p[0], p[1], 0.f }; for (int i = 0; i < n; i++) out[i] = iir1_process(q, in[i]); } break;
	case ST_BUTTER1: { Butter1 q = { p[0], p[1], 0.f, 0.f }; for (int i = 0; i < n; i++) out[i] = butter1_process(q, in[i]); } break;
	case ST_MODAL: { Modal q = { p[0], p[1], 0.f, 0.f, p[2] }; for (int i = 0; i < n; i++) out[i] = modal_process(q, in[i]); } break;
	case ST_FOLLOWER_AR: case ST_FOLLOWER_PEAK: case ST_FOLLOWER_RMS: {
		FollowerAR q = { p[0], p[1], 0.f };
		for (int i = 0; i < n; i++) out[i] = a.prim == ST_FOLLOWER_AR ? follower_ar_process(q, in[i]) : a.prim == ST_FOLLOWER_PEAK ? follower_peak(q, in[i]) : follower_rms(q, in[i]);
	} break;
	}
}

} // namespace klg

extern "C" int klg_selftest(int prim, const float* params, int n_params, const float* in, int n_in, float* out, int n_out, int n) {
	if (!params || !out || n <= 0 || n_out < n) return fail(KLG_ERR_INVALID, "klg_selftest: bad arguments");
	if (klg_ensure_device()) return KLG_ERR_NO_DEVICE;
	float *d_p = nullptr, *d_in = nullptr, *d_out = nullptr, *d_scratch = nullptr;
	const size_t scratch = 1 << 20;
	struct Free { float **a, **b, **c, **d; ~Free() { (void)hipFree(*a); (void)hipFree(*b); (void)hipFree(*c); (void)hipFree(*d); } } release = { &d_p, &d_in, &d_out, &d_scratch };   // also on the error returns of HIP_TRY
	HIP_TRY(hipMalloc(&d_p, (size_t)std::max(n_params, 1) * 4)); HIP_TRY(hipMalloc(&d_in, (size_t)std::max(n_in, 1) * 4));
	HIP_TRY(hipMalloc(&d_out, (size_t)n_out * 4)); HIP_TRY(hipMalloc(&d_scratch, scratch * 4));
	HIP_TRY(hipMemcpy(d_p, params, (size_t)n_params * 4, hipMemcpyHostToDevice));
	if (n_in) HIP_TRY(hipMemcpy(d_in, in, (size_t)n_in * 4, hipMemcpyHostToDevice));
	HIP_TRY(hipMemset(d_out, 0, (size_t)n_out * 4));
	SelfTestArgs a = { prim, n, n_in, d_p, d_in, d_out, d_scratch };
	hipLaunchKernelGGL(klg::klg_selftest_kernel, dim3(1), dim3(64), 0, 0, a);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpy(out, d_out, (size_t)n_out * 4, hipMemcpyDeviceToHost));
	return 0;
}

// Host halves for the self-test (the same klg_host_dsl.hpp code the patches' on() uses), exported so the test can
// obtain increments / coefficients / breakpoint state without re-deriving them in Python.
// kind: 0 Generic::Oscillator::set(f, phase) -> {increment, position}
//       1 Fast::Sine::set(f, phase)          -> {inc bits, pos bits}
//       2 OSM set(f, phase, duty) with ctor duty args[3] (args[4] != 0: duty given) -> {inc, offset, duty, state, delta}
//       3 OnePole LPF(f) / 4 OnePole HPF(f)  -> {b0, b1, a1}
//       5 Biquad design: args = {type(0 LPF,1 HPF,2 BPF peak,3 BPF skirt,4 BRF,5 APF), f, Q} -> {b0 b1 b2 a1 a2}
//       6 ADSR::set(a, d, s, r)              -> {r_out r_target r_rate time A AD S R bits}
//       7 Envelope::set(points) (n <= 3: args = {n, x0, y0, ...}) -> {r_out r_target r_rate time bits}
//       8 pitch -> Frequency
//       9 Butterworth::LPF<1>::set(f) -> {b0, a1}     10 Butterworth::LPF<2>::set(f) -> {b0 b1 b2 a1 a2}
//      11 Modal::set(f, decay[, gain (args[2] != 0)]) -> {a1, a2, gain}     12 Envelope::Follower::AR::set(attack, release) -> {A, R}
extern "C" int klg_selftest_host(int kind, const float* args, int n_args, float sample_rate, float* out, int n_out) {
	if (!args || !out) return fail(KLG_ERR_INVALID, "klg_selftest_host: bad arguments");
	const host::Fs fs(sample_rate);
	auto f2b = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
	switch (kind) {
	case 0: { host::BOscH o; o.set(args[0], args[1], fs); out[0] = o.increment; out[1] = o.position; } return 0;
	case 1: { host::FSineH o; o.set(args[0], args[1], fs); out[0] = f2b((uint32_t)o.inc); out[1] = f2b(o.pos); } return 0;
	case 2: {
		host::OsmH o(args[3]);
		if (args[4] != 0.f) o.set(args[0], args[1], args[2], fs); else o.set(args[0], args[1], fs);
		out[0] = f2b((uint32_t)o.inc); out[1] = f2b(o.offset); out[2] = f2b(o.duty); out[3] = f2b((uint32_t)o.state); out[4] = o.delta;
	} return 0;
	case 3: case 4: {                                                              // OnePole::LPF/HPF::init klang.h:5510-5514, 5537-5542
		const float e = expf(-args[0] * fs.w);
		if (kind == 3) { out[0] = 1 - e; out[1] = 0.f; out[2] = e; }
		else { out[0] = 0.5f * (1.f + e); out[1] = -out[0]; out[2] = e; }
	} return 0;
	case 5: {
		const int type = (int)args[0]; float f = args[1], Q = args[2];
		if (type <= 1) { const BiquadCoef c = design_biquad(type == 1, f, Q, fs); out[0] = c.b0; out[1] = c.b1; out[2] = c.b2; out[3] = c.a1; out[4] = c.a2; return 0; }
		if (type == 5) {                                                            // APF::set(f, r) + init klang.h:5752-5772
			const float a = Q, omega = 2.0f * host::PI_F * f / fs.f, c0 = (float)cos((double)omega);
			out[0] = a * a; out[1] = -2.f * a * c0; out[2] = 1.f; out[3] = out[1]; out[4] = out[0]; return 0;
		}
		if (Q < 0) Q = f / -Q;
		const float w = f * fs.w, cos0 = cosf(w), sin0 = sinf(w);
		if (Q < 0.5) Q = 0.5f;
		const float a = sin0 / (2.f * Q);
		const double a0 = (double)(1.f + a);
		const float inv = (a0 == 0.0f) ? 0.0f : (float)(1.0 / a0);
		out[3] = inv * (-2.f * cos0); out[4] = inv * (1.f - a);
		if (type == 2) { out[0] = inv * a; out[1] = 0; out[2] = inv * -a; }              // BPF constant peak gain 5720-5729
		else if (type == 3) { out[0] = inv * sin0 * 0.5f; out[1] = 0; out[2] = -out[0]; }  // constant skirt gain 5708-5717
		else { out[1] = out[3]; out[0] = out[2] = inv; }                                // BRF 5734-5739
	} return 0;
	case 6: {
		host::AdsrH a; a.set(args[0], args[1], args[2], args[3], fs);
		AdsrRec r; a.pack(r);
		const float v[8] = { r.r_out, r.r_target, r.r_rate, r.time, r.A, r.AD, r.S, r.R };
		for (int i = 0; i < 8; i++) out[i] = v[i];
		out[8] = f2b(a.env.bits());
	} return 0;
	case 7: {
		host::EnvH e; e.set_points((int)args[0], args + 1, fs);
		out[0] = e.r_out; out[1] = e.r_target; out[2] = e.r_rate; out[3] = e.time; out[4] = f2b(e.bits());
	} return 0;
	case 13: {                                                                      // args: rate mode, loop start, loop end, point count, x y ... -> the state set(points) [+ setLoop] leaves
		host::EnvH e; e.rate_mode = args[0] != 0.f; e.set_points((int)args[3], args + 4, fs);
		if (args[1] >= 0.f) e.set_loop((int)args[1], (int)args[2]);
		out[0] = e.r_out; out[1] = e.r_target; out[2] = e.r_rate; out[3] = e.time; out[4] = f2b(graph::env_bits(e.stage, e.point, e.active, e.rate_mode));
	} return 0;
	case 8: out[0] = host::pitch_to_frequency(args[0]); return 0;
	case 9: { host::Butter1H q; q.set(args[0], fs); out[0] = q.b0; out[1] = q.a1; } return 0;
	case 10: host::butter2_design(args[0], fs, out); return 0;
	case 11: { host::ModalH q; if (args[2] != 0.f) q.set(args[0], args[1], args[2], fs); else q.set(args[0], args[1], fs); out[0] = q.a1; out[1] = q.a2; out[2] = q.gain; } return 0;
	case 12: { host::FollowerArH q; q.set(args[0], args[1], fs); out[0] = q.A; out[1] = q.R; } return 0;
	}
	(void)n_args; (void)n_out;
	return fail(KLG_ERR_INVALID, "klg_selftest_host: unknown kind %d", kind);
}
