// oracle/ref/ref_prims.cpp — TEST INFRASTRUCTURE (golden-vector generator).
// Known-answer vectors for every hot-path primitive of SURVEY.md §8(a), produced by
// executing the GENUINE reference header.  Output: tests/golden/prims.kat, a sequence of
// records { u32 name_len, name, u32 count, f32[count] }.
// The stimuli are restated in tests/test_oracle_prims.py (same formulas, numpy) so the C
// restatement and the HIP device primitives can be driven identically.
#include "prelude.h"
#include <klang.h>
#include "ref_common.h"
#include <string>
#include <vector>

using namespace klang;

static FILE* g_out;
static void emit(const std::string& name, const std::vector<float>& v) {
	unsigned n = (unsigned)name.size(), c = (unsigned)v.size();
	fwrite(&n, 4, 1, g_out); fwrite(name.data(), 1, n, g_out);
	fwrite(&c, 4, 1, g_out); fwrite(v.data(), 4, c, g_out);
}
static float bits2f(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static float noise(unsigned n) { return ref_fx_input(1u, 0u, 0u, n, 0xFFFFFFFFu); }  // U(-.5,.5) hash noise

template<class OSC> static std::vector<float> run_osc(OSC& o, int n) {
	std::vector<float> v(n);
	for (int i = 0; i < n; i++) { signal x; o >> x; v[i] = x; }
	return v;
}

int main(int argc, char** argv) {
	if (argc < 2) { fprintf(stderr, "usage: %s out.kat\n", argv[0]); return 1; }
	g_out = fopen(argv[1], "wb");
	klang::fs = SampleRate(48000.f);
	const int N = 1024;
	const float freqs[] = { 27.5f, 110.f, 440.f, 1000.f, 2093.0045f, 7040.f, 15000.f };
	char nm[128];

	// --- a23: Pitch -> Frequency for all MIDI pitches (host-side powf)
	{ std::vector<float> v; for (int p = 0; p < 128; p++) { Pitch pitch((float)p); const param f = pitch -> Frequency; v.push_back(f); } emit("pitch_to_frequency", v); }

	// --- a7/a8: Basic oscillators
	for (float f : freqs) {
		{ Generators::Basic::Sine o; o.set(f, 0.f); snprintf(nm, 128, "basic_sine_%g", f); emit(nm, run_osc(o, N)); }
		{ Generators::Basic::Saw o; o.set(f, 0.f); snprintf(nm, 128, "basic_saw_%g", f); emit(nm, run_osc(o, N)); }
		{ Generators::Basic::Triangle o; o.set(f, 0.f); snprintf(nm, 128, "basic_triangle_%g", f); emit(nm, run_osc(o, N)); }
		{ Generators::Basic::Square o; o.set(f, 0.f); snprintf(nm, 128, "basic_square_%g", f); emit(nm, run_osc(o, N)); }
		{ Generators::Basic::Pulse o; o.set(f, 0.f, 0.25f); snprintf(nm, 128, "basic_pulse25_%g", f); emit(nm, run_osc(o, N)); }
	}
	{ Generators::Basic::Sine o; o.set(440.f, 1.5f); emit("basic_sine_440_phase1.5", run_osc(o, N)); }
	{ Generators::Basic::Sine o; o.set(440.f, 0.f); signal ph(0.25f); o.set(+ph); emit("basic_sine_440_rel0.25", run_osc(o, N)); }
	{ Generators::Basic::Sine o; o.set(50.f, pi); emit("basic_sine_50_phasepi", run_osc(o, N)); }

	// --- a9: noise (glibc rand(), srand(1))
	{ srand(1); Generators::Basic::Noise o; emit("basic_noise_srand1", run_osc(o, 256)); }
	{ srand(1); Generators::Fast::Noise o; emit("fast_noise_srand1", run_osc(o, 256)); }

	// --- a10: Increment / Phase conversions
	{
		std::vector<float> amt, asf;
		for (float f : freqs) { Generators::Fast::Increment inc; inc.set(f); amt.push_back((float)(inc.amount >> 8)); asf.push_back((float)inc); }
		emit("fast_increment_amount_shr8", amt); emit("fast_increment_float", asf);
		std::vector<float> pos, back;
		const float phases[] = { 0.f, 0.5f, 1.5707964f, 3.1415927f, 4.712389f, 6.2831855f, 7.f, 12.566371f };
		for (float p : phases) { Generators::Fast::Phase ph; ph = klang::Phase(p); pos.push_back((float)(ph.position >> 8)); back.push_back((float)ph); }
		emit("fast_phase_position_shr8", pos); emit("fast_phase_float", back);
		std::vector<float> fs_, fm;
		for (unsigned i = 0; i < 1024; i++) { unsigned p = i * 4194304u + 12345u * i; fs_.push_back(Generators::Fast::fastsinp(p)); fm.push_back(fast_modp(p)); }
		emit("fastsinp_grid", fs_); emit("fast_modp_grid", fm);
	}

	// --- a11: Fast::Sine
	for (float f : freqs) { Generators::Fast::Sine o; o.set(f, 0.f); snprintf(nm, 128, "fast_sine_%g", f); emit(nm, run_osc(o, N)); }
	{ Generators::Fast::Sine o; o.set(440.f, 2.f); emit("fast_sine_440_phase2", run_osc(o, N)); }
	{ // phase modulation incl. negative offsets (F3: float->unsigned wrap on baseline x86-64)
		Generators::Fast::Sine o; o.set(440.f, 0.f);
		std::vector<float> v(N);
		for (int i = 0; i < N; i++) { signal m(3.f * noise(i)); o.set(+m); signal x; o >> x; v[i] = x; }
		emit("fast_sine_440_pm_noise3", v);
	}

	// --- a12: OSM oscillators
	for (float f : freqs) {
		{ Generators::Fast::Saw o; o.set(f, 0.f); snprintf(nm, 128, "fast_saw_%g", f); emit(nm, run_osc(o, N)); }
		{ Generators::Fast::Triangle o; o.set(f, 0.f); snprintf(nm, 128, "fast_triangle_%g", f); emit(nm, run_osc(o, N)); }
		{ Generators::Fast::Square o; o.set(f, 0.f); snprintf(nm, 128, "fast_square_%g", f); emit(nm, run_osc(o, N)); }
		{ Generators::Fast::Pulse o; o.set(f, 0.f); snprintf(nm, 128, "fast_pulse_%g", f); emit(nm, run_osc(o, N)); }
		{ Generators::Fast::Saw o; o.set(f, 0.f, 0.05f); snprintf(nm, 128, "fast_saw_duty0.05_%g", f); emit(nm, run_osc(o, N)); }
		{ Generators::Fast::Saw o; o.set(f, 1.f, 0.615f); snprintf(nm, 128, "fast_saw_phase1_duty0.615_%g", f); emit(nm, run_osc(o, N)); }
	}

	// --- a13: OnePole
	for (float f : freqs) {
		{ Filters::OnePole::LPF q; q.set(f); snprintf(nm, 128, "onepole_lpf_coef_%g", f); emit(nm, { q.b0, q.b1, q.a1 });
		  std::vector<float> v(N); for (int i = 0; i < N; i++) { signal x(noise(i)), y; x >> q >> y; v[i] = y; } snprintf(nm, 128, "onepole_lpf_%g", f); emit(nm, v); }
		{ Filters::OnePole::HPF q; q.set(f); snprintf(nm, 128, "onepole_hpf_coef_%g", f); emit(nm, { q.b0, q.b1, q.a1 });
		  std::vector<float> v(N); for (int i = 0; i < N; i++) { signal x(noise(i)), y; x >> q >> y; v[i] = y; } snprintf(nm, 128, "onepole_hpf_%g", f); emit(nm, v); }
	}

	// --- a14: Biquads
	const float Qs[] = { 0.70710678f, 0.3f, 2.f, 10.f };
	for (float f : freqs) for (float Q : Qs) {
#define BIQ(TYPE, tag, PRE) { Filters::Biquad::TYPE q; PRE; q.set(f, Q); snprintf(nm, 128, "biquad_" tag "_coef_%g_%g", f, Q); emit(nm, { q.b0, q.b1, q.b2, q.a1, q.a2 }); \
		std::vector<float> v(256); for (int i = 0; i < 256; i++) { signal x(noise(i)), y; x >> q >> y; v[i] = y; } snprintf(nm, 128, "biquad_" tag "_%g_%g", f, Q); emit(nm, v); }
		BIQ(LPF, "lpf", ) BIQ(HPF, "hpf", ) BIQ(BPF, "bpf", ) BIQ(BPF, "bpfskirt", q = Filters::Biquad::BPF::ConstantSkirtGain) BIQ(BRF, "brf", ) BIQ(APF, "apf", )
#undef BIQ
	}
	{ Filters::Biquad::LPF q; q.set(1000.f); emit("biquad_lpf_coef_default_1000", { q.b0, q.b1, q.b2, q.a1, q.a2 }); }
	{ Filters::Biquad::LPF q; q.set(1000.f, -500.f); emit("biquad_lpf_coef_negQ_1000_500", { q.b0, q.b1, q.b2, q.a1, q.a2 }); }
	{ // swept cutoff every sample (the F6 path of the shipped subtractive.k)
		Filters::Biquad::LPF q; std::vector<float> v(N);
		for (int i = 0; i < N; i++) { signal x(noise(i)), y; const param fc(500.f + 7.f * i); x >> q(fc, 10) >> y; v[i] = y; }
		emit("biquad_lpf_sweep_q10", v);
	}

	// --- a15-a17: Envelope / ADSR
	{ ADSR e; e.set(1e-4f, 1e-4f, .5f, 1e-4f); std::vector<float> v(64); for (int i = 0; i < 64; i++) v[i] = e++; emit("adsr_1e-4", v); }
	{
		ADSR e; e.set(0.01f, 0.1f, 0.7f, 0.25f); std::vector<float> v, st;
		for (int i = 0; i < 24000; i++) { if (i == 9000) e.release(); v.push_back(e++); st.push_back((float)e.getStage()); }
		std::vector<float> dec, sdec; for (int i = 0; i < 24000; i += 8) { dec.push_back(v[i]); sdec.push_back(st[i]); }
		emit("adsr_std_release9000_dec8", dec); emit("adsr_std_release9000_stage_dec8", sdec);
		emit("adsr_std_release9000_head", std::vector<float>(v.begin(), v.begin() + 1024));
		emit("adsr_std_release9000_rel", std::vector<float>(v.begin() + 8990, v.begin() + 9100));
	}
	{ ADSR e; e.set(0.f, 0.f, 1.f, 0.25f); std::vector<float> v; for (int i = 0; i < 2000; i++) { if (i == 1000) e.release(); v.push_back(e++); } emit("adsr_0_0_1_release1000", v); }
	{ ADSR e; e.set(0.001f, 0.25f, 1.f, 0.5f); std::vector<float> v; for (int i = 0; i < 2000; i++) { if (i == 20) e.release(); v.push_back(e++); } emit("adsr_release_during_attack", v); }
	{ ADSR e; e.set(0.01f, 0.1f, 0.7f, 0.25f); std::vector<float> v; for (int i = 0; i < 3000; i++) { if (i == 100) e.release(0.01f, 0.2f); v.push_back(e++); } emit("adsr_release_time_level", v); }
	{ Envelope e = { { 0, 880 }, { 0.01f, 4400 }, { 0.03f, 2200 } }; std::vector<float> v(2048), st(2048); for (int i = 0; i < 2048; i++) { v[i] = e++; st[i] = (float)e.getStage(); } emit("envelope_3pt", v); emit("envelope_3pt_stage", st); }
	{ Envelope e = { { 0, 0 }, { 0.005f, 1 }, { 0.01f, 0.25f }, { 0.02f, 0.5f } }; e.setLoop(1, 3); std::vector<float> v(4096); for (int i = 0; i < 4096; i++) v[i] = e++; emit("envelope_loop_1_3", v); }
	{ Envelope e; std::vector<float> v(16), st(16); for (int i = 0; i < 16; i++) { v[i] = e++; st[i] = (float)e.getStage(); } emit("envelope_default", v); emit("envelope_default_stage", st); }
	{ Envelope e = { { 0, 1.5f }, { 3, 0.5f } }; std::vector<float> v(1024); for (int i = 0; i < 1024; i++) v[i] = e++; emit("envelope_fm_op2", v); }
	{ Envelope e; e.setMode(Envelope::Rate); e = { { 0, 0 }, { 0.001f, 1 }, { 0.0005f, 0.2f } }; std::vector<float> v(4096); for (int i = 0; i < 4096; i++) v[i] = e++; emit("envelope_rate_mode", v); }
	// (round 6) any number of points, loops over later points, Rate mode with a jump point (x == 0), release() in Rate mode (the `time` argument is the RATE there: setTarget({ time, level }, 0) -> setTargetRate)
	{ Envelope e = { { 0, 0 }, { 0.004f, 1 }, { 0.009f, 0.3f }, { 0.013f, 0.8f }, { 0.02f, 0.1f }, { 0.024f, 0.6f }, { 0.05f, 0 } }; e.setLoop(2, 5); std::vector<float> v(8192), st(8192); for (int i = 0; i < 8192; i++) { if (i == 6000) e.resetLoop(); v[i] = e++; st[i] = (float)e.getStage(); } emit("envelope_7pt_loop_2_5", v); emit("envelope_7pt_loop_2_5_stage", st); }
	{ Envelope e = { { 0, 0.5f }, { 0.002f, 1 }, { 0.004f, 0 }, { 0.006f, 0.7f }, { 0.008f, 0.2f }, { 0.01f, 0.9f }, { 0.012f, 0.1f }, { 0.014f, 0.6f }, { 0.016f, 0.3f }, { 0.03f, 0 } }; std::vector<float> v(2048), st(2048); for (int i = 0; i < 2048; i++) { v[i] = e++; st[i] = (float)e.getStage(); } emit("envelope_10pt", v); emit("envelope_10pt_stage", st); }
	{ Envelope e = { { 0, 0 }, { 0.004f, 1 }, { 0.009f, 0.3f }, { 0.013f, 0.8f }, { 0.02f, 0.1f }, { 0.024f, 0.6f } }; e.setLoop(5, 5); std::vector<float> v(2048), st(2048); for (int i = 0; i < 2048; i++) { if (i == 1500) e.release(0.004f, 0.05f); v[i] = e++; st[i] = (float)e.getStage(); } emit("envelope_6pt_hold_5_release", v); emit("envelope_6pt_hold_5_release_stage", st); }
	{ Envelope e; e.setMode(Envelope::Rate); e = { { 0, 0 }, { 0.002f, 1 }, { 0, 0.25f }, { 0.001f, 0.75f }, { 0.0005f, 0.5f }, { 0.004f, 0 } }; std::vector<float> v(4096), st(4096); for (int i = 0; i < 4096; i++) { v[i] = e++; st[i] = (float)e.getStage(); } emit("envelope_rate_6pt_jump", v); emit("envelope_rate_6pt_jump_stage", st); }
	{ Envelope e; e.setMode(Envelope::Rate); e = { { 0, 0 }, { 0.002f, 1 }, { 0.001f, 0.25f }, { 0.003f, 0.75f }, { 0.0005f, 0.5f } }; e.setLoop(1, 3); std::vector<float> v(4096), st(4096); for (int i = 0; i < 4096; i++) { if (i == 3000) e.release(0.0007f, 0.1f); v[i] = e++; st[i] = (float)e.getStage(); } emit("envelope_rate_loop_1_3_release", v); emit("envelope_rate_loop_1_3_release_stage", st); }

	// --- a18: Operator chain (FM.k shape, fixed indices)
	{
		Operator<Generators::Fast::Sine> op1, op2, op3;
		op1.set(220.f, 0); op1 = { { 0, 0 }, { 3, 1 } };
		op2.set(220.f, 0); op2 = { { 0, 1.5f }, { 3, 0.5f } };
		op3.set(440.f, 0);
		std::vector<float> v(N); const param I1 = 3.7f, I2 = 1.37f;
		for (int i = 0; i < N; i++) { signal x; op1 * I1 >> op2 * I2 >> op3 >> x; v[i] = x; }
		emit("operator_chain3", v);
	}

	// --- a19: Delay
	{
		Delay<16> d; std::vector<float> v; d.set(3.5f);
		for (int i = 1; i <= 40; i++) { signal x((float)i), y; x >> d >> y; v.push_back(y); }
		emit("delay16_set3.5", v);
	}
	{
		Delay<16> d; std::vector<float> a, b, c;
		for (int i = 1; i <= 40; i++) { signal x((float)(i * i % 17)); x >> d; a.push_back(d.tap(5)); b.push_back(d.tap(2.25f)); c.push_back(d.lagrange(3.6f)); }
		emit("delay16_tap_int5", a); emit("delay16_tap_2.25", b); emit("delay16_lagrange_3.6", c);
	}
	{
		Delay<1000> d; std::vector<float> v;
		for (int i = 0; i < 3000; i++) { d.set(100.f + 50.f * noise(i)); signal x(noise(i + 7777)), y; x >> d >> y; v.push_back(y); }
		emit("delay1000_modulated_set", v);
	}
	{
		Delay<0> d; d.resize(100); std::vector<float> v;
		for (int i = 0; i < 400; i++) { d.set(33.25f); signal x(noise(i)), y; x >> d >> y; v.push_back(y); }
		emit("delay0_100_set33.25", v);
	}
	{
		Stereo::Delay<64> d; std::vector<float> v;
		for (int i = 0; i < 200; i++) { Stereo::signal x = { noise(i), noise(i + 5000) }; x >> d; Stereo::signal y = d(10.75f); v.push_back(y.l); v.push_back(y.r); }
		emit("stereo_delay64_tap10.75", v);
	}

	// --- a21: Matrix
	{
		constexpr Matrix m = { 0, 1, 1, -1,  -1, 0, -1, 1,  -1, 1, 0, -1,  1, -1, 1, 0 };
		std::vector<float> v;
		for (int i = 0; i < 16; i++) { const signals<4> in = { noise(4 * i), noise(4 * i + 1), noise(4 * i + 2), noise(4 * i + 3) }; const signals<4> o = in >> m; for (int k = 0; k < 4; k++) v.push_back(o[k]); }
		emit("matrix_fdn", v);
	}

	// --- a22: Control::smooth / set clamp
	{
		Control c = Dial("x", 0.001f, 1.f, 0.5f); std::vector<float> v;
		for (int i = 0; i < 512; i++) v.push_back(c.smooth());
		c.set(7.f); v.push_back(c.value); c.set(-7.f); v.push_back(c.value);
		emit("control_smooth_0.5", v);
	}

	// --- row f2: the remaining filters / modifiers that share the hot loop's shape
	auto run_mod = [&](auto& q, int n, float scale) { std::vector<float> v(n); for (int i = 0; i < n; i++) { signal x(scale * noise(i)), y; x >> q >> y; v[i] = y; } return v; };
	{ Filters::DCF q; emit("dcf_default", run_mod(q, N, 1.f)); }
	{ Filters::DCF q; q.set(0.9f); emit("dcf_0.9", run_mod(q, N, 1.f)); }
	{ Filters::IIR<2> q; q.set(-1.2f, 0.5f); emit("iir2", run_mod(q, N, 1.f)); }
	{ Filters::IIR<4> q; q.set(-0.5f, 0.25f, -0.125f, 0.0625f); emit("iir4", run_mod(q, N, 1.f)); }
	{ Filters::IIR<1> q; q.set(0.25f); emit("iir1_0.25", run_mod(q, N, 1.f)); }
	for (float f : freqs) {
		{ Filters::Butterworth::LPF<1> q; q.set(f); snprintf(nm, 128, "butter1_coef_%g", f); emit(nm, { q.b0, q.a1 }); snprintf(nm, 128, "butter1_%g", f); emit(nm, run_mod(q, N, 1.f)); }
		{ Filters::Butterworth::LPF<2> q; q.set(f); snprintf(nm, 128, "butter2_coef_%g", f); emit(nm, { q.b0, q.b1, q.b2, q.a1, q.a2 }); snprintf(nm, 128, "butter2_%g", f); emit(nm, run_mod(q, N, 1.f)); }
	}
	{
		const float modal[3][3] = { { 440.f, 0.5f, 0.f }, { 1000.f, 0.05f, 0.f }, { 110.f, 2.0f, 0.5f } };
		for (int k = 0; k < 3; k++) {
			Modifiers::Modal q;
			if (modal[k][2] != 0.f) q.set(modal[k][0], modal[k][1], modal[k][2]); else q.set(modal[k][0], modal[k][1]);
			snprintf(nm, 128, "modal_coef_%d", k); emit(nm, { q.a1, q.a2, q.gain });
			std::vector<float> v(N);
			for (int i = 0; i < N; i++) { signal x((i % 97) == 0 ? 1.f : 0.25f * noise(i)), y; x >> q >> y; v[i] = y; }   // excitation: clicks on a noise floor
			snprintf(nm, 128, "modal_%d", k); emit(nm, v);
		}
	}
	{
		Envelope::Follower::AR ar; ar.set(0.01f, 0.1f); emit("follower_ar_coef", { ar.A, ar.R });
		std::vector<float> v(N);
		for (int i = 0; i < N; i++) { signal x(fabsf(noise(i)) * ((i / 200) % 2 ? 0.1f : 1.f)), y; x >> ar >> y; v[i] = y; }
		emit("follower_ar", v);
	}
	{ Envelope::Follower q; q = Peak; std::vector<float> v(N); for (int i = 0; i < N; i++) { signal x(noise(i) * ((i / 200) % 2 ? 0.1f : 1.f)), y; x >> q >> y; v[i] = y; } emit("follower_peak", v); }
	{ Envelope::Follower q; q = RMS; std::vector<float> v(N); for (int i = 0; i < N; i++) { signal x(noise(i) * ((i / 200) % 2 ? 0.1f : 1.f)), y; x >> q >> y; v[i] = y; } emit("follower_rms", v); }

	fclose(g_out);
	return 0;
}
