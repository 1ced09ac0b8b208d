/* oracle/ko_kat.c — TEST INFRASTRUCTURE ONLY.
 * Re-creates every known-answer vector of oracle/ref/ref_prims.cpp (same names, same stimuli)
 * with the C restatement, in the same record format, so tests/test_oracle_kat.py can compare the
 * two files record-by-record, bit-for-bit. */
#include "klang_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static FILE* g_out;
static void emit(const char* name, const float* v, unsigned c) {
	unsigned n = (unsigned)strlen(name);
	fwrite(&n, 4, 1, g_out); fwrite(name, 1, n, g_out);
	fwrite(&c, 4, 1, g_out); fwrite(v, 4, c, g_out);
}
static float noise(unsigned n) { return ko_fx_input(1u, 0u, 0u, n, 0xFFFFFFFFu); }

#define N 1024
#define NB 256
static float buf[32768];

int main(int argc, char** argv) {
	if (argc < 2) { fprintf(stderr, "usage: %s out.kat\n", argv[0]); return 1; }
	g_out = fopen(argv[1], "wb");
	ko_set_fs(48000.f);
	const float freqs[] = { 27.5f, 110.f, 440.f, 1000.f, 2093.0045f, 7040.f, 15000.f };
	const int NF = 7;
	char nm[128];

	for (int p = 0; p < 128; p++) buf[p] = ko_pitch_to_frequency((float)p);
	emit("pitch_to_frequency", buf, 128);

	for (int k = 0; k < NF; k++) {
		const float f = freqs[k]; ko_osc o;
		ko_osc_init(&o); ko_osc_set_fp(&o, f, 0.f); for (int i = 0; i < N; i++) buf[i] = ko_basic_sine(&o); snprintf(nm, 128, "basic_sine_%g", f); emit(nm, buf, N);
		ko_osc_init(&o); ko_osc_set_fp(&o, f, 0.f); for (int i = 0; i < N; i++) buf[i] = ko_basic_saw(&o); snprintf(nm, 128, "basic_saw_%g", f); emit(nm, buf, N);
		ko_osc_init(&o); ko_osc_set_fp(&o, f, 0.f); for (int i = 0; i < N; i++) buf[i] = ko_basic_triangle(&o); snprintf(nm, 128, "basic_triangle_%g", f); emit(nm, buf, N);
		ko_osc_init(&o); ko_osc_set_fp(&o, f, 0.f); for (int i = 0; i < N; i++) buf[i] = ko_basic_square(&o); snprintf(nm, 128, "basic_square_%g", f); emit(nm, buf, N);
		ko_osc_init(&o); ko_osc_set_fp(&o, f, 0.f); o.duty = 0.25f; for (int i = 0; i < N; i++) buf[i] = ko_basic_pulse(&o); snprintf(nm, 128, "basic_pulse25_%g", f); emit(nm, buf, N);
	}
	{ ko_osc o; ko_osc_init(&o); ko_osc_set_fp(&o, 440.f, 1.5f); for (int i = 0; i < N; i++) buf[i] = ko_basic_sine(&o); emit("basic_sine_440_phase1.5", buf, N); }
	{ ko_osc o; ko_osc_init(&o); ko_osc_set_fp(&o, 440.f, 0.f); ko_osc_set_rel(&o, 0.25f); for (int i = 0; i < N; i++) buf[i] = ko_basic_sine(&o); emit("basic_sine_440_rel0.25", buf, N); }
	{ ko_osc o; ko_osc_init(&o); ko_osc_set_fp(&o, 50.f, (float)3.1415926535897932384626433832795); for (int i = 0; i < N; i++) buf[i] = ko_basic_sine(&o); emit("basic_sine_50_phasepi", buf, N); }

	srand(1); for (int i = 0; i < 256; i++) buf[i] = ko_basic_noise(); emit("basic_noise_srand1", buf, 256);
	srand(1); for (int i = 0; i < 256; i++) buf[i] = ko_fast_noise(); emit("fast_noise_srand1", buf, 256);

	{
		float a[8], b[8];
		for (int k = 0; k < NF; k++) { const int32_t inc = ko_fast_increment(freqs[k]); a[k] = (float)(inc >> 8); b[k] = ko_fast_increment_float(inc); }
		emit("fast_increment_amount_shr8", a, NF); emit("fast_increment_float", b, NF);
		const float phases[] = { 0.f, 0.5f, 1.5707964f, 3.1415927f, 4.712389f, 6.2831855f, 7.f, 12.566371f };
		for (int k = 0; k < 8; k++) { const uint32_t p = ko_fast_phase(phases[k]); a[k] = (float)(p >> 8); b[k] = ko_fast_phase_float(p); }
		emit("fast_phase_position_shr8", a, 8); emit("fast_phase_float", b, 8);
		for (unsigned i = 0; i < 1024; i++) { const unsigned p = i * 4194304u + 12345u * i; buf[i] = ko_fastsinp(p); buf[2048 + i] = ko_fast_modp(p); }
		emit("fastsinp_grid", buf, 1024); emit("fast_modp_grid", buf + 2048, 1024);
	}

	for (int k = 0; k < NF; k++) { ko_fsine o; ko_fsine_init(&o); ko_fsine_set_fp(&o, freqs[k], 0.f); for (int i = 0; i < N; i++) buf[i] = ko_fsine_process(&o); snprintf(nm, 128, "fast_sine_%g", freqs[k]); emit(nm, buf, N); }
	{ ko_fsine o; ko_fsine_init(&o); ko_fsine_set_fp(&o, 440.f, 2.f); for (int i = 0; i < N; i++) buf[i] = ko_fsine_process(&o); emit("fast_sine_440_phase2", buf, N); }
	{ ko_fsine o; ko_fsine_init(&o); ko_fsine_set_fp(&o, 440.f, 0.f); for (int i = 0; i < N; i++) { ko_fsine_set_rel(&o, 3.f * noise(i)); buf[i] = ko_fsine_process(&o); } emit("fast_sine_440_pm_noise3", buf, N); }

	for (int k = 0; k < NF; k++) {
		const float f = freqs[k]; ko_osm o;
		ko_osm_init(&o, KO_OSM_SAW, 0.f); ko_osm_set_fp(&o, f, 0.f); for (int i = 0; i < N; i++) buf[i] = ko_osm_process(&o); snprintf(nm, 128, "fast_saw_%g", f); emit(nm, buf, N);
		ko_osm_init(&o, KO_OSM_SAW, 1.f); ko_osm_set_fp(&o, f, 0.f); for (int i = 0; i < N; i++) buf[i] = ko_osm_process(&o); snprintf(nm, 128, "fast_triangle_%g", f); emit(nm, buf, N);
		ko_osm_init(&o, KO_OSM_PULSE, 1.0f); ko_osm_set_fp(&o, f, 0.f); for (int i = 0; i < N; i++) buf[i] = ko_osm_process(&o); snprintf(nm, 128, "fast_square_%g", f); emit(nm, buf, N);
		ko_osm_init(&o, KO_OSM_PULSE, 0.5f); ko_osm_set_fp(&o, f, 0.f); for (int i = 0; i < N; i++) buf[i] = ko_osm_process(&o); snprintf(nm, 128, "fast_pulse_%g", f); emit(nm, buf, N);
		ko_osm_init(&o, KO_OSM_SAW, 0.f); ko_osm_set_fpd(&o, f, 0.f, 0.05f); for (int i = 0; i < N; i++) buf[i] = ko_osm_process(&o); snprintf(nm, 128, "fast_saw_duty0.05_%g", f); emit(nm, buf, N);
		ko_osm_init(&o, KO_OSM_SAW, 0.f); ko_osm_set_fpd(&o, f, 1.f, 0.615f); for (int i = 0; i < N; i++) buf[i] = ko_osm_process(&o); snprintf(nm, 128, "fast_saw_phase1_duty0.615_%g", f); emit(nm, buf, N);
	}

	for (int k = 0; k < NF; k++) {
		const float f = freqs[k];
		for (int t = 0; t < 2; t++) {
			ko_onepole q; ko_onepole_init(&q, t); ko_onepole_set(&q, f);
			const float c[3] = { q.b0, q.b1, q.a1 };
			snprintf(nm, 128, "onepole_%s_coef_%g", t ? "hpf" : "lpf", f); emit(nm, c, 3);
			for (int i = 0; i < N; i++) buf[i] = ko_onepole_process(&q, noise(i));
			snprintf(nm, 128, "onepole_%s_%g", t ? "hpf" : "lpf", f); emit(nm, buf, N);
		}
	}

	const float Qs[] = { 0.70710678f, 0.3f, 2.f, 10.f };
	const char* tags[] = { "lpf", "hpf", "bpf", "bpfskirt", "brf", "apf" };
	for (int k = 0; k < NF; k++) for (int qi = 0; qi < 4; qi++) for (int t = 0; t < 6; t++) {
		const float f = freqs[k], Q = Qs[qi];
		ko_biquad q; ko_biquad_init(&q, t);
		if (t == KO_BQ_BPF_SKIRT) { /* `q = ConstantSkirtGain` calls init() on the fresh filter (a = 0, cos0 = 1, sin0 = 0) */
			q.a1 = -2.f; q.a2 = 1.f; q.b0 = 0.f; q.b1 = 0.f; q.b2 = -0.f;
		}
		ko_biquad_set_fq(&q, f, Q);
		const float c[5] = { q.b0, q.b1, q.b2, q.a1, q.a2 };
		snprintf(nm, 128, "biquad_%s_coef_%g_%g", tags[t], f, Q); emit(nm, c, 5);
		for (int i = 0; i < NB; i++) buf[i] = ko_biquad_process(&q, noise(i));
		snprintf(nm, 128, "biquad_%s_%g_%g", tags[t], f, Q); emit(nm, buf, NB);
	}
	{ ko_biquad q; ko_biquad_init(&q, KO_BQ_LPF); ko_biquad_set_f(&q, 1000.f); const float c[5] = { q.b0, q.b1, q.b2, q.a1, q.a2 }; emit("biquad_lpf_coef_default_1000", c, 5); }
	{ ko_biquad q; ko_biquad_init(&q, KO_BQ_LPF); ko_biquad_set_fq(&q, 1000.f, -500.f); const float c[5] = { q.b0, q.b1, q.b2, q.a1, q.a2 }; emit("biquad_lpf_coef_negQ_1000_500", c, 5); }
	{ ko_biquad q; ko_biquad_init(&q, KO_BQ_LPF); for (int i = 0; i < N; i++) { ko_biquad_set_fq(&q, 500.f + 7.f * i, 10.f); buf[i] = ko_biquad_process(&q, noise(i)); } emit("biquad_lpf_sweep_q10", buf, N); }

	{ ko_adsr e; ko_adsr_init(&e); ko_adsr_set(&e, 1e-4f, 1e-4f, .5f, 1e-4f); for (int i = 0; i < 64; i++) buf[i] = ko_env_process(&e.env); emit("adsr_1e-4", buf, 64); }
	{
		static float v[24000], st[24000], dec[3000], sdec[3000];
		ko_adsr e; ko_adsr_init(&e); ko_adsr_set(&e, 0.01f, 0.1f, 0.7f, 0.25f);
		for (int i = 0; i < 24000; i++) { if (i == 9000) ko_adsr_release(&e, 0.f, 0.f); v[i] = ko_env_process(&e.env); st[i] = (float)e.env.stage; }
		int c = 0; for (int i = 0; i < 24000; i += 8) { dec[c] = v[i]; sdec[c] = st[i]; c++; }
		emit("adsr_std_release9000_dec8", dec, c); emit("adsr_std_release9000_stage_dec8", sdec, c);
		emit("adsr_std_release9000_head", v, 1024);
		emit("adsr_std_release9000_rel", v + 8990, 110);
	}
	{ ko_adsr e; ko_adsr_init(&e); ko_adsr_set(&e, 0.f, 0.f, 1.f, 0.25f); for (int i = 0; i < 2000; i++) { if (i == 1000) ko_adsr_release(&e, 0.f, 0.f); buf[i] = ko_env_process(&e.env); } emit("adsr_0_0_1_release1000", buf, 2000); }
	{ ko_adsr e; ko_adsr_init(&e); ko_adsr_set(&e, 0.001f, 0.25f, 1.f, 0.5f); for (int i = 0; i < 2000; i++) { if (i == 20) ko_adsr_release(&e, 0.f, 0.f); buf[i] = ko_env_process(&e.env); } emit("adsr_release_during_attack", buf, 2000); }
	{ ko_adsr e; ko_adsr_init(&e); ko_adsr_set(&e, 0.01f, 0.1f, 0.7f, 0.25f); for (int i = 0; i < 3000; i++) { if (i == 100) ko_adsr_release(&e, 0.01f, 0.2f); buf[i] = ko_env_process(&e.env); } emit("adsr_release_time_level", buf, 3000); }
	{ ko_env e; ko_env_init_default(&e); const float xy[6] = { 0, 880, 0.01f, 4400, 0.03f, 2200 }; ko_env_set_points(&e, 3, xy); for (int i = 0; i < 2048; i++) { buf[i] = ko_env_process(&e); buf[4096 + i] = (float)e.stage; } emit("envelope_3pt", buf, 2048); emit("envelope_3pt_stage", buf + 4096, 2048); }
	{ ko_env e; ko_env_init_default(&e); const float xy[8] = { 0, 0, 0.005f, 1, 0.01f, 0.25f, 0.02f, 0.5f }; ko_env_set_points(&e, 4, xy); ko_env_set_loop(&e, 1, 3); for (int i = 0; i < 4096; i++) buf[i] = ko_env_process(&e); emit("envelope_loop_1_3", buf, 4096); }
	{ ko_env e; ko_env_init_default(&e); for (int i = 0; i < 16; i++) { buf[i] = ko_env_process(&e); buf[64 + i] = (float)e.stage; } emit("envelope_default", buf, 16); emit("envelope_default_stage", buf + 64, 16); }
	{ ko_env e; ko_env_init_default(&e); const float xy[4] = { 0, 1.5f, 3, 0.5f }; ko_env_set_points(&e, 2, xy); for (int i = 0; i < 1024; i++) buf[i] = ko_env_process(&e); emit("envelope_fm_op2", buf, 1024); }
	{ ko_env e; ko_env_init_default(&e); e.mode = KO_ENV_RATE; const float xy[6] = { 0, 0, 0.001f, 1, 0.0005f, 0.2f }; ko_env_set_points(&e, 3, xy); for (int i = 0; i < 4096; i++) buf[i] = ko_env_process(&e); emit("envelope_rate_mode", buf, 4096); }
	/* (round 6, row a16) any number of points, loops over later points, Rate mode with a jump point, release() in Rate mode */
	{ ko_env e; ko_env_init_default(&e); const float xy[14] = { 0, 0, 0.004f, 1, 0.009f, 0.3f, 0.013f, 0.8f, 0.02f, 0.1f, 0.024f, 0.6f, 0.05f, 0 }; ko_env_set_points(&e, 7, xy); ko_env_set_loop(&e, 2, 5);
	  for (int i = 0; i < 8192; i++) { if (i == 6000) ko_env_reset_loop(&e); buf[i] = ko_env_process(&e); buf[8192 + i] = (float)e.stage; } emit("envelope_7pt_loop_2_5", buf, 8192); emit("envelope_7pt_loop_2_5_stage", buf + 8192, 8192); }
	{ ko_env e; ko_env_init_default(&e); const float xy[20] = { 0, 0.5f, 0.002f, 1, 0.004f, 0, 0.006f, 0.7f, 0.008f, 0.2f, 0.01f, 0.9f, 0.012f, 0.1f, 0.014f, 0.6f, 0.016f, 0.3f, 0.03f, 0 }; ko_env_set_points(&e, 10, xy);
	  for (int i = 0; i < 2048; i++) { buf[i] = ko_env_process(&e); buf[4096 + i] = (float)e.stage; } emit("envelope_10pt", buf, 2048); emit("envelope_10pt_stage", buf + 4096, 2048); }
	{ ko_env e; ko_env_init_default(&e); const float xy[12] = { 0, 0, 0.004f, 1, 0.009f, 0.3f, 0.013f, 0.8f, 0.02f, 0.1f, 0.024f, 0.6f }; ko_env_set_points(&e, 6, xy); ko_env_set_loop(&e, 5, 5);
	  for (int i = 0; i < 2048; i++) { if (i == 1500) ko_env_release(&e, 0.004f, 0.05f); buf[i] = ko_env_process(&e); buf[4096 + i] = (float)e.stage; } emit("envelope_6pt_hold_5_release", buf, 2048); emit("envelope_6pt_hold_5_release_stage", buf + 4096, 2048); }
	{ ko_env e; ko_env_init_default(&e); e.mode = KO_ENV_RATE; const float xy[12] = { 0, 0, 0.002f, 1, 0, 0.25f, 0.001f, 0.75f, 0.0005f, 0.5f, 0.004f, 0 }; ko_env_set_points(&e, 6, xy);
	  for (int i = 0; i < 4096; i++) { buf[i] = ko_env_process(&e); buf[4096 + i] = (float)e.stage; } emit("envelope_rate_6pt_jump", buf, 4096); emit("envelope_rate_6pt_jump_stage", buf + 4096, 4096); }
	{ ko_env e; ko_env_init_default(&e); e.mode = KO_ENV_RATE; const float xy[10] = { 0, 0, 0.002f, 1, 0.001f, 0.25f, 0.003f, 0.75f, 0.0005f, 0.5f }; ko_env_set_points(&e, 5, xy); ko_env_set_loop(&e, 1, 3);
	  for (int i = 0; i < 4096; i++) { if (i == 3000) ko_env_release(&e, 0.0007f, 0.1f); buf[i] = ko_env_process(&e); buf[4096 + i] = (float)e.stage; } emit("envelope_rate_loop_1_3_release", buf, 4096); emit("envelope_rate_loop_1_3_release_stage", buf + 4096, 4096); }

	{
		ko_operator op1, op2, op3; ko_operator_init(&op1); ko_operator_init(&op2); ko_operator_init(&op3);
		ko_fsine_set_fp(&op1.osc, 220.f, 0.f); { const float xy[4] = { 0, 0, 3, 1 }; ko_env_set_points(&op1.env, 2, xy); }
		ko_fsine_set_fp(&op2.osc, 220.f, 0.f); { const float xy[4] = { 0, 1.5f, 3, 0.5f }; ko_env_set_points(&op2.env, 2, xy); }
		ko_fsine_set_fp(&op3.osc, 440.f, 0.f);
		for (int i = 0; i < N; i++) {
			op1.amp = 3.7f; const float m1 = ko_operator_process(&op1);
			op2.amp = 1.37f; op2.in = m1; const float m2 = ko_operator_process(&op2);
			op3.in = m2; buf[i] = ko_operator_process(&op3);
		}
		emit("operator_chain3", buf, N);
	}

	{ ko_delay d; ko_delay_create(&d, 16); ko_delay_set(&d, 3.5f); for (int i = 1; i <= 40; i++) { ko_delay_input(&d, (float)i); buf[i - 1] = ko_delay_process(&d); } emit("delay16_set3.5", buf, 40); ko_delay_destroy(&d); }
	{
		ko_delay d; ko_delay_create(&d, 16);
		for (int i = 1; i <= 40; i++) { ko_delay_input(&d, (float)(i * i % 17)); buf[i - 1] = ko_delay_tap_int(&d, 5); buf[100 + i - 1] = ko_delay_tap_float(&d, 2.25f); buf[200 + i - 1] = ko_delay_lagrange(&d, 3.6f); }
		emit("delay16_tap_int5", buf, 40); emit("delay16_tap_2.25", buf + 100, 40); emit("delay16_lagrange_3.6", buf + 200, 40); ko_delay_destroy(&d);
	}
	{ ko_delay d; ko_delay_create(&d, 1000); for (int i = 0; i < 3000; i++) { ko_delay_set(&d, 100.f + 50.f * noise(i)); ko_delay_input(&d, noise(i + 7777)); buf[i] = ko_delay_process(&d); } emit("delay1000_modulated_set", buf, 3000); ko_delay_destroy(&d); }
	{ ko_delay d; ko_delay_create(&d, 100); for (int i = 0; i < 400; i++) { ko_delay_set(&d, 33.25f); ko_delay_input(&d, noise(i)); buf[i] = ko_delay_process(&d); } emit("delay0_100_set33.25", buf, 400); ko_delay_destroy(&d); }
	{
		ko_delay l, r; ko_delay_create(&l, 64); ko_delay_create(&r, 64);
		for (int i = 0; i < 200; i++) { ko_delay_input(&l, noise(i)); ko_delay_input(&r, noise(i + 5000)); ko_stereo_delay_tap_float(&l, &r, 10.75f, &buf[2 * i], &buf[2 * i + 1]); }
		emit("stereo_delay64_tap10.75", buf, 400); ko_delay_destroy(&l); ko_delay_destroy(&r);
	}

	{
		const float m[16] = { 0, 1, 1, -1,  -1, 0, -1, 1,  -1, 1, 0, -1,  1, -1, 1, 0 };
		for (int i = 0; i < 16; i++) { const float in[4] = { noise(4 * i), noise(4 * i + 1), noise(4 * i + 2), noise(4 * i + 3) }; ko_matrix_mul(m, in, &buf[4 * i]); }
		emit("matrix_fdn", buf, 64);
	}
	{
		ko_control c = { 0.001f, 1.f, 0.5f, 0.f };
		for (int i = 0; i < 512; i++) buf[i] = ko_control_smooth(&c);
		ko_control_set(&c, 7.f); buf[512] = c.value; ko_control_set(&c, -7.f); buf[513] = c.value;
		emit("control_smooth_0.5", buf, 514);
	}
	/* --- row f2 --- */
	{ ko_dcf q; ko_dcf_init(&q); for (int i = 0; i < N; i++) buf[i] = ko_dcf_process(&q, noise(i)); emit("dcf_default", buf, N); }
	{ ko_dcf q; ko_dcf_init(&q); q.r = 0.9f; for (int i = 0; i < N; i++) buf[i] = ko_dcf_process(&q, noise(i)); emit("dcf_0.9", buf, N); }
	{ const float c[2] = { -1.2f, 0.5f }; ko_iir q; ko_iir_init(&q, 2, c); for (int i = 0; i < N; i++) buf[i] = ko_iir_process(&q, noise(i)); emit("iir2", buf, N); }
	{ const float c[4] = { -0.5f, 0.25f, -0.125f, 0.0625f }; ko_iir q; ko_iir_init(&q, 4, c); for (int i = 0; i < N; i++) buf[i] = ko_iir_process(&q, noise(i)); emit("iir4", buf, N); }
	{ ko_iir1 q; ko_iir1_init(&q); ko_iir1_set(&q, 0.25f); for (int i = 0; i < N; i++) buf[i] = ko_iir1_process(&q, noise(i)); emit("iir1_0.25", buf, N); }
	for (int k = 0; k < NF; k++) {
		const float f = freqs[k];
		{ ko_butter1 q; ko_butter1_init(&q); ko_butter1_set(&q, f); const float c[2] = { q.b0, q.a1 }; snprintf(nm, 128, "butter1_coef_%g", f); emit(nm, c, 2);
		  for (int i = 0; i < N; i++) buf[i] = ko_butter1_process(&q, noise(i));
		  snprintf(nm, 128, "butter1_%g", f); emit(nm, buf, N); }
		{ ko_biquad q; ko_biquad_init(&q, KO_BQ_LPF); ko_butter2_set(&q, f); const float c[5] = { q.b0, q.b1, q.b2, q.a1, q.a2 }; snprintf(nm, 128, "butter2_coef_%g", f); emit(nm, c, 5);
		  for (int i = 0; i < N; i++) buf[i] = ko_biquad_process(&q, noise(i));
		  snprintf(nm, 128, "butter2_%g", f); emit(nm, buf, N); }
	}
	{
		const float modal[3][3] = { { 440.f, 0.5f, 0.f }, { 1000.f, 0.05f, 0.f }, { 110.f, 2.0f, 0.5f } };
		for (int k = 0; k < 3; k++) {
			ko_modal q; ko_modal_init(&q);
			if (modal[k][2] != 0.f) ko_modal_set_gain(&q, modal[k][0], modal[k][1], modal[k][2]); else ko_modal_set(&q, modal[k][0], modal[k][1]);
			const float c[3] = { q.a1, q.a2, q.gain }; snprintf(nm, 128, "modal_coef_%d", k); emit(nm, c, 3);
			for (int i = 0; i < N; i++) buf[i] = ko_modal_process(&q, (i % 97) == 0 ? 1.f : 0.25f * noise(i));
			snprintf(nm, 128, "modal_%d", k); emit(nm, buf, N);
		}
	}
	{
		ko_follower_ar q; ko_follower_ar_init(&q); ko_follower_ar_set(&q, 0.01f, 0.1f); const float c[2] = { q.A, q.R }; emit("follower_ar_coef", c, 2);
		for (int i = 0; i < N; i++) buf[i] = ko_follower_ar_process(&q, fabsf(noise(i)) * ((i / 200) % 2 ? 0.1f : 1.f));
		emit("follower_ar", buf, N);
	}
	{ ko_follower_ar q; ko_follower_ar_init(&q); ko_follower_ar_set(&q, 0.01f, 0.1f); for (int i = 0; i < N; i++) buf[i] = ko_follower_peak(&q, noise(i) * ((i / 200) % 2 ? 0.1f : 1.f)); emit("follower_peak", buf, N); }
	{ ko_follower_ar q; ko_follower_ar_init(&q); ko_follower_ar_set(&q, 0.01f, 0.1f); for (int i = 0; i < N; i++) buf[i] = ko_follower_rms(&q, noise(i) * ((i / 200) % 2 ? 0.1f : 1.f)); emit("follower_rms", buf, N); }
	fclose(g_out);
	return 0;
}
