/* oracle/klang_oracle.c — TEST INFRASTRUCTURE ONLY.  See klang_oracle.h.
 *
 * CPU restatement of the reference's per-sample signal-graph path in plain C.
 * All citations are file:line into /root/reference/klang.h (v0.7.8) unless a
 * patch file is named.  Written from the behaviour of the reference, operation
 * by operation, with the exact fp32 evaluation order; validated bit-for-bit
 * against tests/golden/ (vectors produced by the genuine header).
 *
 * Compile with -ffp-contract=off: the reference path is bit-stable only without
 * FMA contraction.
 */
#include "klang_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------- constants: klang.h:93-111 (constant), 227-233 (pi, root2) ---------- */
#define KO_PI_D      3.1415926535897932384626433832795
#define KO_PI_F      ((float)KO_PI_D)
#define KO_PI_INV    ((float)(1.0 / KO_PI_D))
#define KO_TWO_PI    (2.f * KO_PI_F)                 /* `2 * pi` as used by Phase/Oscillator */
#define KO_ROOT2_INV ((float)(1.0 / 1.4142135623730950488016887242097))
#define KO_DENORM    1.175494e-38f                   /* DENORMALISE klang.h:90 */

ko_fs_t ko_fs = { 44100.f, 44100, 44100.0, 1.f / 44100.f, 0.f, 22050.f };

/* SampleRate ctor klang.h:1601 */
void ko_set_fs(float sr) {
	ko_fs.f = sr;
	ko_fs.i = (int)(sr + 0.001f);
	ko_fs.d = (double)sr;
	ko_fs.inv = 1.f / sr;
	ko_fs.w = 2.0f * KO_PI_F * ko_fs.inv;
	ko_fs.nyquist = sr / 2.f;
}

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* float -> unsigned as clang emits it on baseline x86-64 (cvttss2si r64 + truncate);
 * reproduces the F3 wrap of negative FM offsets (klang.h:4995-4996). */
static inline uint32_t f2u_wrap(float x) { return (uint32_t)(int64_t)x; }

/* ---------- unit conversion: Pitch::operator-> klang.h:1568-1571, power() 191-217 ---------- */
float ko_pitch_to_frequency(float pitch) {
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
	else p = powf(2.f, e);
	return 440.f * p;
}

/* ---------- Phase::operator+=(float) klang.h:1518-1525 ---------- */
static inline void phase_advance(float* value, float inc) {
	if (inc >= KO_TWO_PI) return;
	*value += inc;
	if (*value > KO_TWO_PI) *value -= KO_TWO_PI;
}

/* ---------- Generic::Oscillator klang.h:2849-2880 ---------- */
void ko_osc_init(ko_osc* o) { o->frequency = 1000.f; o->increment = 0.f; o->position = 0.f; o->offset = 0.f; o->duty = 0.5f; o->out = 0.f; }
void ko_osc_set_f(ko_osc* o, float f) { o->frequency = f; o->increment = f * 2.f * KO_PI_F / ko_fs.f; }          /* 2862-2865 */
void ko_osc_set_fp(ko_osc* o, float f, float phase) { o->position = phase; ko_osc_set_f(o, f); }                /* 2867-2870 */
void ko_osc_set_rel(ko_osc* o, float rel) { o->offset = rel * KO_TWO_PI; }                                      /* 2877-2879 */

/* Generators::Basic klang.h:4899-4944 : compute from current position, then advance */
float ko_basic_sine(ko_osc* o) { o->out = (float)sin((double)(o->position + o->offset)); phase_advance(&o->position, o->increment); return o->out; }
float ko_basic_saw(ko_osc* o) { o->out = o->position * KO_PI_INV - 1.f; phase_advance(&o->position, o->increment); return o->out; }
float ko_basic_triangle(ko_osc* o) { o->out = fabsf(2.f * o->position * KO_PI_INV - 2.f) - 1.f; phase_advance(&o->position, o->increment); return o->out; }
float ko_basic_square(ko_osc* o) { o->out = o->position > KO_PI_F ? 1.f : -1.f; phase_advance(&o->position, o->increment); return o->out; }
float ko_basic_pulse(ko_osc* o) { o->out = o->position > (o->duty * KO_PI_F) ? 1.f : -1.f; phase_advance(&o->position, o->increment); return o->out; }

/* Basic::Noise klang.h:4947-4951 ; Fast::Noise 5357-5366 (libc rand(), F5) */
float ko_basic_noise(void) { return rand() * 2.f / (float)RAND_MAX - 1.f; }
float ko_fast_noise(void) { const uint32_t i = (((uint32_t)rand() & 0x7FFFu) << 1) | 0x43800000u; return u2f(i) - 257.f; }

/* ---------- Generators::Fast ---------- */
int32_t ko_fast_increment(float f) {                       /* Increment::set klang.h:4971-4977 */
	const float FC4 = (float)261.62556530059862;
	const float FC4_FINTMAX = (float)(261.62556530059862 * 2147483648.0);
	const float FBASE = FC4_FINTMAX / ko_fs.f;
	return (int32_t)(2u * (uint32_t)(int32_t)(FBASE / FC4 * f));
}
float ko_fast_increment_float(int32_t amount) {            /* klang.h:4979-4983 (arithmetic shift) */
	const uint32_t i = (uint32_t)((amount >> 9) | 0x3f800000);
	return u2f(i) - 1.f;
}
uint32_t ko_fast_phase(float radians) {                    /* Phase::operator= klang.h:4993-4998 */
	const float p = radians * 2147483648.0f / (2.f * KO_PI_F);
	return f2u_wrap(p);
}
float ko_fast_phase_float(uint32_t pos) { return u2f((pos >> 9) | 0x3f800000u) - 1.f; }   /* klang.h:5004-5007 */
float ko_fast_modp(uint32_t x) { return (u2f((x >> 9) | 0x3f800000u) - 1.f) * KO_TWO_PI; } /* klang.h:1424-1428 */
float ko_polysin(float x) {                                 /* klang.h:5093-5096 */
	const float x2 = x * x;
	return (((-0.00018542f * x2 + 0.0083143f) * x2 - 0.16666f) * x2 + 1.0f) * x;
}
float ko_fastsinp(uint32_t p) {                             /* klang.h:5117-5132 */
	float x = ko_fast_modp(p);
	if (x > 3.f / 2.f * KO_PI_F) x -= KO_TWO_PI;
	else if (x > KO_PI_F / 2.f) x = KO_PI_F - x;
	return ko_polysin(x);
}

/* Fast::Sine klang.h:5135-5172 */
void ko_fsine_init(ko_fsine* o) { o->frequency = 1000.f; o->inc = 0; o->pos = 0; o->off = 0; o->base_offset = 0.f; o->out = 0.f; }
void ko_fsine_set_f(ko_fsine* o, float f) { if (f != o->frequency) { o->frequency = f; o->inc = ko_fast_increment(f); } }  /* 5142-5147 */
void ko_fsine_set_fp(ko_fsine* o, float f, float phase) {                                                             /* 5149-5153 */
	o->pos = ko_fast_phase(phase);
	o->base_offset = 0.f; o->off = ko_fast_phase(0.f);
	ko_fsine_set_f(o, f);
}
void ko_fsine_set_rel(ko_fsine* o, float rel) { o->base_offset = rel * KO_TWO_PI; o->off = ko_fast_phase(o->base_offset); }  /* 5160-5162 */
float ko_fsine_process(ko_fsine* o) { o->out = ko_fastsinp(o->pos + o->off); o->pos += (uint32_t)o->inc; return o->out; }  /* 5164-5167 */

/* ---------- Fast::OSM klang.h:5175-5317 ---------- */
static void osm_coeffs(ko_osm* o) {                          /* OSM::init 5206-5215 */
	o->state = ((uint32_t)(o->offset - (uint32_t)o->inc) < o->duty) ? 3 : 0;
	o->f = o->delta;
	o->omf = 1.f - o->f;
	o->rcpf = 1.f / o->f;
	o->rcpf2 = 2.f * o->rcpf;
	o->col = ko_fast_phase_float(o->duty);
	o->c1 = 1.f / o->col;
	o->c2 = -1.f / (1.0f - o->col);
}
static void osm_set_duty(ko_osm* o, float duty) { o->duty = ko_fast_phase(duty * (2.f * KO_PI_F)); osm_coeffs(o); }   /* 5246-5249 */
static void osm_refresh(ko_osm* o, float f) {
	if (o->frequency != f) { o->frequency = f; o->inc = ko_fast_increment(f); o->delta = ko_fast_increment_float(o->inc); }
}
void ko_osm_init(ko_osm* o, int waveform, float duty) {      /* Osm ctor 5323; Saw/Triangle/Square/Pulse 5348-5354 */
	memset(o, 0, sizeof(*o));
	o->waveform = waveform;
	osm_set_duty(o, duty);
}
void ko_osm_set_f(ko_osm* o, float f) { if (o->frequency != f) { osm_refresh(o, f); osm_coeffs(o); } }             /* 5217-5224 */
void ko_osm_set_fp(ko_osm* o, float f, float phase) { osm_refresh(o, f); o->offset = ko_fast_phase(phase); osm_coeffs(o); }  /* 5226-5234 */
void ko_osm_set_fpd(ko_osm* o, float f, float phase, float duty) { osm_refresh(o, f); o->offset = ko_fast_phase(phase); osm_set_duty(o, duty); }  /* 5236-5244 */

static inline int osm_tick(ko_osm* o) {                      /* 5251-5263 */
	o->state = ((o->state << 1) | (o->offset < o->duty ? 1 : 0)) & 3;
	const int tr = o->state | (o->offset < (uint32_t)o->inc ? 4 : 0);
	o->offset += (uint32_t)o->inc;
	return tr;
}
static inline float sqrf(float x) { return x * x; }
float ko_osm_process(ko_osm* o) {                            /* output() 5266; saw 5290-5302; pulse 5304-5316 */
	float y;
	if (o->waveform == KO_OSM_SAW) {
		const float p = ko_fast_phase_float(o->offset) - o->col;   /* evaluated before tick() (clang, left-to-right) */
		const float f = o->f, omf = o->omf, rcpf = o->rcpf, c1 = o->c1, c2 = o->c2;
		switch (osm_tick(o)) {
		case 3: y = c1 * (p + p - f) + 1.f; break;                                   /* Up */
		case 0: y = c2 * (p + p - f) + 1.f; break;                                   /* Down */
		case 2: y = rcpf * (c2 * sqrf(p) - c1 * sqrf(p - f)) + 1.f; break;          /* UpDown */
		case 5: y = -rcpf * (1.f + c2 * sqrf(p + omf) - c1 * sqrf(p)) + 1.f; break; /* DownUp */
		case 7: y = -rcpf * (1.f + c1 * omf * (p + p + omf)) + 1.f; break;          /* UpDownUp */
		case 4: y = -rcpf * (1.f + c2 * omf * (p + p + omf)) + 1.f; break;          /* DownUpDown */
		default: y = 0.f;
		}
	}
	else {
		const float p = ko_fast_phase_float(o->offset);
		const float rcpf2 = o->rcpf2, col = o->col;
		switch (osm_tick(o)) {
		case 3: y = 1.f; break;
		case 0: y = -1.f; break;
		case 2: y = rcpf2 * (col - p) + 1.f; break;
		case 5: y = rcpf2 * p - 1.f; break;
		case 7: y = rcpf2 * (col - 1.0f) + 1.f; break;
		case 4: y = rcpf2 * col - 1.f; break;
		default: y = 0.f;
		}
	}
	o->out = y;
	return y;
}

/* ---------- Filters::OnePole klang.h:5470-5543 ---------- */
void ko_onepole_init(ko_onepole* q, int type) { q->type = type; q->f = 0; q->a1 = 0; q->b0 = 1; q->b1 = 0; q->z = 0; q->in = 0; q->out = 0; }
void ko_onepole_set(ko_onepole* q, float f) {
	if (q->f != f) {
		q->f = f;
		const float exp0 = expf(-q->f * ko_fs.w);
		if (q->type == KO_OP_LPF) { q->b0 = 1 - exp0; q->a1 = exp0; }                     /* 5510-5514 */
		else { q->b0 = 0.5f * (1.f + exp0); q->b1 = -q->b0; q->a1 = exp0; }              /* 5537-5542 */
	}
}
float ko_onepole_process(ko_onepole* q, float in) {
	q->in = in;
	if (q->type == KO_OP_LPF) q->out = q->b0 * q->in + q->a1 * q->out + KO_DENORM;      /* 5516-5518 */
	else { q->out = q->b0 * q->in + q->b1 * q->z + q->a1 * q->out + KO_DENORM; q->z = q->in; } /* 5499-5502 */
	return q->out;
}

/* ---------- row f2: the remaining filters / modifiers ---------- */
/* Filters::DCF klang.h:5386-5397 */
void ko_dcf_init(ko_dcf* q) { q->r = 0.995f; q->z = 0; q->in = 0; q->out = 0; }
float ko_dcf_process(ko_dcf* q, float in) { q->in = in; q->out = q->in - q->z + q->r * q->out; q->z = q->in; return q->out; }
/* Filters::IIR<ORDER> klang.h:5399-5432 */
void ko_iir_init(ko_iir* q, int order, const float* coeffs) { memset(q, 0, sizeof(*q)); q->order = order; for (int i = 0; i < order; i++) q->a[i] = coeffs[i]; }
float ko_iir_process(ko_iir* q, float in) {
	q->out = in;
	for (int i = 0; i < q->order; i++) q->out -= q->a[i] * q->y[i];
	for (int i = q->order - 1; i > 0; --i) q->y[i] = q->y[i - 1];
	q->y[0] = q->out;
	return q->out;
}
/* Filters::IIR<1> klang.h:5434-5464 */
void ko_iir1_init(ko_iir1* q) { q->a = 1; q->b = 0; q->out = 0; }
void ko_iir1_set(ko_iir1* q, float coeff) { q->a = coeff; q->b = 1.f - q->a; }
float ko_iir1_process(ko_iir1* q, float in) { q->out = in * q->a + q->out * q->b; return q->out; }
/* Filters::Butterworth::LPF<1> klang.h:5786-5799 over OnePole::Filter::set 5489-5494 */
void ko_butter1_init(ko_butter1* q) { q->f = 0; q->a1 = 0; q->b0 = 1; q->z = 0; q->out = 0; }
void ko_butter1_set(ko_butter1* q, float f) {
	if (q->f != f) {
		q->f = f;
		const float c = 1.f / tanf(KO_PI_F * q->f * ko_fs.inv);
		const double a0 = (double)(1.f + c);                        /* constant a0 = { 1.f + c } */
		const float inv = (a0 == 0.0f) ? 0.0f : (float)(1.0 / a0);
		q->b0 = inv;
		q->a1 = (1.f - c) * inv;
	}
}
float ko_butter1_process(ko_butter1* q, float in) { q->out = q->b0 * (in + q->z) - q->a1 * q->out; q->z = in; return q->out; }
/* Filters::Butterworth::LPF<2> klang.h:5801-5811: Biquad::Filter::set(f) (Q = root2.inv) with this init() */
void ko_butter2_set(ko_biquad* q, float f) {
	float Q = KO_ROOT2_INV;
	if (q->f != f || q->Q != Q) {
		q->f = f; q->Q = Q;
		const float w = f * ko_fs.w;
		q->cos0 = cosf(w); q->sin0 = sinf(w);
		if (Q < 0.5) Q = 0.5f;
		q->a = q->sin0 / (2.f * Q);
		const double a0 = (double)(1.f + q->a);
		const float inv = (a0 == 0.0f) ? 0.0f : (float)(1.0 / a0);
		q->b0 = inv * ((1.f - q->cos0) / 2.f);
		q->b1 = inv * (1.f - q->cos0);
		q->b2 = inv * ((1.f - q->cos0) / 2.f);
		q->a1 = inv * (-2.f * q->cos0);
		q->a2 = inv * (1.f - q->a);
	}
}
/* Modifiers::Modal klang.h:5815-5859 */
static float ko_clampf(float x, float lo, float hi) { return x < lo ? lo : (hi < x ? hi : x); }
void ko_modal_init(ko_modal* q) { memset(q, 0, sizeof(*q)); q->gain = 0.05f; }
void ko_modal_set(ko_modal* q, float f, float decay) {
	q->gain = 0.05f;
	const float w = f * ko_fs.w;
	const float d = ko_clampf(expf(-KO_PI_F / (decay * ko_fs.f)), 1e-6f, 0.9999f);
	q->a1 = 2.f * d * ko_clampf(cosf(w), -0.9999f, 0.9999f);
	q->a2 = -d * d;
	q->y2 = 0; q->y1 = 0; q->in = 0; q->out = 0;
}
void ko_modal_set_gain(ko_modal* q, float f, float decay, float gain) { ko_modal_set(q, f, decay); q->gain = ko_clampf(gain * 0.05f, -0.05f, 0.05f); }
float ko_modal_process(ko_modal* q, float in) {
	q->in = in; q->in *= q->gain;                                   /* input() */
	q->out = q->in + q->a1 * q->y1 + q->a2 * q->y2;
	q->y2 = q->y1; q->y1 = q->out; q->in = 0;
	return q->out;
}
/* Envelope::Follower klang.h:5862-5903 */
void ko_follower_ar_init(ko_follower_ar* q) { q->attack = 0; q->release = 0; q->A = 1; q->R = 1; q->out = 0; }
void ko_follower_ar_set(ko_follower_ar* q, float attack, float release) {
	if (q->attack != attack || q->release != release) {
		q->attack = attack; q->release = release;
		q->A = 1.f - (attack == 0.f ? 0.f : expf(-1.0f / (ko_fs.f * attack)));
		q->R = 1.f - (release == 0.f ? 0.f : expf(-1.0f / (ko_fs.f * release)));
	}
}
float ko_follower_ar_process(ko_follower_ar* q, float in) { const float smoothing = in > q->out ? q->A : q->R; q->out = q->out + smoothing * (in - q->out); return q->out; }
float ko_follower_peak(ko_follower_ar* q, float in) { return ko_follower_ar_process(q, fabsf(in)); }
float ko_follower_rms(ko_follower_ar* q, float in) { return sqrtf(ko_follower_ar_process(q, in * in)); }

/* ---------- Filters::Biquad klang.h:5550-5773 ---------- */
void ko_biquad_reset(ko_biquad* q) { q->f = 0; q->Q = 0; q->b0 = 1; q->a1 = q->a2 = q->b1 = q->b2 = 0; q->a = 0; q->z0 = q->z1 = 0; }  /* 5565-5572 */
void ko_biquad_init(ko_biquad* q, int type) { memset(q, 0, sizeof(*q)); q->type = type; q->b0 = 1; q->cos0 = 1; }

static void biquad_coeffs(ko_biquad* q) {
	const float a = q->a, cos0 = q->cos0, sin0 = q->sin0;
	if (q->type == KO_BQ_APF) {                                  /* 5765-5772 */
		const float omega = 2.0f * KO_PI_F * q->f / ko_fs.f;
		const float c = (float)cos((double)omega);
		q->b0 = q->a2 = a * a;
		q->b1 = q->a1 = (-2.f * a * c);
		q->b2 = 1.f;
		return;
	}
	const double a0 = (double)(1.f + a);                         /* constant a0 = { 1.f + a }  klang.h:97 */
	const float inv = (a0 == 0.0f) ? 0.0f : (float)(1.0 / a0);
	q->a1 = inv * (-2.f * cos0);
	q->a2 = inv * (1.f - a);
	switch (q->type) {
	case KO_BQ_LPF: q->b2 = q->b0 = inv * (1.f - cos0) * 0.5f; q->b1 = inv * (1.f - cos0); break;      /* 5658-5665 */
	case KO_BQ_HPF: q->b2 = q->b0 = inv * (1.f + cos0) * 0.5f; q->b1 = inv * -(1.f + cos0); break;     /* 5675-5682 */
	case KO_BQ_BPF_PEAK: q->b0 = inv * a; q->b1 = 0; q->b2 = inv * -a; break;                          /* 5720-5729 */
	case KO_BQ_BPF_SKIRT: q->b0 = inv * sin0 * 0.5f; q->b1 = 0; q->b2 = -q->b0; break;                 /* 5708-5717 */
	case KO_BQ_BRF: q->b1 = q->a1; q->b0 = q->b2 = inv; break;                                          /* 5734-5739 */
	}
}
void ko_biquad_set_fq(ko_biquad* q, float f, float Q) {
	if (q->type == KO_BQ_APF) {                                  /* APF::set 5752-5763 (second arg is the radius r) */
		if (q->f != f || q->a != Q) {
			q->f = f; q->a = Q;
			const float w = f * ko_fs.w;
			q->cos0 = cosf(w); q->sin0 = sinf(w);
			biquad_coeffs(q);
		}
		return;
	}
	if (Q < 0) Q = f / -Q;                                       /* 5584-5586 */
	if (q->f != f || q->Q != Q) {                                /* 5588-5600 */
		q->f = f; q->Q = Q;
		const float w = f * ko_fs.w;
		q->cos0 = cosf(w);
		q->sin0 = sinf(w);
		if (Q < 0.5) Q = 0.5f;
		q->a = q->sin0 / (2.f * Q);
		biquad_coeffs(q);
	}
}
void ko_biquad_set_f(ko_biquad* q, float f) { ko_biquad_set_fq(q, f, q->type == KO_BQ_APF ? 1.f : KO_ROOT2_INV); }  /* 5575, 5749 */
float ko_biquad_process(ko_biquad* q, float in) {               /* TDF-II 5605-5612 */
	q->in = in;
	const float z0 = q->z0, z1 = q->z1;
	const float y = q->b0 * q->in + z0;
	q->z0 = q->b1 * q->in - q->a1 * y + z1;
	q->z1 = q->b2 * q->in - q->a2 * y;
	q->out = y;
	return y;
}

/* ---------- Envelope klang.h:3722-4102 ---------- */
static inline void ramp_set_value(ko_env* e, float v) { e->r_out = v; e->r_target = v; e->r_active = 0; }           /* 3762-3766 */
static inline void ramp_set_target(ko_env* e, float t) { e->r_target = t; e->r_active = (e->r_out != t); }         /* 3756-3759 */
static void env_set_target(ko_env* e, float px, float py, float time) {
	if (e->mode == KO_ENV_TIME) {                                /* setTargetTime 4077-4081 */
		e->time = time;
		ramp_set_target(e, py);
		e->r_rate = fabsf(py - e->r_out) / ((px - time) * ko_fs.f);
	}
	else {                                                       /* setTargetRate 4083-4092 */
		e->time = 0;
		if (px == 0) ramp_set_value(e, py);
		else { ramp_set_target(e, py); e->r_rate = px; }
	}
}
static void env_initialise(ko_env* e) {                          /* 3974-3989 */
	e->point = 0;
	e->timeInc = 1.0f / ko_fs.f;
	e->loop_start = e->loop_end = -1;
	e->stage = KO_ENV_SUSTAIN;
	if (e->npoints) {
		e->out = e->py[0];
		ramp_set_value(e, e->py[0]);
		if (e->npoints > 1) env_set_target(e, e->px[1], e->py[1], e->px[0]);
	}
	else { e->out = 1.0f; ramp_set_value(e, 1.0f); }
}
void ko_env_init_default(ko_env* e) {                            /* Envelope() : ramp(new Linear()) { set(Points(0.f, 1.f)); } */
	memset(e, 0, sizeof(*e));
	e->mode = KO_ENV_TIME;
	ramp_set_value(e, 1.f);                                      /* Ramp(float value = 1.f) 3738-3740 */
	const float xy[2] = { 0.f, 1.f };
	ko_env_set_points(e, 1, xy);
}
void ko_env_set_points(ko_env* e, int n, const float* xy) {     /* 3893-3896 */
	if (n > KO_ENV_MAX_POINTS) { fprintf(stderr, "ko_env_set_points: %d points (KO_ENV_MAX_POINTS = %d)\n", n, KO_ENV_MAX_POINTS); abort(); }
	e->npoints = n;
	for (int i = 0; i < n; i++) { e->px[i] = xy[2 * i]; e->py[i] = xy[2 * i + 1]; }
	env_initialise(e);
}
void ko_env_set_loop(ko_env* e, int start, int end) { if (start >= 0 && end < e->npoints) { e->loop_start = start; e->loop_end = end; } }  /* 3923-3926 */
void ko_env_reset_loop(ko_env* e) {                             /* 3946-3950 */
	e->loop_start = e->loop_end = -1;
	if (e->stage == KO_ENV_SUSTAIN && (e->point + 1) < e->npoints) env_set_target(e, e->px[e->point + 1], e->py[e->point + 1], e->px[e->point]);
}
void ko_env_release(ko_env* e, float time, float level) { e->stage = KO_ENV_RELEASE; env_set_target(e, time, level, 0.f); }            /* 3961-3966 */

float ko_env_process(ko_env* e) {                                /* 4018-4051 */
	/* out = (*ramp)++  — Linear::operator++ 3785-3806: return pre-step value, then step */
	e->out = e->r_out;
	if (e->r_active) {
		if (e->r_target > e->r_out) {
			e->r_out += e->r_rate;
			if (e->r_out >= e->r_target) { e->r_out = e->r_target; e->r_active = 0; }
		}
		else {
			e->r_out -= e->r_rate;
			if (e->r_out <= e->r_target) { e->r_out = e->r_target; e->r_active = 0; }
		}
	}
	switch (e->stage) {
	case KO_ENV_SUSTAIN:
		e->time += e->timeInc;
		if (!e->r_active) {
			const int loop_active = (e->loop_start != -1 && e->loop_end != -1);
			if (loop_active && (e->point + 1) >= e->loop_end) {
				e->point = e->loop_start;
				ramp_set_value(e, e->py[e->point]);
				if (e->loop_start != e->loop_end)
					env_set_target(e, e->px[e->point + 1], e->py[e->point + 1], e->px[e->point]);
			}
			else if ((e->point + 1) < e->npoints) {
				if (e->mode == KO_ENV_RATE || e->time >= e->px[e->point + 1]) {
					e->point++;
					ramp_set_value(e, e->py[e->point]);
					if ((e->point + 1) < e->npoints)
						env_set_target(e, e->px[e->point + 1], e->py[e->point + 1], e->px[e->point]);
				}
			}
			else e->stage = KO_ENV_OFF;
		}
		break;
	case KO_ENV_RELEASE:
		if (!e->r_active) e->stage = KO_ENV_OFF;
		break;
	default: break;
	}
	return e->out;
}

/* ADSR klang.h:4105-4137 */
void ko_adsr_set(ko_adsr* a, float at, float de, float su, float re) {
	a->A = at;
	a->D = de + 0.005f;
	a->S = su;
	a->R = re + 0.005f;
	const float xy[6] = { 0.f, 0.f, a->A, 1.f, a->A + a->D, a->S };
	ko_env_set_points(&a->env, 3, xy);
	ko_env_set_loop(&a->env, 2, 2);
}
void ko_adsr_init(ko_adsr* a) { memset(a, 0, sizeof(*a)); a->env.mode = KO_ENV_TIME; ramp_set_value(&a->env, 1.f); ko_adsr_set(a, 0.5f, 0.5f, 1.f, 0.5f); }
void ko_adsr_release(ko_adsr* a, float time, float level) { ko_env_release(&a->env, time ? time : a->R, level); }

/* ---------- Operator<Fast::Sine> klang.h:4140-4180 ---------- */
void ko_operator_init(ko_operator* op) { ko_fsine_init(&op->osc); op->in = 0.f; ko_env_init_default(&op->env); op->amp = 1.f; }
float ko_operator_process(ko_operator* op) {                     /* 4164-4168 */
	ko_fsine_set_rel(&op->osc, op->in);                          /* OSCILLATOR::set(+in) */
	ko_fsine_process(&op->osc);
	op->osc.out *= ko_env_process(&op->env) * op->amp;
	return op->osc.out;
}

/* ---------- Delay klang.h:3381-3512 (== Delay<0> 3515-3624 with run-time SIZE) ---------- */
int ko_delay_create(ko_delay* d, int size) {
	memset(d, 0, sizeof(*d));
	d->size = size; d->time = 1;
	d->buf = (float*)calloc((size_t)size + 1, sizeof(float));
	return d->buf ? 0 : -1;
}
void ko_delay_destroy(ko_delay* d) { free(d->buf); d->buf = NULL; }
void ko_delay_input(ko_delay* d, float in) {                     /* 3396-3403 */
	d->in = in;
	d->buf[d->position] = in;
	d->position++;
	if (d->position == d->size) d->position = 0;
}
void ko_delay_set(ko_delay* d, float samples) {                  /* 3480-3489 */
	d->time = samples < d->size ? samples : (float)d->size;
	float read = (float)(d->position - 1) - d->time;
	if (read < 0.f) read += d->size;
	d->last_position = (int)read;
	d->last_fraction = read - d->last_position;
}
float ko_delay_process(ko_delay* d) {                            /* tap() 3461-3468, process 3470-3473 */
	const int i = d->last_position;
	const int j = (i + 1) % d->size;
	d->out = d->buf[i] + d->last_fraction * (d->buf[j] - d->buf[i]);
	d->last_position = (d->last_position + 1) % d->size;
	return d->out;
}
float ko_delay_tap_int(const ko_delay* d, int delay) {           /* 3405-3410 */
	int read = (d->position - 1) - delay;
	if (read < 0) read += d->size;
	return d->buf[read];
}
float ko_delay_tap_float(const ko_delay* d, float delay) {       /* 3412-3427 */
	float read = (float)(d->position - 1) - delay;
	if (read < 0.f) read += d->size;
	const int i = (int)read;
	const float fraction = read - i;
	const int j = (i + 1) % d->size;
	return d->buf[i] + fraction * (d->buf[j] - d->buf[i]);
}
float ko_delay_lagrange(const ko_delay* d, float delay) {        /* 3429-3458 */
	const int SIZE = d->size;
	float read = (float)(d->position - 1) - delay;
	if (read < 0.f) read += SIZE;
	const int i = (int)read;
	const float x = read - i;
	const float y0 = d->buf[(i - 1 + SIZE) % SIZE], y1 = d->buf[i], y2 = d->buf[(i + 1) % SIZE], y3 = d->buf[(i + 2) % SIZE];
	const float c0 = (-x * (x - 1) * (x - 2)) / 6.0f;
	const float c1 = ((x + 1) * (x - 1) * (x - 2)) / 2.0f;
	const float c2 = (-x * (x + 1) * (x - 2)) / 2.0f;
	const float c3 = (x * (x + 1) * (x - 1)) / 6.0f;
	return c0 * y0 + c1 * y1 + c2 * y2 + c3 * y3;
}
void ko_stereo_delay_tap_float(const ko_delay* l, const ko_delay* r, float delay, float* outl, float* outr) {  /* Stereo::Delay::tap(float) 4668-4681 */
	const int SIZE = l->size;
	float read = (float)(l->position - 1) - delay;
	if (read < 0.f) read += SIZE;
	const float f = (float)floor((double)read);
	delay = read - f;
	const int i = (int)read;
	const int j = (i == (SIZE - 1)) ? 0 : (i + 1);
	/* `read` within half an ulp below zero rounds to exactly SIZE.  The reference then reads buffer[SIZE] (the pad of `buffer(SIZE + 1, 0)`, klang.h:3391:
	 * 0) and buffer[SIZE + 1] — beyond the buffer's size, inside its power-of-two allocation, never initialised (klang.h:2018-2020, 2062): indeterminate in
	 * the reference.  The restatement (and the device) DEFINE that corner: both elements read as 0. */
	if (i >= SIZE) { *outl = 0.f * (1.f - delay) + 0.f * delay; *outr = *outl; return; }
	*outl = l->buf[i] * (1.f - delay) + l->buf[j] * delay;
	*outr = r->buf[i] * (1.f - delay) + r->buf[j] * delay;
}

/* ---------- Matrix klang.h:1462-1467 ; Control klang.h:1715-1728 ---------- */
void ko_matrix_mul(const float m[16], const float in[4], float out[4]) {
	for (int r = 0; r < 4; r++)
		out[r] = m[4 * r + 0] * in[0] + m[4 * r + 1] * in[1] + m[4 * r + 2] * in[2] + m[4 * r + 3] * in[3];
}
float ko_control_smooth(ko_control* c) { c->smoothed = c->smoothed * 0.999f + (1.f - 0.999f) * c->value; return c->smoothed; }
void ko_control_set(ko_control* c, float x) { c->value = (x < c->min) ? c->min : (c->max < x) ? c->max : x; }

/* ---------- shared synthetic effect input (same definition as oracle/ref/ref_common.h) ---------- */
static inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
float ko_fx_input(uint32_t seed, uint32_t instance, uint32_t ch, uint32_t t, uint32_t burst) {
	if (t >= burst) return 0.f;
	const uint32_t h = hash32(seed ^ hash32(instance * 2u + ch) ^ (t * 0x9e3779b9U));
	return (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

/* =====================================================================
 * Patch level
 * ===================================================================== */
enum { ST_ONSET = 0, ST_SUSTAIN = 1, ST_RELEASE = 2, ST_OFF = 3 };   /* NoteBase::Stage klang.h:4286 */

typedef struct {
	int stage; float pitch, velocity, out;
	union {
		struct { ko_fsine osc; } sine;
		struct { ko_osc osc; } bsine;
		struct { ko_osm osc; ko_biquad lpf; ko_adsr adsr; } s2a;
		struct { ko_osm osc; ko_adsr adsr; ko_env env; ko_biquad filter; } s2b;
		struct { ko_osm osc[7]; ko_adsr adsr; } ss;
		struct { ko_operator op[4]; ko_adsr adsr; } fm;
	} u;
} ko_note;

typedef struct {
	ko_note* notes; int count;
	ko_control controls[16]; int ncontrols;
	unsigned noteOns; unsigned noteStart[128];
} ko_synth;

struct ko_bank { int patch, S, P; float fs; ko_synth* synths; };

static const char* const PATCH_NAMES[KO_PATCH_COUNT] = { "sine", "bsine", "sub2a", "sub2b", "supersaw", "fm3", "fm4", "pingpong", "reverb" };
int ko_patch_from_name(const char* name) {
	for (int i = 0; i < KO_PATCH_COUNT; i++) if (!strcmp(name, PATCH_NAMES[i])) return i;
	return -1;
}

static void dial(ko_synth* s, double mn, double mx, double initial) {   /* Dial() klang.h:1797-1800 */
	ko_control* c = &s->controls[s->ncontrols++];
	c->min = (float)mn; c->max = (float)mx; c->value = (float)initial; c->smoothed = 0.f;
}

static void note_construct(ko_note* n, int patch) {
	memset(n, 0, sizeof(*n));
	n->stage = ST_OFF;
	switch (patch) {
	case KO_PATCH_SINE: ko_fsine_init(&n->u.sine.osc); break;
	case KO_PATCH_BSINE: ko_osc_init(&n->u.bsine.osc); break;
	case KO_PATCH_SUB2A: ko_osm_init(&n->u.s2a.osc, KO_OSM_SAW, 0.f); ko_biquad_init(&n->u.s2a.lpf, KO_BQ_LPF); ko_adsr_init(&n->u.s2a.adsr); break;
	case KO_PATCH_SUB2B: ko_osm_init(&n->u.s2b.osc, KO_OSM_PULSE, 1.0f); ko_adsr_init(&n->u.s2b.adsr); ko_env_init_default(&n->u.s2b.env); ko_biquad_init(&n->u.s2b.filter, KO_BQ_LPF); break;
	case KO_PATCH_SUPERSAW: for (int s = 0; s < 7; s++) ko_osm_init(&n->u.ss.osc[s], KO_OSM_SAW, 0.f); ko_adsr_init(&n->u.ss.adsr); break;
	case KO_PATCH_FM3: case KO_PATCH_FM4: for (int k = 0; k < 4; k++) ko_operator_init(&n->u.fm.op[k]); ko_adsr_init(&n->u.fm.adsr); break;
	}
}

ko_bank* ko_bank_create(int patch, int synths, int notes_per_synth, float fs) {
	if (patch < 0 || patch >= KO_PATCH_PINGPONG || notes_per_synth > 128) return NULL;
	ko_set_fs(fs);
	ko_bank* b = (ko_bank*)calloc(1, sizeof(*b));
	b->patch = patch; b->S = synths; b->P = notes_per_synth; b->fs = fs;
	b->synths = (ko_synth*)calloc((size_t)synths, sizeof(ko_synth));
	for (int i = 0; i < synths; i++) {
		ko_synth* s = &b->synths[i];
		s->count = notes_per_synth;
		s->notes = (ko_note*)calloc((size_t)notes_per_synth, sizeof(ko_note));
		for (int n = 0; n < notes_per_synth; n++) note_construct(&s->notes[n], patch);
		switch (patch) {
		case KO_PATCH_SUPERSAW:                                   /* examples/SuperSaw.k:38-43 */
			dial(s, 0.001, 1, 0.001); dial(s, 0, 1, 0.05); dial(s, 0, 1, 0.6); break;
		case KO_PATCH_FM3:                                        /* examples/FM.k:79-85 */
			dial(s, 0.001, 10.0, 1.0); dial(s, 0.000, 10.0, 0.37); dial(s, 0.000, 10.0, 0.37); dial(s, 0.000, 1.0, 0.5); break;
		case KO_PATCH_FM4:                                        /* oracle/ref/ref_fm.cpp FM4 */
			dial(s, 0.001, 10.0, 1.0); dial(s, 0.000, 10.0, 0.37); dial(s, 0.000, 10.0, 0.37); dial(s, 0.000, 10.0, 0.37); dial(s, 0.000, 1.0, 0.5); break;
		}
	}
	return b;
}
void ko_bank_destroy(ko_bank* b) {
	if (!b) return;
	for (int i = 0; i < b->S; i++) free(b->synths[i].notes);
	free(b->synths); free(b);
}
int ko_bank_voices(const ko_bank* b) { return b->S * b->P; }
void ko_bank_control(ko_bank* b, int synth, int index, float value) {
	if (index < b->synths[synth].ncontrols) ko_control_set(&b->synths[synth].controls[index], value);
}

/* Notes::assign klang.h:4336-4372 */
static int synth_assign(ko_synth* s) {
	for (int i = 0; i < s->count; i++)
		if (s->notes[i].stage == ST_OFF) { s->noteStart[i] = s->noteOns++; return i; }
	int oldest = -1; unsigned oldest_start = 0;
	for (int i = 0; i < s->count; i++)
		if (s->notes[i].stage == ST_RELEASE && (oldest == -1 || s->noteStart[i] < oldest_start)) { oldest = i; oldest_start = s->noteStart[i]; }
	if (oldest != -1) { s->noteStart[oldest] = s->noteOns++; return oldest; }
	oldest = -1; oldest_start = 0;
	for (int i = 0; i < s->count; i++)
		if (oldest == -1 || s->noteStart[i] < oldest_start) { oldest = i; oldest_start = s->noteStart[i]; }
	s->noteStart[oldest] = s->noteOns++;
	return oldest;
}

/* random<double>(min,max) klang.h:236 */
static double random_d(double mn, double mx) { return rand() * ((mx - mn) / (double)RAND_MAX) + mn; }

static void note_on(ko_bank* b, ko_synth* s, ko_note* n) {      /* user on() of each patch */
	const float f = ko_pitch_to_frequency(n->pitch);
	switch (b->patch) {
	case KO_PATCH_SINE: ko_fsine_set_fp(&n->u.sine.osc, f, 0.f); break;          /* oracle/ref/ref_sine.cpp */
	case KO_PATCH_BSINE: ko_osc_set_fp(&n->u.bsine.osc, f, 0.f); break;
	case KO_PATCH_SUB2A:                                                        /* oracle/ref/ref_subtractive.cpp Sub2a */
		ko_osm_set_fp(&n->u.s2a.osc, f, 0.f);
		ko_biquad_reset(&n->u.s2a.lpf);
		ko_biquad_set_fq(&n->u.s2a.lpf, 4.f * f, 2.f);
		ko_adsr_set(&n->u.s2a.adsr, 0.01f, 0.1f, 0.7f, 0.25f);
		break;
	case KO_PATCH_SUB2B: {                                                      /* templates/juce/synth/Source/subtractive.k:14-23 */
		ko_osm_set_fp(&n->u.s2b.osc, f, 0.f);
		ko_adsr_set(&n->u.s2b.adsr, 0.f, 0.f, 1.f, 0.25f);
		const float xy[6] = { 0.f, f * 2.f, 0.25f, f * 10.f, 2.f, f * 5.f };
		ko_env_set_points(&n->u.s2b.env, 3, xy);
		ko_biquad_reset(&n->u.s2b.filter);
	} break;
	case KO_PATCH_SUPERSAW: {                                                   /* examples/SuperSaw.k:12-19 */
		const float detune = (float)(0.01 * (double)s->controls[2].value * (double)f);
		for (int k = 0; k < 7; k++) {
			const double d = (double)((float)(k - 3) * detune) * random_d(0.999, 1.001);
			ko_osm_set_fpd(&n->u.ss.osc[k], f + (float)d, 0.f, s->controls[1].value);
		}
		ko_adsr_set(&n->u.ss.adsr, s->controls[0].value, 0.25f, 1.0f, 0.5f);
	} break;
	case KO_PATCH_FM3: {                                                        /* examples/FM.k:36-55 */
		const float fc = f, fd = fc * s->controls[0].value;
		ko_operator* op = n->u.fm.op;
		ko_fsine_set_fp(&op[0].osc, fd, 0.f); { const float xy[4] = { 0, 0, 3, 1 }; ko_env_set_points(&op[0].env, 2, xy); op[0].env.mode = KO_ENV_TIME; }
		ko_fsine_set_fp(&op[1].osc, fd, 0.f); { const float xy[4] = { 0, 1.5f, 3, 0.5f }; ko_env_set_points(&op[1].env, 2, xy); }
		ko_fsine_set_fp(&op[2].osc, fc, 0.f);
		ko_adsr_set(&n->u.fm.adsr, s->controls[3].value, 0.1f, 1.f, 1.f);
	} break;
	case KO_PATCH_FM4: {                                                        /* oracle/ref/ref_fm.cpp FM4 */
		const float fc = f, fd = fc * s->controls[0].value;
		ko_operator* op = n->u.fm.op;
		ko_fsine_set_fp(&op[0].osc, fd, 0.f); { const float xy[4] = { 0, 0, 3, 1 }; ko_env_set_points(&op[0].env, 2, xy); }
		ko_fsine_set_fp(&op[1].osc, fd, 0.f); { const float xy[4] = { 0, 1.5f, 3, 0.5f }; ko_env_set_points(&op[1].env, 2, xy); }
		ko_fsine_set_fp(&op[2].osc, fd, 0.f); { const float xy[4] = { 0, 1, 2, 0.25f }; ko_env_set_points(&op[2].env, 2, xy); }
		ko_fsine_set_fp(&op[3].osc, fc, 0.f);
		ko_adsr_set(&n->u.fm.adsr, s->controls[4].value, 0.1f, 1.f, 1.f);
	} break;
	}
}

static void note_off(ko_bank* b, ko_note* n) {                  /* user off() */
	switch (b->patch) {
	case KO_PATCH_SINE: case KO_PATCH_BSINE: n->stage = ST_OFF; break;          /* off() { stop(); } */
	case KO_PATCH_SUB2A: ko_adsr_release(&n->u.s2a.adsr, 0.f, 0.f); break;
	case KO_PATCH_SUB2B: ko_adsr_release(&n->u.s2b.adsr, 0.f, 0.f); break;
	case KO_PATCH_SUPERSAW: ko_adsr_release(&n->u.ss.adsr, 0.f, 0.f); break;
	case KO_PATCH_FM3: case KO_PATCH_FM4: ko_adsr_release(&n->u.fm.adsr, 0.f, 0.f); break;
	}
}

int ko_bank_note_on(ko_bank* b, int synth, int pitch, float velocity, long seed) {   /* klang.h:4423-4427 + NoteBase::start 4257-4263 */
	ko_synth* s = &b->synths[synth];
	if (seed >= 0) srand((unsigned)seed);                       /* klang::random(seed) klang.h:239 */
	const int i = synth_assign(s);
	ko_note* n = &s->notes[i];
	n->stage = ST_ONSET;
	n->pitch = (float)pitch; n->velocity = velocity;
	note_on(b, s, n);
	n->stage = ST_SUSTAIN;
	return i;
}
void ko_bank_note_off(ko_bank* b, int synth, int pitch, float velocity) {            /* klang.h:4430-4434 + NoteBase::release 4265-4275 */
	(void)velocity;
	ko_synth* s = &b->synths[synth];
	for (int i = 0; i < s->count; i++) {
		ko_note* n = &s->notes[i];
		if (n->pitch == (float)pitch && n->stage == ST_SUSTAIN) {
			n->stage = ST_RELEASE;
			note_off(b, n);
		}
	}
}

/* user process() of each patch: one sample */
static inline float note_sample(ko_bank* b, ko_synth* s, ko_note* n) {
	switch (b->patch) {
	case KO_PATCH_SINE: n->out = ko_fsine_process(&n->u.sine.osc); break;
	case KO_PATCH_BSINE: n->out = ko_basic_sine(&n->u.bsine.osc); break;
	case KO_PATCH_SUB2A: {
		n->out = ko_biquad_process(&n->u.s2a.lpf, ko_osm_process(&n->u.s2a.osc));
		n->out *= ko_env_process(&n->u.s2a.adsr.env);
		if (n->u.s2a.adsr.env.stage == KO_ENV_OFF) n->stage = ST_OFF;
	} break;
	case KO_PATCH_SUB2B: {                                       /* subtractive.k:29-34 */
		const float fc = ko_env_process(&n->u.s2b.env);          /* filter(env++, 10) evaluated first */
		ko_biquad_set_fq(&n->u.s2b.filter, fc, 10.f);
		n->out = ko_biquad_process(&n->u.s2b.filter, ko_osm_process(&n->u.s2b.osc));
		n->out *= ko_env_process(&n->u.s2b.adsr.env);
		if (n->u.s2b.adsr.env.stage == KO_ENV_OFF) n->stage = ST_OFF;
	} break;
	case KO_PATCH_SUPERSAW: {                                    /* SuperSaw.k:25-33 */
		n->out = 0;
		for (int k = 0; k < 7; k++) n->out += ko_osm_process(&n->u.ss.osc[k]) / 7.f;
		n->out *= ko_env_process(&n->u.ss.adsr.env);
		if (n->u.ss.adsr.env.stage == KO_ENV_OFF) n->stage = ST_OFF;
	} break;
	case KO_PATCH_FM3: {                                         /* FM.k:62-73 */
		ko_operator* op = n->u.fm.op;
		op[0].amp = s->controls[1].value;
		const float m1 = ko_operator_process(&op[0]);
		op[1].amp = s->controls[2].value;
		op[1].in = m1;
		const float m2 = ko_operator_process(&op[1]);
		op[2].in = m2;
		n->out = ko_operator_process(&op[2]);
		n->out *= ko_env_process(&n->u.fm.adsr.env) * 0.1f;
		if (n->u.fm.adsr.env.stage == KO_ENV_OFF) n->stage = ST_OFF;
	} break;
	case KO_PATCH_FM4: {
		ko_operator* op = n->u.fm.op;
		op[0].amp = s->controls[1].value;
		const float m1 = ko_operator_process(&op[0]);
		op[1].amp = s->controls[2].value;
		op[1].in = m1;
		const float m2 = ko_operator_process(&op[1]);
		op[2].amp = s->controls[3].value;
		op[2].in = m2;
		const float m3 = ko_operator_process(&op[2]);
		op[3].in = m3;
		n->out = ko_operator_process(&op[3]);
		n->out *= ko_env_process(&n->u.fm.adsr.env) * 0.1f;
		if (n->u.fm.adsr.env.stage == KO_ENV_OFF) n->stage = ST_OFF;
	} break;
	}
	return n->out;
}

/* Synth::process voice loop (klang.h:4451-4458 / 4842-4848) with Note::process(buffer) (4295-4303 / 4747-4756):
 * a voice that stops mid-block keeps running process() until the block ends. */
void ko_bank_process(ko_bank* b, float* per_voice, float* mix, unsigned char* stages, int n) {
	const int V = b->S * b->P;
	const int accumulate = (b->patch == KO_PATCH_SUB2A || b->patch == KO_PATCH_SUB2B);   /* Stereo::Synth patches */
	for (int v = 0; v < V; v++) {
		ko_synth* s = &b->synths[v / b->P];
		ko_note* note = &s->notes[v % b->P];
		float* dst = per_voice ? per_voice + (size_t)v * n : NULL;
		if (note->stage != ST_OFF) {
			for (int i = 0; i < n; i++) {
				float y = note_sample(b, s, note);
				if (accumulate) y = 0.f + y;       /* Stereo::Mono::Note: `buffer.left += out` into a cleared buffer (-0 -> +0) */
				if (dst) dst[i] = y;
				if (mix) { mix[i] += y; mix[n + i] += y; }
			}
		}
		else if (dst) memset(dst, 0, sizeof(float) * (size_t)n);
		if (stages) stages[v] = (unsigned char)note->stage;
	}
}

/* =====================================================================
 * Scenario runner (file formats documented in oracle/ref/ref_common.h)
 * ===================================================================== */
typedef struct { int block, type, synth; float a, b; long seed; } ko_event;

#define fscanf(...) ((void)!fscanf(__VA_ARGS__))
int ko_run_scenario(const char* scenario_path, const char* out_path) {
	FILE* f = fopen(scenario_path, "r");
	if (!f) return 1;
	char tok[64], patch[64] = "";
	int ver = 0, block = 256, blocks = 1, synths = 1, notes = 1, ndump = 0, nctl = 0, nev = 0, cap = 0;
	int instances = 0, burst = 0; unsigned seed = 0;
	float fs = 48000.f;
	int* dump = NULL; int ctl_i[64]; float ctl_v[64]; ko_event* ev = NULL;
	if ((fscanf)(f, "%63s %d", tok, &ver) != 2 || strcmp(tok, "klgscn")) { fclose(f); return 1; }
	while ((fscanf)(f, "%63s", tok) == 1) {
		if (!strcmp(tok, "end")) break;
		else if (!strcmp(tok, "patch")) fscanf(f, "%63s", patch);
		else if (!strcmp(tok, "fs")) fscanf(f, "%f", &fs);
		else if (!strcmp(tok, "block")) fscanf(f, "%d", &block);
		else if (!strcmp(tok, "blocks")) fscanf(f, "%d", &blocks);
		else if (!strcmp(tok, "synths")) fscanf(f, "%d", &synths);
		else if (!strcmp(tok, "notes")) fscanf(f, "%d", &notes);
		else if (!strcmp(tok, "instances")) fscanf(f, "%d", &instances);
		else if (!strcmp(tok, "burst")) fscanf(f, "%d", &burst);
		else if (!strcmp(tok, "seed")) fscanf(f, "%u", &seed);
		else if (!strcmp(tok, "dump")) { fscanf(f, "%d", &ndump); dump = (int*)malloc(sizeof(int) * (size_t)(ndump + 1)); for (int i = 0; i < ndump; i++) fscanf(f, "%d", &dump[i]); }
		else if (!strcmp(tok, "ctl")) { fscanf(f, "%d %f", &ctl_i[nctl], &ctl_v[nctl]); nctl++; }
		else if (!strcmp(tok, "ev")) {
			if (nev == cap) { cap = cap ? cap * 2 : 256; ev = (ko_event*)realloc(ev, sizeof(ko_event) * (size_t)cap); }
			ko_event* e = &ev[nev++];
			fscanf(f, "%d %d %d %f %f %ld", &e->block, &e->type, &e->synth, &e->a, &e->b, &e->seed);
		}
		else { fclose(f); return 1; }
	}
	fclose(f);
	const int pid = ko_patch_from_name(patch);
	if (pid < 0) return 2;
	FILE* out = fopen(out_path, "wb");
	if (!out) return 3;
	const int N = block, B = blocks;

	if (pid >= KO_PATCH_PINGPONG) {
		const int K = instances;
		ko_fxbank* fx = ko_fxbank_create(pid, K, fs);
		if (!fx) { fclose(out); return 4; }
		for (int k = 0; k < K; k++) for (int c = 0; c < nctl; c++) ko_fxbank_control(fx, k, ctl_i[c], ctl_v[c]);
		const int hdr[5] = { 0x46474C4B, K, N, ndump, B };
		fwrite(hdr, sizeof(int), 5, out);
		float* io = (float*)malloc(sizeof(float) * (size_t)K * 2 * N);
		int evi = 0;
		for (int b = 0; b < B; b++) {
			for (; evi < nev && ev[evi].block <= b; evi++) if (ev[evi].type == 2) ko_fxbank_control(fx, ev[evi].synth, (int)ev[evi].a, ev[evi].b);
			for (int k = 0; k < K; k++) for (int ch = 0; ch < 2; ch++) for (int i = 0; i < N; i++)
				io[((size_t)k * 2 + ch) * N + i] = ko_fx_input(seed, (uint32_t)k, (uint32_t)ch, (uint32_t)(b * N + i), (uint32_t)burst);
			ko_fxbank_process(fx, io, N);
			for (int d = 0; d < ndump; d++) if (dump[d] == b) fwrite(io, sizeof(float), (size_t)K * 2 * N, out);
		}
		free(io); ko_fxbank_destroy(fx);
		fclose(out); free(dump); free(ev);
		return 0;
	}

	ko_bank* bank = ko_bank_create(pid, synths, notes, fs);
	if (!bank) { fclose(out); return 4; }
	for (int s = 0; s < synths; s++) for (int c = 0; c < nctl; c++) ko_bank_control(bank, s, ctl_i[c], ctl_v[c]);
	const int V = synths * notes;
	const int hdr[5] = { 0x4F474C4B, V, N, ndump, B };
	fwrite(hdr, sizeof(int), 5, out);
	float* voice = (float*)malloc(sizeof(float) * (size_t)V * N);
	float* mix = (float*)calloc((size_t)B * 2 * N, sizeof(float));
	unsigned char* stages = (unsigned char*)malloc((size_t)B * V);
	int evi = 0;
	for (int b = 0; b < B; b++) {
		for (; evi < nev && ev[evi].block <= b; evi++) {
			const ko_event* e = &ev[evi];
			if (e->type == 0) ko_bank_note_on(bank, e->synth, (int)e->a, e->b, e->seed);
			else if (e->type == 1) ko_bank_note_off(bank, e->synth, (int)e->a, e->b);
			else if (e->type == 2) ko_bank_control(bank, e->synth, (int)e->a, e->b);
		}
		ko_bank_process(bank, voice, mix + (size_t)b * 2 * N, stages + (size_t)b * V, N);
		for (int d = 0; d < ndump; d++) if (dump[d] == b) fwrite(voice, sizeof(float), (size_t)V * N, out);
	}
	fwrite(mix, sizeof(float), (size_t)B * 2 * N, out);
	fwrite(stages, 1, (size_t)B * V, out);
	fclose(out);
	free(voice); free(mix); free(stages); free(dump); free(ev);
	ko_bank_destroy(bank);
	return 0;
}

/* =====================================================================
 * Effects (Stereo::Effect instances) — implemented in klang_oracle_fx.c
 * ===================================================================== */
