/* oracle/klang_oracle_fx.c — TEST INFRASTRUCTURE ONLY.  See klang_oracle.h.
 *
 * CPU restatement of the two shipped effect patches of BASELINE config 4, operation by operation:
 *   examples/PingPong.k  (Stereo::Effect, klang::basic namespace)
 *   examples/Reverb.k    (Stereo::Effect, klang::optimised namespace)
 * Citations "PingPong.k:NN" / "Reverb.k:NN" are into /root/reference/examples/, "klang.h:NN" into the header.
 * The order of every process()/input()/set() call — including the ones C++'s implicit conversions trigger —
 * follows the DSL rules of SURVEY.md §8(a6) and is pinned bit-for-bit by tests/golden/pingpong_*.npz and
 * reverb_*.npz (produced by the genuine reference).
 */
#include "klang_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PI_F ((float)3.1415926535897932384626433832795)
#define ROOT2_F ((float)1.4142135623730950488016887242097)

static float random_f(float mn, float mx) { return rand() * ((mx - mn) / (float)RAND_MAX) + mn; }   /* klang.h:236 */

/* ---------------------------------------------------------------------------------------------
 * PingPong.k
 * --------------------------------------------------------------------------------------------- */
typedef struct {
	ko_control c[6];
	ko_delay left, right;                 /* Delay<192000> */
	ko_osc lfo;                           /* Basic::Sine */
	ko_biquad dc[2];                      /* Biquad::HPF */
	float delay;                          /* param delay (member) */
} ko_pingpong;

static int pingpong_init(ko_pingpong* p) {
	memset(p, 0, sizeof(*p));
	/* PingPong.k:14-21 ; Dial(name, min, max, initial) klang.h:1797 — `smoothed` starts at 0 */
	const float d[6][3] = { { 0.0f, 0.999f, 0.5f }, { 0.001f, 1.0f, 0.5f }, { 0.0f, 1.0f, 0.0f }, { 0.01f, 1.0f, 0.0f }, { 0.001f, 2.0f, 1.0f }, { 0.f, 1.f, 0.f } };
	for (int i = 0; i < 6; i++) { p->c[i].min = d[i][0]; p->c[i].max = d[i][1]; p->c[i].value = d[i][2]; p->c[i].smoothed = 0.f; }
	if (ko_delay_create(&p->left, 192000) || ko_delay_create(&p->right, 192000)) return -1;
	ko_osc_init(&p->lfo);
	ko_biquad_init(&p->dc[0], KO_BQ_HPF); ko_biquad_init(&p->dc[1], KO_BQ_HPF);
	return 0;
}

static void pingpong_block(ko_pingpong* p, float* L, float* R, int n) {
	/* prepare() PingPong.k:37-41 */
	ko_biquad_set_fq(&p->dc[0], 50.f, 1.f);
	ko_biquad_set_fq(&p->dc[1], 50.f, 1.f);
	for (int i = 0; i < n; i++) {                                         /* Stereo::Effect::process klang.h:4708-4716 */
		const float in_l = L[i], in_r = R[i];
		/* process() PingPong.k:44-71 */
		const float rate = (p->c[3].value * p->c[3].value) * 100.f;
		const float new_delay = ko_control_smooth(&p->c[5]);
		if ((double)fabsf(p->delay - new_delay) > 0.001) {
			p->delay = new_delay;
			ko_control_set(&p->c[1], new_delay);
			ko_osc_set_fp(&p->lfo, rate, PI_F);
		}
		else {
			p->delay = p->c[5].value;
			ko_osc_set_f(&p->lfo, rate);
		}
		const float gain = p->c[0].value;
		const float delay = ko_control_smooth(&p->c[1]);
		const float vibrato = (p->c[2].value * p->c[2].value) * rate * ROOT2_F;
		const float dry = p->c[4].value;
		ko_control_set(&p->c[1], p->c[1].value + ko_basic_sine(&p->lfo) * vibrato * (float)0.00005);

		ko_delay_set(&p->left, delay * ko_fs.f);
		ko_delay_set(&p->right, 0.5f * delay * ko_fs.f);

		/* dry * in.l + (1.f - dry) * ((in.l + right * gain) >> left) >> out.l; */
		const float r1 = ko_delay_process(&p->right);
		ko_delay_input(&p->left, in_l + r1 * gain);
		const float l1 = ko_delay_process(&p->left);
		float out_l = dry * in_l + l1 * (1.f - dry);
		/* dry * in.r + (1.f - dry) * ((in.r + left * gain) >> right) >> out.r; */
		const float l2 = ko_delay_process(&p->left);
		ko_delay_input(&p->right, in_r + l2 * gain);
		const float r2 = ko_delay_process(&p->right);
		float out_r = dry * in_r + r2 * (1.f - dry);

		out_l = ko_biquad_process(&p->dc[0], out_l);
		out_r = ko_biquad_process(&p->dc[1], out_r);
		L[i] = out_l; R[i] = out_r;
	}
}

/* ---------------------------------------------------------------------------------------------
 * Reverb.k
 * --------------------------------------------------------------------------------------------- */
typedef struct { ko_delay delay; ko_biquad filter; float gain, in, out; } ko_fdelay;          /* FilteredDelay Reverb.k:118-134 */
typedef struct { ko_fdelay d[4]; float in, out; } ko_late;                                     /* LateReflections Reverb.k:117-175 */
typedef struct {
	ko_delay dl, dr;                                                                          /* Stereo::Delay<21600> */
	int count; float times[20], gl[20], gr[20];
	float length, size;
	ko_biquad lpf[2], hpf[2];
} ko_early;
typedef struct {
	ko_control c[10]; float cache[10];                                                        /* Controls::value[] klang.h:1878 */
	ko_early early; ko_late mid[2], late[2];
} ko_reverb;

static int reverb_init(ko_reverb* r) {
	memset(r, 0, sizeof(*r));
	/* Reverb.k:100-113 */
	const float d[10][3] = { { 0, 1, 0 }, { 0, 1, 1 }, { 0, 1, 0 }, { 0, 1, 0 }, { 0, 1, 1 }, { 0, 100, 10 }, { 0, 1, 1 }, { 0.01f, 1, 1 }, { 0.01f, 1, 1 }, { 0, 0.2f, 0 } };
	for (int i = 0; i < 10; i++) { r->c[i].min = d[i][0]; r->c[i].max = d[i][1]; r->c[i].value = d[i][2]; }
	if (ko_delay_create(&r->early.dl, 21600) || ko_delay_create(&r->early.dr, 21600)) return -1;
	for (int k = 0; k < 2; k++) { ko_biquad_init(&r->early.lpf[k], KO_BQ_LPF); ko_biquad_init(&r->early.hpf[k], KO_BQ_HPF); }
	ko_late* lr[4] = { &r->mid[0], &r->mid[1], &r->late[0], &r->late[1] };
	for (int a = 0; a < 4; a++) for (int k = 0; k < 4; k++) {
		if (ko_delay_create(&lr[a]->d[k].delay, 192000)) return -1;
		ko_biquad_init(&lr[a]->d[k].filter, KO_BQ_LPF);
	}
	return 0;
}

static void early_update(ko_early* e) {                                   /* Reverb.k:23-52 */
	static const float primes[20] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71 };
	e->count = 10 + (int)(e->size * (float)10.999);
	const float scale = 50.f / primes[e->count - 1];
	const float ms = ko_fs.f / 1000.f;
	for (int r = 0; r < e->count; r++) {
		e->times[r] = ((50.f + primes[r] * scale) * ms * random_f(0.9f, 1.1f));
		const float x = (float)(r + 1.f) / (float)(unsigned)e->count;
		const float g = random_f(0.5f, 1.5f) * expf(-3.f * x);
		const float pan = random_f(0.f, 1.f);
		e->gl[r] = g * (1.f - pan);
		e->gr[r] = g * pan;
	}
}
static void early_set(ko_early* e, float length, float size) {            /* Reverb.k:63-73 */
	length *= 1 / 1000.f;
	if (e->length != length || e->size != size) {
		e->length = length; e->size = size;
		early_update(e);
		ko_biquad_set_f(&e->hpf[0], 100.f); ko_biquad_set_f(&e->hpf[1], 100.f);
		ko_biquad_set_f(&e->lpf[0], 15000.f); ko_biquad_set_f(&e->lpf[1], 15000.f);
	}
}
static void fdelay_set(ko_fdelay* f, float time, float cutoff, float gain) {   /* Reverb.k:123-127 */
	ko_delay_set(&f->delay, time * ko_fs.f / 1000.f);
	ko_biquad_set_f(&f->filter, cutoff);
	f->gain = gain;
}
static void late_set(ko_late* l, const float* delays, float dampening, float gain) {   /* Reverb.k:138-144 */
	for (int k = 0; k < 4; k++) fdelay_set(&l->d[k], delays[k] * random_f(.9f, 1.1f), dampening, gain);
}
static void reflections_set(ko_reverb* r, float length, float size, float dampening1, float dampening2) {   /* Reverb.k:188-214 */
	early_set(&r->early, (length / 10.f) * 1000.f + 50.f, size);
	dampening1 *= 10000.f;
	dampening2 *= dampening1;
	const float delays1[4] = { 7, 11, 13, 17 };
	late_set(&r->mid[0], delays1, dampening1, 0.25f);
	late_set(&r->mid[1], delays1, dampening1, 0.25f);
	const float delays2[4] = { 19, 23, 29, 31 };
	late_set(&r->late[0], delays2, dampening2, 0.35f);
	late_set(&r->late[1], delays2, dampening2, 0.35f);
}

static float fdelay_process(ko_fdelay* f) {                                /* (in >> delay >> filter) * gain >> out  Reverb.k:130-132 */
	ko_delay_input(&f->delay, f->in);
	const float t = ko_delay_process(&f->delay);
	f->out = ko_biquad_process(&f->filter, t) * f->gain;
	return f->out;
}
static float late_process(ko_late* l, float in) {                          /* Reverb.k:153-168 */
	static const float M[16] = { 0, 1, 1, -1,  -1, 0, -1, 1,  -1, 1, 0, -1,  1, -1, 1, 0 };
	l->in = in;
	float dl[4], fb[4];
	for (int k = 0; k < 4; k++) dl[k] = fdelay_process(&l->d[k]);           /* signals<4> delays = { delay[0..3] } : conversion processes each */
	ko_matrix_mul(M, dl, fb);
	for (int k = 0; k < 4; k++) { fb[k] += in; l->d[k].in = fb[k]; }        /* fb = (delays >> matrix) + in ; fb[k] >> delay[k] */
	const float o0 = fdelay_process(&l->d[0]);                              /* the `+` chain processes each FilteredDelay a second time */
	const float o1 = fdelay_process(&l->d[1]);
	const float s01 = o0 + o1;
	const float s012 = fdelay_process(&l->d[2]) + s01;
	l->out = fdelay_process(&l->d[3]) + s012;
	return l->out;
}
static void early_process(ko_early* e, float in_l, float in_r, float* ol, float* or_) {    /* Reverb.k:87-93 */
	const float l = ko_biquad_process(&e->hpf[0], ko_biquad_process(&e->lpf[0], in_l));
	const float r = ko_biquad_process(&e->hpf[1], ko_biquad_process(&e->lpf[1], in_r));
	ko_delay_input(&e->dl, l);
	ko_delay_input(&e->dr, r);
	float sl = 0.f, sr = 0.f;
	for (int d = 0; d < e->count; d++) {
		float tl, tr;
		ko_stereo_delay_tap_float(&e->dl, &e->dr, e->times[d], &tl, &tr);
		sl += tl * e->gl[d];
		sr += tr * e->gr[d];
	}
	*ol = sl; *or_ = sr;
}

static void reverb_block(ko_reverb* r, float* L, float* R, int n) {
	/* prepare() Reverb.k:237-241 : Controls::changed() klang.h:1914-1923 */
	int changed = 0;
	for (int c = 0; c < 10; c++) if (r->c[c].value != r->cache[c]) { r->cache[c] = r->c[c].value; changed = 1; }
	if (changed) {
		srand(272839);
		reflections_set(r, r->c[5].value, r->c[6].value, r->c[7].value, r->c[8].value);
	}
	for (int i = 0; i < n; i++) {
		const float in_l = L[i], in_r = R[i];
		const float dry = r->c[0].value, wet = r->c[4].value;
		/* Reflections::process Reverb.k:223-245 */
		float r1l, r1r;
		early_process(&r->early, in_l, in_r, &r1l, &r1r);
		const float r2l = late_process(&r->mid[0], r1l);
		const float r2r = late_process(&r->mid[1], r1r);
		const float r3l = late_process(&r->late[0], r2l);
		const float r3r = late_process(&r->late[1], r2r);
		const float c1 = r->c[1].value, c2 = r->c[2].value, c3 = r->c[3].value;
		const float refl_l = (r1l * c1 + r2l * c2) + r3l * c3;
		const float refl_r = (r1r * c1 + r2r * c2) + r3r * c3;
		/* (in * dry + (in >> reflections) * wet) >> out  Reverb.k:271
		 * `reflections * wet` is Output<signals<2>>::operator*(TYPE&) = out * SIGNAL(other) (klang.h:2220) and
		 * SIGNAL(other) = signals<2>(const param&) selects the variadic `signals(Args&... initial) : value{ initial... }`
		 * constructor (klang.h:1240-1241) with ONE argument: value = { wet, 0 }.  The right channel of the wet path is
		 * therefore multiplied by 0 in the reference (verified by running it); reproduced here. */
		L[i] = in_l * dry + refl_l * wet;
		R[i] = in_r * dry + refl_r * 0.f;
	}
}

/* ---------------------------------------------------------------------------------------------
 * bank API
 * --------------------------------------------------------------------------------------------- */
struct ko_fxbank { int patch, K; ko_pingpong* pp; ko_reverb* rv; };

ko_fxbank* ko_fxbank_create(int patch, int instances, float fs) {
	if (patch != KO_PATCH_PINGPONG && patch != KO_PATCH_REVERB) return NULL;
	ko_set_fs(fs);
	ko_fxbank* b = (ko_fxbank*)calloc(1, sizeof(*b));
	b->patch = patch; b->K = instances;
	if (patch == KO_PATCH_PINGPONG) {
		b->pp = (ko_pingpong*)calloc((size_t)instances, sizeof(ko_pingpong));
		for (int k = 0; k < instances; k++) if (pingpong_init(&b->pp[k])) return NULL;
	}
	else {
		b->rv = (ko_reverb*)calloc((size_t)instances, sizeof(ko_reverb));
		for (int k = 0; k < instances; k++) if (reverb_init(&b->rv[k])) return NULL;
	}
	return b;
}
void ko_fxbank_destroy(ko_fxbank* b) {
	if (!b) return;
	for (int k = 0; k < b->K; k++) {
		if (b->pp) { ko_delay_destroy(&b->pp[k].left); ko_delay_destroy(&b->pp[k].right); }
		if (b->rv) {
			ko_delay_destroy(&b->rv[k].early.dl); ko_delay_destroy(&b->rv[k].early.dr);
			ko_late* lr[4] = { &b->rv[k].mid[0], &b->rv[k].mid[1], &b->rv[k].late[0], &b->rv[k].late[1] };
			for (int a = 0; a < 4; a++) for (int j = 0; j < 4; j++) ko_delay_destroy(&lr[a]->d[j].delay);
		}
	}
	free(b->pp); free(b->rv); free(b);
}
void ko_fxbank_control(ko_fxbank* b, int instance, int index, float value) {
	if (b->pp && index < 6) ko_control_set(&b->pp[instance].c[index], value);
	if (b->rv && index < 10) ko_control_set(&b->rv[instance].c[index], value);
}
void ko_fxbank_process(ko_fxbank* b, float* io, int n) {
	for (int k = 0; k < b->K; k++) {
		float* L = io + ((size_t)k * 2 + 0) * n;
		float* R = io + ((size_t)k * 2 + 1) * n;
		if (b->pp) pingpong_block(&b->pp[k], L, R, n);
		else reverb_block(&b->rv[k], L, R, n);
	}
}
