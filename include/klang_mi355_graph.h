/* include/klang_mi355_graph.h — "graph patches": the per-sample body of a user Note::process(), recorded as a
 * straight-line program over klang's primitives and compiled for gfx950 at run time.
 *
 * The five shipped patch ids of klang_mi355.h are hand-written kernels.  A graph patch removes that limit for
 * synth notes whose process() is a feed-forward chain of the primitives below: the DSL façade
 * (include/klang/klang.h) runs the user's process() ONCE in recording mode — every `>>`, `++`, arithmetic
 * operator and set() call appends an op instead of computing — and hands the program text to
 * klg_synth_create_graph(), which generates a patch body on top of the same device primitives
 * (klang_amd/csrc/klg_device.hpp) and the same render kernel (klg_render<P>) the shipped patches use, compiles it
 * with hipRTC for gfx950 and loads it.  on()/off() keep running on the host as written; voice state moves with
 * klg_voice_download / klg_voice_upload in the record layout defined here.
 *
 * Program text (one statement per line, '#' starts a comment):
 *     klgg 1
 *     kind effect <channels>              optional: the body of an Effect::process() (1 = klang::Effect, 2 = Stereo::Effect) instead
 *                                         of a Note's; one lane per effect instance, input samples via `in`, output via ret / ret2
 *     ctl <count>                         number of controls of the synth / effect (<= 32)
 *     dial <i> <min> <max> <initial>      Dial(...) of control i              klang.h:1797-1800
 *     node <id> <kind> [size]             a primitive object of the Note / Effect, ids 0,1,2,... in order (size: Delay<SIZE>)
 *     op <code> <dst> <a> <b> <node> <imm>   one op; unused fields are -1; imm = IEEE-754 bits (hex) of a constant
 *     prepare <n>                         optional: the first n ops are Effect::prepare() / Note::prepare() (klang.h:4208-4211) — they run once per
 *                                         block per instance, before the samples; their registers are not visible to the sample ops
 *     ret <reg>                           the register holding `out` at the end of process()   (ret2 <l> <r> for a Stereo::Effect, and for a
 *                                         note whose `out` is a stereo signal — Stereo::Note, klang.h:4721-4733: `buffer++ += out` adds out.l to
 *                                         the left and out.r to the right channel; such a bank renders two samples per voice and sample)
 *     end
 * Data-dependent branches of process() (`if (in > 1) in = 1;`, `if (osc.frequency < fs.nyquist) out += osc / h;`) are
 * structured ops: `cmp` makes a 1.0 / 0.0 register, `if <a>` ... `else` ... `endif` brackets the two sides (registers
 * assigned inside a side are not visible outside it), and the `phi` ops directly after an `endif` merge values:
 * `phi dst a b` = a (a register visible at the end of the `if` side) when the condition held, else b (visible at the end of
 * the `else` side).  The façade finds the sides by running process() once per branch outcome and merging the traces.
 * Registers are single-assignment fp32 values.  Record layout: word 0 = flags (bits 0-1 NoteBase::stage), then the
 * words of node 0, node 1, ... in order (node_words()).
 */
#ifndef KLANG_MI355_GRAPH_H
#define KLANG_MI355_GRAPH_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

namespace klg { namespace graph {

/* node kinds and their record words */
enum NodeKind {
	N_FSINE = 0,    /* Generators::Fast::Sine          klang.h:5135-5172   words: inc, pos, frequency (the set(f) cache) */
	N_SAW = 1,      /* Fast::OSM, saw family (Saw, Triangle)   5175-5354   words: inc, offset, duty, delta, state, frequency */
	N_PULSE = 2,    /* Fast::OSM, pulse family (Square, Pulse)             same words */
	N_LPF = 3,      /* Filters::Biquad::{LPF,HPF,BPF,BRF,APF}  5550-5773   words: b0 b1 b2 a1 a2 z0 z1 f Q  (process() is type-independent; only LPF may be set() per sample) */
	N_ENV = 4,      /* Envelope                         3867-4102           words: r_out r_target r_rate time bits npoints loop px[4] py[4], then — when the node's
	                                                                        argument (its POINT CAPACITY, 5 .. ENV_MAX_POINTS; none = 4) says so — x of points 4 .. cap-1 and
	                                                                        their y.  The first four points sit in registers; the others are read from the record when (and only
	                                                                        when) a segment ends.  bits: stage(2) | point & 7 (3) | ramp active (1) | Rate mode (1: setMode,
	                                                                        klang.h:4064-4092) | point >> 3 (from bit 7) */
	N_ADSR = 5,     /* ADSR                             4105-4137           words: r_out r_target r_rate time bits A AD S R */
	N_PARAM = 6,    /* a signal / param member of the Note that process() reads (and may write)   words: value */
	N_BSINE = 7, N_BSAW, N_BTRI, N_BSQUARE, N_BPULSE,   /* Generators::Basic::*   2849-2880, 4899-4944   words: increment, position, offset, duty, frequency */
	N_OPLPF = 12, N_OPHPF,   /* Filters::OnePole::LPF / HPF   5470-5543           words: b0 b1 a1 z out */
	N_DCF = 14,     /* Filters::DCF                     5386-5397           words: r z out */
	N_IIR1 = 15,    /* Filters::IIR<1>                  5434-5447           words: a b out */
	N_BUTTER1 = 16, /* Filters::Butterworth::LPF<1>     5786-5799           words: b0 a1 z out */
	N_MODAL = 17,   /* Modifiers::Modal                 5815-5859           words: a1 a2 y1 y2 gain */
	N_FOLLOWPEAK = 18, N_FOLLOWRMS,   /* Envelope::Follower (Peak / RMS)  5862-5903   words: A R out */
	N_OPERATOR = 20,   /* Operator<Fast::Sine>          4140-4180           words: inc pos frequency amp + the N_ENV words of its envelope */
	N_DELAY = 21,   /* Delay<SIZE> (effects only)       3381-3512           words: last.position, last.fraction (the read head that set() places and
	                                                                        every process() advances) — the SIZE floats live in a ring per instance in HBM,
	                                                                        position-major over the 64 instances of a wave; the write cursor is the sample counter */
	N_SMOOTH = 22,  /* controls[i].smooth()                1715             words: smoothed (an effect instance's own; a note's is set per block: the Synth's notes share the control) */
	N_WAVETABLE = 23,  /* Wavetable / Sample (synth notes)  3626-3720        words: increment position offset frequency table — `table` is the id
	                                                                        klg_table_upload() returned for this note's samples (HBM; identical tables share an id) */
	N_NDELAY = 24,  /* Delay<SIZE> member of a NOTE (physical models: a delay line per voice)  3381-3512   words: position (write cursor),
	                                                                        last.position, last.fraction (the read head of Delay::process, set by set()), time;
	                                                                        the SIZE floats of each voice's line are contiguous in HBM (voices' cursors never line up) */
	N_IIRN = 25,    /* Filters::IIR<ORDER>, ORDER 2..8  5399-5432           words: a[ORDER] then y[ORDER] (the node's argument is ORDER) */
	N_CTLVAR = 26,  /* a control an EFFECT writes: controls[i].set(x) inside process() (Control::set 1725-1728; PingPong.k:48,60)
	                                                                        words: value — the instance's own copy of the control: every read of controls[i]
	                                                                        (and its smooth()) in the program takes it; klg_fx_set_control overwrites it */
	N_KINDS
};
enum { FSINE_INC = 0, FSINE_POS, FSINE_FREQ, FSINE_WORDS };
enum { OSM_INC = 0, OSM_OFFSET, OSM_DUTY, OSM_DELTA, OSM_STATE, OSM_FREQ, OSM_WORDS };
enum { LPF_B0 = 0, LPF_B1, LPF_B2, LPF_A1, LPF_A2, LPF_Z0, LPF_Z1, LPF_F, LPF_Q, LPF_WORDS };
enum { ENV_OUT = 0, ENV_TARGET, ENV_RATE, ENV_TIME, ENV_BITS, ENV_NPOINTS, ENV_LOOP /* start | end << 8, 0xFF = none (setLoop) */, ENV_PX, ENV_PY = ENV_PX + 4, ENV_WORDS = ENV_PY + 4 /* + env_ext_words(capacity) */ };
enum { ENV_MAX_POINTS = 128, ENV_BIT_RATE = 1u << 6 };   /* (loop indices are bytes with 0xFF = none; the record keeps x and y of every point slot) */
inline int env_capacity(int arg) { return arg > 4 ? arg : 4; }                 /* an Envelope / Operator node's argument: how many points its record holds */
inline int env_ext_words(int arg) { return 2 * (env_capacity(arg) - 4); }      /* x of points 4.., then their y, behind the ENV_WORDS */
inline uint32_t env_bits(int stage, int point, bool active, bool rate) { return (uint32_t)stage | ((uint32_t)(point & 7) << 2) | ((uint32_t)active << 5) | (rate ? (uint32_t)ENV_BIT_RATE : 0u) | ((uint32_t)(point >> 3) << 7); }
inline int env_bits_point(uint32_t b) { return (int)(((b >> 2) & 7u) | ((b >> 7) << 3)); }
enum { ADSR_OUT = 0, ADSR_TARGET, ADSR_RATE, ADSR_TIME, ADSR_BITS, ADSR_A, ADSR_AD, ADSR_S, ADSR_R, ADSR_WORDS };
enum { BOSC_INC = 0, BOSC_POS, BOSC_OFFSET, BOSC_DUTY, BOSC_FREQ, BOSC_WORDS };
enum { OP1_B0 = 0, OP1_B1, OP1_A1, OP1_Z, OP1_OUT, OP1_WORDS };
enum { DCF_R = 0, DCF_Z, DCF_OUT, DCF_WORDS };
enum { IIR1_A = 0, IIR1_B, IIR1_OUT, IIR1_WORDS };
enum { BW1_B0 = 0, BW1_A1, BW1_Z, BW1_OUT, BW1_WORDS };
enum { MODAL_A1 = 0, MODAL_A2, MODAL_Y1, MODAL_Y2, MODAL_GAIN, MODAL_WORDS };
enum { FOLLOW_A = 0, FOLLOW_R, FOLLOW_OUT, FOLLOW_WORDS };
enum { OPER_INC = 0, OPER_POS, OPER_FREQ, OPER_AMP, OPER_ENV, OPER_WORDS = OPER_ENV + ENV_WORDS };
enum { WT_INC = 0, WT_POS, WT_OFFSET, WT_FREQ, WT_TABLE, WT_WORDS };
enum { ND_POS = 0, ND_LASTPOS, ND_LASTFRAC, ND_TIME, ND_WORDS };
enum { ED_LASTPOS = 0, ED_LASTFRAC, ED_WORDS };   /* an effect's Delay: the read head of set() / process() (Delay::last klang.h:3388) */
enum { MAX_WORDS = 2048, MAX_NODES = 256, MAX_OPS = 16384 };   // (round 3: the shipped Reverb.k records — 16 FilteredDelays, 20 stereo taps — at ~330 words / ~3 k ops)

inline bool is_oscillator(int k) { return k == N_FSINE || k == N_SAW || k == N_PULSE || (k >= N_BSINE && k <= N_BPULSE) || k == N_WAVETABLE; }
inline bool is_modifier(int k) { return k == N_LPF || (k >= N_OPLPF && k <= N_FOLLOWRMS) || k == N_IIRN; }
inline int node_words(int kind, int arg = 0) {
	switch (kind) {
	case N_FSINE: return FSINE_WORDS;
	case N_SAW: case N_PULSE: return OSM_WORDS;
	case N_LPF: return LPF_WORDS;
	case N_ENV: return ENV_WORDS + env_ext_words(arg);
	case N_ADSR: return ADSR_WORDS;
	case N_PARAM: return 1;
	case N_BSINE: case N_BSAW: case N_BTRI: case N_BSQUARE: case N_BPULSE: return BOSC_WORDS;
	case N_OPLPF: case N_OPHPF: return OP1_WORDS;
	case N_DCF: return DCF_WORDS;
	case N_IIR1: return IIR1_WORDS;
	case N_BUTTER1: return BW1_WORDS;
	case N_MODAL: return MODAL_WORDS;
	case N_FOLLOWPEAK: case N_FOLLOWRMS: return FOLLOW_WORDS;
	case N_OPERATOR: return OPER_WORDS + env_ext_words(arg);
	case N_DELAY: return ED_WORDS;
	case N_SMOOTH: return 1;
	case N_WAVETABLE: return WT_WORDS;
	case N_NDELAY: return ND_WORDS;
	case N_IIRN: return 2 * arg;
	case N_CTLVAR: return 1;
	}
	return 0;
}
/* the argument a node of this kind keeps (0: none).  An Envelope / Operator with the four built-in point slots has none, so that its program text is what it always was */
inline int node_arg_of(int kind, int arg) {
	if (kind == N_DELAY || kind == N_NDELAY || kind == N_IIRN) return arg;
	if (kind == N_ENV || kind == N_OPERATOR) return arg > 4 ? arg : 0;
	return 0;
}
inline const char* node_name(int kind) {
	static const char* names[N_KINDS] = { "fsine", "saw", "pulse", "lpf", "env", "adsr", "param", "bsine", "bsaw", "btri", "bsquare", "bpulse",
	                                      "oplpf", "ophpf", "dcf", "iir1", "butter1", "modal", "followpeak", "followrms", "operator", "delay", "smooth", "wavetable", "notedelay", "iirn", "ctlvar" };
	return (kind >= 0 && kind < N_KINDS) ? names[kind] : "?";
}

/* ops: dst = code(a, b, node, imm) */
enum OpCode {
	OP_CONST = 0,   /* dst = imm                                                                      */
	OP_CTL,         /* dst = controls[imm]                 (the synth instance's control value)      */
	OP_PARAM,       /* dst = value word of N_PARAM node                                               */
	OP_OSC,         /* dst = oscillator node process()     Fast::Sine / OSM saw / OSM pulse / Basic::*   */
	OP_OSCSET,      /* oscillator node .set(f = a)         Fast::Sine::set 5142-5147 / OSM::set 5217-5224 / Oscillator::set 2862 (per sample: vibrato, FM);
	                   imm 1: .set(f = a, phase = b) 5149-5153 / 5226-5234 / 2867-2870 (hard sync, re-phasing); imm 2: .reset() 5136-5140 / 2859;
	                   imm 3: duty = a — OSM::setDuty 5246-5249 / Basic::Pulse::duty 4936: `set(f, phase, duty)` per sample (PWM) is oscset 1 followed by oscset 3 */
	OP_LPF,         /* dst = (a >> modifier node)          any modifier kind: Biquad::Filter::process 5605-5612, OnePole, DCF, IIR<1>, ... */
	OP_LPFSET,      /* lpf node .set(f = a, Q = b)         Biquad::Filter::set klang.h:5575-5600 + the init() of type imm (0 LPF, 1 HPF, 2 / 3 BPF peak / skirt, 4 BRF, 6 Butterworth<2>) */
	OP_ENV,         /* dst = env/adsr node ++              Envelope::operator++ klang.h:4013-4051     */
	OP_ADD, OP_SUB, OP_MUL, OP_DIV,   /* dst = a op b      fp32, IEEE, no contraction                 */
	OP_NEG,         /* dst = -a                                                                       */
	OP_STOPIF,      /* if (env/adsr node .finished()) stop();   klang.h:4094, 4276-4279               */
	OP_STOP,        /* stop();                                                                        */
	OP_SETPARAM,    /* N_PARAM node = a                    (a member written by process(): next sample reads it) */
	OP_FREQ,        /* dst = oscillator node .frequency    (Oscillator::frequency klang.h:2856, as last set by on() or by oscset) */
	OP_IN,          /* dst = this sample of input channel imm            (effects: `in`, `in.l`, `in.r`)                       */
	OP_DELAYIN,     /* a >> delay node                                  Delay::input klang.h:3396-3403                        */
	OP_DELAYTAP,    /* dst = delay node (a)                             Delay::tap(float) klang.h:3412-3427; imm 1: tap(int) 3405-3410 (a holds the integer); imm 2: one channel of Stereo::Delay::tap(float) 4668-4681; imm 3: lagrange(float) 3429-3458 (effects) */
	OP_SMOOTH,      /* dst = smooth node: controls[imm].smooth()        Control::smooth klang.h:1715                          */
	OP_OPERATOR,    /* dst = operator node process()       modulator a (or -1: none), amp b (or -1: keep)   Operator::process klang.h:4164-4168 */
	OP_CMP,         /* dst = (a REL b) ? 1.0 : 0.0         imm = relation: 0 <, 1 >, 2 <=, 3 >=, 4 ==, 5 !=  (IEEE: false on NaN except !=) */
	OP_IF,          /* if (a != 0) {                       structured; the sides are scopes                   */
	OP_ELSE,        /* } else {                                                                               */
	OP_ENDIF,       /* }                                                                                      */
	OP_PHI,         /* dst = condition of the `if` just closed ? a : b       only directly after `endif` (or another phi of it) */
	OP_NOISE,       /* dst = Noise process()               imm 0: Generators::Basic::Noise klang.h:4947-4951, 1: Fast::Noise 5357-5366.  Both draw
	                   from libc rand(): the bank draws the values on the host, per block, in the order the reference would call rand() with
	                   its instances in one process (instance-major, then sample, then the ops in program order).  Effects only, never inside an `if` */
	OP_DELAYOUT,    /* dst = delay node process()          Delay::process 3470-3473: the read head set by set() (note delays)   */
	OP_TABREAD,     /* dst = table imm [ a ]               Table<float, SIZE>::operator[](float): clamped, linear   klang.h:3365-3377; imm = table id (klg_table_upload) */
	OP_SETCTL,      /* dst = ctlvar node = clamp(a)        controls[imm & 0xFF].set(a): (a < min) ? min : (max < a) ? max : a, the dial's range   Control::set klang.h:1725-1728 (effects);
	                   imm bit 8: `x >> controls[i]` / `controls[i] << x` — the plain assignment of Control::operator<< klang.h:1745-1746 (a METER fed by process(): Vocoder.k:104), no clamp */
	OP_DELAYSET,    /* delay node .set(samples = a)        Delay::set 3480-3489: the read head `samples` behind the write cursor (in process(), or in an
	                   effect's prepare(): placed once per block, then walked by every `delay >> x`) */
	OP_ABS,         /* dst = |a|                           std::abs of a signal: fabsf                                                               */
	/* DOUBLE registers: what `(controls[1] + 0.01232 * c) * fs` is in the reference (examples/Delay/Reverb2.k:43: Control -> float, float + double, double * float
	 * -> double, then Delay::operator()(double) -> tap((float)x)).  A register is a double iff one of these defines it; only these and d2f read one; no phi */
	OP_F2D,         /* dst(double) = (double) a                                                                                                        */
	OP_DCONST,      /* dst(double) = the double whose HIGH word is imm (low word 0)                                                                    */
	OP_DLOW,        /* dst(double) = a with its LOW word replaced by imm: a 64-bit literal is dconst + dlow (an op carries 32 immediate bits)          */
	OP_DADD, OP_DSUB, OP_DMUL, OP_DDIV,   /* dst(double) = a OP b, both doubles                                                                          */
	OP_D2F,         /* dst = (float) a   (round to nearest even)                                                                                       */
	OP_ENVOFF,      /* dst = env/adsr node .finished() ? 1.0 : 0.0     Envelope::finished klang.h:4094 (stage == Off) as a VALUE: `if (adsr.finished()) { ...; stop(); return; }`,
	                   `!adsr.finished()`, `finished() && x > y` — the recorder turns the plain `if (env.finished()) stop();` back into stopif.
	                   imm: the stage asked for — 0 Off (finished()), 1 Sustain, 2 Release (`env == Envelope::Release`, klang.h:3883-3884)                   */
	OP_FUNC,        /* dst(double) = f(a), a a double, f by imm: 0 = tanh — what `tanh(x)` of a float is inside a patch's plain C function: the C library's DOUBLE tanh of the
	                   converted float, and the expression around it stays double (`tanh(c * x) / tanh(c)`, examples/Distortion/Shaping.k:15: f2d, func, ddiv, d2f).
	                   klg_device.hpp glibc_tanh restates glibc 2.35's (float-rounded result equal on all 2^32 floats: tools/verify_tanh_f64.c).
	                   imm & 0xFF: 1 = exp2(a), 2 = pow(B, a) with the constant base B = the float whose pattern is imm & 0xFFFFFF00 (positive, normal) — the C library's DOUBLE
	                   functions, what `pow(2, (signal)osc)` (the pinned compiler calls exp2) and `pow(10, 2 * (x - 1))` of examples/Subtractive/Modular.k:24, 123 are in the
	                   reference build: klg_glibc_pow.hpp (bit-equal as doubles: tools/verify_glibc_pow.cpp) */
	OP_POWC,        /* dst = power(a, e), e the float whose bits are imm, one of 0, +-1 .. +-4: klang's `power(float base, float exp)` with a literal exponent (klang.h:188-218;
	                   Vocoder.k:83 `power(1.f - x, 2.f)`): base == 10 ? (float)exp(e * ln 10) : the product / quotient of bases the reference writes out for these exponents.
	                   (Any other exponent is the C library's powf and is refused by the recorder.) */
	OP_TRUNC,       /* dst = (float)(int)a                 a float forced through an int: klang's `min(20000, x)` returns its FIRST type (klang.h:223; Modular.k:155).  As the x86-64
	                   conversion has it: truncation towards zero; NaN and |a| >= 2^31 give INT_MIN */
	OP_CODES
};
inline const char* op_name(int code) {
	static const char* names[OP_CODES] = { "const", "ctl", "param", "osc", "oscset", "lpf", "lpfset", "env", "add", "sub", "mul", "div", "neg", "stopif", "stop", "setparam", "freq", "in", "delayin", "delaytap", "smooth", "operator", "cmp", "if", "else", "endif", "phi", "noise", "delayout", "tabread", "setctl", "delayset", "abs", "f2d", "dconst", "dlow", "dadd", "dsub", "dmul", "ddiv", "d2f", "envoff", "func", "powc", "trunc" };
	return (code >= 0 && code < OP_CODES) ? names[code] : "?";
}

struct Op { int code, dst, a, b, node; uint32_t imm; };
struct Dial { float min, max, initial; };
enum { GRAPH_MAX_CTL = 32 };     /* = KLG_MAX_CTL (klang_mi355_records.h) */

struct Program {
	int nctl = 0;
	Dial dials[GRAPH_MAX_CTL] = {};
	std::vector<int> nodes;      /* kind of node i */
	std::vector<int> node_arg;   /* Delay<SIZE>: SIZE; IIR<ORDER>: ORDER; Envelope / Operator: point capacity above four; else 0 */
	std::vector<Op> ops;
	int ret = -1, ret_r = -1;    /* ret_r: right channel of a Stereo::Effect / of a Stereo::Note (a note program with ret2) */
	bool stereo_note() const { return channels == 0 && ret_r >= 0; }
	int channels = 0;            /* 0 = synth note; 1 / 2 = effect with that many channels */
	int prepare_ops = 0;         /* the first prepare_ops ops are the effect's prepare(): once per block */
	int arg(int node) const { return node < (int)node_arg.size() ? node_arg[(size_t)node] : 0; }

	int noise_calls() const { int k = 0; for (const Op& o : ops) if (o.code == OP_NOISE) k++; return k; }   /* rand() draws per sample */
	int words() const { int w = 1; for (size_t i = 0; i < nodes.size(); i++) w += node_words(nodes[i], arg((int)i)); return w; }
	int node_word0(int node) const { int w = 1; for (int i = 0; i < node; i++) w += node_words(nodes[(size_t)i], arg(i)); return w; }

	std::string text() const {
		std::string s = "klgg 1\n";
		char line[160];
		if (channels) { snprintf(line, sizeof line, "kind effect %d\n", channels); s += line; }
		snprintf(line, sizeof line, "ctl %d\n", nctl); s += line;
		for (int i = 0; i < nctl; i++) { snprintf(line, sizeof line, "dial %d %.9g %.9g %.9g\n", i, dials[i].min, dials[i].max, dials[i].initial); s += line; }
		for (size_t i = 0; i < nodes.size(); i++) {
			if (arg((int)i)) snprintf(line, sizeof line, "node %zu %s %d\n", i, node_name(nodes[i]), arg((int)i)); else snprintf(line, sizeof line, "node %zu %s\n", i, node_name(nodes[i]));
			s += line;
		}
		for (const Op& o : ops) { snprintf(line, sizeof line, "op %s %d %d %d %d %08x\n", op_name(o.code), o.dst, o.a, o.b, o.node, o.imm); s += line; }
		if (prepare_ops) { snprintf(line, sizeof line, "prepare %d\n", prepare_ops); s += line; }
		if (channels == 2 || stereo_note()) snprintf(line, sizeof line, "ret2 %d %d\nend\n", ret, ret_r); else snprintf(line, sizeof line, "ret %d\nend\n", ret);
		s += line;
		return s;
	}

	/* returns "" on success, else a message */
	std::string parse(const char* text) {
		*this = Program();
		if (!text) return "program is NULL";
		std::string t(text);
		size_t pos = 0; int lineno = 0; bool header = false, ended = false;
		while (pos < t.size() && !ended) {
			size_t e = t.find('\n', pos); if (e == std::string::npos) e = t.size();
			std::string ln = t.substr(pos, e - pos); pos = e + 1; lineno++;
			const size_t hash = ln.find('#'); if (hash != std::string::npos) ln.resize(hash);
			char kw[32] = { 0 }; int n = 0;
			if (sscanf(ln.c_str(), " %31s%n", kw, &n) != 1) continue;
			const char* rest = ln.c_str() + n;
			auto bad = [&](const char* why) { char m[200]; snprintf(m, sizeof m, "graph program line %d: %s: '%s'", lineno, why, ln.c_str()); return std::string(m); };
			if (!strcmp(kw, "klgg")) { int v = 0; if (sscanf(rest, "%d", &v) != 1 || v != 1) return bad("unsupported version"); header = true; }
			else if (!header) return bad("missing 'klgg 1' header");
			else if (!strcmp(kw, "kind")) { char w[32]; if (sscanf(rest, "%31s %d", w, &channels) != 2 || strcmp(w, "effect") || channels < 1 || channels > 2) return bad("expected: kind effect 1|2"); }
			else if (!strcmp(kw, "ctl")) { if (sscanf(rest, "%d", &nctl) != 1 || nctl < 0 || nctl > GRAPH_MAX_CTL) return bad("ctl count must be 0..32"); }
			else if (!strcmp(kw, "dial")) { int i; float a, b, c; if (sscanf(rest, "%d %g %g %g", &i, &a, &b, &c) != 4 || i < 0 || i >= nctl) return bad("bad dial"); dials[i] = { a, b, c }; }
			else if (!strcmp(kw, "node")) {
				int id, size = 0; char kind[32];
				if (sscanf(rest, "%d %31s %d", &id, kind, &size) < 2 || id != (int)nodes.size()) return bad("nodes must be numbered 0,1,2,... in order");
				int k = -1; for (int q = 0; q < N_KINDS; q++) if (!strcmp(kind, node_name(q))) k = q;
				if (k < 0) return bad("unknown node kind");
				if ((int)nodes.size() >= MAX_NODES) return bad("too many nodes");
				if ((k == N_DELAY || k == N_NDELAY) && (size < 2 || size > (1 << 24))) return bad("delay needs its SIZE (2 .. 2^24)");
				if (k == N_IIRN && (size < 2 || size > 8)) return bad("iirn needs its ORDER (2 .. 8)");
				if ((k == N_ENV || k == N_OPERATOR) && size != 0 && (size < 5 || size > ENV_MAX_POINTS)) return bad("an envelope's point capacity is 5 .. 128 (none: four points)");
				nodes.push_back(k); node_arg.push_back(node_arg_of(k, size));
			}
			else if (!strcmp(kw, "op")) {
				char code[32]; Op o; unsigned imm;
				if (sscanf(rest, "%31s %d %d %d %d %x", code, &o.dst, &o.a, &o.b, &o.node, &imm) != 6) return bad("op needs: code dst a b node imm");
				o.imm = imm; o.code = -1; for (int q = 0; q < OP_CODES; q++) if (!strcmp(code, op_name(q))) o.code = q;
				if (o.code < 0) return bad("unknown op code");
				if ((int)ops.size() >= MAX_OPS) return bad("too many ops");
				ops.push_back(o);
			}
			else if (!strcmp(kw, "ret")) { if (sscanf(rest, "%d", &ret) != 1) return bad("bad ret"); }
			else if (!strcmp(kw, "prepare")) { if (sscanf(rest, "%d", &prepare_ops) != 1 || prepare_ops < 0) return bad("bad prepare"); }
			else if (!strcmp(kw, "ret2")) { if (sscanf(rest, "%d %d", &ret, &ret_r) != 2) return bad("bad ret2"); }
			else if (!strcmp(kw, "end")) ended = true;
			else return bad("unknown statement");
		}
		if (!header) return "graph program: missing 'klgg 1' header";
		if (!ended) return "graph program: missing 'end'";
		return validate();
	}

	/* single assignment, defined-before-use, node kinds match their ops */
	std::string validate() const {
		if (words() > MAX_WORDS) return "graph program: the voice record exceeds 2048 words";
		std::vector<char> defined;                                 /* 1 = visible here, 2 = assigned in a branch side that has ended */
		auto def = [&](int r) { return r >= 0 && r < (int)defined.size() && defined[(size_t)r] == 1; };
		struct Side { std::vector<int> regs; bool in_else; std::vector<int> then_side; };
		std::vector<Side> open;                                    /* the `if`s we are inside */
		std::vector<int> then_regs, else_regs;                     /* of the `if` just closed: what its phis may name */
		bool after_endif = false;
		auto in_list = [](const std::vector<int>& l, int r) { for (int x : l) if (x == r) return true; return false; };
		auto kind = [&](int n) { return (n >= 0 && n < (int)nodes.size()) ? nodes[(size_t)n] : -1; };
		std::vector<char> dbl;                                                    // registers that hold a double
		auto is_dbl = [&](int r) { return r >= 0 && r < (int)dbl.size() && dbl[(size_t)r] != 0; };
		char m[160];
		for (size_t i = 0; i < ops.size(); i++) {
			const Op& o = ops[i];
			auto bad = [&](const char* why) { snprintf(m, sizeof m, "graph program op %zu (%s): %s", i, op_name(o.code), why); return std::string(m); };
			bool need_a = false, need_b = false, has_dst = true, dst_dbl = false; int k = kind(o.node);
			switch (o.code) {
			case OP_CONST: break;
			case OP_CTL: if (o.imm >= (uint32_t)nctl) return bad("control index out of range"); break;
			case OP_PARAM: if (k != N_PARAM) return bad("node is not a param"); break;
			case OP_OSC: if (!is_oscillator(k)) return bad("node is not an oscillator"); break;
			case OP_OSCSET: if (!is_oscillator(k)) return bad("node is not an oscillator"); if (o.imm > 3u || (o.imm && k == N_WAVETABLE) || (o.imm == 2u && (k == N_SAW || k == N_PULSE)) || (o.imm == 3u && !(k == N_SAW || k == N_PULSE || k == N_BPULSE))) return bad("this oscillator has no such set() / reset() on the device"); need_a = o.imm != 2u; need_b = o.imm == 1u; has_dst = false; break;
			case OP_LPF: if (!is_modifier(k)) return bad("node is not a modifier"); need_a = true; break;
			case OP_LPFSET: if (k != N_LPF) return bad("node is not an lpf"); if (o.imm == 5 || o.imm > 6) return bad("this biquad type cannot be set() on the device"); need_a = need_b = true; has_dst = false; break;
			case OP_ENV: if (k != N_ENV && k != N_ADSR) return bad("node is not an envelope"); break;
			case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: need_a = need_b = true; break;
			case OP_NEG: need_a = true; break;
			case OP_STOPIF: if (k != N_ENV && k != N_ADSR) return bad("node is not an envelope"); has_dst = false; break;
			case OP_STOP: has_dst = false; break;
			case OP_SETPARAM: if (k != N_PARAM) return bad("node is not a param"); need_a = true; has_dst = false; break;
			case OP_FREQ: if (!is_oscillator(k) && k != N_OPERATOR) return bad("node is not an oscillator"); break;
			case OP_IN: if (channels == 0 || (int)o.imm >= channels) return bad("`in` needs an effect program with that channel"); break;
			case OP_DELAYIN: if (k != N_DELAY && k != N_NDELAY) return bad("node is not a delay"); need_a = true; has_dst = false; break;
			case OP_DELAYTAP: if (k != N_DELAY && k != N_NDELAY) return bad("node is not a delay"); if (o.imm > 3u || (o.imm == 3u && k != N_DELAY)) return bad("unknown tap kind"); need_a = true; break;
			case OP_DELAYOUT: if (k != N_NDELAY && k != N_DELAY) return bad("node is not a delay"); break;
			case OP_DELAYSET: if (k != N_NDELAY && k != N_DELAY) return bad("node is not a delay"); need_a = true; has_dst = false; break;
			case OP_SMOOTH: if (k != N_SMOOTH || o.imm >= (uint32_t)nctl) return bad("node is not a smoothed control"); if (!channels && !open.empty()) return bad("a Note's controls[i].smooth() may not sit inside an `if`: the bank advances the Synth's control by a fixed number of steps per sounding note and sample"); break;
			case OP_OPERATOR: if (k != N_OPERATOR) return bad("node is not an operator"); need_a = o.a >= 0; need_b = o.b >= 0; break;
			case OP_CMP: if (o.imm > 5u) return bad("unknown relation"); need_a = need_b = true; break;
			case OP_NOISE: if (!open.empty() || (int)i < prepare_ops) return bad("Noise may not sit inside an `if` or prepare()"); if (o.imm > 1u) return bad("unknown noise kind"); break;
			case OP_SETCTL: if (k != N_CTLVAR) return bad("node is not a written control"); if (!channels) return bad("only an effect writes its controls"); if ((int)(o.imm & 0xFFu) >= nctl || (o.imm >> 9)) return bad("control index out of range"); need_a = true; break;
			case OP_ABS: case OP_TRUNC: need_a = true; break;
			case OP_POWC: { need_a = true; float e; memcpy(&e, &o.imm, 4); if (!(e == 0.f || e == 1.f || e == 2.f || e == 3.f || e == 4.f || e == -1.f || e == -2.f || e == -3.f || e == -4.f)) return bad("powc: the exponent must be one of 0, +-1 .. +-4"); } break;
			case OP_FUNC: need_a = true; if ((o.imm & 0xFFu) > 2u || ((o.imm & 0xFFu) < 2u && (o.imm >> 8))) return bad("no such function");
				if ((o.imm & 0xFFu) == 2u) { const uint32_t bb = o.imm & 0xFFFFFF00u; float base; memcpy(&base, &bb, 4); if (!(base >= 1.17549435e-38f && base < 3.0e38f)) return bad("func pow: the base must be a positive normal number"); } if (!is_dbl(o.a)) return bad("operand a is not a double"); dst_dbl = true; break;
			case OP_F2D: need_a = true; if (is_dbl(o.a)) return bad("operand a is already a double"); dst_dbl = true; break;
			case OP_DCONST: dst_dbl = true; break;
			case OP_DLOW: need_a = true; if (!is_dbl(o.a)) return bad("operand a is not a double"); dst_dbl = true; break;
			case OP_DADD: case OP_DSUB: case OP_DMUL: case OP_DDIV: need_a = need_b = true; if (!is_dbl(o.a) || !is_dbl(o.b)) return bad("both operands must be doubles"); dst_dbl = true; break;
			case OP_D2F: need_a = true; if (!is_dbl(o.a)) return bad("operand a is not a double"); break;
			case OP_ENVOFF: if (k != N_ENV && k != N_ADSR) return bad("node is not an envelope"); if (o.imm > 2u) return bad("unknown envelope stage"); break;
			case OP_TABREAD: if (channels) return bad("tables are only available to synth notes"); if (o.imm == 0u) return bad("table id 0 is reserved"); need_a = true; break;
			case OP_IF: if ((int)i < prepare_ops) return bad("prepare() may not branch"); need_a = true; has_dst = false; open.push_back({ {}, false, {} }); break;
			case OP_ELSE:
				if (open.empty() || open.back().in_else) return bad("no open `if`");
				has_dst = false; open.back().then_side = open.back().regs; for (int r : open.back().regs) defined[(size_t)r] = 2; open.back().regs.clear(); open.back().in_else = true; break;
			case OP_ENDIF:
				if (open.empty()) return bad("no open `if`");
				has_dst = false;
				if (!open.back().in_else) { then_regs = open.back().regs; else_regs.clear(); } else { then_regs = open.back().then_side; else_regs = open.back().regs; }
				for (int r : open.back().regs) defined[(size_t)r] = 2;
				open.pop_back(); break;
			case OP_PHI:
				if (!after_endif) return bad("a phi must directly follow `endif`");
				if (!(def(o.a) || in_list(then_regs, o.a))) return bad("operand a is not visible at the end of the `if` side");
				if (!(def(o.b) || in_list(else_regs, o.b))) return bad("operand b is not visible at the end of the `else` side");
				break;
			default: return bad("unknown code");
			}
			after_endif = o.code == OP_ENDIF || o.code == OP_PHI;
			if (need_a && !def(o.a)) return bad("operand a is not defined");
			if (need_b && !def(o.b)) return bad("operand b is not defined");
			if (!((o.code >= OP_F2D && o.code <= OP_D2F) || o.code == OP_FUNC) && ((need_a && is_dbl(o.a)) || (need_b && is_dbl(o.b)) || (o.code == OP_PHI && (is_dbl(o.a) || is_dbl(o.b))))) return bad("a double register may only be read by dlow / dadd / dsub / dmul / ddiv / d2f");
			if (has_dst) {
				if (o.dst < 0 || o.dst >= MAX_OPS) return bad("bad destination register");
				if ((int)defined.size() <= o.dst) defined.resize((size_t)o.dst + 1, 0);
				if (defined[(size_t)o.dst]) return bad("register assigned twice");
				defined[(size_t)o.dst] = 1;
				if (dst_dbl) { if ((int)dbl.size() <= o.dst) dbl.resize((size_t)o.dst + 1, 0); dbl[(size_t)o.dst] = 1; }
				if (!open.empty()) open.back().regs.push_back(o.dst);
			}
		}
		if (!open.empty()) return "graph program: an `if` is not closed";
		if (prepare_ops > (int)ops.size()) return "graph program: 'prepare' names more ops than there are";
		{	/* sample ops may not read prepare() registers (prepare runs in another function of the generated patch) */
			std::vector<char> pre; for (int i = 0; i < prepare_ops; i++) if (ops[(size_t)i].dst >= 0) { if ((int)pre.size() <= ops[(size_t)i].dst) pre.resize((size_t)ops[(size_t)i].dst + 1, 0); pre[(size_t)ops[(size_t)i].dst] = 1; }
			auto is_pre = [&](int r) { return r >= 0 && r < (int)pre.size() && pre[(size_t)r]; };
			for (size_t i = (size_t)prepare_ops; i < ops.size(); i++) if (is_pre(ops[i].a) || is_pre(ops[i].b)) return "graph program: a sample op reads a register of prepare()";
			if (is_pre(ret) || is_pre(ret_r)) return "graph program: 'ret' names a register of prepare()";
			for (int i = 0; i < prepare_ops; i++) { const int c = ops[(size_t)i].code; if (c == OP_IN || c == OP_DELAYIN || c == OP_DELAYTAP || c == OP_OSC || c == OP_LPF || c == OP_ENV || c == OP_SMOOTH || c == OP_OPERATOR) return "graph program: prepare() may only compute and set()"; }
		}
		if (!def(ret)) return "graph program: 'ret' names an undefined register";
		if ((channels == 2 || stereo_note()) && !def(ret_r)) return "graph program: 'ret2' names an undefined register";
		if (channels == 1 && ret_r >= 0) return "graph program: 'ret2' in a one-channel effect";
		if (channels == 0) for (int k : nodes) if (k == N_DELAY) return "graph program: delay nodes need an effect program (kind effect); a Note's Delay member is a notedelay";
		if (channels != 0) for (int k : nodes) if (k == N_WAVETABLE || k == N_NDELAY) return "graph program: wavetable / notedelay nodes are only available to synth notes";
		return "";
	}
};

} }  /* namespace klg::graph */
#endif
