// include/klang/klang.h — source-compatible HOST façade of klang's DSL over libklang_mi355.so.
//
// north_star: "keeping the klang.h signal / >> stream-operator and set()/process()/on()/off() API surface so
// existing .k synths and effects compile unchanged: host-side C++ keeps the DSL, voice allocation and event
// dispatch while the per-block process() path calls hand-written CDNA4 HIP kernels through a thin C-ABI shim".
//
// What this header is: a clean-room re-implementation of the part of klang's API a synth patch (.k file) touches —
// signal/param/Control, Pitch -> Frequency, the Generator/Modifier protocol with `>>`, the Fast and Basic
// oscillators, Biquad LPF/HPF, Envelope/ADSR, Operator, Note/Synth (mono and Stereo) with the reference's voice
// allocation — in which
//   * every set()/on()/off() runs on the host exactly as in the reference (klang/host_dsl.hpp),
//   * every per-sample process() is DEVICE code: calling one on the host aborts with a message (there is no CPU
//     rendering path),
//   * Synth::process(float** / float*, int, float*) renders the block on the GPU through the C-ABI
//     (klang_mi355.h), after note events have moved the affected voices' state host <-> lane with
//     klg_voice_download / klg_voice_upload.
// Which kernel renders a Note type:
//   * a hand-written one, when the type is tied to a patch id with KLANG_GPU_BIND (klang/bindings.h holds the bindings of
//     the shipped patches), or
//   * a GENERATED one (graph patch, include/klang_mi355_graph.h): notes.add<T>() runs T::process() ONCE in recording mode —
//     every primitive's process(), every `>>`, `++`, arithmetic operator and filter.set() appends an op to a program
//     instead of computing — and the program is compiled for gfx950 by klg_synth_create_graph().  Supported in a
//     recorded process(): Fast::{Sine,Saw,Triangle,Square,Pulse} with their frequency set in on() or per sample
//     (`osc(f * (1 + lfo * depth))`: vibrato / FM by set(f)), the Basic oscillators, Operator<Sine> chains (`op1 * I >> op2 >> out`),
//     Wavetable / Sample (samples in HBM, klg_table_upload) and Table<float, N> reads with a recorded index, Delay<SIZE> members
//     (a line per voice in HBM: set(time) / clear() in on(), `delay >> x`, `delay << out`, `delay(time)` in process()),
//     every Biquad type, OnePole, DCF, IIR<1>, IIR<2..8>, Butterworth, Modal, Envelope::Follower (Biquad::LPF also set(f, Q) per sample), Envelope
//     (<= 4 points, setLoop) and ADSR `++`, + - * / and unary minus on signals / params / controls / constants, `.out` of a
//     member, signal and param members of the Note (read, and written for next-sample state), `>> out`, `out *= x`,
//     `if (env.finished()) stop();`, and data-dependent `if` / `else if` / `&&` / `!` on comparisons of signals, params and controls
//     (`if (in > 1) in = 1;`, `if (osc.frequency < fs.nyquist) out += osc / h;`): process() is then run once per outcome and the
//     traces are merged into structured if / else / endif + phi ops (gpu::PathMerger).  Anything else (a signal forced to a plain
//     float or int, set(f, phase) / reset() inside process()) stops with a message naming the construct.
// Effects: klang::gpu::EffectBank<FX> records a user Effect / Stereo::Effect the same way (prepare() becomes the per-block prologue);
//   an effect type tied to a hand-written kernel with KLANG_GPU_BIND_FX (the shipped PingPong.k / Reverb.k) is created with
//   klg_fx_create instead.  Stereo::Modifier, Stereo::Bank, Array, signals<N> and Matrix exist so that Reverb.k compiles unchanged.
//
// Reference interface citations (file:line) are into nashaudio/klang's klang.h v0.7.8.
#pragma once
#define KLANG_MI355 1              // this is the MI355X façade, not the reference header (what a host may test to add a KLANG_GPU_BIND line)

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <type_traits>
#include <typeindex>
#include <typeinfo>
#include <memory>
#include <unordered_map>
#include <vector>

#include "../klang_mi355.h"
#include "../klang_mi355_records.h"
#include "../klang_mi355_graph.h"
#include "host_dsl.hpp"

namespace klang {

typedef void event;

[[noreturn]] inline void device_only(const char* what) {
	std::fprintf(stderr, "klang-mi355: %s is per-sample code; it runs on the GPU (libklang_mi355.so), not on the host.\n", what);
	std::abort();
}

// ---- constants (klang.h:93-111, 227-233) ----
struct constant {
	double d; float f; int i; float inv;
	constexpr constant(double v) : d(v), f((float)v), i((int)v), inv(v == 0.0f ? 0.0f : (float)(1.0 / v)) {}
	constexpr operator float() const { return f; }
	float operator^(float x) const { return std::pow(f, x); }                     // klang.h:107-110: the constant raised to a power
	float operator^(int x) const { return static_cast<float>(std::pow(d, x)); }
	float operator^(double x) const { return static_cast<float>(std::pow(d, x)); }
};
constexpr constant pi = { 3.1415926535897932384626433832795 };
constexpr constant ln2 = { 0.6931471805599453094172321214581 };
constexpr constant root2 = { 1.4142135623730950488016887242097 };

template<typename T1, typename T2> inline T1 max(T1 a, T2 b) { return a > b ? a : (T1)b; }                              // klang.h:224 (returns the FIRST type)
template<typename T1, typename T2> inline T1 min(T1 a, T2 b) { return a < b ? a : (T1)b; }                              // klang.h:223
// power(base, exp) klang.h:152-218 (the overload for run-time exponents): an integral base is taken as float; integral exponents 0, +-1 .. +-4 are written-out
// products; a floating exponent first asks whether the base is 10 (then exp(exp * ln 10), the C library's, in double on the float product — see db_to_amplitude
// below, pinned against the reference), then for the same nine exponents; everything else is std::pow
template<typename BASE, typename EXP, std::enable_if_t<std::is_arithmetic_v<BASE> && std::is_arithmetic_v<EXP>, int> = 0>
inline std::conditional_t<std::is_integral_v<BASE>, float, BASE> power(BASE base, EXP exp) {
	if constexpr (std::is_integral_v<BASE>) return power((float)base, exp);
	else {
		auto small = [&](int e) -> BASE { BASE p = 1; const int m = e < 0 ? -e : e; if (m >= 1) p = base; for (int q = 1; q < m; q++) p = p * base; return e < 0 ? (BASE)1 / p : p; };
		if constexpr (std::is_integral_v<EXP>) { if (exp >= -4 && exp <= 4) return small((int)exp); }
		else {
			if (base == (BASE)10) {
				if constexpr (std::is_same_v<EXP, float>) return (BASE)(float)std::exp((double)(exp * 2.3025850929940456840179914546843642076011014886287729760333279009f));
				else return (BASE)std::exp(exp * (EXP)2.3025850929940456840179914546843642076011014886287729760333279009);
			}
			for (int e = -4; e <= 4; e++) if (exp == (EXP)e) return small(e);
		}
		return (BASE)std::pow(base, exp);
	}
}
// (klg_rand_sync: Noise generators draw from the same sequence on the device; the C library gets it back before host code draws — include/klang_mi355.h)
template<typename T> inline T random(const T mn, const T mx) { klg_rand_sync(); return std::rand() * ((mx - mn) / (T)RAND_MAX) + mn; }   // klang.h:236
inline void random(const unsigned seed) { std::srand(seed); klg_random_seed(seed); }                                     // klang.h:239

// =================================================================================================
// Recording a process() body into a graph program (include/klang_mi355_graph.h)
// =================================================================================================
struct signal;
namespace gpu {
// Which objects are ALIVE: every primitive / signal noted into an owner's construction log (below) gets a serial here and takes it
// out again in its destructor, so a log entry whose address has since been destroyed (or re-used by another object) is recognised
// as dead and never touched.
struct LiveMap { std::unordered_map<const void*, uint64_t> serial; uint64_t next = 1; };
inline thread_local LiveMap* live = nullptr;
inline void forget(const void* p) { live->serial.erase(p); }
// a primitive that lives in a lane record: host mirror <-> its block of record words (include/klang_mi355_graph.h)
struct Packable {
	virtual void pack(uint32_t* w) const = 0; virtual void unpack(const uint32_t* w) = 0; virtual ~Packable() { if (live) forget(this); }
	virtual void host_cursor(unsigned long long /*inputs so far*/) {}   // an effect's Delay: its write cursor, which on the device is derived from the sample count (a host-run prepare() places read heads against it)
};
struct Obj { const void* addr; size_t size; int kind; const Packable* packable; int arg; const void* key = nullptr; uint64_t serial = 0; const int* live_arg = nullptr; };   // live_arg: a Delay<0>'s run-time SIZE, read when the program is finished (resize() may come after the constructor)     // a primitive or a signal member seen while a Note / Effect was constructed (arg: Delay SIZE; key / serial: liveness)
// Where construction is noted: the Recorder of a prototype built by notes.add<T>() / gpu::EffectBank<FX> (members = the address range of
// the object), or the construction LOG every Plugin / Note owns — what lets Effect::process(buffer) and Note::process(buffer) find the
// members of an object the HOST constructed (`PingPong pingpong;`), whose type the base class does not know.
struct Sink {
	std::vector<Obj> objs; bool effect = false, tracked = false;
	void note(const void* addr, size_t size, int kind, const Packable* p = nullptr, int arg = 0, const void* key = nullptr, const int* live_arg = nullptr) {
		if (!key) key = p ? (const void*)p : addr;
		uint64_t serial = 0;
		if (tracked) { if (!live) live = new LiveMap(); auto it = live->serial.find(key); serial = it != live->serial.end() ? it->second : (live->serial[key] = live->next++); }
		for (Obj& o : objs) if (o.addr == addr && (!tracked || o.serial == serial)) { o.kind = kind; o.size = size; if (p) o.packable = p; if (arg) o.arg = arg; if (live_arg) o.live_arg = live_arg; return; }       // ADSR refines the Envelope it derives from, an Operator its oscillator
		objs.push_back({ addr, size, kind, p, arg, key, serial, live_arg });
	}
	bool alive(const Obj& o) const { if (!tracked) return true; if (!live) return false; const auto it = live->serial.find(o.key); return it != live->serial.end() && it->second == o.serial; }
};
// The log of one owner (a Plugin or a Note): open from the owner's base-class constructor — which runs before the members of the
// derived class are constructed — until the first thing that can only happen after construction (controls are assigned or read, an
// event, a block, the next owner's constructor).
struct ConstructionLog : Sink { ConstructionLog() { tracked = true; } };
inline thread_local ConstructionLog* log_target = nullptr;
inline thread_local int log_suppress = 0;                    // notes.add<T>() builds 128 notes of a type it has already recorded: nothing to log
inline void close_log() { log_target = nullptr; }
struct Recorder : Sink {
	bool constructing = false, recording = false;
	bool host_prepare = false;                                    // prepare() asked Controls::changed(): it stays HOST code (EffectBank runs it per instance and uploads what it changed)
	klg::graph::Program prog;
	int next_reg = 0;
	std::string error;
	void fail(const std::string& what) { if (error.empty()) error = what; }
	int smooth_node(const void* smoothed_signal) {               // controls[i].smooth() inside an effect: one state word per smoothed control
		for (size_t i = 0; i < objs.size(); i++) if (objs[i].addr == smoothed_signal) return (int)i;
		objs.push_back({ smoothed_signal, sizeof(float) * 2, klg::graph::N_SMOOTH, nullptr, 0 });
		return (int)objs.size() - 1;
	}
	int ctlvar_node(const void* control_value) {                  // controls[i].set(x) inside an effect's process(): the instance's own copy of the control
		for (size_t i = 0; i < objs.size(); i++) if (objs[i].addr == control_value && objs[i].kind == klg::graph::N_CTLVAR) return (int)i;
		objs.push_back({ control_value, sizeof(float) * 2, klg::graph::N_CTLVAR, nullptr, 0 });
		return (int)objs.size() - 1;
	}
	// (Sink::effect: recording an Effect::process() — in / delay / smooth are available)
	// data-dependent branches: process() is run once per outcome (record_paths below); `decisions` is the outcome list this run
	// follows, runs past its end take `true`.  Each test leaves an OP_IF marker (imm = the outcome taken) in the trace.
	std::vector<char> decisions; size_t decision_pos = 0; bool may_branch = false;
	int run_id = 0;                                               // counts the traced runs of process() (PathMerger): what happened "earlier in this run"
	bool decide(int cond_reg) {
		if (!may_branch) { fail("a data-dependent `if` is only supported in process() (not in prepare())"); return true; }
		if (decision_pos >= decisions.size()) decisions.push_back(1);
		const bool take = decisions[decision_pos++] != 0;
		emit(klg::graph::OP_IF, cond_reg, -1, -1, take ? 1u : 0u, false);
		return take;
	}
	int emit(int code, int a, int b, int node, uint32_t imm, bool has_dst) {
		klg::graph::Op o; o.code = code; o.a = a; o.b = b; o.node = node; o.imm = imm; o.dst = has_dst ? next_reg++ : -1;
		prog.ops.push_back(o);
		return o.dst;
	}
	int reg_of(const signal& s);
	// Table<float, SIZE> objects read with a recorded index: slot k (1-based, in order of first use) is uploaded as table id k
	// right after the bank is created (before any Wavetable member uploads its samples)
	struct StaticTable { const void* addr; std::vector<float> data; };
	std::vector<StaticTable> static_tables;
	uint32_t table_slot(const void* addr, const float* data, int n) {
		for (size_t k = 0; k < static_tables.size(); k++) if (static_tables[k].addr == addr) return (uint32_t)k + 1u;
		static_tables.push_back({ addr, std::vector<float>(data, data + n) });
		return (uint32_t)static_tables.size();
	}
	// literals: an op where they are first needed — except while process() is traced once per branch outcome (PathMerger), where
	// they live in a pool of their own (registers CONST_BASE + k), so that the traces of different outcomes line up op for op
	enum { CONST_BASE = 1 << 24 };
	bool pool_consts = false; std::vector<uint32_t> const_pool;
	int const_reg(uint32_t bits) {
		if (!pool_consts) return emit(klg::graph::OP_CONST, -1, -1, -1, bits, true);
		for (size_t k = 0; k < const_pool.size(); k++) if (const_pool[k] == bits) return CONST_BASE + (int)k;
		const_pool.push_back(bits); return CONST_BASE + (int)const_pool.size() - 1;
	}
	std::vector<int> node_of_obj;                               // objs index -> provisional node id (== objs index)
	int node(const void* addr, const char* what) {
		for (size_t i = 0; i < objs.size(); i++) if (objs[i].addr == addr) return (int)i;
		fail(std::string(what) + ": this object is not a member of the Note (only members can be used in a recorded process())");
		return 0;
	}
};
inline thread_local Recorder* rec = nullptr;
inline Sink* constructing() { return (rec && rec->constructing) ? static_cast<Sink*>(rec) : static_cast<Sink*>(log_target); }
inline Recorder* recording() { return (rec && rec->recording) ? rec : nullptr; }
inline uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
// host-side set()/reset()/release() calls inside a recorded process() would have to run per sample on the device
inline bool no_set_while_recording(const char* what) { if (Recorder* r = recording()) { r->fail(std::string(what) + " inside process() is not supported in a recorded graph (set it in on())"); return true; } return false; }
}

namespace gpu {
// `a < b` on signals: a plain bool outside a recording (or between two unrecorded values), a recorded condition inside one —
// testing it (`if`, `&&`, `?:`) asks the recorder which way this run goes
struct Pred {
	bool value; int reg = -1;
	explicit operator bool() const { if (reg >= 0) if (Recorder* r = recording()) return r->decide(reg); return value; }
	Pred operator!() const;
};
}

// ---- signal / relative / param (klang.h:1062-1200, 1357-1371) ----
// `reg` >= 0 only while a process() body is being recorded: the value lives in that register of the program.
struct relative;
struct Control;
struct dsignal;
struct signal {
	float value; int reg = -1;
	signal(const dsignal& d);                                   // (float) of a value computed in double (below)
	signal(const Control& c);                                   // the control's value (a Control also converts to float and int: name the one meant)
	signal(constant c) : value(c.f) { reg_member(); }
	signal(const float v = 0.f) : value(v) { reg_member(); }
	signal(const double v) : value((float)v) { reg_member(); }
	signal(const int v) : value((float)v) { reg_member(); }
	signal(const signal&) = default;
	signal& operator=(const signal&) = default;
	~signal() { if (gpu::live) gpu::forget(this); }           // (host side only: a member signal noted in a construction log leaves the live map)
	void reg_member() { if (gpu::Sink* r = gpu::constructing()) r->note(this, sizeof(signal), klg::graph::N_PARAM); }
	static signal bin(int code, const signal& a, const signal& b, float concrete) {
		gpu::Recorder* r = gpu::recording();
		if (!r || (a.reg < 0 && b.reg < 0)) return signal(concrete);
		signal s(concrete); const int ra = r->reg_of(a), rb = r->reg_of(b);
		s.reg = r->emit(code, ra, rb, -1, 0, true); return s;
	}
	const signal& operator<<(const signal& in) { value = in.value; reg = in.reg; return *this; }
	signal& operator>>(signal& dst) const { dst.value = value; dst.reg = reg; return dst; }
	signal operator+(const signal& x) const { return bin(klg::graph::OP_ADD, *this, x, value + x.value); }
	signal operator-(const signal& x) const { return bin(klg::graph::OP_SUB, *this, x, value - x.value); }
	signal operator*(const signal& x) const { return bin(klg::graph::OP_MUL, *this, x, value * x.value); }
	signal operator/(const signal& x) const { return bin(klg::graph::OP_DIV, *this, x, value / x.value); }
	signal operator-() const { gpu::Recorder* r = gpu::recording(); signal s(-value); if (r && reg >= 0) s.reg = r->emit(klg::graph::OP_NEG, reg, -1, -1, 0, true); return s; }
	signal& operator+=(const signal& x) { return *this = *this + x; }
	signal& operator-=(const signal& x) { return *this = *this - x; }
	signal& operator*=(const signal& x) { return *this = *this * x; }
	signal& operator/=(const signal& x) { return *this = *this / x; }
#define KLANG_SIGNAL_OPS(T) \
	signal& operator+=(T x) { return *this = *this + signal((float)x); } signal& operator-=(T x) { return *this = *this - signal((float)x); } \
	signal& operator*=(T x) { return *this = *this * signal((float)x); } signal& operator/=(T x) { return *this = *this / signal((float)x); } \
	signal operator+(T x) const { return *this + signal((float)x); } signal operator-(T x) const { return *this - signal((float)x); } \
	signal operator*(T x) const { return *this * signal((float)x); } signal operator/(T x) const { return *this / signal((float)x); }
	KLANG_SIGNAL_OPS(float) KLANG_SIGNAL_OPS(double) KLANG_SIGNAL_OPS(int)
	signal operator+(constant c) const { return *this + signal(c.f); } signal operator-(constant c) const { return *this - signal(c.f); }   // `x * root2`: the constant's float (klang.h:93-111)
	signal operator*(constant c) const { return *this * signal(c.f); } signal operator/(constant c) const { return *this / signal(c.f); }
#undef KLANG_SIGNAL_OPS
	// reading a recorded value as a plain float leaves the program: not representable
	void concrete_only(const char* what) const { if (reg >= 0) if (gpu::Recorder* r = gpu::recording()) r->fail(std::string(what) + " of a signal computed in process(): keep it a signal / param (a plain float cannot be recorded)"); }
	operator const float() const { concrete_only("float conversion"); return value; }
	operator float&() { concrete_only("float& conversion"); return value; }
	// `if (mute)`: non-zero (the built-in float -> bool), recordable
	static gpu::Pred cmp(uint32_t rel, const signal& a, const signal& b, bool concrete) {
		gpu::Recorder* r = gpu::recording();
		if (!r || (a.reg < 0 && b.reg < 0)) return gpu::Pred{ concrete, -1 };
		const int ra = r->reg_of(a), rb = r->reg_of(b);
		return gpu::Pred{ concrete, r->emit(klg::graph::OP_CMP, ra, rb, -1, rel, true) };
	}
	explicit operator bool() const { return (bool)cmp(5u, *this, signal(0.f), value != 0.f); }
	explicit operator bool() { return (bool)cmp(5u, *this, signal(0.f), value != 0.f); }
	relative operator+() const;
};
// comparisons (the reference compares through the float conversion; same result, but recordable).  Templates with exactly
// deduced operand types, so that they never compete with the built-in comparisons of plain numbers.
// `x > 0.001` with a double literal is a DOUBLE comparison of the converted float in the reference (klang.h:1115: the float conversion, then the
// built-in operator).  The same decision on floats: with bf = (float)b, (double)x > b  <=>  x >= bf when bf rounded up, x > bf when it
// rounded down (no float lies between b and bf) — likewise for the other three orderings.  (== / != against a double that is not a float
// keep the float literal: never / always true in the reference.)
inline gpu::Pred signal_cmp_double(uint32_t rel, const signal& x, double b, bool concrete) {
	const float bf = (float)b;
	if ((double)bf != b && rel < 4u) {
		const bool up = (double)bf > b;                                     // rel: 0 <  1 >  2 <=  3 >=
		rel = (rel == 1u || rel == 3u) ? (up ? 3u : 1u) : (up ? 0u : 2u);
	}
	return signal::cmp(rel, x, signal(bf), concrete);
}
#define KLANG_SIGNAL_CMP(OP, REL) \
	template<class A, class B, std::enable_if_t<std::is_base_of_v<signal, A> && std::is_base_of_v<signal, B>, int> = 0> inline gpu::Pred operator OP(const A& a, const B& b) { return signal::cmp(REL, a, b, a.value OP b.value); } \
	template<class A, class T, std::enable_if_t<std::is_base_of_v<signal, A> && std::is_arithmetic_v<T>, int> = 0> inline gpu::Pred operator OP(const A& a, T b) { if constexpr (std::is_same_v<T, double>) return signal_cmp_double(REL, a, b, (double)a.value OP b); else return signal::cmp(REL, a, signal((float)b), a.value OP (float)b); } \
	template<class T, class B, std::enable_if_t<std::is_arithmetic_v<T> && std::is_base_of_v<signal, B>, int> = 0> inline gpu::Pred operator OP(T a, const B& b) { if constexpr (std::is_same_v<T, double>) return signal_cmp_double(REL ^ (REL < 4u ? 1u : 0u), b, a, a OP (double)b.value); else return signal::cmp(REL, signal((float)a), b, (float)a OP b.value); }
KLANG_SIGNAL_CMP(<, 0u) KLANG_SIGNAL_CMP(>, 1u) KLANG_SIGNAL_CMP(<=, 2u) KLANG_SIGNAL_CMP(>=, 3u) KLANG_SIGNAL_CMP(==, 4u) KLANG_SIGNAL_CMP(!=, 5u)
#undef KLANG_SIGNAL_CMP
inline gpu::Pred gpu::Pred::operator!() const {
	Recorder* r = reg >= 0 ? recording() : nullptr;
	if (!r) return Pred{ !value, -1 };
	return Pred{ !value, r->emit(klg::graph::OP_CMP, reg, r->const_reg(0u), -1, 4u, true) };   // !c  ==  (c == 0)
}
struct relative : signal {};
inline relative signal::operator+() const { relative r; r.value = value; r.reg = reg; return r; }
inline signal& operator>>(float in, signal& dst) { dst.value = in; dst.reg = -1; return dst; }
// (templates: only a signal / param / ... on the right takes part — `constant` and `Control` keep their float conversions)
#define KLANG_SIGNAL_LEFT(OP, CODE) \
	template<class T, class S, std::enable_if_t<(std::is_same_v<T, float> || std::is_same_v<T, int>) && std::is_base_of_v<signal, S>, int> = 0> \
	inline signal operator OP(T x, const S& s) { return signal::bin(klg::graph::CODE, signal((float)x), s, (float)x OP s.value); }
KLANG_SIGNAL_LEFT(+, OP_ADD) KLANG_SIGNAL_LEFT(-, OP_SUB) KLANG_SIGNAL_LEFT(*, OP_MUL) KLANG_SIGNAL_LEFT(/, OP_DIV)
// (no `double` on the left: the reference has no such operators either, so `0.01 * sig` is built-in DOUBLE arithmetic through the
//  float conversion and stays a double — on()/off() code keeps exactly those roundings; inside a recorded process() write 0.01f)
#undef KLANG_SIGNAL_LEFT
// param + param, Frequency * param, ...: both operands exactly as written (otherwise derived-to-base on one side ties with the
// float conversion on the other and the call is ambiguous under ISO rules, which clang enforces)
#define KLANG_SIGNAL_PAIR(OP, CODE) \
	template<class A, class B, typename = std::enable_if_t<std::is_base_of_v<signal, A> && std::is_base_of_v<signal, B> && !(std::is_same_v<A, signal> && std::is_same_v<B, signal>)>> \
	inline signal operator OP(const A& a, const B& b) { return signal::bin(klg::graph::CODE, a, b, a.value OP b.value); }
KLANG_SIGNAL_PAIR(+, OP_ADD) KLANG_SIGNAL_PAIR(-, OP_SUB) KLANG_SIGNAL_PAIR(*, OP_MUL) KLANG_SIGNAL_PAIR(/, OP_DIV)
#undef KLANG_SIGNAL_PAIR
inline int gpu::Recorder::reg_of(const signal& s) { return s.reg >= 0 ? s.reg : const_reg(gpu::fbits(s.value)); }

// std::abs / abs of a signal (PingPong.k:46 `std::abs(delay - new_delay) > 0.001`): fabsf of the float the reference converts it to — recordable
inline signal abs_of(const signal& x) { gpu::Recorder* r = gpu::recording(); signal s(__builtin_fabsf(x.value)); if (r && x.reg >= 0) s.reg = r->emit(klg::graph::OP_ABS, x.reg, -1, -1, 0, true); return s; }
// min / max with a plain number FIRST and a signal second (klang.h:223-224: `a < b ? a : (T1)b` — the result has the FIRST type, so `min(20000, x)` of Modular.k:155 is an int:
// x truncated).  Recordable: the comparison is a data-dependent branch, the truncation an op
inline signal trunc_of(const signal& x) { gpu::Recorder* r = gpu::recording(); signal s((float)(int)x.value); if (r && x.reg >= 0) s.reg = r->emit(klg::graph::OP_TRUNC, x.reg, -1, -1, 0, true); return s; }
template<class A, std::enable_if_t<std::is_arithmetic_v<A>, int> = 0> inline signal min(A a, const signal& b) {
	if ((bool)(a < b)) return signal((float)a);
	if constexpr (std::is_integral_v<A>) return trunc_of(b); else return signal(b);
}
template<class A, std::enable_if_t<std::is_arithmetic_v<A>, int> = 0> inline signal max(A a, const signal& b) {
	if ((bool)(a > b)) return signal((float)a);
	if constexpr (std::is_integral_v<A>) return trunc_of(b); else return signal(b);
}
// power(signal, literal float): inside a recorded process() one op for the exponents the reference writes out (graph OP_POWC); any other exponent is the C library's powf
inline signal power(const signal& base, float e) {
	signal s(power(base.value, e));
	if (base.reg >= 0) if (gpu::Recorder* r = gpu::recording()) {
		bool small = false; for (int q = -4; q <= 4; q++) small = small || e == (float)q;
		if (!small) r->fail("power(x, e) of a value computed in process() with an exponent other than 0, +-1 .. +-4 (the C library's powf is not restated on the device)");
		else s.reg = r->emit(klg::graph::OP_POWC, base.reg, -1, -1, gpu::fbits(e), true);
	}
	return s;
}
inline signal power(const signal& base, double e) { if (base.reg >= 0 && gpu::recording()) gpu::rec->fail("power(x, double) of a value computed in process()"); return signal(power(base.value, e)); }
inline signal power(const signal& base, int e) { signal p(1.f); const int m = e < 0 ? -e : e; if (m > 4) { base.concrete_only("power(x, n) with |n| > 4"); return signal(power(base.value, e)); } if (m >= 1) p = base; for (int q = 1; q < m; q++) p = p * base; return e < 0 ? signal(1.f) / p : p; }
struct Control;
struct param : signal {
	param(constant c) : signal(c.f) {}
	param(const float v = 0.f) : signal(v) {}
	param(const signal& s) : signal(s) {}
	param(signal& s) : signal(s) {}
	param(const dsignal& d) : signal(d) {}
	param(Control& c);
};

// ---- Control / Controls / Presets (klang.h:1654-1981; UI fields omitted) ----
struct Control {
	enum Type { NONE, ROTARY, BUTTON, TOGGLE, SLIDER, MENU, METER, WHEEL };                                  // klang.h:1657-1667 (what a UI draws; nothing here depends on it)
	struct Size { int x, y, width, height; Size(int x_ = -1, int y_ = -1, int w_ = -1, int h_ = -1) : x(x_), y(y_), width(w_), height(h_) {} };   // klang.h:1670-1685
	std::string name; float min = 0.f, max = 1.f, initial = 0.f;
	Type type = ROTARY; Size size;
	signal value, smoothed; int index = 0;                   // index: position in its Controls (set by Controls::operator=)
	operator signal&() { return value; }
	operator param() const { return param(value); }
	operator float() const { value.concrete_only("float conversion of a Control"); return value.value; }
	// `const int p = controls[0];` (a Menu selecting a row of constants): truncation.  While recording it is a chain of data-dependent
	// branches `v < min + 1 ? min : v < min + 2 ? min + 1 : ...` over the control's range, so every choice gets its own recorded path
	operator int() const;
	signal smooth() {                                                                                        // klang.h:1715
		if (gpu::Recorder* r = gpu::recording()) {
			// an Effect: one lane = one instance, its own smoothed state.  A Note: the control is its Synth's and every sounding note advances
			// it in turn — the bank hands each voice the value its block starts from (klg_set_control_smoothed, klang_mi355.h)
			signal s(smoothed.value); s.reg = r->emit(klg::graph::OP_SMOOTH, -1, -1, r->smooth_node(&smoothed), (uint32_t)index, true); return s;
		}
		smoothed = smoothed.value * 0.999f + (1.f - 0.999f) * value.value; return smoothed;
	}
	bool touched = true;                                     // set() since the last block a gpu::FxRunner sent the controls (a set() overwrites what the effect wrote, even with the same value)
	Control& set(float x) { value = (x < min) ? min : (max < x) ? max : x; touched = true; return *this; }    // klang.h:1725
	// controls[i].set(x) with a value computed in process() (PingPong.k:48,60): recorded in an Effect — the control becomes state of the instance
	Control& set(const signal& x) {
		gpu::Recorder* r = gpu::recording();
		if (!r) return set(x.value);
		if (!r->effect) { r->fail("controls[i].set() inside a Note::process(): the control is the Synth's, shared by its notes"); return *this; }
		const float v = (x.value < min) ? min : (max < x.value) ? max : x.value;
		value.reg = r->emit(klg::graph::OP_SETCTL, r->reg_of(x), -1, r->ctlvar_node(&value), (uint32_t)index, true); value.value = v;
		return *this;
	}
	// `x >> controls[i]` / `controls[i] << x` (klang.h:1745-1746): the plain assignment, no clamp — a METER fed by process() (Vocoder.k:104).  Recorded in an Effect like
	// set(): the control becomes state of the instance (what the UI reads back is that word of its record)
	Control& operator<<(const signal& x) {
		gpu::Recorder* r = gpu::recording();
		if (!r || (x.reg < 0 && !r->effect)) { value = x.value; return *this; }
		if (!r->effect) { r->fail("controls[i] << x inside a Note::process(): the control is the Synth's, shared by its notes"); return *this; }
		value.reg = r->emit(klg::graph::OP_SETCTL, r->reg_of(x), -1, r->ctlvar_node(&value), (uint32_t)index | 0x100u, true); value.value = x.value;
		return *this;
	}
	float range() const { return max - min; }                                                                // klang.h:1719-1721: what a host's 0..1 parameter maps to
	float normalised() const { const float v = value.value; return range() ? (v - min) / range() : (v < 0.f ? 0.f : (1.f < v ? 1.f : v)); }
	void setNormalised(float norm) { value = norm * range() + min; }
};
inline param::param(Control& c) : signal(c.value) {}
inline signal::signal(const Control& c) : value(c.value.value), reg(c.value.reg) {}
inline Control::operator int() const {
	if (value.reg < 0 || !gpu::recording()) return (int)value.value;
	const int lo = (int)min, hi = (int)max;
	if (hi - lo > 15) { gpu::rec->fail("int conversion of a Control with more than 16 values inside process()"); return (int)value.value; }
	for (int k = lo; k < hi; k++) if ((bool)signal::cmp(0u, value, signal((float)(k + 1)), value.value < (float)(k + 1))) return k;
	return hi;
}
// signal (op) Control and Control (op) signal: the control's value (recorded as a control read inside a recorded process()).
// Templates, so that only a signal / param / ... operand takes part (plain numbers keep the Control's float conversion).
#define KLANG_CONTROL_OPS(OP) \
	template<class S, typename = std::enable_if_t<std::is_base_of_v<signal, S>>> inline signal operator OP(const S& a, Control& c) { return static_cast<const signal&>(a) OP c.value; } \
	template<class S, typename = std::enable_if_t<std::is_base_of_v<signal, S>>> inline signal operator OP(Control& c, const S& a) { return c.value OP static_cast<const signal&>(a); }
KLANG_CONTROL_OPS(+) KLANG_CONTROL_OPS(-) KLANG_CONTROL_OPS(*) KLANG_CONTROL_OPS(/)
#undef KLANG_CONTROL_OPS
// comparisons of a control (recordable like those of a signal)
#define KLANG_CONTROL_CMP(OP) \
	template<class T, std::enable_if_t<std::is_arithmetic_v<T>, int> = 0> inline gpu::Pred operator OP(Control& c, T x) { return c.value OP x; } \
	template<class T, std::enable_if_t<std::is_arithmetic_v<T>, int> = 0> inline gpu::Pred operator OP(T x, Control& c) { return x OP c.value; } \
	template<class S, std::enable_if_t<std::is_base_of_v<signal, S>, int> = 0> inline gpu::Pred operator OP(Control& c, const S& s) { return c.value OP static_cast<const signal&>(s); } \
	template<class S, std::enable_if_t<std::is_base_of_v<signal, S>, int> = 0> inline gpu::Pred operator OP(const S& s, Control& c) { return static_cast<const signal&>(s) OP c.value; }
KLANG_CONTROL_CMP(<) KLANG_CONTROL_CMP(>) KLANG_CONTROL_CMP(<=) KLANG_CONTROL_CMP(>=) KLANG_CONTROL_CMP(==) KLANG_CONTROL_CMP(!=)
#undef KLANG_CONTROL_CMP
inline Control Dial(const char* name, float mn = 0.f, float mx = 1.f, float initial = 0.f, Control::Size size = Control::Size()) { Control c; c.name = name; c.min = mn; c.max = mx; c.initial = initial; c.value = initial; c.size = size; return c; }
// the other control kinds (klang.h:1801-1856): on this side of the boundary a control is its range and value (type and size are kept for a host that draws them)
inline Control Slider(const char* name, float mn = 0.f, float mx = 1.f, float initial = 0.f, Control::Size size = Control::Size()) { Control c = Dial(name, mn, mx, initial, size); c.type = Control::SLIDER; return c; }
inline Control Meter(const char* name, float mn = 0.f, float mx = 1.f, float initial = 0.f, Control::Size size = Control::Size()) { Control c = Dial(name, mn, mx, initial, size); c.type = Control::METER; return c; }
inline Control Button(const char* name, Control::Size size = Control::Size()) { Control c = Dial(name, 0.f, 1.f, 0.f, size); c.type = Control::BUTTON; return c; }
inline Control Toggle(const char* name, bool initial = false) { return Dial(name, 0.f, 1.f, initial ? 1.f : 0.f); }
template<typename... Options> inline Control Menu(const char* name, const Options... options) { Control c = Dial(name, 0.f, (float)sizeof...(options) - 1.f, 0.f); c.type = Control::MENU; return c; }
template<typename... Options> inline Control Menu(const char* name, Control::Size size, const Options... options) { Control c = Dial(name, 0.f, (float)sizeof...(options) - 1.f, 0.f, size); c.type = Control::MENU; return c; }   // klang.h:1829-1838
inline Control PitchBend() { return Dial("PITCH\nBEND", 0.f, 16384.f, 8192.f); }
inline Control ModWheel() { return Dial("MOD\nWHEEL", 0.f, 127.f, 0.f); }
struct Group {                                          // klang.h:1853-1873: `{ Dial(..) }` or `{ "name", Dial(..), Dial(..) }`
	const char* name; std::vector<Control> controls;
	template<typename... C> Group(const char* n, C... c) : name(n), controls{ c... } {}
	template<typename... C> Group(C... c) : name(""), controls{ c... } {}
	template<typename... C> Group(const char* n, Control::Size, C... c) : name(n), controls{ c... } {}     // `{ "LFO", { 269, 131 }, Dial(..), .. }`: a frame with a position (klang.h:1866-1871)
	template<typename... C> Group(Control::Size, C... c) : name(""), controls{ c... } {}
};
struct Controls {
	std::vector<Control> items; float cache[128] = { 0 };
	// (controls are assigned in the constructor BODY of a plugin and read by its host: either way every member of the plugin has been
	//  constructed, so the owner's construction log ends here)
	void operator=(std::initializer_list<Group> l) { gpu::close_log(); items.clear(); for (const Group& g : l) for (const Control& c : g.controls) { items.push_back(c); items.back().index = (int)items.size() - 1; } }   // klang.h:1895-1902
	void add(const char* name, Control::Type type = Control::ROTARY, float mn = 0.f, float mx = 1.f, float initial = 0.f, Control::Size size = Control::Size()) {   // klang.h:1904-1912
		gpu::close_log(); Control c; c.name = name; c.type = type; c.min = mn; c.max = mx; c.initial = initial; c.value = initial; c.size = size; items.push_back(c); items.back().index = (int)items.size() - 1;
	}
	void group(const char* /*name*/, unsigned /*start*/, unsigned /*length*/, Control::Size = Control::Size()) {}      // klang.h:1925-1928: a UI frame around controls
	Control& operator[](int i) { gpu::close_log(); return items[(size_t)i]; }
	const Control& operator[](int i) const { return items[(size_t)i]; }
	unsigned size() const { gpu::close_log(); return (unsigned)items.size(); }
	// (while an effect's prepare() is being recorded: this prepare() is host code — tables drawn with rand(), loops over a count, caches compared with != —
	//  and is not recorded at all; gpu::EffectBank runs it on a host mirror of every instance whose dials moved, see there)
	bool changed() { if (gpu::Recorder* r = gpu::recording()) { r->host_prepare = true; return false; } bool c = false; for (size_t i = 0; i < items.size(); i++) if (items[i].value.value != cache[i]) { cache[i] = items[i].value.value; c = true; } return c; }   // klang.h:1914
};
struct Preset { std::string name; std::vector<float> values; Preset(const char* n, std::initializer_list<double> v) : name(n) { for (double x : v) values.push_back((float)x); } };
struct Presets { std::vector<Preset> items; void operator=(std::initializer_list<Preset> l) { items.assign(l.begin(), l.end()); } };

// ---- units (klang.h:1512-1652) ----
struct Conversion : signal { using signal::signal; };
struct Frequency : param { using param::param; Frequency(float f = 1000.f) : param(f) {} };
struct Pitch : param {
	using param::param;
	static inline thread_local Conversion Frequency;
	const Pitch* operator->() { concrete_only("Pitch -> Frequency"); Frequency = klg::host::pitch_to_frequency(value); return this; }             // klang.h:1568-1571
};
// dB <-> linear (klang.h:1609-1652): `GAIN[o]->Amplitude`.  power(10, x) of the reference is ::exp(x * ln 10) on the float product,
// evaluated in double (the global ::exp of <cmath>) and rounded once
inline float db_to_amplitude(float db) { return (float)std::exp((double)((db * 0.05f) * 2.3025850929940456840179914546843642076011014886287729760333279009f)); }
struct dB : param {
	using param::param;
	dB(float gain = 0.f) : param(gain) {}
	static inline thread_local Conversion Amplitude;
	const dB* operator->() const { value_only("dB -> Amplitude"); Amplitude = db_to_amplitude(value); return this; }
	void value_only(const char* what) const { concrete_only(what); }
};
struct Amplitude : param {
	using param::param;
	Amplitude(float a = 1.f) : param(a) {}
	Amplitude(const klang::dB& db) : param(db_to_amplitude(db.value)) { db.value_only("dB -> Amplitude"); }
	static inline thread_local Conversion dB;
	const Amplitude* operator->() const { concrete_only("Amplitude -> dB"); dB = 20.f * log10f(value); return this; }
};
typedef Amplitude Velocity;

struct SampleRate {
	float f; int i; double d; float inv, w, nyquist;
	SampleRate(float sr) : f(sr), i(int(sr + 0.001f)), d((double)sr), inv(1.f / sr), w(2.0f * pi * inv), nyquist(sr / 2.f) {}
	operator float() { return f; }
};
inline SampleRate fs(44100);       // ONE definition per program (the reference's is `static` per translation unit, F7)
inline klg::host::Fs host_fs() { return klg::host::Fs(fs.f); }
// ---- values the reference computes in DOUBLE from a control: `(controls[1] + 0.01232 * c) * fs` (examples/Delay/Reverb2.k:43) is Control -> float, float + double,
// double * float -> a double, which Delay::operator()(double) turns into tap((float)x).  A Control (op) a DOUBLE operand — exactly double: ints and floats keep the
// float arithmetic they have in the reference — yields a dsignal: a double on the host, a double REGISTER of the program while recording (f2d / dconst + dlow /
// dadd / dsub / dmul / ddiv), rounded to float once, where the reference rounds (d2f: becoming a signal / param / delay time). ----
struct dsignal {
	double value = 0.; int reg = -1;
	explicit dsignal(double v = 0.) : value(v) {}                              // (explicit: a plain number never turns into one on its own)
	static int reg_of(gpu::Recorder* r, const dsignal& x) {
		if (x.reg >= 0) return x.reg;
		uint64_t u; std::memcpy(&u, &x.value, 8);
		const int hi = r->emit(klg::graph::OP_DCONST, -1, -1, -1, (uint32_t)(u >> 32), true);
		return (uint32_t)u ? r->emit(klg::graph::OP_DLOW, hi, -1, -1, (uint32_t)u, true) : hi;
	}
	static dsignal from(const signal& s) { dsignal d((double)s.value); if (s.reg >= 0) if (gpu::Recorder* r = gpu::recording()) d.reg = r->emit(klg::graph::OP_F2D, r->reg_of(s), -1, -1, 0, true); return d; }
	static dsignal bin(int code, const dsignal& a, const dsignal& b, double concrete) {
		dsignal d(concrete);
		if (a.reg >= 0 || b.reg >= 0) if (gpu::Recorder* r = gpu::recording()) { const int ra = reg_of(r, a), rb = reg_of(r, b); d.reg = r->emit(code, ra, rb, -1, 0, true); }
		return d;
	}
	operator double() const { if (reg >= 0 && gpu::recording()) gpu::rec->fail("a plain double out of a value computed in process(): keep it in the expression (or make it a signal / param)"); return value; }
};
inline signal::signal(const dsignal& d) : value((float)d.value) { if (d.reg >= 0) if (gpu::Recorder* r = gpu::recording()) reg = r->emit(klg::graph::OP_D2F, d.reg, -1, -1, 0, true); }
#define KLANG_DSIGNAL_OPS(OP, CODE) \
	template<class T, std::enable_if_t<std::is_same_v<T, double>, int> = 0> inline dsignal operator OP(Control& c, T x) { return dsignal::bin(klg::graph::CODE, dsignal::from(c.value), dsignal(x), (double)c.value.value OP x); } \
	template<class T, std::enable_if_t<std::is_same_v<T, double>, int> = 0> inline dsignal operator OP(T x, Control& c) { return dsignal::bin(klg::graph::CODE, dsignal(x), dsignal::from(c.value), x OP (double)c.value.value); } \
	inline dsignal operator OP(const dsignal& a, const dsignal& b) { return dsignal::bin(klg::graph::CODE, a, b, a.value OP b.value); } \
	template<class T, std::enable_if_t<std::is_arithmetic_v<T>, int> = 0> inline dsignal operator OP(const dsignal& a, T x) { return dsignal::bin(klg::graph::CODE, a, dsignal((double)x), a.value OP (double)x); } \
	template<class T, std::enable_if_t<std::is_arithmetic_v<T>, int> = 0> inline dsignal operator OP(T x, const dsignal& b) { return dsignal::bin(klg::graph::CODE, dsignal((double)x), b, (double)x OP b.value); } \
	template<class S, std::enable_if_t<std::is_base_of_v<signal, S>, int> = 0> inline dsignal operator OP(const dsignal& a, const S& s) { return dsignal::bin(klg::graph::CODE, a, dsignal::from(s), a.value OP (double)s.value); } \
	template<class S, std::enable_if_t<std::is_base_of_v<signal, S>, int> = 0> inline dsignal operator OP(const S& s, const dsignal& b) { return dsignal::bin(klg::graph::CODE, dsignal::from(s), b, (double)s.value OP b.value); } \
	inline dsignal operator OP(const dsignal& a, const SampleRate& r) { return dsignal::bin(klg::graph::CODE, a, dsignal((double)r.f), a.value OP (double)r.f); }
KLANG_DSIGNAL_OPS(+, OP_DADD) KLANG_DSIGNAL_OPS(-, OP_DSUB) KLANG_DSIGNAL_OPS(*, OP_DMUL) KLANG_DSIGNAL_OPS(/, OP_DDIV)
#undef KLANG_DSIGNAL_OPS
// `controls[0] * fs`, `mod * fs`: the sample rate is a plain number (its float conversion would un-record a recorded left side)
// (R is deduced, so a plain float never converts into a SampleRate to get here)
template<class S, class R, std::enable_if_t<std::is_base_of_v<signal, S> && std::is_same_v<R, SampleRate>, int> = 0> inline signal operator*(const S& a, const R& r) { return static_cast<const signal&>(a) * signal(r.f); }
template<class S, class R, std::enable_if_t<std::is_base_of_v<signal, S> && std::is_same_v<R, SampleRate>, int> = 0> inline signal operator/(const S& a, const R& r) { return static_cast<const signal&>(a) / signal(r.f); }
template<class R, std::enable_if_t<std::is_same_v<R, SampleRate>, int> = 0> inline signal operator*(Control& c, const R& r) { return c.value * signal(r.f); }
template<class R, std::enable_if_t<std::is_same_v<R, SampleRate>, int> = 0> inline signal operator/(Control& c, const R& r) { return c.value / signal(r.f); }
// Control (op) float / int and the mirror image inside a recorded process() keep the control a control read (same fp32 result as
// the built-in; doubles stay with the built-in double arithmetic)
#define KLANG_CONTROL_NUM(OP) \
	template<class T, std::enable_if_t<std::is_same_v<T, float> || std::is_same_v<T, int>, int> = 0> inline signal operator OP(Control& c, T x) { return c.value OP signal((float)x); } \
	template<class T, std::enable_if_t<std::is_same_v<T, float> || std::is_same_v<T, int>, int> = 0> inline signal operator OP(T x, Control& c) { return signal((float)x) OP c.value; }
KLANG_CONTROL_NUM(+) KLANG_CONTROL_NUM(-) KLANG_CONTROL_NUM(*) KLANG_CONTROL_NUM(/)
#undef KLANG_CONTROL_NUM
// double (op) Control: the built-in DOUBLE arithmetic on the control's float value, spelled out (the control also converts to int —
// `const int p = controls[0];` — so the built-in candidates alone would be ambiguous)
#define KLANG_CONTROL_DBL(OP) \
	template<class T, std::enable_if_t<std::is_same_v<T, double>, int> = 0> inline double operator OP(T x, const Control& c) { return x OP (double)(float)c; } \
	template<class T, std::enable_if_t<std::is_same_v<T, double>, int> = 0> inline double operator OP(const Control& c, T x) { return (double)(float)c OP x; }
KLANG_CONTROL_DBL(+) KLANG_CONTROL_DBL(-) KLANG_CONTROL_DBL(*) KLANG_CONTROL_DBL(/)
#undef KLANG_CONTROL_DBL
// sqr / cube (klang.h:3067-3069: Function<float> objects; applied to a signal they are the same fp32 products)
// tanh of a signal: where the reference's patch code writes `tanh(x)` on a float (a plain C function: examples/Distortion/Shaping.k:15) it is the C library's DOUBLE
// tanh of the converted float (the pinned build imports `tanh`); the device restates glibc's (klg_device.hpp glibc_tanh, tools/verify_tanh_f64.c)
inline dsignal tanh(const signal& x) {                                  // (a double: the expression around it stays double in the reference — `tanh(c * x) / tanh(c)` divides doubles)
	const dsignal a = dsignal::from(x); dsignal d(::tanh(a.value));
	if (a.reg >= 0) if (gpu::Recorder* r = gpu::recording()) d.reg = r->emit(klg::graph::OP_FUNC, a.reg, -1, -1, 0, true);
	return d;
}
// pow(B, x) of a signal with a plain number as the base (examples/Subtractive/Modular.k:24 `pow(10, 2 * (x - 1))`, :123 `pow(2, (signal)osc)`): in the reference std::pow(int, float)
// — the C library's DOUBLE pow of the converted float, and for the base 2 the pinned compiler calls exp2 instead (the reference binary imports `pow` and `exp2`).  A double: the
// expression around it stays double (`pow(2, x) * 0.5`).  On the device: klg_glibc_pow.hpp (graph OP_FUNC 1 / 2)
template<class B, std::enable_if_t<std::is_arithmetic_v<B>, int> = 0> inline dsignal pow(B base, const signal& e) {
	const double b = (double)base; const dsignal a = dsignal::from(e);
	dsignal d(b == 2.0 ? ::exp2(a.value) : ::pow(b, a.value));
	if (a.reg >= 0) if (gpu::Recorder* r = gpu::recording()) {
		const float bf = (float)b; const uint32_t bits = gpu::fbits(bf);
		if (b == 2.0) d.reg = r->emit(klg::graph::OP_FUNC, a.reg, -1, -1, 1u, true);
		else if ((double)bf != b || (bits & 0xFFu) || !(bf >= 1.17549435e-38f && bf < 3.0e38f)) r->fail("pow(B, x) of a value computed in process(): the base must be a positive number with at most 16 significant bits");
		else d.reg = r->emit(klg::graph::OP_FUNC, a.reg, -1, -1, bits | 2u, true);
	}
	return d;
}
inline signal sqr(const signal& x) { return x * x; }
inline signal cube(const signal& x) { return x * x * x; }
// `x >> debug`: the plugin's debug scope (klang.h:3299); nothing to plot here
struct DebugSink { template<class T> void operator<<(const T&) {} };
inline thread_local DebugSink debug;

// ---- Generator / Modifier protocol (klang.h:2180-2329): reading an object as a signal runs its process() ----
namespace Generic {
template<class SIGNAL> struct Input {
	SIGNAL in = { 0.f };
	virtual ~Input() {}
	virtual void operator<<(const SIGNAL& src) { in = src; this->input(); }
	virtual void input(const SIGNAL& src) { in = src; this->input(); }
protected:
	virtual void input() {}
};
template<class SIGNAL> struct Output {
	SIGNAL out = { 0.f };
	virtual ~Output() {}
	template<class T> T& operator>>(T& dst) { this->process(); return dst = out; }
	virtual operator const SIGNAL&() { this->process(); return out; }
	template<class T> SIGNAL operator+(T& o) { this->process(); return out + SIGNAL(o); }
	template<class T> SIGNAL operator*(T& o) { this->process(); return out * SIGNAL(o); }
	template<class T> SIGNAL operator-(T& o) { this->process(); return out - SIGNAL(o); }
	template<class T> SIGNAL operator/(T& o) { this->process(); return out / SIGNAL(o); }
protected:
	virtual void process() = 0;
};
template<class S> inline S operator+(Output<S>& o, float x) { return S(o) + x; }
template<class S> inline S operator*(Output<S>& o, float x) { return S(o) * x; }
template<class S> inline S operator-(Output<S>& o, float x) { return S(o) - x; }
template<class S> inline S operator/(Output<S>& o, float x) { return S(o) / x; }
template<class S> inline S operator+(float x, Output<S>& o) { return S(o) + x; }
template<class S> inline S operator*(float x, Output<S>& o) { return S(o) * x; }
// object (op) signal, also for temporaries (`osc * (a++ + b++)`): read the object, then combine; and signal (op) object.
// Templates on the signal side: only a signal / param / ... written as such takes part (an Operator or an ADSR on that side keeps
// the meaning its own class gives the operator, e.g. `op * adsr` sets the operator's amp).
#define KLANG_OBJECT_OPS(OP) \
	template<class S, typename = std::enable_if_t<std::is_base_of_v<signal, S>>> inline signal operator OP(Output<signal>& o, const S& x) { const signal& a = o; return a OP static_cast<const signal&>(x); } \
	template<class S, typename = std::enable_if_t<std::is_base_of_v<signal, S>>> inline signal operator OP(const S& a, Output<signal>& o) { const signal& b = o; return static_cast<const signal&>(a) OP b; }
KLANG_OBJECT_OPS(+) KLANG_OBJECT_OPS(-) KLANG_OBJECT_OPS(*) KLANG_OBJECT_OPS(/)
#undef KLANG_OBJECT_OPS
template<class SIGNAL> struct Generator : Output<SIGNAL> {
	template<typename... P> Output<SIGNAL>& operator()(P... p) { this->set(p...); return *this; }
	using Output<SIGNAL>::operator>>;
protected:
	virtual void set(param) {}
	virtual void set(relative) {}
	virtual void set(param, param) {}
	virtual void set(param, relative) {}
	virtual void set(param, param, param) {}
	virtual void set(param, param, param, param) {}
};
template<class SIGNAL> struct Modifier : Input<SIGNAL>, Output<SIGNAL> {
	using Input<SIGNAL>::input;
	template<typename... P> Modifier<SIGNAL>& operator()(P... p) { this->set(p...); return *this; }
protected:
	virtual void set(param) {}
	virtual void set(param, param) {}
	virtual void set(param, relative) {}
	virtual void set(param, param, param) {}
};
}
struct Input : Generic::Input<signal> {};
struct Output : Generic::Output<signal> {};
struct Generator : Generic::Generator<signal> {};
struct Modifier : Generic::Modifier<signal> {};
inline signal& operator+=(signal& s, Generic::Output<signal>& o) { s.value += signal(o).value; return s; }
// double (op) object — `0.125 / controls[1] * (in >> hpf[0])` (Vocoder.k:91): the reference's operator(float, Output&) klang.h:2236-2244 — the double rounds to float, then fp32
#define KLANG_DSIGNAL_OBJECT_OPS(OP) \
	inline signal operator OP(const dsignal& d, Generic::Output<signal>& o) { const signal a(d); const signal& b = o; return a OP b; } \
	inline signal operator OP(Generic::Output<signal>& o, const dsignal& d) { const signal& a = o; const signal b(d); return a OP b; }
KLANG_DSIGNAL_OBJECT_OPS(+) KLANG_DSIGNAL_OBJECT_OPS(-) KLANG_DSIGNAL_OBJECT_OPS(*) KLANG_DSIGNAL_OBJECT_OPS(/)
#undef KLANG_DSIGNAL_OBJECT_OPS
// Control (op) object (`controls[1] * lfo`): the control's value and the object's next output (the reference: Control -> signal&, then signal (op) object)
#define KLANG_CONTROL_OBJECT_OPS(OP) inline signal operator OP(Control& c, Generic::Output<signal>& o) { const signal& b = o; return c.value OP b; }
KLANG_CONTROL_OBJECT_OPS(+) KLANG_CONTROL_OBJECT_OPS(-) KLANG_CONTROL_OBJECT_OPS(*) KLANG_CONTROL_OBJECT_OPS(/)
#undef KLANG_CONTROL_OBJECT_OPS

// Function<Args...> (klang.h:2331-2532 as the `optimised` / `basic` namespaces spell it: the signal type first): a C function applied to the signal stream —
// `Function<float, float> f(softclip); in >> f(distort) >> out;` calls softclip(in, distort) per sample (all but the first argument bound by operator(); all of them:
// the first one is the input).  The function is the patch's own code: to be RECORDED its arguments must be the tracing type, i.e. the patch is compiled with
// -DKLANG_GPU_TRACE_FLOAT (the end of this header): `float` in the patch's text is then klang::signal, and so are Args.
// `a >> b`: b.input(a) when b is an Input, else plain assignment (klang.h:4868-4890)
template<class SRC, class DST, typename = std::enable_if_t<!std::is_arithmetic_v<SRC>>>
inline DST& operator>>(SRC& src, DST& dst) {
	if constexpr (std::is_base_of_v<Generic::Input<signal>, DST>) dst.input(src); else dst << src;
	return dst;
}
template<class SRC, class DST, typename = std::enable_if_t<!std::is_arithmetic_v<SRC>>>
inline DST& operator>>(const SRC& src, DST& dst) {
	if constexpr (std::is_base_of_v<Generic::Input<signal>, DST>) dst.input(src); else dst << src;
	return dst;
}

struct GraphStub;
template<typename... Args> struct Function : Modifier {
	static_assert(sizeof...(Args) >= 1, "Function<x, ...>: at least the input");
	static constexpr bool kTraced = (std::is_base_of_v<signal, Args> && ...);
	std::function<signal(Args...)> function;
	std::tuple<Args...> inputs;
	Function() {}
	template<class F, typename = std::enable_if_t<std::is_invocable_v<F, Args...>>> Function(F fn) : function(fn) {}
	template<class F, class... O, typename = std::enable_if_t<std::is_invocable_v<F, Args...>>> Function(F fn, O... o) : function(fn) { with(o...); }
	template<class... O> Function& with(O... o) { static_assert(sizeof...(O) + 1 == sizeof...(Args), "with(): all but the first argument"); inputs = std::tuple<Args...>(in, o...); return *this; }
	template<class... FA> Function& operator()(const FA&... a) {
		if constexpr (sizeof...(FA) == sizeof...(Args)) { inputs = std::tuple<Args...>(a...); in = signal(std::get<0>(inputs)); }
		else { static_assert(sizeof...(FA) + 1 == sizeof...(Args), "Function: only the first argument (the input) may be omitted"); inputs = std::tuple<Args...>(in, a...); }
		return *this;
	}
	GraphStub& operator>>(GraphStub& g) { return g; }                                // `f(distort) >> graph;`: the UI plot
	template<class T> T& operator>>(T& dst) { return klang::operator>>(static_cast<Modifier&>(*this), dst); }   // (the GraphStub overload above hides the plain `>>`: an Input takes input(), anything else is assigned)
	using Modifier::input;
protected:
	void input() override { std::get<0>(inputs) = in; }
	void process() override {
		if (!function) { out = 0.f; return; }
		if constexpr (!kTraced) if (gpu::Recorder* r = gpu::recording()) r->fail("a Function<> over plain floats cannot be recorded: compile the patch with -DKLANG_GPU_TRACE_FLOAT (include/klang/klang.h)");
		out = std::apply(function, inputs);
	}
};

// `abs` (klang.h:3065, 3072: a Function<float> object over fabsf that the header's `#define abs klang::abs` puts in std::abs's place): called — `abs(x)` — or streamed
// through — `x >> abs >> y` (Vocoder.k:102)
struct AbsFunction : Modifier {
	signal operator()(const signal& x) const { return abs_of(x); }
protected:
	void process() override { out = abs_of(in); }
};
inline thread_local AbsFunction abs;

// ---- oscillators: host halves only ----
struct Oscillator : Generator {
	Frequency frequency = 1000.f;
	using Generator::set;
	virtual void reset() {}                                   // klang.h:2859 (a user Oscillator's own phase: it has none here, its members do)
	virtual float host_process() { device_only("rendering this oscillator into a Wavetable on the host (supported: Fast::Sine, Basic::Sine / Saw / Triangle / Square)"); }
};
namespace Generators {
	// White noise: one libc rand() per sample (klang.h:4947-4951 Basic, 5357-5366 Fast).  Every Noise object of the process draws from the
	// one libc sequence: the bank produces each block's values ON THE DEVICE from that sequence's state (glibc's generator restated with jump-ahead,
	// klang_amd/csrc/klg_rand.hpp) in the reference's call order (an Effect bank: instance by instance; a Synth bank: sounding note by sounding
	// note, each through the whole block), and the lanes index them.
	struct NoiseBase : Generator {
		int kind;
		explicit NoiseBase(int k) : kind(k) {}
		void process() override {
			if (gpu::Recorder* r = gpu::recording()) {
				out.reg = r->emit(klg::graph::OP_NOISE, -1, -1, -1, (uint32_t)kind, true); return;
			}
			device_only("Noise::process()");
		}
	};
namespace Basic {
	struct Noise : NoiseBase { Noise() : NoiseBase(0) {} };
	// Generic::Oscillator set() (klang.h:2862-2880) + the five Basic waveforms (4899-4944); process() is device code
	struct Osc : Oscillator, gpu::Packable {
		klg::host::BOscH h; float duty_ = 0.5f; int kind;
		explicit Osc(int k) : kind(k) { if (gpu::Sink* r = gpu::constructing()) r->note(this, sizeof(Osc), k, this); }
		using Oscillator::set;
		void reset() override { if (gpu::Recorder* r = gpu::recording()) { r->emit(klg::graph::OP_OSCSET, -1, -1, r->node(this, "Basic oscillator"), 2, false); return; } h.position = 0; }
		void set(param f) override {
			if (gpu::Recorder* r = gpu::recording()) { r->emit(klg::graph::OP_OSCSET, r->reg_of(f), -1, r->node(this, "Basic oscillator"), 0, false); frequency = f; return; }
			h.frequency = f; h.increment = f * 2.f * pi.f / fs.f; frequency = f;
		}
		void set(param f, param phase) override {
			if (gpu::Recorder* r = gpu::recording()) { r->emit(klg::graph::OP_OSCSET, r->reg_of(f), r->reg_of(phase), r->node(this, "Basic oscillator"), 1, false); frequency = f; return; }   // per sample: re-phasing
			h.set(f, phase, host_fs()); frequency = f;
		}
		void set(param f, relative phase) override { set(f); set(phase); }
		void set(relative phase) override { if (gpu::no_set_while_recording("Basic oscillator set(relative)")) return; h.offset = phase.value * (2 * pi); }
		void process() override { if (gpu::Recorder* r = gpu::recording()) { out.reg = r->emit(klg::graph::OP_OSC, -1, -1, r->node(this, "Basic oscillator"), 0, true); return; } device_only("Basic oscillator process()"); }
		float host_process() override {                                      // klang.h:4899-4944 (one cycle into a Wavetable)
			using namespace klg::graph;
			const float two_pi = 2.f * pi.f;
			float y;
			switch (kind) {
			case N_BSINE: y = (float)std::sin((double)(h.position + h.offset)); break;
			case N_BSAW: y = h.position * pi.inv - 1.f; break;
			case N_BTRI: y = std::fabs(2.f * h.position * pi.inv - 2.f) - 1.f; break;
			case N_BSQUARE: y = h.position > pi.f ? 1.f : -1.f; break;
			default: device_only("rendering a Basic::Pulse into a Wavetable on the host");
			}
			if (!(h.increment >= two_pi)) { h.position += h.increment; if (h.position > two_pi) h.position -= two_pi; }
			return y;
		}
		void pack(uint32_t* w) const override { using namespace klg::graph; w[BOSC_INC] = gpu::fbits(h.increment); w[BOSC_POS] = gpu::fbits(h.position); w[BOSC_OFFSET] = gpu::fbits(h.offset); w[BOSC_DUTY] = gpu::fbits(duty_); w[BOSC_FREQ] = gpu::fbits(h.frequency); }
		void unpack(const uint32_t* w) override { using namespace klg::graph; std::memcpy(&h.increment, &w[BOSC_INC], 4); std::memcpy(&h.position, &w[BOSC_POS], 4); std::memcpy(&duty_, &w[BOSC_DUTY], 4); }
	};
	struct Sine : Osc { Sine() : Osc(klg::graph::N_BSINE) {} };
	struct Saw : Osc { Saw() : Osc(klg::graph::N_BSAW) {} };
	struct Triangle : Osc { Triangle() : Osc(klg::graph::N_BTRI) {} };
	struct Square : Osc { Square() : Osc(klg::graph::N_BSQUARE) {} };
	struct Pulse : Osc {
		Pulse() : Osc(klg::graph::N_BPULSE) {}
		using Osc::set;
		void set(param f, param phase, param duty) override {                                             // klang.h:4932-4935
			Osc::set(f, phase);
			if (gpu::Recorder* r = gpu::recording()) { r->emit(klg::graph::OP_OSCSET, r->reg_of(duty), -1, r->node(this, "Basic oscillator"), 3, false); return; }
			duty_ = duty;
		}
	};
}
namespace Fast {
	struct Noise : NoiseBase { Noise() : NoiseBase(1) {} };
	struct Sine : Oscillator, gpu::Packable {
		klg::host::FSineH h;
		Sine() { if (gpu::Sink* r = gpu::constructing()) r->note(this, sizeof(Sine), klg::graph::N_FSINE, this); }
		using Oscillator::set;
		void reset() override { if (gpu::Recorder* r = gpu::recording()) { r->emit(klg::graph::OP_OSCSET, -1, -1, r->node(this, "Fast::Sine"), 2, false); return; } h.pos = 0; }   // klang.h:5136-5140
		void set(param f) override {
			if (gpu::Recorder* r = gpu::recording()) { r->emit(klg::graph::OP_OSCSET, r->reg_of(f), -1, r->node(this, "Fast::Sine"), 0, false); frequency = f; return; }   // per-sample set(f): vibrato / FM
			if (f != h.frequency) { h.frequency = f; h.inc = klg::host::fast_increment(f, host_fs()); }
			frequency = f;
		}
		void set(param f, param phase) override {
			if (gpu::Recorder* r = gpu::recording()) { r->emit(klg::graph::OP_OSCSET, r->reg_of(f), r->reg_of(phase), r->node(this, "Fast::Sine"), 1, false); frequency = f; return; }
			h.set(f, phase, host_fs()); frequency = f;
		}
		void set(param f, relative phase) override { set(f); set(phase); }
		void set(relative) override { device_only("Fast::Sine::set(relative) [phase modulation]"); }
		void process() override { if (gpu::Recorder* r = gpu::recording()) { out.reg = r->emit(klg::graph::OP_OSC, -1, -1, r->node(this, "Fast::Sine"), 0, true); return; } device_only("Fast::Sine::process()"); }
		float host_process() override { const float y = klg::host::fastsinp_host(h.pos); h.pos += (uint32_t)h.inc; return y; }   // klang.h:5165-5171
		void pack(uint32_t* w) const override { w[klg::graph::FSINE_INC] = (uint32_t)h.inc; w[klg::graph::FSINE_POS] = h.pos; w[klg::graph::FSINE_FREQ] = gpu::fbits(h.frequency); }
		void unpack(const uint32_t* w) override { h.inc = (int32_t)w[klg::graph::FSINE_INC]; h.pos = w[klg::graph::FSINE_POS]; std::memcpy(&h.frequency, &w[klg::graph::FSINE_FREQ], 4); }
	};
	struct Osm : Oscillator, gpu::Packable {
		klg::host::OsmH h; int waveform;             // 0 = saw family, 1 = pulse family
		Osm(int wf, float duty) : h(duty), waveform(wf) { if (gpu::Sink* r = gpu::constructing()) r->note(this, sizeof(Osm), wf ? klg::graph::N_PULSE : klg::graph::N_SAW, this); }
		using Oscillator::set;
		void set(param f) override {
			if (gpu::Recorder* r = gpu::recording()) { r->emit(klg::graph::OP_OSCSET, r->reg_of(f), -1, r->node(this, "Fast oscillator"), 0, false); frequency = f; return; }
			if (h.frequency != f) { h.refresh(f, host_fs()); h.init(); }
			frequency = f;
		}
		void set(param f, param phase) override {
			if (gpu::Recorder* r = gpu::recording()) { r->emit(klg::graph::OP_OSCSET, r->reg_of(f), r->reg_of(phase), r->node(this, "Fast oscillator"), 1, false); frequency = f; return; }   // per sample: hard sync
			h.set(f, phase, host_fs()); frequency = f;
		}
		void set(param f, param phase, param duty) override {                  // klang.h:5236-5244; per sample (PWM): set(f, phase) then setDuty — the second init() recomputes all of the first
			if (gpu::Recorder* r = gpu::recording()) { const int n = r->node(this, "Fast oscillator"); r->emit(klg::graph::OP_OSCSET, r->reg_of(f), r->reg_of(phase), n, 1, false); r->emit(klg::graph::OP_OSCSET, r->reg_of(duty), -1, n, 3, false); frequency = f; return; }
			h.set(f, phase, duty, host_fs()); frequency = f;
		}
		void process() override { if (gpu::Recorder* r = gpu::recording()) { out.reg = r->emit(klg::graph::OP_OSC, -1, -1, r->node(this, "Fast oscillator"), 0, true); return; } device_only("Fast::Osm::process()"); }
		void pack(uint32_t* w) const override { using namespace klg::graph; w[OSM_INC] = (uint32_t)h.inc; w[OSM_OFFSET] = h.offset; w[OSM_DUTY] = h.duty; w[OSM_DELTA] = gpu::fbits(h.delta); w[OSM_STATE] = (uint32_t)h.state; w[OSM_FREQ] = gpu::fbits(h.frequency); }
		void unpack(const uint32_t* w) override { using namespace klg::graph; h.inc = (int32_t)w[OSM_INC]; h.offset = w[OSM_OFFSET]; h.duty = w[OSM_DUTY]; h.state = (int)(w[OSM_STATE] & 3u); std::memcpy(&h.delta, &w[OSM_DELTA], 4); std::memcpy(&h.frequency, &w[OSM_FREQ], 4); }
	};
	struct Saw : Osm { Saw() : Osm(0, 0.f) {} };
	struct Triangle : Osm { Triangle() : Osm(0, 1.f) {} };
	struct Square : Osm { Square() : Osm(1, 1.0f) {} };
	struct Pulse : Osm { Pulse() : Osm(1, 0.5f) {} };
}
}

// ---- filters ----
namespace gpu {
// every modifier is recorded the same way: `in >> node` is one op whose result is the node's `out`
template<class M> inline void record_modifier(M* m, const char* what) {
	Recorder* r = recording();
	const int ri = r->reg_of(m->in);
	m->out.reg = r->emit(klg::graph::OP_LPF, ri, -1, r->node(m, what), 0, true);
}
}
namespace Filters {
namespace Biquad {
	// Biquad::Filter (klang.h:5550-5652): one host design per type, one type-independent process() on the device
	struct Filter : Modifier, gpu::Packable {
		klg::host::BiquadLpfH h;
		explicit Filter(int type) { h.type = type; if (gpu::Sink* r = gpu::constructing()) r->note(this, sizeof(Filter), klg::graph::N_LPF, this); }
		void reset() { if (gpu::no_set_while_recording("Biquad filter reset()")) return; h.reset(); }
		using Modifier::set;
		void set(param f) override { set(f, param(h.type == klg::host::BQ_APF ? 1.f : klg::host::ROOT2_INV)); }
		void set(param f, param Q) override {
			if (gpu::Recorder* r = gpu::recording()) {
				if (h.type == klg::host::BQ_APF) { r->fail("Biquad::APF::set() inside process() / prepare() is not recorded (double-precision design)"); return; }
				const int rf = r->reg_of(f), rq = r->reg_of(Q); r->emit(klg::graph::OP_LPFSET, rf, rq, r->node(this, "Biquad filter"), (uint32_t)h.type, false); return;
			}
			h.set(f, Q, host_fs());
		}
		void process() override { if (gpu::recording()) { gpu::record_modifier(this, "Biquad filter"); return; } device_only("Biquad filter process()"); }
		void pack(uint32_t* w) const override { using namespace klg::graph; const float v[LPF_WORDS] = { h.b0, h.b1, h.b2, h.a1, h.a2, h.z0, h.z1, h.f, h.Q }; for (int i = 0; i < LPF_WORDS; i++) w[i] = gpu::fbits(v[i]); }
		void unpack(const uint32_t* w) override { using namespace klg::graph; float* v[LPF_WORDS] = { &h.b0, &h.b1, &h.b2, &h.a1, &h.a2, &h.z0, &h.z1, &h.f, &h.Q }; for (int i = 0; i < LPF_WORDS; i++) std::memcpy(v[i], &w[i], 4); }
	};
	struct LPF : Filter { LPF() : Filter(klg::host::BQ_LPF) {} };
	struct HPF : Filter { HPF() : Filter(klg::host::BQ_HPF) {} };
	struct BPF : Filter {                                                       // klang.h:5689-5730
		enum Gain { ConstantSkirtGain, ConstantPeakGain };
		BPF() : Filter(klg::host::BQ_BPF_PEAK) {}
		BPF& operator=(Gain g) { h.type = g == ConstantSkirtGain ? klg::host::BQ_BPF_SKIRT : klg::host::BQ_BPF_PEAK; h.init(host_fs()); return *this; }
	};
	struct BRF : Filter { BRF() : Filter(klg::host::BQ_BRF) {} };
	struct APF : Filter { APF() : Filter(klg::host::BQ_APF) {} };
	typedef LPF HCF; typedef LPF HRF; typedef HPF LCF; typedef HPF LRF; typedef BRF BSF;
}
	// Filters::DCF klang.h:5386-5397
	struct DCF : Modifier, gpu::Packable {
		float r = 0.995f, z = 0;
		DCF() { if (gpu::Sink* rr = gpu::constructing()) rr->note(this, sizeof(DCF), klg::graph::N_DCF, this); }
		using Modifier::set;
		void set(float r_) { r = r_; }
		void process() override { if (gpu::recording()) { gpu::record_modifier(this, "DCF"); return; } device_only("DCF::process()"); }
		void pack(uint32_t* w) const override { w[klg::graph::DCF_R] = gpu::fbits(r); w[klg::graph::DCF_Z] = gpu::fbits(z); w[klg::graph::DCF_OUT] = gpu::fbits(out.value); }
		void unpack(const uint32_t* w) override { std::memcpy(&z, &w[klg::graph::DCF_Z], 4); std::memcpy(&out.value, &w[klg::graph::DCF_OUT], 4); }
	};
	// Filters::IIR<ORDER> klang.h:5399-5432: `out = in - sum a[i] * y[i]` over the last ORDER outputs (node kind iirn, ORDER 2..8)
	template<int ORDER> struct IIR : Modifier, gpu::Packable {
		static_assert(ORDER >= 2 && ORDER <= 8, "IIR<ORDER>: orders 2..8 are recorded (IIR<1> is its own node)");
		float a[ORDER] = { };
		float y[ORDER] = { };
		IIR() { if (gpu::Sink* r = gpu::constructing()) r->note(this, sizeof(IIR<ORDER>), klg::graph::N_IIRN, this, ORDER); }
		template<typename... Coeffs> void set(Coeffs... coeffs) {
			static_assert(sizeof...(coeffs) == ORDER, "Incorrect number of coefficients.");
			if (gpu::no_set_while_recording("IIR<ORDER>::set()")) return;
			const float c[ORDER] = { (float)coeffs... };
			for (int i = 0; i < ORDER; i++) a[i] = c[i];
		}
		void process() override { if (gpu::recording()) { gpu::record_modifier(this, "IIR<ORDER>"); return; } device_only("IIR<ORDER>::process()"); }
		void pack(uint32_t* w) const override { for (int i = 0; i < ORDER; i++) { w[i] = gpu::fbits(a[i]); w[ORDER + i] = gpu::fbits(y[i]); } }
		void unpack(const uint32_t* w) override { for (int i = 0; i < ORDER; i++) std::memcpy(&y[i], &w[ORDER + i], 4); }
	};
	template<> struct IIR<1> : Modifier, gpu::Packable {
		float a = 1, b = 0;
		IIR() { if (gpu::Sink* r = gpu::constructing()) r->note(this, sizeof(IIR<1>), klg::graph::N_IIR1, this); }
		using Modifier::set;
		void set(param coeff) override { if (gpu::no_set_while_recording("IIR<1>::set()")) return; a = coeff; b = 1.f - a; }
		void process() override { if (gpu::recording()) { gpu::record_modifier(this, "IIR<1>"); return; } device_only("IIR<1>::process()"); }
		void pack(uint32_t* w) const override { w[klg::graph::IIR1_A] = gpu::fbits(a); w[klg::graph::IIR1_B] = gpu::fbits(b); w[klg::graph::IIR1_OUT] = gpu::fbits(out.value); }
		void unpack(const uint32_t* w) override { std::memcpy(&out.value, &w[klg::graph::IIR1_OUT], 4); }
	};
namespace OnePole {
	// OnePole::Filter / LPF / HPF klang.h:5470-5543
	struct Filter : Modifier, gpu::Packable {
		klg::host::OnePoleH h;
		explicit Filter(bool hpf) { h.hpf = hpf; if (gpu::Sink* r = gpu::constructing()) r->note(this, sizeof(Filter), hpf ? klg::graph::N_OPHPF : klg::graph::N_OPLPF, this); }
		void reset() { if (gpu::no_set_while_recording("OnePole reset()")) return; h.reset(); }
		using Modifier::set;
		void set(param f) override { if (gpu::no_set_while_recording("OnePole set(f)")) return; h.set(f, host_fs()); }
		void process() override { if (gpu::recording()) { gpu::record_modifier(this, "OnePole filter"); return; } device_only("OnePole filter process()"); }
		void pack(uint32_t* w) const override { using namespace klg::graph; w[OP1_B0] = gpu::fbits(h.b0); w[OP1_B1] = gpu::fbits(h.b1); w[OP1_A1] = gpu::fbits(h.a1); w[OP1_Z] = gpu::fbits(h.z); w[OP1_OUT] = gpu::fbits(out.value); }
		void unpack(const uint32_t* w) override { std::memcpy(&h.z, &w[klg::graph::OP1_Z], 4); std::memcpy(&out.value, &w[klg::graph::OP1_OUT], 4); }
	};
	struct LPF : Filter { LPF() : Filter(false) {} };
	struct HPF : Filter { HPF() : Filter(true) {} };
}
namespace Butterworth {
	template<int ORDER> struct LPF;
	template<> struct LPF<1> : Modifier, gpu::Packable {                        // klang.h:5786-5799
		klg::host::Butter1H h;
		LPF() { if (gpu::Sink* r = gpu::constructing()) r->note(this, sizeof(LPF<1>), klg::graph::N_BUTTER1, this); }
		using Modifier::set;
		void set(param f) override { if (gpu::no_set_while_recording("Butterworth::LPF<1>::set()")) return; h.set(f, host_fs()); }
		void process() override { if (gpu::recording()) { gpu::record_modifier(this, "Butterworth::LPF<1>"); return; } device_only("Butterworth::LPF<1>::process()"); }
		void pack(uint32_t* w) const override { using namespace klg::graph; w[BW1_B0] = gpu::fbits(h.b0); w[BW1_A1] = gpu::fbits(h.a1); w[BW1_Z] = gpu::fbits(h.z); w[BW1_OUT] = gpu::fbits(out.value); }
		void unpack(const uint32_t* w) override { std::memcpy(&h.z, &w[klg::graph::BW1_Z], 4); std::memcpy(&out.value, &w[klg::graph::BW1_OUT], 4); }
	};
	template<> struct LPF<2> : Biquad::Filter { LPF() : Biquad::Filter(klg::host::BQ_BUTTER2) {} };   // klang.h:5801-5811
}
}
namespace Modifiers {
	// Modifiers::Modal klang.h:5815-5859
	struct Modal : Modifier, gpu::Packable {
		klg::host::ModalH h;
		Modal() { if (gpu::Sink* r = gpu::constructing()) r->note(this, sizeof(Modal), klg::graph::N_MODAL, this); }
		using Modifier::set;
		void set(param f, param decay) override { if (gpu::no_set_while_recording("Modal::set()")) return; h.set(f, decay, host_fs()); in = 0; out = 0; }
		void set(param f, param decay, param gain) override { if (gpu::no_set_while_recording("Modal::set()")) return; h.set(f, decay, gain.value, host_fs()); in = 0; out = 0; }
		void process() override { if (gpu::recording()) { gpu::record_modifier(this, "Modal"); return; } device_only("Modal::process()"); }
		void pack(uint32_t* w) const override { using namespace klg::graph; w[MODAL_A1] = gpu::fbits(h.a1); w[MODAL_A2] = gpu::fbits(h.a2); w[MODAL_Y1] = gpu::fbits(h.y1); w[MODAL_Y2] = gpu::fbits(h.y2); w[MODAL_GAIN] = gpu::fbits(h.gain); }
		void unpack(const uint32_t* w) override { std::memcpy(&h.y1, &w[klg::graph::MODAL_Y1], 4); std::memcpy(&h.y2, &w[klg::graph::MODAL_Y2], 4); }
	};
}
enum Mode { Peak, RMS, Mean };

// ---- Envelope / ADSR (klang.h:3722-4137) ----
// Points: any number on the host (the reference keeps a std::vector<Point>, klang.h:4093).  A lane record holds `capacity` point slots, fixed when the
// program is recorded: KLANG_GPU_ENV_POINTS (default 16; 4 .. 128), or more when the member is given more points while its owner is constructed
// (`Envelope env = { ... }`, a constructor body).  An envelope that holds more points than its record when a note starts STOPS with a message that
// names the macro — nothing is ever truncated.
#ifndef KLANG_GPU_ENV_POINTS
#define KLANG_GPU_ENV_POINTS 16
#endif
static_assert(KLANG_GPU_ENV_POINTS >= 4 && KLANG_GPU_ENV_POINTS <= klg::graph::ENV_MAX_POINTS, "KLANG_GPU_ENV_POINTS: 4 .. 128 point slots per Envelope record");
struct Envelope : Generator, gpu::Packable {
	struct Follower;
	struct Point { float x, y; Point() : x(0), y(0) {} template<class A, class B> Point(A a, B b) : x(float(a)), y(float(b)) {} };
	// Envelope::Points(0, 1)(1, 0)... (klang.h:3818-3851): the inline point list
	struct Points : Point {
		std::vector<Point> rest;
		Points(float x_, float y_) { x = x_; y = y_; }
		Points& operator()(float x_, float y_) { rest.push_back(Point(x_, y_)); return *this; }
		int count() const { return 1 + (int)rest.size(); }
	};
	// Ramp / Linear (klang.h:3731-3807): what an Envelope steps from point to point.  On the GPU an envelope's ramp IS the linear one (klg_device.hpp Env); the types are
	// here so that patches that name them compile and behave on the host as in the reference, `env.set(new Envelope::Linear())` is what it is there (the default ramp
	// again + initialise()), and a USER ramp — a subclass with its own operator++ — stops with a message instead of rendering a line.
	struct Ramp : Generator {
		float target = 1.f, rate = 0.f; bool active = false;
		Ramp(float value = 1.f) { setValue(value); }
		Ramp(float start, float target_, float time) { setValue(start); setTarget(target_); setTime(time); }
		virtual ~Ramp() {}
		bool isActive() const { return active; }
		virtual void setTarget(float target_) { target = target_; active = (out.value != target_); }
		virtual void setValue(float value) { out.value = value; target = value; active = false; }
		virtual void setRate(float rate_) { rate = rate_; }
		virtual void setTime(float time) { rate = time ? 1.f / (time * fs.f) : 0.f; }
		virtual signal operator++(int) = 0;
		void process() override {}
	};
	struct Linear : Ramp {
		using Ramp::Ramp;
		signal operator++(int) override {                                              // klang.h:3784-3806
			const signal output = out;
			if (active) {
				if (target > out.value) { out.value += rate; if (out.value >= target) { out.value = target; active = false; } }
				else { out.value -= rate; if (out.value <= target) { out.value = target; active = false; } }
			}
			return output;
		}
	};
	void set(Ramp* ramp) {                                                             // klang.h:4057-4060 (takes ownership, then initialise())
		std::unique_ptr<Ramp> own(ramp);
		if (!ramp || typeid(*ramp) != typeid(Linear)) {
			std::fprintf(stderr, "klang-mi355: Envelope::set(Ramp*) with a user-defined Ramp: the device envelope steps the Linear ramp (klang.h:3781-3807) and has no form for another operator++ — not rendered as a line instead\n");
			std::abort();
		}
		initialise();
	}
	enum Stage { Sustain, Release, Off };
	enum Mode { Time, Rate };
	klg::host::EnvH h;
	int capacity = KLANG_GPU_ENV_POINTS;                                           // point slots of this member's lane record (read when the program is finished)
	void reg_member() { if (gpu::Sink* r = gpu::constructing()) r->note(this, sizeof(Envelope), klg::graph::N_ENV, this, 0, nullptr, &capacity); }
	Envelope() { reg_member(); const float one[2] = { 0.f, 1.f }; h.set_points(1, one, host_fs()); }
	Envelope(const Points& p) { reg_member(); set(p); }
	Envelope(std::initializer_list<Point> p) { reg_member(); assign(p.begin(), (int)p.size()); }
	Envelope(const Envelope& in) : Generator(), gpu::Packable() { reg_member(); const std::vector<Point> p = in.points(); assign(p.data(), (int)p.size()); }   // klang.h:3879: the POINTS of the other envelope, initialised afresh (Time mode, no loop)
	Envelope& operator=(const Envelope& in) { h = in.h; out.value = in.out.value; grow(h.npoints); return *this; }
	Envelope& operator=(std::initializer_list<Point> p) { assign(p.begin(), (int)p.size()); return *this; }
	std::vector<Point> points() const { std::vector<Point> p((size_t)h.npoints); for (int i = 0; i < h.npoints; i++) p[(size_t)i] = Point(h.px[(size_t)i], h.py[(size_t)i]); return p; }
	void grow(int n) { if (n > capacity && gpu::constructing()) capacity = n; }   // (only while the owner is constructed: every note of a type then has the same record)
	void assign(const Point* p, int n) {                                          // set(points) + initialise()  klang.h:3893-3896, 3974-3989
		if (gpu::no_set_while_recording("Envelope::set(points)")) return;
		std::vector<float> xy((size_t)(2 * n + 2));
		for (int i = 0; i < n; i++) { xy[(size_t)(2 * i)] = p[i].x; xy[(size_t)(2 * i + 1)] = p[i].y; }
		h.set_points(n, xy.data(), host_fs());
		out.value = h.r_out;
		grow(n);
	}
	using Generator::set;
	void set(const std::vector<Point>& p) { assign(p.data(), (int)p.size()); }                                    // klang.h:3893-3896
	void set(const Points& p) { std::vector<Point> v; v.push_back(p); v.insert(v.end(), p.rest.begin(), p.rest.end()); assign(v.data(), (int)v.size()); }   // 3899-3909
	void initialise() { if (gpu::no_set_while_recording("Envelope::initialise()")) return; h.initialise(host_fs()); out.value = h.r_out; }   // 3974-3989
	void sequence() {                                                              // relative -> absolute times  klang.h:3912-3920
		if (gpu::no_set_while_recording("Envelope::sequence()")) return;
		float time = 0.f;
		for (int i = 0; i < h.npoints; i++) { const float delta = h.px[(size_t)i]; time += delta + 0.00001f; h.px[(size_t)i] = time; }
		h.initialise(host_fs()); out.value = h.r_out;
	}
	void resize(float length) {                                                    // klang.h:3992-4005 (as written there: length / (fs * old length))
		if (gpu::no_set_while_recording("Envelope::resize()")) return;
		const float old_length = getLength();
		if (old_length == 0.0) return;
		const float multiplier = length / (fs.f * old_length);
		for (int i = 0; i < h.npoints; i++) h.px[(size_t)i] *= multiplier;
		h.initialise(host_fs()); out.value = h.r_out;
	}
	float getLength() const { return h.npoints ? h.px[(size_t)(h.npoints - 1)] : 0.f; }   // klang.h:3958
	const Point operator[](int i) const { return Point(h.px[(size_t)i], h.py[(size_t)i]); }   // klang.h:4054-4056
	// value at a time in seconds, klang.h:3929-3943 — host arithmetic on the points (a lookup table such as SynTHX.k's `transposition.at(x)`)
	signal at(param time) const {
		if (time.reg >= 0 && gpu::recording()) { gpu::recording()->fail("Envelope::at(t) of a value computed inside process() is not supported in a recorded graph"); return signal(0.f); }
		if (h.npoints == 0) return 0;
		float lx = 0.f, ly = h.py[0];
		for (int i = 0; i < h.npoints; i++) {
			const float x = h.px[(size_t)i], y = h.py[(size_t)i];
			if (x >= time.value) { const float dx = x - lx, dy = y - ly, t = time.value - lx; return dx == 0 ? ly : (ly + t * dy / dx); }
			lx = x; ly = y;
		}
		return h.py[(size_t)(h.npoints - 1)];
	}
	void setMode(Mode m) { if (gpu::no_set_while_recording("Envelope::setMode()")) return; h.rate_mode = (m == Rate); }   // klang.h:4064-4071 (takes effect at the next setTarget, as there)
	Mode mode() const { return h.rate_mode ? Rate : Time; }
	void setStage(Stage st) { if (gpu::no_set_while_recording("Envelope::setStage()")) return; h.stage = (int)st; }   // klang.h:3952
	Stage getStage() const { return (Stage)h.stage; }                              // (host state: in a recorded process() ask `env == Envelope::Off` / finished())
	void setTarget(const Point& p, float time = 0.f) { if (gpu::no_set_while_recording("Envelope::setTarget()")) return; h.set_target(p.x, p.y, time, host_fs()); }   // klang.h:4008-4010
	virtual void release(float time, float level = 0.f) { if (gpu::no_set_while_recording("Envelope::release()")) return; h.stage = klg::ENV_RELEASE; h.set_target(time, level, 0.f, host_fs()); }   // klang.h:3961-3966
	void setLoop(int startPoint, int endPoint) { if (gpu::no_set_while_recording("Envelope::setLoop()")) return; h.set_loop(startPoint, endPoint); }   // klang.h:3923-3926
	void resetLoop() {                                                             // klang.h:3946-3950
		if (gpu::no_set_while_recording("Envelope::resetLoop()")) return;
		h.loop_start = h.loop_end = -1;
		if (h.stage == klg::ENV_SUSTAIN && (h.point + 1) < h.npoints) h.set_target(h.px[(size_t)(h.point + 1)], h.py[(size_t)(h.point + 1)], h.px[(size_t)h.point], host_fs());
	}
	// klang.h:4094.  In a recorded process() the test is a VALUE (envoff): `if (env.finished()) stop();`, `if (env.finished()) { ...; stop(); return; }`,
	// `!env.finished()`, `env.finished() && x > y` all record; the plain stop() idiom is folded back into `stopif` (gpu::fold_stop_idiom)
	gpu::Pred stage_is(Stage st) const {
		if (gpu::Recorder* r = gpu::recording()) return gpu::Pred{ h.stage == (int)st, r->emit(klg::graph::OP_ENVOFF, -1, -1, r->node(this, "Envelope"), st == Off ? 0u : (st == Sustain ? 1u : 2u), true) };
		return gpu::Pred{ h.stage == (int)st, -1 };
	}
	gpu::Pred finished() const { return stage_is(Off); }
	gpu::Pred operator==(Stage st) const { return stage_is(st); }                  // klang.h:3883
	gpu::Pred operator!=(Stage st) const { return !stage_is(st); }                 // klang.h:3884
	signal& operator++(int) { this->process(); return out; }
	void process() override { if (gpu::Recorder* r = gpu::recording()) { out.reg = r->emit(klg::graph::OP_ENV, -1, -1, r->node(this, "Envelope"), 0, true); return; } device_only("Envelope::process()"); }
	[[noreturn]] void too_many_points() const {
		std::fprintf(stderr, "klang-mi355: an Envelope holds %d points but its lane record was recorded with %d point slots.  Give the member its points while the note is constructed, or compile with "
			"-DKLANG_GPU_ENV_POINTS=%d (at most %d).\n", h.npoints, capacity, h.npoints, (int)klg::graph::ENV_MAX_POINTS);
		std::abort();
	}
	void pack(uint32_t* w) const override {
		using namespace klg::graph;
		if (h.npoints > capacity || h.npoints > (int)ENV_MAX_POINTS) too_many_points();
		w[ENV_OUT] = gpu::fbits(h.r_out); w[ENV_TARGET] = gpu::fbits(h.r_target); w[ENV_RATE] = gpu::fbits(h.r_rate); w[ENV_TIME] = gpu::fbits(h.time); w[ENV_NPOINTS] = (uint32_t)h.npoints;
		w[ENV_BITS] = env_bits(h.stage, h.point, h.active, h.rate_mode);
		w[ENV_LOOP] = (uint32_t)(h.loop_start & 0xFF) | ((uint32_t)(h.loop_end & 0xFF) << 8);
		for (int i = 0; i < 4; i++) { w[ENV_PX + i] = gpu::fbits(h.px[(size_t)i]); w[ENV_PY + i] = gpu::fbits(h.py[(size_t)i]); }
		const int ext = env_capacity(capacity) - 4;                                // x of points 4.., then their y (zeros behind the last point)
		for (int i = 0; i < ext; i++) { const bool has = 4 + i < h.npoints; w[ENV_WORDS + i] = has ? gpu::fbits(h.px[(size_t)(4 + i)]) : 0u; w[ENV_WORDS + ext + i] = has ? gpu::fbits(h.py[(size_t)(4 + i)]) : 0u; }
	}
	void unpack(const uint32_t* w) override {
		std::memcpy(&h.r_out, &w[0], 4); std::memcpy(&h.r_target, &w[1], 4); std::memcpy(&h.r_rate, &w[2], 4); std::memcpy(&h.time, &w[3], 4);
		const uint32_t bits = w[4]; h.stage = (int)(bits & 3u); h.point = klg::graph::env_bits_point(bits); h.active = ((bits >> 5) & 1u) != 0;
	}
};
struct ADSR : Envelope {
	klg::host::AdsrH a;
	enum Mode { Time, Rate }; Mode mode = (Mode)0;                                     // (klang.h:4112: a member the reference declares and never reads; Envelope::setMode is what changes the ramps)
	ADSR() { if (gpu::Sink* r = gpu::constructing()) r->note(static_cast<Envelope*>(this), sizeof(ADSR), klg::graph::N_ADSR, this); set(0.5, 0.5, 1, 0.5); }
	using Envelope::set;
	float A = 0.f, D = 0.f, S = 0.f, R = 0.f;                                  // klang.h:4100-4103: the envelope's times as set() keeps them (`if (env.R > 0.01) env.release();`, Modular.k:93 — event code)
	void set(param attack, param decay, param sustain, param release) override { if (gpu::no_set_while_recording("ADSR::set()")) return; const bool rate = h.rate_mode; a.env.rate_mode = rate; a.set(attack, decay, sustain, release, host_fs()); h = a.env; A = a.A; D = a.D; S = a.S; R = a.R; }
	// The ADSR lane record is the shape ADSR::set() gives the envelope — (0,0) (A,1) (A+D,S), loop (2,2), Time mode — with A, A+D, S, R as its words.  An ADSR object
	// that was given other points / another loop / Rate mode through the Envelope interface has no such record: stop, do not play something else.
	void pack(uint32_t* w) const override {
		using namespace klg::graph;
		auto same = [](float x, float y) { return gpu::fbits(x) == gpu::fbits(y); };
		const bool shaped = !h.rate_mode && h.npoints == 3 && same(h.px[0], 0.f) && same(h.py[0], 0.f) && same(h.px[1], a.A) && same(h.py[1], 1.f) && same(h.py[2], a.S) && (h.stage != klg::ENV_SUSTAIN || (h.loop_start == 2 && h.loop_end == 2));
		if (!shaped) { std::fprintf(stderr, "klang-mi355: an ADSR whose points, loop or mode were changed through the Envelope interface (operator=, setLoop, resetLoop, setMode(Rate)) has no GPU record: use an Envelope member for that shape\n"); std::abort(); }
		w[ADSR_OUT] = gpu::fbits(h.r_out); w[ADSR_TARGET] = gpu::fbits(h.r_target); w[ADSR_RATE] = gpu::fbits(h.r_rate); w[ADSR_TIME] = gpu::fbits(h.time); w[ADSR_BITS] = h.bits();
		w[ADSR_A] = gpu::fbits(a.A); w[ADSR_AD] = gpu::fbits(h.px[2]); w[ADSR_S] = gpu::fbits(a.S); w[ADSR_R] = gpu::fbits(a.R);
	}
	void release(float time = 0.f, float level = 0.f) override { Envelope::release(time ? time : a.R, level); }
	gpu::Pred operator==(Envelope::Stage st) const { return stage_is(st); }    // klang.h:4135
};

// ---- Envelope::Follower (klang.h:5862-5903): the AR smoother with abs / square-sqrt around it ----
struct Envelope::Follower : Modifier, gpu::Packable {
	klg::host::FollowerArH ar; klang::Mode mode = RMS;
	Follower() { if (gpu::Sink* r = gpu::constructing()) r->note(this, sizeof(Follower), klg::graph::N_FOLLOWRMS, this); set(0.01f, 0.1f); }
	using Modifier::set;
	void set(param attack, param release) override { if (gpu::no_set_while_recording("Envelope::Follower::set()")) return; ar.set(attack, release, host_fs()); }
	Follower& operator=(klang::Mode m) {                                              // klang.h:5892-5895 (choose before the Synth is created: the mode is part of the recorded program)
		mode = m;
		if (gpu::Sink* r = gpu::rec ? static_cast<gpu::Sink*>(gpu::rec) : static_cast<gpu::Sink*>(gpu::log_target)) r->note(this, sizeof(Follower), m == RMS ? klg::graph::N_FOLLOWRMS : klg::graph::N_FOLLOWPEAK, this);
		return *this;
	}
	void process() override { if (gpu::recording()) { gpu::record_modifier(this, "Envelope::Follower"); return; } device_only("Envelope::Follower::process()"); }
	void pack(uint32_t* w) const override { w[klg::graph::FOLLOW_A] = gpu::fbits(ar.A); w[klg::graph::FOLLOW_R] = gpu::fbits(ar.R); w[klg::graph::FOLLOW_OUT] = gpu::fbits(out.value); }
	void unpack(const uint32_t* w) override { std::memcpy(&out.value, &w[klg::graph::FOLLOW_OUT], 4); }
	// (Envelope::Follower::Window<N>, klang.h:5905-5949, is not here: the reference's own template cannot be instantiated — `sum * window.inv >> sqrt` is an ambiguous
	//  `Function<float> << const double` (klang.h:5946 through 4888; clang 19, the compiler every fixture of tests/golden was made with) — so no patch can hold one
	//  and there is no reference output to be identical to.)
};

// ---- FM operator (klang.h:4140-4180) ----
template<class OSC> struct Operator : OSC, Input {
	Envelope env; Amplitude amp = 1.f;
	Operator() { if (gpu::Sink* r = gpu::constructing()) r->note(static_cast<OSC*>(this), sizeof(Operator), klg::graph::N_OPERATOR, static_cast<gpu::Packable*>(static_cast<OSC*>(this)), 0, nullptr, &env.capacity); }   // (the node's argument: its envelope's point slots)
	Operator& operator()(param f) { OSC::set(f); return *this; }
	Operator& operator()(param f, relative phase) { OSC::set(f, phase); return *this; }
	// klang.h:4149-4157: `op = { {0,0}, {3,1} }` builds a TEMPORARY Envelope and copy-assigns it — points, state AND mode: an operator envelope is back in Time mode
	// after every assignment (set Rate mode after it, then initialise())
	Operator& operator=(std::initializer_list<Envelope::Point> p) { env.h.rate_mode = false; env = p; return *this; }
	Operator& operator=(const Envelope::Points& p) { env.h.rate_mode = false; env.set(p); return *this; }
	Operator& operator=(const Envelope& e) { env = e; return *this; }
	Operator& operator*(signal a) { amp = a; return *this; }                     // (while recording, `a` may be a recorded value: the operator's amp operand)
	template<class S, typename = std::enable_if_t<std::is_base_of_v<signal, S>>> Operator& operator*(const S& a) { amp = static_cast<const signal&>(a); return *this; }   // exact for param / Frequency / ...
	Operator& operator>>(Operator& carrier) { carrier << *this; return carrier; }
	void process() override {
		if (gpu::Recorder* r = gpu::recording()) {
			const int ri = in.reg, ra = amp.reg;                                  // a concrete modulator / amp stays what on() left in the record
			const int rin = (ri >= 0 || in.value != 0.f) ? r->reg_of(in) : -1;
			this->out.reg = r->emit(klg::graph::OP_OPERATOR, rin, ra, r->node(static_cast<OSC*>(this), "Operator"), 0, true);
			return;
		}
		device_only("Operator::process()");
	}
	void pack(uint32_t* w) const override {
		using namespace klg::graph;
		w[OPER_INC] = (uint32_t)this->h.inc; w[OPER_POS] = this->h.pos; w[OPER_FREQ] = gpu::fbits(this->h.frequency); w[OPER_AMP] = gpu::fbits(amp.value);
		env.pack(w + OPER_ENV);
	}
	void unpack(const uint32_t* w) override {
		using namespace klg::graph;
		this->h.pos = w[OPER_POS]; std::memcpy(&amp.value, &w[OPER_AMP], 4);
		env.unpack(w + OPER_ENV);
	}
};
template<class OSC> inline const signal& operator>>(signal modulator, Operator<OSC>& carrier) { carrier << modulator; return carrier; }   // klang.h:4176-4180

// ---- Table / graph: host-side conveniences a patch's on() may touch (klang.h:3303-3378, 2943-3024); no device role ----
template<typename TYPE> struct Result {
	TYPE* y; int i = 0; TYPE sum = 0;
	Result(TYPE* array, int index) : y(&array[index]), i(index) {}
	TYPE& operator[](int index) { return *(y + index); }
	operator TYPE const() { return *y; }
	Result& operator=(const TYPE& in) { *y = in; return *this; }
	TYPE& operator++(int) { i++; return *++y; }
};
#define FUNCTION(type) (void(*)(type, klang::Result<type>&))[](type x, klang::Result<type>& y)
// klang::buffer (klang.h:1983-2136): a cursor over caller-owned samples (`buffer(float*, int)`: what a host hands to
// Effect::process(buffer) / Note::process(buffer)) or over an owned array padded to the next power of two (`buffer(int size)`).  In the
// reference a signal IS a float, so the cursor hands out `signal&` into the sample array; here a signal also carries its recording
// register, so element access goes through `sample`, a reference-like proxy with the same reads, writes and compound assignments
// (`buffer++ = out`, `buffer += x`, `buffer[i] = y`, `signal s = buffer`).
class buffer {
protected:
	static constexpr unsigned capacity(unsigned n) { n--; n |= n >> 1; n |= n >> 2; n |= n >> 4; n |= n >> 8; n |= n >> 16; return n + 1; }    // klang.h:1986-1997
	unsigned mask = 0xFFFFFFFFu; bool owned = false;
	float* samples; float* ptr; float* end;
public:
	struct sample {                                                            // `signal&` of the reference: one element of the array
		float* p;
		operator signal() const { return signal(*p); }
		operator float() const { return *p; }
		sample& operator=(const signal& in) { *p = in.value; return *this; }
		sample& operator=(const sample& in) { *p = *in.p; return *this; }
		sample& operator=(float in) { *p = in; return *this; }
		sample& operator+=(const signal& in) { *p += in.value; return *this; }
		sample& operator-=(const signal& in) { *p -= in.value; return *this; }
		sample& operator*=(const signal& in) { *p *= in.value; return *this; }
		sample& operator/=(const signal& in) { *p /= in.value; return *this; }
	};
	typedef klang::signal signal;
	const int size;
	buffer(float* data, int size_) : samples(data), size(size_) { rewind(); }                                            // klang.h:2007-2010 (not owned: mask stays all ones)
	buffer(float* data, int size_, float initial) : samples(data), size(size_) { rewind(); set(initial); }
	buffer(int size_ = 1, float initial = 0) : mask(capacity((unsigned)size_) - 1u), owned(true), samples(new float[capacity((unsigned)size_)]), size(size_) { rewind(); set(initial); }
	// a copy is a view of the same samples at the same cursor (process(buffer) takes its buffer BY VALUE: every note restarts where the caller's cursor stands)
	buffer(const buffer& b) : mask(b.mask), owned(false), samples(b.samples), ptr(b.ptr), end(b.end), size(b.size) {}
	virtual ~buffer() { if (owned) delete[] samples; }
	void attach(const buffer& b, int n = 0) { samples = b.samples; rewind(); end = samples + n; }
	void rewind(int offset = 0) { ptr = samples + offset; end = samples + size; }                                       // klang.h:2035-2041
	void clear() { std::memset(samples, 0, sizeof(float) * (size_t)size); }
	void clear(int n) { std::memset(samples, 0, sizeof(float) * (size_t)(n < size ? n : size)); }
	int offset() const { return int(samples - ptr); }                                                                   // (the reference's sign: klang.h:2051-2053)
	void set(float value = 0) { if (value == 0) clear(); else for (int i = 0; i < size; i++) samples[i] = value; }
	sample operator[](int i) { return sample{ &samples[(unsigned)i & mask] }; }                                         // klang.h:2062-2064 (masked)
	float operator[](int i) const { return samples[i]; }                                                                // klang.h:2080-2082
	signal operator[](float o) const {                                                                                  // klang.h:2070-2078 (linear read, wraps at size - 1)
		const float f = std::floor(o), frac = o - f;
		const int i = (int)o, j = (i == size - 1) ? 0 : i + 1;
		return signal(samples[i] * (1.f - frac) + samples[j] * frac);
	}
	operator signal() const { return signal(*ptr); }                                                                    // klang.h:2084-2090
	explicit operator double() const { return *ptr; }
	bool finished() const { return ptr == end; }                                                                        // klang.h:2096
	sample operator++(int) { return sample{ ptr++ }; }                                                                  // klang.h:2100-2102
	sample operator=(const signal& in) { *ptr = in.value; return sample{ ptr }; }                                       // klang.h:2104-2106
	sample operator+=(const signal& in) { *ptr += in.value; return sample{ ptr }; }
	sample operator*=(const signal& in) { *ptr *= in.value; return sample{ ptr }; }
	buffer& operator=(const buffer& in) { std::memcpy(samples, in.samples, (size_t)(size < in.size ? size : in.size) * sizeof(float)); return *this; }   // klang.h:2118-2121
	buffer& operator<<(const signal& in) { *ptr = in.value; return *this; }
	float* data() { return samples; } const float* data() const { return samples; }
	float* cursor() const { return ptr; } int remaining() const { return int(end - ptr); }                              // (what the block entry points hand to the GPU)
	void advance(int n) { ptr += n; }
};
// ---- Wavetable / Sample (klang.h:3626-3720): the samples live in HBM (klg_table_upload); a note's record names them by id ----
namespace gpu { inline thread_local klg_synth* upload_target = nullptr; }   // the bank a voice record is being packed for (SynthCore sets it)
class Wavetable : public Oscillator, public gpu::Packable {
protected:
	std::vector<signal> samples; int size;
	float increment = 0.f, position = 0.f, offset = 0.f;
	mutable int table_id = -1; mutable bool dirty = true;
	void announce() { if (gpu::Sink* r = gpu::constructing()) r->note(this, sizeof(Wavetable), klg::graph::N_WAVETABLE, this); }
public:
	using Oscillator::set;
	Wavetable(int size_ = 2048) : samples((size_t)size_), size(size_) { announce(); }
	template<typename TYPE, std::enable_if_t<std::is_base_of_v<Oscillator, TYPE>, int> = 0>
	Wavetable(TYPE oscillator, int size_ = 2048) : samples((size_t)size_), size(size_) { announce(); operator=(oscillator); }
	signal& operator[](int index) { dirty = true; return samples[(size_t)index]; }
	template<typename TYPE, std::enable_if_t<std::is_base_of_v<Oscillator, TYPE>, int> = 0>
	Wavetable& operator=(TYPE& oscillator) {                                  // one cycle of the oscillator klang.h:3646-3651
		oscillator.set(param(fs.f / (float)size));
		for (int s = 0; s < size; s++) samples[(size_t)s] = oscillator.host_process();
		dirty = true; return *this;
	}
	void set(param f) override {                                             // klang.h:3655-3658
		if (gpu::Recorder* r = gpu::recording()) { r->emit(klg::graph::OP_OSCSET, r->reg_of(f), -1, r->node(this, "Wavetable"), 0, false); frequency = f; return; }
		frequency = f; increment = f * ((float)size / fs.f);
	}
	void set(param f, param phase) override { if (gpu::no_set_while_recording("Wavetable::set(f, phase)")) return; position = phase * float(size); set(f); }
	void set(relative phase) override { if (gpu::no_set_while_recording("Wavetable::set(relative)")) return; offset = phase.value * float(size); }
	void set(param f, relative phase) override { set(f); set(phase); }
	void process() override { if (gpu::Recorder* r = gpu::recording()) { out.reg = r->emit(klg::graph::OP_OSC, -1, -1, r->node(this, "Wavetable"), 0, true); return; } device_only("Wavetable::process()"); }
	void pack(uint32_t* w) const override {
		using namespace klg::graph;
		if (dirty || table_id < 0) {
			if (!gpu::upload_target) { std::fprintf(stderr, "klang-mi355: a Wavetable can only be packed for a GPU bank\n"); std::abort(); }
			std::vector<float> f((size_t)size); for (int s = 0; s < size; s++) f[(size_t)s] = samples[(size_t)s].value;
			table_id = klg_table_upload(gpu::upload_target, f.data(), size, 1);
			if (table_id < 0) { std::fprintf(stderr, "klang-mi355: klg_table_upload: %s\n", klg_last_error()); std::abort(); }
			dirty = false;
		}
		w[WT_INC] = gpu::fbits(increment); w[WT_POS] = gpu::fbits(position); w[WT_OFFSET] = gpu::fbits(offset); w[WT_FREQ] = gpu::fbits(frequency.value); w[WT_TABLE] = (uint32_t)table_id;
	}
	void unpack(const uint32_t* w) override { std::memcpy(&increment, &w[klg::graph::WT_INC], 4); std::memcpy(&position, &w[klg::graph::WT_POS], 4); }
};
// Generators::Wavetables (klang.h:5369-5380): a cycle of Basic::Sine / Basic::Saw rendered into 2,048 samples by the constructor
namespace Generators { namespace Wavetables {
	struct Sine : public Wavetable { Sine() : Wavetable(Basic::Sine()) {} };
	struct Saw : public Wavetable { Saw() : Wavetable(Basic::Saw()) {} };
} }
// Sample (klang.h:3683-3720): plays an attached buffer at one sample per sample, whatever the frequency
class Sample : public Wavetable {
public:
	Sample() : Wavetable(2) {}
	Sample& operator=(const klang::buffer& b) { samples.assign(b.data(), b.data() + b.size); size = b.size; dirty = true; return *this; }   // klang.h:3696-3700 (the samples are copied to HBM when the note is packed)
	Sample& operator=(const std::vector<float>& data) { samples.assign(data.begin(), data.end()); size = (int)data.size(); dirty = true; return *this; }
	void set(param f) override { if (gpu::no_set_while_recording("Sample::set(f)")) return; frequency = f; increment = 1.f; }
	void set(param f, param phase) override { if (gpu::no_set_while_recording("Sample::set(f, phase)")) return; position = phase * 44100.f; set(f); }
	void set(relative phase) override { if (gpu::no_set_while_recording("Sample::set(relative)")) return; offset = phase.value * 44100.f; }
	void set(param f, relative phase) override { set(f); set(phase); }
};

// Array<T, N> (klang.h:1373-1444): a counted fixed-capacity array
// Array<TYPE, CAPACITY> (klang.h:245-330).  Its elements and its count are MEMBERS a recorded process() may read (`for (d = 0; d < times.count; d++)
// out += delay(times[d]) * gains[d];` examples/Reverb.k:90-91): float elements are kept as params (a record word each), the count is a word too, and
// `d < count` is a recorded comparison — one traced path per possible count, merged into nested ifs (PathMerger) — that is false outright at CAPACITY.
template<int CAPACITY> struct ArrayCount {
	param v;
	ArrayCount() : v(0.f) {}
	ArrayCount& operator=(unsigned n) { if (gpu::no_set_while_recording("Array::count = n")) return *this; v.value = (float)n; v.reg = -1; return *this; }
	ArrayCount& operator=(int n) { return *this = (unsigned)n; }
	ArrayCount& operator=(const ArrayCount& o) { return *this = (unsigned)o.v.value; }
	ArrayCount& operator++() { return *this = (unsigned)v.value + 1u; }
	unsigned operator++(int) { const unsigned n = (unsigned)v.value; *this = n + 1u; return n; }
	operator unsigned() const { v.concrete_only("the integer value of an Array's count"); return (unsigned)v.value; }
	template<class T, std::enable_if_t<std::is_integral_v<T>, int> = 0> friend gpu::Pred operator<(T d, const ArrayCount& c) {
		if ((long long)d >= (long long)CAPACITY) return gpu::Pred{ false, -1 };
		return (float)d < c.v;
	}
};
template<typename TYPE, int CAPACITY> struct Array {
	using Item = std::conditional_t<std::is_same_v<TYPE, float>, param, TYPE>;
	Item items[CAPACITY] = {}; ArrayCount<CAPACITY> count;
	void add(const TYPE& v) { const unsigned n = count; if (n < (unsigned)CAPACITY) { items[n] = v; count = n + 1u; } }
	Item& operator[](int i) { return items[i]; } const Item& operator[](int i) const { return items[i]; }
	unsigned size() const { return count; }
};
template<typename TYPE, int SIZE> struct Table {
	TYPE items[SIZE] = {}; unsigned count = 0;
	void add(const TYPE& v) { if (count < (unsigned)SIZE) items[count++] = v; }
	Table(TYPE (*function)(TYPE)) { for (int x = 0; x < SIZE; x++) add(function((TYPE)x)); }
	Table(void (*function)(TYPE x, Result<TYPE>& y)) { count = SIZE; Result<TYPE> y(items, 0); for (int x = 0; x < SIZE; x++) { function((TYPE)x, y); y.sum += items[x]; y++; } }
	Table(std::initializer_list<TYPE> values) { for (TYPE v : values) add(v); }
	TYPE operator[](int index) const { return items[index]; }
	// a recorded index: the clamped linear read runs on the device (OP_TABREAD), the samples go to HBM with the bank
	template<class S, std::enable_if_t<std::is_base_of_v<signal, S> && std::is_same_v<TYPE, float>, int> = 0>
	signal operator[](const S& index) const {
		gpu::Recorder* r = gpu::recording();
		if (!r || index.reg < 0) return signal((*this)[(float)index.value]);
		signal y((*this)[(float)index.value]);
		y.reg = r->emit(klg::graph::OP_TABREAD, index.reg, -1, -1, r->table_slot(this, items, SIZE), true);
		return y;
	}
	TYPE operator[](float index) const {
		if (index < 0) return items[0];
		if (index >= (SIZE - 1)) return items[SIZE - 1];
		const float x = std::floor(index); const int i = int(x);
		return items[i] + (index - x) * (items[i + 1] - items[i]);
	}
};
struct GraphStub { void clear() {} template<class... A> void add(A...) {} template<class... A> void plot(A...) {} template<class... A> GraphStub& operator()(A...) { return *this; } };   // the UI line plotter: nothing to draw on here (`graph(-2, 2)`: its axes)
template<class R, class... A> inline GraphStub& operator>>(R (*)(A...), GraphStub& g) { return g; }   // `hardclip >> graph(-2, 2);` (Distortion/Functions.k:22): plotting a function
inline thread_local GraphStub graph;

// =================================================================================================
// GPU binding of a Note type: found by ADL on the note pointer / reference (see klang/bindings.h)
// =================================================================================================
inline int klang_gpu_patch(const void*) { return -1; }
inline void klang_gpu_pack(const void*, uint32_t*) {}
inline void klang_gpu_unpack(void*, const uint32_t*) {}
#define KLANG_GPU_BIND(NOTE, PATCH, BINDER) \
	inline int klang_gpu_patch(const NOTE*) { return PATCH; } \
	inline void klang_gpu_pack(const NOTE* n, uint32_t* w) { BINDER::pack(*n, w); } \
	inline void klang_gpu_unpack(NOTE* n, const uint32_t* w) { BINDER::unpack(*n, w); }

// the same for an Effect type and a klg_patch id of an effect kernel (KLG_PATCH_PINGPONG, KLG_PATCH_REVERB): gpu::EffectBank<FX> then
// creates the bank with klg_fx_create instead of recording FX::process()
inline int klang_gpu_fx_patch(const void*) { return -1; }
// (KLANG_GPU_BIND_FX is defined with the effect classes below: it also registers the type for Effect::process(buffer))

struct NoteBinding { int patch; void (*pack)(const void*, uint32_t*); void (*unpack)(void*, const uint32_t*); };

// A recorded Note type: the program, and where each node's object sits inside a Note of that type (every instance of the
// type has the same layout, so the offsets found on the prototype serve all of them).
namespace gpu {
struct GraphLayout {
	struct Member { size_t offset; int kind; int word0; bool shared = false; int ctl = -1; bool ctl_smoothed = false; };   // ctl: an EFFECT's smoothed / written control — it lives in the object's Controls (a vector), not inside the object   // offset: of the Packable subobject (primitives) or of the signal (params); shared: a Note's smoothed control — it lives in the Synth, the bank sets the word per block
	std::vector<Member> members;
	std::string program;
	int words = 0;
	std::vector<std::vector<float>> tables;                        // tabread slot k + 1 -> samples (uploaded by the Synth when the bank is created)
	bool host_prepare = false;                                     // an effect whose prepare() stays host code (Controls::changed())
	std::vector<int> delay_inputs;                                 // per member: an effect's Delay takes this many inputs per sample (its write cursor = samples x inputs, modulo SIZE)
	// (own: the Controls of the object being packed, when it is not the recorded prototype — a host mirror of an instance: member offsets into the prototype's Controls mean nothing there)
	void pack(const void* note, uint32_t* w, const Controls* own = nullptr) const {
		for (const Member& m : members) {
			const char* obj = (const char*)note + m.offset;
			if (m.shared) { w[m.word0] = 0u; continue; }
			if (m.ctl >= 0 && own) { const Control& c = own->items[(size_t)m.ctl]; w[m.word0] = fbits(m.ctl_smoothed ? c.smoothed.value : c.value.value); continue; }
			if (m.kind == klg::graph::N_PARAM || m.kind == klg::graph::N_SMOOTH || m.kind == klg::graph::N_CTLVAR) w[m.word0] = fbits(reinterpret_cast<const signal*>(obj)->value);
			else reinterpret_cast<const Packable*>(obj)->pack(w + m.word0);
		}
	}
	// a note's delay line keeps its cursors across notes (the reference never resets Delay::position): read them back even from a voice that is Off
	void unpack_delays(void* note, const uint32_t* w) const {
		for (const Member& m : members) if (m.kind == klg::graph::N_NDELAY) reinterpret_cast<Packable*>((char*)note + m.offset)->unpack(w + m.word0);
	}
	void unpack(void* note, const uint32_t* w, Controls* own = nullptr) const {
		for (const Member& m : members) {
			char* obj = (char*)note + m.offset;
			if (m.shared) continue;
			if (m.ctl >= 0 && own) { Control& c = own->items[(size_t)m.ctl]; std::memcpy(m.ctl_smoothed ? &c.smoothed.value : &c.value.value, &w[m.word0], 4); continue; }
			if (m.kind == klg::graph::N_PARAM || m.kind == klg::graph::N_SMOOTH || m.kind == klg::graph::N_CTLVAR) std::memcpy(&reinterpret_cast<signal*>(obj)->value, &w[m.word0], 4);
			else reinterpret_cast<Packable*>(obj)->unpack(w + m.word0);
		}
	}
};
}

namespace gpu {                                                              // what a note's Delay::clear() needs to find its line: set by SynthCore around events
inline thread_local int current_voice = -1; inline thread_local const void* current_note = nullptr; inline thread_local const GraphLayout* current_layout = nullptr;
}
namespace gpu {
// ---- data-dependent branches: one run of process() per outcome, merged into structured if / else / endif + phi ops ----
// `run` executes process() and its epilogue (write-backs, then the OP_OUT marker naming the output registers).  The first
// trace takes `true` everywhere; at each OP_IF marker the other outcome is traced too, the two traces rejoin where their
// tails become the same ops again (the longest common tail: merging more of the tail is always valid, the operands that
// differ become phis), the parts in between are the two sides.  Sides may contain further branches (handled recursively).
struct PathMerger {
	using Op = klg::graph::Op;
	enum { OP_OUT = klg::graph::OP_CODES };                       // recorder-internal: a = out (left), b = right or -1; becomes `ret`
	Recorder& R; size_t base_ops; int base_reg; std::function<void()> run;
	std::vector<Op> out; int next = 0, pseudo = 1 << 20, runs = 0;
	PathMerger(Recorder& r, std::function<void()> f) : R(r), base_ops(r.prog.ops.size()), base_reg(r.next_reg), run(std::move(f)), next(r.next_reg) {}
	std::vector<Op> trace(const std::vector<char>& D) {
		R.prog.ops.resize(base_ops); R.next_reg = base_reg; R.decisions = D; R.decision_pos = 0; R.run_id++;
		if (++runs > 2048) { R.fail("process() has too many data-dependent branches to record"); return {}; }
		run();
		return std::vector<Op>(R.prog.ops.begin() + (std::ptrdiff_t)base_ops, R.prog.ops.end());
	}
	std::map<int, int> const_final;                               // pool index -> register of the final program
	int map_reg(const std::map<int, int>& m, int r) {
		if (r < base_reg) return r;                                   // -1, or a register of the common prologue
		if (r >= Recorder::CONST_BASE) { const auto it = const_final.find(r - Recorder::CONST_BASE); if (it != const_final.end()) return it->second; return const_final[r - Recorder::CONST_BASE] = next++; }
		const auto it = m.find(r);
		if (it == m.end()) { R.fail("a value computed inside one side of an `if` is used after it in a way that cannot be merged"); return 0; }
		return it->second;
	}
	void copy_op(const Op& o, std::map<int, int>& m) { Op q = o; q.a = map_reg(m, o.a); q.b = map_reg(m, o.b); if (o.dst >= 0) { q.dst = next++; m[o.dst] = q.dst; } out.push_back(q); }
	static bool same(const Op& a, const Op& b) { return a.code == b.code && a.node == b.node && a.imm == b.imm && (a.dst >= 0) == (b.dst >= 0); }
	// a trace: `cur` is rewritten as phis replace operands of its tail, `orig` keeps the registers as the run numbered them
	// (two runs that share a prefix number it identically, so `orig` is what traces are compared on)
	struct Trace { std::vector<Op> cur, orig; size_t size() const { return cur.size(); } };
	// longest tail of T[from..) / F[from..) that is the same ops with corresponding operands; `occ` (optional) receives the
	// operand occurrences (index into the tail, field 0 = a / 1 = b) whose values differ between the traces: the phis
	// (p0: pseudo registers >= p0 are phis made inside the sides of this `if` — out of scope after it, so they need a phi too)
	size_t common_tail(const Trace& T, const Trace& F, size_t from, int r_if, int p0, std::vector<std::pair<size_t, int>>* occ) {
		size_t S = 0;
		const size_t lim = std::min(T.size(), F.size()) - from;
		while (S < lim && same(T.cur[T.size() - 1 - S], F.cur[F.size() - 1 - S])) S++;
		for (;;) {
			const size_t sT = T.size() - S, sF = F.size() - S;
			std::map<int, int> def; std::set<int> fdef;
			if (occ) occ->clear();
			bool ok = true;
			for (size_t k = 0; k < S && ok; k++) {
				const Op& a = T.orig[sT + k]; const Op& b = F.orig[sF + k];
				const int xs[2] = { a.a, a.b }, ys[2] = { b.a, b.b };
				for (int f = 0; f < 2 && ok; f++) {
					const int x = xs[f], y = ys[f];
					if (x < 0 && y < 0) continue;
					const bool dx = def.count(x) != 0, dy = fdef.count(y) != 0;
					if ((x < 0) != (y < 0) || dx != dy || (dx && def[x] != y)) { S = S - k - 1; ok = false; break; }
					if (dx) continue;
					const int xc = f ? T.cur[sT + k].b : T.cur[sT + k].a, yc = f ? F.cur[sF + k].b : F.cur[sF + k].a;
					if (x == y && (x < r_if || x >= Recorder::CONST_BASE) && !(xc >= p0 && xc < Recorder::CONST_BASE) && !(yc >= p0 && yc < Recorder::CONST_BASE)) continue;   // the same value, computed before the branch (or the same literal)
					if (occ) occ->push_back({ k, f });
				}
				if (ok && a.dst >= 0) { def[a.dst] = b.dst; fdef.insert(b.dst); }
			}
			if (ok) return S;
		}
	}
	void emit_range(Trace& T, size_t lo, size_t hi, std::map<int, int>& m) {
		using namespace klg::graph;
		for (size_t i = lo; i < hi && R.error.empty();) {
			const Op o = T.cur[i];
			if (o.code != OP_IF) { copy_op(o, m); i++; continue; }
			std::vector<char> D;
			for (size_t q = 0; q < i; q++) if (T.cur[q].code == OP_IF) D.push_back((char)T.cur[q].imm);
			D.push_back(0);
			Trace F; F.cur = trace(D); F.orig = F.cur;
			if (!R.error.empty()) return;
			bool prefix = F.size() > i && F.cur[i].code == OP_IF && F.cur[i].imm == 0 && o.imm == 1;
			for (size_t q = 0; prefix && q < i; q++) prefix = same(T.orig[q], F.orig[q]) && T.orig[q].dst == F.orig[q].dst;
			if (!prefix) { R.fail("process() does not record the same ops when run again (does it depend on random() or host state?)"); return; }
			int r_if = base_reg;
			for (size_t q = 0; q < i; q++) if (T.orig[q].dst >= r_if) r_if = T.orig[q].dst + 1;
			const int p0 = pseudo;
			const size_t S = common_tail(T, F, i + 1, r_if, p0, nullptr);
			const size_t sT = T.size() - S, sF = F.size() - S;
			if (std::getenv("KLANG_MI355_DEBUG_PATHS")) {
				std::fprintf(stderr, "if at %zu: |T| %zu |F| %zu tail %zu -> sides T[%zu,%zu) F[%zu,%zu) hi %zu r_if %d\n", i, T.size(), F.size(), S, i + 1, sT, i + 1, sF, hi, r_if);
				for (size_t q = i; q < std::min(T.size(), i + 14); q++) std::fprintf(stderr, "   T %s %d %d %d n%d %x | F %s %d %d %d n%d %x\n", op_name(T.orig[q].code), T.orig[q].dst, T.orig[q].a, T.orig[q].b, T.orig[q].node, T.orig[q].imm,
					q < F.size() ? op_name(F.orig[q].code) : "-", q < F.size() ? F.orig[q].dst : 0, q < F.size() ? F.orig[q].a : 0, q < F.size() ? F.orig[q].b : 0, q < F.size() ? F.orig[q].node : 0, q < F.size() ? F.orig[q].imm : 0);
			}
			if (sT > hi) { R.fail("an `if` inside process() does not rejoin the code after it (early return?)"); return; }
			Op c = o; c.a = map_reg(m, o.a); c.imm = 0; out.push_back(c);
			std::map<int, int> mT = m, mF = m;
			emit_range(T, i + 1, sT, mT);
			Op e = o; e.code = OP_ELSE; e.a = -1; e.imm = 0; out.push_back(e);
			emit_range(F, i + 1, sF, mF);
			e.code = OP_ENDIF; out.push_back(e);
			if (!R.error.empty()) return;
			std::vector<std::pair<size_t, int>> occ;
			if (common_tail(T, F, i + 1, r_if, p0, &occ) != S) { R.fail("internal: the branch join moved"); return; }
			// after the join only what was visible before the `if` and the phis are: registers of the sides are out of scope
			// (a later run that took the `if` side names its values by their raw registers: those map to a phi that merged them)
			std::map<std::pair<int, int>, int> phi_of;                     // (then value, else value) -> pseudo register of the tail
			for (const auto& oc : occ) {
				Op& a = T.cur[sT + oc.first]; const Op& b = F.cur[sF + oc.first];
				int& x = oc.second ? a.b : a.a; const int y = oc.second ? b.b : b.a;
				const int x_raw = oc.second ? T.orig[sT + oc.first].b : T.orig[sT + oc.first].a;
				const auto key = std::make_pair(x, y);
				auto it = phi_of.find(key);
				if (it == phi_of.end()) {
					Op ph; ph.code = OP_PHI; ph.node = -1; ph.imm = 0; ph.a = map_reg(mT, x); ph.b = map_reg(mF, y); ph.dst = next++;
					out.push_back(ph);
					it = phi_of.emplace(key, pseudo++).first;
					m[it->second] = ph.dst;
					if (x_raw >= r_if && x_raw < Recorder::CONST_BASE) m[x_raw] = ph.dst;
				}
				x = it->second;
			}
			i = sT;
		}
	}
	// records every path; on return R.prog.ops = prologue + the structured body, R.prog.ret / ret_r are set
	void record() {
		R.pool_consts = true;
		Trace T; T.cur = trace({}); T.orig = T.cur;
		if (!R.error.empty()) return;
		std::map<int, int> m;
		emit_range(T, 0, T.size(), m);
		if (!R.error.empty()) return;
		if (out.empty() || out.back().code != OP_OUT) { R.fail("internal: the recorded body does not end with its output"); return; }
		R.prog.ret = out.back().a; R.prog.ret_r = out.back().b; out.pop_back();
		for (const Op& q : out) if (q.code == OP_OUT) { R.fail("internal: output marker inside a branch"); return; }
		R.prog.ops.resize(base_ops);
		for (const auto& kv : const_final) { Op c; c.code = klg::graph::OP_CONST; c.dst = kv.second; c.a = c.b = c.node = -1; c.imm = R.const_pool[(size_t)kv.first]; R.prog.ops.push_back(c); }
		R.prog.ops.insert(R.prog.ops.end(), out.begin(), out.end());
		R.next_reg = next;
		R.pool_consts = false;
	}
};
}

namespace gpu {
// dead-code elimination, node numbering and member layout shared by the note and the effect recorder
// `if (env.finished()) stop();` (klang.h:4276-4279 in every shipped note) arrives as envoff / if / stop / else / endif: one `stopif` — which a body
// without other branches may test once per block instead of per sample, and which has a two-voices-per-lane form (klg_graph.hpp)
inline void fold_stop_idiom(std::vector<klg::graph::Op>& ops) {
	using namespace klg::graph;
	for (size_t i = 0; i + 4 < ops.size(); i++) {
		if (ops[i].code != OP_ENVOFF || ops[i + 1].code != OP_IF || ops[i + 1].a != ops[i].dst || ops[i + 2].code != OP_STOP || ops[i + 3].code != OP_ELSE || ops[i + 4].code != OP_ENDIF) continue;
		if (i + 5 < ops.size() && ops[i + 5].code == OP_PHI) continue;
		bool read = false;
		for (size_t q = 0; q < ops.size() && !read; q++) if (q != i + 1 && (ops[q].a == ops[i].dst || ops[q].b == ops[i].dst)) read = true;
		if (read) continue;
		Op s = ops[i]; s.code = OP_STOPIF; s.dst = -1; s.a = s.b = -1; s.imm = 0;
		ops[i] = s; ops.erase(ops.begin() + (std::ptrdiff_t)i + 1, ops.begin() + (std::ptrdiff_t)i + 5);
	}
}
inline void finish_program(Recorder& R, const char* lo, GraphLayout& L) {
	using namespace klg::graph;
	fold_stop_idiom(R.prog.ops);
	// ---- dead code: pure ops nobody reads, params nobody reads (and their write-backs), primitives nobody uses ----
	std::vector<Op>& ops = R.prog.ops;
	std::vector<char> keep(ops.size(), 1), used;
	auto pure = [](int c) { return c == OP_CONST || c == OP_CTL || c == OP_PARAM || c == OP_FREQ || c == OP_IN || c == OP_ADD || c == OP_SUB || c == OP_MUL || c == OP_DIV || c == OP_NEG || c == OP_CMP || c == OP_ENVOFF || c == OP_PHI || c == OP_TABREAD || c == OP_FUNC || (c >= OP_F2D && c <= OP_D2F); };
	for (bool changed = true; changed;) {
		changed = false;
		used.assign(MAX_OPS + 1, 0); used[(size_t)R.prog.ret] = 1; if (R.prog.ret_r >= 0) used[(size_t)R.prog.ret_r] = 1;
		for (size_t i = ops.size(); i-- > 0;) {                         // registers are defined before use: one backward sweep
			if (!keep[i]) continue;
			const Op& o = ops[i];
			if (pure(o.code) && !used[(size_t)o.dst]) { keep[i] = 0; changed = true; continue; }
			if (o.a >= 0) used[(size_t)o.a] = 1;
			if (o.b >= 0) used[(size_t)o.b] = 1;
		}
		std::vector<char> param_read(R.objs.size(), 0);
		for (size_t i = 0; i < ops.size(); i++) if (keep[i] && ops[i].code == OP_PARAM) param_read[(size_t)ops[i].node] = 1;
		for (size_t i = 0; i < ops.size(); i++) if (keep[i] && ops[i].code == OP_SETPARAM && !param_read[(size_t)ops[i].node]) { keep[i] = 0; changed = true; }
	}
	std::vector<int> node_id(R.objs.size(), -1);
	std::vector<char> node_used(R.objs.size(), 0);
	for (size_t i = 0; i < ops.size(); i++) if (keep[i] && ops[i].node >= 0) node_used[(size_t)ops[i].node] = 1;
	for (size_t i = 0; i < R.objs.size(); i++) if (node_used[i]) { node_id[i] = (int)R.prog.nodes.size(); R.prog.nodes.push_back(R.objs[i].kind); R.prog.node_arg.push_back(klg::graph::node_arg_of(R.objs[i].kind, R.objs[i].live_arg ? *R.objs[i].live_arg : R.objs[i].arg)); }
	std::vector<Op> out_ops;
	int kept_prepare = 0;
	for (size_t i = 0; i < ops.size(); i++) if (keep[i]) { Op o = ops[i]; if (o.node >= 0) o.node = node_id[(size_t)o.node]; out_ops.push_back(o); if ((int)i < R.prog.prepare_ops) kept_prepare++; }
	ops = out_ops;
	R.prog.prepare_ops = kept_prepare;
	const std::string verr = R.prog.validate();
	if (!verr.empty()) { std::fprintf(stderr, "klang-mi355: the recorded program is invalid: %s\n%s", verr.c_str(), R.prog.text().c_str()); std::abort(); }
	for (size_t i = 0; i < R.objs.size(); i++) if (node_used[i]) {
		const void* at = (R.objs[i].kind == N_PARAM || R.objs[i].kind == N_SMOOTH || R.objs[i].kind == N_CTLVAR || !R.objs[i].packable) ? R.objs[i].addr : (const void*)R.objs[i].packable;
		L.members.push_back({ (size_t)((const char*)at - lo), R.objs[i].kind, R.prog.node_word0(node_id[i]), R.objs[i].kind == N_SMOOTH && !R.effect });
	}
	L.program = R.prog.text();
	L.words = R.prog.words();
	for (const auto& t : R.static_tables) L.tables.push_back(t.data);
}
}

// ---- Controller / Plugin / Effect / NoteBase (klang.h:4182-4292) ----
struct Controller {
protected:
	virtual event control(int, float) {}
	virtual event preset(int) {}
	virtual event midi(int, int, int) {}
public:
	virtual ~Controller() {}
	virtual void onControl(int index, float value) { control(index, value); }
	virtual void onPreset(int index) { preset(index); }
	virtual void onMIDI(int status, int byte1, int byte2) { midi(status, byte1, byte2); }                  // klang.h:4191
};
namespace gpu {
// Every Plugin and every Note owns the log of its own construction (gpu::ConstructionLog): the base-class constructor below runs
// BEFORE the members of the user's class are constructed, so from here on each primitive / signal member announces itself into
// this log — which is how Effect::process(buffer) / Note::process(buffer) later find the members of an object that the HOST
// constructed (`PingPong pingpong;` in a plugin processor), with no template parameter naming its type.
struct Owner {
	ConstructionLog log;
	explicit Owner(bool effect) { log.effect = effect; if (!(rec && rec->constructing) && !log_suppress) log_target = &log; }
	Owner(const Owner& o) { log.effect = o.log.effect; }                     // a copy has a log of its own (empty: its members were copied, not constructed)
	Owner& operator=(const Owner&) { return *this; }
	~Owner() { if (log_target == &log) log_target = nullptr; }
};
// the members among what a sink saw: inside [lo, hi) (a prototype built under a Recorder) or still alive (an owner's log); signals
// inside a primitive belong to the primitive
inline std::vector<Obj> member_objs(const Sink& sink, const char* lo, const char* hi) {
	std::vector<Obj> in, kept;
	for (const Obj& o : sink.objs) {
		const char* a = (const char*)o.addr;
		if (hi ? (a < lo || a >= hi) : !sink.alive(o)) continue;
		in.push_back(o);
	}
	for (const Obj& o : in) {
		const char* a = (const char*)o.addr;
		bool inside = false;
		if (o.kind == klg::graph::N_PARAM) for (const Obj& q : in) if (q.kind != klg::graph::N_PARAM && a >= (const char*)q.addr && a < (const char*)q.addr + q.size) inside = true;
		if (!inside) kept.push_back(o);
	}
	return kept;
}
// effect types tied to a hand-written kernel, by dynamic type (what Effect::process(buffer) can look up from `*this`)
inline std::map<std::type_index, int>& fx_bindings() { static std::map<std::type_index, int> m; return m; }
inline bool register_fx(const std::type_info& t, int patch) { fx_bindings()[std::type_index(t)] = patch; return true; }
inline int fx_binding(const std::type_info& t) { const auto it = fx_bindings().find(std::type_index(t)); return it == fx_bindings().end() ? -1 : it->second; }
}
#define KLANG_GPU_BIND_FX_CAT2(a, b) a##b
#define KLANG_GPU_BIND_FX_CAT(a, b) KLANG_GPU_BIND_FX_CAT2(a, b)
#define KLANG_GPU_BIND_FX(FX, PATCH) inline int klang_gpu_fx_patch(const FX*) { return PATCH; } \
	static const bool KLANG_GPU_BIND_FX_CAT(klang_gpu_fx_registered_, __LINE__) = klang::gpu::register_fx(typeid(FX), PATCH);

struct Plugin : Controller, gpu::Owner { Plugin() : gpu::Owner(true) {} Controls controls; Presets presets; };

namespace gpu {
// a member's write-back whose value is the member's own read at the top of the sample (process() did not write it on any path): nothing to store
inline void drop_idle_writebacks(Recorder& R, const std::vector<int>& first_reg) {
	std::vector<klg::graph::Op>& ops = R.prog.ops;
	size_t w = (size_t)R.prog.prepare_ops;
	for (size_t q = w; q < ops.size(); q++) {
		const klg::graph::Op& o = ops[q];
		if (o.code == klg::graph::OP_SETPARAM && o.node >= 0 && (size_t)o.node < first_reg.size() && o.a == first_reg[(size_t)o.node]) continue;
		ops[w++] = o;
	}
	ops.resize(w);
}
// ---- recording a Note's process() (+ prepare()) into a graph program ----
// `objs`: the note's members (member_objs).  prepare() — host code that runs once per block in the reference (klang.h:4292-4296) —
// becomes the program's per-block prologue, like an effect's.
inline thread_local bool quiet_recording = false;                              // (a body re-recorded only to be compared: SynthCore::check_body)
template<class NOTEBASE> inline void record_note(NOTEBASE* nb, std::vector<Obj> objs, Controls& ctl, const char* lo, GraphLayout& L, const char* type_name) {
	using namespace klg::graph;
	Recorder R; rec = &R;
	R.objs = std::move(objs);
	R.prog.nctl = (int)ctl.items.size() < (int)klg::KLG_MAX_CTL ? (int)ctl.items.size() : (int)klg::KLG_MAX_CTL;
	for (int c = 0; c < R.prog.nctl; c++) R.prog.dials[c] = { ctl.items[(size_t)c].min, ctl.items[(size_t)c].max, ctl.items[(size_t)c].initial };
	R.recording = true;
	std::vector<int> first_reg(R.objs.size(), -1);
	// `osc.frequency` read inside process(): the node's current frequency (what on() or a recorded set(f) left there)
	std::vector<Oscillator*> oscs; std::vector<int> osc_node;
	for (size_t i = 0; i < R.objs.size(); i++) if (is_oscillator(R.objs[i].kind) || R.objs[i].kind == N_OPERATOR)
		if (Oscillator* o = const_cast<Oscillator*>(dynamic_cast<const Oscillator*>(R.objs[i].packable))) { oscs.push_back(o); osc_node.push_back((int)i); }
	auto fresh_inputs = [&]() {                                              // member / control / frequency reads of the phase being recorded
		for (size_t i = 0; i < R.objs.size(); i++) if (R.objs[i].kind == N_PARAM) { signal* sg = (signal*)R.objs[i].addr; first_reg[i] = sg->reg = R.emit(OP_PARAM, -1, -1, (int)i, 0, true); }
		for (int c = 0; c < R.prog.nctl; c++) ctl.items[(size_t)c].value.reg = R.emit(OP_CTL, -1, -1, -1, (uint32_t)c, true);
		for (size_t q = 0; q < oscs.size(); q++) oscs[q]->frequency.reg = R.emit(OP_FREQ, -1, -1, osc_node[q], 0, true);
	};
	fresh_inputs();
	nb->prepare();
	for (size_t i = 0; i < R.objs.size(); i++) if (R.objs[i].kind == N_PARAM) { signal* sg = (signal*)R.objs[i].addr; if (sg->reg != first_reg[i]) R.emit(OP_SETPARAM, R.reg_of(*sg), -1, (int)i, 0, false); }
	R.prog.prepare_ops = (int)R.prog.ops.size();
	fresh_inputs();
	std::vector<float> value0(R.objs.size(), 0.f); std::vector<int> freq_reg;
	for (size_t i = 0; i < R.objs.size(); i++) if (R.objs[i].kind == N_PARAM) value0[i] = ((signal*)R.objs[i].addr)->value;
	for (Oscillator* o : oscs) freq_reg.push_back(o->frequency.reg);
	PathMerger paths(R, [&]() {                                          // one run of process(): every data-dependent `if` outcome gets its own (gpu::PathMerger)
		for (size_t i = 0; i < R.objs.size(); i++) if (R.objs[i].kind == N_PARAM) { signal* sg = (signal*)R.objs[i].addr; sg->reg = first_reg[i]; sg->value = value0[i]; }
		for (size_t q = 0; q < oscs.size(); q++) oscs[q]->frequency.reg = freq_reg[q];
		R.may_branch = true;
		nb->run_process();
		R.may_branch = false;
		// `out` at the end of process(): one register, or two for a Stereo::Note whose out is {l, r} (klang.h:4721-4733) -> `ret2`
		const int ret = R.reg_of(nb->out_channel(0)), ret_r = nb->out_channels() == 2 ? R.reg_of(nb->out_channel(1)) : -1;
		for (size_t i = 0; i < R.objs.size(); i++) if (R.objs[i].kind == N_PARAM) {
			signal* sg = (signal*)R.objs[i].addr;
			R.emit(OP_SETPARAM, R.reg_of(*sg), -1, (int)i, 0, false);     // written by process(): the next sample reads it.  (EVERY member, also one this run left alone: the runs of
		}                                                                     // different branch outcomes then end in the same ops and rejoin right after the `if` — drop_idle_writebacks removes the rest)
		R.emit(PathMerger::OP_OUT, ret, ret_r, -1, 0, false);
	});
	paths.record();
	drop_idle_writebacks(R, first_reg);
	for (size_t i = 0; i < R.objs.size(); i++) if (R.objs[i].kind == N_PARAM) { signal* sg = (signal*)R.objs[i].addr; sg->reg = -1; sg->value = value0[i]; }
	for (int c = 0; c < R.prog.nctl; c++) ctl.items[(size_t)c].value.reg = -1;
	for (Oscillator* o : oscs) o->frequency.reg = -1;
	R.recording = false;
	rec = nullptr;
	if (!R.error.empty()) { std::fprintf(stderr, "klang-mi355: cannot record %s::process() as a graph patch: %s\n", type_name, R.error.c_str()); std::abort(); }
	finish_program(R, lo, L);
	if (std::getenv("KLANG_MI355_DUMP_GRAPH") && !quiet_recording) std::fprintf(stderr, "klang-mi355: recorded %s::process():\n%s", type_name, L.program.c_str());
}

// ---- recording an Effect's prepare() + process() into a `kind effect` program ----
// `identity` (optional) is set when the body is `out = in` and nothing else (the default Effect::process(), klang.h:4206: what a Synth
// that has no post-processing of its own inherits) — no program is built then.
inline void record_effect(std::vector<Obj> objs, Controls& ctl, int channels, signal* const ins[2], signal* const outs[2], const std::function<void()>& prepare,
                          const std::function<void()>& process, const char* lo, GraphLayout& layout, const char* type_name, bool* identity = nullptr) {
	using namespace klg::graph;
	Recorder R; R.effect = true; rec = &R;
	R.objs = std::move(objs);
	R.prog.channels = channels;
	R.prog.nctl = (int)ctl.items.size() < (int)klg::KLG_MAX_CTL ? (int)ctl.items.size() : (int)klg::KLG_MAX_CTL;
	for (int c = 0; c < R.prog.nctl; c++) R.prog.dials[c] = { ctl.items[(size_t)c].min, ctl.items[(size_t)c].max, ctl.items[(size_t)c].initial };
	// prepare() is recorded too: it becomes the program's per-block prologue (`prepare <n>`), so `filter.set(controls[2])`
	// follows each instance's own control.  Member params it assigns are written to the record and read back by process().
	R.recording = true;
	std::vector<int> first_reg(R.objs.size(), -1);
	std::vector<Oscillator*> oscs; std::vector<int> osc_node;
	for (size_t i = 0; i < R.objs.size(); i++) if (is_oscillator(R.objs[i].kind)) if (Oscillator* o = const_cast<Oscillator*>(dynamic_cast<const Oscillator*>(R.objs[i].packable))) { oscs.push_back(o); osc_node.push_back((int)i); }
	auto is_io = [&](const signal* sg) { return sg == ins[0] || sg == ins[1] || sg == outs[0] || sg == outs[1]; };
	auto fresh_inputs = [&]() {                                              // control / member / frequency reads of the phase being recorded
		for (size_t i = 0; i < R.objs.size(); i++) if (R.objs[i].kind == N_PARAM) { signal* sg = (signal*)R.objs[i].addr; first_reg[i] = sg->reg = R.emit(OP_PARAM, -1, -1, (int)i, 0, true); }
		for (int c = 0; c < R.prog.nctl; c++) ctl.items[(size_t)c].value.reg = R.emit(OP_CTL, -1, -1, -1, (uint32_t)c, true);
		for (size_t q = 0; q < oscs.size(); q++) oscs[q]->frequency.reg = R.emit(OP_FREQ, -1, -1, osc_node[q], 0, true);
	};
	const size_t nobjs0 = R.objs.size();
	fresh_inputs();
	prepare();
	if (!R.error.empty() && !R.host_prepare) {
		// prepare() does something a recorded prologue cannot express (Vocoder.k:51-78: Pitch -> Frequency and power() on values read from dials, constants built
		// from them, loops of std::pow): it stays HOST code, exactly as an effect whose prepare() asks Controls::changed() — EffectBank / FxRunner run it on a host
		// mirror of every instance whose dials moved and upload what it changed.  Nothing of the attempt is kept.
		if (!std::getenv("KLANG_MI355_QUIET")) std::fprintf(stderr, "klang-mi355: note: %s::prepare() stays host code (it runs per instance on the host whenever a dial moved): %s\n", type_name, R.error.c_str());
		R.host_prepare = true; R.error.clear(); R.prog.ops.clear(); R.next_reg = 0; R.objs.resize(nobjs0);
		for (size_t i = 0; i < R.objs.size(); i++) if (R.objs[i].kind == N_PARAM) { ((signal*)R.objs[i].addr)->reg = -1; first_reg[i] = -1; }
	}
	for (size_t i = 0; i < R.objs.size(); i++) if (R.objs[i].kind == N_PARAM) { signal* sg = (signal*)R.objs[i].addr; if (sg->reg != first_reg[i]) R.emit(OP_SETPARAM, R.reg_of(*sg), -1, (int)i, 0, false); }
	R.prog.prepare_ops = (int)R.prog.ops.size();
	fresh_inputs();
	for (int c = 0; c < channels; c++) ins[c]->reg = R.emit(OP_IN, -1, -1, -1, (uint32_t)c, true);      // `in` is this sample of the block
	std::vector<float> value0(R.objs.size(), 0.f); std::vector<int> freq_reg, in_reg;
	for (size_t i = 0; i < R.objs.size(); i++) if (R.objs[i].kind == N_PARAM) value0[i] = ((signal*)R.objs[i].addr)->value;
	for (Oscillator* o : oscs) freq_reg.push_back(o->frequency.reg);
	for (int c = 0; c < channels; c++) in_reg.push_back(ins[c]->reg);
	std::vector<int> ctl_reg; std::vector<float> ctl_value;
	for (int c = 0; c < R.prog.nctl; c++) { ctl_reg.push_back(ctl.items[(size_t)c].value.reg); ctl_value.push_back(ctl.items[(size_t)c].value.value); }
	PathMerger paths(R, [&]() {                                          // one run of process() per outcome of its data-dependent `if`s
		for (size_t i = 0; i < R.objs.size(); i++) if (R.objs[i].kind == N_PARAM) { signal* sg = (signal*)R.objs[i].addr; sg->reg = first_reg[i]; sg->value = value0[i]; }
		for (size_t q = 0; q < oscs.size(); q++) oscs[q]->frequency.reg = freq_reg[q];
		for (int c = 0; c < channels; c++) ins[c]->reg = in_reg[(size_t)c];
		for (int c = 0; c < R.prog.nctl; c++) { ctl.items[(size_t)c].value.reg = ctl_reg[(size_t)c]; ctl.items[(size_t)c].value.value = ctl_value[(size_t)c]; }   // (process() may write its controls)
		R.may_branch = true;
		process();
		R.may_branch = false;
		const int ret = R.reg_of(*outs[0]), ret_r = channels == 2 ? R.reg_of(*outs[1]) : -1;
		for (size_t i = 0; i < R.objs.size(); i++) if (R.objs[i].kind == N_PARAM) {
			signal* sg = (signal*)R.objs[i].addr;
			if (!(sg == ins[0] || sg == ins[1])) R.emit(OP_SETPARAM, R.reg_of(*sg), -1, (int)i, 0, false);     // (every member: see record_note)
		}
		R.emit(PathMerger::OP_OUT, ret, ret_r, -1, 0, false);
	});
	paths.record();
	drop_idle_writebacks(R, first_reg);
	(void)is_io;
	for (size_t i = 0; i < R.objs.size(); i++) if (R.objs[i].kind == N_PARAM) { signal* sg = (signal*)R.objs[i].addr; sg->reg = -1; sg->value = value0[i]; }
	for (int c = 0; c < channels; c++) { ins[c]->reg = -1; outs[c]->reg = -1; }
	for (int c = 0; c < R.prog.nctl; c++) { ctl.items[(size_t)c].value.reg = -1; ctl.items[(size_t)c].value.value = ctl_value[(size_t)c]; }
	for (Oscillator* o : oscs) o->frequency.reg = -1;
	R.recording = false; rec = nullptr;
	if (!R.error.empty()) { std::fprintf(stderr, "klang-mi355: cannot record %s::process() as a graph effect: %s\n", type_name, R.error.c_str()); std::abort(); }
	if (identity) {
		*identity = R.prog.ret == in_reg[0] && (channels < 2 || R.prog.ret_r == in_reg[1]);
		for (const Op& o : R.prog.ops) if (o.code == OP_SETPARAM || o.code == OP_DELAYIN || o.code == OP_OSCSET || o.code == OP_LPFSET) *identity = false;
		if (*identity) return;
	}
	finish_program(R, lo, layout);
	for (GraphLayout::Member& m : layout.members) if (m.kind == N_SMOOTH || m.kind == N_CTLVAR) for (size_t c = 0; c < ctl.items.size(); c++) {
		if (lo + m.offset == (const char*)&ctl.items[c].value) { m.ctl = (int)c; m.ctl_smoothed = false; }
		if (lo + m.offset == (const char*)&ctl.items[c].smoothed) { m.ctl = (int)c; m.ctl_smoothed = true; }
	}
	layout.host_prepare = R.host_prepare;
	layout.delay_inputs.assign(layout.members.size(), 0);             // (member j is node j of the finished program)
	for (size_t i = (size_t)R.prog.prepare_ops; i < R.prog.ops.size(); i++) if (R.prog.ops[i].code == OP_DELAYIN && R.prog.ops[i].node >= 0 && (size_t)R.prog.ops[i].node < layout.delay_inputs.size()) layout.delay_inputs[(size_t)R.prog.ops[i].node]++;
	if (std::getenv("KLANG_MI355_DUMP_GRAPH")) std::fprintf(stderr, "klang-mi355: recorded %s::process():\n%s", type_name, layout.program.c_str());
}
}

// =================================================================================================
// Effects (klang.h:4190-4216 Effect, 4703-4717 Stereo::Effect): the DSL side, rendered as recorded graph effects
// =================================================================================================
// Delay<SIZE> (klang.h:3381-3512).  In an Effect: a ring per instance in HBM, cursor = the sample counter (`x >> delay`, `delay << x`,
// `delay(time)`, `(x >> delay)(time)`); on the host the object only takes part in the recording.  In a Note (physical models): a
// `notedelay` node — the line lives in HBM per voice, its cursors (write position, the read head of set() / process()) in the record.
// Delay<0> (klang.h:3515-3624): the same object with a run-time SIZE, given by resize(samples).  On the GPU a line's SIZE fixes its ring in HBM when the bank is
// created: resize() belongs where the reference's examples have it — the constructor, or the first prepare() — and the size is read when the recording is
// finished; a resize() to ANOTHER size afterwards (the reference reallocates a cleared buffer) is refused loudly.
namespace gpu {
template<int N> struct DelaySize { static constexpr int SIZE = N; const int* live_size() const { return nullptr; } };
template<> struct DelaySize<0> {
	int SIZE = 0; bool sized = false;
	const int* live_size() const { return &SIZE; }
	void resize(int samples) {
		if (samples == SIZE) return;
		if (sized) { std::fprintf(stderr, "klang-mi355: Delay<0>::resize(%d) of a line that was sized to %d samples: on the GPU a line keeps the ring it was given\n", samples, SIZE); std::abort(); }
		SIZE = samples; sized = true;
	}
};
}
template<int SIZE_> struct Delay : Modifier, gpu::Packable, gpu::DelaySize<SIZE_> {
	using gpu::DelaySize<SIZE_>::SIZE;
	bool in_note = false;
	float time = 1.f; int position = 0; struct { int position = 0; float fraction = 0.f; } last;      // host mirror (notes)
	Delay() {
		if (gpu::Sink* r = gpu::constructing()) { in_note = !r->effect; r->note(this, sizeof(Delay), in_note ? klg::graph::N_NDELAY : klg::graph::N_DELAY, this, SIZE, static_cast<const gpu::Packable*>(this), this->live_size()); }
		else in_note = true;                                                        // every further Note of a recorded type
	}
	// Delay::lagrange(float) klang.h:3429-3458: third-order Lagrange interpolation over the four samples around the read position
	template<typename TIME> signal lagrange(const TIME& delay) {
		if (gpu::Recorder* r = gpu::recording()) { signal t; if constexpr (std::is_arithmetic_v<TIME>) t = signal((float)delay); else t = signal(const_cast<TIME&>(delay)); signal s; s.reg = r->emit(klg::graph::OP_DELAYTAP, r->reg_of(t), -1, r->node(this, "Delay"), 3, true); return s; }
		device_only("Delay::lagrange()"); return signal();
	}
	using Generic::Input<signal>::input;
	using Modifier::set;
	void input() override {
		if (gpu::Recorder* r = gpu::recording()) { r->emit(klg::graph::OP_DELAYIN, r->reg_of(in), -1, r->node(this, "Delay"), 0, false); return; }
		device_only("Delay::input()");
	}
	template<typename TIME> signal operator()(const TIME& delay) {                 // klang.h:3491-3509: tap(int) for integers, tap(float) otherwise
		if constexpr (std::is_integral_v<TIME>) {                                   // tap(int) klang.h:3405-3410: the sample `delay` positions back, no interpolation
			if (gpu::Recorder* r = gpu::recording()) { signal s; s.reg = r->emit(klg::graph::OP_DELAYTAP, r->reg_of(signal((float)delay)), -1, r->node(this, "Delay"), 1, true); return s; }
			device_only("Delay::operator()");
		}
		if (gpu::Recorder* r = gpu::recording()) { signal t; if constexpr (std::is_arithmetic_v<TIME>) t = signal((float)delay); else if constexpr (std::is_same_v<TIME, dsignal>) t = signal(delay); else t = signal(const_cast<TIME&>(delay)); signal s; s.reg = r->emit(klg::graph::OP_DELAYTAP, r->reg_of(t), -1, r->node(this, "Delay"), 0, true); return s; }
		device_only("Delay::operator()");
	}
	// one channel of Stereo::Delay::tap(float) klang.h:4668-4681: its own interpolation form (a * (1 - f) + b * f), not Delay::tap(float)'s
	template<typename TIME> signal tap_stereo_form(const TIME& delay) {
		if (gpu::Recorder* r = gpu::recording()) { signal t; if constexpr (std::is_arithmetic_v<TIME>) t = signal((float)delay); else t = signal(const_cast<TIME&>(delay)); signal s; s.reg = r->emit(klg::graph::OP_DELAYTAP, r->reg_of(t), -1, r->node(this, "Delay"), 2, true); return s; }
		device_only("Stereo::Delay::operator()"); return signal();
	}
	void set(param samples) override {                                             // klang.h:3480-3489: place the read head `samples` behind the write cursor
		if (gpu::Recorder* r = gpu::recording()) {                                  // per sample: the read head follows a control / an LFO (PingPong.k:62-63)
			r->emit(klg::graph::OP_DELAYSET, r->reg_of(samples), -1, r->node(this, "Delay"), 0, false); return;
		}
		time = samples.value < SIZE ? samples.value : (float)SIZE;
		float read = static_cast<float>(position - 1) - time;
		if (read < 0.f) read += SIZE;
		last.position = static_cast<int>(read);
		last.fraction = read - last.position;
	}
	void clear() {                                                                 // klang.h:3392-3394, on the voice's line in HBM
		if (gpu::no_set_while_recording("Delay::clear()")) return;
		if (!gpu::upload_target || gpu::current_voice < 0 || !gpu::current_layout) return;          // not attached to a voice yet: the line is still all zeros
		int ordinal = 0; const size_t me = (size_t)((const char*)static_cast<const gpu::Packable*>(this) - (const char*)gpu::current_note);
		for (const auto& m : gpu::current_layout->members) if (m.kind == klg::graph::N_NDELAY && m.offset < me) ordinal++;
		if (klg_voice_delay_clear(gpu::upload_target, gpu::current_voice, ordinal)) { std::fprintf(stderr, "klang-mi355: klg_voice_delay_clear: %s\n", klg_last_error()); std::abort(); }
	}
	void process() override {
		if (gpu::Recorder* r = gpu::recording()) {
			out.reg = r->emit(klg::graph::OP_DELAYOUT, -1, -1, r->node(this, "Delay"), 0, true); return;
		}
		device_only("Delay::process()");
	}
	void pack(uint32_t* w) const override {
		using namespace klg::graph;
		if (!in_note) { w[ED_LASTPOS] = (uint32_t)last.position; w[ED_LASTFRAC] = gpu::fbits(last.fraction); return; }      // an effect's Delay: only its read head is state of the record
		w[ND_POS] = (uint32_t)position; w[ND_LASTPOS] = (uint32_t)last.position; w[ND_LASTFRAC] = gpu::fbits(last.fraction); w[ND_TIME] = gpu::fbits(time);
	}
	void unpack(const uint32_t* w) override { using namespace klg::graph; if (!in_note) { last.position = (int)w[ED_LASTPOS]; return; } position = (int)w[ND_POS]; last.position = (int)w[ND_LASTPOS]; }
	void host_cursor(unsigned long long inputs) override { if (!in_note) position = (int)(inputs % (unsigned long long)SIZE); }
	unsigned int max() const { return SIZE; }
};

namespace gpu {
// One instance of an effect on the GPU, built the first time its owner hands it a block: the engine behind Effect::process(buffer),
// Stereo::Effect::process(Stereo::buffer) and the post-processing pass of a Synth (klang.h:4208-4216, 4708-4716, 4451, 4851).
// The effect's type is tied to a hand-written kernel (KLANG_GPU_BIND_FX: klg_fx_create) or its prepare() / process() are recorded
// from the members its construction log names (klg_fx_create_graph); blocks then go through klg_fx_process.
struct FxRunner {
	klg_fx* h = nullptr; GraphLayout layout; bool built = false, identity = false; int channels = 1, cap = 1024;
	std::vector<float> io, sent;
	// an effect whose prepare() stays host code (Controls::changed(): see gpu::EffectBank) — here the host's own object is the mirror of the one instance
	void* host_obj = nullptr; std::function<void()> host_prepare_fn; unsigned long long samples = 0; int device_controls = (int)klg::KLG_MAX_CTL; std::vector<uint32_t> before, after;
	FxRunner() {}
	FxRunner(const FxRunner&) {}
	FxRunner& operator=(const FxRunner&) { return *this; }
	~FxRunner() { if (h) klg_fx_destroy(h); }
	[[noreturn]] static void die(const char* what) { std::fprintf(stderr, "klang-mi355: %s: %s\n", what, klg_last_error()); std::abort(); }
	template<class FX> void build(FX* fx, Owner& owner, Controls& ctl, int channels_, signal* const ins[2], signal* const outs[2], bool may_be_identity) {
		if (built) return;
		built = true; channels = channels_;
		close_log();
		const int bound = fx_binding(typeid(*fx));
		if (bound >= 0 && !std::getenv("KLANG_MI355_FORCE_GRAPH")) {
			if (channels != 2) { std::fprintf(stderr, "klang-mi355: the effect kernels of the library are stereo\n"); std::abort(); }
			h = klg_fx_create(bound, 1, fs.f, cap);
			if (!h) die("klg_fx_create");
			return;
		}
		const char* lo = (const char*)dynamic_cast<const void*>(fx);
		record_effect(member_objs(owner.log, nullptr, nullptr), ctl, channels, ins, outs, [fx]() { fx->prepare(); }, [fx]() { fx->run_process(); }, lo, layout, typeid(*fx).name(), may_be_identity ? &identity : nullptr);
		if (identity) return;
		std::vector<uint32_t> words((size_t)layout.words, 0u);
		layout.pack(lo, words.data());
		h = klg_fx_create_graph(layout.program.c_str(), 1, fs.f, cap, words.data());
		if (!h) die("klg_fx_create_graph");
		device_controls = (int)ctl.items.size() < (int)klg::KLG_MAX_CTL ? (int)ctl.items.size() : (int)klg::KLG_MAX_CTL;
		if (layout.host_prepare) { host_obj = (void*)lo; host_prepare_fn = [fx]() { fx->prepare(); }; before.resize((size_t)layout.words); after.resize((size_t)layout.words); }
	}
	void host_prepare() {                                                            // (gpu::EffectBank::host_prepare, for the one instance the host object itself mirrors)
		if (klg_fx_download_record(h, 0, before.data(), before.size() * 4)) die("klg_fx_download_record");
		layout.unpack(host_obj, before.data());
		for (size_t j = 0; j < layout.members.size(); j++) if (layout.members[j].kind == klg::graph::N_DELAY)
			reinterpret_cast<Packable*>((char*)host_obj + layout.members[j].offset)->host_cursor(samples * (unsigned long long)layout.delay_inputs[j]);
		layout.pack(host_obj, before.data());
		host_prepare_fn();
		layout.pack(host_obj, after.data());
		for (int w = 0; w < layout.words;) {
			if (after[(size_t)w] == before[(size_t)w]) { w++; continue; }
			int e = w + 1; while (e < layout.words && after[(size_t)e] != before[(size_t)e]) e++;
			if (klg_fx_upload_words(h, 0, w, e - w, after.data() + w)) die("klg_fx_upload_words");
			w = e;
		}
	}
	// the host's control values -> the instance: whatever was set() since the last block (even to the same value: a set() overwrites
	// what the effect itself may have written to the control, as in the reference) or differs from what was sent
	void sync_controls(Controls& ctl) {
		const int n = (int)ctl.items.size();
		if ((int)sent.size() != n) sent.assign((size_t)n, std::nanf(""));
		for (int c = 0; c < n; c++) {
			Control& k = ctl.items[(size_t)c];
			if (!k.touched && k.value.value == sent[(size_t)c]) continue;
			if (host_obj && c >= device_controls) { sent[(size_t)c] = k.value.value; k.touched = false; continue; }   // (a dial only the host-run prepare() reads)
			if (klg_fx_set_control(h, 0, c, k.value.value)) die("klg_fx_set_control");
			sent[(size_t)c] = k.value.value; k.touched = false;
		}
	}
	void run(Controls& ctl, float* const* ch, int n) {                               // ch[channels][n], processed in place
		if (identity || n <= 0) return;
		sync_controls(ctl);
		for (int at = 0; at < n; at += cap) {                                        // (prepare() runs once per klg_fx_process call: once per <= 1024 samples)
			const int m = n - at < cap ? n - at : cap;
			io.resize((size_t)channels * (size_t)m);
			for (int c = 0; c < channels; c++) std::memcpy(&io[(size_t)c * (size_t)m], ch[c] + at, (size_t)m * sizeof(float));
			if (host_obj) host_prepare();                                             // (Controls::changed() inside decides whether anything happens)
			if (klg_fx_process(h, io.data(), m)) die("klg_fx_process");
			samples += (unsigned long long)m;
			for (int c = 0; c < channels; c++) std::memcpy(ch[c] + at, &io[(size_t)c * (size_t)m], (size_t)m * sizeof(float));
		}
		// what the effect wrote to its own controls (PingPong.k:48,60) comes back to the host's Control objects, as in the reference
		for (int c = 0; c < (int)ctl.items.size() && c < (int)klg::KLG_MAX_CTL; c++) {
			float v = 0.f;
			if (klg_fx_get_control(h, 0, c, &v)) die("klg_fx_get_control");
			if (v != ctl.items[(size_t)c].value.value) { ctl.items[(size_t)c].value.value = v; sent[(size_t)c] = v; }
		}
	}
};
}

// klang::Effect (klang.h:4203-4217): `in` and `out` are the Modifier's signals; process() is the per-sample body (device code, recorded),
// prepare() runs once per block; process(buffer) is the host's entry point — the whole block goes to the GPU.
struct Effect : Plugin, Modifier {
	enum { channels = 1 };
	gpu::FxRunner gpu_fx;
	virtual void prepare() {}
	virtual void process() override { out = in; }                                  // klang.h:4206
	void run_process() { this->process(); }
	virtual void process(buffer buffer) {                                          // klang.h:4208-4216: prepare(); per sample { in = *buf; process(); *buf++ = out; }
		signal* ins[2] = { &in, &in }; signal* outs[2] = { &out, &out };
		gpu_fx.build(this, *this, controls, 1, ins, outs, false);
		float* ch[1] = { buffer.cursor() };
		gpu_fx.run(controls, ch, buffer.remaining());
	}
};
namespace Stereo {
	struct signal {                                                          // klang.h:4487-4558 (the operators the shipped effects use)
		klang::signal l, r;
		signal(klang::signal l_ = 0.f, klang::signal r_ = 0.f) : l(l_), r(r_) { reg(); }
		void reg() { l.reg_member(); r.reg_member(); }                             // (a member of an Effect — `Array<stereo::signal, 20> gains` — is state of its record: the channels were copy-constructed, which notes nothing)
		klang::signal& operator[](int c) { return c == 0 ? l : r; }                // klang.h:4530
		signal operator+(const signal& x) const { return { l + x.l, r + x.r }; } signal operator-(const signal& x) const { return { l - x.l, r - x.r }; }
		signal operator*(const signal& x) const { return { l * x.l, r * x.r }; } signal operator/(const signal& x) const { return { l / x.l, r / x.r }; }
		signal operator*(const klang::signal& x) const { return { l * x, r * x }; } signal operator/(const klang::signal& x) const { return { l / x, r / x }; }
		signal operator*(float x) const { return { l * x, r * x }; }
		signal(float x) : l(x), r(x) { reg(); } signal(int x) : l((float)x), r((float)x) { reg(); } signal(double x) : l((float)x), r((float)x) { reg(); }
		signal& operator+=(const signal& x) { l = l + x.l; r = r + x.r; return *this; }
		klang::signal mono() const { return (l + r) * 0.5f; }                      // klang.h:4556
	};
	inline signal operator*(Control& c, const signal& x) { return { c.value * x.l, c.value * x.r }; }
	// Stereo::frame (klang.h:4487-4517): one sample of each channel of a Stereo::buffer, by reference
	struct frame {
		klang::buffer::sample l, r;
		frame(klang::buffer::sample left, klang::buffer::sample right) : l(left), r(right) {}
		frame& operator+=(const frame& x) { l += klang::signal(x.l); r += klang::signal(x.r); return *this; } frame& operator-=(const frame& x) { l -= klang::signal(x.l); r -= klang::signal(x.r); return *this; }
		frame& operator*=(const frame& x) { l *= klang::signal(x.l); r *= klang::signal(x.r); return *this; } frame& operator/=(const frame& x) { l /= klang::signal(x.l); r /= klang::signal(x.r); return *this; }
		frame& operator+=(const signal& x) { l += x.l; r += x.r; return *this; } frame& operator-=(const signal& x) { l -= x.l; r -= x.r; return *this; }
		frame& operator*=(const signal& x) { l *= x.l; r *= x.r; return *this; } frame& operator/=(const signal& x) { l /= x.l; r /= x.r; return *this; }
		frame& operator=(const signal& x) { l = x.l; r = x.r; return *this; }
		frame& operator=(const klang::signal& x) { l = x; r = x; return *this; }
		operator signal() const { return { klang::signal(l), klang::signal(r) }; }
	};
	// Stereo::buffer (klang.h:4578-4640): two mono buffers advanced in lock-step (`left` by reference, `right` a copy, as declared there)
	struct buffer {
		typedef Stereo::signal signal;
		klang::buffer& left; klang::buffer right;
		buffer(const buffer& b) : left(b.left), right(b.right) { rewind(); }
		buffer(klang::buffer& l, klang::buffer& r) : left(l), right(r) { rewind(); }
		operator const signal() const { return { klang::signal(left), klang::signal(right) }; }
		operator frame() { return { klang::buffer::sample{ left.cursor() }, klang::buffer::sample{ right.cursor() } }; }
		bool finished() const { return left.finished() && right.finished(); }
		frame operator++(int) { return { left++, right++ }; }
		frame operator=(const signal& in) { return { left = in.l, right = in.r }; }
		buffer& operator=(const buffer& in) { left = in.left; right = in.right; return *this; }
		buffer& operator+=(const signal& in) { left += in.l; right += in.r; return *this; }
		buffer& operator=(const frame& in) { left = klang::signal(in.l); right = klang::signal(in.r); return *this; }
		buffer& operator+=(const frame& in) { left += klang::signal(in.l); right += klang::signal(in.r); return *this; }
		buffer& operator*=(const frame& in) { left *= klang::signal(in.l); right *= klang::signal(in.r); return *this; }
		buffer& operator=(const klang::signal in) { left = in; right = in; return *this; }
		buffer& operator+=(const klang::signal in) { left += in; right += in; return *this; }
		buffer& operator*=(const klang::signal in) { left *= in; right *= in; return *this; }
		frame operator[](int index) { return { left[index], right[index] }; }
		signal operator[](int index) const { const klang::buffer& l = left; const klang::buffer& r = right; return { klang::signal(l[index]), klang::signal(r[index]) }; }
		klang::buffer& channel(int index) { return index == 1 ? right : left; }
		void clear() { left.clear(); right.clear(); }
		void clear(int size) { left.clear(size); right.clear(size); }
		void rewind() { left.rewind(); right.rewind(); }
	};
	// Stereo::Modifier / Stereo::Bank<T> (klang.h:4560-4645): the types the shipped Reverb.k is written in.  They are here so that the
	// file compiles unchanged; a patch built from them is tied to its hand-written kernel (KLANG_GPU_BIND_FX) — their process() is
	// never recorded.
	struct Modifier {
		Stereo::signal in, out;
		virtual ~Modifier() {}
		virtual void process() { out = in; }
		void operator<<(const Stereo::signal& x) { in = x; process(); }
		operator const Stereo::signal&() { return out; }
		// `modifier * x` with x a signal / param: the reference's Modifier multiplies by ITS signal type, and x becomes a Stereo::signal through signals<2>'s variadic
		// constructor (klang.h:1238-1241: value{ x }), i.e. { x, 0 } — the right channel is multiplied by 0 (examples/Reverb.k:271 `(in >> reflections) * wet`: the wet
		// right channel is silent in the reference; reproduced, as the hand-written kernel does)
		Stereo::signal operator*(const klang::signal& x) { return { out.l * x, out.r * klang::signal(0.f) }; }
	};
	template<class TYPE> struct Bank {
		TYPE items[2];
		Stereo::signal out;
		template<typename... P> void set(P... p) { items[0].set(p...); items[1].set(p...); }
		void operator<<(const Stereo::signal& x) { x.l >> items[0]; x.r >> items[1]; out = { klang::signal(items[0]), klang::signal(items[1]) }; }
		operator const Stereo::signal&() { return out; }
		TYPE& operator[](int i) { return items[i]; }
	};
	// Stereo::Delay<SIZE> (klang.h:4646-4699): a left and a right Delay<SIZE> advanced together
	template<int SIZE> struct Delay {
		klang::Delay<SIZE> items[2]; klang::Delay<SIZE>& l; klang::Delay<SIZE>& r;
		Delay() : l(items[0]), r(items[1]) {}
		void operator<<(const signal& x) { x.l >> items[0]; x.r >> items[1]; }                  // `delay << out`: each line takes its channel
		signal operator()(const signal& time) { return { items[0](time.l), items[1](time.r) }; }   // stereo time: a tap per side (klang.h:4692-4693)
		template<typename TIME, std::enable_if_t<!std::is_same_v<TIME, signal>, int> = 0>
		signal operator()(const TIME& time) {                                                   // klang.h:4686-4696: tap(int) for integers; otherwise Stereo::Delay::tap(float) — both lines at one cursor, its own interpolation form
			if constexpr (std::is_integral_v<TIME>) return { items[0](time), items[1](time) };
			else return { items[0].tap_stereo_form(time), items[1].tap_stereo_form(time) };
		}
		unsigned int max() const { return SIZE; }
	};
	inline signal& operator>>(const signal& x, signal& dst) { dst = x; return dst; }
	// Stereo::Effect (klang.h:4703-4717)
	struct Effect : Plugin {
		enum { channels = 2 };
		Stereo::signal in, out;
		gpu::FxRunner gpu_fx;
		virtual ~Effect() {}
		virtual void prepare() {}
		virtual void process() { out = in; }                                       // klang.h:4707
		void run_process() { this->process(); }
		virtual void process(Stereo::buffer buffer) {                              // klang.h:4708-4716
			klang::signal* ins[2] = { &in.l, &in.r }; klang::signal* outs[2] = { &out.l, &out.r };
			gpu_fx.build(this, *this, controls, 2, ins, outs, false);
			float* ch[2] = { buffer.left.cursor(), buffer.right.cursor() };
			const int n = buffer.left.remaining() < buffer.right.remaining() ? buffer.left.remaining() : buffer.right.remaining();
			gpu_fx.run(controls, ch, n);
		}
	};
}
namespace stereo = Stereo;
namespace Mono { typedef klang::Modifier Modifier; typedef klang::Generator Generator; typedef klang::signal signal; typedef klang::buffer buffer; }
namespace mono = Mono;

// =================================================================================================
// Notes
// =================================================================================================
namespace gpu {
// A note the host uses on its own (`MyNote note; note.start(p, v); if (!note.process(buffer)) note.stop();`, klang.h:4295-4306 and the
// reference's README): a one-voice bank built from the note's construction log the first time it is needed.
struct SoloVoice {
	klg_synth* h = nullptr; GraphLayout layout; std::vector<uint32_t> words; std::vector<float> pv; Controls no_controls; bool rendered = false;
	~SoloVoice() { if (h) klg_synth_destroy(h); }
};
template<class T> struct NoteHooks {                                       // does the Note type T define control() / preset() / midi() of its own?
	struct Probe : T {
		static constexpr bool control_() { return !std::is_same_v<decltype(&Probe::control), void (Controller::*)(int, float)>; }
		static constexpr bool preset_() { return !std::is_same_v<decltype(&Probe::preset), void (Controller::*)(int)>; }
		static constexpr bool midi_() { return !std::is_same_v<decltype(&Probe::midi), void (Controller::*)(int, int, int)>; }
	};
	static constexpr unsigned mask() { return (Probe::control_() ? 1u : 0u) | (Probe::preset_() ? 2u : 0u) | (Probe::midi_() ? 4u : 0u); }
};
}

template<class SYNTH> class NoteBase : public Controller, public gpu::Owner {
	SYNTH* synth = nullptr;
protected:
	virtual event on(Pitch, Velocity) {}
	virtual event off(Velocity = 0) { stage = Off; }
	SYNTH* getSynth() { return synth; }
public:
	struct ControlsRef { Controls* c = nullptr; Control& operator[](int i) { return (*c)[i]; } unsigned size() { return c ? c->size() : 0; } } controls;
	Pitch pitch; Velocity velocity;
	enum Stage { Onset, Sustain, Release, Off } stage = Off;
	gpu::SoloVoice* solo = nullptr;                                         // only when the host drives this note by itself (no Synth)
	NoteBase() : gpu::Owner(false) {}
	virtual ~NoteBase() { delete solo; }
	bool attached() const { return synth != nullptr; }
	void attach(SYNTH* s) { synth = s; controls.c = &s->controls; init(); }      // klang.h:4245-4249
	virtual void init() {}
	virtual void solo_pull() {}                                             // lane -> host mirror / host mirror -> lane of a note used on its own
	virtual void solo_push() {}
	virtual void start(Pitch p, Velocity v) { if (!synth) solo_pull(); stage = Onset; pitch = p; velocity = v; on(pitch, velocity); stage = Sustain; if (!synth) solo_push(); }    // klang.h:4257-4263
	virtual bool release(Velocity v = 0) {                                   // klang.h:4265-4275
		if (stage == Off) return true;
		if (stage != Release) { if (!synth) solo_pull(); stage = Release; off(v); if (!synth) solo_push(); }
		return stage == Off;
	}
	virtual bool stop(Velocity = 0) {
		if (gpu::Recorder* r = gpu::recording()) {                 // `if (adsr.finished()) stop();` / `stop();` in a recorded process()
			r->emit(klg::graph::OP_STOP, -1, -1, -1, 0, false);
			return true;
		}
		stage = Off; if (!synth && solo) solo_push(); return true;
	}
	bool finished() const { return stage == Off; }
	virtual void controlChange(int controller, int value) { midi(0xB0, controller, value); }   // klang.h:4289
};

namespace gpu {
// the block of a note used on its own: record + create on first use, then one klg_process_voices per <= 1024 samples
template<class NOTE> inline void solo_ensure(NOTE* note) {
	if (note->solo) return;
	close_log();
	SoloVoice* s = note->solo = new SoloVoice();
	const char* lo = (const char*)dynamic_cast<const void*>(note);
	record_note(note, member_objs(note->log, nullptr, nullptr), s->no_controls, lo, s->layout, typeid(*note).name());
	s->h = klg_synth_create_graph(s->layout.program.c_str(), 1, 1, fs.f, 1024);
	if (!s->h) { std::fprintf(stderr, "klang-mi355: klg_synth_create_graph: %s\n", klg_last_error()); std::abort(); }
	for (size_t k = 0; k < s->layout.tables.size(); k++) if (klg_table_upload(s->h, s->layout.tables[k].data(), (int)s->layout.tables[k].size(), 0) != (int)k + 1) { std::fprintf(stderr, "klang-mi355: klg_table_upload: %s\n", klg_last_error()); std::abort(); }
	s->words.assign(klg_synth_state_bytes(s->h) / 4, 0u);
}
template<class NOTE> inline void solo_pull(NOTE* note) {
	solo_ensure(note);
	SoloVoice* s = note->solo; const char* lo = (const char*)dynamic_cast<const void*>(note);
	if (klg_voice_download(s->h, 0, s->words.data(), s->words.size() * 4)) { std::fprintf(stderr, "klang-mi355: klg_voice_download: %s\n", klg_last_error()); std::abort(); }
	bool used = (s->words[0] & 3u) != (uint32_t)klg::ST_OFF || note->stage != NOTE::Off;     // (see SynthCore::with_voice)
	for (size_t w = 1; w < s->words.size() && !used; w++) used = s->words[w] != 0u;
	if (used) s->layout.unpack((void*)lo, s->words.data());
	upload_target = s->h; current_voice = 0; current_note = lo; current_layout = &s->layout;
}
template<class NOTE> inline void solo_push(NOTE* note) {
	solo_ensure(note);
	SoloVoice* s = note->solo; const char* lo = (const char*)dynamic_cast<const void*>(note);
	upload_target = s->h; current_voice = 0; current_note = lo; current_layout = &s->layout;
	s->layout.pack(lo, s->words.data());
	s->words[0] = (s->words[0] & ~3u) | (uint32_t)note->stage;
	if (klg_voice_upload(s->h, 0, s->words.data(), s->words.size() * 4)) { std::fprintf(stderr, "klang-mi355: klg_voice_upload: %s\n", klg_last_error()); std::abort(); }
}
// renders the note's next n samples into s->pv and brings the note's stage back (a note the GPU stopped is Off)
template<class NOTE> inline const float* solo_render(NOTE* note, int at, int m) {
	(void)at;
	SoloVoice* s = note->solo;
	s->pv.resize((size_t)m * (size_t)klg_synth_note_channels(s->h));       // [channels][m] for a note with a stereo out
	if (klg_process_voices(s->h, s->pv.data(), nullptr, 0, m)) { std::fprintf(stderr, "klang-mi355: klg_process_voices: %s\n", klg_last_error()); std::abort(); }
	uint8_t st = 0;
	if (klg_voice_stages(s->h, &st, 1)) { std::fprintf(stderr, "klang-mi355: klg_voice_stages: %s\n", klg_last_error()); std::abort(); }
	if (st == klg::ST_OFF) note->stage = NOTE::Off;
	return s->pv.data();
}
[[noreturn]] inline void note_of_a_synth() {
	std::fprintf(stderr, "klang-mi355: Note::process(buffer) on a note that belongs to a Synth: the Synth renders all its voices in one launch (Synth::process)\n"); std::abort();
}
}

// =================================================================================================
// Synth: host voice allocation + event dispatch; blocks rendered by libklang_mi355.so
// =================================================================================================
namespace gpu { enum MixMode { Sum = 0, LastActiveVoice = 1 }; }          // how the voices of a MONO Synth combine (see klang::Synth below)

template<class NOTEBASE> struct SynthCore : Plugin {
	struct Slot { NOTEBASE* note = nullptr; NoteBinding b = { -1, nullptr, nullptr }; const gpu::GraphLayout* graph = nullptr; const char* lo = nullptr; int type = 0; };   // lo: the most derived object's address; type: which notes.add<T>() made it
	// SEVERAL NOTE TYPES (round 6; Notes::add<TYPE>, klang.h:4323-4330, takes several): slots in add order, `assign()` over all of them as the reference's (4336-4372); every
	// type has its own kernel and record layout, hence its own bank of `count` voices in which only that type's slots ever sound.  A block renders every bank into the same
	// buffers.  What notes of different types would share through their Synth — a smooth()ed control, the rand() sequence of Noise generators — is ordered per bank, not per slot.
	struct TypeBank { klg_synth* bank = nullptr; std::vector<uint32_t> words; std::vector<uint8_t> stages; };
	std::vector<TypeBank> types;
	// NOTE VARIANTS (-DKLANG_GPU_NOTE_VARIANTS).  A recorded body is one program per Note TYPE — but a note's process() may depend on HOST state that its on() sets: a pointer to
	// one of several member oscillators (examples/Additive/Inheritance.k: `Additive* osc` chosen by a Menu in on(), `*osc >> out` in process()), an `int` that selects a
	// branch.  With the switch nothing is recorded at notes.add<T>(); instead every event of a note (noteOn, noteOff, a hook) is followed by a recording of THAT note's
	// process() with its host state as the event left it.  Equal program texts share a bank ("variant"); a note whose text changed moves to the other variant's bank (its
	// record travels through the host mirror, as at any event); a block renders every variant's bank into the same buffers.  What notes of DIFFERENT variants would share
	// through their Synth — a smooth()ed control, the rand() sequence of Noise generators — is ordered per bank, not per slot: such patches keep to one variant.
#ifdef KLANG_GPU_NOTE_VARIANTS
	static constexpr bool kVariants = true;
#else
	static constexpr bool kVariants = false;
#endif
	struct Variant { gpu::GraphLayout* layout = nullptr; klg_synth* bank = nullptr; std::vector<uint32_t> words; std::vector<uint8_t> stages; };
	std::vector<Variant> variants; std::vector<int> slot_variant;
	struct NotesT {
		std::vector<gpu::Obj> proto_objs; const char* proto_lo = nullptr; size_t proto_size = 0; std::string type_name;   // (variants: the prototype's members, re-based onto each note)
		SynthCore* owner; std::vector<Slot> items; unsigned noteOns = 0; unsigned noteStart[128] = { 0 };
		unsigned count = 0;
		std::vector<gpu::GraphLayout*> layouts;
		const std::type_info* note_type = nullptr; unsigned hooks = 0;
		std::vector<const std::type_info*> type_ids; std::vector<const gpu::GraphLayout*> type_layout;
		struct Proto { std::vector<gpu::Obj> objs; const char* lo = nullptr; size_t size = 0; std::string name; };   // a recorded type's prototype: its members, re-based onto each note (check_body)
		std::vector<Proto> type_proto;
		template<class T> void add(int n) {
			gpu::close_log();
			int ti = -1;
			for (size_t k = 0; k < type_ids.size(); k++) if (*type_ids[k] == typeid(T)) ti = (int)k;
			if (ti < 0) { ti = (int)type_ids.size(); type_ids.push_back(&typeid(T)); type_layout.push_back(nullptr); type_proto.push_back(Proto()); }
			cur_type = ti;
			if (owner->gpu) { std::fprintf(stderr, "klang-mi355: notes.add<%s>() after the Synth's first event or block: the banks exist by then (add every Note type in the Synth's constructor)\n", typeid(T).name()); std::abort(); }
			// (per-note recorded bodies, -DKLANG_GPU_NOTE_VARIANTS, keep to one Note type: a variant is a program of THE type's members)
			if (kVariants && note_type && *note_type != typeid(T)) { std::fprintf(stderr, "klang-mi355: notes.add<%s>() after notes.add<%s>() with -DKLANG_GPU_NOTE_VARIANTS: per-note recorded bodies keep to ONE Note type (build without the switch, or use a Synth per type)\n", typeid(T).name(), note_type->name()); std::abort(); }
			note_type = &typeid(T); hooks |= gpu::NoteHooks<T>::mask();
			const bool bound = klang_gpu_patch((const T*)nullptr) >= 0 && !std::getenv("KLANG_MI355_FORCE_GRAPH");
			const gpu::GraphLayout* layout = type_layout[(size_t)ti];
			type_name = typeid(T).name();
			for (int i = 0; i < n && items.size() < 128; i++) {
				T* t = nullptr;
				if (kVariants && !bound) {                                        // (see NOTE VARIANTS above: members noted once, nothing recorded yet)
					if (!proto_lo) { gpu::Recorder C; gpu::rec = &C; C.constructing = true; t = new T(); C.constructing = false; gpu::rec = nullptr; proto_lo = (const char*)t; proto_size = sizeof(T); proto_objs = gpu::member_objs(C, proto_lo, proto_lo + sizeof(T)); }
					else { gpu::log_suppress++; t = new T(); gpu::log_suppress--; }
					t->attach(static_cast<typename T::synth_type*>(owner));
				}
				else if (!bound && !layout) { gpu::GraphLayout* l = new gpu::GraphLayout(); layouts.push_back(l); t = record<T>(*l); layout = l; type_layout[(size_t)ti] = l; }   // the prototype becomes the type's first note
				else { gpu::log_suppress++; t = new T(); gpu::log_suppress--; t->attach(static_cast<typename T::synth_type*>(owner)); }
				Slot s; s.note = t; s.graph = layout; s.lo = (const char*)t; s.type = ti;
				s.b.patch = bound ? klang_gpu_patch((const T*)t) : -1;
				s.b.pack = [](const void* p, uint32_t* w) { klang_gpu_pack((const T*)p, w); };
				s.b.unpack = [](void* p, const uint32_t* w) { klang_gpu_unpack((T*)p, w); };
				items.push_back(s); count = (unsigned)items.size();
			}
		}
		// Construct the prototype Note with every primitive / signal member announcing itself, run its process() once in
		// recording mode, and turn what was recorded into a graph program + the member layout of the Note type.
		int cur_type = 0;
		template<class T> T* record(gpu::GraphLayout& L) {
			gpu::Recorder C;
			gpu::rec = &C; C.constructing = true;
			T* t = new T();
			C.constructing = false; gpu::rec = nullptr;
			t->attach(static_cast<typename T::synth_type*>(owner));
			const char* lo = (const char*)t;
			NOTEBASE* nb = t;
			Proto& P = type_proto[(size_t)cur_type];
			P.objs = gpu::member_objs(C, lo, lo + sizeof(T)); P.lo = lo; P.size = sizeof(T); P.name = typeid(T).name();
			gpu::record_note(nb, P.objs, owner->controls, lo, L, typeid(T).name());
			return t;
		}
		NOTEBASE* operator[](int i) { return items[(size_t)i].note; }
		int assign() {                                                       // Notes::assign klang.h:4336-4372
			for (unsigned i = 0; i < count; i++) if (items[i].note->stage == NOTEBASE::Off) { noteStart[i] = noteOns++; return (int)i; }
			int oldest = -1; unsigned oldest_start = 0;
			for (unsigned i = 0; i < count; i++) if (items[i].note->stage == NOTEBASE::Release && (oldest == -1 || noteStart[i] < oldest_start)) { oldest = (int)i; oldest_start = noteStart[i]; }
			if (oldest != -1) { noteStart[oldest] = noteOns++; return oldest; }
			oldest = -1; oldest_start = 0;
			for (unsigned i = 0; i < count; i++) if (oldest == -1 || noteStart[i] < oldest_start) { oldest = (int)i; oldest_start = noteStart[i]; }
			noteStart[oldest] = noteOns++;
			return oldest;
		}
		~NotesT() { for (auto& s : items) delete s.note; for (auto* l : layouts) delete l; }
	} notes;
	klg_synth* gpu = nullptr;
	std::vector<uint32_t> words;
	std::vector<uint8_t> stages;
	gpu::MixMode mix = gpu::Sum;
	gpu::FxRunner post;                                                      // the Synth's own process() (post-processing of the mix), if it has one

	SynthCore() { notes.owner = this; }
	~SynthCore() { if (kVariants) { for (auto& v : variants) { if (v.bank) klg_synth_destroy(v.bank); delete v.layout; } } else for (auto& t : types) if (t.bank) klg_synth_destroy(t.bank); }

	virtual bool mono_synth() const { return false; }
	void fail(const char* what) { std::fprintf(stderr, "klang-mi355: %s: %s\n", what, klg_last_error()); std::abort(); }
	void ensure_gpu() {
		if (gpu) return;
		gpu::close_log();
		if (kVariants && notes.count && notes.proto_lo) {                       // (banks are made per variant, at the first event that needs one)
			if (slot_variant.size() != notes.count) slot_variant.assign(notes.count, -1);
			if (variants.empty()) if (const char* e = std::getenv("KLANG_MI355_MONO_MIX")) if (mono_synth() && (!std::strcmp(e, "last") || !std::strcmp(e, "reference"))) mix = gpu::LastActiveVoice;
			return;
		}
		if (!notes.count) { std::fprintf(stderr, "klang-mi355: Synth has no notes (call notes.add<T>(n))\n"); std::abort(); }
		if (const char* e = std::getenv("KLANG_MI355_MONO_MIX")) if (mono_synth() && (!std::strcmp(e, "last") || !std::strcmp(e, "reference"))) mix = gpu::LastActiveVoice;   // (no source change needed to get the reference's literal mono behaviour)
		types.assign(notes.type_ids.size(), TypeBank());
		for (size_t ti = 0; ti < types.size(); ti++) {                          // one bank per Note type (see SEVERAL NOTE TYPES above), every bank with all `count` slots
			const Slot* first = nullptr;
			for (const Slot& sl : notes.items) if (sl.type == (int)ti) { first = &sl; break; }
			if (!first) continue;
			klg_synth* bank = nullptr;
			if (const gpu::GraphLayout* g = first->graph) {                      // recorded process(): compiled for gfx950 now (hipRTC)
				bank = klg_synth_create_graph(g->program.c_str(), 1, (int)notes.count, fs.f, 1024);
				if (!bank) fail("klg_synth_create_graph");
				for (size_t k = 0; k < g->tables.size(); k++) if (klg_table_upload(bank, g->tables[k].data(), (int)g->tables[k].size(), 0) != (int)k + 1) fail("klg_table_upload (Table read by process())");
			}
			else {
				if (first->b.patch < 0) { std::fprintf(stderr, "klang-mi355: no GPU kernel is bound to this Note type\n"); std::abort(); }
				bank = klg_synth_create(first->b.patch, 1, (int)notes.count, fs.f, 1024);
				if (!bank) fail("klg_synth_create");
			}
			if (klg_synth_set_mix_mode(bank, (int)mix)) fail("klg_synth_set_mix_mode");   // (LastActiveVoice with several types: the bank's own last sounding slot; render_voices() keeps the block of the bank that holds the synth's last one)
			types[ti].bank = bank;
			types[ti].words.resize(klg_synth_state_bytes(bank) / 4);
			types[ti].stages.assign(notes.count, (uint8_t)klg::ST_OFF);
			if (klg_synth_note_channels(bank) != klg_synth_note_channels(types[0].bank)) { std::fprintf(stderr, "klang-mi355: the Note types of one Synth must all have a mono or all a stereo `out`\n"); std::abort(); }
		}
		gpu = types[0].bank;
		stages.resize(notes.count);
		sync_controls();
		push_smoothed();
	}
	void sync_controls() { if (kVariants && notes.proto_lo) { for (auto& v : variants) for (unsigned c = 0; c < controls.items.size() && (int)c < klg_synth_controls(v.bank); c++) klg_set_control(v.bank, 0, (int)c, controls.items[c].value.value); return; }
		for (auto& t : types) for (unsigned c = 0; c < controls.items.size() && (int)c < klg_synth_controls(t.bank); c++) klg_set_control(t.bank, 0, (int)c, controls.items[c].value.value); }
	// Control::smoothed (klang.h:1707): the bank advances it (every sounding note's smooth() calls, in order); the host objects follow
	void push_smoothed() { if (kVariants && notes.proto_lo) return; for (auto& t : types) for (unsigned c = 0; c < controls.items.size() && (int)c < klg_synth_controls(t.bank); c++) klg_set_control_smoothed(t.bank, 0, (int)c, controls.items[c].smoothed.value); }
	void pull_smoothed() { if (kVariants && notes.proto_lo) return; for (unsigned c = 0; c < controls.items.size() && (int)c < klg_synth_controls(gpu); c++) klg_get_control_smoothed(gpu, 0, (int)c, &controls.items[c].smoothed.value); }
	// ---- NOTE VARIANTS: the event path and the block ----
	std::vector<gpu::Obj> rebased_objs(const Slot& s) const {
		std::vector<gpu::Obj> objs = notes.proto_objs;
		const ptrdiff_t d = s.lo - notes.proto_lo;
		auto move = [&](const void* p) -> const void* { const char* c = (const char*)p; return (c >= notes.proto_lo && c < notes.proto_lo + notes.proto_size) ? c + d : c; };
		for (auto& o : objs) { o.addr = move(o.addr); if (o.packable) o.packable = (const gpu::Packable*)move(o.packable); if (o.key) o.key = move(o.key); if (o.live_arg) o.live_arg = (const int*)move(o.live_arg); }
		return objs;
	}
	int variant_of(gpu::GraphLayout* L) {                                    // takes L (kept by a new variant, deleted when its text is known)
		for (size_t i = 0; i < variants.size(); i++) if (variants[i].layout->program == L->program) { delete L; return (int)i; }
		Variant v; v.layout = L;
		v.bank = klg_synth_create_graph(L->program.c_str(), 1, (int)notes.count, fs.f, 1024);
		if (!v.bank) fail("klg_synth_create_graph (note variant)");
		for (size_t k = 0; k < L->tables.size(); k++) if (klg_table_upload(v.bank, L->tables[k].data(), (int)L->tables[k].size(), 0) != (int)k + 1) fail("klg_table_upload (Table read by process())");
		if (klg_synth_set_mix_mode(v.bank, (int)mix)) fail("klg_synth_set_mix_mode (note variant)");   // (LastActiveVoice: the bank's own last sounding slot; render_voices() keeps the block of the bank that holds the synth's last one)
		v.words.assign(klg_synth_state_bytes(v.bank) / 4, 0u); v.stages.assign(notes.count, (uint8_t)klg::ST_OFF);
		variants.push_back(v);
		if (!gpu) gpu = v.bank;                                               // (what asks a bank for the number of controls)
		for (unsigned c = 0; c < controls.items.size() && (int)c < klg_synth_controls(v.bank); c++) { klg_set_control(v.bank, 0, (int)c, controls.items[c].value.value); klg_set_control_smoothed(v.bank, 0, (int)c, controls.items[c].smoothed.value); }
		return (int)variants.size() - 1;
	}
	template<class F> void with_voice_variant(int n, F&& event_code) {
		Slot& s = notes.items[(size_t)n];
		const int v = slot_variant[(size_t)n];
		if (v >= 0) {
			Variant& V = variants[(size_t)v];
			if (klg_voice_download(V.bank, n, V.words.data(), V.words.size() * 4)) fail("klg_voice_download");
			bool used = (V.words[0] & 3u) != (uint32_t)klg::ST_OFF || s.note->stage != NOTEBASE::Off;
			for (size_t w = 1; w < V.words.size() && !used; w++) used = V.words[w] != 0u;
			if (used) V.layout->unpack(s.note, V.words.data());
			gpu::upload_target = V.bank; gpu::current_voice = n; gpu::current_note = s.note; gpu::current_layout = V.layout;
		}
		event_code(s.note);
		gpu::GraphLayout* L = new gpu::GraphLayout();
		gpu::record_note(s.note, rebased_objs(s), controls, s.lo, *L, notes.type_name.c_str());   // process() with the host state this event left
		const int v2 = variant_of(L);
		Variant& W = variants[(size_t)v2];
		if (v >= 0 && v2 != v) {                                               // the voice leaves its old variant's bank (Off there) with everything its record held
			Variant& V = variants[(size_t)v];
			if (V.words.size() == W.words.size()) W.words = V.words;
			std::vector<uint32_t> off(V.words.size(), 0u); off[0] = (uint32_t)klg::ST_OFF;
			if (klg_voice_upload(V.bank, n, off.data(), off.size() * 4)) fail("klg_voice_upload");
		}
		else if (v < 0) std::fill(W.words.begin(), W.words.end(), 0u);
		W.layout->pack(s.note, W.words.data());
		W.words[0] = (W.words[0] & ~3u) | (uint32_t)s.note->stage;
		if (klg_voice_upload(W.bank, n, W.words.data(), W.words.size() * 4)) fail("klg_voice_upload");
		slot_variant[(size_t)n] = v2; s.graph = W.layout;
	}
	// A recorded body is ONE program per Note type, recorded from the prototype at notes.add<T>().  A process() that follows HOST state of its note — a pointer or an `int` that
	// on() sets and process() branches on (Additive/Inheritance.k, Subtractive/Modular.k), a plain `float` member used as a factor — would silently play the prototype's body for
	// every note.  So after every event the note's process() is recorded AGAIN, with its host state as the event left it, and compared with the bank's program: a difference
	// stops the run with the switch that handles it (-DKLANG_GPU_NOTE_VARIANTS: one bank per body).  -DKLANG_GPU_TRUST_BODIES skips the check (a host that has run it once).
	void check_body(const Slot& s) {
#ifndef KLANG_GPU_TRUST_BODIES
		if (!s.graph || (size_t)s.type >= notes.type_proto.size() || !notes.type_proto[(size_t)s.type].lo) return;
		const typename NotesT::Proto& P = notes.type_proto[(size_t)s.type];
		std::vector<gpu::Obj> objs = P.objs;
		const ptrdiff_t d = s.lo - P.lo;
		auto move = [&](const void* p) -> const void* { const char* c = (const char*)p; return (c >= P.lo && c < P.lo + P.size) ? c + d : c; };
		for (auto& o : objs) { o.addr = move(o.addr); if (o.packable) o.packable = (const gpu::Packable*)move(o.packable); if (o.key) o.key = move(o.key); if (o.live_arg) o.live_arg = (const int*)move(o.live_arg); }
		gpu::GraphLayout L2;
		gpu::quiet_recording = true;
		gpu::record_note(s.note, std::move(objs), controls, s.lo, L2, P.name.c_str());
		gpu::quiet_recording = false;
		if (L2.program == s.graph->program) return;
		std::fprintf(stderr, "klang-mi355: %s::process() follows HOST state of its note: recorded again after this event it is a different program from the one recorded at notes.add<T>() "
			"(a pointer, an int or a plain float that on() / off() / a hook sets and process() reads).  One program per Note type cannot play that: compile with -DKLANG_GPU_NOTE_VARIANTS "
			"(one bank per body), or keep such state in `param` / `signal` members.\n", P.name.c_str());
		if (std::getenv("KLANG_MI355_DUMP_GRAPH")) std::fprintf(stderr, "---- at notes.add<T>():\n%s---- after this event:\n%s", s.graph->program.c_str(), L2.program.c_str());
		std::abort();
#else
		(void)s;
#endif
	}
	// host mirror <- lane ; run the event ; lane <- host mirror
	template<class F> void with_voice(int n, F&& event_code) {
		ensure_gpu();
		if (kVariants && notes.proto_lo) { with_voice_variant(n, event_code); return; }
		Slot& s = notes.items[(size_t)n];
		klg_synth* const gpu = types[(size_t)s.type].bank; std::vector<uint32_t>& words = types[(size_t)s.type].words;   // the slot's own type's bank
		if (klg_voice_download(gpu, n, words.data(), words.size() * 4)) fail("klg_voice_download");
		// the lane's record -> the host mirror.  A note keeps ALL its member state from one note to the next in the reference (filter memories,
		// oscillator phases, delay cursors: nothing is reset unless on() does it), so a slot that has sounded before is unpacked whatever its
		// stage; only a slot that was never started — an all-zero record — leaves the freshly constructed object as it is
		bool used = (words[0] & 3u) != (uint32_t)klg::ST_OFF || s.note->stage != NOTEBASE::Off;
		for (size_t w = 1; w < words.size() && !used; w++) used = words[w] != 0u;
		if (used) { if (s.graph) s.graph->unpack(s.note, words.data()); else s.b.unpack(s.note, words.data()); }
		gpu::upload_target = gpu; gpu::current_voice = n; gpu::current_note = s.note; gpu::current_layout = s.graph;   // Wavetable uploads / Delay::clear() of this voice
		event_code(s.note);
		check_body(s);
		if (s.graph) s.graph->pack(s.note, words.data()); else s.b.pack(s.note, words.data());
		words[0] = (words[0] & ~3u) | (uint32_t)s.note->stage;
		if (klg_voice_upload(gpu, n, words.data(), words.size() * 4)) fail("klg_voice_upload");
	}
	virtual event noteOn(int pitch, float velocity) {                        // klang.h:4423-4427
		ensure_gpu(); refresh_stages();
		const int n = notes.assign();
		with_voice(n, [&](NOTEBASE* note) { note->start((float)pitch, velocity); });
	}
	virtual event noteOff(int pitch, float velocity) {                       // klang.h:4430-4434
		ensure_gpu();
		for (unsigned n = 0; n < notes.count; n++)
			if (notes[(int)n]->pitch == pitch && notes[(int)n]->stage == NOTEBASE::Sustain)
				with_voice((int)n, [&](NOTEBASE* note) { note->release(velocity); });
	}
	// events go to the synth and to every sounding note (klang.h:4399-4421, 4779-4811).  A note's own hook runs on the host mirror of
	// its voice (lane -> host -> hook -> lane), and only for a Note type that defines the hook at all (gpu::NoteHooks).
	template<class F> void to_sounding_notes(unsigned hook, F&& f) {
		if (!(notes.hooks & hook) || !notes.count) return;
		ensure_gpu(); refresh_stages();
		for (unsigned n = 0; n < notes.count; n++) if (notes[(int)n]->stage != NOTEBASE::Off) with_voice((int)n, f);
	}
	event onControl(int index, float value) override { control(index, value); to_sounding_notes(1u, [&](NOTEBASE* note) { note->onControl(index, value); }); if (gpu) sync_controls(); }
	event onPreset(int index) override { preset(index); to_sounding_notes(2u, [&](NOTEBASE* note) { note->onPreset(index); }); if (gpu) sync_controls(); }
	event onMIDI(int status, int byte1, int byte2) override { midi(status, byte1, byte2); to_sounding_notes(4u, [&](NOTEBASE* note) { note->onMIDI(status, byte1, byte2); }); }
	// MIDI input for external calls (templates/juce/synth/Source/klang.h:3921-3928)
	virtual void input(int status, int byte1, int byte2) {
		if (status == 0x90 && byte2 > 0) noteOn(byte1, byte2 / 127.f);
		else if (status == 0x80 || (status == 0x90 && byte2 == 0)) noteOff(byte1, byte2 / 127.f);
		else onMIDI(status, byte1, byte2);
	}
	void refresh_stages() {
		if (kVariants && notes.proto_lo) {
			for (size_t v = 0; v < variants.size(); v++) {
				if (klg_voice_stages(variants[v].bank, variants[v].stages.data(), (int)notes.count)) fail("klg_voice_stages");
				for (unsigned n = 0; n < notes.count; n++) if (slot_variant[n] == (int)v && variants[v].stages[n] == klg::ST_OFF) notes[(int)n]->stage = NOTEBASE::Off;
			}
			return;
		}
		for (size_t ti = 0; ti < types.size(); ti++) {
			if (klg_voice_stages(types[ti].bank, types[ti].stages.data(), (int)notes.count)) fail("klg_voice_stages");
			for (unsigned n = 0; n < notes.count; n++) if (notes.items[n].type == (int)ti) { stages[n] = types[ti].stages[n]; if (stages[n] == klg::ST_OFF) notes[(int)n]->stage = NOTEBASE::Off; }    // `if (!note->process(..)) note->stop()`
		}
	}
	std::vector<float> unheard;                                               // (note variants + LastActiveVoice: where the banks that are not heard render)
	float* per_voice_sink = nullptr;                                          // tests: every voice's own block ([voice][note channels][length]) is written here too
	void render_voices(float* const* buffers, int channels, int length) {
		ensure_gpu();
		sync_controls();
		if (kVariants && notes.proto_lo) {                                       // every variant's bank adds its sounding voices to the block
			std::vector<float> tmp;
			// a mono Synth with mix = LastActiveVoice (klang.h:4450-4457: every sounding note overwrites the block in slot order): the block is the one of
			// the bank that holds the synth's LAST sounding slot — the stages are the host's own at this point (events run on the host mirror, the end of
			// the last block came back in refresh_stages()); every other bank renders too (its notes' state moves on) into samples nobody hears.
			int heard = -1;
			if (mix == gpu::LastActiveVoice) for (unsigned n = 0; n < notes.count; n++) if (notes[(int)n]->stage != NOTEBASE::Off && slot_variant[n] >= 0) heard = slot_variant[n];
			float* sink[2] = { nullptr, nullptr };                                  // (`unheard` is a member: nothing is allocated per block once it has its size)
			if (mix == gpu::LastActiveVoice) { if (unheard.size() < (size_t)channels * (size_t)length) unheard.resize((size_t)channels * (size_t)length); for (int c = 0; c < channels && c < 2; c++) sink[c] = unheard.data() + (size_t)c * (size_t)length; }
			for (size_t v = 0; v < variants.size(); v++) {
				float* const* dst = (mix == gpu::LastActiveVoice && (int)v != heard) ? sink : buffers;
				if (per_voice_sink) {
					const size_t row = (size_t)klg_synth_note_channels(variants[v].bank) * (size_t)length;
					tmp.assign(row * notes.count, 0.f);
					if (klg_process_voices(variants[v].bank, tmp.data(), dst, channels, length)) fail("klg_process_voices");
					for (unsigned n = 0; n < notes.count; n++) if (slot_variant[n] == (int)v) std::memcpy(per_voice_sink + (size_t)n * row, tmp.data() + (size_t)n * row, row * sizeof(float));
				}
				else if (klg_process(variants[v].bank, dst, channels, length, nullptr)) fail("klg_process");
			}
			if (per_voice_sink) for (unsigned n = 0; n < notes.count; n++) if (slot_variant[n] < 0) std::memset(per_voice_sink + (size_t)n * (size_t)length * (variants.empty() ? 1 : (size_t)klg_synth_note_channels(variants[0].bank)), 0, (size_t)length * (variants.empty() ? 1 : (size_t)klg_synth_note_channels(variants[0].bank)) * sizeof(float));
			refresh_stages();
			return;
		}
		if (types.size() > 1) {                                                  // several Note types: every type's bank adds its sounding voices to the block (as the variants' banks above)
			std::vector<float> tmp;
			int heard = -1;
			if (mix == gpu::LastActiveVoice) for (unsigned n = 0; n < notes.count; n++) if (notes[(int)n]->stage != NOTEBASE::Off) heard = notes.items[n].type;
			float* sink[2] = { nullptr, nullptr };
			if (mix == gpu::LastActiveVoice) { if (unheard.size() < (size_t)channels * (size_t)length) unheard.resize((size_t)channels * (size_t)length); for (int c = 0; c < channels && c < 2; c++) sink[c] = unheard.data() + (size_t)c * (size_t)length; }
			const size_t row = (size_t)klg_synth_note_channels(gpu) * (size_t)length;
			for (size_t ti = 0; ti < types.size(); ti++) {
				float* const* dst = (mix == gpu::LastActiveVoice && (int)ti != heard) ? sink : buffers;
				if (per_voice_sink) {
					tmp.assign(row * notes.count, 0.f);
					if (klg_process_voices(types[ti].bank, tmp.data(), dst, channels, length)) fail("klg_process_voices");
					for (unsigned n = 0; n < notes.count; n++) if (notes.items[n].type == (int)ti) std::memcpy(per_voice_sink + (size_t)n * row, tmp.data() + (size_t)n * row, row * sizeof(float));
				}
				else if (klg_process(types[ti].bank, dst, channels, length, nullptr)) fail("klg_process");
			}
			refresh_stages();
			pull_smoothed();
			return;
		}
		if (per_voice_sink) { if (klg_process_voices(gpu, per_voice_sink, buffers, channels, length)) fail("klg_process_voices"); }
		else if (klg_process(gpu, buffers, channels, length, nullptr)) fail("klg_process");
		refresh_stages();
		pull_smoothed();
	}
};

// klang::Note / klang::Synth (mono, klang.h:4292-4466).  A Synth is an Effect in the reference: its own process() post-processes the mix.
struct Synth;
struct Note : NoteBase<Synth>, Generator {
	typedef Synth synth_type;
	virtual void prepare() {}
	virtual void process() override = 0;
	void run_process() { this->process(); }
	signal& out_channel(int) { return out; } int out_channels() const { return 1; }
	void solo_pull() override { gpu::solo_pull(this); }
	void solo_push() override { gpu::solo_push(this); }
	virtual bool process(buffer buffer) {                                    // klang.h:4295-4303: per sample { process(); buffer++ = out; } — a mono note OVERWRITES
		if (attached()) gpu::note_of_a_synth();
		gpu::solo_ensure(this);
		if (!solo->rendered) { gpu::solo_push(this); solo->rendered = true; }
		while (!buffer.finished()) {
			const int m = buffer.remaining() < 1024 ? buffer.remaining() : 1024;
			const float* y = gpu::solo_render(this, 0, m);
			std::memcpy(buffer.cursor(), y, (size_t)m * sizeof(float));
			buffer.advance(m);
		}
		return !finished();
	}
	virtual bool process(buffer* buffers) { return this->process(buffers[0]); }   // klang.h:4304-4306
};
struct Synth : SynthCore<Note> {
	typedef klang::Note Note;
	signal in = { 0.f }, out = { 0.f };
	bool mono_synth() const override { return true; }
	// post processing (klang.h:4438-4439): the Synth's own per-sample process() over the mixed block, identity unless overridden
	virtual void prepare() {}
	virtual void process() { out = in; }
	void run_process() { this->process(); }
	virtual void process(buffer buffer) {
		signal* ins[2] = { &in, &in }; signal* outs[2] = { &out, &out };
		post.build(this, *this, controls, 1, ins, outs, true);
		float* ch[1] = { buffer.cursor() };
		post.run(controls, ch, buffer.remaining());
	}
	// klang.h:4440-4466.  In the reference every sounding note OVERWRITES the block in turn (`buffer++ = out`, klang.h:4299), so what a
	// mono Synth returns is its last sounding note alone: that is `mix = gpu::LastActiveVoice`.  The default here is the SUM of the voices
	// (what Stereo::Synth does) — set `mix` before the first block / event to get the reference's literal behaviour.
	virtual void process(float* buffer, int length, float* parameters = nullptr) {
		if (parameters) for (unsigned c = 0; c < controls.items.size(); c++) controls.items[c].set(parameters[c]);
		float* b[1] = { buffer }; render_voices(b, 1, length);
		klang::buffer mono(buffer, length);
		this->process(mono);
		if (parameters) for (unsigned c = 0; c < controls.items.size(); c++) parameters[c] = controls.items[c].value.value;
	}
};

namespace Stereo {
	struct Synth;
	// Stereo::Note (klang.h:4721-4739): `out` is a STEREO signal; a block is `process(); buffer++ += out;` — out.l is added to the left,
	// out.r to the right channel.  On the device such a note renders two samples per sample (`ret2` of its recorded program).
	// Stereo::Mono::Note (klang.h:4741-4757) keeps a mono `out` that goes to both channels (`left += out; right += out`).
	struct Note : NoteBase<Synth> {
		typedef Synth synth_type;
		Stereo::signal out;
		Note() { out.l.reg_member(); out.r.reg_member(); }                       // (members of `out`: what process() leaves there is what the next sample finds)
		virtual void prepare() {}
		virtual void process() = 0;
		void run_process() { this->process(); }
		virtual klang::signal& out_channel(int c) { return c ? out.r : out.l; }
		virtual int out_channels() const { return 2; }
		void solo_pull() override { gpu::solo_pull(this); }
		void solo_push() override { gpu::solo_push(this); }
		virtual bool process(Stereo::buffer buffer) {                            // klang.h:4727-4734 / 4747-4756: a note ADDS to the buffer
			if (attached()) gpu::note_of_a_synth();
			gpu::solo_ensure(this);
			if (!solo->rendered) { gpu::solo_push(this); solo->rendered = true; }
			while (!buffer.finished()) {
				int m = buffer.left.remaining() < buffer.right.remaining() ? buffer.left.remaining() : buffer.right.remaining();
				if (m > 1024) m = 1024;
				if (m <= 0) break;
				const float* y = gpu::solo_render(this, 0, m);
				const float* yr = out_channels() == 2 ? y + m : y;                 // per-voice block of a stereo note: [2][m]
				float* l = buffer.left.cursor(); float* r = buffer.right.cursor();
				for (int i = 0; i < m; i++) { l[i] += y[i]; r[i] += yr[i]; }
				buffer.left.advance(m); buffer.right.advance(m);
			}
			return !finished();
		}
		virtual bool process(klang::buffer* buffers) { Stereo::buffer b = { buffers[0], buffers[1] }; return this->process(b); }   // klang.h:4735-4738
	};
	namespace Mono {
		struct Note : Stereo::Note, klang::Generator {
			using klang::Generator::out;
			void process() override = 0;
			klang::signal& out_channel(int) override { return out; }
			int out_channels() const override { return 1; }
		};
	}
	struct Synth : SynthCore<Note> {
		typedef Stereo::Note Note;
		struct Mono { typedef Stereo::Mono::Note Note; };
		Stereo::signal in, out;
		virtual void prepare() {}
		virtual void process() { out = in; }                                       // post processing (klang.h:4826-4827)
		void run_process() { this->process(); }
		virtual void process(Stereo::buffer buffer) {
			klang::signal* ins[2] = { &in.l, &in.r }; klang::signal* outs[2] = { &out.l, &out.r };
			post.build(this, *this, controls, 2, ins, outs, true);
			float* ch[2] = { buffer.left.cursor(), buffer.right.cursor() };
			post.run(controls, ch, buffer.left.remaining() < buffer.right.remaining() ? buffer.left.remaining() : buffer.right.remaining());
		}
		virtual void process(float** buffers, int length, float* parameters = nullptr) {   // klang.h:4830-4858
			if (parameters) for (unsigned c = 0; c < controls.items.size(); c++) controls.items[c].set(parameters[c]);
			render_voices(buffers, 2, length);
			klang::buffer left(buffers[0], length), right(buffers[1], length);
			Stereo::buffer both(left, right);
			this->process(both);
			if (parameters) for (unsigned c = 0; c < controls.items.size(); c++) parameters[c] = controls.items[c].value.value;
		}
		void output(float** buffers, int length, float* parameters = nullptr) { process(buffers, length, parameters); }                            // v0.7.2 template name (templates/juce/synth/Source/klang.h:3931)
	};
}

// signals<N> / Matrix (klang.h:1273-1337, 1446-1470): a row of signals and the 4 x 4 feedback matrix of Reverb.k
struct Matrix { float v[4][4]; constexpr float operator()(int r, int c) const { return v[r][c]; } };
template<int N> struct signals {
	signal value[N];
	template<typename... A, std::enable_if_t<sizeof...(A) == N, int> = 0> signals(A&... a) : value{ signal(a)... } {}
	signals() {}
	signal& operator[](int i) { return value[i]; } const signal& operator[](int i) const { return value[i]; }
	signals operator+(const signal& x) const { signals s; for (int i = 0; i < N; i++) s.value[i] = value[i] + x; return s; }
	signals operator>>(const Matrix& m) const {                   // Matrix::operator>> klang.h:1457-1466 (rows = outputs)
		static_assert(N == 4, "Matrix is 4 x 4");
		signals s;
		for (int r = 0; r < 4; r++) s.value[r] = value[0] * m(r, 0) + value[1] * m(r, 1) + value[2] * m(r, 2) + value[3] * m(r, 3);
		return s;
	}
};

namespace gpu {
// `instances` copies of a user effect FX, rendered on the GPU (bank scale: BASELINE config 4).  The constructor builds ONE FX object
// with every primitive / signal member announcing itself, runs prepare() and then process() once in recording mode
// (include/klang_mi355_graph.h, `kind effect`), packs the object's state into the record every instance starts from and creates the bank.
template<class FX> struct EffectBank {
	FX* fx = nullptr; klg_fx* h = nullptr; GraphLayout layout; int instances; int channels = FX::channels;
	explicit EffectBank(int instances_, int max_block = 1024) : instances(instances_) {
		// an effect type tied to a hand-written kernel (KLANG_GPU_BIND_FX: the shipped PingPong.k / Reverb.k) is not recorded
		const int bound = klang_gpu_fx_patch((const FX*)nullptr);
		if (bound >= 0 && !std::getenv("KLANG_MI355_FORCE_GRAPH")) {
			fx = new FX();
			close_log();
			h = klg_fx_create(bound, instances, fs.f, max_block);
			if (!h) { std::fprintf(stderr, "klang-mi355: klg_fx_create: %s\n", klg_last_error()); std::abort(); }
			return;
		}
		Recorder C; C.effect = true; rec = &C;
		C.constructing = true; fx = new FX(); C.constructing = false; rec = nullptr;
		const char* lo = (const char*)fx;
		signal* ins[2]; signal* outs[2];
		if constexpr (FX::channels == 2) { ins[0] = &fx->in.l; ins[1] = &fx->in.r; outs[0] = &fx->out.l; outs[1] = &fx->out.r; }
		else { ins[0] = ins[1] = &fx->in; outs[0] = outs[1] = &fx->out; }
		FX* f = fx;
		record_effect(member_objs(C, lo, lo + sizeof(FX)), fx->controls, channels, ins, outs, [f]() { f->prepare(); }, [f]() { f->run_process(); }, lo, layout, typeid(FX).name());
		std::vector<uint32_t> words((size_t)layout.words, 0u);
		layout.pack(fx, words.data());
		if (std::getenv("KLANG_MI355_DUMP_GRAPH")) { std::fprintf(stderr, "klang-mi355: initial record:"); for (uint32_t w : words) std::fprintf(stderr, " %08x", w); std::fprintf(stderr, "\n"); }
		h = klg_fx_create_graph(layout.program.c_str(), instances, fs.f, max_block, words.data());
		if (!h) { std::fprintf(stderr, "klang-mi355: klg_fx_create_graph: %s\n", klg_last_error()); std::abort(); }
		device_controls = (int)fx->controls.items.size() < (int)klg::KLG_MAX_CTL ? (int)fx->controls.items.size() : (int)klg::KLG_MAX_CTL;      // (record_effect: the controls a program knows)
	}
	~EffectBank() { if (h) klg_fx_destroy(h); delete fx; for (FX* m : mirror) delete m; }
	void set(int instance, int control, float value) {
		if (layout.host_prepare) {                                // prepare() is host code: the dial goes to the instance's host mirror, prepare() runs before the next block
			mirror_of(instance)->controls[control].set(value);
			if (!dirty[(size_t)instance]) { dirty[(size_t)instance] = 1; touched.push_back(instance); }
			if (control >= device_controls) return;                  // (a dial only prepare() reads: the program does not know it)
		}
		if (klg_fx_set_control(h, instance, control, value)) { std::fprintf(stderr, "klang-mi355: %s\n", klg_last_error()); std::abort(); }
	}
	void process(float* io /* [instances][channels][n] */, int n) {
		if (layout.host_prepare) host_prepare();
		if (klg_fx_process(h, io, n)) { std::fprintf(stderr, "klang-mi355: klg_fx_process: %s\n", klg_last_error()); std::abort(); }
		samples += (unsigned long long)n;
	}

	// ---- an effect whose prepare() is HOST code (it asks Controls::changed(): examples/Reverb.k:238-241 — then reseeds rand(), draws tap tables in a
	// loop over a count, compares cached values with !=, calls libm).  It stays host code, as in the reference: every instance has a host MIRROR (an FX
	// object of its own, with its own Controls cache and member caches — what prepare() does depends on which dials moved since ITS last call); before a
	// block, the mirror of every instance whose dials were set takes the instance's record as the device last left it (klg_fx_download_record), its
	// Delays learn where their write cursors stand (samples x inputs per sample: Delay::set() places the read head against it), prepare() runs, and the
	// words it changed are uploaded (klg_fx_upload_words).  process() is recorded once, from the prototype, and reads those members from the record. ----
	std::vector<FX*> mirror; std::vector<char> dirty; std::vector<int> touched;
	unsigned long long samples = 0; int device_controls = 0;
	FX* mirror_of(int k) {
		if (mirror.empty()) { mirror.assign((size_t)instances, nullptr); dirty.assign((size_t)instances, 0); }
		if (!mirror[(size_t)k]) { mirror[(size_t)k] = new FX(); close_log(); }
		return mirror[(size_t)k];
	}
	void host_prepare() {
		if (mirror.empty()) { mirror_of(0); for (int k = 0; k < instances; k++) { dirty[(size_t)k] = 1; touched.push_back(k); } }   // the first block: every instance's prepare() finds its dials changed (klang.h:1914)
		if (touched.empty()) return;
		std::vector<uint32_t> before((size_t)layout.words), after((size_t)layout.words);
		for (int k : touched) {
			FX* m = mirror_of(k);
			if (klg_fx_download_record(h, k, before.data(), before.size() * 4)) { std::fprintf(stderr, "klang-mi355: %s\n", klg_last_error()); std::abort(); }
			layout.unpack(m, before.data(), &m->controls);
			for (size_t j = 0; j < layout.members.size(); j++) if (layout.members[j].kind == klg::graph::N_DELAY)
				reinterpret_cast<Packable*>((char*)m + layout.members[j].offset)->host_cursor(samples * (unsigned long long)layout.delay_inputs[j]);
			layout.pack(m, before.data(), &m->controls);            // (what the mirror holds — not everything a primitive keeps on the device has a host side: only what prepare() CHANGES goes back)
			m->prepare();
			layout.pack(m, after.data(), &m->controls);
			for (int w = 0; w < layout.words;) {
				if (after[(size_t)w] == before[(size_t)w]) { w++; continue; }
				int e = w + 1; while (e < layout.words && after[(size_t)e] != before[(size_t)e]) e++;
				if (klg_fx_upload_words(h, k, w, e - w, after.data() + w)) { std::fprintf(stderr, "klang-mi355: %s\n", klg_last_error()); std::abort(); }
				w = e;
			}
			if (std::getenv("KLANG_MI355_DUMP_GRAPH") && k == 0 && samples == 0) { std::fprintf(stderr, "klang-mi355: prepared record:"); for (uint32_t w : after) std::fprintf(stderr, " %08x", w); std::fprintf(stderr, "\n"); }
			if (std::getenv("KLANG_MI355_DUMP_GRAPH")) { int nchg = 0; for (int w = 0; w < layout.words; w++) nchg += after[(size_t)w] != before[(size_t)w]; std::fprintf(stderr, "klang-mi355: host prepare() of instance %d at sample %llu changed %d of %d words\n", k, samples, nchg, layout.words); }
			dirty[(size_t)k] = 0;
		}
		touched.clear();
	}
};
}

namespace optimised { using namespace klang; using namespace Generators::Fast; using namespace Modifiers; using namespace Filters; using namespace Filters::Biquad; }   // klang.h:6145-6152
namespace basic { using namespace klang; using namespace Generators::Basic; using namespace Modifiers; using namespace Filters; using namespace Filters::Biquad; }       // klang.h:6136-6143
namespace minimal { using namespace klang; }

}  // namespace klang

// `std::abs(x)` of a signal: in the reference the signal converts to float and takes std::abs(float).  A recorded value has no float to
// convert to, so the call is given the signal itself (an exact match beats the float conversion); same value, recordable.
namespace std { inline ::klang::signal abs(const ::klang::signal& x) { return ::klang::abs_of(x); } }
// `abs` in a patch's text names klang's object (klang.h:3072 `#define abs klang::abs`), and what a patch writes as `std::abs(x)` is therefore `std::klang::abs(x)`: the
// original math function (klang.h:56-62).  Both as the reference has them — the macro at the very end of this header.
namespace std { namespace klang {
	inline ::klang::signal abs(const ::klang::signal& x) { return ::klang::abs_of(x); }
	template<class T, std::enable_if_t<std::is_arithmetic_v<T>, int> = 0> inline T abs(T x) { return std::abs(x); }
} }

// ---- plain-`float` C functions applied to signals (examples/Distortion/Functions.k: `float hardclip(float x) {...}`, `hardclip(in * gain) >> out`) ----
// A tracing facade cannot see into `float f(float)`: the signal converts to a float and the recorded world ends there.  A patch of that kind is compiled
// with -DKLANG_GPU_TRACE_FLOAT: from here on — i.e. in the patch's OWN text, which follows this header — the word `float` names klang::signal, the value
// that records what is done to it (arithmetic, comparisons in `if`: one traced run of process() per outcome).  The .k file itself stays as it is; whoever
// includes it puts `#undef float` behind the include (include/klang/bindings.h and the test drivers do).
// What the renaming must never do SILENTLY: change the meaning of `sizeof(float)`, or let a `float*` that is really a signal* go through the C library's
// byte copiers.  Both are compile errors with a message: `sizeof` of the tracing type (or of an array of it) in the patch's text does not compile, and memcpy / memmove /
// memset / memcmp on klang::signal objects are deleted overloads — with or without the switch: a signal here is a value AND its place in a recording, not four bytes.
// (`union { float f; unsigned u; }` does not compile either: the tracing type has constructors.)
namespace klang { namespace gpu {
template<class T> struct traced_sizeof_guard {
	static_assert(!std::is_same<typename std::remove_cv<typename std::remove_all_extents<typename std::remove_reference<T>::type>::type>::type, ::klang::signal>::value,
	              "klang-mi355: sizeof(float) in a patch compiled with -DKLANG_GPU_TRACE_FLOAT would be the size of the TRACING type (a value and its register), not 4: "
	              "say sizeof(std::uint32_t) / 4 where bytes are meant, or move the byte-level code into a function compiled without the switch");
	static constexpr std::size_t zero = 0;
};
} }
void* memcpy(::klang::signal*, const void*, std::size_t) = delete;            // (a signal is a value and its place in a recording: not bytes)
void* memcpy(void*, const ::klang::signal*, std::size_t) = delete;
void* memcpy(::klang::signal*, const ::klang::signal*, std::size_t) = delete;
void* memmove(::klang::signal*, const void*, std::size_t) = delete;
void* memmove(void*, const ::klang::signal*, std::size_t) = delete;
void* memmove(::klang::signal*, const ::klang::signal*, std::size_t) = delete;
void* memset(::klang::signal*, int, std::size_t) = delete;
int memcmp(const ::klang::signal*, const void*, std::size_t) = delete;
int memcmp(const void*, const ::klang::signal*, std::size_t) = delete;
int memcmp(const ::klang::signal*, const ::klang::signal*, std::size_t) = delete;
namespace std { using ::memcpy; using ::memmove; using ::memset; using ::memcmp; }
#define abs klang::abs
#ifdef KLANG_GPU_TRACE_FLOAT
#define float ::klang::signal
#define sizeof(...) (sizeof(__VA_ARGS__) + ::klang::gpu::traced_sizeof_guard<__typeof__(__VA_ARGS__)>::zero)
#endif
