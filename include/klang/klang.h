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
// A Note type is tied to its kernel with KLANG_GPU_BIND (klang/bindings.h holds the bindings of the shipped patches).
// Effects (Stereo::Effect patches) are reached through klg_fx_* directly; their DSL façade is future work.
//
// Reference interface citations (file:line) are into nashaudio/klang's klang.h v0.7.8.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <string>
#include <type_traits>
#include <vector>

#include "../klang_mi355.h"
#include "../klang_mi355_records.h"
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
};
constexpr constant pi = { 3.1415926535897932384626433832795 };
constexpr constant ln2 = { 0.6931471805599453094172321214581 };
constexpr constant root2 = { 1.4142135623730950488016887242097 };

template<typename T> inline T random(const T mn, const T mx) { return std::rand() * ((mx - mn) / (T)RAND_MAX) + mn; }   // klang.h:236
inline void random(const unsigned seed) { std::srand(seed); klg_random_seed(seed); }                                     // klang.h:239

// ---- signal / relative / param (klang.h:1062-1200, 1357-1371) ----
struct relative;
struct signal {
	float value;
	signal(constant c) : value(c.f) {}
	signal(const float v = 0.f) : value(v) {}
	signal(const double v) : value((float)v) {}
	signal(const int v) : value((float)v) {}
	const signal& operator<<(const signal& in) { value = in.value; return *this; }
	signal& operator>>(signal& dst) const { dst.value = value; return dst; }
	signal& operator+=(const signal& x) { value += x.value; return *this; }
	signal& operator-=(const signal& x) { value -= x.value; return *this; }
	signal& operator*=(const signal& x) { value *= x.value; return *this; }
	signal& operator/=(const signal& x) { value /= x.value; return *this; }
#define KLANG_SIGNAL_OPS(T) \
	signal& operator+=(T x) { value += (float)x; return *this; } signal& operator-=(T x) { value -= (float)x; return *this; } \
	signal& operator*=(T x) { value *= (float)x; return *this; } signal& operator/=(T x) { value /= (float)x; return *this; } \
	signal operator+(T x) const { return value + (float)x; } signal operator-(T x) const { return value - (float)x; } \
	signal operator*(T x) const { return value * (float)x; } signal operator/(T x) const { return value / (float)x; }
	KLANG_SIGNAL_OPS(float) KLANG_SIGNAL_OPS(double) KLANG_SIGNAL_OPS(int)
#undef KLANG_SIGNAL_OPS
	operator const float() const { return value; }
	operator float&() { return value; }
	relative operator+() const;
};
struct relative : signal {};
inline relative signal::operator+() const { relative r; r.value = value; return r; }
inline signal& operator>>(float in, signal& dst) { dst.value = in; return dst; }

struct Control;
struct param : signal {
	param(constant c) : signal(c.f) {}
	param(const float v = 0.f) : signal(v) {}
	param(const signal& s) : signal(s) {}
	param(signal& s) : signal(s) {}
	param(Control& c);
};

// ---- Control / Controls / Presets (klang.h:1654-1981; UI fields omitted) ----
struct Control {
	std::string name; float min = 0.f, max = 1.f, initial = 0.f;
	signal value, smoothed;
	operator signal&() { return value; }
	operator param() const { return param(value); }
	operator float() const { return value.value; }
	signal smooth() { smoothed = smoothed.value * 0.999f + (1.f - 0.999f) * value.value; return smoothed; }   // klang.h:1715
	Control& set(float x) { value = (x < min) ? min : (max < x) ? max : x; return *this; }                    // klang.h:1725
};
inline param::param(Control& c) : signal(c.value) {}
inline Control Dial(const char* name, float mn = 0.f, float mx = 1.f, float initial = 0.f) { Control c; c.name = name; c.min = mn; c.max = mx; c.initial = initial; c.value = initial; return c; }
struct Controls {
	std::vector<Control> items; float cache[128] = { 0 };
	void operator=(std::initializer_list<Control> l) { items.assign(l.begin(), l.end()); }
	Control& operator[](int i) { return items[(size_t)i]; }
	unsigned size() const { return (unsigned)items.size(); }
	bool changed() { bool c = false; for (size_t i = 0; i < items.size(); i++) if (items[i].value.value != cache[i]) { cache[i] = items[i].value.value; c = true; } return c; }   // klang.h:1914
};
struct Preset { std::string name; std::vector<float> values; Preset(const char* n, std::initializer_list<double> v) : name(n) { for (double x : v) values.push_back((float)x); } };
struct Presets { std::vector<Preset> items; void operator=(std::initializer_list<Preset> l) { items.assign(l.begin(), l.end()); } };

// ---- units (klang.h:1512-1652) ----
struct Conversion : signal { using signal::signal; };
struct Frequency : param { using param::param; Frequency(float f = 1000.f) : param(f) {} };
struct Pitch : param {
	using param::param;
	static inline thread_local Conversion Frequency;
	const Pitch* operator->() { Frequency = klg::host::pitch_to_frequency(value); return this; }             // klang.h:1568-1571
};
struct Amplitude : param { using param::param; Amplitude(float a = 1.f) : param(a) {} };
typedef Amplitude Velocity;

struct SampleRate {
	float f; int i; double d; float inv, w, nyquist;
	SampleRate(float sr) : f(sr), i(int(sr + 0.001f)), d((double)sr), inv(1.f / sr), w(2.0f * pi * inv), nyquist(sr / 2.f) {}
	operator float() { return f; }
};
inline SampleRate fs(44100);       // ONE definition per program (the reference's is `static` per translation unit, F7)
inline klg::host::Fs host_fs() { return klg::host::Fs(fs.f); }

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

// ---- oscillators: host halves only ----
struct Oscillator : Generator {
	Frequency frequency = 1000.f;
	using Generator::set;
};
namespace Generators {
namespace Basic {
	struct Sine : Oscillator {
		klg::host::BOscH h;
		using Oscillator::set;
		void set(param f) override { h.frequency = f; h.increment = f * 2.f * pi.f / fs.f; }
		void set(param f, param phase) override { h.set(f, phase, host_fs()); }
		void set(relative phase) override { h.offset = phase.value * (2 * pi); }
		void process() override { device_only("Basic::Sine::process()"); }
	};
}
namespace Fast {
	struct Sine : Oscillator {
		klg::host::FSineH h;
		using Oscillator::set;
		void set(param f) override { if (f != h.frequency) { h.frequency = f; h.inc = klg::host::fast_increment(f, host_fs()); } }
		void set(param f, param phase) override { h.set(f, phase, host_fs()); }
		void set(param f, relative phase) override { set(f); set(phase); }
		void set(relative) override { device_only("Fast::Sine::set(relative) [phase modulation]"); }
		void process() override { device_only("Fast::Sine::process()"); }
	};
	struct Osm : Oscillator {
		klg::host::OsmH h; int waveform;             // 0 = saw family, 1 = pulse family
		Osm(int wf, float duty) : h(duty), waveform(wf) {}
		using Oscillator::set;
		void set(param f) override { if (h.frequency != f) { h.refresh(f, host_fs()); h.init(); } }
		void set(param f, param phase) override { h.set(f, phase, host_fs()); }
		void set(param f, param phase, param duty) override { h.set(f, phase, duty, host_fs()); }
		void process() override { device_only("Fast::Osm::process()"); }
	};
	struct Saw : Osm { Saw() : Osm(0, 0.f) {} };
	struct Triangle : Osm { Triangle() : Osm(0, 1.f) {} };
	struct Square : Osm { Square() : Osm(1, 1.0f) {} };
	struct Pulse : Osm { Pulse() : Osm(1, 0.5f) {} };
}
}

// ---- filters ----
namespace Filters { namespace Biquad {
	struct LPF : Modifier {
		klg::host::BiquadLpfH h;
		void reset() { h.reset(); }
		using Modifier::set;
		void set(param f) override { h.set(f, klg::host::ROOT2_INV, host_fs()); }
		void set(param f, param Q) override { h.set(f, Q, host_fs()); }
		void process() override { device_only("Biquad::LPF::process()"); }
	};
} }

// ---- Envelope / ADSR (klang.h:3722-4137) ----
struct Envelope : Generator {
	struct Point { float x, y; Point() : x(0), y(0) {} template<class A, class B> Point(A a, B b) : x(float(a)), y(float(b)) {} };
	enum Stage { Sustain, Release, Off };
	klg::host::EnvH h;
	Envelope() { const float one[2] = { 0.f, 1.f }; h.set_points(1, one, host_fs()); }
	Envelope(std::initializer_list<Point> p) { assign(p); }
	Envelope& operator=(std::initializer_list<Point> p) { assign(p); return *this; }
	void assign(std::initializer_list<Point> p) {
		float xy[8]; int n = 0;
		for (const Point& q : p) if (n < 4) { xy[2 * n] = q.x; xy[2 * n + 1] = q.y; n++; }
		h.set_points(n, xy, host_fs());
	}
	virtual void release(float time, float level = 0.f) { h.stage = klg::ENV_RELEASE; h.set_target(time, level, 0.f, host_fs()); }   // klang.h:3961-3966
	bool finished() const { return h.stage == klg::ENV_OFF; }
	signal& operator++(int) { this->process(); return out; }
	void process() override { device_only("Envelope::process()"); }
};
struct ADSR : Envelope {
	klg::host::AdsrH a;
	ADSR() { set(0.5, 0.5, 1, 0.5); }
	using Envelope::set;
	void set(param attack, param decay, param sustain, param release) override { a.set(attack, decay, sustain, release, host_fs()); h = a.env; }
	void release(float time = 0.f, float level = 0.f) override { Envelope::release(time ? time : a.R, level); }
};

// ---- FM operator (klang.h:4140-4180) ----
template<class OSC> struct Operator : OSC, Input {
	Envelope env; Amplitude amp = 1.f;
	Operator& operator()(param f) { OSC::set(f); return *this; }
	Operator& operator=(std::initializer_list<Envelope::Point> p) { env = p; return *this; }
	Operator& operator*(signal a) { amp = a; return *this; }
	Operator& operator>>(Operator& carrier) { carrier << *this; return carrier; }
	void process() override { device_only("Operator::process()"); }
};

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

struct NoteBinding { int patch; void (*pack)(const void*, uint32_t*); void (*unpack)(void*, const uint32_t*); };

// ---- Controller / Plugin / Effect / NoteBase (klang.h:4182-4292) ----
struct Controller {
protected:
	virtual event control(int, float) {}
	virtual event preset(int) {}
public:
	virtual ~Controller() {}
	virtual void onControl(int index, float value) { control(index, value); }
	virtual void onPreset(int index) { preset(index); }
};
struct Plugin : Controller { Controls controls; Presets presets; };

template<class SYNTH> class NoteBase : public Controller {
	SYNTH* synth = nullptr;
protected:
	virtual event on(Pitch, Velocity) {}
	virtual event off(Velocity = 0) { stage = Off; }
public:
	struct ControlsRef { Controls* c = nullptr; Control& operator[](int i) { return (*c)[i]; } unsigned size() { return c ? c->size() : 0; } } controls;
	Pitch pitch; Velocity velocity;
	enum Stage { Onset, Sustain, Release, Off } stage = Off;
	void attach(SYNTH* s) { synth = s; controls.c = &s->controls; }
	virtual void start(Pitch p, Velocity v) { stage = Onset; pitch = p; velocity = v; on(pitch, velocity); stage = Sustain; }    // klang.h:4257-4263
	virtual bool release(Velocity v = 0) { if (stage == Off) return true; if (stage != Release) { stage = Release; off(v); } return stage == Off; }
	virtual bool stop(Velocity = 0) { stage = Off; return true; }
	bool finished() const { return stage == Off; }
};

// =================================================================================================
// Synth: host voice allocation + event dispatch; blocks rendered by libklang_mi355.so
// =================================================================================================
template<class NOTEBASE> struct SynthCore : Plugin {
	struct Slot { NOTEBASE* note = nullptr; NoteBinding b = { -1, nullptr, nullptr }; };
	struct NotesT {
		SynthCore* owner; std::vector<Slot> items; unsigned noteOns = 0; unsigned noteStart[128] = { 0 };
		unsigned count = 0;
		template<class T> void add(int n) {
			for (int i = 0; i < n && items.size() < 128; i++) {
				T* t = new T(); t->attach(static_cast<typename T::synth_type*>(owner));
				Slot s; s.note = t;
				s.b.patch = klang_gpu_patch((const T*)t);
				s.b.pack = [](const void* p, uint32_t* w) { klang_gpu_pack((const T*)p, w); };
				s.b.unpack = [](void* p, const uint32_t* w) { klang_gpu_unpack((T*)p, w); };
				items.push_back(s); count = (unsigned)items.size();
			}
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
		~NotesT() { for (auto& s : items) delete s.note; }
	} notes;
	klg_synth* gpu = nullptr;
	std::vector<uint32_t> words;
	std::vector<uint8_t> stages;

	SynthCore() { notes.owner = this; }
	~SynthCore() { if (gpu) klg_synth_destroy(gpu); }

	void fail(const char* what) { std::fprintf(stderr, "klang-mi355: %s: %s\n", what, klg_last_error()); std::abort(); }
	void ensure_gpu() {
		if (gpu) return;
		if (!notes.count) { std::fprintf(stderr, "klang-mi355: Synth has no notes (call notes.add<T>(n))\n"); std::abort(); }
		const int patch = notes.items[0].b.patch;
		if (patch < 0) { std::fprintf(stderr, "klang-mi355: no GPU kernel is bound to this Note type: use KLANG_GPU_BIND (klang/bindings.h)\n"); std::abort(); }
		gpu = klg_synth_create(patch, 1, (int)notes.count, fs.f, 1024);
		if (!gpu) fail("klg_synth_create");
		words.resize(klg_synth_state_bytes(gpu) / 4);
		stages.resize(notes.count);
		sync_controls();
	}
	void sync_controls() { for (unsigned c = 0; c < controls.size() && (int)c < klg_synth_controls(gpu); c++) klg_set_control(gpu, 0, (int)c, controls[(int)c].value.value); }
	// host mirror <- lane ; run the event ; lane <- host mirror
	template<class F> void with_voice(int n, F&& event_code) {
		ensure_gpu();
		Slot& s = notes.items[(size_t)n];
		if (klg_voice_download(gpu, n, words.data(), words.size() * 4)) fail("klg_voice_download");
		if ((words[0] & 3u) != (uint32_t)klg::ST_OFF || s.note->stage != NOTEBASE::Off) s.b.unpack(s.note, words.data());
		event_code(s.note);
		s.b.pack(s.note, words.data());
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
	event onControl(int index, float value) override { control(index, value); if (gpu) sync_controls(); }
	void refresh_stages() {
		if (klg_voice_stages(gpu, stages.data(), (int)notes.count)) fail("klg_voice_stages");
		for (unsigned n = 0; n < notes.count; n++) if (stages[n] == klg::ST_OFF) notes[(int)n]->stage = NOTEBASE::Off;    // `if (!note->process(..)) note->stop()`
	}
	void render(float* const* buffers, int channels, int length, float* parameters) {
		ensure_gpu();
		if (parameters) for (unsigned c = 0; c < controls.size(); c++) controls[(int)c].set(parameters[c]);
		sync_controls();
		if (klg_process(gpu, buffers, channels, length, nullptr)) fail("klg_process");
		refresh_stages();
		if (parameters) for (unsigned c = 0; c < controls.size(); c++) parameters[c] = controls[(int)c].value.value;
	}
};

struct Synth;
struct Note : NoteBase<Synth>, Generator { typedef Synth synth_type; virtual void process() override = 0; };
struct Synth : SynthCore<Note> {
	typedef klang::Note Note;
	virtual void process(float* buffer, int length, float* parameters = nullptr) { float* b[1] = { buffer }; render(b, 1, length, parameters); }   // klang.h:4440-4466 (voices are SUMMED)
};

namespace Stereo {
	struct Synth;
	struct Note : NoteBase<Synth>, klang::Generator { typedef Synth synth_type; virtual void process() override = 0; };
	namespace Mono { typedef Stereo::Note Note; }
	struct Synth : SynthCore<Note> {
		typedef Stereo::Note Note;
		struct Mono { typedef Stereo::Note Note; };
		virtual void process(float** buffers, int length, float* parameters = nullptr) { render(buffers, 2, length, parameters); }                 // klang.h:4830-4858
		void output(float** buffers, int length, float* parameters = nullptr) { process(buffers, length, parameters); }                            // v0.7.2 template name
	};
}

namespace optimised { using namespace klang; using namespace Generators::Fast; using namespace Filters::Biquad; }
namespace basic { using namespace klang; using namespace Generators::Basic; using namespace Filters::Biquad; }
namespace minimal { using namespace klang; }

}  // namespace klang
