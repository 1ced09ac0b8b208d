// oracle/ref/ref_sine.cpp — TEST INFRASTRUCTURE (golden-vector generator).
// BASELINE.json config 1: a single-oscillator Note ("sine" = Generators::Fast::Sine,
// "bsine" = Generators::Basic::Sine) inside a mono klang::Synth, written against the reference API.
#include "prelude.h"
#include <klang.h>

static int g_notes = 1;

template<class OSC>
struct SineSynth : klang::Synth {
	struct MyNote : klang::Note {
		OSC osc;
		void on(klang::Pitch pitch, klang::Amplitude velocity) {
			const klang::param f = pitch -> Frequency;
			osc(f, 0);
		}
		void off(klang::Amplitude velocity) { stop(); }
		void process() { osc >> out; }
	};
	SineSynth() { notes.template add<MyNote>(g_notes); }
};

#define REF_WITH_KLANG
#include "ref_common.h"

int main(int argc, char** argv) {
	if (argc < 3) { fprintf(stderr, "usage: %s scenario out.bin\n", argv[0]); return 1; }
	RefScenario s;
	if (!ref_load(argv[1], s)) return 1;
	g_notes = s.notes;
	if (s.patch == "sine") return run_synth<SineSynth<klang::Generators::Fast::Sine>, 0>(s, argv[2]);
	if (s.patch == "bsine") return run_synth<SineSynth<klang::Generators::Basic::Sine>, 0>(s, argv[2]);
	fprintf(stderr, "unknown patch %s\n", s.patch.c_str());
	return 1;
}
