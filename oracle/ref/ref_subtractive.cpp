// oracle/ref/ref_subtractive.cpp — TEST INFRASTRUCTURE (golden-vector generator).
// Drives the genuine reference header with
//   patch "sub2b": the shipped templates/juce/synth/Source/subtractive.k (included by path, unmodified)
//   patch "sub2a": the config-2 patch of SURVEY.md §8(d) (Saw >> LPF >> ADSR, static cutoff),
//                  written here against the reference API.
#include "prelude.h"
#include <klang.h>
#include "templates/juce/synth/Source/subtractive.k"   // -I/root/reference ; brings `using namespace klang::optimised`

static int g_notes = 128;

struct Sub2a : Stereo::Synth {
	struct MyNote : Mono::Note {
		Saw osc;
		LPF lpf;
		ADSR adsr;

		event on(Pitch pitch, Amplitude velocity) {
			const param f = pitch -> Frequency;
			osc(f, 0);
			lpf.reset();
			lpf.set(4 * f, 2);
			adsr(0.01, 0.1, 0.7, 0.25);
		}

		event off(Amplitude velocity) {
			adsr.release();
		}

		void process() {
			osc >> lpf >> out;
			out *= adsr++;
			if (adsr.finished())
				stop();
		}
	};

	Sub2a() { notes.add<MyNote>(g_notes); }
};

#define REF_WITH_KLANG
#include "ref_common.h"

int main(int argc, char** argv) {
	if (argc < 3) { fprintf(stderr, "usage: %s scenario out.bin\n", argv[0]); return 1; }
	RefScenario s;
	if (!ref_load(argv[1], s)) return 1;
	if (s.patch == "sub2a") { g_notes = s.notes; return run_synth<Sub2a, 1>(s, argv[2]); }
	if (s.patch == "sub2b") return run_synth<Subtractive, 1>(s, argv[2]);
	fprintf(stderr, "unknown patch %s\n", s.patch.c_str());
	return 1;
}
