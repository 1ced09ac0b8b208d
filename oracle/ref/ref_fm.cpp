// oracle/ref/ref_fm.cpp — TEST INFRASTRUCTURE (golden-vector generator).
// Drives the genuine reference header with
//   patch "fm3": the shipped examples/FM.k (included by path, unmodified; 3 operators)
//   patch "fm4": the 4-operator chain BASELINE.json config 5 names — the same note body as FM.k
//                with one more Operator<Sine> in series, written here
//                against the reference API.
#include "prelude.h"
#include <klang.h>
#include "examples/FM.k"

struct FM4 : Synth {
	struct MyNote : Note {
		Operator<Sine> op1, op2, op3, op4;
		ADSR adsr;

		event on(Pitch p, Velocity v) {
			const param fc = p -> Frequency;
			const Frequency fd = fc * controls[0];

			op1.set(fd, 0);
			op1 = { {0,0}, {3,1} };

			op2.set(fd, 0);
			op2 = { {0,1.5}, {3,0.5} };

			op3.set(fd, 0);
			op3 = { {0,1}, {2,0.25} };

			op4.set(fc, 0);

			adsr(controls[4], 0.1, 1, 1);
		}

		event off(Velocity v) {
			adsr.release();
		}

		void process() {
			const param I1 = controls[1];
			const param I2 = controls[2];
			const param I3 = controls[3];

			op1 * I1 >> op2 * I2 >> op3 * I3 >> op4 >> out;

			out *= adsr++ * 0.1f;
			if (adsr.finished())
				stop();
		}
	};

	FM4() {
		controls = {
			Dial("Mod Freq", 0.001, 10.0, 1.0),
			Dial("Mod Index 1", 0.000, 10.0, 0.37),
			Dial("Mod Index 2", 0.000, 10.0, 0.37),
			Dial("Mod Index 3", 0.000, 10.0, 0.37),
			Dial("Attack", 0.000, 1.0, 0.5),
		};
		notes.add<MyNote>(32);
	}
};

#define REF_WITH_KLANG
#include "ref_common.h"

int main(int argc, char** argv) {
	if (argc < 3) { fprintf(stderr, "usage: %s scenario out.bin\n", argv[0]); return 1; }
	RefScenario s;
	if (!ref_load(argv[1], s)) return 1;
	if (s.patch == "fm3") return run_synth<FM, 0>(s, argv[2]);
	if (s.patch == "fm4") return run_synth<FM4, 0>(s, argv[2]);
	fprintf(stderr, "unknown patch %s\n", s.patch.c_str());
	return 1;
}
