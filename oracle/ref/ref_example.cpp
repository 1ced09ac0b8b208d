// oracle/ref/ref_example.cpp — TEST INFRASTRUCTURE (golden-vector generator).
// Drives the genuine reference header with ONE of the reference's shipped example synths, included by path, unmodified:
//   -DEXAMPLE_FILE='"examples/Subtractive/Filter.k"' -DEXAMPLE_TYPE=Filter -DEXAMPLE_NAME='"ex_filter"'
// (one binary per example: the examples all live in the global namespace).  These patches have NO hand-written kernel in
// the library: they exercise the recorded-graph path (include/klang_mi355_graph.h) against the reference.
#include "prelude.h"
#include <klang.h>
#include EXAMPLE_FILE

#define REF_WITH_KLANG
#include "ref_common.h"

int main(int argc, char** argv) {
	if (argc < 3) { fprintf(stderr, "usage: %s scenario out.bin\n", argv[0]); return 1; }
	RefScenario s;
	if (!ref_load(argv[1], s)) return 1;
#ifndef EXAMPLE_NOTE_KIND
#define EXAMPLE_NOTE_KIND 0        // 2: the patch's notes are Stereo::Notes with a stereo `out` (ref_common.h)
#endif
	if (s.patch == EXAMPLE_NAME) return run_synth<EXAMPLE_TYPE, EXAMPLE_NOTE_KIND>(s, argv[2]);
	fprintf(stderr, "unknown patch %s\n", s.patch.c_str());
	return 1;
}
