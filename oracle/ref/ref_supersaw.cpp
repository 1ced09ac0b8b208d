// oracle/ref/ref_supersaw.cpp — TEST INFRASTRUCTURE (golden-vector generator).
// Drives the genuine reference header with the shipped examples/SuperSaw.k (included by path, unmodified).
#include "prelude.h"
#include <klang.h>
#include "examples/SuperSaw.k"

#define REF_WITH_KLANG
#include "ref_common.h"

int main(int argc, char** argv) {
	if (argc < 3) { fprintf(stderr, "usage: %s scenario out.bin\n", argv[0]); return 1; }
	RefScenario s;
	if (!ref_load(argv[1], s)) return 1;
	if (s.patch == "supersaw") return run_synth<SuperSaw, 0>(s, argv[2]);
	fprintf(stderr, "unknown patch %s\n", s.patch.c_str());
	return 1;
}
