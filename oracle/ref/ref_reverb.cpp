// oracle/ref/ref_reverb.cpp — TEST INFRASTRUCTURE (golden-vector generator).
// Drives the genuine reference header with the shipped examples/Reverb.k (included by path, unmodified).
#include "prelude.h"
#include <klang.h>
#include "examples/Reverb.k"

#define REF_WITH_KLANG
#include "ref_common.h"

int main(int argc, char** argv) {
	if (argc < 3) { fprintf(stderr, "usage: %s scenario out.bin\n", argv[0]); return 1; }
	RefScenario s;
	if (!ref_load(argv[1], s)) return 1;
	if (s.patch == "reverb") return run_effect<Reverb>(s, argv[2]);
	fprintf(stderr, "unknown patch %s\n", s.patch.c_str());
	return 1;
}
