// oracle/ref/ref_fxexample.cpp — TEST INFRASTRUCTURE (golden-vector generator).
// Drives the genuine reference header with ONE of the reference's shipped example EFFECTS, included by path, unmodified:
//   -DEXAMPLE_FILE='"examples/Delay/Echo.k"' -DEXAMPLE_TYPE=::Echo -DEXAMPLE_CH=1 -DEXAMPLE_NAME='"fx_echo"'
// These effects have no hand-written kernel: they pin the recorded graph-effect path (klang::gpu::EffectBank).
#include "prelude.h"
#include <klang.h>
#include EXAMPLE_FILE

#define REF_WITH_KLANG
#include "ref_common.h"

int main(int argc, char** argv) {
	if (argc < 3) { fprintf(stderr, "usage: %s scenario out.bin\n", argv[0]); return 1; }
	RefScenario s;
	if (!ref_load(argv[1], s)) return 1;
	if (s.patch == EXAMPLE_NAME) return run_effect_example<EXAMPLE_TYPE, EXAMPLE_CH>(s, argv[2]);
	fprintf(stderr, "unknown patch %s\n", s.patch.c_str());
	return 1;
}
