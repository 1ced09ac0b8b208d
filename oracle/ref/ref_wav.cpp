// oracle/ref/ref_wav.cpp — TEST INFRASTRUCTURE: decodes a WAV file with the GENUINE reference header's File::WAV (klang.h:5991-6099)
// and prints the samples it hands to a Sample, one hex float per line after the count.  Used by oracle/gen_golden_wav.py only.
#include "prelude.h"
#include <klang.h>
#include <cstdio>
int main(int argc, char** argv) {
	if (argc != 2) return 2;
	klang::File::WAV wav;
	if (!wav.load(argv[1])) { std::fprintf(stderr, "load failed\n"); return 1; }
	klang::variable::buffer buffer;
	if (!(wav >> buffer)) { std::fprintf(stderr, "decode failed\n"); return 1; }
	std::printf("%d\n", (int)buffer.size);
	const float* p = buffer.data();
	for (int i = 0; i < (int)buffer.size; i++) { unsigned u; std::memcpy(&u, &p[i], 4); std::printf("%08x\n", u); }
	return 0;
}
