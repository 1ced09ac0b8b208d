// tests/hosts/note_host.cpp — the reference's "Usage in a C++ project" (README.md:133-142): ONE note object driven by the host itself,
//     note.start(pitch, velocity);  note.release(velocity);  klang::buffer buffer(pfBuffer, numSamples);  if (!note.klang::Note::process(buffer))      // (qualified, like the effect template: the note's own process() hides the overload) note.stop();
// i.e. Note::process(buffer) -> bool (klang.h:4295-4303) with no Synth around it.  Compiled against the genuine reference header
// (oracle/_ref/ref_host_note_*: golden vectors) and against the façade (tests/cpp/_bin/facade_host_note_*: the block is rendered by
// libklang_mi355.so).  Reads a synth scenario (one note: events of synth 0), writes float32 [B][N] + uint8 finished[B].
#include <cstdio>
#include <string>
#include <vector>

#include PATCH_FILE

struct Ev { int block, type, synth; float a, b; long seed; };

int main(int argc, char** argv) {
	if (argc < 3) { std::fprintf(stderr, "usage: %s scenario out.bin\n", argv[0]); return 1; }
	FILE* f = std::fopen(argv[1], "r");
	if (!f) return 1;
	char tok[64]; int ver; float fsr = 48000.f; int block = 256, blocks = 1; std::vector<Ev> ev;
	if (std::fscanf(f, "%63s %d", tok, &ver) != 2) return 1;
	while (std::fscanf(f, "%63s", tok) == 1) {
		std::string t(tok); int k;
		if (t == "end") break;
		else if (t == "patch") (void)!std::fscanf(f, "%63s", tok);
		else if (t == "fs") (void)!std::fscanf(f, "%f", &fsr);
		else if (t == "block") (void)!std::fscanf(f, "%d", &block);
		else if (t == "blocks") (void)!std::fscanf(f, "%d", &blocks);
		else if (t == "synths" || t == "notes") (void)!std::fscanf(f, "%d", &k);
		else if (t == "dump") { (void)!std::fscanf(f, "%d", &k); for (int i = 0; i < k; i++) { int d; (void)!std::fscanf(f, "%d", &d); } }
		else if (t == "ctl") { int i; float v; (void)!std::fscanf(f, "%d %f", &i, &v); }
		else if (t == "ev") { Ev e; (void)!std::fscanf(f, "%d %d %d %f %f %ld", &e.block, &e.type, &e.synth, &e.a, &e.b, &e.seed); ev.push_back(e); }
	}
	std::fclose(f);
	klang::fs = klang::SampleRate(fsr);
	HOST_NOTE_TYPE note;
	const int N = block;
	std::vector<float> pfBuffer((size_t)N);
	std::vector<unsigned char> finished;
	FILE* out = std::fopen(argv[2], "wb");
	if (!out) return 1;
	size_t evi = 0;
	for (int b = 0; b < blocks; b++) {
		for (; evi < ev.size() && ev[evi].block <= b; evi++) {
			const Ev& e = ev[evi];
			if (e.seed >= 0) klang::random((unsigned)e.seed);
			if (e.type == 0) note.start(e.a, e.b);                  // Note On
			else if (e.type == 1) note.release(e.b);                 // Note Off
		}
		for (int i = 0; i < N; i++) pfBuffer[(size_t)i] = 0.25f;   // a mono note OVERWRITES whatever the host left in the buffer
		if (!note.finished()) {
#ifdef KLANG_REF_BLOCK_SESSION_NOTE
			KLANG_REF_BLOCK_SESSION_NOTE(N);
#endif
			klang::buffer buffer(pfBuffer.data(), N);
			if (!note.klang::Note::process(buffer))      // (qualified, like the effect template: the note's own process() hides the overload)
				note.stop();
		}
		std::fwrite(pfBuffer.data(), 4, (size_t)N, out);
		finished.push_back(note.finished() ? 1 : 0);
	}
	std::fwrite(finished.data(), 1, finished.size(), out);
	std::fclose(out);
	return 0;
}
