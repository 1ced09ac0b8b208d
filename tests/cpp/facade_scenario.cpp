// tests/cpp/facade_scenario.cpp — drives a .k patch THROUGH THE DSL FAÇADE (include/klang/klang.h): the patch's own
// on()/off() run on the host, blocks are rendered by libklang_mi355.so.  Reads a scenario (tests/scenario_io.py format,
// one synth instance), writes the stereo mix per block + note stages:  int32 'KLGM', N, B, P ; float32 [B][2][N] ; uint8 [B][P]
// and, built with -DDUMP_VOICES=<channels of a note's out>, every voice's own block of the scenario's dump blocks:
// int32 ndump, NC ; float32 [ndump][P][NC][N]
#include <cstdio>
#include <string>
#include <vector>

#ifndef KLANG_TEST_NOTES
#define KLANG_TEST_NOTES 16
#endif
#include PATCH_FILE
#include <klang/bindings.h>
BIND_LINE

struct Ev { int block, type, synth; float a, b; long seed; };

int main(int argc, char** argv) {
	if (argc < 3) { std::fprintf(stderr, "usage: %s scenario out.bin\n", argv[0]); return 1; }
	FILE* f = std::fopen(argv[1], "r");
	if (!f) return 1;
	char tok[64]; int ver; float fsr = 48000.f; int block = 256, blocks = 1, notes = 1; std::vector<Ev> ev; std::vector<int> dump; std::vector<std::pair<int, float>> ctl;
	if (std::fscanf(f, "%63s %d", tok, &ver) != 2) return 1;
	while (std::fscanf(f, "%63s", tok) == 1) {
		std::string t(tok); int k; 
		if (t == "end") break;
		else if (t == "patch") (void)!std::fscanf(f, "%63s", tok);
		else if (t == "fs") (void)!std::fscanf(f, "%f", &fsr);
		else if (t == "block") (void)!std::fscanf(f, "%d", &block);
		else if (t == "blocks") (void)!std::fscanf(f, "%d", &blocks);
		else if (t == "synths") (void)!std::fscanf(f, "%d", &k);
		else if (t == "notes") (void)!std::fscanf(f, "%d", &notes);
		else if (t == "dump") { (void)!std::fscanf(f, "%d", &k); for (int i = 0; i < k; i++) { int d; (void)!std::fscanf(f, "%d", &d); dump.push_back(d); } }
		else if (t == "ctl") { int i; float v; (void)!std::fscanf(f, "%d %f", &i, &v); ctl.push_back({ i, v }); }
		else if (t == "ev") { Ev e; (void)!std::fscanf(f, "%d %d %d %f %f %ld", &e.block, &e.type, &e.synth, &e.a, &e.b, &e.seed); ev.push_back(e); }
	}
	std::fclose(f);
	klang::fs = klang::SampleRate(fsr);
	SYNTH_TYPE synth;
	for (auto& c : ctl) synth.controls[c.first].set(c.second);
	const int N = block, B = blocks, P = (int)synth.notes.count;
	std::vector<float> mix((size_t)B * 2 * N, 0.f);
	std::vector<unsigned char> stages((size_t)B * P);
#ifdef DUMP_VOICES
	std::vector<float> pv((size_t)P * DUMP_VOICES * N), dumps;
	synth.per_voice_sink = pv.data();
#endif
	size_t evi = 0;
	for (int b = 0; b < B; b++) {
		for (; evi < ev.size() && ev[evi].block <= b; evi++) {
			const Ev& e = ev[evi];
			if (e.synth != 0) continue;
			if (e.type == 0) {
				if (e.seed >= 0) klang::random((unsigned)e.seed);
				synth.noteOn((int)e.a, e.b);
				if (std::getenv("KLG_FACADE_DEBUG")) { std::printf("on p=%d:", (int)e.a); for (size_t i = 0; i < synth.words.size() && i < 12; i++) std::printf(" %08x", synth.words[i]); std::printf("\n"); }
			}
			else if (e.type == 1) synth.noteOff((int)e.a, e.b);
			else if (e.type == 2) { synth.controls[(int)e.a].set(e.b); synth.onControl((int)e.a, synth.controls[(int)e.a].value); }
		}
		float* bufs[2] = { &mix[((size_t)b * 2 + 0) * N], &mix[((size_t)b * 2 + 1) * N] };
#ifdef MONO_SYNTH
		synth.process(bufs[0], N);
		for (int i = 0; i < N; i++) bufs[1][i] = bufs[0][i];
#else
		synth.process(bufs, N);
#endif
		for (int p = 0; p < P; p++) stages[(size_t)b * P + p] = (unsigned char)synth.notes[p]->stage;
#ifdef DUMP_VOICES
		for (int d : dump) if (d == b) dumps.insert(dumps.end(), pv.begin(), pv.end());
#endif
	}
	FILE* o = std::fopen(argv[2], "wb");
	const int hdr[4] = { 0x4D474C4B, N, B, P };
	std::fwrite(hdr, 4, 4, o); std::fwrite(mix.data(), 4, mix.size(), o); std::fwrite(stages.data(), 1, stages.size(), o);
#ifdef DUMP_VOICES
	const int vh[2] = { (int)(dumps.size() / pv.size()), DUMP_VOICES };
	std::fwrite(vh, 4, 2, o); std::fwrite(dumps.data(), 4, dumps.size(), o);
#endif
	std::fclose(o);
	return 0;
}
