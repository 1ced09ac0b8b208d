// tests/hosts/synth_host.cpp — a plugin host for ONE synth object: the block callback of the reference's JUCE synth template
// (templates/juce/synth/Source/PluginProcessor.cpp:153-182) minus JUCE, against the v0.7.8 entry points: the processor owns the synth,
// clears the output block (:165-166), passes the block's note events on (noteOn / noteOff), and calls the synth's REAL block entry —
// Synth::process(float*, int) for a mono klang::Synth (klang.h:4440-4466: every sounding note OVERWRITES the block in note order, then the
// Synth's own post-processing runs), Stereo::Synth::process(float**, int) (4830-4858: notes accumulate) otherwise.
// Compiled against the genuine reference header (oracle/_ref/ref_host_synth_*: golden vectors) and against the façade
// (facade_host_synth_*: GPU).  Writes int32 'KLGM', N, B, P ; float32 [B][2][N] ; uint8 stages [B][P] (the layout of facade_scenario.cpp).
#include <cstdio>
#include <string>
#include <vector>

#include PATCH_FILE

struct Ev { int block, type, synth; float a, b; long seed; };

struct KlangSynthAudioProcessor {
	HOST_SYNTH_TYPE synth;
	void processBlock(const std::vector<Ev>& midiMessages, float** channels, int numSamples) {
		for (int c = 0; c < 2; c++) for (int i = 0; i < numSamples; i++) channels[c][i] = 0.f;       // buffer.clear(i, 0, numSamples)
		for (const Ev& e : midiMessages) {
			if (e.type == 0) { if (e.seed >= 0) klang::random((unsigned)e.seed); synth.noteOn((int)e.a, e.b); }
			else if (e.type == 1) synth.noteOff((int)e.a, e.b);
			else if (e.type == 2) { synth.controls[(int)e.a].set(e.b); synth.onControl((int)e.a, synth.controls[(int)e.a].value); }
		}
#ifdef KLANG_REF_BLOCK_SESSION_NOTE
		KLANG_REF_BLOCK_SESSION_NOTE(numSamples);
#endif
#ifdef HOST_MONO
		synth.klang::Synth::process(channels[0], numSamples);             // (qualified, like the effect template: a Synth's own process() hides the overloads)
		for (int i = 0; i < numSamples; i++) channels[1][i] = channels[0][i];
#else
		synth.klang::Stereo::Synth::process(channels, numSamples);
#endif
	}
};

int main(int argc, char** argv) {
	if (argc < 3) { std::fprintf(stderr, "usage: %s scenario out.bin\n", argv[0]); return 1; }
	FILE* f = std::fopen(argv[1], "r");
	if (!f) return 1;
	char tok[64]; int ver; float fsr = 48000.f; int block = 256, blocks = 1; std::vector<Ev> ev; std::vector<std::pair<int, float>> ctl;
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
		else if (t == "ctl") { int i; float v; (void)!std::fscanf(f, "%d %f", &i, &v); ctl.push_back({ i, v }); }
		else if (t == "ev") { Ev e; (void)!std::fscanf(f, "%d %d %d %f %f %ld", &e.block, &e.type, &e.synth, &e.a, &e.b, &e.seed); ev.push_back(e); }
	}
	std::fclose(f);
	klang::fs = klang::SampleRate(fsr);
	static KlangSynthAudioProcessor processor;
	for (auto& c : ctl) processor.synth.controls[c.first].set(c.second);
	const int N = block, B = blocks, P = (int)processor.synth.notes.count;
	std::vector<float> mix((size_t)B * 2 * N, 0.f);
	std::vector<unsigned char> stages((size_t)B * P);
	size_t evi = 0;
	for (int b = 0; b < B; b++) {
		std::vector<Ev> now;
		for (; evi < ev.size() && ev[evi].block <= b; evi++) if (ev[evi].synth == 0) now.push_back(ev[evi]);
		float* channels[2] = { &mix[((size_t)b * 2 + 0) * N], &mix[((size_t)b * 2 + 1) * N] };
		processor.processBlock(now, channels, N);
		for (int p = 0; p < P; p++) stages[(size_t)b * P + p] = (unsigned char)processor.synth.notes[p]->stage;
	}
	FILE* o = std::fopen(argv[2], "wb");
	const int hdr[4] = { 0x4D474C4B, N, B, P };
	std::fwrite(hdr, 4, 4, o); std::fwrite(mix.data(), 4, mix.size(), o); std::fwrite(stages.data(), 1, stages.size(), o);
	std::fclose(o);
	return 0;
}
