// tests/hosts/effect_host.cpp — a plugin host for ONE effect object, with the block callback of the reference's JUCE effect template
// (templates/juce/effect/Source/PluginProcessor.cpp:153-178) minus JUCE: the processor owns the effect as a plain member
// (PluginProcessor.h: `PingPong pingpong;`), publishes one parameter per control (constructor, :24-46), and per block wraps the channel
// pointers in klang::buffer / klang::stereo::buffer, copies the parameters into the controls and calls
// `pingpong.klang::Stereo::Effect::process(buffers)`.
//
// The SAME source is compiled twice (tests/cpp/Makefile `hosts`, oracle/Makefile `ref`):
//   * against the genuine reference header (-I/root/reference, oracle/ref/prelude.h force-included)  -> oracle/_ref/ref_host_fx_*:
//     generates the golden vectors (oracle/gen_golden_hosts.py);
//   * against the DSL façade of this repo (include/klang/klang.h)                                       -> oracle/_ref/facade_host_fx_*:
//     the same calls end in klg_fx_process on the GPU.
// usage: effect_host scenario instance in.bin out.bin [--set-on-change]
//   scenario: tests/golden/*.scn (effect flavour); `instance`: which instance's control events to follow; in.bin: float32 [B][CH][N].
//   Default = the template's behaviour: every control is set() from its parameter EVERY block.  --set-on-change sets a control only in
//   the block its parameter changed (what the fixtures of oracle/gen_golden_fxexamples.py did).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include PATCH_FILE
#ifdef KLANG_GPU_TRACE_FLOAT
#undef float                 // (include/klang/klang.h: the patch's own text was compiled with `float` = the tracing signal)
#undef sizeof                // (... and with sizeof guarded against that type)
#endif
#ifdef KLANG_MI355
HOST_BIND_LINE
#endif

struct KlangEffectAudioProcessor {
	HOST_FX_TYPE pingpong;                                       // PluginProcessor.h
	std::vector<float> parameters;                               // getParameters()[c]->getValue()
	std::vector<char> changed;
	bool setOnChange = false;

	KlangEffectAudioProcessor() {
		// Register Klang Effect's parameters (PluginProcessor.cpp:24-46)
		for (unsigned int c = 0; c < pingpong.controls.size(); c++) {
			const klang::Control& control = pingpong.controls[c];
			parameters.push_back(control.initial);
			changed.push_back(0);
		}
	}
	void setParameter(int c, float value) { parameters[(size_t)c] = value; changed[(size_t)c] = 1; }

	void processBlock(float* const* channels, int numSamples) {
#ifdef KLANG_REF_BLOCK_SESSION
		KLANG_REF_BLOCK_SESSION(numSamples);                     // harness obligation of the genuine header
#endif
		// Setup the buffer for Klang (PluginProcessor.cpp:168-171)
		klang::buffer left(channels[0], numSamples);
#if HOST_FX_CHANNELS == 2
		klang::buffer right(channels[1], numSamples);
		klang::stereo::buffer buffers(left, right);
#endif
		// Update the Klang Synth's parameters (PluginProcessor.cpp:173-175)
		for (unsigned int c = 0; c < pingpong.controls.size(); c++)
			if (!setOnChange || changed[c]) { pingpong.controls[c].set(parameters[c]); changed[c] = 0; }
#if HOST_FX_CHANNELS == 2
		pingpong.klang::Stereo::Effect::process(buffers);      // PluginProcessor.cpp:177
#else
		pingpong.klang::Effect::process(left);
#endif
	}
};

struct Ev { int block, type, inst; float a, b; long seed; };

int main(int argc, char** argv) {
	if (argc < 5) { std::fprintf(stderr, "usage: %s scenario instance in.bin out.bin [--set-on-change]\n", argv[0]); return 1; }
	FILE* f = std::fopen(argv[1], "r");
	if (!f) return 1;
	const int instance = std::atoi(argv[2]);
	char tok[64]; int ver; float fsr = 48000.f; int block = 256, blocks = 1; std::vector<Ev> ev; std::vector<std::pair<int, float>> ctl;
	if (std::fscanf(f, "%63s %d", tok, &ver) != 2) return 1;
	while (std::fscanf(f, "%63s", tok) == 1) {
		std::string t(tok); int k; unsigned u;
		if (t == "end") break;
		else if (t == "patch") (void)!std::fscanf(f, "%63s", tok);
		else if (t == "fs") (void)!std::fscanf(f, "%f", &fsr);
		else if (t == "block") (void)!std::fscanf(f, "%d", &block);
		else if (t == "blocks") (void)!std::fscanf(f, "%d", &blocks);
		else if (t == "instances" || t == "burst" || t == "synths" || t == "notes") (void)!std::fscanf(f, "%d", &k);
		else if (t == "seed") (void)!std::fscanf(f, "%u", &u);
		else if (t == "dump") { (void)!std::fscanf(f, "%d", &k); for (int i = 0; i < k; i++) { int d; (void)!std::fscanf(f, "%d", &d); } }
		else if (t == "ctl") { int i; float v; (void)!std::fscanf(f, "%d %f", &i, &v); ctl.push_back({ i, v }); }
		else if (t == "ev") { Ev e; (void)!std::fscanf(f, "%d %d %d %f %f %ld", &e.block, &e.type, &e.inst, &e.a, &e.b, &e.seed); ev.push_back(e); }
	}
	std::fclose(f);
	klang::fs = klang::SampleRate(fsr);
	static KlangEffectAudioProcessor processor;                  // (static: the effect owns megabytes of delay line under the genuine header)
	for (int i = 5; i < argc; i++) if (!std::strcmp(argv[i], "--set-on-change")) processor.setOnChange = true;
	for (auto& c : ctl) processor.setParameter(c.first, c.second);
	const int CH = HOST_FX_CHANNELS, N = block;
	std::vector<float> io((size_t)CH * N);
	FILE* in = std::fopen(argv[3], "rb"); FILE* out = std::fopen(argv[4], "wb");
	if (!in || !out) return 1;
	size_t evi = 0;
	for (int b = 0; b < blocks; b++) {
		for (; evi < ev.size() && ev[evi].block <= b; evi++) if (ev[evi].type == 2 && ev[evi].inst == instance) processor.setParameter((int)ev[evi].a, ev[evi].b);
		if (std::fread(io.data(), 4, io.size(), in) != io.size()) return 2;
		float* channels[2] = { io.data(), io.data() + (CH == 2 ? N : 0) };
		processor.processBlock(channels, N);
		std::fwrite(io.data(), 4, io.size(), out);
	}
	std::fclose(in); std::fclose(out);
	return 0;
}
