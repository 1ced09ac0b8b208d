// tests/cpp/facade_fx_scenario.cpp — drives an effect patch (.k) THROUGH THE DSL FAÇADE: klang::gpu::EffectBank<FX> records the
// patch's process(), the blocks are rendered by libklang_mi355.so.  Reads an effect scenario (tests/scenario_io.py format:
// instances / block / blocks / ctl / control events) and the input blocks float32 [B][K][CH][N] prepared by the test
// (the same hash noise the reference fixtures were generated with); writes the processed blocks in the same layout.
#include <cstdio>
#include <string>
#include <vector>

#include PATCH_FILE
#ifdef KLANG_GPU_TRACE_FLOAT
#undef float                 // (include/klang/klang.h: the patch's own text was compiled with `float` = the tracing signal)
#undef sizeof                // (... and with sizeof guarded against that type)
#endif
#ifdef FX_BIND_LINE
FX_BIND_LINE
#endif

struct Ev { int block, type, inst; float a, b; long seed; };

int main(int argc, char** argv) {
	if (argc < 4) { std::fprintf(stderr, "usage: %s scenario in.bin out.bin\n", argv[0]); return 1; }
	FILE* f = std::fopen(argv[1], "r");
	if (!f) return 1;
	char tok[64]; int ver; float fsr = 48000.f; int block = 256, blocks = 1, K = 1; std::vector<Ev> ev; std::vector<std::pair<int, float>> ctl;
	if (std::fscanf(f, "%63s %d", tok, &ver) != 2) return 1;
	while (std::fscanf(f, "%63s", tok) == 1) {
		std::string t(tok); int k; unsigned u;
		if (t == "end") break;
		else if (t == "patch") (void)!std::fscanf(f, "%63s", tok);
		else if (t == "fs") (void)!std::fscanf(f, "%f", &fsr);
		else if (t == "block") (void)!std::fscanf(f, "%d", &block);
		else if (t == "blocks") (void)!std::fscanf(f, "%d", &blocks);
		else if (t == "instances") (void)!std::fscanf(f, "%d", &K);
		else if (t == "burst") (void)!std::fscanf(f, "%d", &k);
		else if (t == "seed") (void)!std::fscanf(f, "%u", &u);
		else if (t == "synths" || t == "notes") (void)!std::fscanf(f, "%d", &k);
		else if (t == "dump") { (void)!std::fscanf(f, "%d", &k); for (int i = 0; i < k; i++) { int d; (void)!std::fscanf(f, "%d", &d); } }
		else if (t == "ctl") { int i; float v; (void)!std::fscanf(f, "%d %f", &i, &v); ctl.push_back({ i, v }); }
		else if (t == "ev") { Ev e; (void)!std::fscanf(f, "%d %d %d %f %f %ld", &e.block, &e.type, &e.inst, &e.a, &e.b, &e.seed); ev.push_back(e); }
	}
	std::fclose(f);
	klang::fs = klang::SampleRate(fsr);
	klang::gpu::EffectBank<FX_TYPE> bank(K, block);
	const int CH = bank.channels, N = block;
	for (auto& c : ctl) for (int k = 0; k < K; k++) bank.set(k, c.first, c.second);
	std::vector<float> io((size_t)K * CH * N);
	FILE* in = std::fopen(argv[2], "rb"); FILE* out = std::fopen(argv[3], "wb");
	if (!in || !out) return 1;
	size_t evi = 0;
	for (int b = 0; b < blocks; b++) {
		for (; evi < ev.size() && ev[evi].block <= b; evi++) if (ev[evi].type == 2) bank.set(ev[evi].inst, (int)ev[evi].a, ev[evi].b);
		if (std::fread(io.data(), 4, io.size(), in) != io.size()) return 2;
		bank.process(io.data(), N);
		std::fwrite(io.data(), 4, io.size(), out);
	}
	std::fclose(in); std::fclose(out);
	return 0;
}
