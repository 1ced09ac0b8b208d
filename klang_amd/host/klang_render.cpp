// klang_amd/host/klang_render.cpp — headless host: Standard MIDI File in, WAV out.
// replaces: the JUCE wrapper of the reference (templates/juce/synth/Source/PluginProcessor.cpp:153-182): per block it hands the
// block's MIDI messages to the synth (0x90 with velocity > 0 -> noteOn(pitch, velocity / 127), 0x80 or 0x90 with velocity 0
// -> noteOff; the template's klang.h:3921-3928), copies the host parameters into the controls and asks the synth for the
// block's audio.  Here the synth is a .k patch compiled UNCHANGED against the DSL façade (include/klang/klang.h): on()/off()
// run on the host, the blocks are rendered by libklang_mi355.so on the GPU.
//
// Build (one binary per patch, like the reference's one plugin per patch):
//   clang++ -std=c++17 -O2 -ffp-contract=off -DPATCH_FILE='"patch.k"' -DSYNTH_TYPE=MySynth [-DMONO_SYNTH] klang_render.cpp \
//           -Iinclude -Iinclude/klang -Lklang_amd -lklang_mi355 -o klang_render_mysynth
// Run:
//   klang_render in.mid out.wav [--fs 48000] [--block 256] [--tail 2.0] [--pcm16] [--preset i] [--control i=value]...
//                [--cc number=control]... [--channel c] [--events]
//     --preset i        start from presets[i] (klang.h:1940-1981: its values are set on the controls, then onPreset(i))
//     --control i=v     controls[i].set(v) before the first block
//     --cc n=i          MIDI controller n drives controls[i] (0..127 -> setNormalised(v / 127), klang.h:1721); program change
//                       p selects presets[p]
//     --events          print the merged, time-stamped MIDI messages and exit (no GPU needed)
//   klang_render --wav-info file.wav      decode a WAV file (the reader a Sample's data comes through) and print its shape
//   klang_render --wav-dump file.wav      ... and print channel 0 as the bit patterns of its floats
// Events that fall inside a block are delivered before that block is rendered (as the reference's processBlock does: it
// ignores the messages' sample offsets).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include <klang/host/smf.hpp>
#include <klang/host/wav.hpp>

#ifdef PATCH_FILE
#include PATCH_FILE
#include <klang/bindings.h>
#ifdef BIND_LINE
BIND_LINE
#endif
#endif

int main(int argc, char** argv) {
	const char* in = nullptr; const char* out = nullptr;
	float fsr = 48000.f, tail = 2.f; int block = 256, preset = -1, channel = -1; bool pcm16 = false, events_only = false;
	std::vector<std::pair<int, float>> initial; std::map<int, int> cc;
	if (argc == 3 && std::string(argv[1]) == "--wav-info") {
		klang::host::WavData w;
		if (!klang::host::wav_read(argv[2], w)) { std::fprintf(stderr, "%s: %s\n", argv[2], w.error.c_str()); return 1; }
		std::printf("rate %d channels %zu frames %zu\n", w.sample_rate, w.channels.size(), w.channels.empty() ? (size_t)0 : w.channels[0].size());
		for (size_t c = 0; c < w.channels.size(); c++) { double sum = 0, peak = 0; for (float x : w.channels[c]) { sum += x; peak = std::fabs(x) > peak ? std::fabs(x) : peak; } std::printf("channel %zu sum %.9g peak %.9g first %.9g\n", c, sum, peak, w.channels[c].empty() ? 0.0 : (double)w.channels[c][0]); }
		return 0;
	}
	if (argc == 3 && std::string(argv[1]) == "--wav-dump") {                     // channel 0 as the bit patterns of its floats (parity with the reference's decoder)
		klang::host::WavData w;
		if (!klang::host::wav_read(argv[2], w)) { std::fprintf(stderr, "%s: %s\n", argv[2], w.error.c_str()); return 1; }
		const std::vector<float> none; const std::vector<float>& c0 = w.channels.empty() ? none : w.channels[0];
		std::printf("%zu\n", c0.size());
		for (float x : c0) { unsigned u; std::memcpy(&u, &x, 4); std::printf("%08x\n", u); }
		return 0;
	}
	for (int i = 1; i < argc; i++) {
		const std::string a = argv[i];
		auto need = [&](const char* what) { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", what); std::exit(2); } return argv[++i]; };
		if (a == "--fs") fsr = (float)std::atof(need("--fs"));
		else if (a == "--block") block = std::atoi(need("--block"));
		else if (a == "--tail") tail = (float)std::atof(need("--tail"));
		else if (a == "--preset") preset = std::atoi(need("--preset"));
		else if (a == "--channel") channel = std::atoi(need("--channel"));
		else if (a == "--pcm16") pcm16 = true;
		else if (a == "--events") events_only = true;
		else if (a == "--control") { int k; float v; if (std::sscanf(need("--control"), "%d=%f", &k, &v) != 2) { std::fprintf(stderr, "--control i=value\n"); return 2; } initial.push_back({ k, v }); }
		else if (a == "--cc") { int n, k; if (std::sscanf(need("--cc"), "%d=%d", &n, &k) != 2) { std::fprintf(stderr, "--cc number=control\n"); return 2; } cc[n] = k; }
		else if (!in) in = argv[i];
		else if (!out) out = argv[i];
		else { std::fprintf(stderr, "unexpected argument %s\n", argv[i]); return 2; }
	}
	if (!in || (!out && !events_only) || block < 1 || block > 1024 || !(fsr > 0.f)) {
		std::fprintf(stderr, "usage: %s in.mid out.wav [--fs 48000] [--block 256 (<= 1024)] [--tail seconds] [--pcm16] [--preset i] [--control i=v] [--cc n=i] [--channel c] [--events]\n", argv[0]);
		return 2;
	}
	klang::host::SmfFile smf;
	if (!smf.load(in)) { std::fprintf(stderr, "%s: %s\n", in, smf.error.c_str()); return 1; }
	if (events_only) {
		std::printf("format %d tracks %d division %d events %zu\n", smf.format, smf.tracks, smf.division, smf.events.size());
		for (const auto& e : smf.events) std::printf("%.9f %llu %d %02x %d %d\n", e.seconds, (unsigned long long)e.tick, e.track, e.status, e.data1, e.data2);
		return 0;
	}
#ifndef PATCH_FILE
	std::fprintf(stderr, "this binary was built without a patch (-DPATCH_FILE / -DSYNTH_TYPE): only --events is available\n");
	return 2;
#else
	klang::fs = klang::SampleRate(fsr);
	SYNTH_TYPE synth;
	auto apply_preset = [&](int p) {
		if (p < 0 || p >= (int)synth.presets.items.size()) return;
		const auto& v = synth.presets.items[(size_t)p].values;
		for (size_t c = 0; c < v.size() && c < synth.controls.size(); c++) { synth.controls[(int)c].set(v[c]); synth.onControl((int)c, synth.controls[(int)c].value); }
		synth.onPreset(p);
	};
	apply_preset(preset);
	for (const auto& kv : initial) if (kv.first >= 0 && kv.first < (int)synth.controls.size()) { synth.controls[kv.first].set(kv.second); synth.onControl(kv.first, synth.controls[kv.first].value); }
	const double last = smf.events.empty() ? 0.0 : smf.events.back().seconds;
	const long long total = (long long)std::ceil((last + (double)tail) * (double)fsr);
	const long long blocks = (total + block - 1) / block;
	std::vector<std::vector<float>> audio(2, std::vector<float>((size_t)(blocks * block), 0.f));
	size_t evi = 0; long long notes_on = 0;
	for (long long b = 0; b < blocks; b++) {
		const long long end = (b + 1) * block;                                   // messages stamped before the end of this block belong to it
		for (; evi < smf.events.size() && (long long)std::floor(smf.events[evi].seconds * (double)fsr) < end; evi++) {
			const auto& e = smf.events[evi];
			if (channel >= 0 && (e.status & 0x0F) != channel) continue;
			const int kind = e.status & 0xF0;
			if (kind == 0x90 && e.data2 > 0) { synth.noteOn(e.data1, e.data2 / 127.f); notes_on++; }
			else if (kind == 0x80 || (kind == 0x90 && e.data2 == 0)) synth.noteOff(e.data1, e.data2 / 127.f);
			else if (kind == 0xB0) { const auto it = cc.find(e.data1); if (it != cc.end() && it->second >= 0 && it->second < (int)synth.controls.size()) { synth.controls[it->second].setNormalised(e.data2 / 127.f); synth.onControl(it->second, synth.controls[it->second].value); } }
			else if (kind == 0xC0) apply_preset(e.data1);
		}
		float* bufs[2] = { audio[0].data() + b * block, audio[1].data() + b * block };
#ifdef MONO_SYNTH
		synth.process(bufs[0], block);
		std::memcpy(bufs[1], bufs[0], sizeof(float) * (size_t)block);
#else
		synth.process(bufs, block);
#endif
	}
	std::string err;
	if (!klang::host::wav_write(out, audio, (int)fsr, !pcm16, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
	double peak = 0; for (const auto& c : audio) for (float x : c) peak = std::fabs(x) > peak ? std::fabs(x) : peak;
	std::printf("%s: %lld blocks of %d at %.0f Hz, %lld note-ons, peak %.4f -> %s\n", in, blocks, block, (double)fsr, notes_on, peak, out);
	return 0;
#endif
}
