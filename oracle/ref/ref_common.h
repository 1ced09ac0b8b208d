// oracle/ref/ref_common.h — TEST INFRASTRUCTURE, not product code.
//
// Scenario-driven harness around the GENUINE reference header
// (/root/reference/klang.h, included by path at build time — never copied).
// Each oracle/ref/ref_<patch>.cpp includes prelude.h, <klang.h>, one patch,
// then this file, and instantiates run_synth<>/run_effect<>.  The binaries
// land in oracle/_ref/ (git-ignored) and are only ever used to
//   (1) generate the golden vectors under tests/golden/ (oracle/gen_golden.py)
//   (2) pin the C restatement in oracle/klang_oracle.c.
//
// Harness obligations taken from SURVEY.md §8(c): klang::fs set in the same TU,
// one klang::Debug::Session per processed block, pre-cleared buffers,
// klang::random(seed) before every noteOn that carries a seed, single thread.
//
// Scenario file (text tokens):
//   klgscn 1
//   patch <name> | fs <f> | block <N> | blocks <B> | synths <S> | notes <P>
//   dump <k> b0 .. bk-1          blocks whose per-voice output is written
//   ctl <idx> <value>            initial control value (all synth instances)
//   ev <block> <type> <synth> <a> <b> <seed>
//        type 0 noteOn(pitch=a, velocity=b)  [srand(seed) first if seed>=0]
//        type 1 noteOff(pitch=a, velocity=b)
//        type 2 control idx=a value=b (controls[a].set(b) + onControl)
//   end
// Output file (little-endian):
//   int32 magic 'KLGO', V, N, ndump, B
//   float32 [ndump][V][N]   per-voice block output (zeros for Off voices)
//   float32 [B][2][N]       stereo mix = fp32 sum over voices in index order
//   uint8   [B][V]          NoteBase::stage after each block
// Synths whose notes have a STEREO `out` (Stereo::Note, klang.h:4721-4733; run_synth<.., 2>): magic 'KLGS' and
//   float32 [ndump][V][2][N] per-voice block output, left then right; the mix sums each channel over the voices
#pragma once
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

struct RefEvent { int block, type, synth; float a, b; long seed; };
struct RefScenario {
	std::string patch;
	float fs = 48000.f;
	int block = 256, blocks = 1, synths = 1, notes = 1;
	std::vector<int> dump;
	std::vector<std::pair<int, float>> ctl;
	std::vector<RefEvent> ev;
	// effect scenarios
	int instances = 0, burst = 0;
	unsigned seed = 0;
};

static bool ref_load(const char* path, RefScenario& s) {
	FILE* f = fopen(path, "r");
	if (!f) { fprintf(stderr, "cannot open %s\n", path); return false; }
	char tok[64];
	int ver = 0;
	if (fscanf(f, "%63s %d", tok, &ver) != 2 || std::string(tok) != "klgscn") { fclose(f); return false; }
	while (fscanf(f, "%63s", tok) == 1) {
		std::string t(tok);
		if (t == "end") break;
		else if (t == "patch") { fscanf(f, "%63s", tok); s.patch = tok; }
		else if (t == "fs") fscanf(f, "%f", &s.fs);
		else if (t == "block") fscanf(f, "%d", &s.block);
		else if (t == "blocks") fscanf(f, "%d", &s.blocks);
		else if (t == "synths") fscanf(f, "%d", &s.synths);
		else if (t == "notes") fscanf(f, "%d", &s.notes);
		else if (t == "instances") fscanf(f, "%d", &s.instances);
		else if (t == "burst") fscanf(f, "%d", &s.burst);
		else if (t == "seed") fscanf(f, "%u", &s.seed);
		else if (t == "dump") { int k = 0; fscanf(f, "%d", &k); s.dump.resize(k); for (int i = 0; i < k; i++) fscanf(f, "%d", &s.dump[i]); }
		else if (t == "ctl") { int i; float v; fscanf(f, "%d %f", &i, &v); s.ctl.push_back({ i, v }); }
		else if (t == "ev") { RefEvent e; fscanf(f, "%d %d %d %f %f %ld", &e.block, &e.type, &e.synth, &e.a, &e.b, &e.seed); s.ev.push_back(e); }
		else { fprintf(stderr, "bad token %s\n", tok); fclose(f); return false; }
	}
	fclose(f);
	return true;
}

static inline bool ref_is_dump(const RefScenario& s, int b) {
	for (int d : s.dump) if (d == b) return true;
	return false;
}

// Deterministic synthetic effect input shared by harness, oracle and GPU tests:
// lowbias32 integer hash -> U(-0.5,0.5) for the first `burst` samples, then silence.
static inline unsigned ref_hash32(unsigned x) {
	x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
static inline float ref_fx_input(unsigned seed, unsigned instance, unsigned ch, unsigned t, unsigned burst) {
	if (t >= burst) return 0.f;
	const unsigned h = ref_hash32(seed ^ ref_hash32(instance * 2u + ch) ^ (t * 0x9e3779b9U));
	return (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

#ifdef REF_WITH_KLANG

// NOTE_KIND: 0 = klang::Note (mono, process(buffer) overwrites),
//            1 = Stereo::Mono::Note (process(Stereo::buffer) accumulates L and R)
//            2 = Stereo::Note with a stereo `out` (process(Stereo::buffer): `buffer++ += out`, klang.h:4727-4733)
template<class SYNTH, int NOTE_KIND>
static int run_synth(const RefScenario& s, const char* outpath) {
	klang::fs = klang::SampleRate(s.fs);   // fs is static per TU (klang.h:1593-1604)
	const int N = s.block, P = s.notes, S = s.synths, V = S * P, B = s.blocks;

	std::vector<SYNTH*> synths(S);
	for (int i = 0; i < S; i++) {
		synths[i] = new SYNTH();
		if ((int)synths[i]->notes.count != P) { fprintf(stderr, "patch has %d notes per synth, scenario says %d\n", (int)synths[i]->notes.count, P); return 2; }
		for (auto& c : s.ctl) synths[i]->controls[c.first].set(c.second);
	}

	FILE* out = fopen(outpath, "wb");
	if (!out) return 3;
	constexpr int NC = NOTE_KIND == 2 ? 2 : 1;              // channels of a voice's own output
	const int hdr[5] = { NOTE_KIND == 2 ? 0x53474C4B : 0x4F474C4B, V, N, (int)s.dump.size(), B };
	fwrite(hdr, sizeof(int), 5, out);

	std::vector<float> voice((size_t)V * N * NC), mix((size_t)B * 2 * N, 0.f), tmpL(N), tmpR(N);
	std::vector<unsigned char> stages((size_t)B * V);

	size_t evi = 0;
	for (int b = 0; b < B; b++) {
		for (; evi < s.ev.size() && s.ev[evi].block <= b; evi++) {
			const RefEvent& e = s.ev[evi];
			SYNTH* sy = synths[e.synth];
			if (e.type == 0) { if (e.seed >= 0) klang::random((unsigned)e.seed); sy->noteOn((int)e.a, e.b); }
			else if (e.type == 1) sy->noteOff((int)e.a, e.b);
			else if (e.type == 2) { sy->controls[(int)e.a].set(e.b); sy->onControl((int)e.a, sy->controls[(int)e.a].value); }
		}
		std::fill(voice.begin(), voice.end(), 0.f);
		float* mixL = &mix[((size_t)b * 2 + 0) * N];
		float* mixR = &mix[((size_t)b * 2 + 1) * N];
		for (int v = 0; v < V; v++) {
			SYNTH* sy = synths[v / P];
			auto* note = sy->notes[v % P];
			float* dst = &voice[(size_t)v * N * NC];
			if (note->stage != SYNTH::Note::Off) {
				klang::Debug::Session session(nullptr, N, klang::Debug::Buffer::Synth);
				bool alive;
				if constexpr (NOTE_KIND == 0) {
					klang::buffer mono(dst, N);
					alive = note->process(mono);
				}
				else {
					std::fill(tmpL.begin(), tmpL.end(), 0.f);
					std::fill(tmpR.begin(), tmpR.end(), 0.f);
					klang::buffer left(tmpL.data(), N), right(tmpR.data(), N);
					klang::Stereo::buffer st(left, right);
					alive = note->process(st);
					for (int i = 0; i < N; i++) dst[i] = tmpL[i];   // Mono::Note writes L == R
					if constexpr (NOTE_KIND == 2) for (int i = 0; i < N; i++) dst[N + i] = tmpR[i];
				}
				if (!alive) note->stop();
				for (int i = 0; i < N; i++) { mixL[i] += dst[i]; mixR[i] += dst[(NC - 1) * N + i]; }
			}
			stages[(size_t)b * V + v] = (unsigned char)note->stage;
		}
		if (ref_is_dump(s, b)) fwrite(voice.data(), sizeof(float), voice.size(), out);
	}
	fwrite(mix.data(), sizeof(float), mix.size(), out);
	fwrite(stages.data(), 1, stages.size(), out);
	fclose(out);
	for (auto* sy : synths) delete sy;
	return 0;
}

// Stereo effect instances: K independent EFFECT objects, in-place stereo blocks.
// Output: int32 magic 'KLGF', K, N, ndump, B ; float32 [ndump][K][2][N]
template<class EFFECT>
static int run_effect(const RefScenario& s, const char* outpath) {
	klang::fs = klang::SampleRate(s.fs);
	const int N = s.block, K = s.instances, B = s.blocks;
	std::vector<EFFECT*> fx(K);
	for (int k = 0; k < K; k++) {
		fx[k] = new EFFECT();
		for (auto& c : s.ctl) fx[k]->controls[c.first].set(c.second);
	}
	FILE* out = fopen(outpath, "wb");
	if (!out) return 3;
	const int hdr[5] = { 0x46474C4B, K, N, (int)s.dump.size(), B };
	fwrite(hdr, sizeof(int), 5, out);
	std::vector<float> io((size_t)K * 2 * N);
	size_t evi = 0;
	for (int b = 0; b < B; b++) {
		for (; evi < s.ev.size() && s.ev[evi].block <= b; evi++) {
			const RefEvent& e = s.ev[evi];
			if (e.type == 2) fx[e.synth]->controls[(int)e.a].set(e.b);
		}
		for (int k = 0; k < K; k++) {
			float* L = &io[((size_t)k * 2 + 0) * N];
			float* R = &io[((size_t)k * 2 + 1) * N];
			for (int i = 0; i < N; i++) {
				L[i] = ref_fx_input(s.seed, k, 0, (unsigned)(b * N + i), s.burst);
				R[i] = ref_fx_input(s.seed, k, 1, (unsigned)(b * N + i), s.burst);
			}
			klang::Debug::Session session(nullptr, N, klang::Debug::Buffer::Effect);  // F8: rewinds the TLS debug buffer
			klang::buffer left(L, N), right(R, N);
			klang::Stereo::buffer st(left, right);
			fx[k]->klang::Stereo::Effect::process(st);
		}
		if (ref_is_dump(s, b)) fwrite(io.data(), sizeof(float), io.size(), out);
	}
	fclose(out);
	for (auto* e : fx) delete e;
	return 0;
}

// Example effects (klang::Effect = 1 channel, Stereo::Effect = 2): every block is written.
// Output: int32 magic 'KLGX', K, N, B, CH ; float32 [B][K][CH][N]
template<class EFFECT, int CH>
static int run_effect_example(const RefScenario& s, const char* outpath) {
	klang::fs = klang::SampleRate(s.fs);
	const int N = s.block, K = s.instances, B = s.blocks;
	std::vector<EFFECT*> fx(K);
	for (int k = 0; k < K; k++) {
		fx[k] = new EFFECT();
		for (auto& c : s.ctl) fx[k]->controls[c.first].set(c.second);
	}
	FILE* out = fopen(outpath, "wb");
	if (!out) return 3;
	const int hdr[5] = { 0x58474C4B, K, N, B, CH };
	fwrite(hdr, sizeof(int), 5, out);
	std::vector<float> io((size_t)K * CH * N);
	size_t evi = 0;
	for (int b = 0; b < B; b++) {
		for (; evi < s.ev.size() && s.ev[evi].block <= b; evi++) {
			const RefEvent& e = s.ev[evi];
			if (e.type == 2) fx[e.synth]->controls[(int)e.a].set(e.b);
		}
		for (int k = 0; k < K; k++) {
			float* ch[2] = { &io[((size_t)k * CH + 0) * N], &io[((size_t)k * CH + (CH - 1)) * N] };
			for (int c = 0; c < CH; c++) for (int i = 0; i < N; i++) ch[c][i] = ref_fx_input(s.seed, k, c, (unsigned)(b * N + i), s.burst);
			klang::Debug::Session session(nullptr, N, klang::Debug::Buffer::Effect);
			if constexpr (CH == 1) { klang::buffer mono(ch[0], N); fx[k]->klang::Effect::process(mono); }
			else { klang::buffer left(ch[0], N), right(ch[1], N); klang::Stereo::buffer st(left, right); fx[k]->klang::Stereo::Effect::process(st); }
		}
		fwrite(io.data(), sizeof(float), io.size(), out);
	}
	fclose(out);
	for (auto* e : fx) delete e;
	return 0;
}

#endif // REF_WITH_KLANG
