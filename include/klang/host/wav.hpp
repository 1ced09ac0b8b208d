// include/klang/host/wav.hpp — RIFF/WAVE writer and reader for the headless host.
// replaces: File::WAV decode of the reference (klang.h:5991-6099, the data a Sample plays) and the audio device the JUCE
// wrapper writes to.  Writer: 32-bit float or 16-bit PCM, interleaved.  Reader: PCM 8/16/24/32 and float 32/64, any channel
// count, returned de-interleaved as floats in [-1, 1).  Mono PCM 8 / 16 / 32 and float 32 — everything File::WAV decodes — give
// the reference's floats bit for bit (tests/test_host_render.py::test_wav_reader_matches_the_reference_decoder, fixtures from
// oracle/gen_golden_wav.py); the reference hands a multi-channel file to its buffer as the first `frames` samples of the
// INTERLEAVED stream (klang.h:6085-6095: `decode<T>(.., buffer.size)` with STEP 1), this reader de-interleaves instead.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace klang { namespace host {

inline bool wav_write(const char* path, const std::vector<std::vector<float>>& channels, int sample_rate, bool float32, std::string* error = nullptr) {
	const uint32_t ch = (uint32_t)channels.size(), frames = ch ? (uint32_t)channels[0].size() : 0, bytes = float32 ? 4u : 2u;
	FILE* f = std::fopen(path, "wb");
	if (!f) { if (error) *error = std::string("cannot create ") + path; return false; }
	auto u32 = [&](uint32_t v) { std::fwrite(&v, 4, 1, f); };
	auto u16 = [&](uint16_t v) { std::fwrite(&v, 2, 1, f); };
	const uint32_t data = frames * ch * bytes;
	std::fwrite("RIFF", 1, 4, f); u32(36 + data); std::fwrite("WAVEfmt ", 1, 8, f); u32(16);
	u16(float32 ? 3 : 1); u16((uint16_t)ch); u32((uint32_t)sample_rate); u32((uint32_t)sample_rate * ch * bytes); u16((uint16_t)(ch * bytes)); u16((uint16_t)(bytes * 8));
	std::fwrite("data", 1, 4, f); u32(data);
	for (uint32_t i = 0; i < frames; i++) for (uint32_t c = 0; c < ch; c++) {
		const float x = channels[c][i];
		if (float32) std::fwrite(&x, 4, 1, f);
		else { float y = x * 32767.f; y = y > 32767.f ? 32767.f : (y < -32768.f ? -32768.f : y); const int16_t q = (int16_t)std::lrintf(y); std::fwrite(&q, 2, 1, f); }
	}
	std::fclose(f);
	return true;
}

struct WavData { int sample_rate = 0; std::vector<std::vector<float>> channels; std::string error; };

inline bool wav_read(const char* path, WavData& out) {
	FILE* f = std::fopen(path, "rb");
	if (!f) { out.error = std::string("cannot open ") + path; return false; }
	std::vector<uint8_t> d; uint8_t buf[65536]; size_t n;
	while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
	std::fclose(f);
	auto le32 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24); };
	auto le16 = [&](size_t o) { return (int)((uint32_t)d[o] | ((uint32_t)d[o + 1] << 8)); };
	if (d.size() < 12 || std::memcmp(d.data(), "RIFF", 4) || std::memcmp(d.data() + 8, "WAVE", 4)) { out.error = "not a RIFF/WAVE file"; return false; }
	int fmt = 0, ch = 0, bits = 0; size_t data_at = 0, data_len = 0;
	for (size_t o = 12; o + 8 <= d.size();) {
		const uint32_t len = le32(o + 4);
		if (!std::memcmp(d.data() + o, "fmt ", 4) && len >= 16 && o + 24 <= d.size()) {
			fmt = le16(o + 8); ch = le16(o + 10); out.sample_rate = (int)le32(o + 12); bits = le16(o + 22);
			if (fmt == 0xFFFE && len >= 26 && o + 34 <= d.size()) fmt = le16(o + 32);          // WAVE_FORMAT_EXTENSIBLE: first two bytes of the sub-format GUID
		}
		else if (!std::memcmp(d.data() + o, "data", 4)) { data_at = o + 8; data_len = len; if (data_at + data_len > d.size()) data_len = d.size() - data_at; break; }
		o += 8 + (size_t)len + (len & 1u);
	}
	if (!ch || !data_at || (fmt != 1 && fmt != 3)) { out.error = "unsupported WAVE encoding (PCM and IEEE float only)"; return false; }
	const size_t bps = (size_t)bits / 8, frames = bps ? data_len / (bps * (size_t)ch) : 0;
	out.channels.assign((size_t)ch, std::vector<float>(frames));
	for (size_t i = 0; i < frames; i++) for (int c = 0; c < ch; c++) {
		const uint8_t* p = d.data() + data_at + (i * (size_t)ch + (size_t)c) * bps;
		float x = 0.f;
		if (fmt == 3 && bits == 32) std::memcpy(&x, p, 4);
		else if (fmt == 3 && bits == 64) { double v; std::memcpy(&v, p, 8); x = (float)v; }
		else if (bits == 8) x = ((float)p[0] - 128.f) * (1.f / 255.f);                       // as the reference decodes unsigned samples: (x - 2^7) / (2^8 - 1), klang.h:6070-6075
		else if (bits == 16) x = (float)(int16_t)(p[0] | (p[1] << 8)) / 32768.f;
		else if (bits == 24) x = (float)(((int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24)) >> 8) / 8388608.f;
		else if (bits == 32) { int32_t v; std::memcpy(&v, p, 4); x = (float)((double)v / 2147483648.0); }
		else { out.error = "unsupported sample width"; return false; }
		out.channels[(size_t)c][i] = x;
	}
	return true;
}

} }
