// include/klang/host/smf.hpp — Standard MIDI File reader for the headless host.
// replaces: the MIDI side of the JUCE wrapper (templates/juce/synth/Source/PluginProcessor.cpp:166-168 hands each block's
// juce::MidiBuffer to the synth message by message).  Formats 0 and 1, PPQ and SMPTE divisions, tempo changes (meta 0x51),
// running status, sysex and other meta events skipped.  The result is the channel messages of all tracks merged in time
// order (ties keep track order, then file order), each stamped with its time in seconds.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace klang { namespace host {

struct MidiEvent { double seconds; uint64_t tick; int track, order; uint8_t status, data1, data2; };

struct SmfFile {
	int format = 0, tracks = 0, division = 480;
	std::vector<MidiEvent> events;
	std::string error;

	bool load(const char* path) {
		FILE* f = std::fopen(path, "rb");
		if (!f) { error = std::string("cannot open ") + path; return false; }
		std::vector<uint8_t> d; uint8_t buf[65536]; size_t n;
		while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
		std::fclose(f);
		return parse(d.data(), d.size());
	}

	bool parse(const uint8_t* d, size_t n) {
		events.clear(); error.clear();
		auto be32 = [&](size_t o) { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; };
		auto be16 = [&](size_t o) { return (int)(((uint32_t)d[o] << 8) | d[o + 1]); };
		if (n < 14 || d[0] != 'M' || d[1] != 'T' || d[2] != 'h' || d[3] != 'd' || be32(4) < 6) { error = "not a Standard MIDI File (no MThd)"; return false; }
		format = be16(8); tracks = be16(10); division = be16(12);
		if (format > 1) { error = "SMF format 2 (independent patterns) is not supported"; return false; }
		size_t o = 8 + be32(4);
		struct Tempo { uint64_t tick; uint32_t us_per_quarter; };
		std::vector<Tempo> tempo;
		int order = 0;
		for (int t = 0; t < tracks; t++) {
			if (o + 8 > n || d[o] != 'M' || d[o + 1] != 'T' || d[o + 2] != 'r' || d[o + 3] != 'k') { error = "truncated file (missing MTrk)"; return false; }
			const size_t end = o + 8 + be32(o + 4);
			if (end > n) { error = "truncated track"; return false; }
			size_t p = o + 8; uint64_t tick = 0; uint8_t running = 0;
			auto varlen = [&](uint32_t& v) { v = 0; for (int k = 0; k < 4 && p < end; k++) { const uint8_t b = d[p++]; v = (v << 7) | (b & 0x7Fu); if (!(b & 0x80u)) return true; } return false; };
			while (p < end) {
				uint32_t delta;
				if (!varlen(delta)) { error = "bad delta time"; return false; }
				tick += delta;
				if (p >= end) break;
				uint8_t st = d[p];
				if (st == 0xFF) {                                                   // meta
					if (p + 2 > end) { error = "truncated meta event"; return false; }
					const uint8_t type = d[p + 1]; p += 2; uint32_t len;
					if (!varlen(len) || p + len > end) { error = "bad meta length"; return false; }
					if (type == 0x51 && len == 3) tempo.push_back({ tick, ((uint32_t)d[p] << 16) | ((uint32_t)d[p + 1] << 8) | d[p + 2] });
					p += len; running = 0;
					if (type == 0x2F) break;
				}
				else if (st == 0xF0 || st == 0xF7) { p++; uint32_t len; if (!varlen(len) || p + len > end) { error = "bad sysex length"; return false; } p += len; running = 0; }
				else {
					if (st & 0x80u) { running = st; p++; } else st = running;
					if (!(st & 0x80u)) { error = "data byte without a running status"; return false; }
					const int nd = ((st & 0xF0u) == 0xC0u || (st & 0xF0u) == 0xD0u) ? 1 : 2;
					if (p + nd > end) { error = "truncated channel message"; return false; }
					MidiEvent e; e.seconds = 0; e.tick = tick; e.track = t; e.order = order++; e.status = st; e.data1 = d[p]; e.data2 = nd == 2 ? d[p + 1] : 0;
					p += nd;
					events.push_back(e);
				}
			}
			o = end;
		}
		std::stable_sort(tempo.begin(), tempo.end(), [](const Tempo& a, const Tempo& b) { return a.tick < b.tick; });
		std::stable_sort(events.begin(), events.end(), [](const MidiEvent& a, const MidiEvent& b) { return a.tick != b.tick ? a.tick < b.tick : (a.track != b.track ? a.track < b.track : a.order < b.order); });
		// ticks -> seconds
		if (division & 0x8000) {                                                    // SMPTE: frames per second x ticks per frame
			const int fps = 256 - ((division >> 8) & 0xFF), tpf = division & 0xFF;
			for (auto& e : events) e.seconds = (double)e.tick / ((double)fps * tpf);
		}
		else {
			size_t ti = 0; uint64_t t0 = 0; double s0 = 0.0; double us = 500000.0;    // 120 bpm until told otherwise
			for (auto& e : events) {
				while (ti < tempo.size() && tempo[ti].tick <= e.tick) { s0 += (double)(tempo[ti].tick - t0) * us / (1e6 * division); t0 = tempo[ti].tick; us = tempo[ti].us_per_quarter; ti++; }
				e.seconds = s0 + (double)(e.tick - t0) * us / (1e6 * division);
			}
		}
		return true;
	}
};

} }
