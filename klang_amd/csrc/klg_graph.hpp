// klang_amd/csrc/klg_graph.hpp — graph patches (include/klang_mi355_graph.h): program text -> HIP source of a patch
// body over the device primitives of klg_device.hpp -> hipRTC (gfx950) -> code object with klg_render<PatchGen, *>.
//
// The generated struct has exactly the shape of the hand-written patches in klg_patches.hpp (Rec / kStoreMask / Live /
// begin / sample / end), so the render kernel, the voice mix, the record planes in HBM and the host protocol are the
// ones the shipped patches use; only sample() comes from the recorded program.  hipRTC is loaded with dlopen at the
// first use (a process that never creates a graph patch does not need it).  Compiling needs no GPU.
#pragma once
#include <dlfcn.h>
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstring>

#include <map>
#include <mutex>
#include <regex>
#include <string>
#include <vector>

#include <algorithm>
#include <cstdarg>
#include <functional>

#include "../../include/klang_mi355_graph.h"

namespace klg { namespace graphrt {

using graph::Program;
using graph::Op;
} }
#include "klg_graph_staged.hpp"
namespace klg { namespace graphrt {

// ------------------------------------------------------------------------------------------------
// source generation
// ------------------------------------------------------------------------------------------------
inline std::string fmt(const char* f, ...) {
	char b[512]; va_list ap; va_start(ap, f); vsnprintf(b, sizeof b, f, ap); va_end(ap); return std::string(b);
}

// Can this program run two voices per lane?  Only node kinds / ops that have packed forms in klg_device_x2.hpp.
inline bool x2_eligible(const Program& g) {
	using namespace graph;
	if (g.channels || g.prepare_ops || g.stereo_note()) return false;
	for (int k : g.nodes) if (!(k == N_FSINE || k == N_SAW || k == N_PULSE || k == N_LPF || k == N_ENV || k == N_ADSR || k == N_PARAM)) return false;
	for (const Op& o : g.ops) switch (o.code) {
		case OP_CONST: case OP_CTL: case OP_PARAM: case OP_OSC: case OP_LPF: case OP_ENV: case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: case OP_NEG:
		case OP_STOPIF: case OP_STOP: case OP_SETPARAM: break;
		default: return false;
	}
	return true;
}

inline std::string generate_source(const Program& g, bool x2 = false, StagedPlan* staged = nullptr, int staged_G = 0) {
	using namespace graph;
	// type names of the generated body: one voice per lane, or two (the packed primitives overload the scalar names)
	const std::string TF = x2 ? "f2" : "float", TI = x2 ? "i2" : "int", TU = x2 ? "u2" : "uint32_t", T2 = x2 ? "2" : "";
	const int NW = g.words();
	std::vector<bool> swept(g.nodes.size(), false), written(g.nodes.size(), false), retuned(g.nodes.size(), false), dutied(g.nodes.size(), false);   // dutied: a duty set per sample (oscset 3) is written back
	std::vector<bool> reset_head(g.nodes.size(), false), head_used(g.nodes.size(), false);
	for (const Op& o : g.ops) if (o.code == OP_DELAYSET || o.code == OP_DELAYOUT) head_used[(size_t)o.node] = true;
	for (const Op& o : g.ops) if (o.code == OP_DELAYSET) reset_head[(size_t)o.node] = true;
	for (const Op& o : g.ops) { if (o.code == OP_LPFSET) swept[(size_t)o.node] = true; if (o.code == OP_SETPARAM) written[(size_t)o.node] = true; if (o.code == OP_OSCSET) { retuned[(size_t)o.node] = true; if (o.imm == 3u) dutied[(size_t)o.node] = true; } }
	const bool fx = g.channels > 0;
	static_assert((int)GRAPH_MAX_CTL == (int)KLG_MAX_CTL, "one control count");
	int ctlvar[KLG_MAX_CTL]; for (int& v : ctlvar) v = -1;                    // control index -> the ctlvar node holding the instance's own copy (controls the effect writes)
	for (const Op& o : g.ops) if (o.code == OP_SETCTL) ctlvar[o.imm & 0xFFu] = o.node;
	auto fbits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
	std::vector<long long> ring_off(g.nodes.size(), 0); std::vector<int> inputs(g.nodes.size(), 0);   // Delay nodes: first row in the group's ring tile, inputs per sample
	{ long long rows = 0; for (size_t i = 0; i < g.nodes.size(); i++) if (g.nodes[i] == N_DELAY || g.nodes[i] == N_NDELAY) { ring_off[i] = rows; rows += g.arg((int)i) + 1; } }   // SIZE + 1: the pad element (klg_delay.hpp)
	for (const Op& o : g.ops) if (o.code == OP_DELAYIN) inputs[(size_t)o.node]++;
	// effects: position-major rows of 64 instances (all instances share the cursor: one coalesced row per access).  notes: each voice's
	// line is contiguous — voices start at different times and have different lengths, so their cursors never line up; a lane walking its
	// own line re-uses each 64-byte sector for 16 samples (measured 8x over the position-major layout, tools/pluck_bench.py)
	// Effects: position-major rows of `ring_row` instances.  64 (a wave of klg_fx_graph<P>) unless the body has a staged form (klg_graph_staged.hpp): then a row is as
	// wide as the staged kernel's workgroup (16 / 32 / 64 instances) — a workgroup owns its rows, and a chunk's 32 positions of a line are 32 consecutive rows:
	// contiguous memory.  (With 64-wide rows four 16-instance workgroups share every row and each fetches the 128-byte lines around its 64 bytes: the recorded
	// Reverb.k at 4,096 instances moved 2.28 GB per launch for 0.45 GB of algorithmic bytes.  Every instance's lines contiguous — the notes' layout — was
	// measured too: fewer bytes, but 288 distant regions per workgroup instead of 18, and 0.82 ms instead of 0.63.)  `ring_row` is decided below, before any op is emitted.
	int ring_row = 64;
	auto ring = [&](int node) { return fx ? fmt("Ring{ c.ring + (size_t)%lldll * %d, %d, %d }", ring_off[(size_t)node], ring_row, ring_row, g.arg(node)) : fmt("Ring{ c.ring + (size_t)%lldll, 1, %d }", ring_off[(size_t)node], g.arg(node)); };
	uint64_t mask[graph::MAX_WORDS / 64] = { 1ull };                      // word 0 (flags) is always written back
	auto mark = [&](int w, int n) { for (int i = w; i < w + n; i++) mask[i >> 6] |= 1ull << (i & 63); };
	std::string live = fx ? "\tstruct Live { int unused_; int sidx;" : "\tstruct Live { " + TI + " stage; float tinc;" + (g.noise_calls() ? " int sidx;" : ""), begin, end, body;
	const int noise_calls = g.noise_calls();
	std::vector<int> noise_index(g.ops.size(), 0);                          // OP_NOISE: which of the sample's draws (program order)
	{ int nk = 0; for (size_t oi = 0; oi < g.ops.size(); oi++) if (g.ops[oi].code == OP_NOISE) noise_index[oi] = nk++; }
	std::vector<std::string> node_begin(g.nodes.size()), node_end(g.nodes.size());   // each node's share of begin() / end() (the staged effect kernel loads and commits nodes one by one)
	std::string ctl_begin;
	for (size_t i = 0; i < g.nodes.size(); i++) {
		const int k = g.nodes[i], w0 = g.node_word0((int)i);
		const size_t begin_at = begin.size(), end_at = end.size();
		struct Slice { std::string& from; size_t at; std::string& to; ~Slice() { to = from.substr(at); } } slice_b = { begin, begin_at, node_begin[i] }, slice_e = { end, end_at, node_end[i] };
		const std::string n = fmt("L.n%zu", i);
		auto R = [&](int off) { return fmt("r.w[%d]", w0 + off); };
		auto F = [&](int off) { return fmt("u2f(r.w[%d])", w0 + off); };
		auto W = [&](int off, const std::string& expr) { return fmt("\t\tr.w[%d] = ", w0 + off) + expr + ";\n"; };
		switch (k) {
		case N_FSINE:
			live += " FSine" + T2 + fmt(" n%zu; ", i) + TF + fmt(" n%zuf;", i);
			begin += "\t\t" + n + ".inc = to_i(" + R(FSINE_INC) + "); " + n + ".pos = " + R(FSINE_POS) + "; " + n + "f = " + F(FSINE_FREQ) + ";\n";
			end += W(FSINE_POS, n + ".pos");
			mark(w0 + FSINE_POS, 1);
			if (retuned[i]) { end += W(FSINE_INC, "to_u(" + n + ".inc)") + W(FSINE_FREQ, "f2u(" + n + "f)"); mark(w0 + FSINE_INC, 1); mark(w0 + FSINE_FREQ, 1); }
			break;
		case N_SAW: case N_PULSE:
			live += " Osm" + T2 + fmt(" n%zu; ", i) + TF + fmt(" n%zuf; bool n%zud0;", i, i);
			begin += "\t\t" + n + ".inc = to_i(" + R(OSM_INC) + "); " + n + ".offset = " + R(OSM_OFFSET) + "; " + n + ".duty = " + R(OSM_DUTY) + "; " + n + ".delta = " + F(OSM_DELTA) + "; "
				+ n + ".state = to_i(" + R(OSM_STATE) + " & 3u); " + n + "f = " + F(OSM_FREQ) + "; osm_derive(" + n + "); " + n + "d0 = osm_is_duty0(" + n + ");\n";
			end += W(OSM_OFFSET, n + ".offset") + W(OSM_STATE, "to_u(" + n + ".state)");
			mark(w0 + OSM_OFFSET, 1); mark(w0 + OSM_STATE, 1);
			if (retuned[i]) { end += W(OSM_INC, "to_u(" + n + ".inc)") + W(OSM_DELTA, "f2u(" + n + ".delta)") + W(OSM_FREQ, "f2u(" + n + "f)"); mark(w0 + OSM_INC, 1); mark(w0 + OSM_DELTA, 1); mark(w0 + OSM_FREQ, 1); }
			if (dutied[i]) { end += W(OSM_DUTY, n + ".duty"); mark(w0 + OSM_DUTY, 1); }
			break;
		case N_LPF:
			live += " Biquad" + T2 + fmt(" n%zu;", i) + " BiquadSweep" + T2 + fmt(" n%zus;", i);
			begin += "\t\t" + n + ".b0 = " + F(LPF_B0) + "; " + n + ".b1 = " + F(LPF_B1) + "; " + n + ".b2 = " + F(LPF_B2) + "; " + n + ".a1 = " + F(LPF_A1) + "; " + n + ".a2 = " + F(LPF_A2) + "; "
				+ n + ".z0 = " + F(LPF_Z0) + "; " + n + ".z1 = " + F(LPF_Z1) + "; " + n + "s.f = " + F(LPF_F) + "; " + n + "s.Q = " + F(LPF_Q) + ";\n";
			end += W(LPF_Z0, "f2u(" + n + ".z0)") + W(LPF_Z1, "f2u(" + n + ".z1)");
			mark(w0 + LPF_Z0, 2);
			if (swept[i]) {
				end += W(LPF_B0, "f2u(" + n + ".b0)") + W(LPF_B1, "f2u(" + n + ".b1)") + W(LPF_B2, "f2u(" + n + ".b2)") + W(LPF_A1, "f2u(" + n + ".a1)") + W(LPF_A2, "f2u(" + n + ".a2)")
					+ W(LPF_F, "f2u(" + n + "s.f)") + W(LPF_Q, "f2u(" + n + "s.Q)");
				mark(w0, LPF_WORDS);
			}
			break;
		case N_ENV: {
			// four point slots in registers; a node with more (its argument) reads the others from the voice's record when a segment ends (PtsN / PtsNx2, klg_device.hpp)
			const int cap = env_capacity(g.arg((int)i)); const bool far = cap > 4;
			live += (x2 ? std::string(" Env2") : std::string(" Env")) + fmt(" n%zu; ", i) + (x2 ? (far ? "PtsNx2" : "Pts4x2") : (far ? "PtsN" : "Pts4")) + fmt(" n%zup; ", i) + TI + fmt(" n%zunp, n%zuls, n%zule; ", i, i, i) + TF + fmt(" n%zuhy;", i);
			if (!x2) live += fmt(" float n%zugs, n%zugt;", i, i);                     // event-free chunks: this envelope's step and time step (quiet())
			const std::string h = far ? n + "p.head" : n + "p";
			begin += "\t\t" + n + ".r_out = " + F(ENV_OUT) + "; " + n + ".r_target = " + F(ENV_TARGET) + "; " + n + ".r_rate = " + F(ENV_RATE) + "; " + n + ".time = " + F(ENV_TIME) + "; env_unpack_rt(" + n + ", " + n + "np, " + R(ENV_BITS) + ", " + R(ENV_NPOINTS) + "); "
				+ n + "ls = loop_index(" + R(ENV_LOOP) + " & 0xFFu); " + n + "le = loop_index((" + R(ENV_LOOP) + " >> 8) & 0xFFu);\n";
			begin += "\t\t" + h + ".x0 = " + F(ENV_PX) + "; " + h + ".x1 = " + F(ENV_PX + 1) + "; " + h + ".x2 = " + F(ENV_PX + 2) + "; " + h + ".x3 = " + F(ENV_PX + 3) + "; "
				+ h + ".y0 = " + F(ENV_PY) + "; " + h + ".y1 = " + F(ENV_PY + 1) + "; " + h + ".y2 = " + F(ENV_PY + 2) + "; " + h + ".y3 = " + F(ENV_PY + 3) + ";\n";
			if (far) begin += "\t\t" + n + fmt("p.ext = c.rec + (size_t)%d * c.stride; ", w0 + ENV_WORDS) + n + "p.stride = c.stride; " + n + fmt("p.slots = %d;\n", cap - 4);
			begin += "\t\t" + n + "hy = env_hold_y(" + n + "p, " + n + "ls);\n";
			end += W(ENV_OUT, "f2u(" + n + ".r_out)") + W(ENV_TARGET, "f2u(" + n + ".r_target)") + W(ENV_RATE, "f2u(" + n + ".r_rate)") + W(ENV_TIME, "f2u(" + n + ".time)") + W(ENV_BITS, "env_pack_rt(" + n + ", " + n + "np)");
			mark(w0 + ENV_OUT, 5);
		} break;
		case N_ADSR:
			live += " Adsr" + T2 + fmt(" n%zu;", i);
			if (!x2) live += fmt(" float n%zugs, n%zugt;", i, i);
			begin += "\t\t" + n + ".e.r_out = " + F(ADSR_OUT) + "; " + n + ".e.r_target = " + F(ADSR_TARGET) + "; " + n + ".e.r_rate = " + F(ADSR_RATE) + "; " + n + ".e.time = " + F(ADSR_TIME) + "; env_unpack(" + n + ".e, " + R(ADSR_BITS) + ");\n";
			begin += "\t\tadsr_set_points(" + n + ", " + F(ADSR_A) + ", " + F(ADSR_AD) + ", " + F(ADSR_S) + ", " + F(ADSR_R) + "); adsr_derive(" + n + ", c.fs);\n";
			end += W(ADSR_OUT, "f2u(" + n + ".e.r_out)") + W(ADSR_TARGET, "f2u(" + n + ".e.r_target)") + W(ADSR_RATE, "f2u(" + n + ".e.r_rate)") + W(ADSR_TIME, "f2u(" + n + ".e.time)") + W(ADSR_BITS, "env_pack(" + n + ".e)");
			mark(w0 + ADSR_OUT, 5);
			break;
		case N_BSINE: case N_BSAW: case N_BTRI: case N_BSQUARE: case N_BPULSE:
			live += fmt(" BOsc n%zu; float n%zud, n%zuf;", i, i, i);
			begin += "\t\t" + n + ".increment = " + F(BOSC_INC) + "; " + n + ".position = " + F(BOSC_POS) + "; " + n + ".offset = " + F(BOSC_OFFSET) + "; " + n + "d = " + F(BOSC_DUTY) + "; " + n + "f = " + F(BOSC_FREQ) + ";\n";
			end += W(BOSC_POS, "f2u(" + n + ".position)");
			mark(w0 + BOSC_POS, 1);
			if (retuned[i]) { end += W(BOSC_INC, "f2u(" + n + ".increment)") + W(BOSC_FREQ, "f2u(" + n + "f)"); mark(w0 + BOSC_INC, 1); mark(w0 + BOSC_FREQ, 1); }
			if (dutied[i]) { end += W(BOSC_DUTY, "f2u(" + n + "d)"); mark(w0 + BOSC_DUTY, 1); }
			break;
		case N_OPLPF: case N_OPHPF:
			live += fmt(" OnePole n%zu;", i);
			begin += "\t\t" + n + ".b0 = " + F(OP1_B0) + "; " + n + ".b1 = " + F(OP1_B1) + "; " + n + ".a1 = " + F(OP1_A1) + "; " + n + ".z = " + F(OP1_Z) + "; " + n + ".out = " + F(OP1_OUT) + ";\n";
			end += W(OP1_Z, "f2u(" + n + ".z)") + W(OP1_OUT, "f2u(" + n + ".out)");
			mark(w0 + OP1_Z, 2);
			break;
		case N_DCF:
			live += fmt(" Dcf n%zu;", i);
			begin += "\t\t" + n + ".r = " + F(DCF_R) + "; " + n + ".z = " + F(DCF_Z) + "; " + n + ".out = " + F(DCF_OUT) + ";\n";
			end += W(DCF_Z, "f2u(" + n + ".z)") + W(DCF_OUT, "f2u(" + n + ".out)");
			mark(w0 + DCF_Z, 2);
			break;
		case N_IIR1:
			live += fmt(" Iir1 n%zu;", i);
			begin += "\t\t" + n + ".a = " + F(IIR1_A) + "; " + n + ".b = " + F(IIR1_B) + "; " + n + ".out = " + F(IIR1_OUT) + ";\n";
			end += W(IIR1_OUT, "f2u(" + n + ".out)");
			mark(w0 + IIR1_OUT, 1);
			break;
		case N_IIRN: {
			const int order = g.arg((int)i);
			live += fmt(" Iir<%d> n%zu;", order, i);
			begin += "\t\t";
			for (int q = 0; q < order; q++) begin += n + fmt(".a[%d] = ", q) + F(q) + "; " + n + fmt(".y[%d] = ", q) + F(order + q) + "; ";
			begin += "\n";
			for (int q = 0; q < order; q++) end += W(order + q, "f2u(" + n + fmt(".y[%d])", q));
			mark(w0 + order, order);
		} break;
		case N_BUTTER1:
			live += fmt(" Butter1 n%zu;", i);
			begin += "\t\t" + n + ".b0 = " + F(BW1_B0) + "; " + n + ".a1 = " + F(BW1_A1) + "; " + n + ".z = " + F(BW1_Z) + "; " + n + ".out = " + F(BW1_OUT) + ";\n";
			end += W(BW1_Z, "f2u(" + n + ".z)") + W(BW1_OUT, "f2u(" + n + ".out)");
			mark(w0 + BW1_Z, 2);
			break;
		case N_MODAL:
			live += fmt(" Modal n%zu;", i);
			begin += "\t\t" + n + ".a1 = " + F(MODAL_A1) + "; " + n + ".a2 = " + F(MODAL_A2) + "; " + n + ".y1 = " + F(MODAL_Y1) + "; " + n + ".y2 = " + F(MODAL_Y2) + "; " + n + ".gain = " + F(MODAL_GAIN) + ";\n";
			end += W(MODAL_Y1, "f2u(" + n + ".y1)") + W(MODAL_Y2, "f2u(" + n + ".y2)");
			mark(w0 + MODAL_Y1, 2);
			break;
		case N_FOLLOWPEAK: case N_FOLLOWRMS:
			live += fmt(" FollowerAR n%zu;", i);
			begin += "\t\t" + n + ".A = " + F(FOLLOW_A) + "; " + n + ".R = " + F(FOLLOW_R) + "; " + n + ".out = " + F(FOLLOW_OUT) + ";\n";
			end += W(FOLLOW_OUT, "f2u(" + n + ".out)");
			mark(w0 + FOLLOW_OUT, 1);
			break;
		case N_OPERATOR: {
			const int cap = env_capacity(g.arg((int)i)); const bool far = cap > 4;
			live += fmt(" FSine n%zu; float n%zua, n%zuf; Env n%zue; %s n%zup; int n%zunp, n%zuls, n%zule; float n%zuhy, n%zugs, n%zugt;", i, i, i, i, far ? "PtsN" : "Pts4", i, i, i, i, i, i, i);
			const int e0 = OPER_ENV;
			const std::string h = far ? n + "p.head" : n + "p";
			begin += "\t\t" + n + ".inc = (int32_t)" + R(OPER_INC) + "; " + n + ".pos = " + R(OPER_POS) + "; " + n + "a = " + F(OPER_AMP) + "; " + n + "f = " + F(OPER_FREQ) + ";\n";
			begin += "\t\t" + n + "e.r_out = " + F(e0 + ENV_OUT) + "; " + n + "e.r_target = " + F(e0 + ENV_TARGET) + "; " + n + "e.r_rate = " + F(e0 + ENV_RATE) + "; " + n + "e.time = " + F(e0 + ENV_TIME) + "; env_unpack_rt(" + n + "e, " + n + "np, " + R(e0 + ENV_BITS) + ", " + R(e0 + ENV_NPOINTS) + "); "
				+ n + "ls = (int)(" + R(e0 + ENV_LOOP) + " & 0xFFu); " + n + "le = (int)((" + R(e0 + ENV_LOOP) + " >> 8) & 0xFFu); "
				+ n + "ls = " + n + "ls == 255 ? -1 : " + n + "ls; " + n + "le = " + n + "le == 255 ? -1 : " + n + "le;\n";
			begin += "\t\t" + h + ".x0 = " + F(e0 + ENV_PX) + "; " + h + ".x1 = " + F(e0 + ENV_PX + 1) + "; " + h + ".x2 = " + F(e0 + ENV_PX + 2) + "; " + h + ".x3 = " + F(e0 + ENV_PX + 3) + "; "
				+ h + ".y0 = " + F(e0 + ENV_PY) + "; " + h + ".y1 = " + F(e0 + ENV_PY + 1) + "; " + h + ".y2 = " + F(e0 + ENV_PY + 2) + "; " + h + ".y3 = " + F(e0 + ENV_PY + 3) + ";\n";
			if (far) begin += "\t\t" + n + fmt("p.ext = c.rec + (size_t)%d * c.stride; ", w0 + e0 + ENV_WORDS) + n + "p.stride = c.stride; " + n + fmt("p.slots = %d;\n", cap - 4);
			begin += "\t\t" + n + "hy = env_hold_y(" + n + "p, " + n + "ls);\n";
			end += W(OPER_POS, n + ".pos") + W(OPER_AMP, "f2u(" + n + "a)")
				+ W(e0 + ENV_OUT, "f2u(" + n + "e.r_out)") + W(e0 + ENV_TARGET, "f2u(" + n + "e.r_target)") + W(e0 + ENV_RATE, "f2u(" + n + "e.r_rate)") + W(e0 + ENV_TIME, "f2u(" + n + "e.time)") + W(e0 + ENV_BITS, "env_pack_rt(" + n + "e, " + n + "np)");
			mark(w0 + OPER_POS, 1); mark(w0 + OPER_AMP, 1); mark(w0 + e0 + ENV_OUT, 5);
		} break;
		case N_WAVETABLE:
			live += fmt(" WTab n%zu; float n%zuf;", i, i);
			begin += "\t\t" + n + ".inc = " + F(WT_INC) + "; " + n + ".pos = " + F(WT_POS) + "; " + n + ".off = " + F(WT_OFFSET) + "; " + n + "f = " + F(WT_FREQ) + "; wavetable_load(" + n + ", c.tables, " + R(WT_TABLE) + ");\n";
			end += W(WT_POS, "f2u(" + n + ".pos)");
			mark(w0 + WT_POS, 1);
			if (retuned[i]) { end += W(WT_INC, "f2u(" + n + ".inc)") + W(WT_FREQ, "f2u(" + n + "f)"); mark(w0 + WT_INC, 1); mark(w0 + WT_FREQ, 1); }
			break;
		case N_NDELAY:                                                   // a note's delay line: its own write cursor and read head (klang.h:3386-3388, 3475-3478)
			live += fmt(" int n%zupos; Tap n%zut; float n%zutime;", i, i, i);
			begin += "\t\t" + n + "pos = (int)" + R(ND_POS) + "; " + n + "t.position = (int)" + R(ND_LASTPOS) + "; " + n + "t.fraction = " + F(ND_LASTFRAC) + "; " + n + "time = " + F(ND_TIME) + ";\n";
			end += W(ND_POS, "(uint32_t)" + n + "pos") + W(ND_LASTPOS, "(uint32_t)" + n + "t.position");
			mark(w0 + ND_POS, 2);
			if (reset_head[i]) { end += W(ND_LASTFRAC, "f2u(" + n + "t.fraction)"); mark(w0 + ND_LASTFRAC, 1); }   // set() inside process()
			break;
		case N_DELAY:
			live += fmt(" int n%zupos; Tap n%zut;", i, i);                   // (t: the read head of set() / process(), kept in the record)
			begin += "\t\t" + n + fmt("pos = (int)((c.samples * %dull) %% %dull); ", inputs[i], g.arg((int)i)) + n + "t.position = (int)" + R(ED_LASTPOS) + "; " + n + "t.fraction = " + F(ED_LASTFRAC) + ";\n";     // Delay::position: one step per input()
			if (head_used[i]) { end += W(ED_LASTPOS, "(uint32_t)" + n + "t.position") + W(ED_LASTFRAC, "f2u(" + n + "t.fraction)"); mark(w0 + ED_LASTPOS, 2); }
			break;
		case N_CTLVAR:
		case N_SMOOTH:
			live += fmt(" float n%zu;", i);
			begin += "\t\t" + n + " = " + F(0) + ";\n";
			end += W(0, "f2u(" + n + ")"); mark(w0, 1);
			break;
		case N_PARAM:
			live += " " + TF + fmt(" n%zu;", i);
			begin += "\t\t" + n + " = " + F(0) + ";\n";
			if (written[i]) { end += W(0, "f2u(" + n + ")"); mark(w0, 1); }
			break;
		}
	}
	if (fx) {                                                         // an effect's dials are the block's (klg_fx_set_control uploads them before the launch): read once, not in every sample
		bool used[KLG_MAX_CTL] = {};
		for (const Op& o : g.ops) if ((o.code == OP_CTL || o.code == OP_SMOOTH) && ctlvar[o.imm & 0xFFu] < 0 && o.imm < (unsigned)KLG_MAX_CTL) used[o.imm] = true;
		for (unsigned i = 0; i < (unsigned)KLG_MAX_CTL; i++) if (used[i]) { live += fmt(" float ctl%u;", i); ctl_begin += fmt("\t\tL.ctl%u = c.ctl[%u];\n", i, i); }
		begin += ctl_begin;
	}
	live += " };\n";
	std::map<int, uint32_t> const_of;                                    // single-assignment registers holding a literal
	for (const Op& o : g.ops) if (o.code == OP_CONST) const_of[o.dst] = o.imm;
	int if_depth = 0; std::vector<std::string> stop_at_end;
	// structured branches: the phis that follow an `endif` are assigned at the end of each side of their `if`
	std::vector<int> match_else(g.ops.size(), -1), match_endif(g.ops.size(), -1), if_of(g.ops.size(), -1);
	{
		std::vector<int> st;
		for (size_t oi = 0; oi < g.ops.size(); oi++) {
			const int c = g.ops[oi].code;
			if (c == OP_IF) st.push_back((int)oi);
			else if (c == OP_ELSE && !st.empty()) { match_else[(size_t)st.back()] = (int)oi; if_of[oi] = st.back(); }
			else if (c == OP_ENDIF && !st.empty()) { match_endif[(size_t)st.back()] = (int)oi; if_of[oi] = st.back(); st.pop_back(); }
		}
	}
	auto phis_of = [&](int if_index) { std::vector<const Op*> v; const int e = match_endif[(size_t)if_index]; for (size_t q = (size_t)e + 1; e >= 0 && q < g.ops.size() && g.ops[q].code == OP_PHI; q++) v.push_back(&g.ops[q]); return v; };
	// ---- effects: taps requested where their rows are first known (klg_delay.hpp "HOISTED TAPS") ----
	struct Hoist { int tap, at, m, k; std::vector<int> writes; };            // tap op, the op it is requested in front of, process() calls / input()s of the line in between
	std::vector<Hoist> hoists; std::vector<int> hoist_of(g.ops.size(), -1);
	std::vector<bool> names_row(g.ops.size(), false);                        // delayin ops whose row a later tap compares with
	{
		const char* he = getenv("KLG_FX_HOIST");
		const int first = g.prepare_ops > 0 ? g.prepare_ops : 0;
		std::vector<int> region(g.ops.size(), 0), region_start(1, first), stack(1, 0);
		for (size_t oi = 0; oi < g.ops.size(); oi++) {                           // region of an op: the branch side it stands in (0: the top level)
			const int c = g.ops[oi].code;
			if (c == OP_IF) { region[oi] = stack.back(); stack.push_back((int)region_start.size()); region_start.push_back((int)oi + 1); }
			else if (c == OP_ELSE && stack.size() > 1) { stack.pop_back(); region[oi] = stack.back(); stack.push_back((int)region_start.size()); region_start.push_back((int)oi + 1); }
			else if (c == OP_ENDIF && stack.size() > 1) { stack.pop_back(); region[oi] = stack.back(); }
			else region[oi] = stack.back();
		}
		std::vector<int> def_at(2 * g.ops.size() + 64, -1);
		for (size_t oi = 0; oi < g.ops.size(); oi++) if (g.ops[oi].dst >= 0 && (size_t)g.ops[oi].dst < def_at.size()) def_at[(size_t)g.ops[oi].dst] = (int)oi;
		if (fx && !(he && he[0] == '0')) for (size_t t = (size_t)first; t < g.ops.size(); t++) {
			const Op& o = g.ops[t];
			const bool out = o.code == OP_DELAYOUT, tapf = o.code == OP_DELAYTAP && (o.imm == 0u || o.imm == 2u);
			if (!out && !tapf) continue;
			if (o.node < 0 || g.nodes[(size_t)o.node] != N_DELAY) continue;
			const int R = region[t];
			int at = std::max(region_start[(size_t)R], first);
			bool ok = true;
			if (tapf) { const int dd = (o.a >= 0 && (size_t)o.a < def_at.size()) ? def_at[(size_t)o.a] : -1; if (dd >= (int)t) ok = false; else if (dd >= at) { if (region[(size_t)dd] != R) ok = false; at = dd + 1; if (g.ops[(size_t)dd].code == OP_PHI) { while (at < (int)t && g.ops[(size_t)at].code == OP_PHI) at++; } } }
			if (out) for (int q = (int)t - 1; q >= at; q--) if (g.ops[(size_t)q].code == OP_DELAYSET && g.ops[(size_t)q].node == o.node) { if (region[(size_t)q] != R) ok = false; at = q + 1; break; }
			Hoist h; h.tap = (int)t; h.at = at; h.m = 0; h.k = 0;
			for (int q = at; ok && q < (int)t; q++) {
				const Op& x = g.ops[(size_t)q];
				if (x.node != o.node) continue;
				if (x.code != OP_DELAYIN && x.code != OP_DELAYOUT && x.code != OP_DELAYSET) continue;
				if (region[(size_t)q] != R) { ok = false; break; }                     // (a conditional input() / process() / set() of the line in between: not predicted)
				if (x.code == OP_DELAYIN) { h.k++; h.writes.push_back(q); }
				else if (x.code == OP_DELAYOUT) h.m++;
				else ok = false;
			}
			if (!ok || (int)t - at < 2) continue;                                  // (nothing to request ahead of)
			for (int w : h.writes) names_row[(size_t)w] = true;
			hoist_of[t] = (int)hoists.size(); hoists.push_back(h);
		}
		// A tap is worth requesting early when something it would otherwise stand behind WAITS for another tap: an input() in between whose value comes from a
		// tap of this sample (PingPong.k's cross-feed; a feedback matrix).  Where no such input() lies in between the round trip is paid once wherever the request
		// stands, and requesting early only costs the compiler its own pairing of samples (the echo: 0.087 -> 0.109 ms) — those taps stay where they are.
		{
			std::vector<bool> dep(def_at.size(), false), node_dep(g.nodes.size(), false), waits(g.ops.size(), false);
			for (size_t oi = (size_t)first; oi < g.ops.size(); oi++) {
				const Op& x = g.ops[oi];
				auto D = [&](int r) { return r >= 0 && (size_t)r < dep.size() && dep[(size_t)r]; };
				bool dd = D(x.a) || D(x.b) || x.code == OP_DELAYOUT || x.code == OP_DELAYTAP;
				if (x.node >= 0 && (size_t)x.node < node_dep.size()) { if (x.code == OP_SETPARAM || x.code == OP_SETCTL) node_dep[(size_t)x.node] = node_dep[(size_t)x.node] || D(x.a); else if (node_dep[(size_t)x.node]) dd = true; }
				if (x.dst >= 0 && (size_t)x.dst < dep.size()) dep[(size_t)x.dst] = dd;
				if (x.code == OP_DELAYIN && D(x.a)) waits[oi] = true;
			}
			std::vector<Hoist> kept; std::fill(hoist_of.begin(), hoist_of.end(), -1); std::fill(names_row.begin(), names_row.end(), false);
			for (const Hoist& h : hoists) {
				bool behind = false;
				for (int q = h.at; q < h.tap; q++) if (waits[(size_t)q]) behind = true;
				if (!behind) continue;
				for (int w : h.writes) names_row[(size_t)w] = true;
				hoist_of[(size_t)h.tap] = (int)kept.size(); kept.push_back(h);
			}
			hoists.swap(kept);
		}
	}
	auto emit_hoists = [&](std::string& body, int at) {                       // the requests that stand in front of op `at`
		for (size_t q = 0; q < hoists.size(); q++) if (hoists[q].at == at) {
			const Hoist& h = hoists[q]; const Op& o = g.ops[(size_t)h.tap];
			const std::string n = fmt("L.n%d", o.node), rg = ring(o.node);
			const int SZ = g.arg(o.node);
			if (o.code == OP_DELAYOUT) {
				body += fmt("\t\tRows2 h%zu; h%zu.i = ", q, q) + n + "t.position;";
				for (int i = 0; i < h.m; i++) body += fmt(" h%zu.i = ring_succ(h%zu.i, %d);", q, q, SZ);
				body += fmt(" h%zu.j = ring_succ(h%zu.i, %d);\n", q, q, SZ);
			}
			else {
				const std::string pos = h.k ? "((" + n + fmt("pos + %d) %% %d)", h.k, SZ) : n + "pos";
				body += fmt("\t\tconst Rows2 h%zu = ", q) + (o.imm == 2u ? "delay_tap_stereo_rows(" : "delay_tap_float_rows(") + fmt("%d, ", SZ) + pos + fmt(", r%d);\n", o.a);
			}
			body += fmt("\t\tconst float h%zua = ", q) + rg + fmt(".rd(h%zu.i), h%zub = ", q, q) + rg + fmt(".rd(h%zu.j);\n", q);
		}
	};
	auto hazard_of = [&](int q) {
		std::string e = "false";
		for (int w : hoists[(size_t)q].writes) e += fmt(" || h%d.i == wr%d || h%d.j == wr%d", q, w, q, w);
		return e;
	};
	std::string prologue;                                                // Effect::prepare(): once per block, at the end of begin()
	// one op's statement(s) appended to `body`.  assign: `r<dst> = ...` into a register declared elsewhere instead of `const float r<dst> = ...`
	// (the staged effect kernel declares registers that live across its phases, or inside a branch, up front)
	auto emit_op = [&](size_t oi, std::string& body, bool assign) {
		const Op& o = g.ops[oi];
		const std::string d = assign ? fmt("\t\tr%d = ", o.dst) : "\t\tconst " + TF + fmt(" r%d = ", o.dst), n = fmt("L.n%d", o.node), a = fmt("r%d", o.a), b = fmt("r%d", o.b);
		const std::string dd = assign ? fmt("\t\tr%d = ", o.dst) : fmt("\t\tconst double r%d = ", o.dst);
		const int k = (o.node >= 0 && o.node < (int)g.nodes.size()) ? g.nodes[(size_t)o.node] : -1;
		switch (o.code) {
		case OP_CONST: body += d + "kf<" + TF + fmt(">(0x%08xu);\n", o.imm); break;
		case OP_CTL: body += d + (ctlvar[o.imm & 0xFFu] >= 0 ? fmt("L.n%d;\n", ctlvar[o.imm & 0xFFu]) : fx ? fmt("L.ctl%u;\n", o.imm) : fmt("ctl_read(c, %uu);\n", o.imm)); break;   // (a control the effect writes: its own copy)
		case OP_SETCTL: {                                                  // Control::set klang.h:1725-1728: (x < min) ? min : (max < x) ? max : x — as two selects (min < max: at most one of the tests holds; a NaN passes both)
			if (o.imm & 0x100u) { body += d + a + ";\n\t\t" + n + fmt(" = r%d;\n", o.dst); break; }   // Control::operator<< klang.h:1745-1746: the plain assignment (a meter)
			const uint32_t mn = fbits(g.dials[o.imm & 0xFFu].min), mx = fbits(g.dials[o.imm & 0xFFu].max);
			body += fmt("\t\tconst float r%dh = (u2f(0x%08xu) < ", o.dst, mx) + a + fmt(") ? u2f(0x%08xu) : ", mx) + a + ";\n";
			body += d + "(" + a + fmt(" < u2f(0x%08xu)) ? u2f(0x%08xu) : r%dh;\n\t\t", mn, mn, o.dst) + n + fmt(" = r%d;\n", o.dst);
		} break;
		case OP_ABS: body += d + "__builtin_fabsf(" + a + ");\n"; break;
		case OP_TRUNC: body += d + "(__builtin_fabsf(" + a + ") < 2147483648.f) ? __builtin_truncf(" + a + ") + 0.f : -2147483648.f;\n"; break;   // (float)(int)x as cvttss2si has it (+ 0.f: an int has no negative zero)
		case OP_POWC: {                                                    // klang.h:188-218 with a literal float exponent: the test for base 10 first, then the written-out products
			float e; memcpy(&e, &o.imm, 4);
			const float ten = (float)exp((double)(e * 2.3025850929940456840179914546843642076011014886287729760333279009f));   // (the C library's exp, here on the host: the value the reference computes at run time)
			const int m = (int)(e < 0 ? -e : e);
			std::string p = m == 0 ? "1.f" : a; for (int q = 1; q < m; q++) p += " * " + a;
			if (m > 1) p = "(" + p + ")";
			if (e < 0) p = "1.f / " + p;
			body += d + "(" + a + fmt(" == 10.f) ? u2f(0x%08xu) : ", fbits(ten)) + p + ";\n";
		} break;
		case OP_PARAM: body += d + n + ";\n"; break;
		case OP_OSC: {
			std::string e;
			switch (k) {
			case N_FSINE: e = "fsine_process(" + n + ", 0u)"; break;
			case N_SAW: e = retuned[(size_t)o.node] ? "osm_saw(" + n + ")" : "(" + n + "d0 ? osm_saw_duty0(" + n + ") : osm_saw(" + n + "))"; break;   // d0: decided once per block (begin)
			case N_PULSE: e = "osm_pulse(" + n + ")"; break;
			case N_BSINE: e = "basic_sine(" + n + ")"; break;
			case N_BSAW: e = "basic_saw(" + n + ")"; break;
			case N_BTRI: e = "basic_triangle(" + n + ")"; break;
			case N_BSQUARE: e = "basic_square(" + n + ")"; break;
			case N_BPULSE: e = "basic_pulse(" + n + ", " + n + "d)"; break;
			case N_WAVETABLE: e = "wavetable_process(" + n + ")"; break;
			}
			body += d + e + ";\n";
		} break;
		case OP_OSCSET:
			if (o.imm == 3u) body += k == N_BPULSE ? "\t\t" + n + "d = " + a + ";\n" : "\t\tosm_set_duty(" + n + ", " + a + ");\n";   // Basic::Pulse::duty 4936 / OSM::setDuty 5246-5249
			else if (o.imm == 2u) body += "\t\t" + n + (k == N_FSINE ? ".pos = 0u;\n" : ".position = 0.f;\n");                               // reset(): Fast::Sine 5136-5140 (= set(frequency, 0)), Oscillator 2859
			else if (o.imm == 1u) {                                                                                                    // set(f, phase)
				if (k >= N_BSINE && k <= N_BPULSE) body += "\t\t" + n + ".position = " + b + "; " + n + "f = " + a + "; " + n + ".increment = " + a + " * 2.f * KLG_PI_F / c.fs.f;\n";   // klang.h:2867-2870
				else body += "\t\t" + std::string(k == N_FSINE ? "fsine_set_fp(" : "osm_set_fp(") + n + ", " + n + "f, " + a + ", " + b + ", c.fs.f);\n";
			}
			else if (k == N_WAVETABLE) body += "\t\twavetable_set_f(" + n + ", " + n + "f, " + a + ", c.fs.f);\n";
			else if (k >= N_BSINE && k <= N_BPULSE) body += "\t\t" + n + "f = " + a + "; " + n + ".increment = " + a + " * 2.f * KLG_PI_F / c.fs.f;\n";   // Oscillator::set(f) klang.h:2862-2865
			else body += "\t\t" + std::string(k == N_FSINE ? "fsine_set_f(" : "osm_set_f(") + n + ", " + n + "f, " + a + ", c.fs.f);\n";
			break;
		case OP_LPF: {
			const char* fn = k == N_LPF ? "biquad_process" : k == N_OPLPF ? "onepole_lpf_process" : k == N_OPHPF ? "onepole_process" : k == N_DCF ? "dcf_process" : k == N_IIR1 ? "iir1_process" : k == N_IIRN ? "iir_process"
				: k == N_BUTTER1 ? "butter1_process" : k == N_MODAL ? "modal_process" : k == N_FOLLOWPEAK ? "follower_peak" : "follower_rms";
			body += d + fn + "(" + n + ", " + a + ");\n";
		} break;
		case OP_LPFSET: body += (o.imm == 0 ? "\t\tbiquad_lpf_set(" : fmt("\t\tbiquad_set<%u>(", o.imm)) + n + ", " + n + "s, " + a + ", " + b + ", c.fs.w);\n"; break;
		case OP_ENV: body += d + (k == N_ADSR ? "adsr_process(" + n + ", c.fs)" : "env_process_rt(" + n + ", " + n + "p, " + n + "np, " + n + "ls, " + n + "le, " + n + "hy, c.fs)") + ";\n"; break;
		case OP_ADD: body += d + a + " + " + b + ";\n"; break;
		case OP_SUB: body += d + a + " - " + b + ";\n"; break;
		case OP_MUL: body += d + a + " * " + b + ";\n"; break;
		case OP_DIV: {                                              // constant divisors of the verified set: klg_device.hpp div_const
			const auto cv = const_of.find(o.b);
			const uint32_t y = cv == const_of.end() ? 0u : cv->second;
			if (y == 0x40e00000u || y == 0x40400000u || y == 0x40a00000u || y == 0x41100000u || y == 0x40200000u) body += d + fmt("div_const<0x%08xu>(", y) + a + ");\n";
			else body += d + a + " / " + b + ";\n";
		} break;
		case OP_NEG: body += d + "-" + a + ";\n"; break;
		// double registers (include/klang_mi355_graph.h): IEEE double arithmetic, contraction off like everything else
		case OP_F2D: body += dd + "(double)" + a + ";\n"; break;
		case OP_FUNC:
			if ((o.imm & 0xFFu) == 0u) body += dd + "glibc_tanh(" + a + ");\n";
			else if ((o.imm & 0xFFu) == 1u) body += dd + "klg::glibc::exp2(" + a + ");\n";
			else {                                                              // pow(B, a), B constant: log(B) = hi + lo from the host's restatement, as literals
				const uint32_t bb = o.imm & 0xFFFFFF00u; float base; memcpy(&base, &bb, 4);
				double lo; const double hi = klg::glibc::pow_log((double)base, &lo);
				uint64_t ub, uh, ul; const double db = (double)base; memcpy(&ub, &db, 8); memcpy(&uh, &hi, 8); memcpy(&ul, &lo, 8);
				body += dd + fmt("klg::glibc::pow_of_log(klg::glibc::as_f64(0x%016llxull), ", (unsigned long long)ub) + a + fmt(", klg::glibc::as_f64(0x%016llxull), klg::glibc::as_f64(0x%016llxull));\n", (unsigned long long)uh, (unsigned long long)ul);
			}
			break;
		case OP_DCONST: body += dd + fmt("__longlong_as_double(0x%08x00000000ll);\n", o.imm); break;
		case OP_DLOW: body += dd + "__longlong_as_double(__double_as_longlong(" + a + fmt(") | 0x%08xll);\n", o.imm); break;
		case OP_DADD: body += dd + a + " + " + b + ";\n"; break;
		case OP_DSUB: body += dd + a + " - " + b + ";\n"; break;
		case OP_DMUL: body += dd + a + " * " + b + ";\n"; break;
		case OP_DDIV: body += dd + a + " / " + b + ";\n"; break;
		case OP_D2F: body += d + "(float)" + a + ";\n"; break;
		case OP_ENVOFF: body += d + (o.imm == 0u ? "env_is_off(" : o.imm == 1u ? "env_is_sustain(" : "env_is_release(") + n + (k == N_ADSR ? ".e" : "") + ".stage) ? 1.f : 0.f;\n"; break;     // Envelope::finished klang.h:4094; `env == Envelope::Sustain / Release` 3883
		case OP_CMP: { static const char* rel[6] = { "<", ">", "<=", ">=", "==", "!=" }; body += d + "(" + a + " " + rel[o.imm <= 5u ? o.imm : 0u] + " " + b + ") ? 1.f : 0.f;\n"; } break;
		case OP_IF:
			if_depth++;
			for (const Op* ph : phis_of((int)oi)) body += fmt("\t\tfloat r%d;\n", ph->dst);
			body += "\t\tif (" + a + " != 0.f) {\n";
			break;
		case OP_ELSE:
			for (const Op* ph : phis_of(if_of[oi])) body += fmt("\t\tr%d = r%d;\n", ph->dst, ph->a);
			body += "\t\t} else {\n";
			break;
		case OP_ENDIF:
			if (match_else[(size_t)if_of[oi]] < 0) { for (const Op* ph : phis_of(if_of[oi])) body += fmt("\t\tr%d = r%d;\n", ph->dst, ph->a); body += "\t\t} else {\n"; }
			for (const Op* ph : phis_of(if_of[oi])) body += fmt("\t\tr%d = r%d;\n", ph->dst, ph->b);
			body += "\t\t}\n";
			break;
		case OP_NOISE:                                                    // effects: straight from the block's draws; notes: from the wave's LDS copy of this group of samples (klg_render)
			if (fx) body += d + (o.imm ? "fast_noise(" : "basic_noise(") + fmt("c.rand[(size_t)(L.sidx * %d + %d) * c.rstride]);\n", noise_calls, noise_index[oi]);
			else body += d + (o.imm ? "fast_noise(" : "basic_noise(") + fmt("c.nz[((L.sidx & (KLG_NZ_GROUP - 1)) * %d + %d) * 64]);\n", noise_calls, noise_index[oi]);
			break;
		case OP_DELAYOUT:
			if (hoist_of[oi] >= 0) body += d + "delay_process_h(" + ring(o.node) + ", " + n + fmt("t, h%d, h%da, h%db, ", hoist_of[oi], hoist_of[oi], hoist_of[oi]) + hazard_of(hoist_of[oi]) + ");\n";
			else body += d + "delay_process(" + ring(o.node) + ", " + n + "t);\n";
			break;
		case OP_TABREAD: body += d + fmt("table_read(c.tables, %uu, ", o.imm) + a + ");\n"; break;
		case OP_PHI: break;                                         // assigned at the end of both sides (above)
		case OP_STOPIF: {
			// outside a branch the test may wait for the end of the block: an envelope that is Off stays Off, and the note's stage is only read there
			const std::string t = "stage_off_if(env_is_off(" + n + (k == N_ADSR ? ".e" : "") + ".stage), ";
			if (if_depth == 0 && !fx) stop_at_end.push_back(t); else body += "\t\tL.stage = " + t + "L.stage);\n";
		} break;
		case OP_STOP: body += "\t\tL.stage = stage_all_off(L.stage);\n"; break;
		case OP_SETPARAM: body += "\t\t" + n + " = " + a + ";\n"; break;
		case OP_FREQ: body += d + n + "f;\n"; break;
		case OP_IN: body += d + (o.imm ? "in1" : "in0") + ";\n"; break;
		case OP_DELAYIN:
			if (names_row[oi]) body += fmt("\t\tconst int wr%zu = ", oi) + n + "pos;\n";     // (the row this input() writes: hoisted taps of the line compare theirs with it)
			body += "\t\t{ const Ring q = " + ring(o.node) + "; q.wr(" + n + "pos, " + a + "); " + n + "pos = (" + n + fmt("pos + 1 == %d) ? 0 : ", g.arg(o.node)) + n + "pos + 1; }\n"; break;   // Delay::input klang.h:3396-3403
		case OP_DELAYSET: body += "\t\t" + n + "t = delay_set(" + n + fmt("pos, %d, ", g.arg(o.node)) + a + ");\n"; break;   // Delay::set klang.h:3480-3489
		case OP_DELAYTAP:
			if (hoist_of[oi] >= 0) { body += d + (o.imm == 2u ? "delay_tap_stereo_h(" : "delay_tap_float_h(") + ring(o.node) + ", " + n + "pos, " + a + fmt(", h%d, h%da, h%db, ", hoist_of[oi], hoist_of[oi], hoist_of[oi]) + hazard_of(hoist_of[oi]) + ");\n"; break; }
			body += d + (o.imm == 1u ? "delay_tap_int(" + ring(o.node) + ", " + n + "pos, (int)" + a + ");\n" : (o.imm == 3u ? "delay_lagrange(" : o.imm == 2u ? "delay_tap_stereo(" : "delay_tap_float(") + ring(o.node) + ", " + n + "pos, " + a + ");\n"); break;   // imm 1: tap(int), klang.h:3405-3410; 3: lagrange(float) 3429-3458
		case OP_SMOOTH: body += "\t\t" + n + " = " + n + " * 0.999f + (1.f - 0.999f) * " + (ctlvar[o.imm & 0xFFu] >= 0 ? fmt("L.n%d", ctlvar[o.imm & 0xFFu]) : fx ? fmt("L.ctl%u", o.imm) : fmt("c.ctl[%u]", o.imm)) + ";\n" + d + n + ";\n"; break;   // Control::smooth klang.h:1715
		case OP_OPERATOR:                                           // OSC::set(+in); OSC::process(); out *= env++ * amp   klang.h:4164-4168
			if (o.b >= 0) body += "\t\t" + n + "a = " + b + ";\n";
			body += d + (o.a >= 0 ? "fsine_process_rel(" + n + ", " + a + ")" : "fsine_process(" + n + ", 0u)") + " * (env_process_rt(" + n + "e, " + n + "p, " + n + "np, " + n + "ls, " + n + "le, " + n + "hy, c.fs) * " + n + "a);\n";
			break;
		}
	};
	auto staged_plan = [&](bool dry) {
		StagedPlan plan;
		const char* se = getenv("KLG_FX_STAGED"), *ge = getenv("KLG_FX_STAGED_G"), *ce = getenv("KLG_FX_STAGED_C");
		if (se && se[0] == '0') { plan.ok = false; plan.why = "KLG_FX_STAGED=0"; return plan; }
		StagedInput in;
		in.g = &g; in.node_begin = &node_begin; in.node_end = &node_end; in.ctl_begin = ctl_begin; in.ring_off = &ring_off; in.inputs = &inputs; in.ctlvar = ctlvar;
		if (dry) in.emit_op = [](size_t, std::string&, bool) {}; else in.emit_op = emit_op;
		// instances per workgroup: the caller's choice by bank size (klg_fx_create_graph), the chunk that goes with it (a wider workgroup takes a shorter chunk:
		// 512 lanes for the parallel levels either way); a plan that does not fit at one width is tried at the next narrower one
		int G = ge ? atoi(ge) : (staged_G ? staged_G : 16);
		for (;;) {
			StagedInput tryin = in; tryin.G = G;
			if (ce) tryin.C = atoi(ce); else { tryin.C = G >= 64 ? 8 : G == 32 ? 16 : 32; tryin.C_is_a_preference = true; }
			plan = plan_staged(tryin);
			// (a wide workgroup is worth it for bodies whose serial levels dominate — PingPong.k; one that only fits with a chunk shorter than its width's
			//  own — the recorded Reverb.k: 68 values through LDS — is better off narrow and long: 2.5 ms at 16 x 32 against 3.9 at 32 x 8, 16,384 instances)
			if (ge || G <= 16 || (plan.ok && plan.C >= tryin.C)) break;
			G /= 2;
		}
		return plan;
	};
	bool has_staged = false;
	if (fx && staged) { const StagedPlan dry = staged_plan(true); has_staged = dry.ok; if (dry.ok) ring_row = dry.G; }   // the ring layout of BOTH kernels of this code object follows from the staged form's width
	for (size_t oi = 0; oi < g.ops.size(); oi++) {
		if ((int)oi == g.prepare_ops && g.prepare_ops) { prologue = body; body.clear(); }
		emit_hoists(body, (int)oi);
		emit_op(oi, body, false);
	}
	if (g.prepare_ops && (int)g.ops.size() == g.prepare_ops) { prologue = body; body.clear(); }
	const std::string begin_core = begin;                                 // (effects: begin() without the per-block prologue — what a chunk walked by the plain body inside the staged kernel starts from)
	begin += prologue;
	std::string s;
	s += "// generated by klg_graph.hpp from a recorded klang process() body (include/klang_mi355_graph.h)\n";
	s += fx ? "#include \"klg_fx.hpp\"\n" : "#include \"klg_render_x2.hpp\"\n#include \"klg_delay.hpp\"\n#include \"klg_render_sp.hpp\"\n";
	s += "#pragma clang fp contract(off)\nnamespace klg {\n";
	s += "struct PatchGen {\n";
	s += "\tstruct Rec { " + TU + fmt(" w[%d]; };\n\tstatic constexpr int kWords = %d;\n", NW, NW);
	s += fmt("\tstatic constexpr uint64_t kStoreMask = 0x%llxull;\n\tstatic constexpr uint64_t kStoreMask2 = 0x%llxull;\n", (unsigned long long)mask[0], (unsigned long long)mask[1]);
	if (NW > 128) {                                                       // longer records: one mask per 64 words (klg_kernels.hpp patch_stores)
		s += "\tstatic __device__ __forceinline__ constexpr bool stores_word(int w) { constexpr uint64_t m[] = {";
		for (int i = 0; i < (NW + 63) / 64; i++) s += fmt(" 0x%llxull,", (unsigned long long)mask[i]);
		s += " }; return ((m[w >> 6] >> (w & 63)) & 1ull) != 0; }\n";
	}
	s += live;
	if (!fx && noise_calls) { begin += "\t\tL.sidx = 0;\n"; body += "\t\tL.sidx++;\n"; s += fmt("\tstatic constexpr int kNoiseDraws = %d;\n", noise_calls); }   // a voice's draws of the block: [sample][Noise generator in process() order]
	if (!fx) {
		// delay lines in notes: a wave keeps 64 voices x (read head, tap, write) x one 64-byte sector live while it walks its lines; at full
		// occupancy the waves of an XCD hold more live sectors than its 4 MB L2 and every access becomes an HBM sector.  One wave per SIMD
		// keeps the live set in L2: 3.3e10 voice*samples/s at 1 Mi voices against 2.5e10 / 2.4e10 / 2.2e10 with 2 / 3 / 4 (measured,
		// tools/pluck_bench.py; KLG_NDELAY_WAVES overrides)
		bool nd = false; for (int k : g.nodes) if (k == N_NDELAY) nd = true;
		const char* e = getenv("KLG_NDELAY_WAVES");
		if (nd) s += fmt("\tstatic constexpr int kWavesPerEu = %d;\n", e ? atoi(e) : 1);
	}
	if (fx) {
		s += fmt("\tstatic constexpr int kChannels = %d;\n", g.channels);
		s += fmt("\tstatic constexpr int kRingRow = %d;              // instances per row of the position-major delay lines (FxCtx::ring = the instance's column of its group's tile)\n", ring_row);
		s += "\tstatic __device__ __forceinline__ void begin(Live& L, const Rec& r, const FxCtx& c) {\n\t\tL.unused_ = 0; L.sidx = 0; (void)r; (void)c;\n" + begin + "\t}\n";
		s += "\tstatic __device__ __forceinline__ void begin_core(Live& L, const Rec& r, const FxCtx& c) {\n\t\tL.unused_ = 0; L.sidx = 0; (void)r; (void)c;\n" + begin_core + "\t}\n";
		s += "\tstatic __device__ __forceinline__ void sample(Live& L, const FxCtx& c, float in0, float in1, float& out0, float& out1) {\n\t\t(void)in0; (void)in1;\n" + body
			+ fmt("\t\tout0 = r%d;\n", g.ret) + (g.channels == 2 ? fmt("\t\tout1 = r%d;\n", g.ret_r) : std::string("\t\t(void)out1;\n")) + "\t\tL.sidx++;\n\t}\n";
		s += "\tstatic __device__ __forceinline__ void end(const Live& L, Rec& r) {\n\t\t(void)L; (void)r;\n" + end + "\t}\n};\n";
		// the sample-parallel form of the same body (klg_graph_staged.hpp), when the program has one: KLG_FX_STAGED=0 never, KLG_FX_STAGED_G / _C force the shape
		if (staged) {
			*staged = staged_plan(false);
			if (staged->ok != has_staged || (staged->ok && staged->G != ring_row)) { staged->ok = false; staged->why = "the plan changed between its two passes"; }     // (cannot happen: the plan does not depend on the ops' text)
			if (staged->ok) s += staged->source;
			else s += "// no sample-parallel form (klg_graph_staged.hpp): " + staged->why + "\n";
		}
		s += "}\n";
	}
	else {
		const std::string ctx = x2 ? "BlockCtx2" : "BlockCtx";
		s += "\tstatic __device__ __forceinline__ void begin(Live& L, const Rec& r, const " + ctx + "& c) {\n\t\tL.stage = to_i(r.w[0] & 3u); L.tinc = c.fs.timeInc;\n" + begin + "\t}\n";
		// a Stereo::Note returns both channels of its `out` (klang.h:4721-4733)
		const bool st = g.stereo_note();
		const std::string RT = st ? "Out2" : TF, retline = st ? fmt("\t\treturn Out2{ r%d, r%d };\n\t}\n", g.ret, g.ret_r) : fmt("\t\treturn r%d;\n\t}\n", g.ret);
		if (st) s += "\tstatic constexpr bool kStereo = true;\n";
		s += "\tstatic __device__ __forceinline__ " + RT + " sample(Live& L, const " + ctx + "& c) {\n" + body + retline;
		{
			// the same body with every ADSR holding (adsr_hold): what the render kernel runs for a chunk when quiet() says every envelope
			// of the wave only has its Sustain clock to advance.  (Envelope nodes have no such form: their patches never are quiet; a
			// body with branches keeps one form.)
			std::string quiet_test, d0_test, qbody = body; bool has_env = false, has_adsr = false, has_if = false;
			for (size_t i = 0; i < g.nodes.size(); i++) {
				if (g.nodes[i] == N_ENV || g.nodes[i] == N_OPERATOR) has_env = true;
				if (g.nodes[i] == N_ADSR) { has_adsr = true; quiet_test += fmt(" & adsr_quiet(L.n%zu)", i); }
				if (g.nodes[i] == N_SAW && !retuned[i]) d0_test += fmt(" && L.n%zud0", i);
			}
			for (const Op& o : g.ops) if (o.code == OP_IF) has_if = true;
			// One voice per lane: EVENT-FREE chunks (klg_patches.hpp: env_safe / env_glide).  When no envelope of any sounding voice of the wave
			// can reach a segment end inside the chunk, every ADSR / Envelope / Operator envelope is `out = value; value += step; time += tstep`
			// — holding at sustain is the special case step = -0.0.  quiet() decides per chunk and leaves the steps in Live.
			const bool glide = !x2 && (has_adsr || has_env) && !has_if;
			const bool quiet = glide || (has_adsr && !has_env && !has_if);
			std::string glide_test;
			if (glide) {
				for (size_t i = 0; i < g.nodes.size(); i++) {
					const std::string n = fmt("L.n%zu", i);
					if (g.nodes[i] == N_ADSR) glide_test += "\t\tsafe = env_safe(" + n + ".e, " + n + ".e.point == 2, " + n + "gs, " + n + "gt, L.tinc) && safe;\n";
					else if (g.nodes[i] == N_ENV) glide_test += "\t\tsafe = env_safe(" + n + ", env_settled(" + n + ", " + n + "ls, " + n + "le, " + n + "hy), " + n + "gs, " + n + "gt, L.tinc) && safe;\n";
					else if (g.nodes[i] == N_OPERATOR) glide_test += "\t\tsafe = env_safe(" + n + "e, env_settled(" + n + "e, " + n + "ls, " + n + "le, " + n + "hy), " + n + "gs, " + n + "gt, L.tinc) && safe;\n";
				}
				qbody = std::regex_replace(qbody, std::regex("adsr_process\\((L\\.n[0-9]+), c\\.fs\\)"), "env_glide($1.e, $1gs, $1gt)");
				qbody = std::regex_replace(qbody, std::regex("env_process_rt\\((L\\.n[0-9]+)e, [^)]*\\)"), "env_glide($1e, $1gs, $1gt)");
				qbody = std::regex_replace(qbody, std::regex("env_process_rt\\((L\\.n[0-9]+), [^)]*\\)"), "env_glide($1, $1gs, $1gt)");
			}
			else for (size_t at = 0; quiet && (at = qbody.find("adsr_process(", at)) != std::string::npos;) qbody.replace(at, 13, "adsr_hold(");
			// ... and with every saw in its duty-0 form (the per-block decision n<k>d0 folded into the choice of body)
			std::string fbody = qbody;
			for (size_t at = 0; (at = fbody.find("d0 ? osm_saw_duty0(", at)) != std::string::npos;) {
				const size_t open = fbody.rfind('(', at), colon = fbody.find(" : osm_saw(", at), close = fbody.find("))", colon);
				const std::string call = fbody.substr(at + 5, colon - (at + 5));            // osm_saw_duty0(L.nK)
				fbody.replace(open, close + 2 - open, call);
				at = open;
			}
			s += std::string("\tstatic constexpr bool kHasQuiet = ") + (quiet ? "true" : "false") + ";\n";
			// 0: the full body; 1: every envelope of the wave holds (or glides); 2: ... and every saw is in its duty-0 form
			if (glide) s += "\tstatic __device__ __forceinline__ int quiet(Live& L) {\n\t\tbool safe = true;\n" + glide_test + "\t\tif (__ballot(L.stage != (int)ST_OFF && !safe) != 0ull) return 0;\n\t\treturn (true" + d0_test + ") ? 2 : 1;\n\t}\n";
			else if (x2) s += "\tstatic __device__ __forceinline__ int quiet(const Live& L) {\n\t\t(void)L; const i2 q = (i2)(-1)" + (quiet ? quiet_test : std::string("")) + ";\n\t\tif (__ballot((q.x & q.y) == 0) != 0ull) return 0;\n\t\treturn (true" + d0_test + ") ? 2 : 1;\n\t}\n";
			else s += "\tstatic __device__ __forceinline__ int quiet(const Live& L) {\n\t\t(void)L; const bool q = true" + (quiet ? quiet_test : std::string("")) + ";\n\t\tif (__ballot(!q) != 0ull) return 0;\n\t\treturn (true" + d0_test + ") ? 2 : 1;\n\t}\n";
			s += "\tstatic __device__ __forceinline__ " + RT + " sample_quiet(Live& L, const " + ctx + "& c) {\n" + (quiet ? qbody : body) + retline;
			s += "\tstatic __device__ __forceinline__ " + RT + " sample_fast(Live& L, const " + ctx + "& c) {\n" + (quiet ? fbody : body) + retline;
		}
		// ---- the SAMPLE-PARALLEL tile of the same body (klg_render_sp.hpp: klg_render_gsp<PatchGen, ..>), when the program has one ----
		// sample j of a tile of cnt <= SLOTS samples, from the state at the tile's start: oscillators closed-form, envelopes and filters walked by every lane of the voice
		// together (each keeps its own sample's value), everything else per lane.  The SAME primitives in the same order per value as sample(): the same bits.
		if (!x2) {
			std::string why;
			std::vector<int> uses(g.nodes.size(), 0);
			for (const Op& o : g.ops) if (o.node >= 0 && (o.code == OP_OSC || o.code == OP_LPF || o.code == OP_ENV || o.code == OP_OPERATOR)) uses[(size_t)o.node]++;
			if (st) why = "a stereo `out`";
			else if (g.prepare_ops) why = "a prepare()";
			for (size_t i = 0; i < g.nodes.size() && why.empty(); i++) {
				const int k = g.nodes[i];
				if (!(k == N_FSINE || k == N_SAW || k == N_PULSE || k == N_ENV || k == N_ADSR || k == N_PARAM || k == N_OPERATOR || is_modifier(k))) why = std::string("a ") + node_name(k) + " node";
				else if (uses[i] > 1) why = std::string("a ") + node_name(k) + " node processed twice per sample";
				else if (k == N_PARAM && written[i]) why = "a member written by process()";
			}
			for (const Op& o : g.ops) if (why.empty()) switch (o.code) {
				case OP_CONST: case OP_CTL: case OP_PARAM: case OP_OSC: case OP_LPF: case OP_ENV: case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: case OP_NEG: case OP_ABS: case OP_TRUNC: case OP_POWC:
				case OP_STOPIF: case OP_FREQ: case OP_OPERATOR: case OP_CMP: break;
				default: why = std::string("op ") + op_name(o.code);
			}
			if (const char* e = getenv("KLG_GRAPH_SP")) if (e[0] == '0') why = "KLG_GRAPH_SP=0";
			if (!why.empty()) s += "\t// no sample-parallel tile (klg_render_sp.hpp): " + why + "\n";
			else {
				// the envelope-like nodes of the body, in node order: what is a chain through the samples and has no input
				struct EnvNode { size_t i; std::string ev, settled, full; };
				std::vector<EnvNode> envs;
				for (size_t i = 0; i < g.nodes.size(); i++) {
					const std::string n = fmt("L.n%zu", i);
					if (!uses[i]) continue;
					if (g.nodes[i] == N_ADSR) envs.push_back({ i, n + ".e", n + ".e.point == 2", "adsr_process(" + n + ", c.fs)" });
					else if (g.nodes[i] == N_ENV) envs.push_back({ i, n, "env_settled(" + n + ", " + n + "ls, " + n + "le, " + n + "hy)", "env_process_rt(" + n + ", " + n + "p, " + n + "np, " + n + "ls, " + n + "le, " + n + "hy, c.fs)" });
					else if (g.nodes[i] == N_OPERATOR) envs.push_back({ i, n + "e", "env_settled(" + n + "e, " + n + "ls, " + n + "le, " + n + "hy)", "env_process_rt(" + n + "e, " + n + "p, " + n + "np, " + n + "ls, " + n + "le, " + n + "hy, c.fs)" });
				}
				const int K = (int)envs.size();
				std::string t = fmt("\tstatic constexpr bool kHasSp = true;\n\tstatic constexpr int kSpEnvs = %d;\n", K);
				t += "\ttemplate<int SLOTS> static __device__ __forceinline__ float sp_tile(Live& L, const BlockCtx& c, const int j, const int cnt, float* X, float* E) {\n\t\t(void)c; (void)X; (void)E;\n";
				// 1. the envelopes of the tile, 32 samples at a time (what env_safe looks ahead).  While no envelope of any voice of the wave has an event inside those
				//    samples, an envelope is `value += step` per sample — and the K envelopes of a voice are K LANES of it: lane k walks envelope k through the samples,
				//    leaves the values in the wave's LDS (E[k][sample]) and the state it arrives at behind them; every lane then takes the K values of ITS sample.  (One walk
				//    for all of a voice's envelopes instead of one per envelope repeated in every lane.)  With fewer lanes per voice than envelopes, or an event in reach:
				//    every lane walks every envelope itself through env_glide / the full Envelope::process.
				for (int k = 0; k < K; k++) t += fmt("\t\tfloat e%zu = 0.f;\n", envs[(size_t)k].i);
				if (K) {
					t += "\t\tfor (int h = 0; h < cnt; h += KLG_CHUNK_MAX) {\n\t\t\tconst int hl = (cnt - h < KLG_CHUNK_MAX) ? (cnt - h) : KLG_CHUNK_MAX;\n";
					t += "\t\t\tbool safe = true;\n";
					for (int k = 0; k < K; k++) t += fmt("\t\t\tfloat st%d = 0.f, ts%d = 0.f; safe = env_safe(", k, k) + envs[(size_t)k].ev + ", " + envs[(size_t)k].settled + fmt(", st%d, ts%d, L.tinc) && safe;\n", k, k);
					t += "\t\t\tsafe = safe || L.stage == (int)ST_OFF;\n";
					t += fmt("\t\t\tif (SLOTS >= %d && __ballot(!safe) == 0ull) {\n", 2 * K);       // (a lane per envelope, and two words per envelope for where it ends)
					t += "\t\t\t\tfloat r = 0.f, sr = 0.f, tm = 0.f, stm = 0.f;\n";
					for (int k = 0; k < K; k++) t += fmt("\t\t\t\tif (j == %d) { r = ", k) + envs[(size_t)k].ev + fmt(".r_out; sr = st%d; tm = ", k) + envs[(size_t)k].ev + fmt(".time; stm = ts%d; }\n", k);
					t += fmt("\t\t\t\tif (j < %d) {\n\t\t\t\t\tfloat* row = E + j * SLOTS + h;\n", K);
					// (a whole chunk: written out — as a loop every sample pays a jump on a wave nothing else runs beside: 7 instructions and ~45 cycles a sample instead of 3 and ~15)
					t += "\t\t\t\t\tconstexpr int WHOLE = SLOTS < KLG_CHUNK_MAX ? SLOTS : KLG_CHUNK_MAX;\n\t\t\t\t\tif (hl == WHOLE) {\n#pragma unroll\n\t\t\t\t\t\tfor (int i = 0; i < WHOLE; i++) { row[i] = r; r = sc_add(r, sr); tm = sc_add(tm, stm); }\n\t\t\t\t\t}\n";
					t += "\t\t\t\t\telse for (int i = 0; i < hl; i++) { row[i] = r; r = sc_add(r, sr); tm = sc_add(tm, stm); }\n";
					t += fmt("\t\t\t\t\tE[%d * SLOTS + 2 * j] = r; E[%d * SLOTS + 2 * j + 1] = tm;\n\t\t\t\t}\n\t\t\t\twave_sync();\n", K, K);
					for (int k = 0; k < K; k++) {
						t += fmt("\t\t\t\te%zu = (j >= h && j < h + hl) ? E[%d * SLOTS + j] : e%zu; ", envs[(size_t)k].i, k, envs[(size_t)k].i) + envs[(size_t)k].ev + fmt(".r_out = E[%d * SLOTS + %d]; ", K, 2 * k) + envs[(size_t)k].ev + fmt(".time = E[%d * SLOTS + %d];\n", K, 2 * k + 1);
					}
					t += "\t\t\t\twave_sync();\n\t\t\t}\n\t\t\telse {\n";
					for (int k = 0; k < K; k++) {
						const EnvNode& e = envs[(size_t)k];
						t += fmt("\t\t\t\tif (safe) { for (int i = 0; i < hl; i++) { const float v = env_glide(", 0) + e.ev + fmt(", st%d, ts%d); e%zu = (h + i == j) ? v : e%zu; } }\n", k, k, e.i, e.i);
						t += "\t\t\t\telse { for (int i = 0; i < hl; i++) { const float v = " + e.full + fmt("; e%zu = (h + i == j) ? v : e%zu; } }\n", e.i, e.i);
					}
					t += "\t\t\t}\n\t\t}\n";
				}
				// 2. the ops in program order, a lane per sample; a filter's recurrence over the tile's inputs (through the wave's LDS: four at a time in a full tile)
				std::string adv;                                              // the closed-form nodes moved on by cnt samples, behind the tile
				for (size_t oi = 0; oi < g.ops.size(); oi++) {
					const Op& o = g.ops[oi];
					const int k = o.node >= 0 ? g.nodes[(size_t)o.node] : -1;
					const std::string n = fmt("L.n%d", o.node), d = fmt("\t\tconst float r%d = ", o.dst), a = fmt("r%d", o.a), b = fmt("r%d", o.b);
					switch (o.code) {
					case OP_STOPIF: break;                                       // (in end(): the envelope's stage after the block)
					case OP_ENV: t += d + fmt("e%d;\n", o.node); break;
					case OP_OSC:
						if (k == N_FSINE) { t += fmt("\t\tFSine t%zu = ", oi) + n + fmt("; t%zu.pos += (uint32_t)t%zu.inc * (uint32_t)j;\n", oi, oi) + d + fmt("fsine_process(t%zu, 0u);\n", oi); adv += "\t\t" + n + ".pos += (uint32_t)" + n + ".inc * (uint32_t)cnt;\n"; }
						else {
							t += fmt("\t\tOsm t%zu = ", oi) + n + fmt("; osm_jump(t%zu, j);\n", oi);
							t += d + (k == N_PULSE ? fmt("osm_pulse(t%zu)", oi) : (retuned[(size_t)o.node] ? fmt("osm_saw(t%zu)", oi) : "(" + n + fmt("d0 ? osm_saw_duty0(t%zu) : osm_saw(t%zu))", oi, oi))) + ";\n";
							adv += "\t\tosm_advance(" + n + ", cnt);\n";
						}
						break;
					case OP_OPERATOR:
						if (o.b >= 0) t += "\t\t" + n + "a = " + b + ";\n";
						t += fmt("\t\tFSine t%zu = ", oi) + n + fmt("; t%zu.pos += (uint32_t)t%zu.inc * (uint32_t)j;\n", oi, oi);
						t += d + (o.a >= 0 ? fmt("fsine_process_rel(t%zu, ", oi) + a + ")" : fmt("fsine_process(t%zu, 0u)", oi)) + fmt(" * (e%d * ", o.node) + n + "a);\n";
						adv += "\t\t" + n + ".pos += (uint32_t)" + n + ".inc * (uint32_t)cnt;\n";
						break;
					case OP_LPF: {
						const char* fn = k == N_LPF ? "biquad_process" : k == N_OPLPF ? "onepole_lpf_process" : k == N_OPHPF ? "onepole_process" : k == N_DCF ? "dcf_process" : k == N_IIR1 ? "iir1_process" : k == N_IIRN ? "iir_process"
							: k == N_BUTTER1 ? "butter1_process" : k == N_MODAL ? "modal_process" : k == N_FOLLOWPEAK ? "follower_peak" : "follower_rms";
						if (k == N_LPF) {
							// a biquad: its feed-forward products b0 x, b1 x, b2 x are taken by the lane that owns the sample (three rows of the wave's LDS), the recurrence every
							// lane walks is what is left — y = p0 + z0; z0 = p1 - a1 y + z1; z1 = p2 - a2 y, the two products with y and the two subtractions one packed operation each
							// ({ b1 x, b2 x } lie side by side) — the same operations on the same values in the same order per value (biquad_process; klg_render_sub2a_sp.hpp does this by hand)
							const std::string x12 = "X12_" + std::to_string(o.dst);
							t += "\t\tf2* const " + x12 + " = sp_pairs(X, j);\n";
							t += "\t\tX[j] = " + n + ".b0 * " + a + "; { const f2 p12 = { " + n + ".b1 * " + a + ", " + n + ".b2 * " + a + " }; " + x12 + "[j] = p12; }\n\t\twave_sync();\n" + fmt("\t\tfloat r%d = 0.f;\n", o.dst);
							t += "\t\tconst f2 a12_" + std::to_string(o.dst) + " = { " + n + ".a1, " + n + ".a2 };\n";
							t += "\t\tif (SLOTS >= 4 && cnt == SLOTS) {\n\t\t\ttypedef float f4_ __attribute__((ext_vector_type(4)));\n#pragma unroll\n\t\t\tfor (int i0 = 0; i0 < SLOTS; i0 += 4) {\n";
							t += "\t\t\t\tconst f4_ p0 = *reinterpret_cast<const f4_*>(X + i0), pa = *reinterpret_cast<const f4_*>(" + x12 + " + i0), pb = *reinterpret_cast<const f4_*>(" + x12 + " + i0 + 2);\n";
							t += fmt("\t\t\t\t{ const float v = biquad_step_pk(%s, a12_%d, p0[0], f2{ pa.x, pa.y }); r%d = sp_keep<SLOTS>(r%d, v, i0); }\n", n.c_str(), o.dst, o.dst, o.dst);
							t += fmt("\t\t\t\t{ const float v = biquad_step_pk(%s, a12_%d, p0[1], f2{ pa.z, pa.w }); r%d = sp_keep<SLOTS>(r%d, v, i0 + 1); }\n", n.c_str(), o.dst, o.dst, o.dst);
							t += fmt("\t\t\t\t{ const float v = biquad_step_pk(%s, a12_%d, p0[2], f2{ pb.x, pb.y }); r%d = sp_keep<SLOTS>(r%d, v, i0 + 2); }\n", n.c_str(), o.dst, o.dst, o.dst);
							t += fmt("\t\t\t\t{ const float v = biquad_step_pk(%s, a12_%d, p0[3], f2{ pb.z, pb.w }); r%d = sp_keep<SLOTS>(r%d, v, i0 + 3); }\n\t\t\t}\n\t\t}\n", n.c_str(), o.dst, o.dst, o.dst);
							t += fmt("\t\telse for (int i = 0; i < cnt; i++) { const float v = biquad_step_pk(%s, a12_%d, X[i], %s[i]); r%d = (i == j) ? v : r%d; }\n\t\twave_sync();\n", n.c_str(), o.dst, x12.c_str(), o.dst, o.dst);
							break;
						}
						t += "\t\tX[j] = " + a + ";\n\t\twave_sync();\n" + fmt("\t\tfloat r%d = 0.f;\n", o.dst);
						t += "\t\tif (SLOTS >= 4 && cnt == SLOTS) {\n\t\t\ttypedef float f4_ __attribute__((ext_vector_type(4)));\n#pragma unroll\n\t\t\tfor (int i0 = 0; i0 < SLOTS; i0 += 4) {\n\t\t\t\tconst f4_ x4 = *reinterpret_cast<const f4_*>(X + i0);\n";
						t += std::string("#pragma unroll\n\t\t\t\tfor (int q = 0; q < 4; q++) { const float v = ") + fn + "(" + n + fmt(", x4[q]); r%d = sp_keep<SLOTS>(r%d, v, i0 + q); }\n\t\t\t}\n\t\t}\n", o.dst, o.dst);
						t += std::string("\t\telse for (int i = 0; i < cnt; i++) { const float v = ") + fn + "(" + n + fmt(", X[i]); r%d = (i == j) ? v : r%d; }\n\t\twave_sync();\n", o.dst, o.dst);
					} break;
					default: emit_op(oi, t, false); break;                       // arithmetic, literals, controls, members, `osc.frequency`: per lane, as in sample()
					}
				}
				t += adv + fmt("\t\treturn r%d;\n\t}\n", g.ret);
				s += t;
			}
		}
		std::string stage_expr = "L.stage";
		for (const std::string& t : stop_at_end) stage_expr = t + stage_expr + ")";
		s += "\tstatic __device__ __forceinline__ void end(const Live& L, Rec& r) {\n\t\tr.w[0] = to_u(" + stage_expr + ");\n" + end + "\t}\n";
		s += "\tstatic __device__ __forceinline__ void release(Rec&, float) {}\n};\n}\n";
	}
	return s;
}

// ------------------------------------------------------------------------------------------------
// hipRTC (dlopen)
// ------------------------------------------------------------------------------------------------
struct Rtc {
	void* lib = nullptr;
	int (*CreateProgram)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
	int (*AddNameExpression)(void*, const char*) = nullptr;
	int (*CompileProgram)(void*, int, const char**) = nullptr;
	int (*GetProgramLogSize)(void*, size_t*) = nullptr;
	int (*GetProgramLog)(void*, char*) = nullptr;
	int (*GetCodeSize)(void*, size_t*) = nullptr;
	int (*GetCode)(void*, char*) = nullptr;
	int (*GetLoweredName)(void*, const char*, const char**) = nullptr;
	int (*DestroyProgram)(void**) = nullptr;
	int (*Version)(int*, int*) = nullptr;
	std::string error;
	bool load() {
		if (lib) return true;
		const char* names[] = { "libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so" };
		for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
		if (!lib) { error = std::string("cannot load libhiprtc.so: ") + dlerror(); return false; }
		bool ok = true;
		auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) { ok = false; error = std::string("libhiprtc.so lacks ") + n; } return p; };
		CreateProgram = (decltype(CreateProgram))sym("hiprtcCreateProgram");
		AddNameExpression = (decltype(AddNameExpression))sym("hiprtcAddNameExpression");
		CompileProgram = (decltype(CompileProgram))sym("hiprtcCompileProgram");
		GetProgramLogSize = (decltype(GetProgramLogSize))sym("hiprtcGetProgramLogSize");
		GetProgramLog = (decltype(GetProgramLog))sym("hiprtcGetProgramLog");
		GetCodeSize = (decltype(GetCodeSize))sym("hiprtcGetCodeSize");
		GetCode = (decltype(GetCode))sym("hiprtcGetCode");
		GetLoweredName = (decltype(GetLoweredName))sym("hiprtcGetLoweredName");
		DestroyProgram = (decltype(DestroyProgram))sym("hiprtcDestroyProgram");
		Version = (decltype(Version))dlsym(lib, "hiprtcVersion");
		if (!ok) { dlclose(lib); lib = nullptr; }
		return ok;
	}
};

struct Compiled { std::vector<char> code; std::string name[2]; std::string source; int words = 0; int channels = 0; int note_channels = 1;   // note_channels: 2 = the notes' `out` is stereo (ret2 in a note program)
	 long long ring_rows = 0; int noise_calls = 0; struct Smooth { int word, ctl, calls; }; std::vector<Smooth> smooths; int ctlvar_word[KLG_MAX_CTL] = { -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1 };   // ctlvar_word[i]: the record word of control i's own copy (an effect that writes it), else -1
	 bool x2 = false; std::vector<std::pair<long long, int>> delays;
	 bool sp = false;                                                    // the code object also holds klg_render_gsp<PatchGen, pv, 1 | 8> (klg_render_sp.hpp): sp_kernel_name()
	 bool staged = false; std::string staged_why; int staged_G = 0, staged_C = 0, staged_threads = 0, staged_lds = 0, staged_levels = 0, staged_slots = 0; };   // staged: the code object also holds klg_fx_staged (klg_graph_staged.hpp)   // delays: (first ring row, SIZE) of each delay node in node order   // name[pv] (effects: name[0] only); x2: two voices per lane (klg_render_x2<P>)

// the sample-parallel note kernels of a code object, by their (Itanium-mangled) names: klg::klg_render_gsp<klg::PatchGen, PER_VOICE, VPW>(klg::RenderArgs)
inline std::string sp_kernel_name(bool per_voice, int vpw) { char b[96]; snprintf(b, sizeof b, "_ZN3klg14klg_render_gspINS_8PatchGenELb%dELi%dEEEvNS_10RenderArgsE", per_voice ? 1 : 0, vpw); return b; }
// directory holding klg_kernels.hpp etc.: next to the shared library (klang_amd/csrc), or $KLG_GRAPH_SRC
inline std::string source_dir() {
	if (const char* e = getenv("KLG_GRAPH_SRC")) return e;
	Dl_info info;
	if (dladdr((const void*)&source_dir, &info) && info.dli_fname) {
		std::string p = info.dli_fname;
		const size_t slash = p.rfind('/');
		return (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/csrc";
	}
	return "klang_amd/csrc";
}

// ---- code objects on disk -----------------------------------------------------------------------------------------------------------------
// hipRTC takes ~2 s per program; a plug-in host that instantiates Reverb.k pays that on every start unless the code object of an earlier process is
// still good.  It is when NOTHING that went into it has changed: the generated source (which carries the program and every switch that shapes it), the
// headers it includes (every *.hpp next to the library + the two record / graph headers: their bytes are hashed), the compiler (hiprtcVersion) and its options.
// Files: $KLG_CACHE_DIR, else $XDG_CACHE_HOME/klang_mi355, else $HOME/.cache/klang_mi355; KLG_CACHE=0 turns the cache off.  A file that does not verify
// (magic, sizes, the stored hash of its own payload) is ignored and rewritten.  Written to a temporary name and renamed: concurrent processes are safe.
inline uint64_t fnv1a(const void* data, size_t n, uint64_t h = 1469598103934665603ull) { const unsigned char* p = (const unsigned char*)data; for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; } return h; }
inline std::string cache_dir() {
	const char* off = getenv("KLG_CACHE"); if (off && off[0] == '0') return "";
	std::string d;
	if (const char* e = getenv("KLG_CACHE_DIR")) d = e;
	else if (const char* x = getenv("XDG_CACHE_HOME")) d = std::string(x) + "/klang_mi355";
	else if (const char* h = getenv("HOME")) d = std::string(h) + "/.cache/klang_mi355";
	if (d.empty()) return "";
	std::string cur;
	for (size_t i = 0; i <= d.size(); i++) if (i == d.size() || d[i] == '/') { cur = d.substr(0, i); if (!cur.empty()) (void)mkdir(cur.c_str(), 0755); }
	return access(d.c_str(), W_OK) == 0 ? d : "";
}
inline uint64_t headers_hash() {
	static const uint64_t h = []() {
		uint64_t x = 1469598103934665603ull;
		const std::string dir = source_dir();
		std::vector<std::string> files;
		if (DIR* dp = opendir(dir.c_str())) { while (dirent* e = readdir(dp)) { const std::string n = e->d_name; if (n.size() > 4 && n.substr(n.size() - 4) == ".hpp") files.push_back(dir + "/" + n); } closedir(dp); }
		files.push_back(dir + "/../../include/klang_mi355_records.h"); files.push_back(dir + "/../../include/klang_mi355_graph.h");
		std::sort(files.begin(), files.end());
		for (const std::string& f : files) if (FILE* fp = fopen(f.c_str(), "rb")) { char buf[65536]; size_t n; x = fnv1a(f.data() + dir.size(), f.size() - dir.size(), x); while ((n = fread(buf, 1, sizeof buf, fp)) > 0) x = fnv1a(buf, n, x); fclose(fp); }
		return x;
	}();
	return h;
}
inline std::string cache_file(const Rtc& rtc, const std::string& source, const std::string& options) {
	const std::string dir = cache_dir();
	if (dir.empty()) return "";
	int major = 0, minor = 0; if (rtc.Version) (void)rtc.Version(&major, &minor);
	uint64_t a = fnv1a(source.data(), source.size()), b = fnv1a(source.data(), source.size(), 0x9E3779B97F4A7C15ull);
	a = fnv1a(options.data(), options.size(), a); const uint64_t hh = headers_hash(); a = fnv1a(&hh, sizeof hh, a); a = fnv1a(&major, sizeof major, a); a = fnv1a(&minor, sizeof minor, a);
	char name[96]; snprintf(name, sizeof name, "/%016llx%016llx_%zu.klgco", (unsigned long long)a, (unsigned long long)b, source.size());
	return dir + name;
}
inline bool cache_load(const std::string& path, Compiled& c) {
	FILE* fp = path.empty() ? nullptr : fopen(path.c_str(), "rb");
	if (!fp) return false;
	bool ok = false;
	char magic[8] = {}; uint64_t sizes[3] = {}, sum = 0;
	if (fread(magic, 1, 8, fp) == 8 && memcmp(magic, "KLGCO01\n", 8) == 0 && fread(sizes, 8, 3, fp) == 3 && sizes[0] < 4096 && sizes[1] < 4096 && sizes[2] > 0 && sizes[2] < (1ull << 30)) {
		std::string n0(sizes[0], '\0'), n1(sizes[1], '\0'); std::vector<char> code(sizes[2]);
		if (fread(&n0[0], 1, n0.size(), fp) == n0.size() && fread(&n1[0], 1, n1.size(), fp) == n1.size() && fread(code.data(), 1, code.size(), fp) == code.size() && fread(&sum, 8, 1, fp) == 1
		    && sum == fnv1a(code.data(), code.size(), fnv1a(n1.data(), n1.size(), fnv1a(n0.data(), n0.size())))) { c.name[0] = n0; c.name[1] = n1; c.code.swap(code); ok = true; }
	}
	fclose(fp);
	return ok;
}
inline void cache_store(const std::string& path, const Compiled& c) {
	if (path.empty()) return;
	const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
	FILE* fp = fopen(tmp.c_str(), "wb");
	if (!fp) return;
	const uint64_t sizes[3] = { c.name[0].size(), c.name[1].size(), c.code.size() };
	const uint64_t sum = fnv1a(c.code.data(), c.code.size(), fnv1a(c.name[1].data(), c.name[1].size(), fnv1a(c.name[0].data(), c.name[0].size())));
	const bool ok = fwrite("KLGCO01\n", 1, 8, fp) == 8 && fwrite(sizes, 8, 3, fp) == 3 && fwrite(c.name[0].data(), 1, sizes[0], fp) == sizes[0] && fwrite(c.name[1].data(), 1, sizes[1], fp) == sizes[1]
		&& fwrite(c.code.data(), 1, sizes[2], fp) == sizes[2] && fwrite(&sum, 8, 1, fp) == 1;
	if (fclose(fp) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
}

// program text -> code object (cached per process by program text).  Returns "" on success.
inline std::string compile(const char* text, const Compiled** out, bool x2 = false, int staged_G = 0) {
	static std::mutex mu;
	static std::map<std::string, Compiled> cache;
	static Rtc rtc;
	std::lock_guard<std::mutex> lock(mu);
	Program g;
	const std::string perr = g.parse(text);
	if (!perr.empty()) return perr;
	if (x2 && !x2_eligible(g)) return "graph program: not every node / op has a two-voices-per-lane form";
	auto envs = [](const char* n) { const char* e = getenv(n); return std::string(e ? e : ""); };
	const std::string key = (x2 ? "x2\n" : "") + std::string(envs("KLG_GRAPH_SP") == "0" ? "nosp\n" : "") + (g.channels ? "staged G" + std::to_string(staged_G) + " " + envs("KLG_FX_STAGED") + "," + envs("KLG_FX_STAGED_G") + "," + envs("KLG_FX_STAGED_C") + "," + envs("KLG_FX_STAGED_LDS") + "," + envs("KLG_FX_STAGED_SKIP") + "," + envs("KLG_FX_STAGED_STAMP") + "," + envs("KLG_FX_STAGED_PIPE") + "," + envs("KLG_FX_STAGED_BATCH") + "," + envs("KLG_FX_STAGED_NEAR") + "," + envs("KLG_FX_STAGED_PACK") + "," + envs("KLG_FX_STAGED_TILES") + "," + envs("KLG_FX_STAGED_RETRY") + "\n" : std::string()) + g.text();
	auto it = cache.find(key);
	if (it != cache.end()) { *out = &it->second; return ""; }
	if (!rtc.load()) return rtc.error;
	Compiled c;
	StagedPlan plan;
	c.source = generate_source(g, x2, g.channels ? &plan : nullptr, staged_G);
	c.staged = plan.ok; c.staged_why = plan.why; c.staged_G = plan.G; c.staged_C = plan.C; c.staged_threads = plan.threads; c.staged_lds = plan.lds_bytes; c.staged_levels = plan.levels; c.staged_slots = plan.slots;
	c.words = g.words(); c.channels = g.channels; c.x2 = x2; c.note_channels = g.stereo_note() ? 2 : 1;
	c.sp = !g.channels && !x2 && c.source.find("kHasSp") != std::string::npos;
	for (size_t i = 0; i < g.nodes.size(); i++) if (g.nodes[i] == graph::N_DELAY || g.nodes[i] == graph::N_NDELAY) { c.delays.push_back({ c.ring_rows, g.arg((int)i) }); c.ring_rows += g.arg((int)i) + 1; }   // (+ the pad element of every line: klg_delay.hpp)
	c.noise_calls = g.noise_calls();
	for (const graph::Op& o : g.ops) if (o.code == graph::OP_SETCTL) c.ctlvar_word[o.imm & 0xFFu] = g.node_word0(o.node);
	for (size_t i = 0; i < g.nodes.size(); i++) if (g.nodes[i] == graph::N_SMOOTH) {            // controls[ctl].smooth(): state word, control, calls per sample
		Compiled::Smooth sm = { g.node_word0((int)i), -1, 0 };
		for (const graph::Op& o : g.ops) if (o.code == graph::OP_SMOOTH && o.node == (int)i) { sm.ctl = (int)o.imm; sm.calls++; }
		if (sm.calls) c.smooths.push_back(sm);
	}
	const char* const extra_opt = getenv("KLG_RTC_EXTRA");
	const std::string disk = cache_file(rtc, c.source, std::string("--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off ") + (extra_opt ? extra_opt : ""));
	if (cache_load(disk, c)) {
		if (getenv("KLG_GRAPH_DEBUG")) fprintf(stderr, "klang-mi355: code object from %s\n", disk.c_str());
		auto ins = cache.emplace(key, std::move(c));
		*out = &ins.first->second;
		return "";
	}
	void* prog = nullptr;
	if (rtc.CreateProgram(&prog, c.source.c_str(), "klg_graph_patch.hip", 0, nullptr, nullptr) != 0) return "hiprtcCreateProgram failed";
	const char* expr[2] = { "klg::klg_render<klg::PatchGen, false>", "klg::klg_render<klg::PatchGen, true>" };
	if (g.channels) expr[0] = expr[1] = "klg::klg_fx_graph<klg::PatchGen>";
	if (x2) { expr[0] = "klg::klg_render_x2<klg::PatchGen, false>"; expr[1] = "klg::klg_render_x2<klg::PatchGen, true>"; }
	rtc.AddNameExpression(prog, expr[0]); if (!g.channels) rtc.AddNameExpression(prog, expr[1]);
	if (c.sp) for (const char* e : { "klg::klg_render_gsp<klg::PatchGen, false, 1>", "klg::klg_render_gsp<klg::PatchGen, true, 1>", "klg::klg_render_gsp<klg::PatchGen, false, 4>", "klg::klg_render_gsp<klg::PatchGen, true, 4>",
	                                "klg::klg_render_gsp<klg::PatchGen, false, 8>", "klg::klg_render_gsp<klg::PatchGen, true, 8>" }) rtc.AddNameExpression(prog, e);
	const std::string inc = "-I" + source_dir();
	const char* extra = getenv("KLG_RTC_EXTRA");                             // (measurement: one more compiler option for the generated kernels)
	const char* opts[] = { "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", inc.c_str(), extra };
	const int rc = rtc.CompileProgram(prog, (extra && extra[0]) ? 6 : 5, opts);
	if (rc != 0) {
		size_t n = 0; rtc.GetProgramLogSize(prog, &n);
		std::string log(n, '\0'); if (n) rtc.GetProgramLog(prog, &log[0]);
		rtc.DestroyProgram(&prog);
		return "hipRTC compilation of the graph patch failed (headers expected in " + source_dir() + "; set KLG_GRAPH_SRC):\n" + log;
	}
	size_t n = 0; rtc.GetCodeSize(prog, &n);
	c.code.resize(n); rtc.GetCode(prog, c.code.data());
	for (int i = 0; i < 2; i++) { const char* ln = nullptr; rtc.GetLoweredName(prog, expr[i], &ln); c.name[i] = ln ? ln : ""; }
	rtc.DestroyProgram(&prog);
	if (c.name[0].empty() || c.name[1].empty()) return "hipRTC: lowered kernel names not found";
	cache_store(disk, c);
	auto ins = cache.emplace(key, std::move(c));
	*out = &ins.first->second;
	return "";
}

} }  // namespace klg::graphrt
