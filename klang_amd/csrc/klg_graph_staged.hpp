// klang_amd/csrc/klg_graph_staged.hpp — the SAMPLE-PARALLEL form of a recorded effect (include/klang_mi355_graph.h, `kind effect`).
//
// klg_fx_graph<P> (klg_fx.hpp) walks an effect's samples one after the other with one lane per instance: a bank of 4,096 instances is 64 waves on a chip of
// 1,024 SIMDs, each running one sample's whole dependent chain (~330 instructions for PingPong.k, ~750 recorded ops for Reverb.k) 256 times per block.  Most
// of a sample does not depend on the sample before it.  What does is found here, from the op list as recorded:
//
//   * a dependency graph over the ops of ONE sample: register def -> use, branch condition -> the ops inside the branch, the ops of a stateful node in program
//     order, and the edge that crosses samples: a stateful node's last op -> its first op of the next sample (an oscillator's phase, a filter's z, a smoother, a
//     control or member the body writes).  A Delay's input() -> a later tap is NOT an edge: the write cursor is the sample counter (closed form), and whether a
//     tap reads a row its own chunk writes is checked at run time (klg_delay.hpp "STAGED EFFECTS");
//   * its strongly connected components in topological order.  A component that is a single op without a self edge is PARALLEL: it runs with a lane per
//     (sample of the chunk, instance).  Every other component is SERIAL: its ops run in program order inside `for s < C` on one lane per instance;
//   * levels: parallel components sit on even levels, serial ones on odd levels, a component one level above everything it waits for that is of the other kind
//     (or of its own kind where that costs nothing).  A workgroup — G instances x C samples = G * C lanes — runs level after level with a barrier in between;
//     serial components of one level that do not feed each other go to different waves.  Values between levels travel through LDS ([C][G] floats per value,
//     slots reused once a value is dead) or stay in the lane's registers (parallel -> parallel);
//   * every ring read of a chunk is issued, and checked, before the chunk's first ring write: a chunk whose check fails has changed nothing and is walked in
//     sample order by the plain body (PatchGen::begin_core / sample / end on one lane per instance), as is a ragged last chunk.
//
// Node state lives in an LDS copy of the G records between chunks (a serial component loads its nodes at the start of its loop and commits them once the
// chunk's check has passed), so the plain body finds everything where it expects it.  What is not handled is refused (StagedPlan::why) and the bank keeps
// klg_fx_graph<P>: a conditional input() / set() / process() of a Delay, a Delay op inside a serial component, a double register that would have to cross levels.
#pragma once

namespace klg { namespace graphrt {

struct StagedPlan {
	bool ok = false; std::string why;
	int G = 16, C = 32, threads = 0, lds_bytes = 0, levels = 0, slots = 0, serial_ops = 0, parallel_ops = 0;
	std::string source;                                                  // the kernel (appended to the generated translation unit, inside namespace klg)
};

struct StagedInput {
	const graph::Program* g;
	std::function<void(size_t, std::string&, bool)> emit_op;             // generate_source's op emitter (text in terms of L. / c. / r<N>)
	const std::vector<std::string>* node_begin; const std::vector<std::string>* node_end;
	std::string ctl_begin;
	const std::vector<long long>* ring_off; const std::vector<int>* inputs;
	const int* ctlvar;                                                   // control index -> ctlvar node or -1
	int G, C;                                                            // requested width / chunk (0: choose)
};

inline StagedPlan plan_staged(const StagedInput& in) {
	using namespace graph;
	StagedPlan P;
	const Program& g = *in.g;
	auto refuse = [&](const std::string& w) { P.ok = false; P.why = w; return P; };
	if (g.channels < 1) return refuse("not an effect program");
	const int first = g.prepare_ops, N = (int)g.ops.size();
	if (N - first < 1) return refuse("no sample ops");

	// ---- virtual ops: the sample ops, with Basic::Sine's process() split into its phase walk (state) and the sine of the argument (pure) ----
	enum { V_OSCARG = OP_CODES + 1, V_OSCEVAL };
	struct VOp { int code, dst, a, b, node; uint32_t imm; int orig; std::vector<std::pair<int, int>> path; int parent_if = -1; };   // path: (vop index of the `if`, side 0 then / 1 else), outermost first
	std::vector<VOp> V;
	int maxreg = -1;
	for (const Op& o : g.ops) maxreg = std::max(maxreg, std::max(o.dst, std::max(o.a, o.b)));
	std::vector<char> is_dbl((size_t)maxreg + 2 + (size_t)N, 0);
	for (const Op& o : g.ops) if (o.code == OP_F2D || o.code == OP_DCONST || o.code == OP_DLOW || (o.code >= OP_DADD && o.code <= OP_DDIV)) is_dbl[(size_t)o.dst] = 1;
	std::vector<bool> written(g.nodes.size(), false);
	for (int i = first; i < N; i++) if (g.ops[(size_t)i].code == OP_SETPARAM) written[(size_t)g.ops[(size_t)i].node] = true;
	{
		std::vector<std::pair<int, int>> path;
		for (int i = first; i < N; i++) {
			const Op& o = g.ops[(size_t)i];
			if (o.code == OP_STOP || o.code == OP_STOPIF || o.code == OP_TABREAD) return refuse("op without a staged form");
			VOp v; v.code = o.code; v.dst = o.dst; v.a = o.a; v.b = o.b; v.node = o.node; v.imm = o.imm; v.orig = i;
			if (o.code == OP_ELSE) { if (path.empty()) return refuse("unbalanced else"); path.back().second = 1; v.path = path; V.push_back(v); continue; }
			if (o.code == OP_ENDIF) { if (path.empty()) return refuse("unbalanced endif"); path.pop_back(); v.path = path; V.push_back(v); continue; }
			v.path = path;
			if (o.code == OP_OSC && o.node >= 0 && g.nodes[(size_t)o.node] == N_BSINE) {
				VOp arg = v; arg.code = V_OSCARG; arg.dst = ++maxreg; V.push_back(arg);
				v.code = V_OSCEVAL; v.a = arg.dst; v.node = -1; V.push_back(v);
				continue;
			}
			V.push_back(v);
			if (o.code == OP_IF) path.push_back({ (int)V.size() - 1, 0 });
		}
		if (!path.empty()) return refuse("unbalanced if");
	}
	const int NV = (int)V.size();
	std::vector<int> def_at((size_t)maxreg + 2, -1);
	for (int i = 0; i < NV; i++) if (V[(size_t)i].dst >= 0 && V[(size_t)i].code != OP_IF && V[(size_t)i].code != OP_ELSE && V[(size_t)i].code != OP_ENDIF) def_at[(size_t)V[(size_t)i].dst] = i;
	auto is_struct = [&](int c) { return c == OP_IF || c == OP_ELSE || c == OP_ENDIF; };
	// the `if` a phi belongs to: the one whose endif directly precedes it (phis of one `if` follow each other)
	std::vector<int> phi_if((size_t)NV, -1);
	{
		std::vector<int> open; int last_closed = -1;
		for (int i = 0; i < NV; i++) {
			const int c = V[(size_t)i].code;
			if (c == OP_IF) open.push_back(i);
			else if (c == OP_ENDIF) { last_closed = open.back(); open.pop_back(); }
			else if (c == OP_PHI) phi_if[(size_t)i] = last_closed;
		}
	}

	// ---- delay lines: every input() / set() / process() unconditional; static position of each op inside its sample ----
	const size_t NN = g.nodes.size();
	std::vector<int> k_in(NN, 0), outs(NN, 0), set_at(NN, -1);
	std::vector<int> in_index((size_t)NV, 0), out_index((size_t)NV, 0);      // DELAYIN: which input of the sample; taps / set: inputs of the line before it; DELAYOUT: process() calls before it
	std::vector<bool> head_used(NN, false);
	for (const Op& o : g.ops) if ((o.code == OP_DELAYSET || o.code == OP_DELAYOUT) && o.node >= 0) head_used[(size_t)o.node] = true;     // (prepare() included: what end() stores)
	for (int i = 0; i < NV; i++) {
		const VOp& v = V[(size_t)i];
		const bool dl = v.code == OP_DELAYIN || v.code == OP_DELAYTAP || v.code == OP_DELAYOUT || v.code == OP_DELAYSET;
		if (!dl) continue;
		if (v.node < 0 || g.nodes[(size_t)v.node] != N_DELAY) return refuse("delay op on a node that is not an effect Delay");
		if (v.code != OP_DELAYTAP && !v.path.empty()) return refuse("a conditional input() / set() / process() of a Delay (its cursor would depend on the samples)");
		in_index[(size_t)i] = k_in[(size_t)v.node]; out_index[(size_t)i] = outs[(size_t)v.node];
		if (v.code == OP_DELAYIN) k_in[(size_t)v.node]++;
		if (v.code == OP_DELAYOUT) outs[(size_t)v.node]++;
		if (v.code == OP_DELAYSET) { if (set_at[(size_t)v.node] >= 0 || outs[(size_t)v.node] > 0) return refuse("a Delay that is set() twice per sample, or after its process()"); set_at[(size_t)v.node] = i; }
	}
	for (size_t d = 0; d < NN; d++) if (g.nodes[d] == N_DELAY && (k_in[d] * in.C >= g.arg((int)d) || outs[d] * 1024 >= g.arg((int)d))) return refuse("a Delay shorter than a chunk's inputs");

	// ---- block invariants: literals, dials, members process() only reads, and plain arithmetic on those outside any branch.  They belong to no level: whoever
	// needs one computes it (a lane of a parallel level once per chunk, a serial loop in front of its samples) — nothing of them travels through LDS ----
	std::vector<char> inv((size_t)NV, 0);
	for (int i = 0; i < NV; i++) {
		const VOp& v = V[(size_t)i];
		auto opinv = [&](int r) { const int d = (r >= 0 && (size_t)r < def_at.size()) ? def_at[(size_t)r] : -1; return d >= 0 && d < i && inv[(size_t)d]; };
		switch (v.code) {
		case OP_CONST: case OP_DCONST: inv[(size_t)i] = 1; break;
		case OP_CTL: inv[(size_t)i] = in.ctlvar[v.imm & 7u] < 0; break;
		case OP_PARAM: inv[(size_t)i] = !written[(size_t)v.node]; break;
		case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: case OP_CMP: case OP_DADD: case OP_DSUB: case OP_DMUL: case OP_DDIV: inv[(size_t)i] = v.path.empty() && opinv(v.a) && opinv(v.b); break;
		case OP_NEG: case OP_ABS: case OP_F2D: case OP_D2F: case OP_DLOW: inv[(size_t)i] = v.path.empty() && opinv(v.a); break;
		}
	}

	// ---- stateful nodes of an op ----
	auto nodes_of = [&](const VOp& v, int out[2]) {
		int n = 0;
		switch (v.code) {
		case OP_OSC: case V_OSCARG: case OP_OSCSET: case OP_FREQ: case OP_LPF: case OP_LPFSET: case OP_ENV: case OP_OPERATOR: case OP_SETCTL: out[n++] = v.node; break;
		case OP_SMOOTH: out[n++] = v.node; if (in.ctlvar[v.imm & 7u] >= 0) out[n++] = in.ctlvar[v.imm & 7u]; break;
		case OP_CTL: if (in.ctlvar[v.imm & 7u] >= 0) out[n++] = in.ctlvar[v.imm & 7u]; break;
		case OP_PARAM: case OP_SETPARAM: if (written[(size_t)v.node]) out[n++] = v.node; break;
		}
		return n;
	};

	// ---- edges ----
	std::vector<std::vector<int>> succ((size_t)NV);
	std::vector<char> self_edge((size_t)NV, 0);
	auto edge = [&](int u, int v) { if (u < 0 || v < 0) return; if (u == v) { self_edge[(size_t)u] = 1; return; } succ[(size_t)u].push_back(v); };
	auto def = [&](int r) { const int d = (r >= 0 && (size_t)r < def_at.size()) ? def_at[(size_t)r] : -1; return (d >= 0 && inv[(size_t)d]) ? -1 : d; };   // (an invariant is nobody's predecessor)
	for (int i = 0; i < NV; i++) {
		const VOp& v = V[(size_t)i];
		if (is_struct(v.code) || inv[(size_t)i]) continue;
		bool ua = v.a >= 0, ub = v.b >= 0;
		if (v.code == OP_OPERATOR) { ua = v.a >= 0; ub = v.b >= 0; }
		if (ua) edge(def(v.a), i);
		if (ub) edge(def(v.b), i);
		for (const auto& pe : v.path) edge(def(V[(size_t)pe.first].a), i);                     // control: the conditions of every enclosing branch
		if (v.code == OP_PHI && phi_if[(size_t)i] >= 0) edge(def(V[(size_t)phi_if[(size_t)i]].a), i);
		if (v.code == OP_DELAYOUT && set_at[(size_t)v.node] >= 0) edge(set_at[(size_t)v.node], i);   // the head this process() walks
	}
	{
		std::vector<int> first_op(NN, -1), last_op(NN, -1);
		for (int i = 0; i < NV; i++) {
			if (is_struct(V[(size_t)i].code)) continue;
			int ns[2]; const int n = nodes_of(V[(size_t)i], ns);
			for (int q = 0; q < n; q++) {
				const size_t nd = (size_t)ns[q];
				if (last_op[nd] >= 0) edge(last_op[nd], i);
				if (first_op[nd] < 0) first_op[nd] = i;
				last_op[nd] = i;
			}
		}
		for (size_t nd = 0; nd < NN; nd++) if (first_op[nd] >= 0) edge(last_op[nd], first_op[nd]);     // the next sample (a lone op: a self edge)
	}

	// ---- strongly connected components (Tarjan, iterative); comp ids come out in reverse topological order ----
	std::vector<int> comp((size_t)NV, -1); int ncomp = 0;
	{
		std::vector<int> index((size_t)NV, -1), low((size_t)NV, 0), stack, it((size_t)NV, 0); std::vector<char> on((size_t)NV, 0); int counter = 0;
		for (int root = 0; root < NV; root++) {
			if (index[(size_t)root] >= 0 || is_struct(V[(size_t)root].code) || inv[(size_t)root]) continue;
			std::vector<int> call(1, root);
			index[(size_t)root] = low[(size_t)root] = counter++; stack.push_back(root); on[(size_t)root] = 1;
			while (!call.empty()) {
				const int u = call.back();
				if (it[(size_t)u] < (int)succ[(size_t)u].size()) {
					const int w = succ[(size_t)u][(size_t)it[(size_t)u]++];
					if (index[(size_t)w] < 0) { index[(size_t)w] = low[(size_t)w] = counter++; stack.push_back(w); on[(size_t)w] = 1; call.push_back(w); }
					else if (on[(size_t)w]) low[(size_t)u] = std::min(low[(size_t)u], index[(size_t)w]);
				}
				else {
					if (low[(size_t)u] == index[(size_t)u]) { int w; do { w = stack.back(); stack.pop_back(); on[(size_t)w] = 0; comp[(size_t)w] = ncomp; } while (w != u); ncomp++; }
					call.pop_back();
					if (!call.empty()) low[(size_t)call.back()] = std::min(low[(size_t)call.back()], low[(size_t)u]);
				}
			}
		}
	}
	std::vector<int> csize((size_t)ncomp, 0); std::vector<char> cserial((size_t)ncomp, 0);
	for (int i = 0; i < NV; i++) if (comp[(size_t)i] >= 0) { csize[(size_t)comp[(size_t)i]]++; if (self_edge[(size_t)i]) cserial[(size_t)comp[(size_t)i]] = 1; }
	for (int c = 0; c < ncomp; c++) if (csize[(size_t)c] > 1) cserial[(size_t)c] = 1;
	for (int i = 0; i < NV; i++) {
		const VOp& v = V[(size_t)i];
		if (comp[(size_t)i] < 0 || !cserial[(size_t)comp[(size_t)i]]) continue;
		if (v.code == OP_DELAYIN || v.code == OP_DELAYTAP || v.code == OP_DELAYOUT || v.code == OP_DELAYSET) return refuse("a Delay op inside a serial component (its time depends on state that depends on it)");
		if (v.code == OP_NOISE || v.code == OP_IN) return refuse("input inside a serial component");
	}

	// ---- levels (even: parallel, odd: serial) ----
	std::vector<int> level((size_t)NV, 0);
	std::vector<int> clevel((size_t)ncomp, -1);
	std::vector<std::vector<int>> pred((size_t)NV);
	for (int u = 0; u < NV; u++) for (int w : succ[(size_t)u]) pred[(size_t)w].push_back(u);
	std::vector<std::vector<int>> members((size_t)ncomp);
	for (int i = 0; i < NV; i++) if (comp[(size_t)i] >= 0) members[(size_t)comp[(size_t)i]].push_back(i);
	int guard_level = -2;
	for (int c = ncomp - 1; c >= 0; c--) {                                       // topological order
		const bool ser = cserial[(size_t)c] != 0;
		int lv = ser ? 1 : 0;
		for (int m : members[(size_t)c]) for (int p : pred[(size_t)m]) {
			const int pc = comp[(size_t)p]; if (pc == c) continue;
			const int pl = clevel[(size_t)pc]; const bool pser = cserial[(size_t)pc] != 0;
			lv = std::max(lv, (ser == pser) ? pl : pl + 1);
		}
		clevel[(size_t)c] = lv;
		for (int m : members[(size_t)c]) { level[(size_t)m] = lv; const int code = V[(size_t)m].code; if (code == OP_DELAYTAP || code == OP_DELAYOUT) guard_level = std::max(guard_level, lv); }
	}
	// ring writes only after every ring read of the chunk has been issued and checked
	for (int i = 0; i < NV; i++) if (V[(size_t)i].code == OP_DELAYIN) { level[(size_t)i] = std::max(level[(size_t)i], guard_level + 2); clevel[(size_t)comp[(size_t)i]] = level[(size_t)i]; }
	int max_level = 0;
	for (int i = 0; i < NV; i++) if (comp[(size_t)i] >= 0) max_level = std::max(max_level, level[(size_t)i]);
	const int out_level = (max_level + 1) & ~1;                                  // the level that writes `out` into the tile and commits the delay heads (parallel)
	const int last_level = std::max(out_level, guard_level + 2);
	// ---- serial components -> strands (those of one level that feed each other stay together) -> waves ----
	std::vector<int> strand((size_t)ncomp, -1);
	{
		std::vector<int> parent((size_t)ncomp); for (int c = 0; c < ncomp; c++) parent[(size_t)c] = c;
		std::function<int(int)> find = [&](int x) { while (parent[(size_t)x] != x) { parent[(size_t)x] = parent[(size_t)parent[(size_t)x]]; x = parent[(size_t)x]; } return x; };
		for (int u = 0; u < NV; u++) for (int w : succ[(size_t)u]) {
			const int a = comp[(size_t)u], b = comp[(size_t)w];
			if (a != b && cserial[(size_t)a] && cserial[(size_t)b] && clevel[(size_t)a] == clevel[(size_t)b]) parent[(size_t)find(a)] = find(b);
		}
		for (int c = 0; c < ncomp; c++) if (cserial[(size_t)c]) strand[(size_t)c] = find(c);
	}
	int G = in.G > 0 ? in.G : 16, C = in.C > 0 ? in.C : 32;
	const int CH = g.channels, NW = g.words();

	// the plan for a given chunk length: waves, slots, LDS bytes; the source is generated once the chunk length fits the LDS budget
	for (;; C /= 2) {
		if (C < 8) return refuse("the values that cross levels do not fit the LDS budget at any chunk length");
		const int NT = G * C, NWV = NT / 64;
		if (NT < 64 || NT > 1024 || (NT & 63)) return refuse("G x C must be 64 .. 1024 lanes");
		// strands of a level -> waves, heaviest first onto the lightest wave
		std::vector<int> wave_of((size_t)ncomp, 0);
		for (int lv = 1; lv <= max_level; lv += 2) {
			std::map<int, int> weight;
			for (int c = 0; c < ncomp; c++) if (cserial[(size_t)c] && clevel[(size_t)c] == lv) weight[strand[(size_t)c]] += csize[(size_t)c];
			std::vector<std::pair<int, int>> order; for (const auto& kv : weight) order.push_back({ kv.second, kv.first });
			std::sort(order.begin(), order.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first != y.first ? x.first > y.first : x.second < y.second; });
			std::vector<int> load((size_t)NWV, 0); std::map<int, int> wave_of_strand;
			for (const auto& o : order) { int best = 0; for (int w = 1; w < NWV; w++) if (load[(size_t)w] < load[(size_t)best]) best = w; load[(size_t)best] += o.first; wave_of_strand[o.second] = best; }
			for (int c = 0; c < ncomp; c++) if (cserial[(size_t)c] && clevel[(size_t)c] == lv) wave_of[(size_t)c] = wave_of_strand[strand[(size_t)c]];
		}
		auto ser_of = [&](int i) { return comp[(size_t)i] >= 0 && cserial[(size_t)comp[(size_t)i]] != 0; };
		auto wave_of_op = [&](int i) { return comp[(size_t)i] >= 0 ? wave_of[(size_t)comp[(size_t)i]] : -1; };
		// ---- where each register lives ----
		struct Reg { int def = -1; bool slot = false, chunk = false; int last = -1; int slot_id = -1; };
		std::vector<Reg> regs((size_t)maxreg + 2);
		auto use = [&](int r, int at_level, bool at_ser, int at_wave) {
			if (r < 0 || (size_t)r >= regs.size()) return;
			const int d = def_at[(size_t)r]; if (d < 0 || inv[(size_t)d]) return;  // (a prepare() register cannot be named here: Program::validate)
			Reg& R = regs[(size_t)r]; R.def = d;
			const int dl = level[(size_t)d]; const bool dser = ser_of(d);
			if (!dser && !at_ser) { if (at_level != dl) R.chunk = true; }
			else if (dser && at_ser && dl == at_level && wave_of_op(d) == at_wave) {}
			else { R.slot = true; R.last = std::max(R.last, at_level); }
		};
		for (int i = 0; i < NV; i++) {
			const VOp& v = V[(size_t)i];
			if (is_struct(v.code) || inv[(size_t)i]) continue;
			const bool s = ser_of(i); const int lv = level[(size_t)i], w = s ? wave_of_op(i) : -1;
			if (v.a >= 0) use(v.a, lv, s, w);
			if (v.b >= 0) use(v.b, lv, s, w);
			for (const auto& pe : v.path) use(V[(size_t)pe.first].a, lv, s, w);
			if (v.code == OP_PHI && phi_if[(size_t)i] >= 0) use(V[(size_t)phi_if[(size_t)i]].a, lv, s, w);
		}
		use(g.ret, out_level, false, -1);
		if (CH == 2) use(g.ret_r, out_level, false, -1);
		for (size_t r = 0; r < regs.size(); r++) if (regs[r].slot && is_dbl[r]) return refuse("a double register would have to cross levels");
		// slots: intervals [def level, last use level], reused when disjoint
		int nslots = 0;
		{
			std::vector<int> order;
			for (size_t r = 0; r < regs.size(); r++) if (regs[r].slot) order.push_back((int)r);
			std::sort(order.begin(), order.end(), [&](int x, int y) { const int lx = level[(size_t)regs[(size_t)x].def], ly = level[(size_t)regs[(size_t)y].def]; return lx != ly ? lx < ly : x < y; });
			std::vector<int> free_after;                                            // per slot: the level after which it is free
			for (int r : order) {
				const int dl = level[(size_t)regs[(size_t)r].def];
				int s = -1;
				for (size_t q = 0; q < free_after.size(); q++) if (free_after[q] < dl) { s = (int)q; break; }
				if (s < 0) { s = (int)free_after.size(); free_after.push_back(0); }
				free_after[(size_t)s] = regs[(size_t)r].last; regs[(size_t)r].slot_id = s;
			}
			nslots = (int)free_after.size();
		}
		const long long lds_words = (long long)NW * G + (long long)CH * C * G + (long long)nslots * C * G + 4;
		const char* le = getenv("KLG_FX_STAGED_LDS");
		const long long budget = le ? atoll(le) : 100 * 1024;                 // (gfx950 grants a workgroup up to 160 KB)
		if (lds_words * 4 > budget) { if (in.C > 0) return refuse("the requested chunk length does not fit the LDS budget"); continue; }

		// =========================================================== source ===========================================================
		std::string s;
		auto F = [](const char* f, ...) { char b[1024]; va_list ap; va_start(ap, f); vsnprintf(b, sizeof b, f, ap); va_end(ap); return std::string(b); };
		auto ring = [&](int node) { return F("Ring{ c.ring + (size_t)%lldll * 64, 64, %d }", (*in.ring_off)[(size_t)node], g.arg(node)); };
		auto ty = [&](int r) { return std::string(is_dbl[(size_t)r] ? "double" : "float"); };
		// registers that must be declared ahead of their defining statement at the level they are defined in (inside a branch there)
		auto in_branch = [&](int i) { return !V[(size_t)i].path.empty(); };
		// the statement(s) of virtual op i, in the context (L, c) it is emitted in
		auto op_text = [&](int i, bool assign) {
			const VOp& v = V[(size_t)i];
			std::string b;
			const std::string d = assign ? F("\t\tr%d = ", v.dst) : "\t\tconst " + ty(std::max(v.dst, 0)) + F(" r%d = ", v.dst);
			const int SZ = v.node >= 0 ? g.arg(v.node) : 0;
			auto pos = [&](int j) { return j ? F("ring_at(d%dp0, %d, %d)", v.node, j, SZ) : F("d%dp0", v.node); };
			switch (v.code) {
			case V_OSCARG: b += d + F("basic_sine_arg(L.n%d);\n", v.node); break;
			case V_OSCEVAL: b += d + F("basic_sine_of(r%d);\n", v.a); break;
			case OP_PHI: b += d + F("(r%d != 0.f) ? r%d : r%d;\n", V[(size_t)phi_if[(size_t)i]].a, v.a, v.b); break;
			case OP_DELAYIN: b += "\t\t{ const Ring q = " + ring(v.node) + "; q.wr(" + pos(in_index[(size_t)i]) + F(", r%d); }\n", v.a); break;
			case OP_DELAYSET: b += F("\t\td%dt = delay_set(", v.node) + pos(in_index[(size_t)i]) + F(", %d, r%d);\n", SZ, v.a); break;
			case OP_DELAYOUT:
				if (set_at[(size_t)v.node] >= 0) b += d + "staged_process(" + ring(v.node) + F(", ring_walk(d%dt.position, %d, %d), d%dt.fraction, d%dw, bad);\n", v.node, out_index[(size_t)i], SZ, v.node, v.node);
				else b += d + "staged_process(" + ring(v.node) + F(", ring_walk(d%dh.position, ps * %d + %d, %d), d%dh.fraction, d%dw, bad);\n", v.node, outs[(size_t)v.node], out_index[(size_t)i], SZ, v.node, v.node);
				break;
			case OP_DELAYTAP:
				if (v.imm == 1u) b += d + "staged_tap_int(" + ring(v.node) + ", " + pos(in_index[(size_t)i]) + F(", (int)r%d, d%dw, bad);\n", v.a, v.node);
				else b += d + (v.imm == 2u ? "staged_tap_stereo(" : "staged_tap_float(") + ring(v.node) + ", " + pos(in_index[(size_t)i]) + F(", r%d, d%dw, bad);\n", v.a, v.node);
				break;
			default: in.emit_op((size_t)v.orig, b, assign); break;
			}
			return b;
		};
		// ops of one (level, wave) in program order, with the branches they stand in re-opened around them
		auto emit_ops = [&](int lv, int wave /* -1: parallel */, const std::string& slot_index, std::string& out, const std::vector<char>& predeclared) {
			std::vector<std::pair<int, int>> open;
			auto close_to = [&](size_t keep) { while (open.size() > keep) { out += "\t\t}\n"; open.pop_back(); } };
			for (int i = 0; i < NV; i++) {
				const VOp& v = V[(size_t)i];
				if (is_struct(v.code) || inv[(size_t)i] || level[(size_t)i] != lv) continue;
				if ((wave >= 0) != ser_of(i) || (wave >= 0 && wave_of_op(i) != wave)) continue;
				size_t common = 0;
				while (common < open.size() && common < v.path.size() && open[common] == v.path[common]) common++;
				close_to(common);
				for (size_t q = common; q < v.path.size(); q++) { out += F("\t\tif (%s(r%d != 0.f)) {\n", v.path[q].second ? "!" : "", V[(size_t)v.path[q].first].a); open.push_back(v.path[q]); }
				const bool has_dst = v.dst >= 0 && v.code != OP_OSCSET && v.code != OP_LPFSET && v.code != OP_SETPARAM && v.code != OP_DELAYIN && v.code != OP_DELAYSET;
				const bool assign = has_dst && predeclared[(size_t)v.dst] != 0;
				out += op_text(i, assign);
				if (has_dst && regs[(size_t)v.dst].slot) out += F("\t\tSL(%d)[%s] = r%d;\n", regs[(size_t)v.dst].slot_id, slot_index.c_str(), v.dst);
			}
			close_to(0);
		};
		// registers an op of (level, wave) reads that come from elsewhere
		auto inputs_of = [&](int lv, int wave, std::vector<int>& from_slot) {
			std::vector<char> seen(regs.size(), 0);
			auto want = [&](int r) {
				if (r < 0 || (size_t)r >= regs.size() || seen[(size_t)r]) return;
				const int d = def_at[(size_t)r]; if (d < 0 || inv[(size_t)d]) return;
				const bool here = level[(size_t)d] == lv && ((wave >= 0) == ser_of(d)) && (wave < 0 || wave_of_op(d) == wave);
				if (here) return;
				if (wave < 0 && !ser_of(d)) return;                                   // parallel -> parallel: the lane's own register
				seen[(size_t)r] = 1; from_slot.push_back(r);
			};
			for (int i = 0; i < NV; i++) {
				const VOp& v = V[(size_t)i];
				if (is_struct(v.code) || inv[(size_t)i] || level[(size_t)i] != lv || (wave >= 0) != ser_of(i) || (wave >= 0 && wave_of_op(i) != wave)) continue;
				want(v.a); want(v.b);
				for (const auto& pe : v.path) want(V[(size_t)pe.first].a);
				if (v.code == OP_PHI) want(V[(size_t)phi_if[(size_t)i]].a);
			}
			if (wave < 0 && lv == out_level) { want(g.ret); if (CH == 2) want(g.ret_r); }
		};

		// the block invariants the ops of (level, wave) read, in program order
		auto inv_prelude = [&](int lv, int wave) {
			std::vector<char> need((size_t)NV, 0);
			auto want = [&](int r) { const int d = (r >= 0 && (size_t)r < def_at.size()) ? def_at[(size_t)r] : -1; if (d >= 0 && inv[(size_t)d]) need[(size_t)d] = 1; };
			for (int i = 0; i < NV; i++) {
				const VOp& v = V[(size_t)i];
				if (is_struct(v.code) || inv[(size_t)i] || level[(size_t)i] != lv || (wave >= 0) != ser_of(i) || (wave >= 0 && wave_of_op(i) != wave)) continue;
				want(v.a); want(v.b);
				for (const auto& pe : v.path) want(V[(size_t)pe.first].a);
				if (v.code == OP_PHI) want(V[(size_t)phi_if[(size_t)i]].a);
			}
			if (wave < 0 && lv == out_level) { want(g.ret); if (CH == 2) want(g.ret_r); }
			for (int i = NV - 1; i >= 0; i--) if (need[(size_t)i]) { want(V[(size_t)i].a); want(V[(size_t)i].b); }     // operands come earlier in program order
			std::string t;
			for (int i = 0; i < NV; i++) if (need[(size_t)i]) t += op_text(i, false);
			return t;
		};

		s += F("\n// ---- the staged form (klg_graph_staged.hpp): %d instances x %d samples per workgroup, %d levels, %d values through LDS ----\n", G, C, last_level + 1, nslots);
		s += "__device__ __forceinline__ int ring_at(int p0, int j, int size) { const int p = p0 + j; return p >= size ? p - size : p; }\n";
		s += "struct StagedWords { uint32_t* p; __device__ __forceinline__ uint32_t& operator[](int i) const { return p[i * " + std::to_string(G) + "]; } };\n";
		s += "struct StagedRec { StagedWords w; };\n";
		s += F("extern \"C\" __global__ __launch_bounds__(%d) void klg_fx_staged(const FxGraphArgs a) {\n", NT);
		s += F("\tconstexpr int G = %d, C = %d, NT = %d, NW = %d, CH = %d;\n", G, C, NT, NW, CH);
		s += "\tusing P = PatchGen;\n\textern __shared__ float lds[];\n";
		s += "\tuint32_t* const srec = reinterpret_cast<uint32_t*>(lds);                 // [NW][G]: the G records between chunks\n";
		s += "\tfloat* const tile = lds + NW * G;                                         // [CH][C][G]: the caller's block, chunk by chunk\n";
		s += "\tfloat* const slots = tile + CH * C * G;                                   // [slots][C][G]: values that cross levels\n";
		s += F("\tint* const flag = reinterpret_cast<int*>(slots + %d * C * G);\n", nslots);
		s += "#define SL(k) (slots + (k) * (C * G))\n";
		s += "\tconst int t = threadIdx.x, ps = t / G, pg = t % G, wv = t >> 6, ln = t & 63, k0 = blockIdx.x * G;\n";
		s += "\tconst int sg = ln < G ? ln : 0;                                           // the instance a lane of a serial level works for\n";
		s += "\tfor (int i = t; i < NW * G; i += NT) srec[i] = a.state[(size_t)(i / G) * a.kpad + k0 + (i % G)];\n";
		s += "\tFxCtx cp, cs;\n\tcp.fs = cs.fs = a.fs; cp.samples = cs.samples = a.samples;\n";
		s += "\tcp.ctl = a.controls + (size_t)(k0 + pg) * KLG_MAX_CTL; cs.ctl = a.controls + (size_t)(k0 + sg) * KLG_MAX_CTL;\n";
		s += "\tfloat* const ring0 = a.rings + (size_t)(k0 / 64) * a.ring_rows * 64 + (k0 % 64);\n\tcp.ring = ring0 + pg; cs.ring = ring0 + sg;\n";
		s += "\tcp.rand = a.rand ? a.rand + (size_t)(k0 + pg < a.K ? k0 + pg : 0) * (size_t)a.rand_per_instance : nullptr;\n";
		s += "\tcs.rand = a.rand ? a.rand + (size_t)(k0 + sg < a.K ? k0 + sg : 0) * (size_t)a.rand_per_instance : nullptr;\n";
		s += "\tP::Live Lp, Ls;\n\tLp.unused_ = 0; Lp.sidx = 0; Ls.unused_ = 0; Ls.sidx = 0;\n";
		s += "\t__syncthreads();\n";
		// the plain body over [from, from + count) of the block, on one lane per instance (wave 0): prepare() at the head of the block, chunks whose check failed, a ragged tail
		s += "\tauto plain = [&](int from, int count, bool with_prepare) {\n";
		s += "\t\tif (wv != 0 || ln >= G) return;\n\t\tP::Rec rec;\n#pragma unroll\n\t\tfor (int w = 0; w < NW; w++) rec.w[w] = srec[w * G + ln];\n";
		s += "\t\tP::Live L; FxCtx c = cs; c.samples = a.samples + (unsigned long long)from;\n";
		s += "\t\tif (with_prepare) P::begin(L, rec, c); else P::begin_core(L, rec, c);\n\t\tL.sidx = from;\n";
		s += "\t\tfor (int q = 0; q < count; q++) {\n\t\t\tconst float in0 = tile[(0 * C + q) * G + ln], in1 = CH > 1 ? tile[(1 * C + q) * G + ln] : 0.f;\n\t\t\tfloat out0 = 0.f, out1 = 0.f;\n";
		s += "\t\t\tP::sample(L, c, in0, in1, out0, out1);\n\t\t\ttile[(0 * C + q) * G + ln] = out0;\n\t\t\tif (CH > 1) tile[(1 * C + q) * G + ln] = out1;\n\t\t}\n";
		s += "\t\tP::end(L, rec);\n#pragma unroll\n\t\tfor (int w = 0; w < NW; w++) if (patch_stores<P>(w)) srec[w * G + ln] = rec.w[w];\n\t};\n";
		if (g.prepare_ops > 0) s += "\tplain(0, 0, true);                                                           // Effect::prepare(): once per block\n\t__syncthreads();\n";
		// what the lanes hold for the whole block: the dials, the members process() only reads
		{
			std::string pb, sb;
			for (size_t nd = 0; nd < NN; nd++) if (g.nodes[nd] == N_PARAM && !written[nd]) pb += (*in.node_begin)[nd];
			s += "\t{ auto& L = Lp; const FxCtx& c = cp; const StagedRec r = { { srec + pg } }; (void)L; (void)c; (void)r;\n" + in.ctl_begin + pb + "\t}\n";
			s += "\t{ auto& L = Ls; const FxCtx& c = cs; const StagedRec r = { { srec + sg } }; (void)L; (void)c; (void)r;\n" + in.ctl_begin + pb + "\t}\n";
		}
		s += "\tfor (int s0 = 0; s0 < a.n; s0 += C) {\n\t\tconst int cl = (a.n - s0 < C) ? (a.n - s0) : C;\n";
		// the caller's rows of this chunk -> tile (a row's C samples are contiguous in the caller's block)
		s += "\t\tfor (int i = t; i < G * CH * C; i += NT) { const int row = i / C, q = i % C, gi = row / CH, ch = row % CH;\n";
		s += "\t\t\ttile[(ch * C + q) * G + gi] = (q < cl && k0 + gi < a.K) ? a.io[((size_t)(k0 + gi) * CH + ch) * a.n + s0 + q] : 0.f; }\n";
		s += "\t\tif (t == 0) *flag = 0;\n\t\t__syncthreads();\n";
		s += "\t\tbool ok = cl == C;\n\t\tint bad = 0; (void)bad;\n";
		s += "\t\tconst float in0 = tile[(0 * C + ps) * G + pg], in1 = CH > 1 ? tile[(1 * C + ps) * G + pg] : 0.f; (void)in0; (void)in1;\n";
		s += "\t\tLp.sidx = s0 + ps;\n";
		// delay lines: this lane's cursor at the start of its sample, the rows the chunk writes, the head
		for (size_t nd = 0; nd < NN; nd++) if (g.nodes[nd] == N_DELAY) {
			const int SZ = g.arg((int)nd), w0 = g.node_word0((int)nd);
			s += F("\t\tconst int d%zup0 = (int)(((a.samples + (unsigned long long)(s0 + ps)) * %dull) %% %dull); (void)d%zup0;\n", nd, k_in[nd], SZ, nd);
			s += F("\t\tconst RingWindow d%zuw = { (int)(((a.samples + (unsigned long long)s0) * %dull) %% %dull), %d * C }; (void)d%zuw;\n", nd, k_in[nd], SZ, k_in[nd], nd);
			if (set_at[nd] >= 0) s += F("\t\tTap d%zut = { 0, 0.f }; (void)d%zut;\n", nd, nd);
			else if (outs[nd] > 0) s += F("\t\tconst Tap d%zuh = { (int)srec[%d * G + pg], u2f(srec[%d * G + pg]) };\n", nd, w0 + ED_LASTPOS, w0 + ED_LASTFRAC);
		}
		// registers that live across levels of a lane
		for (size_t r = 0; r < regs.size(); r++) if (regs[r].def >= 0 && regs[r].chunk) s += "\t\t" + ty((int)r) + F(" r%zu = 0; (void)r%zu;\n", r, r);
		std::vector<std::string> deferred;                                          // commits of serial levels at or below the guard
		bool guard_emitted = guard_level < 0;
		for (int lv = 0; lv <= last_level; lv++) {
			std::string code;
			if (!(lv & 1)) {                                                           // ---- a parallel level ----
				std::vector<int> from_slot; inputs_of(lv, -1, from_slot);
				std::vector<char> pre(regs.size(), 0);
				std::string decl, body;
				for (int r : from_slot) decl += "\t\tconst float " + F("r%d = SL(%d)[t];\n", r, regs[(size_t)r].slot_id);
				decl += inv_prelude(lv, -1);
				for (int i = 0; i < NV; i++) if (!is_struct(V[(size_t)i].code) && !inv[(size_t)i] && level[(size_t)i] == lv && !ser_of(i) && V[(size_t)i].dst >= 0 && (size_t)V[(size_t)i].dst < regs.size()) {
					const int r = V[(size_t)i].dst;
					if (regs[(size_t)r].chunk && regs[(size_t)r].def == i) pre[(size_t)r] = 1;
					else if (in_branch(i) && def_at[(size_t)r] == i) { pre[(size_t)r] = 1; decl += "\t\t" + ty(r) + F(" r%d = 0; (void)r%d;\n", r, r); }
				}
				emit_ops(lv, -1, "t", body, pre);
				if (lv > guard_level && !deferred.empty()) { for (const std::string& d : deferred) body += d; deferred.clear(); }
				if (lv == guard_level) body += "\t\tif (bad) *flag = 1;\n";
				if (lv == out_level) {
					body += F("\t\ttile[(0 * C + ps) * G + pg] = r%d;\n", g.ret);
					if (CH == 2) body += F("\t\ttile[(1 * C + ps) * G + pg] = r%d;\n", g.ret_r);
				}
				if (lv == last_level) {                                                  // the read heads as the chunk leaves them (Delay::last)
					for (size_t nd = 0; nd < NN; nd++) if (g.nodes[nd] == N_DELAY && head_used[nd]) {
						const int SZ = g.arg((int)nd), w0 = g.node_word0((int)nd);
						if (set_at[nd] >= 0) body += F("\t\tif (ps == C - 1) { srec[%d * G + pg] = (uint32_t)ring_walk(d%zut.position, %d, %d); srec[%d * G + pg] = f2u(d%zut.fraction); }\n", w0 + ED_LASTPOS, nd, outs[nd], SZ, w0 + ED_LASTFRAC, nd);
						else if (outs[nd] > 0) body += F("\t\tif (ps == 0) srec[%d * G + pg] = (uint32_t)ring_walk(d%zuh.position, C * %d, %d);\n", w0 + ED_LASTPOS, nd, outs[nd], SZ);
					}
				}
				if (!decl.empty() || !body.empty()) code += "\t\tif (ok) { auto& L = Lp; const FxCtx& c = cp; (void)L; (void)c;\n" + decl + body + "\t\t}\n";
			}
			else {                                                                     // ---- a serial level: one loop per wave that has work ----
				for (int w = 0; w < NWV; w++) {
					std::vector<int> mine;
					for (int i = 0; i < NV; i++) if (!is_struct(V[(size_t)i].code) && !inv[(size_t)i] && level[(size_t)i] == lv && ser_of(i) && wave_of_op(i) == w) mine.push_back(i);
					if (mine.empty()) continue;
					std::vector<char> node_here(NN, 0);
					for (int i : mine) { int ns[2]; const int n = nodes_of(V[(size_t)i], ns); for (int q = 0; q < n; q++) node_here[(size_t)ns[q]] = 1; }
					std::string load, commit, loop, decl;
					for (size_t nd = 0; nd < NN; nd++) if (node_here[nd]) { load += (*in.node_begin)[nd]; commit += (*in.node_end)[nd]; }
					std::vector<int> from_slot; inputs_of(lv, w, from_slot);
					std::vector<char> pre(regs.size(), 0);
					for (int r : from_slot) decl += "\t\tconst " + ty(r) + F(" r%d = SL(%d)[q];\n", r, regs[(size_t)r].slot_id);
					for (int i : mine) if (V[(size_t)i].dst >= 0 && in_branch(i) && def_at[(size_t)V[(size_t)i].dst] == i) { const int r = V[(size_t)i].dst; pre[(size_t)r] = 1; decl += "\t\t" + ty(r) + F(" r%d = 0; (void)r%d;\n", r, r); }
					emit_ops(lv, w, "q", loop, pre);
					P.serial_ops += (int)mine.size();
					code += F("\t\tif (ok && wv == %d && ln < G) { auto& L = Ls; const FxCtx& c = cs; const StagedRec r = { { srec + ln } }; (void)L; (void)c; (void)r;\n", w) + load + inv_prelude(lv, w);
					code += "\t\tfor (int sq = 0; sq < C; sq++) { const int q = sq * G + ln; (void)q;\n" + decl + loop + "\t\t}\n";
					const std::string cm = F("\t\tif (ok && wv == %d && ln < G) { auto& L = Ls; StagedRec r = { { srec + ln } }; (void)L; (void)r;\n", w) + commit + "\t\t}\n";
					if (lv > guard_level) code += commit + "\t\t}\n"; else { code += "\t\t}\n"; deferred.push_back(cm); }
				}
			}
			if (code.empty() && !(lv == guard_level)) continue;
			s += F("\t\t// ---- level %d (%s) ----\n", lv, (lv & 1) ? "serial" : "parallel") + code;
			if (lv < last_level || true) s += "\t\t__syncthreads();\n";
			if (lv == guard_level && !guard_emitted) { s += "\t\tok = ok && *flag == 0;                                                   // every ring read of the chunk lies outside the rows the chunk writes\n"; guard_emitted = true; }
		}
		if (!deferred.empty()) { for (const std::string& d : deferred) s += d; s += "\t\t__syncthreads();\n"; }
		s += "\t\tif (!ok) { plain(s0, cl, false); __syncthreads(); }\n";
		s += "\t\tfor (int i = t; i < G * CH * C; i += NT) { const int row = i / C, q = i % C, gi = row / CH, ch = row % CH;\n";
		s += "\t\t\tif (q < cl && k0 + gi < a.K) a.io[((size_t)(k0 + gi) * CH + ch) * a.n + s0 + q] = tile[(ch * C + q) * G + gi]; }\n";
		s += "\t\t__syncthreads();\n\t}\n";
		s += "\tfor (int i = t; i < NW * G; i += NT) if (k0 + (i % G) < a.K && patch_stores<P>(i / G)) a.state[(size_t)(i / G) * a.kpad + k0 + (i % G)] = srec[i];\n";
		s += "#undef SL\n}\n";
		for (int i = 0; i < NV; i++) if (!is_struct(V[(size_t)i].code) && !inv[(size_t)i] && !ser_of(i)) P.parallel_ops++;
		P.ok = true; P.G = G; P.C = C; P.threads = NT; P.lds_bytes = (int)(lds_words * 4); P.levels = last_level + 1; P.slots = nslots; P.source = s;
		return P;
	}
}

} }
