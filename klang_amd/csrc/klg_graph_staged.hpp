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
	int G = 16, C = 32, threads = 0, lds_bytes = 0, levels = 0, slots = 0, serial_ops = 0, parallel_ops = 0, retry_min = 0; bool pipelined = false;   // retry_min: the shortest part a chunk that failed its ring check is tried again in (0: never — straight to the plain body)
	std::string prefix_commit;                                           // (scratch of plan_staged)
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
	bool C_is_a_preference = false;                                      // C: where the search for a chunk length that fits the LDS budget starts (else: that length or none)
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
	// ... and a biquad's set(f, Q) into what it computes from (f, Q) alone — the cached Q, five coefficients: pure, V_LPFQ / V_LPFCOEF — and what it does to the
	// filter (V_LPFAPPLY: compare with the cached pair, assign): a cutoff swept per sample (WahWah.k) leaves its cosf / sinf / divisions out of the filter's loop
	enum { V_OSCARG = OP_CODES + 1, V_OSCEVAL, V_LPFQ, V_LPFCOEF, V_LPFAPPLY };
	struct VOp { int code, dst, a, b, node; uint32_t imm; int orig; std::vector<std::pair<int, int>> path; int parent_if = -1; std::vector<int> x; };   // x: further operands (V_LPFAPPLY: the five coefficients)   // path: (vop index of the `if`, side 0 then / 1 else), outermost first
	std::vector<VOp> V;
	int maxreg = -1;
	for (const Op& o : g.ops) maxreg = std::max(maxreg, std::max(o.dst, std::max(o.a, o.b)));
	std::vector<char> is_dbl((size_t)maxreg + 2 + 8 * (size_t)N, 0);
	for (const Op& o : g.ops) if (o.code == OP_F2D || o.code == OP_DCONST || o.code == OP_DLOW || o.code == OP_FUNC || (o.code >= OP_DADD && o.code <= OP_DDIV)) is_dbl[(size_t)o.dst] = 1;
	std::vector<bool> written(g.nodes.size(), false);
	for (int i = first; i < N; i++) if (g.ops[(size_t)i].code == OP_SETPARAM) written[(size_t)g.ops[(size_t)i].node] = true;
	// (which registers are block invariants, on the recorded ops: a set(f, Q) of dials only stays one op — its test against the cached pair is all a sample costs)
	std::vector<char> reg_inv((size_t)maxreg + 2, 0);
	{
		int depth = 0;
		auto ri = [&](int r) { return r >= 0 && reg_inv[(size_t)r] != 0; };
		for (int i = first; i < N; i++) {
			const Op& o = g.ops[(size_t)i];
			if (o.code == OP_IF) depth++; else if (o.code == OP_ENDIF) depth--;
			if (o.dst < 0) continue;
			switch (o.code) {
			case OP_CONST: case OP_DCONST: reg_inv[(size_t)o.dst] = 1; break;
			case OP_CTL: reg_inv[(size_t)o.dst] = in.ctlvar[o.imm & 0xFFu] < 0; break;
			case OP_PARAM: reg_inv[(size_t)o.dst] = !written[(size_t)o.node]; break;
			case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: case OP_CMP: case OP_DADD: case OP_DSUB: case OP_DMUL: case OP_DDIV: reg_inv[(size_t)o.dst] = depth == 0 && ri(o.a) && ri(o.b); break;
			case OP_NEG: case OP_ABS: case OP_TRUNC: case OP_POWC: case OP_FUNC: case OP_F2D: case OP_D2F: case OP_DLOW: reg_inv[(size_t)o.dst] = depth == 0 && ri(o.a); break;
			default: break;
			}
		}
	}
	{
		std::vector<std::pair<int, int>> path;
		for (int i = first; i < N; i++) {
			const Op& o = g.ops[(size_t)i];
			if (o.code == OP_STOP || o.code == OP_STOPIF || o.code == OP_TABREAD || o.code == OP_ENVOFF) return refuse("op without a staged form");
			if ((o.code == OP_ENV || o.code == OP_OPERATOR) && o.node >= 0 && env_capacity(g.arg(o.node)) > 4) return refuse("an Envelope with more than four point slots has no staged form");
			VOp v; v.code = o.code; v.dst = o.dst; v.a = o.a; v.b = o.b; v.node = o.node; v.imm = o.imm; v.orig = i;
			if (o.code == OP_ELSE) { if (path.empty()) return refuse("unbalanced else"); path.back().second = 1; v.path = path; V.push_back(v); continue; }
			if (o.code == OP_ENDIF) { if (path.empty()) return refuse("unbalanced endif"); path.pop_back(); v.path = path; V.push_back(v); continue; }
			v.path = path;
			if (o.code == OP_OSC && o.node >= 0 && g.nodes[(size_t)o.node] == N_BSINE) {
				VOp arg = v; arg.code = V_OSCARG; arg.dst = ++maxreg; V.push_back(arg);
				v.code = V_OSCEVAL; v.a = arg.dst; v.node = -1; V.push_back(v);
				continue;
			}
			if (o.code == OP_LPFSET && o.node >= 0 && !(o.a >= 0 && o.b >= 0 && reg_inv[(size_t)o.a] && reg_inv[(size_t)o.b])) {
				VOp q = v; q.code = V_LPFQ; q.node = -1; q.dst = ++maxreg; V.push_back(q);
				VOp ap = v; ap.code = V_LPFAPPLY; ap.b = q.dst; ap.dst = -1;
				for (int k = 0; k < 5; k++) { VOp c = v; c.code = V_LPFCOEF; c.node = -1; c.b = q.dst; c.dst = ++maxreg; c.imm = o.imm | ((uint32_t)k << 8); V.push_back(c); ap.x.push_back(c.dst); }
				V.push_back(ap);
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
	// lines that take the effect's own input (`in >> delay`, one input() per sample): what a tap may read of its own chunk is the chunk's `in`, known up front —
	// such a tap needs no check and never sends its chunk to the plain body (klg_delay.hpp staged_tap_float_fetch_near): near_ch[line] = the channel, or -1
	std::vector<int> near_ch(NN, -1);
	bool any_near = false;
	{
		const char* e = getenv("KLG_FX_STAGED_NEAR");
		if (!(e && e[0] == '0')) for (int i = 0; i < NV; i++) {
			const VOp& v = V[(size_t)i];
			if (v.code != OP_DELAYIN || v.node < 0 || k_in[(size_t)v.node] != 1) continue;
			const int d = (v.a >= 0 && (size_t)v.a < def_at.size()) ? def_at[(size_t)v.a] : -1;
			if (d >= 0 && V[(size_t)d].code == OP_IN && V[(size_t)d].path.empty()) near_ch[(size_t)v.node] = (int)V[(size_t)d].imm;
		}
		for (int i = 0; i < NV; i++) { const VOp& v = V[(size_t)i]; if (v.code == OP_DELAYTAP && v.imm == 0u && v.node >= 0 && near_ch[(size_t)v.node] >= 0) any_near = true; }
	}

	// ---- block invariants: literals, dials, members process() only reads, and plain arithmetic on those outside any branch.  They belong to no level: whoever
	// needs one computes it (a lane of a parallel level once per chunk, a serial loop in front of its samples) — nothing of them travels through LDS ----
	std::vector<char> inv((size_t)NV, 0);
	for (int i = 0; i < NV; i++) {
		const VOp& v = V[(size_t)i];
		auto opinv = [&](int r) { const int d = (r >= 0 && (size_t)r < def_at.size()) ? def_at[(size_t)r] : -1; return d >= 0 && d < i && inv[(size_t)d]; };
		switch (v.code) {
		case OP_CONST: case OP_DCONST: inv[(size_t)i] = 1; break;
		case OP_CTL: inv[(size_t)i] = in.ctlvar[v.imm & 0xFFu] < 0; break;
		case OP_PARAM: inv[(size_t)i] = !written[(size_t)v.node]; break;
		case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: case OP_CMP: case OP_DADD: case OP_DSUB: case OP_DMUL: case OP_DDIV: inv[(size_t)i] = v.path.empty() && opinv(v.a) && opinv(v.b); break;
		case OP_NEG: case OP_ABS: case OP_TRUNC: case OP_POWC: case OP_FUNC: case OP_F2D: case OP_D2F: case OP_DLOW: inv[(size_t)i] = v.path.empty() && opinv(v.a); break;
		case V_LPFQ: case V_LPFCOEF: inv[(size_t)i] = v.path.empty() && opinv(v.a) && opinv(v.b); break;   // (dials only: once per chunk by whoever needs them)
		}
	}

	// ---- stateful nodes of an op ----
	auto nodes_of = [&](const VOp& v, int out[2]) {
		int n = 0;
		switch (v.code) {
		case OP_OSC: case V_OSCARG: case OP_OSCSET: case OP_FREQ: case OP_LPF: case OP_LPFSET: case V_LPFAPPLY: case OP_ENV: case OP_OPERATOR: case OP_SETCTL: out[n++] = v.node; break;
		case OP_SMOOTH: out[n++] = v.node; if (in.ctlvar[v.imm & 0xFFu] >= 0) out[n++] = in.ctlvar[v.imm & 0xFFu]; break;
		case OP_CTL: if (in.ctlvar[v.imm & 0xFFu] >= 0) out[n++] = in.ctlvar[v.imm & 0xFFu]; break;
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
		for (int r : v.x) edge(def(r), i);
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

	// ---- the control path runs a chunk AHEAD ----
	// Ops that nothing of the chunk's audio reaches — no ring read, no `in`, no input() upstream of them: dial smoothers, LFOs, the delay times they set — form the
	// PREFIX; the rest is the SUFFIX.  While the suffix levels of chunk c run, the prefix levels of chunk c + 1 run beside them (serial components on other waves,
	// parallel ones in the same lanes): a chunk then costs the longer of the two level sequences, not their sum.  No rollback is needed when chunk c fails its ring
	// check and is walked by the plain body: the prefix of chunk c + 1 depends on nothing that walk computes differently (it is the same arithmetic on the same
	// state), so what was computed ahead stays valid.  The prefix keeps its own copy of the records (end of chunk x in copy x & 1), handed to the architectural
	// copy when chunk x is complete.
	std::vector<std::vector<int>> pred((size_t)NV);
	for (int u = 0; u < NV; u++) for (int w : succ[(size_t)u]) pred[(size_t)w].push_back(u);
	std::vector<char> pfx((size_t)NV, 0);
	bool pipelined = false;
	{
		std::vector<char> sfx((size_t)NV, 0); std::vector<int> work;
		for (int i = 0; i < NV; i++) { const int c = V[(size_t)i].code; if (comp[(size_t)i] >= 0 && (c == OP_DELAYOUT || c == OP_DELAYTAP || c == OP_DELAYIN || c == OP_IN)) { sfx[(size_t)i] = 1; work.push_back(i); } }
		while (!work.empty()) { const int u = work.back(); work.pop_back(); for (int w : succ[(size_t)u]) if (!sfx[(size_t)w]) { sfx[(size_t)w] = 1; work.push_back(w); } }
		bool serial_prefix = false, any_suffix = false;
		for (int i = 0; i < NV; i++) if (comp[(size_t)i] >= 0) { if (!sfx[(size_t)i]) { pfx[(size_t)i] = 1; if (cserial[(size_t)comp[(size_t)i]]) serial_prefix = true; } else any_suffix = true; }
		const char* pe = getenv("KLG_FX_STAGED_PIPE");
		pipelined = serial_prefix && any_suffix && !(pe && pe[0] == '0');
		if (!pipelined) std::fill(pfx.begin(), pfx.end(), 0);
	}
	auto cpfx = [&](int c, const std::vector<std::vector<int>>& members) { return !members[(size_t)c].empty() && pfx[(size_t)members[(size_t)c][0]]; };

	// ---- levels (even: parallel, odd: serial), counted inside each of the two groups ----
	std::vector<int> level((size_t)NV, 0);
	std::vector<int> clevel((size_t)ncomp, -1);
	std::vector<std::vector<int>> members((size_t)ncomp);
	for (int i = 0; i < NV; i++) if (comp[(size_t)i] >= 0) members[(size_t)comp[(size_t)i]].push_back(i);
	int guard_level = -2, pmax = -1;
	for (int c = ncomp - 1; c >= 0; c--) {                                       // topological order
		const bool ser = cserial[(size_t)c] != 0, pf = cpfx(c, members);
		int lv = ser ? 1 : 0;
		for (int m : members[(size_t)c]) for (int p : pred[(size_t)m]) {
			const int pc = comp[(size_t)p]; if (pc == c || cpfx(pc, members) != pf) continue;   // (what the prefix hands to the suffix was computed an iteration earlier)
			const int pl = clevel[(size_t)pc]; const bool pser = cserial[(size_t)pc] != 0;
			lv = std::max(lv, (ser == pser) ? pl : pl + 1);
		}
		clevel[(size_t)c] = lv;
		if (pf) pmax = std::max(pmax, lv);
		for (int m : members[(size_t)c]) { level[(size_t)m] = lv; const int code = V[(size_t)m].code; if (code == OP_DELAYTAP || code == OP_DELAYOUT) guard_level = std::max(guard_level, lv); }
	}
	// ring writes only after every ring read of the chunk has been issued and checked
	for (int i = 0; i < NV; i++) if (V[(size_t)i].code == OP_DELAYIN) { level[(size_t)i] = std::max(level[(size_t)i], guard_level + 2); clevel[(size_t)comp[(size_t)i]] = level[(size_t)i]; }
	int max_level = 0;
	for (int i = 0; i < NV; i++) if (comp[(size_t)i] >= 0 && !pfx[(size_t)i]) max_level = std::max(max_level, level[(size_t)i]);
	const int out_level = (max_level + 1) & ~1;                                  // the level that writes `out` into the tile and commits the delay heads (parallel)
	const int last_level = std::max(out_level, guard_level + 2);
	// ---- serial components -> strands (those of one level and group that feed each other stay together) -> waves ----
	std::vector<int> strand((size_t)ncomp, -1);
	{
		std::vector<int> parent((size_t)ncomp); for (int c = 0; c < ncomp; c++) parent[(size_t)c] = c;
		std::function<int(int)> find = [&](int x) { while (parent[(size_t)x] != x) { parent[(size_t)x] = parent[(size_t)parent[(size_t)x]]; x = parent[(size_t)x]; } return x; };
		for (int u = 0; u < NV; u++) for (int w : succ[(size_t)u]) {
			const int a = comp[(size_t)u], b = comp[(size_t)w];
			if (a != b && cserial[(size_t)a] && cserial[(size_t)b] && clevel[(size_t)a] == clevel[(size_t)b] && cpfx(a, members) == cpfx(b, members)) parent[(size_t)find(a)] = find(b);
		}
		for (int c = 0; c < ncomp; c++) if (cserial[(size_t)c]) strand[(size_t)c] = find(c);
	}
	int G = in.G > 0 ? in.G : 16, C = in.C > 0 ? in.C : 32;
	const int CH = g.channels, NW = g.words();

	// the plan for a given chunk length: waves, slots, LDS bytes; the source is generated once the chunk length fits the LDS budget
	for (;; C /= 2) {
		if (C < 8) return refuse("the values that cross levels do not fit the LDS budget at any chunk length");
		// The serial loops get waves of their own (NSW of them, after the NTP = G x C lanes of the parallel levels): a loop of the control path then runs BESIDE the
		// audio path's parallel level of the same interval — which mostly waits for its ring rows — instead of after it on one of its waves.
		// In interval k the audio path runs its level k and the control path (of the next chunk) its level k + poff: poff = 1 when the control path has nothing on
		// level 0 (the usual case: its level 0 would be arithmetic on dials, and that is invariant), so that its serial levels meet the audio path's parallel ones.
		int poff = 0;
		if (pipelined) { bool l0 = false; for (int i = 0; i < NV; i++) if (comp[(size_t)i] >= 0 && pfx[(size_t)i] && level[(size_t)i] == 0) l0 = true; poff = l0 ? 0 : 1; }
		const int NTP = G * C;
		if (NTP < 64 || NTP > 1024 || (NTP & 63)) return refuse("G x C must be 64 .. 1024 lanes");
		// RingS (klg_delay.hpp) addresses a line's rows with 32-bit byte offsets from a wave-uniform base: (SIZE + 1) rows of G floats must stay below 4 GB
		for (size_t d = 0; d < NN; d++) if (g.nodes[d] == N_DELAY && ((unsigned long long)g.arg((int)d) + 1ull) * (unsigned long long)G * 4ull >= (1ull << 32)) return refuse("a Delay line of 2^32 bytes or more per workgroup (32-bit row offsets)");
		int NSW = 1;
		for (int lv = 1; lv <= std::max(max_level, pmax); lv += 2) for (int grp = 0; grp < 2; grp++) {
			std::vector<int> st; for (int c = 0; c < ncomp; c++) if (cserial[(size_t)c] && clevel[(size_t)c] == lv && (cpfx(c, members) ? 1 : 0) == grp) st.push_back(strand[(size_t)c]);
			std::sort(st.begin(), st.end()); st.erase(std::unique(st.begin(), st.end()), st.end());
			NSW = std::max(NSW, (int)st.size());
		}
		if (const char* e = getenv("KLG_FX_STAGED_SW")) NSW = std::max(1, atoi(e));
		NSW = std::min(NSW, std::min(4, (1024 - NTP) / 64));
		// (only a pipelined plan has anything to run beside a serial loop: without one the loops take the waves of the parallel levels — more waves would only cost
		// registers: the recorded Reverb.k with 8 + 4 waves per workgroup spilled and ran 1.7 x slower than with 4)
		const bool own_waves = pipelined && NSW >= 1;
		const int NT = own_waves ? NTP + 64 * NSW : NTP, NWV = own_waves ? NSW : NTP / 64;
		// strands of a level -> serial waves, heaviest first onto the lightest wave; where both groups have loops in one interval the audio path takes the lower half
		auto ser_of_comp = [&](int c) { return c >= 0 && cserial[(size_t)c] != 0; };
		auto F = [](const char* f, ...) { char b[1024]; va_list ap; va_start(ap, f); vsnprintf(b, sizeof b, f, ap); va_end(ap); return std::string(b); };
		std::vector<int> wave_of((size_t)ncomp, 0);
		// PACKS.  A serial loop runs on G of a wave's 64 lanes, and an instruction costs its four cycles whatever the lanes: strands of one level that are the SAME
		// code on different nodes (Reverb.k's sixteen damping filters, Chorus.k's ten LFOs, an equaliser's bands) run as ONE loop with 64 / G of them side by side
		// in the lanes — lane = (strand of the pack, instance) —: the text is the first strand's, what differs (the nodes' record words, the LDS slots of what
		// comes in and goes out) is chosen by the lane's quarter.  A strand qualifies when it has no branch, no invariant operand, and nothing per-instance but
		// its nodes (no control, no Noise).  op_sub: -1 a strand on its own, p >= 0 the first strand of pack p (what is emitted), -2 the others.
		struct Pack { int lv = 0; bool pf = false; int wave = 0; std::vector<std::vector<int>> ops; };
		std::vector<Pack> packs;
		std::vector<int> op_sub((size_t)NV, -1);
		const int QPACK = []() { const char* e = getenv("KLG_FX_STAGED_PACK"); return !(e && e[0] == '0'); }() ? 64 / G : 1;
		for (int lv = 1; lv <= std::max(max_level, pmax); lv += 2) {
			for (int grp = 0; grp < 2; grp++) {
				bool mine = false, other = false;                                      // the other group's serial level of the same interval
				const int olv = grp ? lv - poff : lv + poff;
				for (int c = 0; c < ncomp; c++) if (cserial[(size_t)c]) { const int cg = cpfx(c, members) ? 1 : 0; if (cg == grp && clevel[(size_t)c] == lv) mine = true; if (pipelined && cg != grp && clevel[(size_t)c] == olv) other = true; }
				if (!mine) continue;
				const bool split = other && NWV >= 2;
				const int w0 = (split && grp == 1) ? NWV / 2 : 0, w1 = (split && grp == 0) ? NWV / 2 : NWV;
				std::map<int, int> weight;
				for (int c = 0; c < ncomp; c++) if (cserial[(size_t)c] && clevel[(size_t)c] == lv && (cpfx(c, members) ? 1 : 0) == grp) weight[strand[(size_t)c]] += csize[(size_t)c];
				// packs of this (level, group): strands with the same signature, 64 / G at a time
				std::map<int, int> unit_of;                                            // strand -> the strand whose wave it shares (its pack's first)
				std::vector<size_t> new_packs;
				if (QPACK >= 2) {
					std::map<int, std::vector<int>> sops;
					for (int i = 0; i < NV; i++) { const int c = comp[(size_t)i]; if (c >= 0 && cserial[(size_t)c] && clevel[(size_t)c] == lv && (cpfx(c, members) ? 1 : 0) == grp && !is_struct(V[(size_t)i].code) && !inv[(size_t)i]) sops[strand[(size_t)c]].push_back(i); }
					std::map<std::string, std::vector<int>> by_sig;
					for (const auto& kv : sops) {
						const std::vector<int>& ops = kv.second;
						std::map<int, int> posof; for (size_t q = 0; q < ops.size(); q++) posof[ops[q]] = (int)q;
						std::string sig; bool ok = true;
						for (int i : ops) {
							const VOp& v = V[(size_t)i];
							switch (v.code) {
							case OP_OSC: case V_OSCARG: case OP_OSCSET: case OP_LPF: case V_LPFAPPLY: case OP_ENV: case OP_OPERATOR: case OP_PARAM: case OP_SETPARAM: case OP_FREQ:
							case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: case OP_NEG: case OP_ABS: case OP_TRUNC: case OP_POWC: case OP_FUNC: case OP_CMP: break;
							default: ok = false;
							}
							if (!v.path.empty()) ok = false;
							if (!ok) break;
							sig += F("|%d,%u,%d,%d,%d", v.code, v.imm, v.node >= 0 ? g.nodes[(size_t)v.node] : -1, v.node >= 0 ? g.arg(v.node) : 0, v.dst >= 0 ? 1 : 0);
							std::vector<int> rs = { v.a, v.b }; rs.insert(rs.end(), v.x.begin(), v.x.end());
							for (int r : rs) {
								if (r < 0) { sig += ",-"; continue; }
								const int d = (size_t)r < def_at.size() ? def_at[(size_t)r] : -1;
								if (d < 0 || inv[(size_t)d] || is_dbl[(size_t)r]) { ok = false; break; }
								const auto it = posof.find(d);
								if (it != posof.end()) sig += F(",i%d", it->second);
								else sig += F(",e%d%d", pfx[(size_t)d] ? 1 : 0, ser_of_comp(comp[(size_t)d]) ? 1 : 0);
							}
						}
						if (ok && !ops.empty()) by_sig[sig].push_back(kv.first);
					}
					for (const auto& kv : by_sig) {
						const std::vector<int>& st = kv.second;
						for (size_t b0 = 0; b0 + 1 < st.size(); b0 += (size_t)QPACK) {
							const size_t n = std::min((size_t)QPACK, st.size() - b0);
							if (n < 2) break;
							Pack pk; pk.lv = lv; pk.pf = grp != 0;
							for (size_t q = 0; q < n; q++) { pk.ops.push_back(sops[st[b0 + q]]); unit_of[st[b0 + q]] = st[b0]; }
							new_packs.push_back(packs.size()); packs.push_back(pk);
						}
					}
				}
				std::vector<std::pair<int, int>> order; for (const auto& kv : weight) { const auto u = unit_of.find(kv.first); if (u == unit_of.end() || u->second == kv.first) order.push_back({ kv.second, kv.first }); }
				std::sort(order.begin(), order.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first != y.first ? x.first > y.first : x.second < y.second; });
				std::vector<int> load((size_t)NWV, 0); std::map<int, int> wave_of_strand;
				for (const auto& o : order) { int best = w0; for (int w = w0 + 1; w < w1; w++) if (load[(size_t)w] < load[(size_t)best]) best = w; load[(size_t)best] += o.first; wave_of_strand[o.second] = best; }
				for (const auto& u : unit_of) wave_of_strand[u.first] = wave_of_strand[u.second];
				for (size_t pi : new_packs) {
					Pack& pk = packs[pi];
					pk.wave = wave_of_strand[strand[(size_t)comp[(size_t)pk.ops[0][0]]]];
					for (size_t q = 0; q < pk.ops.size(); q++) for (int i : pk.ops[q]) op_sub[(size_t)i] = q == 0 ? (int)pi : -2;
				}
				for (int c = 0; c < ncomp; c++) if (cserial[(size_t)c] && clevel[(size_t)c] == lv && (cpfx(c, members) ? 1 : 0) == grp) wave_of[(size_t)c] = wave_of_strand[strand[(size_t)c]];
			}
		}
		auto ser_of = [&](int i) { return comp[(size_t)i] >= 0 && cserial[(size_t)comp[(size_t)i]] != 0; };
		auto wave_of_op = [&](int i) { return comp[(size_t)i] >= 0 ? wave_of[(size_t)comp[(size_t)i]] : -1; };
		auto live_op = [&](int i) { return !is_struct(V[(size_t)i].code) && !inv[(size_t)i] && comp[(size_t)i] >= 0; };
		// is op i one of (group, level, wave)?  wave -1: the parallel ops of the level
		int cur_sub = -1;                                                          // which serial ops a block is being generated for: -1 the strands on their own, p the pack p (its first strand)
		auto in_block = [&](int i, bool pf, int lv, int wave) { return live_op(i) && (pfx[(size_t)i] != 0) == pf && level[(size_t)i] == lv && ((wave >= 0) == ser_of(i)) && (wave < 0 || (wave_of_op(i) == wave && op_sub[(size_t)i] == cur_sub)); };
		// ---- where each register lives ----
		// chunk: a lane's own register across levels of its group; xrot: parallel prefix -> parallel suffix (computed an iteration ahead, handed over at the top of the
		// loop); slot: through LDS — a prefix value in two buffers (the chunk's parity), a suffix value in a slot that is reused once it is dead
		struct Reg { int def = -1; bool slot = false, chunk = false, xrot = false; int last = -1; int slot_id = -1; };
		std::vector<Reg> regs((size_t)maxreg + 2);
		auto use = [&](int r, bool at_pf, int at_level, bool at_ser, int at_wave) {
			if (r < 0 || (size_t)r >= regs.size()) return;
			const int d = def_at[(size_t)r]; if (d < 0 || inv[(size_t)d]) return;  // (a prepare() register cannot be named here: Program::validate)
			Reg& R = regs[(size_t)r]; R.def = d;
			const int dl = level[(size_t)d]; const bool dser = ser_of(d), dpf = pfx[(size_t)d] != 0;
			if (dpf != at_pf) { if (!dser && !at_ser) R.xrot = true; else { R.slot = true; R.last = 1 << 20; } return; }
			if (!dser && !at_ser) { if (at_level != dl) R.chunk = true; }
			else if (dser && at_ser && dl == at_level && wave_of_op(d) == at_wave) {}
			else { R.slot = true; R.last = std::max(R.last, at_level); }
		};
		for (int i = 0; i < NV; i++) {
			const VOp& v = V[(size_t)i];
			if (!live_op(i)) continue;
			const bool s = ser_of(i), pf = pfx[(size_t)i] != 0; const int lv = level[(size_t)i], w = s ? wave_of_op(i) : -1;
			if (v.a >= 0) use(v.a, pf, lv, s, w);
			if (v.b >= 0) use(v.b, pf, lv, s, w);
			for (int r : v.x) use(r, pf, lv, s, w);
			for (const auto& pe : v.path) use(V[(size_t)pe.first].a, pf, lv, s, w);
			if (v.code == OP_PHI && phi_if[(size_t)i] >= 0) use(V[(size_t)phi_if[(size_t)i]].a, pf, lv, s, w);
		}
		use(g.ret, false, out_level, false, -1);
		if (CH == 2) use(g.ret_r, false, out_level, false, -1);
		// a pack's strands store the same results: where one strand's result crosses levels, every strand's does (a slot each; one nobody reads is written all the same)
		for (const Pack& pk : packs) for (size_t pos = 0; pos < pk.ops[0].size(); pos++) {
			bool any = false; int last = -1;
			for (const auto& ops : pk.ops) { const int r = V[(size_t)ops[pos]].dst; if (r >= 0 && def_at[(size_t)r] == ops[pos] && regs[(size_t)r].slot) { any = true; last = std::max(last, regs[(size_t)r].last); } }
			if (any) for (const auto& ops : pk.ops) { const int r = V[(size_t)ops[pos]].dst; if (r >= 0 && def_at[(size_t)r] == ops[pos]) { Reg& R = regs[(size_t)r]; R.def = ops[pos]; R.slot = true; R.last = std::max(R.last, last); } }
		}
		for (size_t r = 0; r < regs.size(); r++) if (regs[r].slot && is_dbl[r]) return refuse("a double register would have to cross levels");
		// slots: suffix values by interval [def level, last use level], reused when disjoint; prefix values one (double-buffered) slot each
		int nslots = 0, npslots = 0;
		{
			std::vector<int> order;
			for (size_t r = 0; r < regs.size(); r++) if (regs[r].slot) { if (pfx[(size_t)regs[r].def]) regs[r].slot_id = npslots++; else order.push_back((int)r); }
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
		long long lds_words = (long long)NW * G * (pipelined ? 3 : 1) + (long long)CH * C * G * (any_near ? 2 : 1) + (long long)(nslots + 2 * npslots) * C * G + 4;   // (any_near: a second copy of the chunk's `in`)
		const char* le = getenv("KLG_FX_STAGED_LDS");
		const long long budget = le ? atoll(le) : 160 * 1024;
		// two tiles, by the parity of the chunk, where they fit: a chunk's results leave from one while the next chunk's input lands in the other — no barrier between them
		const char* te = getenv("KLG_FX_STAGED_TILES");
		const bool two_tiles = !(te && te[0] == '1') && (lds_words + (long long)CH * C * G) * 4 <= budget;
		if (two_tiles) lds_words += (long long)CH * C * G;                 // gfx950 grants a workgroup up to 160 KB; a plan that needs most of it (the recorded Reverb.k: 68 values x 32 samples x 16 instances) is one workgroup of 8 waves per CU — measured 0.61 against 0.83 ms at 4,096 instances, 2.5 against 4.4 ms at 16,384, with half the chunk
		// A chunk whose ring check fails is tried again in PARTS before it goes to the plain body: halves, then quarters (never below 8 samples: the serial
		// loops take their inputs eight at a time) — a tap 20 samples behind the cursor fails a 32-sample chunk and passes its halves; the plain body (one lane
		// per instance, the whole sample's dependent chain) then walks only what no part could take.  A part runs the audio path's levels on its samples; of the
		// control path — computed a chunk ahead: its values for this chunk stay where they are — it runs the SERIAL loops again, on the architectural records
		// (the same arithmetic on the same inputs), so that the records are complete at the end of every part and the plain body can take over at any of them.
		// Nothing of this is on the path of a chunk that passes its check.
		const char* re = getenv("KLG_FX_STAGED_RETRY");
		const int CMIN = std::max(8, C / 4);
		const bool retry = !(re && re[0] == '0') && C >= 16 && guard_level >= 0;
		if (lds_words * 4 > budget) { if (in.C > 0 && !in.C_is_a_preference) return refuse("the requested chunk length does not fit the LDS budget"); continue; }

		// =========================================================== source ===========================================================
		std::string s;
		bool part = false;                                                         // generating the text of a PART of a chunk (the retry after a failed check): samples OFF .. OFF + CC - 1 of the chunk
		auto ring = [&](int node) { return F("RingS{ (const char*)(ring0 + (size_t)%lldll * %d), (unsigned)pg * 4u, %du, %d }", (*in.ring_off)[(size_t)node], G, G * 4, g.arg(node)); };   // rows of G instances: this workgroup's own (PatchGen::kRingRow); a wave-uniform base + 32-bit offsets (klg_delay.hpp RingS)
		auto ty = [&](int r) { return std::string(is_dbl[(size_t)r] ? "double" : "float"); };
		auto in_branch = [&](int i) { return !V[(size_t)i].path.empty(); };
		// where a register's LDS copy is, for code of group `pf` (the prefix works on the NEXT chunk: the other parity)
		std::map<int, std::string> slot_override;                                  // a pack's loop: the slot of (its first strand's) register r, by the lane's quarter
		auto slot_ref = [&](int r, bool reader_pf) {
			if (!slot_override.empty()) { const auto it = slot_override.find(r); if (it != slot_override.end()) return it->second; }
			const Reg& R = regs[(size_t)r];
			if (pfx[(size_t)R.def]) return F("SLP(%d, %s)", R.slot_id, (reader_pf && !part) ? "parn" : "parc");
			return F("SL(%d)", R.slot_id);
		};
		std::string inst = "pg";                                                   // which instance the code being generated works for: a lane of a parallel level (pg) or of a serial loop (ln)
		// the statement(s) of virtual op i, in the context (L, c) it is emitted in
		// tap_mode (ring reads only): 0 the whole read; 1 its first half — rows found, checked, loads issued into tf<i> —; 2 the arithmetic on what they returned
		auto op_text = [&](int i, bool assign, int tap_mode = 0) {
			const VOp& v = V[(size_t)i];
			std::string b;
			std::string d = assign ? F("\t\tr%d = ", v.dst) : "\t\tconst " + ty(std::max(v.dst, 0)) + F(" r%d = ", v.dst);
			if (tap_mode && (v.code == OP_DELAYOUT || v.code == OP_DELAYTAP)) {
				const char* fn = v.code == OP_DELAYOUT ? "staged_process" : v.imm == 1u ? "staged_tap_int" : v.imm == 3u ? "staged_lagrange" : v.imm == 2u ? "staged_tap_stereo" : "staged_tap_float";
				if (tap_mode == 2 && !(v.code == OP_DELAYTAP && v.imm == 0u && near_ch[(size_t)v.node] >= 0)) return d + fn + F("_finish(tf%d);\n", i);
			}
			const int SZ = v.node >= 0 ? g.arg(v.node) : 0;
			auto pos = [&](int j) { return j ? F("ring_at(d%dp0, %d, %d)", v.node, j, SZ) : F("d%dp0", v.node); };
			switch (v.code) {
			case OP_PARAM:                                                           // a member process() only reads: from the LDS copy of the records where it is needed — as fields of the
				if (!written[(size_t)v.node]) { b += d + F("u2f(srec[%d * G + %s]);\n", g.node_word0(v.node), inst.c_str()); break; }   // lanes' Live structs the recorded Reverb.k's 77 of them were held (three times) for the whole block: 1,123 spilled registers, 3.7 KB of scratch per lane, 5 x the algorithmic traffic
				in.emit_op((size_t)v.orig, b, assign); break;
			case V_OSCARG: b += d + F("basic_sine_arg(L.n%d);\n", v.node); break;
			case V_OSCEVAL: b += d + F("basic_sine_of(r%d);\n", v.a); break;
			case V_LPFQ: b += d + F("biquad_q(r%d, r%d);\n", v.a, v.b); break;
			case V_LPFCOEF: b += d + F("biquad_coef<%u, %u>(r%d, r%d, c.fs.w);\n", v.imm & 0xffu, v.imm >> 8, v.a, v.b); break;
			case V_LPFAPPLY: b += F("\t\tbiquad_apply(L.n%d, L.n%ds, r%d, r%d, r%d, r%d, r%d, r%d, r%d);\n", v.node, v.node, v.a, v.b, v.x[0], v.x[1], v.x[2], v.x[3], v.x[4]); break;
			case OP_PHI: b += d + F("(r%d != 0.f) ? r%d : r%d;\n", V[(size_t)phi_if[(size_t)i]].a, v.a, v.b); break;
			case OP_DELAYIN: b += "\t\t{ const RingS q = " + ring(v.node) + "; q.wr(" + pos(in_index[(size_t)i]) + F(", r%d); }\n", v.a); break;
			case OP_DELAYSET: b += F("\t\td%dt = delay_set(", v.node) + pos(in_index[(size_t)i]) + F(", %d, r%d);\n", SZ, v.a); break;
			case OP_DELAYOUT:
				if (tap_mode == 1) d = F("\t\ttf%d = ", i);
				if (set_at[(size_t)v.node] >= 0) b += d + (tap_mode == 1 ? "staged_process_fetch(" : "staged_process(") + ring(v.node) + F(", ring_walk(d%dt.position, %d, %d), d%dt.fraction, d%dw, bad);\n", v.node, out_index[(size_t)i], SZ, v.node, v.node);
				else b += d + (tap_mode == 1 ? "staged_process_fetch(" : "staged_process(") + ring(v.node) + F(", ring_walk(d%dh.position, %s * %d + %d, %d), d%dh.fraction, d%dw, bad);\n", v.node, part ? "(ps - OFF)" : "ps", outs[(size_t)v.node], out_index[(size_t)i], SZ, v.node, v.node);
				break;
			case OP_DELAYTAP:
				if (v.imm == 0u && near_ch[(size_t)v.node] >= 0) {                      // a line fed by `in`: no check (see near_ch)
					const std::string args = ring(v.node) + ", " + pos(in_index[(size_t)i]) + F(", r%d, d%dnw0, ps + %d, incopy + (%d * C) * G + pg, G);\n", v.a, v.node, in_index[(size_t)i], near_ch[(size_t)v.node]);
					if (tap_mode == 2) b += d + F("staged_tap_float_finish_near(tf%d);\n", i);
					else if (tap_mode == 1) b += F("\t\ttf%d = staged_tap_float_fetch_near(", i) + args;
					else b += d + "staged_tap_float_near(" + args;
					break;
				}
				if (tap_mode == 1) d = F("\t\ttf%d = ", i);
				if (v.imm == 1u) b += d + (tap_mode == 1 ? "staged_tap_int_fetch(" : "staged_tap_int(") + ring(v.node) + ", " + pos(in_index[(size_t)i]) + F(", (int)r%d, d%dw, bad);\n", v.a, v.node);
				else b += d + std::string(v.imm == 3u ? "staged_lagrange" : v.imm == 2u ? "staged_tap_stereo" : "staged_tap_float") + (tap_mode == 1 ? "_fetch(" : "(") + ring(v.node) + ", " + pos(in_index[(size_t)i]) + F(", r%d, d%dw, bad);\n", v.a, v.node);
				break;
			default: in.emit_op((size_t)v.orig, b, assign); break;
			}
			return b;
		};
		// A serial loop pays for every branch of its body in every sample (exec-mask bookkeeping and a jump on a chain nothing overlaps).  What stands inside a
		// recorded `if` there is mostly plain arithmetic into single-assignment registers — only read under the same condition, or by a phi: computed
		// unconditionally — and assignments to state (`L.x = e`: a member, a control, an oscillator's set()): `L.x = cond ? e : L.x`.  Statements of any other shape
		// (a call that takes a node by reference: a filter, an oscillator step) keep their branch.  Returns false when the op's text has such a statement.
		auto predicated = [&](const std::string& text, const std::string& cond, std::string& out) {
			std::string res; size_t at = 0;
			while (at < text.size()) {
				size_t e = text.find(";\n", at); if (e == std::string::npos) return false;
				std::string st = text.substr(at, e - at); at = e + 2;
				// (several statements may share a line: `a = x; b = y;\n`)
				std::vector<std::string> parts; { size_t b = 0; for (;;) { const size_t q = st.find("; ", b); if (q == std::string::npos) { parts.push_back(st.substr(b)); break; } parts.push_back(st.substr(b, q - b)); b = q + 2; } }
				for (std::string p : parts) {
					const size_t ns = p.find_first_not_of(" \t"); if (ns == std::string::npos) continue; p = p.substr(ns);
					const size_t eq = p.find(" = "); if (eq == std::string::npos) return false;
					const std::string lhs = p.substr(0, eq), rhs = p.substr(eq + 3);
					if (rhs.find("(L.n") != std::string::npos || rhs.find("SL(") != std::string::npos) return false;
					if (lhs.rfind("L.", 0) == 0) res += "\t\t" + lhs + " = (" + cond + ") ? (" + rhs + ") : " + lhs + ";\n";
					else if (lhs.rfind("const float r", 0) == 0 || lhs.rfind("const double r", 0) == 0 || (lhs[0] == 'r' && lhs.find_first_not_of("0123456789", 1) == std::string::npos)) res += "\t\t" + p + ";\n";
					else return false;
				}
			}
			out += res; return true;
		};
		// `if (c) osc.set(f, pi); else osc.set(f);` (PingPong.k:52-55) leaves, predicated, `X = c ? e : X` and later `X = !c ? e : X` for the same state X and the same
		// expression e — the frequency and the increment either set() stores: X = e whatever c says.  Written so, e being a block invariant, X stops being carried from
		// sample to sample at all (four selects a sample on the recorded PingPong.k's control chain).  Only the plain shape: one condition and its negation, the same text
		// for e, nothing in between that names X.
		auto merge_complementary = [](std::string& body) {
			std::vector<std::string> lines; { size_t at = 0; while (at < body.size()) { size_t e = body.find('\n', at); if (e == std::string::npos) e = body.size() - 1; lines.push_back(body.substr(at, e - at + 1)); at = e + 1; } }
			struct Sel { bool ok = false; std::string lhs, cond, rhs; };
			auto parse = [](const std::string& ln) {
				Sel r;
				if (ln.rfind("\t\tL.", 0) != 0) return r;
				const size_t eq = ln.find(" = ("); if (eq == std::string::npos) return r;
				r.lhs = ln.substr(2, eq - 2);
				size_t at = eq + 3; int depth = 0; size_t q = at;
				for (; q < ln.size(); q++) { if (ln[q] == '(') depth++; else if (ln[q] == ')') { if (--depth == 0) break; } }
				if (q >= ln.size()) return r;
				r.cond = ln.substr(at + 1, q - at - 1);
				const std::string mid = " ? (", tail = ") : " + r.lhs + ";\n";
				if (ln.compare(q + 1, mid.size(), mid) != 0 || ln.size() < tail.size() + q + 1 + mid.size() || ln.compare(ln.size() - tail.size(), tail.size(), tail) != 0) return r;
				r.rhs = ln.substr(q + 1 + mid.size(), ln.size() - tail.size() - (q + 1 + mid.size()));
				r.ok = r.cond.find("&&") == std::string::npos;
				return r;
			};
			bool changed = false;
			for (size_t i = 0; i < lines.size(); i++) {
				const Sel a = parse(lines[i]); if (!a.ok) continue;
				for (size_t j = i + 1; j < lines.size(); j++) {
					if (lines[j].find(a.lhs) == std::string::npos) continue;
					const Sel b = parse(lines[j]);
					if (b.ok && b.lhs == a.lhs && b.rhs == a.rhs && (b.cond == "!" + a.cond || a.cond == "!" + b.cond)) { lines[j] = "\t\t" + a.lhs + " = (" + a.rhs + ");\n"; lines[i].clear(); changed = true; }
					break;                                                                   // (the first later line that names X decides)
				}
			}
			if (changed) { body.clear(); for (const std::string& l : lines) body += l; }
		};
		// ops of one block in program order, with the branches they stand in re-opened around them
		auto emit_ops = [&](bool pf, int lv, int wave /* -1: parallel */, const std::string& slot_index, std::string& out, const std::vector<char>& predeclared) {
			std::vector<std::pair<int, int>> open;
			auto close_to = [&](size_t keep) { while (open.size() > keep) { out += "\t\t}\n"; open.pop_back(); } };
			static const bool ifconv = []() { const char* e = getenv("KLG_FX_STAGED_IFCONV"); return !(e && e[0] == '0'); }();
			const bool batch_taps = []() { const char* e = getenv("KLG_FX_STAGED_BATCH"); return !(e && e[0] == '0'); }();   // (read per plan: the compile cache is keyed by it)
			// A parallel level's ring reads, many under way at a time.  Written one after the other, every read is two loads, a wait for both and the
			// interpolation — a memory round trip per tap, fifty in a row in the recorded Reverb.k's first level.  Nothing in a parallel level has a side effect
			// another op of the level could see, so its ops may run in any order their operands allow: a read's first half (rows, check, loads: tf<i>) is emitted
			// where the read stands, its second half (the arithmetic) and everything that needs it wait until 2 x BATCH reads are under way; then the oldest
			// BATCH are finished — behind a compiler barrier that keeps the loads above it — and what waited for them follows.
			if (wave < 0 && batch_taps) {
				std::vector<int> here; bool any_tap = false;
				auto is_tap = [&](int i) { return V[(size_t)i].code == OP_DELAYOUT || V[(size_t)i].code == OP_DELAYTAP; };
				for (int i = 0; i < NV; i++) if (in_block(i, pf, lv, wave)) { here.push_back(i); if (is_tap(i)) any_tap = true; }
				if (any_tap) {
					const int BATCH = []() { const char* e = getenv("KLG_FX_STAGED_BATCH"); const int b = e ? atoi(e) : 0; return b >= 2 ? b : 8; }();
					std::vector<char> inblk((size_t)NV, 0), done((size_t)NV, 0);
					for (int i : here) inblk[(size_t)i] = 1;
					auto ready = [&](int i) {
						const VOp& v = V[(size_t)i];
						bool ok = true;
						auto rd = [&](int r) { const int j = (r >= 0 && (size_t)r < def_at.size()) ? def_at[(size_t)r] : -1; if (j >= 0 && inblk[(size_t)j] && !done[(size_t)j]) ok = false; };
						rd(v.a); rd(v.b);
						for (int r : v.x) rd(r);
						for (const auto& pe : v.path) rd(V[(size_t)pe.first].a);
						if (v.code == OP_PHI && phi_if[(size_t)i] >= 0) rd(V[(size_t)phi_if[(size_t)i]].a);
						if (v.code == OP_DELAYOUT && v.node >= 0 && set_at[(size_t)v.node] >= 0) { const int sa = set_at[(size_t)v.node]; if (inblk[(size_t)sa] && !done[(size_t)sa]) ok = false; }   // (the head a process() walks is what set() left: d<n>t)
						return ok;
					};
					for (int i : here) if (is_tap(i)) out += F("\t\tTapFetch tf%d;\n", i);
					auto stmt = [&](int i, const std::string& text) {
						const VOp& v = V[(size_t)i];
						size_t common = 0;
						while (common < open.size() && common < v.path.size() && open[common] == v.path[common]) common++;
						close_to(common);
						for (size_t q = common; q < v.path.size(); q++) { out += F("\t\tif (%s(r%d != 0.f)) {\n", v.path[q].second ? "!" : "", V[(size_t)v.path[q].first].a); open.push_back(v.path[q]); }
						out += text;
					};
					auto has_dst_of = [&](const VOp& v) { return v.dst >= 0 && v.code != OP_OSCSET && v.code != OP_LPFSET && v.code != OP_SETPARAM && v.code != OP_DELAYIN && v.code != OP_DELAYSET; };
					// A read inside a recorded `if` is taken OUTSIDE it: what is assigned under a branch and used behind the join lives in scratch (a store and a wait
					// per load).  Every row a read computes is a row of its line whatever its operand holds; only its check counts under the branch's condition alone.
					auto cond_of = [&](const VOp& v) { std::string c; for (const auto& pe : v.path) c += (c.empty() ? "" : " && ") + F("%s(r%d != 0.f)", pe.second ? "!" : "", V[(size_t)pe.first].a); return c; };
					auto fetch = [&](int i) {
						const VOp& v = V[(size_t)i];
						close_to(0);
						if (v.path.empty()) { out += op_text(i, false, 1); return; }
						std::string text = op_text(i, false, 1);
						const size_t at = text.rfind(", bad)");
						if (at != std::string::npos) text.replace(at, 6, ", badl)");
						out += "\t\t{ int badl = 0;\n" + text + "\t\tbad |= (" + cond_of(v) + ") ? badl : 0; }\n";
					};
					auto finish = [&](int i) {
						const VOp& v = V[(size_t)i];
						const bool assign = predeclared[(size_t)v.dst] != 0;
						std::string text = op_text(i, assign, 2);
						if (regs[(size_t)v.dst].slot) text += "\t\t" + slot_ref(v.dst, pf) + F("[%s] = r%d;\n", slot_index.c_str(), v.dst);
						close_to(0);
						out += text;
						done[(size_t)i] = 1;
					};
					auto plain_op = [&](int i) {
						const VOp& v = V[(size_t)i];
						const bool has_dst = has_dst_of(v), assign = has_dst && predeclared[(size_t)v.dst] != 0;
						std::string text = op_text(i, assign);
						if (has_dst && regs[(size_t)v.dst].slot) text += "\t\t" + slot_ref(v.dst, pf) + F("[%s] = r%d;\n", slot_index.c_str(), v.dst);
						stmt(i, text);
						done[(size_t)i] = 1;
					};
					const std::string fence = "\t\tasm volatile(\"\" ::: \"memory\");                                       // (the loads above are not moved below this line)\n";
					std::vector<int> pending, waiting;                                         // reads under way (oldest first); ops whose operands are not there yet (program order)
					auto take = [&](int i) { if (is_tap(i)) { fetch(i); pending.push_back(i); } else plain_op(i); };
					auto drain = [&](size_t n) {
						if (n > pending.size()) n = pending.size();
						if (n) { close_to(0); out += fence; for (size_t q = 0; q < n; q++) finish(pending[q]); pending.erase(pending.begin(), pending.begin() + (std::ptrdiff_t)n); }
						for (bool moved = true; moved;) {                                        // what waited for them, in program order (a read among it goes under way)
							moved = false;
							for (size_t q = 0; q < waiting.size();) {
								if (pending.size() >= 2 * (size_t)BATCH && is_tap(waiting[q])) { q++; continue; }
								if (ready(waiting[q])) { const int i = waiting[q]; waiting.erase(waiting.begin() + (std::ptrdiff_t)q); take(i); moved = true; }
								else q++;
							}
						}
					};
					for (int i : here) {
						if (waiting.empty() ? ready(i) : (is_tap(i) && ready(i))) take(i);       // (an op behind a waiting one waits too unless it is a read: consumers keep their order)
						else waiting.push_back(i);
						if (pending.size() >= 2 * (size_t)BATCH) drain((size_t)BATCH);
					}
					while (!pending.empty() || !waiting.empty()) {
						const size_t before = pending.size() + waiting.size();
						drain(pending.size() > (size_t)BATCH ? (size_t)BATCH : pending.size());
						if (pending.size() + waiting.size() == before && pending.empty()) break;   // (cannot happen: the ops of a block are in def-before-use order)
					}
					close_to(0);
					return;
				}
			}
			for (int i = 0; i < NV; i++) {
				const VOp& v = V[(size_t)i];
				if (!in_block(i, pf, lv, wave)) continue;
				if (wave >= 0 && ifconv && !v.path.empty() && v.code != V_OSCEVAL) {      // a serial loop: without the branch where the op allows it
					const bool has_dst = v.dst >= 0 && v.code != OP_OSCSET && v.code != OP_LPFSET && v.code != OP_SETPARAM && v.code != OP_DELAYIN && v.code != OP_DELAYSET;
					std::string cond;
					for (const auto& pe : v.path) cond += (cond.empty() ? "" : " && ") + F("%s(r%d != 0.f)", pe.second ? "!" : "", V[(size_t)pe.first].a);
					std::string conv;
					if (predicated(op_text(i, has_dst && predeclared[(size_t)v.dst] != 0), cond, conv)) {
						close_to(0);
						out += conv;
						if (has_dst && regs[(size_t)v.dst].slot) out += "\t\t" + slot_ref(v.dst, pf) + F("[%s] = r%d;\n", slot_index.c_str(), v.dst);
						continue;
					}
				}
				size_t common = 0;
				while (common < open.size() && common < v.path.size() && open[common] == v.path[common]) common++;
				close_to(common);
				for (size_t q = common; q < v.path.size(); q++) { out += F("\t\tif (%s(r%d != 0.f)) {\n", v.path[q].second ? "!" : "", V[(size_t)v.path[q].first].a); open.push_back(v.path[q]); }
				const bool has_dst = v.dst >= 0 && v.code != OP_OSCSET && v.code != OP_LPFSET && v.code != OP_SETPARAM && v.code != OP_DELAYIN && v.code != OP_DELAYSET;
				const bool assign = has_dst && predeclared[(size_t)v.dst] != 0;
				out += op_text(i, assign);
				if (has_dst && regs[(size_t)v.dst].slot) out += "\t\t" + slot_ref(v.dst, pf) + F("[%s] = r%d;\n", slot_index.c_str(), v.dst);
			}
			close_to(0);
		};
		// every register the block reads (operands, branch conditions, phi conditions) -> f(r)
		auto for_reads = [&](bool pf, int lv, int wave, const std::function<void(int)>& f) {
			for (int i = 0; i < NV; i++) {
				const VOp& v = V[(size_t)i];
				if (!in_block(i, pf, lv, wave)) continue;
				f(v.a); f(v.b);
				for (int r : v.x) f(r);
				for (const auto& pe : v.path) f(V[(size_t)pe.first].a);
				if (v.code == OP_PHI) f(V[(size_t)phi_if[(size_t)i]].a);
			}
			if (!pf && wave < 0 && lv == out_level) { f(g.ret); if (CH == 2) f(g.ret_r); }
		};
		// registers the block reads from LDS
		auto inputs_of = [&](bool pf, int lv, int wave, std::vector<int>& from_slot) {
			std::vector<char> seen(regs.size(), 0);
			for_reads(pf, lv, wave, [&](int r) {
				if (r < 0 || (size_t)r >= regs.size() || seen[(size_t)r]) return;
				const int d = def_at[(size_t)r]; if (d < 0 || inv[(size_t)d]) return;
				if (in_block(d, pf, lv, wave)) return;
				if (wave < 0 && !ser_of(d)) return;                                   // parallel -> parallel: a register of the lane (chunk / xrot)
				seen[(size_t)r] = 1; from_slot.push_back(r);
			});
		};
		// the block invariants the block reads, in program order
		auto inv_prelude = [&](bool pf, int lv, int wave) {
			std::vector<char> need((size_t)NV, 0);
			auto want = [&](int r) { const int d = (r >= 0 && (size_t)r < def_at.size()) ? def_at[(size_t)r] : -1; if (d >= 0 && inv[(size_t)d]) need[(size_t)d] = 1; };
			for_reads(pf, lv, wave, want);
			for (int i = NV - 1; i >= 0; i--) if (need[(size_t)i]) { want(V[(size_t)i].a); want(V[(size_t)i].b); for (int r : V[(size_t)i].x) want(r); }     // operands come earlier in program order
			std::string t;
			for (int i = 0; i < NV; i++) if (need[(size_t)i]) t += op_text(i, false);
			return t;
		};
		const char* skip_env = getenv("KLG_FX_STAGED_SKIP");                         // (measurement only: a bit mask of suffix levels whose code is left out — the result is wrong, the time is what the others cost)
		const unsigned skip_mask = skip_env ? (unsigned)strtoul(skip_env, nullptr, 0) : 0u;
		std::vector<std::string> deferred;                                          // commits of suffix serial levels at or below the guard
		// ---- one block of code: the parallel ops of (group, level), or the serial loops of (group, level), one per wave that has work ----
		bool first_pass = true;                                                      // (the control path's blocks are generated twice: for chunk 0 ahead of the loop, and inside it)
		auto block = [&](bool pf, int lv) {
			std::string code;
			const char* cond = (pf && !part) ? "pre" : "ok";                        // (a part of a chunk runs the control path's serial loops with the audio path's: above)
			if (!pf && lv < 32 && ((skip_mask >> lv) & 1u)) return code;
			inst = (lv & 1) ? "ln" : "pg";
			if (!(lv & 1)) {
				std::vector<int> from_slot; inputs_of(pf, lv, -1, from_slot);
				std::vector<char> predecl(regs.size(), 0);
				std::string decl, body;
				if (pf) {                                                              // the prefix works on the next chunk: its own cursor positions, its registers under their plain names
					decl += "\t\tLq.sidx = s0 + C + ps;\n";
					std::vector<char> dn(NN, 0);
					for (int i = 0; i < NV; i++) if (in_block(i, pf, lv, -1) && V[(size_t)i].code == OP_DELAYSET) dn[(size_t)V[(size_t)i].node] = 1;
					for (size_t nd = 0; nd < NN; nd++) if (dn[nd]) decl += F("\t\tconst int d%zup0 = (int)((d%zub + (unsigned)(s0 + C + ps) * %du) %% %du); Tap& d%zut = xn_d%zut;\n", nd, nd, k_in[nd], g.arg((int)nd), nd, nd);
					std::vector<char> bound(regs.size(), 0);
					auto bind = [&](int r) {
						if (r < 0 || (size_t)r >= regs.size() || bound[(size_t)r]) return;
						const Reg& R = regs[(size_t)r]; if (R.def < 0 || !pfx[(size_t)R.def] || ser_of(R.def) || !(R.chunk || R.xrot)) return;
						bound[(size_t)r] = 1; predecl[(size_t)r] = 1;
						decl += "\t\t" + ty(r) + F("& r%d = %s_r%d; (void)r%d;\n", r, R.xrot ? "xn" : "pc", r, r);
					};
					for_reads(pf, lv, -1, bind);
					for (int i = 0; i < NV; i++) if (in_block(i, pf, lv, -1) && V[(size_t)i].dst >= 0 && def_at[(size_t)V[(size_t)i].dst] == i) bind(V[(size_t)i].dst);
				}
				for (int r : from_slot) decl += "\t\tconst float " + F("r%d = ", r) + slot_ref(r, pf) + "[t];\n";
				decl += inv_prelude(pf, lv, -1);
				for (int i = 0; i < NV; i++) if (in_block(i, pf, lv, -1) && V[(size_t)i].dst >= 0 && (size_t)V[(size_t)i].dst < regs.size() && def_at[(size_t)V[(size_t)i].dst] == i) {
					const int r = V[(size_t)i].dst;
					if (predecl[(size_t)r]) continue;
					if (!pf && regs[(size_t)r].chunk) predecl[(size_t)r] = 1;                          // (declared at the top of the chunk)
					else if (in_branch(i)) { predecl[(size_t)r] = 1; decl += "\t\t" + ty(r) + F(" r%d = 0; (void)r%d;\n", r, r); }
				}
				emit_ops(pf, lv, -1, "t", body, predecl);
				if (!pf) {
					if (lv > guard_level && !deferred.empty()) { for (const std::string& d : deferred) code += d; deferred.clear(); }
					if (lv == guard_level) body += "\t\tif (bad && k0 + pg < a.K) *flag = 1;                                    // (the lanes past the bank's last instance — a last workgroup that is not full — have no say: their dials are zeros, their taps sit on the cursor)\n";
					if (lv == out_level) {
						body += F("\t\ttile[(0 * C + ps) * G + pg] = r%d;\n", g.ret);
						if (CH == 2) body += F("\t\ttile[(1 * C + ps) * G + pg] = r%d;\n", g.ret_r);
					}
					if (lv == last_level) {                                                  // the read heads as the chunk leaves them (Delay::last)
						for (size_t nd = 0; nd < NN; nd++) if (g.nodes[nd] == N_DELAY && head_used[nd]) {
							const int SZ = g.arg((int)nd), w0 = g.node_word0((int)nd);
							if (set_at[nd] >= 0) body += F("\t\tif (ps == %s) { srec[%d * G + pg] = (uint32_t)ring_walk(d%zut.position, %d, %d); srec[%d * G + pg] = f2u(d%zut.fraction); }\n", part ? "OFF + CC - 1" : "C - 1", w0 + ED_LASTPOS, nd, outs[nd], SZ, w0 + ED_LASTFRAC, nd);
							else if (outs[nd] > 0) body += F("\t\tif (ps == %s) srec[%d * G + pg] = (uint32_t)ring_walk(d%zuh.position, %s * %d, %d);\n", part ? "OFF" : "0", w0 + ED_LASTPOS, nd, part ? "CC" : "C", outs[nd], SZ);
						}
					}
				}
				if (!decl.empty() || !body.empty()) code += F("\t\tif (%s && t < NTP%s) { auto& L = %s; const FxCtx& c = cp; (void)L; (void)c;\n", cond, part ? " && ps >= OFF && ps < OFF + CC" : "", pf ? "Lq" : "Lp") + decl + body + "\t\t}\n";
			}
			else {
				for (int w = 0; w < NWV; w++) {
				std::vector<int> subs = { -1 };
				for (size_t pi = 0; pi < packs.size(); pi++) if (packs[pi].lv == lv && packs[pi].pf == pf && packs[pi].wave == w) subs.push_back((int)pi);
				for (int sub : subs) {
					cur_sub = sub; slot_override.clear();
					std::vector<int> mine;
					for (int i = 0; i < NV; i++) if (in_block(i, pf, lv, w)) mine.push_back(i);
					if (mine.empty()) continue;
					const Pack* pk = sub >= 0 ? &packs[(size_t)sub] : nullptr;
					const int K = pk ? (int)pk->ops.size() : 1;
					const std::string li = pk ? "gi" : "ln";                               // the instance a lane works for
					auto qsel = [&](const std::vector<long long>& v) {                      // a number by the lane's quarter
						bool same = true; for (long long x : v) if (x != v[0]) same = false;
						if (same) return F("%lld", v[0]);
						std::string t = F("%lld", v.back());
						for (size_t q = v.size() - 1; q-- > 0;) t = F("(qq == %zu ? %lld : ", q, v[q]) + t + ")";
						return t;
					};
					std::vector<char> node_here(NN, 0);
					for (int i : mine) { int ns[2]; const int n = nodes_of(V[(size_t)i], ns); for (int q = 0; q < n; q++) node_here[(size_t)ns[q]] = 1; }
					std::string load, commit, loop, decl, pack_decl;
					if (pk) {
						pack_decl += "\t\tconst int qq = ln / G, gi = ln - qq * G; (void)qq; (void)gi;\n";
						// the corresponding nodes of the pack's other strands: by position in the op lists
						std::vector<std::vector<long long>> w0s(NN);
						for (size_t pos = 0; pos < pk->ops[0].size(); pos++) {
							int n0[2]; const int c0 = nodes_of(V[(size_t)pk->ops[0][pos]], n0);
							for (int q = 0; q < c0; q++) if (w0s[(size_t)n0[q]].empty()) for (int m = 0; m < K; m++) { int nm[2]; nodes_of(V[(size_t)pk->ops[(size_t)m][pos]], nm); w0s[(size_t)n0[q]].push_back((long long)g.node_word0(nm[q]) - (long long)g.node_word0(n0[q])); }
						}
						auto shifted = [&](std::string text, size_t nd) {                     // r.w[N] -> r.w[N + dw<nd>]
							const std::string key = "r.w[";
							for (size_t at = 0; (at = text.find(key, at)) != std::string::npos;) { const size_t e = text.find(']', at); text.insert(e, F(" + dw%zu", nd)); at = e; }
							return text;
						};
						for (size_t nd = 0; nd < NN; nd++) if (node_here[nd]) { pack_decl += F("\t\tconst int dw%zu = ", nd) + qsel(w0s[nd]) + F("; (void)dw%zu;\n", nd); load += shifted((*in.node_begin)[nd], nd); commit += shifted((*in.node_end)[nd], nd); }
					}
					else for (size_t nd = 0; nd < NN; nd++) if (node_here[nd]) { load += (*in.node_begin)[nd]; commit += (*in.node_end)[nd]; }
					std::vector<int> from_slot; inputs_of(pf, lv, w, from_slot);
					if (pk) {
						// the slots of what comes in and goes out, by quarter: the same operand / result of the same op in every strand of the pack
						auto slots_of = [&](int r, size_t pos, int field /* 0 a, 1 b, 2.. x, -1 dst */) {
							std::vector<long long> ids; bool dpf = false;
							for (int m = 0; m < K; m++) {
								const VOp& vm = V[(size_t)pk->ops[(size_t)m][pos]];
								const int rm = field < 0 ? vm.dst : field == 0 ? vm.a : field == 1 ? vm.b : vm.x[(size_t)field - 2];
								ids.push_back(regs[(size_t)rm].slot_id); dpf = pfx[(size_t)regs[(size_t)rm].def] != 0;
							}
							if (dpf) slot_override[r] = F("(pslots + ((%s) * 2 + %s) * (C * G))", qsel(ids).c_str(), (pf && !part) ? "parn" : "parc");
							else slot_override[r] = F("(slots + (%s) * (C * G))", qsel(ids).c_str());
						};
						for (size_t pos = 0; pos < pk->ops[0].size(); pos++) {
							const VOp& v0 = V[(size_t)pk->ops[0][pos]];
							std::vector<int> rs = { v0.a, v0.b }; rs.insert(rs.end(), v0.x.begin(), v0.x.end());
							for (size_t f = 0; f < rs.size(); f++) if (rs[f] >= 0 && std::find(from_slot.begin(), from_slot.end(), rs[f]) != from_slot.end() && !slot_override.count(rs[f])) slots_of(rs[f], pos, (int)f);
							if (v0.dst >= 0 && regs[(size_t)v0.dst].slot && def_at[(size_t)v0.dst] == pk->ops[0][pos]) slots_of(v0.dst, pos, -1);
						}
					}
					std::vector<char> predecl(regs.size(), 0);
					// the loop's inputs are fetched U samples at a time in front of the U samples that use them: one LDS round trip per batch instead of one (or more)
					// in every sample of a chain nothing else overlaps (a power of two, so that it divides C; a long loop body — many components side by side —
					// hides the round trip by itself and stays as it is)
					int U = 8; while (U > 1 && U * (int)from_slot.size() > 64) U /= 2;
					if (mine.size() > 48 || U > C) U = 1;
					std::string fetch;
					for (int r : from_slot) { fetch += F("\t\tfloat i%d[%d];\n#pragma unroll\n\t\tfor (int u = 0; u < %d; u++) i%d[u] = ", r, U, U, r) + slot_ref(r, pf) + "[(sb + u) * G + " + li + "];\n"; decl += "\t\tconst " + ty(r) + F(" r%d = i%d[u];\n", r, r); }
					for (int i : mine) if (V[(size_t)i].dst >= 0 && in_branch(i) && def_at[(size_t)V[(size_t)i].dst] == i) { const int r = V[(size_t)i].dst; predecl[(size_t)r] = 1; decl += "\t\t" + ty(r) + F(" r%d = 0; (void)r%d;\n", r, r); }
					emit_ops(pf, lv, w, "q", loop, predecl);
					merge_complementary(loop);
					if (first_pass && !part) P.serial_ops += (int)mine.size() * K;
					// the suffix works on the architectural records; the prefix on its own two copies: from the one its previous chunk left, into the other
					const std::string from = (pf && !part) ? "srecp + (parn ^ 1) * (NW * G)" : "srec", to = (pf && !part) ? "srecp + parn * (NW * G)" : "srec";
					const std::string lanes = F("ln < %d", G * K);
					code += F("\t\tif (%s && sw == %d && %s) { auto& L = Ls; const FxCtx& c = cs; (void)L; (void)c;\n", cond, w, lanes.c_str()) + pack_decl + F("\t\t{ const StagedRec r = { { %s + %s } }; (void)r;\n", from.c_str(), li.c_str()) + load + "\t\t}\n" + inv_prelude(pf, lv, w);
					code += (part ? F("\t\tfor (int sb = OFF; sb < OFF + CC; sb += %d) {\n", U) : F("\t\tfor (int sb = 0; sb < C; sb += %d) {\n", U)) + fetch + F("#pragma unroll\n\t\tfor (int u = 0; u < %d; u++) { const int q = (sb + u) * G + %s; (void)q;\n", U, li.c_str()) + decl + loop + "\t\t}\n\t\t}\n";
					const std::string cm = F("\t\t{ StagedRec r = { { %s + %s } }; (void)r;\n", to.c_str(), li.c_str()) + commit + "\t\t}\n";
					if (pf || lv > guard_level) code += cm + "\t\t}\n";
					else { code += "\t\t}\n"; deferred.push_back(F("\t\tif (ok && sw == %d && %s) { auto& L = Ls; (void)L;\n", w, lanes.c_str()) + pack_decl + cm + "\t\t}\n"); }
					if (pf && first_pass) {                                                // ... and, once its chunk is complete, from its copy to the architectural one
						P.prefix_commit += F("\t\tif (ok && sw == %d && %s) { auto& L = Ls; (void)L;\n", w, lanes.c_str()) + pack_decl + F("\t\t{ const StagedRec r = { { srecp + parc * (NW * G) + %s } }; (void)r;\n", li.c_str()) + load + F("\t\t}\n\t\t{ StagedRec r = { { srec + %s } }; (void)r;\n", li.c_str()) + commit + "\t\t}\n\t\t}\n";
					}
				}
				cur_sub = -1; slot_override.clear();
				}
			}
			return code;
		};

		s += F("\n// ---- the staged form (klg_graph_staged.hpp): %d instances x %d samples per workgroup, %d levels%s, %d values through LDS ----\n", G, C, last_level + 1, pipelined ? F(" beside the %d of the next chunk's control path", pmax + 1).c_str() : "", nslots + npslots);
		s += "__device__ __forceinline__ int ring_at(int p0, int j, int size) { const int p = p0 + j; return p >= size ? p - size : p; }\n";
		s += "struct StagedWords { uint32_t* p; __device__ __forceinline__ uint32_t& operator[](int i) const { return p[i * " + std::to_string(G) + "]; } };\n";
		s += "struct StagedRec { StagedWords w; };\n";
		s += F("extern \"C\" __global__ __launch_bounds__(%d) void klg_fx_staged(const FxGraphArgs a) {\n", NT);
		s += F("\tconstexpr int G = %d, C = %d, NTP = %d, NT = %d, NW = %d, CH = %d, SW0 = %d;\n", G, C, NTP, NT, NW, CH, own_waves ? NTP / 64 : 0);
		const char* stamp_env = getenv("KLG_FX_STAGED_STAMP");                       // (measurement only: workgroup 0 prints what its chunks spent between the barriers, 10 ns units: top of chunk, each level, tail)
		const bool stamp = stamp_env && stamp_env[0] == '1';
		s += "\tusing P = PatchGen;\n\textern __shared__ float lds[];\n";
		if (stamp) s += "\tconst long long tstart = wall_clock64();\n\tint pcount[5] = { 0, 0, 0, 0, 0 };                                              // chunks that failed their check; parts that passed; parts cut in two; plain walks; catch-ups of the control path\n";
		s += "\tuint32_t* const srec = reinterpret_cast<uint32_t*>(lds);                 // [NW][G]: the G records between chunks\n";
		s += F("\tuint32_t* const srecp = srec + NW * G;                                    // [2][NW][G]: the control path's own copies (it runs a chunk ahead)%s\n", pipelined ? "" : " — unused");
		s += F("\tfloat* const tile0 = lds + NW * G * %d;                                   // [%d][CH][C][G]: the caller's block, chunk by chunk (two: by the chunk's parity)\n", pipelined ? 3 : 1, two_tiles ? 2 : 1);
		s += "\tfloat* tile = tile0; int tile_turn = 0; (void)tile_turn;\n";
		s += F("\tfloat* const slots = tile0 + %d * CH * C * G;                              // [slots][C][G]: values that cross levels\n", two_tiles ? 2 : 1);
		s += F("\tfloat* const pslots = slots + %d * C * G;                                 // [slots][2][C][G]: those of the control path, by the parity of their chunk\n", nslots);
		s += F("\tfloat* const incopy = pslots + %d * C * G;                                  // [CH][C][G]: the chunk's `in` as it came (the tile is overwritten by the outputs)%s\n", 2 * npslots, any_near ? "" : " — unused");
		s += F("\tint* const flag = reinterpret_cast<int*>(incopy + %d);\n\t(void)incopy;\n", any_near ? CH * C * G : 0);
		s += "#define SL(k) (slots + (k) * (C * G))\n#define SLP(k, par) (pslots + ((k) * 2 + (par)) * (C * G))\n";
		s += "\tconst int t = threadIdx.x, tp = t < NTP ? t : 0, ps = tp / G, pg = tp % G, wv = t >> 6, sw = wv - SW0, ln = t & 63, k0 = blockIdx.x * G;   // sw: which wave of the serial levels\n";
		s += "\tconst int sg = ln < G ? ln : 0;                                           // the instance a lane of a serial level works for\n";
		s += "\t(void)srecp; (void)pslots;\n";
		s += "\tfor (int i = t; i < NW * G; i += NT) srec[i] = a.state[(size_t)(i / G) * a.kpad + k0 + (i % G)];\n";
		s += "\tFxCtx cp, cs;\n\tcp.fs = cs.fs = a.fs; cp.samples = cs.samples = a.samples;\n";
		s += "\tunsigned long long samples0 = a.samples; float* io = a.io;                // of the block being processed (a span: klg_fx_render_device)\n";
		s += "\tcp.ctl = a.controls + (size_t)(k0 + pg) * KLG_MAX_CTL; cs.ctl = a.controls + (size_t)(k0 + sg) * KLG_MAX_CTL;\n";
		s += "\tfloat* const ring0 = a.rings + (size_t)blockIdx.x * a.ring_rows * G;             // this workgroup's ring tile: [line][position][G]\n\tcp.ring = ring0 + pg; cs.ring = ring0 + sg;\n";
		s += "\tcp.rand = cs.rand = nullptr; cp.rstride = cs.rstride = a.rstride;\n";
		s += "\tP::Live Lp, Lq, Ls;\n\tLp.unused_ = 0; Lp.sidx = 0; Lq.unused_ = 0; Lq.sidx = 0; Ls.unused_ = 0; Ls.sidx = 0;\n";
		s += "\t__syncthreads();\n";
		// the plain body over [from, from + count) of the block, on one lane per instance (wave 0): prepare() at the head of the block, chunks whose check failed, a ragged tail
		s += "\tauto plain = [&](int from, int count, bool with_prepare, int tq0) {             // tq0: where `from` stands in the chunk's tile\n";
		s += "\t\tif (wv != 0 || ln >= G) return;\n\t\tP::Rec rec;\n#pragma unroll\n\t\tfor (int w = 0; w < NW; w++) rec.w[w] = srec[w * G + ln];\n";
		s += "\t\tP::Live L; FxCtx c = cs; c.samples = samples0 + (unsigned long long)from;\n";
		s += "\t\tif (with_prepare) P::begin(L, rec, c); else P::begin_core(L, rec, c);\n\t\tL.sidx = from;\n";
		s += "\t\tfor (int q = 0; q < count; q++) {\n\t\t\tconst float in0 = tile[(0 * C + tq0 + q) * G + ln], in1 = CH > 1 ? tile[(1 * C + tq0 + q) * G + ln] : 0.f;\n\t\t\tfloat out0 = 0.f, out1 = 0.f;\n";
		s += "\t\t\tP::sample(L, c, in0, in1, out0, out1);\n\t\t\ttile[(0 * C + tq0 + q) * G + ln] = out0;\n\t\t\tif (CH > 1) tile[(1 * C + tq0 + q) * G + ln] = out1;\n\t\t}\n";
		s += "\t\tP::end(L, rec);\n#pragma unroll\n\t\tfor (int w = 0; w < NW; w++) if (patch_stores<P>(w)) srec[w * G + ln] = rec.w[w];\n\t};\n";
		if (retry) s += "\tint part_cc = C / 2;                                                          // the largest part that passed in the last chunk that failed its check (0: none)\n";
		s += "\tconst int nblk = a.blocks > 1 ? a.blocks : 1;\n\tfor (int blk = 0; blk < nblk; blk++) {                                          // Effect::process(buffer), block after block (klang.h:4208-4216)\n";
		s += "\tsamples0 = a.samples + (unsigned long long)blk * (unsigned long long)a.n; io = a.io + (size_t)blk * a.block_stride; cp.samples = cs.samples = samples0;\n";
		s += "\tif (a.rand) { cp.rand = a.rand + (size_t)blk * (size_t)a.K + (size_t)(k0 + pg < a.K ? k0 + pg : 0); cs.rand = a.rand + (size_t)blk * (size_t)a.K + (size_t)(k0 + sg < a.K ? k0 + sg : 0); }   // this block's columns of the span's draws\n";
		if (g.prepare_ops > 0) s += "\tplain(0, 0, true, 0);                                                           // Effect::prepare(): once per block\n\t__syncthreads();\n";
		if (pipelined) s += "\tfor (int i = t; i < NW * G; i += NT) { srecp[i] = srec[i]; srecp[NW * G + i] = srec[i]; }\n\t__syncthreads();\n";
		// what the lanes hold for the whole block: the dials, the members process() only reads
		{
			const std::string pb;                                                     // (members process() only reads are taken from the LDS records where they are used: op_text)
			s += "\t{ auto& L = Lp; const FxCtx& c = cp; const StagedRec r = { { srec + pg } }; (void)L; (void)c; (void)r;\n" + in.ctl_begin + pb + "\t}\n";
			s += "\t{ auto& L = Lq; const FxCtx& c = cp; const StagedRec r = { { srec + pg } }; (void)L; (void)c; (void)r;\n" + in.ctl_begin + pb + "\t}\n";
			s += "\t{ auto& L = Ls; const FxCtx& c = cs; const StagedRec r = { { srec + sg } }; (void)L; (void)c; (void)r;\n" + in.ctl_begin + pb + "\t}\n";
		}
		// registers handed from the control path to the audio path: computed during the previous iteration (xn), taken at the top of this one (xc)
		std::string xdecl, xrot, pcdecl, cdecl, cdecl_sub;
		for (size_t r = 0; r < regs.size(); r++) if (regs[r].def >= 0 && !ser_of(regs[r].def)) {
			const bool dpf = pfx[(size_t)regs[r].def] != 0;
			if (dpf && regs[r].xrot) { xdecl += "\t" + ty((int)r) + F(" xn_r%zu = 0, xc_r%zu = 0; (void)xc_r%zu;\n", r, r, r); xrot += F("\t\txc_r%zu = xn_r%zu;\n", r, r); cdecl += "\t\tconst " + ty((int)r) + F(" r%zu = xc_r%zu; (void)r%zu;\n", r, r, r); }
			else if (dpf && regs[r].chunk) pcdecl += "\t\t" + ty((int)r) + F(" pc_r%zu = 0; (void)pc_r%zu;\n", r, r);
			else if (!dpf && regs[r].chunk) { cdecl += "\t\t" + ty((int)r) + F(" r%zu = 0; (void)r%zu;\n", r, r); cdecl_sub += "\t\t" + ty((int)r) + F(" r%zu = 0; (void)r%zu;\n", r, r); }
		}
		for (size_t nd = 0; nd < NN; nd++) if (g.nodes[nd] == N_DELAY && set_at[nd] >= 0 && pfx[(size_t)set_at[nd]]) { xdecl += F("\tTap xn_d%zut = { 0, 0.f }, xc_d%zut = { 0, 0.f }; (void)xc_d%zut;\n", nd, nd, nd); xrot += F("\t\txc_d%zut = xn_d%zut;\n", nd, nd); }
		s += xdecl;
		// every Delay's write cursor is the sample counter (one step per input()): where it stands at the start of the block (64-bit once), 32-bit from there
		for (size_t nd = 0; nd < NN; nd++) if (g.nodes[nd] == N_DELAY) s += F("\tconst unsigned d%zub = (unsigned)((samples0 * %dull) %% %dull);\n", nd, k_in[nd], g.arg((int)nd));
		// the caller's rows of a chunk (a row's C samples are contiguous in the caller's block: G * CH * C = CH * NT values, CH per thread) are requested a chunk
		// ahead and put into the tile at the top of their chunk
		s += "\tfloat nx[CH];\n";
		s += "\tauto fetch = [&](int s0) {\n#pragma unroll\n\t\tfor (int j = 0; j < CH; j++) { const int i = tp + j * NTP, row = i / C, q = i % C, gi = row / CH, ch = row % CH;\n";
		s += "\t\t\tnx[j] = (t < NTP && s0 + q < a.n && k0 + gi < a.K) ? io[((size_t)(k0 + gi) * CH + ch) * a.n + s0 + q] : 0.f; }\n\t};\n";
		s += "\tfetch(0);\n";
		if (pipelined) {                                                           // the control path of chunk 0
			s += "\t{ const int s0 = -C; const bool pre = a.n >= C; const int parn = 0; (void)parn;\n" + pcdecl;
			for (int lv = 0; lv <= pmax; lv++) { const std::string code = block(true, lv); if (code.empty()) continue; s += F("\t\t// ---- control path of chunk 0, level %d ----\n", lv) + code + "\t\t__syncthreads();\n"; }
			s += "\t}\n";
			first_pass = false;
		}
		if (stamp) s += "\tlong long tacc[16] = { 0 }; const long long thead = wall_clock64() - tstart; long long tprev = wall_clock64();\n";
		s += "\tfor (int s0 = 0; s0 < a.n; s0 += C) {\n\t\tconst int cl = (a.n - s0 < C) ? (a.n - s0) : C;\n";
		// The thread index is laundered through an empty asm once per chunk: otherwise every per-lane LDS address of the chunk's body (a value's slot + this lane's
		// place in it: the slots lie beyond an instruction's 16-bit offset) is loop-invariant, gets hoisted out of the chunk loop and then spilled — the recorded
		// Reverb.k's kernel held ~500 of them in scratch (3.7 KB per lane; 1,123 spilled registers) and moved 5 x its algorithmic bytes
		if (two_tiles) s += "\t\ttile = tile0 + (tile_turn & 1) * (CH * C * G); tile_turn++;          // (counted across the blocks of a span)\n";
		s += "\t\tint tv = threadIdx.x; asm volatile(\"\" : \"+v\"(tv));\n";
		s += "\t\tconst int t = tv, tp = t < NTP ? t : 0, ps = tp / G, pg = tp % G, wv = t >> 6, sw = wv - SW0, ln = t & 63; (void)ps; (void)pg; (void)sw; (void)ln; (void)wv;\n";
		s += "#pragma unroll\n\t\tfor (int j = 0; j < CH; j++) { const int i = tp + j * NTP, row = i / C, q = i % C, gi = row / CH, ch = row % CH; if (t < NTP) { tile[(ch * C + q) * G + gi] = nx[j];" + std::string(any_near ? " incopy[(ch * C + q) * G + gi] = nx[j];" : "") + " } }\n";
		s += "\t\tif (t == 0) *flag = 0;\n\t\t__syncthreads();\n";
		s += "\t\tif (s0 + C < a.n) fetch(s0 + C);\n";
		// (Leaving out the attempt on the WHOLE chunk after two failures in a row — taps that stay inside the chunk fail it every time — was measured and lost: the
		//  attempt's ring reads are what brings the parts' rows into the cache: fx_short.k at 12 / 22 samples 60.2 -> 62.6 / 44.7 -> 46.0 us, PingPong.k random dials 72.7 -> 80.7)
		if (retry) s += "\t\tconst bool probe = ((s0 / C) & 7) == 0; (void)probe;\n\t\tbool ok = cl == C;\n\t\tint bad = 0; (void)bad;\n";
		else s += "\t\tbool ok = cl == C;\n\t\tint bad = 0; (void)bad;\n";
		s += "\t\tconst bool pre = s0 + 2 * C <= a.n; const int parc = (s0 / C) & 1, parn = parc ^ 1; (void)pre; (void)parc; (void)parn;\n";
		s += "\t\tconst float in0 = tile[(0 * C + ps) * G + pg], in1 = CH > 1 ? tile[(1 * C + ps) * G + pg] : 0.f; (void)in0; (void)in1;\n";
		s += "\t\tLp.sidx = s0 + ps;\n";
		s += xrot + pcdecl;
		// delay lines: this lane's cursor at the start of its sample, the rows the chunk writes, the head
		for (size_t nd = 0; nd < NN; nd++) if (g.nodes[nd] == N_DELAY) {
			const int SZ = g.arg((int)nd), w0 = g.node_word0((int)nd);
			s += F("\t\tconst int d%zup0 = (int)((d%zub + (unsigned)(s0 + ps) * %du) %% %du); (void)d%zup0;\n", nd, nd, k_in[nd], SZ, nd);
			s += F("\t\tconst RingWindow d%zuw = { (int)((d%zub + (unsigned)s0 * %du) %% %du), %d * C }; (void)d%zuw;\n", nd, nd, k_in[nd], SZ, k_in[nd], nd);
			s += F("\t\tconst int d%zunw0 = d%zuw.w0; (void)d%zunw0;\n", nd, nd, nd);
			if (set_at[nd] >= 0 && pfx[(size_t)set_at[nd]]) s += F("\t\tconst Tap d%zut = xc_d%zut; (void)d%zut;\n", nd, nd, nd);
			else if (set_at[nd] >= 0) s += F("\t\tTap d%zut = { 0, 0.f }; (void)d%zut;\n", nd, nd);
			else if (outs[nd] > 0) s += F("\t\tconst Tap d%zuh = { (int)srec[%d * G + pg], u2f(srec[%d * G + pg]) };\n", nd, w0 + ED_LASTPOS, w0 + ED_LASTFRAC);
		}
		s += cdecl;
		bool guard_emitted = guard_level < 0;
		if (stamp) s += "\t\t{ const long long now = wall_clock64(); tacc[0] += now - tprev; tprev = now; }\n";
		for (int lv = 0; lv <= std::max(last_level, pipelined ? pmax - poff : -1); lv++) {
			std::string code;
			if (lv <= last_level) code += block(false, lv);
			if (pipelined && lv + poff <= pmax) code += block(true, lv + poff);
			if (code.empty() && lv != guard_level) continue;
			s += F("\t\t// ---- interval %d ----\n", lv) + code + "\t\t__syncthreads();\n";
			if (stamp) s += F("\t\t{ const long long now = wall_clock64(); tacc[%d] += now - tprev; tprev = now; }\n", lv + 1);
			if (lv == guard_level && !guard_emitted) { s += "\t\tok = ok && *flag == 0;                                                   // every ring read of the chunk lies outside the rows the chunk writes\n"; guard_emitted = true; }
		}
		{ std::string rest; for (const std::string& d : deferred) rest += d; deferred.clear(); rest += P.prefix_commit; if (!rest.empty()) s += rest + "\t\t__syncthreads();\n"; }
		if (retry) {
			// the chunk again in parts (the audio path alone): halves, a half that fails in quarters, and only a part of CMIN samples that fails goes to the plain body
			// (a ragged last chunk takes the same way out with nothing to try: one copy of the plain body.  What the part's code needs of the chunk's own values is
			// taken again rather than kept — the thread index laundered once more, `in` from the tile, the cursors —: kept, they would stay live across every level
			// of every chunk for the sake of a path that is almost never taken: the recorded PingPong.k's kernel then spilled 36 registers, 9 this way)
			if (stamp) s += "\t\tif (!ok && cl == C) pcount[0]++;\n";
			// (the parts start at the size that got through the last failed chunk — none: the plain body at once, as if there were no parts — and one chunk in
			//  eight starts at halves again: what a short comb costs is its parts, not the attempts that come before them — fx_short.k, 4,096 instances, taps 5 / 12
			//  samples behind the cursor: 131.8 -> 116.6 / 60.2 -> 55.9 us per block; a failed chunk straight to the plain body: 110.6 / 108.7)
			s += "\t\tif (!ok) {\n\t\tconst int start_cc = probe ? C / 2 : part_cc;\n";
			s += "\t\tconst bool try_parts = cl == C && start_cc != 0;\n";
			s += "\t\tint OFF = 0, CC = try_parts ? start_cc : cl, ctl_at = 0, best = 0; (void)ctl_at; (void)best;   // ctl_at: the sample of the chunk the control path's ARCHITECTURAL records stand at\n\t\twhile (OFF < cl) {\n";
			s += "\t\tint tv = threadIdx.x; asm volatile(\"\" : \"+v\"(tv));\n";
			s += "\t\tconst int t = tv, tp = t < NTP ? t : 0, ps = tp / G, pg = tp % G, wv = t >> 6, sw = wv - SW0, ln = t & 63; (void)ps; (void)pg; (void)sw; (void)ln; (void)wv;\n";
			s += "\t\tif (t == 0) *flag = 0;\n\t\t__syncthreads();\n\t\tbool ok = try_parts; int bad = 0; (void)bad;\n";
			s += "\t\tconst float in0 = tile[(0 * C + ps) * G + pg], in1 = CH > 1 ? tile[(1 * C + ps) * G + pg] : 0.f; (void)in0; (void)in1;\n";
			for (size_t nd = 0; nd < NN; nd++) if (g.nodes[nd] == N_DELAY) {
				const int SZ = g.arg((int)nd), w0 = g.node_word0((int)nd);
				s += F("\t\tconst int d%zup0 = (int)((d%zub + (unsigned)(s0 + ps) * %du) %% %du); (void)d%zup0;\n", nd, nd, k_in[nd], SZ, nd);
				s += F("\t\tconst RingWindow d%zuw = { (int)((d%zub + (unsigned)(s0 + OFF) * %du) %% %du), %d * CC }; (void)d%zuw;\n", nd, nd, k_in[nd], SZ, k_in[nd], nd);
				if (set_at[nd] >= 0 && pfx[(size_t)set_at[nd]]) {}
				else if (set_at[nd] >= 0) s += F("\t\tTap d%zut = { 0, 0.f }; (void)d%zut;\n", nd, nd);
				else if (outs[nd] > 0) s += F("\t\tconst Tap d%zuh = { (int)srec[%d * G + pg], u2f(srec[%d * G + pg]) };\n", nd, w0 + ED_LASTPOS, w0 + ED_LASTFRAC);
			}
			s += cdecl_sub;
			part = true;
			bool guard_seen = guard_level < 0;
			for (int lv = 0; lv <= last_level; lv++) {
				const std::string code = block(false, lv);
				if (code.empty() && lv != guard_level) continue;
				s += F("\t\t// ---- a part of the chunk, level %d ----\n", lv) + code + "\t\t__syncthreads();\n";
				if (lv == guard_level && !guard_seen) { s += "\t\tok = ok && *flag == 0;\n"; guard_seen = true; }
			}
			{ std::string rest; for (const std::string& d : deferred) rest += d; deferred.clear(); if (!rest.empty()) s += rest + "\t\t__syncthreads();\n"; }
			// The plain body starts from the architectural records.  The control path's are where the last plain walk left them (ctl_at; the chunk's start at
			// first) — parts that passed ran the audio path alone —: its serial loops run over [ctl_at, OFF) on those records first (the same arithmetic on the
			// inputs the run a chunk ahead left in LDS), and only then: nothing of the control path is repeated for a chunk whose parts all pass.
			std::string catchup;
			if (pipelined) for (int lv = 1; lv <= pmax; lv += 2) catchup += block(true, lv);
			part = false;
			if (stamp) s += F("\t\tif (ok) pcount[1]++; else if (try_parts && CC > %d) pcount[2]++; else { if (cl == C) pcount[3]++; if (OFF > ctl_at) pcount[4]++; }\n", CMIN);
			s += F("\t\tif (ok) { OFF += CC; best = CC > best ? CC : best; } else if (try_parts && CC > %d) CC >>= 1; else {\n", CMIN);
			if (!catchup.empty()) s += "\t\tif (OFF > ctl_at) { const int part_at = OFF; { const int OFF = ctl_at, CC = part_at - ctl_at; const bool ok = true; (void)ok;\n" + catchup + "\t\t} __syncthreads(); }\n";
			s += "\t\tplain(s0 + OFF, CC, false, OFF); OFF += CC; ctl_at = OFF; }\n";
			s += "\t\t__syncthreads();                                                          // (the next part resets the flag this one's lanes have read)\n\t\t}\n\t\tif (cl == C) part_cc = best;\n";
			if (!P.prefix_commit.empty()) s += "\t\tif (cl == C) { const bool ok = true; (void)ok;                                  // the control path's records at the end of the chunk (where a plain walk ended the chunk they are there already: the same values)\n" + P.prefix_commit + "\t\t}\n\t\t__syncthreads();\n";
			s += "\t\t}\n";
		}
		else s += "\t\tif (!ok) { plain(s0, cl, false, 0); __syncthreads(); }\n";
		s += "\t\tfor (int i = t; i < G * CH * C; i += NT) { const int row = i / C, q = i % C, gi = row / CH, ch = row % CH;\n";
		s += "\t\t\tif (q < cl && k0 + gi < a.K) io[((size_t)(k0 + gi) * CH + ch) * a.n + s0 + q] = tile[(ch * C + q) * G + gi]; }\n";
		if (!two_tiles) s += "\t\t__syncthreads();\n";                            // (two tiles: the next chunk's input goes to the other one, and this one is not touched before the barriers of that chunk)
		if (stamp) s += "\t\t{ const long long now = wall_clock64(); tacc[15] += now - tprev; tprev = now; }\n";
		s += "\t}\n";

		if (stamp) s += "\tif (t == 0 && blockIdx.x == 0) { printf(\"staged stamps (10 ns): head %lld |\", thead); for (int i = 0; i < 16; i++) printf(\" %lld\", tacc[i]); printf(\"\\n\"); }\n";
		if (stamp && retry) s += "\tif (t == 0 && blockIdx.x == 0) printf(\"staged parts: %d chunks failed their check, %d parts passed, %d cut in two, %d walked by the plain body, %d control catch-ups\\n\", pcount[0], pcount[1], pcount[2], pcount[3], pcount[4]);\n";
		s += "\t}                                                                          // (the next block of the span)\n";
		s += "\tfor (int i = t; i < NW * G; i += NT) if (k0 + (i % G) < a.K && patch_stores<P>(i / G)) a.state[(size_t)(i / G) * a.kpad + k0 + (i % G)] = srec[i];\n";
		s += "#undef SL\n#undef SLP\n}\n";
		for (int i = 0; i < NV; i++) if (live_op(i) && !ser_of(i)) P.parallel_ops++;
		P.ok = true; P.G = G; P.C = C; P.threads = NT; P.lds_bytes = (int)(lds_words * 4); P.levels = last_level + 1; P.slots = nslots + npslots; P.pipelined = pipelined; P.retry_min = retry ? CMIN : 0; P.source = s;
		return P;
	}
}

} }
