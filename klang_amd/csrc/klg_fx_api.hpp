// klang_amd/csrc/klg_fx_api.hpp — C-ABI entry points of the effect banks (included by klg_api.hip).
//
// Host side: controls (clamping, Controls::changed()), and the patches' prepare() code — everything that runs
// once per block or per control change in the reference (rand(), expf, cosf/sinf coefficient design, tap-table
// construction) stays on the CPU and is shipped to the lanes as state-word updates.  Per-sample code is device only.
#pragma once

struct FxUpdate { int k, word, bits; };

struct RvFilterCache { float f = 0.f, Q = 0.f; };
struct RvHost {                                        // host mirror of one Reverb instance's prepare() state
	float cache[10] = { 0 };                           // Controls::value[] klang.h:1878
	float e_length = 0.f, e_size = 0.f;                // EarlyReflections::length / size
};

// a bank sharded BY INSTANCE over the devices of klg_init (effects shard by instance, no collective): contiguous ranges,
// one ordinary single-device bank per device; this handle only routes
struct FxMulti { std::vector<struct klg_fx*> shard; std::vector<int> first; };   // shard i owns instances [first[i], first[i + 1])

struct klg_fx {
	int patch = 0, K = 0, max_block = 0, nctl = 0, words = 0;
	int device = 0;                                    // the GPU this bank (or shard) lives on
	FxMulti* multi = nullptr;
	size_t kpad = 0;
	host::Fs fs;
	hipStream_t stream = nullptr;
	float *d_state = nullptr, *d_rings = nullptr, *d_rings2 = nullptr, *d_io = nullptr;
	// Reverb (layout 1): the block's early-reflection sums [kpad][2][max_block] (klg_fx_reverb_early -> klg_fx_reverb_q<true>), two buffers in turn:
	// block b + 1's sums are computed on `early_stream` while block b's recursive kernel runs (they depend on samples >= 45 ms old only)
	float* d_early[2] = { nullptr, nullptr }; unsigned early_turn = 0; bool upd_flushed = false;
	hipStream_t early_stream = nullptr; hipEvent_t early_ready = nullptr, q_done[2] = { nullptr, nullptr }, upd_done = nullptr; bool q_used[2] = { false, false };
	bool early_alloc_failed = false; int rv_prev_n = 1 << 30; hipStream_t rv_prev_stream = nullptr;   // (mode 2: length and stream of the previous Reverb block)
	int* d_upd = nullptr; size_t d_upd_cap = 0;
	unsigned long long samples = 0;                    // samples processed so far (defines every write cursor)
	unsigned long long pp_touched_at = 0;              // PingPong: `samples` when a dial / record word was last written (a fresh bank: 0 — its smoothers start converging there)
	int* pp_done = nullptr;                            // PingPong spans in two launches: [kpad / 16] which workgroups the first one rendered (klg_fx.hpp PingPongArgs::done)
	std::vector<host::ControlH> controls;              // [K][nctl]
	std::vector<FxUpdate> upd;
	std::vector<RvHost> rv;
	std::vector<int> rv_touched; std::vector<unsigned char> rv_flag;   // Reverb instances whose dials were set since their last prepare() (Controls::changed() is only evaluated for those)
	BiquadCoef pp_dc;
	int lds_limit = 64 * 1024;                         // hipDeviceAttributeMaxSharedMemoryPerBlock of this bank's device (gfx950: 160 KB)
	int rv_layout = 1;                                 // Reverb ring layout: 1 = a contiguous ring per (instance, line) [klg_fx_reverb_q] — the only one since klg_fx_reverb16 was retired
	bool timing = false; std::vector<hipEvent_t> tev; int launches = 0;
	// graph effects (klg_graph.hpp, `kind effect`): hipRTC code object, per-instance controls in HBM
	const graphrt::Compiled* graph = nullptr;
	hipModule_t module = nullptr; hipFunction_t graph_fn = nullptr, staged_fn = nullptr;   // staged_fn: klg_fx_staged of the same code object, when the body has a sample-parallel form
	int channels = 2;
	float* d_controls = nullptr; std::vector<float> h_controls; bool controls_dirty = false;
	// Noise ops: the span's rand() draws [n * draws][blocks * K], produced on the device (klg_rand_fill); rand_done: the last launch that read them
	int* d_rand = nullptr; size_t rand_cap = 0; hipEvent_t rand_done = nullptr; hipStream_t rand_stream = nullptr;
};
enum { KLG_PATCH_FXGRAPH = 1001 };

static inline int f2i(float f) { int i; std::memcpy(&i, &f, 4); return i; }

static void fx_free(klg_fx* f) {
	if (!f) return;
	if (f->multi) { for (klg_fx* sh : f->multi->shard) { DeviceGuard bound(sh->device); fx_free(sh); } delete f->multi; delete f; return; }
	if (f->stream) (void)hipStreamSynchronize(f->stream);
	void* dev[] = { f->d_state, f->d_rings, f->d_rings2, f->d_io, f->d_upd, f->d_controls, f->d_early[0], f->d_early[1] };
	if (f->early_stream) { (void)hipStreamSynchronize(f->early_stream); (void)hipStreamDestroy(f->early_stream); }
	for (hipEvent_t e : { f->early_ready, f->q_done[0], f->q_done[1], f->upd_done }) if (e) (void)hipEventDestroy(e);
	for (void* p : dev) if (p) (void)hipFree(p);
	if (f->module) (void)hipModuleUnload(f->module);
	for (auto e : f->tev) (void)hipEventDestroy(e);
	if (f->d_rand) (void)hipFree(f->d_rand);
	if (f->pp_done) (void)hipFree(f->pp_done);
	if (f->rand_done) (void)hipEventDestroy(f->rand_done);
	if (f->stream) (void)hipStreamDestroy(f->stream);
	delete f;
}

static const DialDef PP_DIALS[6] = { { 0.0f, 0.999f, 0.5f }, { 0.001f, 1.0f, 0.5f }, { 0.0f, 1.0f, 0.0f }, { 0.01f, 1.0f, 0.0f }, { 0.001f, 2.0f, 1.0f }, { 0.f, 1.f, 0.f } };   // PingPong.k:14-21
static const DialDef RV_DIALS[10] = { { 0, 1, 0 }, { 0, 1, 1 }, { 0, 1, 0 }, { 0, 1, 0 }, { 0, 1, 1 }, { 0, 100, 10 }, { 0, 1, 1 }, { 0.01f, 1, 1 }, { 0.01f, 1, 1 }, { 0, 0.2f, 0 } };   // Reverb.k:100-113

// Biquad coefficient design on the host (klang.h:5584-5600 + LPF/HPF::init 5658-5682); glibc cosf/sinf
static BiquadCoef design_biquad(bool hpf, float f, float Q, const host::Fs& fs) {
	if (Q < 0) Q = f / -Q;
	const float w = f * fs.w;
	const float cos0 = cosf(w), sin0 = sinf(w);
	if (Q < 0.5) Q = 0.5f;
	const float a = sin0 / (2.f * Q);
	const double a0 = (double)(1.f + a);
	const float inv = (a0 == 0.0f) ? 0.0f : (float)(1.0 / a0);
	BiquadCoef c;
	c.a1 = inv * (-2.f * cos0);
	c.a2 = inv * (1.f - a);
	if (!hpf) { c.b2 = c.b0 = inv * (1.f - cos0) * 0.5f; c.b1 = inv * (1.f - cos0); }
	else { c.b2 = c.b0 = inv * (1.f + cos0) * 0.5f; c.b1 = inv * -(1.f + cos0); }
	return c;
}

// `make(device, instances)` = the single-device creator; the bank's instances are dealt to the devices in contiguous ranges
template<class MAKE> static klg_fx* fx_multi_create(int instances, int max_block, MAKE&& make) {
	const std::vector<int> devs = g_devices;
	const int n = (int)std::min<size_t>(devs.size(), (size_t)instances);
	klg_fx* r = new klg_fx(); r->multi = new FxMulti(); r->K = instances; r->max_block = max_block; r->device = devs[0];
	const int base = instances / n, extra = instances % n;
	r->multi->first.push_back(0);
	for (int i = 0; i < n; i++) {
		const int count = base + (i < extra ? 1 : 0);
		klg_fx* sh = make(devs[(size_t)i], count);
		if (!sh) { const std::string why = g_err; fx_free(r); fail(KLG_ERR_NOMEM, "multi-device effect bank: shard %d on device %d: %s", i, devs[(size_t)i], why.c_str()); return nullptr; }
		r->multi->shard.push_back(sh); r->multi->first.push_back(r->multi->first.back() + count);
	}
	const klg_fx* s0 = r->multi->shard[0];
	r->patch = s0->patch; r->nctl = s0->nctl; r->words = s0->words; r->channels = s0->channels; r->fs = s0->fs; r->graph = s0->graph;
	return r;
}
static int fx_shard_of(const klg_fx* f, int instance, int* local) {
	const FxMulti& m = *f->multi;
	for (size_t i = 0; i + 1 < m.first.size(); i++) if (instance >= m.first[i] && instance < m.first[i + 1]) { *local = instance - m.first[i]; return (int)i; }
	return -1;
}

static klg_fx* fx_create_on(int device, int patch_id, int instances, float sample_rate, int max_block);
extern "C" klg_fx* klg_fx_create(int patch_id, int instances, float sample_rate, int max_block) {
	if (patch_id != KLG_PATCH_PINGPONG && patch_id != KLG_PATCH_REVERB) { fail(KLG_ERR_INVALID, "klg_fx_create: patch %d is not an effect patch", patch_id); return nullptr; }
	if (instances <= 0 || max_block <= 0 || max_block > MAX_BLOCK || !(sample_rate > 0.f)) { fail(KLG_ERR_INVALID, "klg_fx_create: bad arguments"); return nullptr; }
	if (default_device() < 0) return nullptr;
	if (g_devices.size() > 1) return fx_multi_create(instances, max_block, [&](int device, int count) { return fx_create_on(device, patch_id, count, sample_rate, max_block); });
	return fx_create_on(g_device, patch_id, instances, sample_rate, max_block);
}
// the same on a NAMED device, whatever klg_init() said: one bank on one GPU (a rank of a sharded effect bank; several banks of one process on different GPUs).
// `program` non-NULL: a recorded effect (klg_fx_create_graph's arguments), else patch_id as for klg_fx_create
static klg_fx* fx_create_graph_on(int device, const char* program, int instances, float sample_rate, int max_block, const void* initial_record);
extern "C" klg_fx* klg_fx_create_on(int device, int patch_id, const char* program, int instances, float sample_rate, int max_block, const void* initial_record) {
	if (instances <= 0 || max_block <= 0 || max_block > MAX_BLOCK || !(sample_rate > 0.f)) { fail(KLG_ERR_INVALID, "klg_fx_create_on: bad arguments"); return nullptr; }
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); fail(KLG_ERR_NO_DEVICE, "klg_fx_create_on: no HIP device visible: libklang_mi355 has no CPU fallback"); return nullptr; }
	if (device < 0 || device >= count) { fail(KLG_ERR_INVALID, "klg_fx_create_on: device %d of %d", device, count); return nullptr; }
	if (program) return fx_create_graph_on(device, program, instances, sample_rate, max_block, initial_record);
	if (patch_id != KLG_PATCH_PINGPONG && patch_id != KLG_PATCH_REVERB) { fail(KLG_ERR_INVALID, "klg_fx_create_on: patch %d is not an effect patch", patch_id); return nullptr; }
	return fx_create_on(device, patch_id, instances, sample_rate, max_block);
}
static klg_fx* fx_create_on(int device, int patch_id, int instances, float sample_rate, int max_block) {
	RandGuard rg;
	DeviceGuard bound(device);
	if (!bound.ok) return nullptr;
	klg_fx* f = new klg_fx();
	f->device = device;
	f->patch = patch_id; f->K = instances; f->max_block = max_block;
	f->kpad = ((size_t)instances + FX_WG - 1) / FX_WG * FX_WG;
	f->fs = host::Fs(sample_rate);
	const bool pp = patch_id == KLG_PATCH_PINGPONG;
	{ int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0) f->lds_limit = v; }
	f->nctl = pp ? 6 : 10;
	f->words = pp ? (int)PP_WORDS : (int)RV_WORDS;
	if (!pp) {
		// klg_fx_reverb_q — one independent wave per four instances, a contiguous ring per (instance, line) — serves every Reverb bank (round 2's
		// sixteen-waves-per-64-instances kernel, which took banks above 8,192 instances, lost to it at every size that fits in memory and was retired in round 5)
		f->rv_layout = 1;
	}
	const size_t ring1 = pp ? (size_t)2 * PP_ROWS * f->kpad : (size_t)2 * RV_ESTRIDE * f->kpad;     // (Reverb: lines + the mirror tails of klg_fx_reverb_q)
	const size_t ring2 = pp ? 0 : (size_t)16 * RV_FSTRIDE * f->kpad;
	bool ok = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) == hipSuccess;
	ok = ok && hipMalloc(&f->d_state, (size_t)f->words * f->kpad * 4) == hipSuccess;
	ok = ok && hipMalloc(&f->d_rings, ring1 * 4) == hipSuccess;
	ok = ok && (ring2 == 0 || hipMalloc(&f->d_rings2, ring2 * 4) == hipSuccess);
	ok = ok && hipMalloc(&f->d_io, (size_t)f->kpad * 2 * max_block * 4) == hipSuccess;
	ok = ok && hipMemset(f->d_state, 0, (size_t)f->words * f->kpad * 4) == hipSuccess;
	ok = ok && hipMemset(f->d_rings, 0, ring1 * 4) == hipSuccess;                      // Delay() : buffer(SIZE + 1, 0)
	ok = ok && (ring2 == 0 || hipMemset(f->d_rings2, 0, ring2 * 4) == hipSuccess);
	if (!ok) { fail(KLG_ERR_NOMEM, "klg_fx_create: device allocation failed (%zu ring bytes): %s", (ring1 + ring2) * 4, hipGetErrorString(hipGetLastError())); fx_free(f); return nullptr; }
	f->controls.resize((size_t)instances * f->nctl);
	const DialDef* dials = pp ? PP_DIALS : RV_DIALS;
	for (int k = 0; k < instances; k++) for (int c = 0; c < f->nctl; c++) {
		f->controls[(size_t)k * f->nctl + c] = { dials[c].min, dials[c].max, dials[c].initial };
		if (pp) f->upd.push_back({ k, c, f2i(dials[c].initial) });
		else if (c < 5) f->upd.push_back({ k, RV_CTL + c, f2i(dials[c].initial) });
	}
	if (pp) f->pp_dc = design_biquad(true, 50.f, 1.f, f->fs);                          // dcfilter[k].set(50, 1)  PingPong.k:39-40
	else { f->rv.resize(instances); f->rv_flag.assign(instances, 1); f->rv_touched.resize(instances); for (int k = 0; k < instances; k++) f->rv_touched[k] = k; }
	return f;
}

// replaces: constructing `instances` copies of a user Effect whose process() body was recorded (include/klang_mi355_graph.h,
// `kind effect 1|2`).  `initial_record` (the program's record: Program::words() 32-bit words; NULL = zeros) is the state of one freshly
// constructed instance; every instance starts from it.  Delay<SIZE> members become zero-filled rings in HBM.
extern "C" klg_fx* klg_fx_create_graph(const char* program, int instances, float sample_rate, int max_block, const void* initial_record) {
	if (instances <= 0 || max_block <= 0 || max_block > MAX_BLOCK || !(sample_rate > 0.f)) { fail(KLG_ERR_INVALID, "klg_fx_create_graph: bad arguments"); return nullptr; }
	if (default_device() < 0) return nullptr;
	if (g_devices.size() > 1) return fx_multi_create(instances, max_block, [&](int device, int count) { return fx_create_graph_on(device, program, count, sample_rate, max_block, initial_record); });
	return fx_create_graph_on(g_device, program, instances, sample_rate, max_block, initial_record);
}
static klg_fx* fx_create_graph_on(int device, const char* program, int instances, float sample_rate, int max_block, const void* initial_record) {
	RandGuard rg;
	DeviceGuard bound(device);
	if (!bound.ok) return nullptr;
	const graphrt::Compiled* c = nullptr;
	// instances per workgroup of the staged form, by bank size (measured with the recorded PingPong.k, tools/staged_sweep.py: 4,096 instances 55 us at 16 per
	// workgroup, 16,384: 0.23 ms at 16, 0.17 at 32, 0.12 at 64; 65,536: 0.91 / 0.65 / 0.48 — a serial level costs a workgroup the same whatever its width, so a
	// bank that fills the chip anyway takes the widest): the widths of klg_fx_pingpong_x
	const size_t kp = ((size_t)instances + FX_WG - 1) / FX_WG * FX_WG;
	const std::string err = graphrt::compile(program, &c, false, kp <= 4096 ? 16 : kp <= 8192 ? 32 : 64);
	if (!err.empty()) { fail(KLG_ERR_INVALID, "klg_fx_create_graph: %s", err.c_str()); return nullptr; }
	if (!c->channels) { fail(KLG_ERR_INVALID, "klg_fx_create_graph: the program is a synth note body (no `kind effect` line): use klg_synth_create_graph"); return nullptr; }
	graph::Program g; (void)g.parse(program);
	klg_fx* f = new klg_fx();
	f->device = device;
	f->patch = KLG_PATCH_FXGRAPH; f->K = instances; f->max_block = max_block; f->graph = c; f->channels = c->channels;
	f->kpad = ((size_t)instances + FX_WG - 1) / FX_WG * FX_WG;
	f->fs = host::Fs(sample_rate);
	f->nctl = g.nctl; f->words = c->words;
	const size_t ring = (size_t)std::max<long long>(c->ring_rows, 1) * f->kpad;
	bool ok = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) == hipSuccess;
	ok = ok && hipMalloc(&f->d_state, (size_t)f->words * f->kpad * 4) == hipSuccess;
	ok = ok && hipMalloc(&f->d_rings, ring * 4) == hipSuccess;
	ok = ok && hipMalloc(&f->d_io, (size_t)f->kpad * f->channels * max_block * 4) == hipSuccess;
	ok = ok && hipMalloc(&f->d_controls, f->kpad * KLG_MAX_CTL * 4) == hipSuccess;
	ok = ok && hipMemset(f->d_rings, 0, ring * 4) == hipSuccess;                       // Delay() : buffer(SIZE + 1, 0)
	ok = ok && hipModuleLoadData(&f->module, c->code.data()) == hipSuccess;
	ok = ok && hipModuleGetFunction(&f->graph_fn, f->module, c->name[0].c_str()) == hipSuccess;
	{ int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0) f->lds_limit = v; }
	// the sample-parallel form of the same body (klg_graph_staged.hpp): planned for gfx950's 160 KB of LDS — a device that grants a workgroup less keeps the one-lane-per-instance
	// kernel of the same code object (same bits, klg_fx_graph<P> with kRingRow = the plan's G), as does a kernel that cannot be given its dynamic LDS
	if (ok && c->staged && c->staged_lds <= f->lds_limit && hipModuleGetFunction(&f->staged_fn, f->module, "klg_fx_staged") != hipSuccess) { (void)hipGetLastError(); f->staged_fn = nullptr; }
	if (ok && c->staged && !f->staged_fn && c->staged_lds > f->lds_limit) fprintf(stderr, "klang-mi355: the staged form of this effect needs %d bytes of LDS per workgroup, the device grants %d: one lane per instance\n", c->staged_lds, f->lds_limit);
	if (!ok) { fail(KLG_ERR_NOMEM, "klg_fx_create_graph: device allocation / module load failed (%zu ring bytes): %s", ring * 4, hipGetErrorString(hipGetLastError())); fx_free(f); return nullptr; }
	std::vector<uint32_t> init((size_t)f->words * f->kpad, 0u);
	if (initial_record) { const uint32_t* r = (const uint32_t*)initial_record; for (int w = 0; w < f->words; w++) std::fill(init.begin() + (size_t)w * f->kpad, init.begin() + (size_t)(w + 1) * f->kpad, r[w]); }
	if (hipMemcpy(f->d_state, init.data(), init.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { fail(KLG_ERR_HIP, "klg_fx_create_graph: state upload failed"); fx_free(f); return nullptr; }
	f->controls.resize((size_t)instances * std::max(1, f->nctl));
	f->h_controls.assign(f->kpad * KLG_MAX_CTL, 0.f);
	for (int k = 0; k < instances; k++) for (int i = 0; i < f->nctl; i++) {
		f->controls[(size_t)k * f->nctl + i] = { g.dials[i].min, g.dials[i].max, g.dials[i].initial };
		f->h_controls[(size_t)k * KLG_MAX_CTL + i] = g.dials[i].initial;
	}
	f->controls_dirty = true;
	return f;
}

// replaces nothing (diagnostics): how a graph effect bank runs its recorded body
extern "C" int klg_fx_graph_form(const klg_fx* f, int* instances_per_workgroup, int* samples_per_chunk, int* levels, int* lds_values, char* why, size_t why_cap) {
	if (!f) return fail(KLG_ERR_INVALID, "klg_fx_graph_form: NULL handle");
	if (f->multi) return klg_fx_graph_form(f->multi->shard[0], instances_per_workgroup, samples_per_chunk, levels, lds_values, why, why_cap);
	if (!f->graph) return fail(KLG_ERR_INVALID, "klg_fx_graph_form: not a graph effect bank (klg_fx_create_graph)");
	const bool staged = f->staged_fn != nullptr;
	if (instances_per_workgroup) *instances_per_workgroup = staged ? f->graph->staged_G : FX_WG;
	if (samples_per_chunk) *samples_per_chunk = staged ? f->graph->staged_C : 1;
	if (levels) *levels = staged ? f->graph->staged_levels : 1;
	if (lds_values) *lds_values = staged ? f->graph->staged_slots : 0;
	if (why && why_cap) { snprintf(why, why_cap, "%s", staged ? "" : f->graph->staged_why.c_str()); }
	return staged ? 1 : 0;
}
extern "C" void klg_fx_destroy(klg_fx* f) { if (!f) return; DeviceGuard bound(f->device); fx_free(f); }
extern "C" size_t klg_fx_state_bytes(const klg_fx* f) {
	if (!f) return 0;
	if (f->multi) return klg_fx_state_bytes(f->multi->shard[0]);
	if (f->graph) return (size_t)f->words * 4 + (size_t)f->graph->ring_rows * 4;
	return (size_t)f->words * 4 + (f->patch == KLG_PATCH_PINGPONG ? (size_t)2 * 192000 * 4 : ((size_t)2 * RV_ESIZE + (size_t)16 * RV_FSIZE) * 4);
}

static int fx_flush_updates(klg_fx* f, hipStream_t st);
extern "C" int klg_fx_set_control(klg_fx* f, int instance, int index, float value) {
	if (!f || instance < 0 || instance >= f->K || index < 0 || index >= f->nctl) return fail(KLG_ERR_INVALID, "klg_fx_set_control: instance %d / control %d out of range", instance, index);
	if (f->multi) { int li = 0; const int sh = fx_shard_of(f, instance, &li); return klg_fx_set_control(f->multi->shard[(size_t)sh], li, index, value); }
	host::ControlH& c = f->controls[(size_t)instance * f->nctl + index];
	c.set(value);                                                                   // Control::set clamps (klang.h:1725-1728)
	if (f->graph) {
		f->h_controls[(size_t)instance * KLG_MAX_CTL + index] = c.value; f->controls_dirty = true;
		// a control the effect itself writes lives in the instance's record: a host set() overwrites the effect's value (klang.h:4211-4212)
		if (f->graph->ctlvar_word[index] >= 0) f->upd.push_back({ instance, f->graph->ctlvar_word[index], f2i(c.value) });
		return 0;
	}
	if (f->patch == KLG_PATCH_PINGPONG) f->upd.push_back({ instance, index, f2i(c.value) });
	else {
		if (index < 5) f->upd.push_back({ instance, RV_CTL + index, f2i(c.value) });
		if (!f->rv_flag[instance]) { f->rv_flag[instance] = 1; f->rv_touched.push_back(instance); }
	}
	return 0;
}
// ---- an instance's record, for effects whose prepare() is HOST code (include/klang/klang.h EffectBank: Controls::changed(), rand() tables,
// loops over a count — examples/Reverb.k:238-241): the facade runs prepare() on its own mirror of the instance, starting from the record as
// the device last left it, and uploads the words prepare() changed ----
extern "C" int klg_fx_record_words(const klg_fx* f) {
	if (!f) return fail(KLG_ERR_INVALID, "klg_fx_record_words: null bank");
	return f->multi ? klg_fx_record_words(f->multi->shard[0]) : f->words;
}
extern "C" int klg_fx_download_record(klg_fx* f, int instance, void* words, size_t bytes) {
	if (!f || !words || instance < 0 || instance >= f->K) return fail(KLG_ERR_INVALID, "klg_fx_download_record: bad arguments");
	if (f->multi) { int li = 0; const int sh = fx_shard_of(f, instance, &li); return klg_fx_download_record(f->multi->shard[(size_t)sh], li, words, bytes); }
	if (bytes != (size_t)f->words * 4) return fail(KLG_ERR_INVALID, "klg_fx_download_record: the record is %d bytes", f->words * 4);
	KLG_BIND(f);
	if (int rc = fx_flush_updates(f, f->stream)) return rc;
	HIP_TRY(hipStreamSynchronize(f->stream));
	HIP_TRY(hipMemcpy2D(words, 4, (const float*)f->d_state + instance, f->kpad * sizeof(float), 4, (size_t)f->words, hipMemcpyDeviceToHost));   // word w of instance k: state[w][k]
	return 0;
}
extern "C" int klg_fx_upload_words(klg_fx* f, int instance, int first, int count, const void* values) {
	if (!f || !values || instance < 0 || instance >= f->K || first < 0 || count < 0) return fail(KLG_ERR_INVALID, "klg_fx_upload_words: bad arguments");
	if (f->multi) { int li = 0; const int sh = fx_shard_of(f, instance, &li); return klg_fx_upload_words(f->multi->shard[(size_t)sh], li, first, count, values); }
	if (first + count > f->words) return fail(KLG_ERR_INVALID, "klg_fx_upload_words: words %d .. %d of a record of %d", first, first + count - 1, f->words);
	const int* v = (const int*)values;
	for (int i = 0; i < count; i++) f->upd.push_back({ instance, first + i, v[i] });       // applied, in order, before the next block (klg_fx_apply_updates)
	return 0;
}
// the control's value as the effect sees it: a control the effect writes itself (PingPong.k:48,60 controls[1].set(..)) is read back from the
// instance's state — the parameter sync OUT of Effect::process(float*, int, float* parameters) klang.h:4213-4215
extern "C" int klg_fx_get_control(klg_fx* f, int instance, int index, float* value) {
	if (!f || !value || instance < 0 || instance >= f->K || index < 0 || index >= f->nctl) return fail(KLG_ERR_INVALID, "klg_fx_get_control: instance %d / control %d out of range", instance, index);
	if (f->multi) { int li = 0; const int sh = fx_shard_of(f, instance, &li); return klg_fx_get_control(f->multi->shard[(size_t)sh], li, index, value); }
	int word = -1;
	if (f->graph) word = f->graph->ctlvar_word[index];
	else if (f->patch == KLG_PATCH_PINGPONG && index == 1) word = 1;                  // klg_fx_pingpong*: state word 1 is controls[1]
	if (word < 0) { *value = f->controls[(size_t)instance * f->nctl + index].value; return 0; }
	KLG_BIND(f);
	if (int rc = fx_flush_updates(f, f->stream)) return rc;
	HIP_TRY(hipStreamSynchronize(f->stream));
	HIP_TRY(hipMemcpy(value, (const float*)f->d_state + (size_t)word * f->kpad + instance, sizeof(float), hipMemcpyDeviceToHost));
	return 0;
}

// ---- Reverb.k prepare() on the host (Reverb.k:237-241 -> Reflections::set 188-214) ----
static float random_f(float mn, float mx) { return rand() * ((mx - mn) / (float)RAND_MAX) + mn; }   // klang::random<float> klang.h:236

static void rv_push_coef(klg_fx* f, int k, int word0, const BiquadCoef& c) {
	const float v[5] = { c.b0, c.b1, c.b2, c.a1, c.a2 };
	for (int i = 0; i < 5; i++) f->upd.push_back({ k, word0 + i, f2i(v[i]) });
}
static void rv_prepare(klg_fx* f, int k) {
	RvHost& h = f->rv[k];
	const host::ControlH* c = &f->controls[(size_t)k * 10];
	bool changed = false;                                                           // Controls::changed() klang.h:1914-1923
	for (int i = 0; i < 10; i++) if (c[i].value != h.cache[i]) { h.cache[i] = c[i].value; changed = true; }
	if (!changed) return;
	klg_random_seed(272839);                                                        // random(272839)  Reverb.k:239 (whatever stream a device held is superseded: klg_api.hip RngChain)
	const float length = c[5].value, size = c[6].value;
	float dampening1 = c[7].value, dampening2 = c[8].value;
	const host::Fs& fs = f->fs;
	// early.set((length / 10.f) * 1000.f + 50.f, size)  Reverb.k:189, 63-73
	{
		float el = (length / 10.f) * 1000.f + 50.f;
		el *= 1 / 1000.f;
		if (h.e_length != el || h.e_size != size) {
			h.e_length = el; h.e_size = size;
			static const float primes[20] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71 };
			const int count = 10 + (int)(size * (float)10.999);                        // Reverb.k:26
			const float scale = 50.f / primes[count - 1];
			const float ms = fs.f / 1000.f;
			f->upd.push_back({ k, RV_ECOUNT, count });
			for (int r = 0; r < count; r++) {
				const float t = ((50.f + primes[r] * scale) * ms * random_f(0.9f, 1.1f));
				const float x = (float)(r + 1.f) / (float)(unsigned)count;
				const float g = random_f(0.5f, 1.5f) * expf(-3.f * x);
				const float pan = random_f(0.f, 1.f);
				f->upd.push_back({ k, RV_ETIMES + r, f2i(t) });
				f->upd.push_back({ k, RV_EGL + r, f2i(g * (1.f - pan)) });
				f->upd.push_back({ k, RV_EGR + r, f2i(g * pan) });
			}
			rv_push_coef(f, k, RV_EHPF, design_biquad(true, 100.f, host::ROOT2_INV, fs));     // hpf.set(100)
			rv_push_coef(f, k, RV_ELPF, design_biquad(false, 15000.f, host::ROOT2_INV, fs));  // lpf.set(15000)
		}
	}
	dampening1 *= 10000.f;
	dampening2 *= dampening1;
	const int fpos = (int)((2ull * f->samples) % RV_FSIZE);                          // every FilteredDelay line has had 2 inputs per sample
	const float delays1[4] = { 7, 11, 13, 17 }, delays2[4] = { 19, 23, 29, 31 };
	for (int a = 0; a < 4; a++) {                                                   // mid[0], mid[1], late[0], late[1]  Reverb.k:198-207
		const float* delays = a < 2 ? delays1 : delays2;
		const float damp = a < 2 ? dampening1 : dampening2;
		const float gain = a < 2 ? 0.25f : 0.35f;
		for (int j = 0; j < 4; j++) {                                               // FilteredDelay::set Reverb.k:123-127
			const float time = delays[j] * random_f(.9f, 1.1f);
			const float samples = time * fs.f / 1000.f;
			const float t = samples < RV_FSIZE ? samples : (float)RV_FSIZE;            // Delay::set klang.h:3480-3489
			float read = (float)(fpos - 1) - t;
			if (read < 0.f) read += RV_FSIZE;
			const int lastp = (int)read;
			const float lastf = read - lastp;
			const int w0 = RV_FD + (a * 4 + j) * FD_WORDS;
			f->upd.push_back({ k, w0 + FD_LASTP, lastp });
			f->upd.push_back({ k, w0 + FD_LASTF, f2i(lastf) });
			f->upd.push_back({ k, w0 + FD_GAIN, f2i(gain) });
			rv_push_coef(f, k, w0 + FD_COEF, design_biquad(false, damp, host::ROOT2_INV, fs));   // filter.set(cutoff)
		}
	}
}

static int fx_flush_updates(klg_fx* f, hipStream_t st) {
	f->upd_flushed = !f->upd.empty();
	if (f->upd.empty()) return 0;
	{	// several updates of one word in a batch: the LAST one wins (the scatter kernel has no ordering)
		std::stable_sort(f->upd.begin(), f->upd.end(), [](const FxUpdate& a, const FxUpdate& b) { return a.k != b.k ? a.k < b.k : a.word < b.word; });
		size_t o = 0;
		for (size_t i = 0; i < f->upd.size(); i++) {
			if (i + 1 < f->upd.size() && f->upd[i + 1].k == f->upd[i].k && f->upd[i + 1].word == f->upd[i].word) continue;
			f->upd[o++] = f->upd[i];
		}
		f->upd.resize(o);
	}
	const size_t bytes = f->upd.size() * sizeof(FxUpdate);
	if (bytes > f->d_upd_cap) {
		HIP_TRY(hipStreamSynchronize(st));
		if (f->d_upd) (void)hipFree(f->d_upd);
		f->d_upd_cap = bytes * 2;
		HIP_TRY(hipMalloc(&f->d_upd, f->d_upd_cap));
	}
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipMemcpyAsync(f->d_upd, f->upd.data(), bytes, hipMemcpyHostToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));
	const int count = (int)f->upd.size();
	hipLaunchKernelGGL(klg_fx_apply_updates, dim3((count + 255) / 256), dim3(256), 0, st, f->d_state, f->kpad, (const int*)f->d_upd, count);
	HIP_TRY(hipGetLastError());
	f->upd.clear();
	return 0;
}

static int fx_enqueue_graph(klg_fx* f, float* d_io, int n, hipStream_t st, int blocks = 1) {
	if (int rc = fx_flush_updates(f, st)) return rc;
	if (f->controls_dirty) {
		HIP_TRY(hipStreamSynchronize(st));
		HIP_TRY(hipMemcpyAsync(f->d_controls, f->h_controls.data(), f->h_controls.size() * 4, hipMemcpyHostToDevice, st));
		HIP_TRY(hipStreamSynchronize(st));
		f->controls_dirty = false;
	}
	FxGraphArgs a;
	a.state = (uint32_t*)f->d_state; a.kpad = f->kpad; a.K = f->K; a.rings = f->d_rings; a.ring_rows = (size_t)f->graph->ring_rows;
	a.io = d_io; a.n = n; a.controls = f->d_controls;
	a.fs.f = f->fs.f; a.fs.w = f->fs.w; a.fs.timeInc = 1.0f / f->fs.f;
	a.samples = f->samples;
	a.rand = nullptr; a.rstride = 0;
	a.blocks = blocks; a.block_stride = (size_t)f->K * (size_t)f->channels * (size_t)n;      // (blocks > 1: fx_span_in_one_launch() said so)
	const int draws = f->graph->noise_calls;
	if (draws > 0) {
		// Generators::*::Noise call libc rand() once per sample (klang.h:4949, 5363), and the reference's `K` effect objects of one process would draw in the order
		// they are processed — instance 0's whole block, then instance 1's, ... block after block: (block, instance) is the rank, n * draws values each, produced on
		// the device from the stream's state (klg_rand_fill): no host draw, no copy.
		const size_t ranks = (size_t)f->K * (size_t)blocks, rstride = (ranks + 63) / 64 * 64, need = rstride * (size_t)n * (size_t)draws;
		if (need > f->rand_cap) {
			RandGuard rg;                                                  // (allocations must not disturb the C library's generator)
			HIP_TRY(hipStreamSynchronize(st));
			if (f->rand_done) HIP_TRY(hipEventSynchronize(f->rand_done)); else HIP_TRY(hipEventCreateWithFlags(&f->rand_done, hipEventDisableTiming));
			if (f->d_rand) (void)hipFree(f->d_rand);
			f->d_rand = nullptr; f->rand_cap = 0;
			if (hipMalloc((void**)&f->d_rand, need * sizeof(int)) != hipSuccess) return fail(KLG_ERR_NOMEM, "the Noise generators' draws (%zu instances x blocks, %d samples, %d generators: %.2f GB) could not be allocated", ranks, n, draws, need * 4 / 1e9);
			f->rand_cap = need;
		}
		else if (f->rand_stream != st) HIP_TRY(hipStreamWaitEvent(st, f->rand_done, 0));   // (the buffer's last reader ran on another stream)
		if (int rc = rng_fill(f->device, st, f->d_rand, rstride, nullptr, (unsigned)ranks, (unsigned)ranks, n * draws)) return rc;
		a.rand = f->d_rand; a.rstride = rstride;
	}
	void* params[] = { &a };
	// G instances x C samples per workgroup, level by level (klg_graph_staged.hpp) — or, for a body that has no such form (Compiled::staged_why), one lane per
	// instance walking the samples in order.  Same bits either way (tests/test_gpu_fx_facade.py runs both).
	if (f->staged_fn) {
		TimedLaunch timed(f);
		const hipError_t e = klg_module_launch(f->staged_fn, (unsigned)(f->kpad / (size_t)f->graph->staged_G), (unsigned)f->graph->staged_threads, (unsigned)f->graph->staged_lds, st, params);
		if (e != hipSuccess && blocks <= 1) {                              // (the launch was refused — its LDS, its registers —: nothing ran; the other kernel of the code object renders the same bits)
			(void)hipGetLastError();
			fprintf(stderr, "klang-mi355: launching the staged form failed (%s): one lane per instance from here on\n", hipGetErrorString(e));
			f->staged_fn = nullptr;
			HIP_TRY(klg_module_launch(f->graph_fn, (unsigned)(f->kpad / FX_WG), FX_WG, 0, st, params));
		}
		else HIP_TRY(e);
	}
	else { TimedLaunch timed(f); HIP_TRY(klg_module_launch(f->graph_fn, (unsigned)(f->kpad / FX_WG), FX_WG, 0, st, params)); }
	if (draws > 0) { HIP_TRY(hipEventRecord(f->rand_done, st)); f->rand_stream = st; }
	f->samples += (unsigned long long)n * (unsigned long long)blocks;
	return 0;
}
// Can a span of blocks of n samples go out as ONE launch?  The staged form of a recorded effect walks the blocks itself (prepare() at the head of each; a
// span's Noise draws are made by one klg_rand_fill ahead of it).  PingPong's pipelined kernel takes the span as one long block (PingPong.k's prepare() only sets the DC filters
// and no dial moves inside a span): blocks of whole chunks.  Reverb's kernel stages a whole block in LDS: it stays a launch (two) per block.
static bool fx_span_in_one_launch(const klg_fx* f, int n) {
	if (f->graph) return f->staged_fn != nullptr;
	if (f->patch != KLG_PATCH_PINGPONG) return false;
	const char* e1 = getenv("KLG_FX_PINGPONG1"); const char* e2 = getenv("KLG_FX_ABLATE");
	return n % PPX_CHUNK == 0 && !(e1 && e1[0] == '1') && !(e2 && atoi(e2));
}

// the early-sum buffers, second stream and events of KLG_FX_REVERB_EARLY modes 1 / 2: allocated the first time such a mode is selected (banks above 2,048
// instances default to mode 0 and never pay the 2 x kpad x 2 x max_block floats: 256 MB at 16,384 instances)
static bool rv_early_alloc(klg_fx* f) {
	if (f->d_early[0] && f->d_early[1] && f->early_stream && f->early_ready && f->upd_done && f->q_done[0] && f->q_done[1]) return true;
	if (f->early_alloc_failed) return false;
	RandGuard rg;
	bool ok = true;
	for (int i = 0; i < 2; i++) {
		if (!f->d_early[i]) ok = ok && hipMalloc(&f->d_early[i], (size_t)f->kpad * 2 * f->max_block * 4) == hipSuccess;
		if (!f->q_done[i]) ok = ok && hipEventCreateWithFlags(&f->q_done[i], hipEventDisableTiming) == hipSuccess;
	}
	if (!f->early_stream) ok = ok && hipStreamCreateWithFlags(&f->early_stream, hipStreamNonBlocking) == hipSuccess;
	if (!f->early_ready) ok = ok && hipEventCreateWithFlags(&f->early_ready, hipEventDisableTiming) == hipSuccess;
	if (!f->upd_done) ok = ok && hipEventCreateWithFlags(&f->upd_done, hipEventDisableTiming) == hipSuccess;
	if (!ok) { (void)hipGetLastError(); f->early_alloc_failed = true; }
	return ok;
}

static int fx_enqueue(klg_fx* f, float* d_io, int n, hipStream_t st, int blocks = 1) {
	if (f->graph) return fx_enqueue_graph(f, d_io, n, st, blocks);
	if (f->patch == KLG_PATCH_REVERB && !f->rv_touched.empty()) {                    // prepare(): `if (controls.changed())` Reverb.k:238 — a dial changes only through klg_fx_set_control
		std::sort(f->rv_touched.begin(), f->rv_touched.end());                        // (instance order: every changed instance re-seeds and draws from rand())
		for (int k : f->rv_touched) { rv_prepare(f, k); f->rv_flag[k] = 0; }
		f->rv_touched.clear();
	}
	if (!f->upd.empty()) f->pp_touched_at = f->samples;                              // a dial or a record word was written since the last block (PingPong: which kernel a span takes, below)
	if (int rc = fx_flush_updates(f, st)) return rc;
	// kernel timing: PingPong is one launch, timed by events attached to its dispatch; a Reverb block is the early-sum kernel and klg_fx_reverb_q
	// (the pair is what a block costs): events recorded around both on the stream
	const bool bracket = f->timing && f->patch != KLG_PATCH_PINGPONG;
	if (bracket) {
		if ((int)f->tev.size() < 2 * (f->launches + 1)) { hipEvent_t e0, e1; HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); f->tev.push_back(e0); f->tev.push_back(e1); }
		HIP_TRY(hipEventRecord(f->tev[2 * f->launches], st));
	}
	const dim3 grid((unsigned)(f->kpad / FX_WG)), block(FX_WG);
	if (f->patch == KLG_PATCH_PINGPONG) {
		PingPongArgs a;
		a.pass = 0; a.done = nullptr;
		a.state = f->d_state; a.kpad = f->kpad; a.K = f->K; a.rings = f->d_rings;
		a.position = (int)(f->samples % 192000ull);
		a.io = d_io; a.n = n * blocks; a.nb = blocks > 1 ? n : 0; a.block_stride = (size_t)f->K * 2 * (size_t)n;
		a.fs.f = f->fs.f; a.fs.w = f->fs.w; a.fs.timeInc = 1.0f / f->fs.f;
		a.dc = f->pp_dc; a.c1_min = PP_DIALS[1].min; a.c1_max = PP_DIALS[1].max;
		static const int ablate = []() { const char* e = getenv("KLG_FX_ABLATE"); return e ? atoi(e) : 0; }();
		a.ablate = ablate;
		static const bool single_wave = []() { const char* e = getenv("KLG_FX_PINGPONG1"); return e && e[0] == '1'; }();
		if (single_wave || a.ablate) {                                                  // one wave per 64 instances (A/B reference, ablation)
#ifdef KLG_AB_KERNELS
			TimedLaunch timed(f); KLG_LAUNCH(klg_fx_pingpong, grid, block, 0, st, a);
#else
			return fail(KLG_ERR_INVALID, "KLG_FX_PINGPONG1 / KLG_FX_ABLATE ask for the one-wave A/B reference kernel; this library was built without -DKLG_AB_KERNELS");
#endif
		}
		// the pipeline's time is a workgroup's instruction count on its one CU: a quarter / a half of a ring group per workgroup while
		// the bank does not fill the chip that way either (klg_fx_pingpong_x<G>; KLG_FX_PINGPONG_G = 16 / 32 / 64 forces the width)
		else {
			const char* const forced_env = getenv("KLG_FX_PINGPONG_G");                 // (read per launch: the tests switch it)
			const int forced = forced_env ? atoi(forced_env) : 0;
			const int G = forced == 16 || forced == 32 || forced == 64 ? forced : (f->kpad <= 4096 ? 16 : f->kpad <= 8192 ? 32 : 64);
			// Which compilation of the kernel (klg_fx.hpp: MODE).  The moving-dials pipeline only ever runs for spans of PPX_MOVING_MIN chunks and more: shorter launches
			// take the kernel compiled without it (the same paths otherwise).  A longer span goes out TWICE unless a dial was touched within the last 8,192 samples
			// (then it surely moves): first the stationary-only kernel (no scratch, 11 waves' worth of code), in which every workgroup whose span is stationary — decided
			// on the device, on the dials as they are — renders it and says so in pp_done; then the full kernel for the workgroups that were not (the others leave at
			// once).  KLG_FX_PINGPONG_MV = 1 / 0 forces the full kernel alone / the kernel without the moving-dials pipeline alone (A/B; the same bits every way).
			static const int force_mv = []() { const char* e = getenv("KLG_FX_PINGPONG_MV"); return e ? atoi(e) : -1; }();
			const int chunks = (n * blocks + PPX_CHUNK - 1) / PPX_CHUNK;
			const bool long_span = chunks >= PPX_MOVING_MIN;
			const bool touched = f->samples - f->pp_touched_at < 8192ull;
			// (two launches cost a span ~2.5 us: below 64 chunks — eight 256-sample blocks — a span whose dials have been left alone takes the kernel without the moving-dials pipeline alone)
			int plan = !long_span ? 0 : touched ? 1 : chunks >= 2 * PPX_MOVING_MIN ? 2 : 0;   // 0: the kernel without the moving-dials pipeline alone, 1: the full kernel alone, 2: both
			if (force_mv == 1) plan = 1; else if (force_mv == 0) plan = 0;
			if (plan == 2 && !f->pp_done) { if (hipMalloc((void**)&f->pp_done, (f->kpad / 16) * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); f->pp_done = nullptr; plan = 1; } }
			auto launch = [&](auto mode_c, int pass) {
				constexpr int MV = decltype(mode_c)::value;
				a.pass = pass; a.done = f->pp_done;
				TimedLaunch timed2(f);                                                  // (each launch its own pair of events: a two-launch span's time is their sum)
				if (G == 16) KLG_LAUNCH((klg_fx_pingpong_x<16, MV>), dim3((unsigned)(f->kpad / 16)), dim3(PPX_THREADS), 0, st, a);
				else if (G == 32) KLG_LAUNCH((klg_fx_pingpong_x<32, MV>), dim3((unsigned)(f->kpad / 32)), dim3(PPX_THREADS), 0, st, a);
				else KLG_LAUNCH((klg_fx_pingpong_x<64, MV>), grid, dim3(PPX_THREADS), 0, st, a);   // control / audio / filter pipeline over twelve waves
			};
			if (plan == 0) launch(IntTag<PPX_NO_MOVING>{}, 0);
			else if (plan == 1) launch(IntTag<PPX_FULL>{}, 0);
			else { launch(IntTag<PPX_STATIONARY>{}, 1); launch(IntTag<PPX_FULL>{}, 2); }
		}
	}
	else {
		ReverbArgs a;
		a.state = f->d_state; a.kpad = f->kpad; a.K = f->K; a.early_rings = f->d_rings; a.fd_rings = f->d_rings2;
		a.epos = (int)(f->samples % (unsigned long long)RV_ESIZE);
		a.fpos = (int)((2ull * f->samples) % (unsigned long long)RV_FSIZE);
		a.io = d_io; a.n = n;
		static const bool single_wave = []() { const char* e = getenv("KLG_FX_REVERB1"); return e && e[0] == '1'; }();
		a.layout = f->rv_layout; a.early_sums = nullptr;
		// the production kernels request ring rows ahead of their use (reverb_q: 8 samples = 16 positions; reverb16: one sample): safe while the
		// shortest line (7 ms * 0.9) is longer than that.  reverb_q also computes a block's early sums before the block's early-line writes:
		// right while every tap reads further back than the block is long — the shortest tap is (50 ms + ...) * random(0.9, 1.1) > 44.9 ms.
		const bool taps_behind_block = (float)(n + 2) < 0.0449f * f->fs.f;
		// klg_fx_reverb_q stages the block in dynamic LDS (75 KB at n = 1024): more than a device grants (not on gfx950's 160 KB) -> the single-lane kernel, loudly once
		const size_t q_lds = (size_t)(RVQ_WG / 64) * (RVQ_TILE_ROWS * (((n + 3) & ~3) + 4) + RVQ_XQ_FLOATS) * sizeof(float);
		const bool lds_fits = q_lds <= (size_t)f->lds_limit;
		if (f->rv_layout && !lds_fits) { static bool told = false; if (!told) { told = true; fprintf(stderr, "klang-mi355: klg_fx_reverb_q needs %zu bytes of LDS per workgroup for %d-sample blocks, the device grants %d: using the single-lane kernel\n", q_lds, n, f->lds_limit); } }
		if (single_wave || f->fs.f < 16000.f || (f->rv_layout && (!taps_behind_block || !lds_fits))) hipLaunchKernelGGL(klg_fx_reverb, grid, block, 0, st, a);   // one lane walks the whole graph (A/B reference; either layout)
		else if (f->rv_layout) {
			const dim3 qgrid((unsigned)((f->kpad + 4 * (RVQ_WG / 64) - 1) / (4 * (RVQ_WG / 64))));
			const size_t qlds = (size_t)(RVQ_WG / 64) * (RVQ_TILE_ROWS * (((n + 3) & ~3) + 4) + RVQ_XQ_FLOATS) * sizeof(float);
			// KLG_FX_REVERB_EARLY: 0 = the early sums inside klg_fx_reverb_q (phase 1 of the lone wave), 1 = klg_fx_reverb_early ahead of it on the same
			// stream, 2 = on a second stream, beside the previous block's recursive kernel.  Measured (tools/reverb_modes.py, profiles/r03_reverb_modes.jsonl,
			// us per 256-sample block, modes 0 / 1 / 2): 1,024 instances 82 / 71 / 78, 4,096: 138 / 145 / 151, 8,192: 257 / 284 / 286.  The sums are not
			// instruction work that a fuller chip absorbs: they are 168 MB of the block's 302 MB of ring READS at 4,096 instances (20 taps x 2 lines x 257
			// positions per instance), and from 4,096 instances on the recursive kernel already keeps the memory system busy — a second kernel only adds
			// its own launch and an 8 MB round trip of the sums.  Below that the lone wave's phase 1 is latency-bound and the separate launch wins.
			static const int forced_mode = []() { const char* e = getenv("KLG_FX_REVERB_EARLY"); return e ? atoi(e) : -1; }();
			int early_mode = forced_mode >= 0 ? forced_mode : (f->kpad <= 2048 ? 1 : 0);
			if (early_mode != 0 && (f->K > 65535 || !rv_early_alloc(f))) early_mode = 0;   // (klg_fx_reverb_early's grid has one row of workgroups per instance: gridDim.y <= 65535)
			// Mode 2 runs THIS block's sums beside the PREVIOUS block's recursive kernel, which is still writing the previous block's early-line samples: nothing
			// the sums read may be that young — the previous block's length counts too — and the previous block must have gone to the same stream (the only
			// ordering mode 2 keeps is q_done of two blocks ago).  Otherwise: the same kernels on one stream (mode 1).
			if (early_mode == 2 && (f->rv_prev_stream != st || !((float)(f->rv_prev_n + n + 2) < 0.0449f * f->fs.f))) early_mode = 1;
			f->rv_prev_n = n; f->rv_prev_stream = st;
			if (early_mode == 0) hipLaunchKernelGGL(klg_fx_reverb_q<false>, qgrid, dim3(RVQ_WG), qlds, st, a);   // one wave per four instances
			else {
				const unsigned turn = f->early_turn++ & 1u;
				a.early_sums = f->d_early[turn];
				const dim3 egrid((unsigned)((n + RVE_WG - 1) / RVE_WG), (unsigned)f->K);
				if (early_mode == 1) hipLaunchKernelGGL(klg_fx_reverb_early, egrid, dim3(RVE_WG), 0, st, a);
				else {
					// The sums of this block read early-line samples >= 45 ms old (the host checked taps_behind_block) and the tap words.  They wait (a) for
					// state updates flushed on `st` in THIS call (a dial moved: rare), (b) for the recursive kernel that last read this buffer (two blocks
					// ago) — and for nothing else on `st`: in a stream of blocks they run beside the previous block's recursive kernel.
					if (f->upd_flushed) { HIP_TRY(hipEventRecord(f->upd_done, st)); HIP_TRY(hipStreamWaitEvent(f->early_stream, f->upd_done, 0)); }
					if (f->q_used[turn]) HIP_TRY(hipStreamWaitEvent(f->early_stream, f->q_done[turn], 0));
					hipLaunchKernelGGL(klg_fx_reverb_early, egrid, dim3(RVE_WG), 0, f->early_stream, a);
					HIP_TRY(hipEventRecord(f->early_ready, f->early_stream));
					HIP_TRY(hipStreamWaitEvent(st, f->early_ready, 0));
				}
				hipLaunchKernelGGL(klg_fx_reverb_q<true>, qgrid, dim3(RVQ_WG), qlds, st, a);
				HIP_TRY(hipEventRecord(f->q_done[turn], st)); f->q_used[turn] = true;       // (recorded in mode 1 as well: a later mode-2 block's sums wait for the last reader of this buffer whichever mode it ran in)
			}
		}
		else return fail(KLG_ERR_INVALID, "klg_fx: no Reverb kernel for ring layout %d", f->rv_layout);
	}
	HIP_TRY(hipGetLastError());
	if (bracket) { HIP_TRY(hipEventRecord(f->tev[2 * f->launches + 1], st)); f->launches++; }
	f->samples += (unsigned long long)n * (unsigned long long)blocks;
	return 0;
}

extern "C" int klg_fx_process(klg_fx* f, float* io, int n) {
	if (!f || !io || n <= 0 || n > f->max_block) return fail(KLG_ERR_INVALID, "klg_fx_process: bad arguments (n=%d)", n);
	if (f->multi) {
		// every shard takes ITS instances' rows of the caller's [K][channels][n] block: enqueue on all devices, then wait for all (no collective:
		// effect instances are independent).  Host work that is ordered across instances (Reverb's prepare(), Noise draws) runs shard by shard
		// = in instance order, as in one bank.
		const FxMulti& m = *f->multi;
		for (size_t i = 0; i < m.shard.size(); i++) {
			klg_fx* sh = m.shard[i]; DeviceGuard bound(sh->device); if (!bound.ok) return KLG_ERR_NO_DEVICE;
			float* part = io + (size_t)m.first[i] * sh->channels * n; const size_t bytes = (size_t)sh->K * sh->channels * n * 4;
			HIP_TRY(hipMemcpyAsync(sh->d_io, part, bytes, hipMemcpyHostToDevice, sh->stream));
			if (int rc = fx_enqueue(sh, sh->d_io, n, sh->stream)) return rc;
			HIP_TRY(hipMemcpyAsync(part, sh->d_io, bytes, hipMemcpyDeviceToHost, sh->stream));
		}
		for (klg_fx* sh : m.shard) { DeviceGuard bound(sh->device); HIP_TRY(hipStreamSynchronize(sh->stream)); }
		return 0;
	}
	KLG_BIND(f);
	const size_t bytes = (size_t)f->K * f->channels * n * 4;
	HIP_TRY(hipMemcpyAsync(f->d_io, io, bytes, hipMemcpyHostToDevice, f->stream));
	if (int rc = fx_enqueue(f, f->d_io, n, f->stream)) return rc;
	HIP_TRY(hipMemcpyAsync(io, f->d_io, bytes, hipMemcpyDeviceToHost, f->stream));
	HIP_TRY(hipStreamSynchronize(f->stream));
	return 0;
}
extern "C" int klg_fx_process_device(klg_fx* f, float* d_io, int n, void* hip_stream) {
	if (!f || !d_io || n <= 0 || n > f->max_block) return fail(KLG_ERR_INVALID, "klg_fx_process_device: bad arguments (n=%d)", n);
	if (f->multi) return fail(KLG_ERR_INVALID, "klg_fx_process_device: this bank is sharded over %zu devices (klg_init) and a device block lives on ONE of them: use klg_fx_process (host block), or one bank per GPU", f->multi->shard.size());
	KLG_BIND(f);
	return fx_enqueue(f, d_io, n, hip_stream ? (hipStream_t)hip_stream : f->stream);
}
// replaces: the host's block loop around an effect for a stream known in advance (templates/juce/effect/Source/PluginProcessor.cpp:153-178 called `blocks`
// times: offline rendering, a benchmark): see include/klang_mi355.h
extern "C" int klg_fx_render_device(klg_fx* f, float* d_io, int blocks, int n, void* hip_stream) {
	if (!f || !d_io || blocks <= 0 || n <= 0 || n > f->max_block) return fail(KLG_ERR_INVALID, "klg_fx_render_device: bad arguments (blocks=%d, n=%d)", blocks, n);
	if (f->multi) return fail(KLG_ERR_INVALID, "klg_fx_render_device: this bank is sharded over %zu devices (klg_init) and a device buffer lives on ONE of them: one bank per GPU", f->multi->shard.size());
	KLG_BIND(f);
	hipStream_t st = hip_stream ? (hipStream_t)hip_stream : f->stream;
	const size_t stride = (size_t)f->K * (size_t)f->channels * (size_t)n;
	if (blocks > 1 && fx_span_in_one_launch(f, n)) {
		// (the cursor arithmetic of a launch is 32-bit from the start of its span: spans of at most 2^20 samples per launch)
		int per = std::max(1, (1 << 20) / n);
		if (f->graph && f->graph->noise_calls > 0) per = (int)std::max<size_t>(1, std::min<size_t>((size_t)per, ((size_t)64 << 20) / ((size_t)n * (size_t)f->graph->noise_calls * (size_t)f->K)));   // (... and their draws in at most 256 MB)
		for (int b = 0; b < blocks; b += per) if (int rc = fx_enqueue(f, d_io + (size_t)b * stride, n, st, std::min(per, blocks - b))) return rc;
		return 0;
	}
	for (int b = 0; b < blocks; b++) if (int rc = fx_enqueue(f, d_io + (size_t)b * stride, n, st)) return rc;
	return 0;
}
extern "C" int klg_fx_sync(klg_fx* f) {
	if (!f) return fail(KLG_ERR_INVALID, "klg_fx_sync: NULL handle");
	if (f->multi) { for (klg_fx* sh : f->multi->shard) if (int rc = klg_fx_sync(sh)) return rc; return 0; }
	KLG_BIND(f);
	HIP_TRY(hipDeviceSynchronize());
	return 0;
}
extern "C" int klg_fx_timing_begin(klg_fx* f) { if (!f) return fail(KLG_ERR_INVALID, "NULL handle"); if (f->multi) { for (klg_fx* sh : f->multi->shard) klg_fx_timing_begin(sh); return 0; } f->timing = true; f->launches = 0; return 0; }
extern "C" int klg_fx_timing_end(klg_fx* f, int* launches, float* total_ms) {
	if (!f || !launches || !total_ms) return fail(KLG_ERR_INVALID, "klg_fx_timing_end: bad arguments");
	if (f->multi) {                                                     // the slowest shard (they run concurrently)
		int l = 0; float ms = 0.f; *launches = 0; *total_ms = 0.f;
		for (klg_fx* sh : f->multi->shard) { if (int rc = klg_fx_timing_end(sh, &l, &ms)) return rc; if (ms > *total_ms) { *total_ms = ms; *launches = l; } }
		return 0;
	}
	KLG_BIND(f);
	HIP_TRY(hipDeviceSynchronize());
	float total = 0.f;
	for (int i = 0; i < f->launches; i++) { float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, f->tev[2 * i], f->tev[2 * i + 1])); total += ms; }
	*launches = f->launches; *total_ms = total;
	f->timing = false;
	return 0;
}
