// klang_amd/csrc/klg_api.hip — libklang_mi355.so: the C-ABI of include/klang_mi355.h.
//
// Host side (this file): voice allocation (Notes::assign), note lifecycle (NoteBase::start/release/stop),
// controls, the patches' on()/off() code and event dispatch — all on the CPU, as in the reference.
// Device side (klg_kernels.hpp): everything per sample.  There is no CPU rendering path.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/klang_mi355.h"
#include "../../include/klang/host_dsl.hpp"
#include "klg_kernels.hpp"
#include "klg_rand_dev.hpp"
#include "klg_graph.hpp"
#include "klg_fx.hpp"
#include "klg_render_x2.hpp"
#include "klg_render_lanes.hpp"
#include "klg_render_supersaw_sp.hpp"
#include "klg_render_sub2a_sp.hpp"

// Kernel timing (klg_timing_* / klg_fx_timing_*): the dominant kernel of a call is launched with its two events ATTACHED TO THE DISPATCH
// (hipExtLaunchKernelGGL / hipExtModuleLaunchKernel): they hold the kernel's own start and end, what rocprofv3's kernel trace reads.  Events recorded
// around the launch on the stream add the dispatch latency and the marker's completion — 2 - 3 us, which is 15 % of a 17 us PingPong block and made
// a 1,024-voice bank's "kernel time" longer than its whole step.  Outside timing the two are null and the launch is an ordinary one.
static thread_local hipEvent_t g_time0 = nullptr, g_time1 = nullptr;
#define KLG_LAUNCH(kernel, grid, block, lds, st, ...) hipExtLaunchKernelGGL(kernel, grid, block, lds, st, g_time0, g_time1, 0, __VA_ARGS__)
static hipError_t klg_module_launch(hipFunction_t fn, unsigned groups, unsigned threads, unsigned lds, hipStream_t st, void** params) {
	if (!g_time0) return hipModuleLaunchKernel(fn, groups, 1, 1, threads, 1, 1, lds, st, params, nullptr);
	return hipExtModuleLaunchKernel(fn, groups * threads, 1, 1, threads, 1, 1, lds, st, params, nullptr, g_time0, g_time1, 0);
}
struct TimedLaunch {                                                // binds a handle's next pair of events to the launches made in its scope
	template<class H> explicit TimedLaunch(H* h) {
		if (!h->timing) return;
		if ((int)h->tev.size() < 2 * (h->launches + 1)) { hipEvent_t e0 = nullptr, e1 = nullptr; if (hipEventCreate(&e0) != hipSuccess) return; if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return; } h->tev.push_back(e0); h->tev.push_back(e1); }
		g_time0 = h->tev[2 * h->launches]; g_time1 = h->tev[2 * h->launches + 1]; h->launches++;
	}
	~TimedLaunch() { g_time0 = g_time1 = nullptr; }
};
struct TimedAux {                                                   // the same for a block's OTHER launches (its event kernel, the voice-mix reduce): a second list of events
	template<class H> explicit TimedAux(H* h) {
		if (!h->timing) return;
		if ((int)h->tev_aux.size() < 2 * (h->launches_aux + 1)) { hipEvent_t e0 = nullptr, e1 = nullptr; if (hipEventCreate(&e0) != hipSuccess) return; if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return; } h->tev_aux.push_back(e0); h->tev_aux.push_back(e1); }
		g_time0 = h->tev_aux[2 * h->launches_aux]; g_time1 = h->tev_aux[2 * h->launches_aux + 1]; h->launches_aux++;
	}
	~TimedAux() { g_time0 = g_time1 = nullptr; }
};

#pragma clang fp contract(off)

using namespace klg;

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int g_device = -1;                      // the DEFAULT GPU of this process (klg_init's first id, else 0).  A bank remembers the GPU it was created
                                               // on (klg_synth::device / klg_fx::device); nothing below ever swaps this value around a call.

static int fail(int code, const char* fmt, ...) {
	char buf[512];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
	g_err = buf;
	return code;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(KLG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

extern "C" const char* klg_last_error(void) { return g_err.c_str(); }
extern "C" int klg_version(void) { return 100; }

// HIP/HSA runtime initialisation draws from libc's random() state (observed: the first hipMalloc after srand(seed)
// shifts the rand() sequence).  The reference's patches use that same global stream on the host (klang::random /
// SuperSaw.k:17), so every entry point that may initialise the runtime or allocate runs under this guard: it parks
// the caller's generator state and restores it on exit (rand() and random() share state in glibc).
// (The generator state is process-wide: guards of two host threads must not interleave — the second would park the FIRST guard's
// temporary buffer as "the caller's state" and put it back after that buffer is gone.  One guard at a time, re-entrant per thread.)
struct RandGuard {
	char buf[128]; char* prev;
	static std::recursive_mutex& mu() { static std::recursive_mutex m; return m; }
	RandGuard() { mu().lock(); prev = initstate(1u, buf, sizeof buf); }
	~RandGuard() { if (prev) setstate(prev); mu().unlock(); }
	RandGuard(const RandGuard&) = delete; RandGuard& operator=(const RandGuard&) = delete;
};

// the default device, resolved once (-1 + error: no GPU, and no CPU fallback)
static int default_device() {
	if (g_device >= 0) return g_device;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { fail(KLG_ERR_NO_DEVICE, "no HIP device visible: libklang_mi355 has no CPU fallback"); return -1; }
	g_device = 0;
	return 0;
}
// Scoped device binding: every entry point that touches a bank makes the BANK's GPU current for the calling thread (hipSetDevice is
// per thread) and puts back what the thread had on the way out — two banks on two GPUs may be driven from two host threads, and a
// host that shares the process (torch, another library) finds its current device untouched.
struct DeviceGuard {
	int prev = -1; bool ok = true;
	explicit DeviceGuard(int dev) {
		if (hipGetDevice(&prev) != hipSuccess) prev = -1;
		if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; else prev = -1;
		if (!ok) fail(KLG_ERR_NO_DEVICE, "hipSetDevice(%d) failed", dev);
	}
	~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
	DeviceGuard(const DeviceGuard&) = delete; DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define KLG_BIND(handle) DeviceGuard klg_bound_((handle)->device); if (!klg_bound_.ok) return KLG_ERR_NO_DEVICE
// entry points without a handle (klg_selftest): the default device becomes (and stays) current
int klg_ensure_device() {
	if (default_device() < 0) return KLG_ERR_NO_DEVICE;
	if (hipSetDevice(g_device) != hipSuccess) return fail(KLG_ERR_NO_DEVICE, "hipSetDevice(%d) failed", g_device);
	return 0;
}

// The devices of this process.  One id: every bank lives on that GPU (one process per GPU: what bench.py and torch.distributed hosts do).
// Several ids: a synth bank created afterwards is SHARDED over them inside the library — contiguous ranges of synth instances, one
// shard (state, stream, event queue) per device, and ONE RCCL all-reduce of the [2][n] stereo block per klg_process (§ multi-device banks
// below).  The same id may be listed twice (two shards on one GPU: how the sharding logic is tested on a 1-GPU box; those shards are
// combined by a device-side add instead of RCCL, which cannot put one GPU in a communicator twice).
static std::vector<int> g_devices;
extern "C" int klg_init(const int* device_ids, int n_devices) {
	RandGuard rg;
	if (!device_ids || n_devices < 1 || n_devices > 64) return fail(KLG_ERR_INVALID, "klg_init: 1..64 devices (got %d)", n_devices);
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(KLG_ERR_NO_DEVICE, "no HIP device visible: libklang_mi355 has no CPU fallback");
	for (int i = 0; i < n_devices; i++) if (device_ids[i] < 0 || device_ids[i] >= count) return fail(KLG_ERR_INVALID, "device id %d out of range (0..%d)", device_ids[i], count - 1);
	g_device = device_ids[0];                                     // banks created from now on; existing banks stay where they are
	g_devices.assign(device_ids, device_ids + n_devices);
	return 0;
}

// ------------------------------------------------------------------------------------------------
// where the rand() stream lives (klg_rand_dev.hpp)
// ------------------------------------------------------------------------------------------------
// The reference has ONE rand() sequence per process: klang::random(seed) seeds it, a patch's on() draws detune from it on the host (SuperSaw.k:17),
// Reverb.k's prepare() re-seeds it and draws its tap tables, and every Noise generator draws from it once per sample (klang.h:4949, 5363).  Here the
// per-sample draws happen on the device.  The stream is therefore in one of two places: in the C library (host code draws with rand() itself), or on a
// device as 31 words that the Noise kernels advance block after block with no host round trip.  It moves to a device when a block with Noise
// generators is enqueued, from device to device in the order shards are processed, and back into the C library — one 124-byte copy, after the last
// Noise block has finished — when host code is about to draw: the library's own host draws call rng_to_host() first, the DSL facade's
// klang::random() calls klg_rand_sync().
struct RngChain {
	bool on_device = false; int device = -1;
	std::map<int, uint32_t*> d_state;                              // per device: 32 words
	std::map<int, hipEvent_t> last; std::map<int, bool> used; std::map<int, hipStream_t> last_stream;   // (per device: the stream of the last launch that used that device's words)   // after the last launch that used the state on that device (used: the event has been recorded)
	std::map<std::pair<int, unsigned long long>, uint32_t*> tables;    // (device, per) -> klg_rand::jump_table(per) in HBM
};
static RngChain g_rng;
// the stream's state in device memory of `device`, current as of what has been enqueued, with `st` ordered behind whoever used it last
static int rng_acquire(int device, hipStream_t st, uint32_t** state) {
	std::lock_guard<std::recursive_mutex> lock(RandGuard::mu());
	RngChain& g = g_rng;
	if (!g.d_state.count(device)) {
		RandGuard rg;
		uint32_t* p = nullptr; hipEvent_t e = nullptr;
		HIP_TRY(hipMalloc((void**)&p, 32 * sizeof(uint32_t))); HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		g.d_state[device] = p; g.last[device] = e;
	}
	uint32_t* const mine = g.d_state[device];
	if (!g.on_device) {
		klg_rand::State s;
		if (!klg_rand::libc_state(s)) return fail(KLG_ERR_INVALID, "the C library's rand() is not running its default generator (initstate() with a small state?): the Noise generators have no stream to continue");
		RandStateArg arg; std::memcpy(arg.x, s.x, sizeof arg.x);
		if (g.used.count(device) && g.last_stream[device] != st) HIP_TRY(hipStreamWaitEvent(st, g.last[device], 0));   // (an earlier sequence's last Noise block may still be reading these words on another stream)
		hipLaunchKernelGGL(klg_rand_set, dim3(1), dim3(64), 0, st, mine, arg);
		HIP_TRY(hipGetLastError());
	}
	else if (g.device != device) {
		HIP_TRY(hipStreamWaitEvent(st, g.last[g.device], 0));
		HIP_TRY(hipMemcpyPeerAsync(mine, device, g.d_state[g.device], g.device, klg_rand::DEG * sizeof(uint32_t), st));
	}
	else if (g.last_stream[device] != st) HIP_TRY(hipStreamWaitEvent(st, g.last[device], 0));
	g.on_device = true; g.device = device; g.last_stream[device] = st;
	*state = mine;
	return 0;
}
static int rng_release(int device, hipStream_t st) {                // after the launches that read / advanced the state
	std::lock_guard<std::recursive_mutex> lock(RandGuard::mu());
	HIP_TRY(hipEventRecord(g_rng.last[device], st));
	g_rng.used[device] = true;
	return 0;
}
static int rng_table(int device, unsigned long long per, const uint32_t** table) {
	std::lock_guard<std::recursive_mutex> lock(RandGuard::mu());
	auto it = g_rng.tables.find({ device, per });
	if (it == g_rng.tables.end()) {
		RandGuard rg;
		const std::vector<uint32_t> t = klg_rand::jump_table(per);
		uint32_t* d = nullptr;
		HIP_TRY(hipMalloc((void**)&d, t.size() * 4)); HIP_TRY(hipMemcpy(d, t.data(), t.size() * 4, hipMemcpyHostToDevice));
		it = g_rng.tables.emplace(std::make_pair(device, per), d).first;
	}
	*table = it->second;
	return 0;
}
// host code is about to call rand(): the C library gets the stream back (waits for the last Noise block)
static int rng_to_host() {
	std::lock_guard<std::recursive_mutex> lock(RandGuard::mu());
	RngChain& g = g_rng;
	if (!g.on_device) return 0;
	DeviceGuard bound(g.device); if (!bound.ok) return KLG_ERR_NO_DEVICE;
	klg_rand::State s;
	HIP_TRY(hipEventSynchronize(g.last[g.device]));
	HIP_TRY(hipMemcpy(s.x, g.d_state[g.device], sizeof s.x, hipMemcpyDeviceToHost));
	klg_rand::libc_set_state(s);
	g.on_device = false;
	return 0;
}
extern "C" int klg_rand_sync(void) { return rng_to_host(); }
extern "C" void klg_random_seed(unsigned seed) {                   // klang::random(seed) klang.h:239
	{ std::lock_guard<std::recursive_mutex> lock(RandGuard::mu()); g_rng.on_device = false; }      // (whatever a device held is superseded)
	srand(seed);
}
// `ranks` x `per` draws of the stream, in order, into device memory: out[i * rstride + r] = the (r * per + i)-th rand() from here — what `ranks` Noise
// objects processing `per / draws` samples one after the other would draw (klang.h:4842-4848)
static int rng_fill(int device, hipStream_t st, int* d_out, size_t rstride, const unsigned* d_count, unsigned ranks_bound, unsigned count_imm, int per) {
	// (the whole sequence under the one lock — acquire, the two launches, the event —: two host threads with Noise banks on one device must not interleave their
	//  launches on the 31 state words; the mutex is recursive, rng_acquire / rng_release take it again)
	std::lock_guard<std::recursive_mutex> lock(RandGuard::mu());
	uint32_t* state = nullptr; const uint32_t* table = nullptr;
	if (int rc = rng_table(device, (unsigned long long)per, &table)) return rc;
	if (int rc = rng_acquire(device, st, &state)) return rc;
	RandFillArgs f; f.state = state; f.table = table; f.count = d_count; f.count_imm = count_imm; f.per = per; f.out = d_out; f.rstride = rstride;
	if (ranks_bound > 0) hipLaunchKernelGGL(klg_rand_fill, dim3((ranks_bound + 255u) / 256u), dim3(256), 0, st, f);
	hipLaunchKernelGGL(klg_rand_advance, dim3(1), dim3(64), 0, st, state, table, d_count, count_imm);
	HIP_TRY(hipGetLastError());
	return rng_release(device, st);
}
extern "C" int klg_rand_fill_device(int* d_out, size_t rstride, unsigned ranks, int per, void* hip_stream) {
	if (!d_out || per <= 0 || rstride < ranks || ranks >= (1u << 30)) return fail(KLG_ERR_INVALID, "klg_rand_fill_device: bad arguments");
	if (default_device() < 0) return KLG_ERR_NO_DEVICE;
	int device = g_device;
	{ hipPointerAttribute_t at; if (hipPointerGetAttributes(&at, d_out) == hipSuccess) device = at.device; else (void)hipGetLastError(); }
	DeviceGuard bound(device); if (!bound.ok) return KLG_ERR_NO_DEVICE;
	return rng_fill(device, (hipStream_t)hip_stream, d_out, rstride, nullptr, ranks, ranks, per);
}

// ------------------------------------------------------------------------------------------------
// patch table
// ------------------------------------------------------------------------------------------------
struct DialDef { float min, max, initial; };
struct PatchInfo { int words; int ncontrols; DialDef dials[KLG_MAX_CTL]; int fsines; int note_channels; };   // note_channels: 0 / 1 = the notes' `out` is mono, 2 = stereo (graph patches with ret2)

static const PatchInfo* patch_info(int id) {
	static const PatchInfo T[KLG_PATCH_COUNT] = {
		/* SINE     */ { (int)sizeof(PatchSine::Rec) / 4, 0, {}, 1 },
		/* BSINE    */ { (int)sizeof(PatchBSine::Rec) / 4, 0, {}, 0 },
		/* SUB2A    */ { (int)sizeof(PatchSub2a::Rec) / 4, 0, {}, 0 },
		/* SUB2B    */ { (int)sizeof(PatchSub2b::Rec) / 4, 0, {}, 0 },
		/* SUPERSAW */ { (int)sizeof(PatchSuperSaw::Rec) / 4, 3, { { 0.001f, 1.f, 0.001f }, { 0.f, 1.f, 0.05f }, { 0.f, 1.f, 0.6f } }, 0 },   // SuperSaw.k:38-43
		/* FM3      */ { (int)sizeof(PatchFM<3>::Rec) / 4, 4, { { 0.001f, 10.f, 1.0f }, { 0.f, 10.f, 0.37f }, { 0.f, 10.f, 0.37f }, { 0.f, 1.f, 0.5f } }, 3 },  // FM.k:79-85
		/* FM4      */ { (int)sizeof(PatchFM<4>::Rec) / 4, 5, { { 0.001f, 10.f, 1.0f }, { 0.f, 10.f, 0.37f }, { 0.f, 10.f, 0.37f }, { 0.f, 10.f, 0.37f }, { 0.f, 1.f, 0.5f } }, 4 },
		/* PINGPONG */ { 0, 0, {}, 0 },
		/* REVERB   */ { 0, 0, {}, 0 },
	};
	return (id >= 0 && id < KLG_PATCH_COUNT) ? &T[id] : nullptr;
}

// ------------------------------------------------------------------------------------------------
struct HostVoice { uint8_t stage = ST_OFF; float pitch = 0.f, velocity = 0.f; host::FSineH fs[4]; host::OsmH osm[7]; };
struct Event { int voice, type, payload; unsigned seq; };

struct klg_synth {
	int patch = 0, S = 0, P = 0, V = 0, W = 0, max_block = 0, nctl = 0;
	int note_ch = 1;                             // channels of a voice's own output: 2 for banks of Stereo::Notes whose `out` is {l, r} (klang.h:4721-4733)
	size_t stride = 0;
	host::Fs fs;
	SampleRate dfs;
	hipStream_t stream = nullptr;
	uint32_t* d_state = nullptr;
	float *d_controls = nullptr, *d_partials = nullptr, *d_mix = nullptr, *d_per_voice = nullptr;
	uint32_t* d_scratch_rec = nullptr;
	int grid = 0;
	struct Multi* multi = nullptr;               // a bank sharded over several devices (klg_init with more than one id): this handle only routes
	int device = 0;                              // the GPU this bank (or shard) lives on
	unsigned* d_ticket = nullptr; bool fuse_reduce = false;   // banks of <= KLG_FUSE_MAX_ROWS workgroups: the last one to finish adds the partial rows to the mix (no klg_reduce launch)
	int mix_mode = 0; int* d_solo = nullptr;      // klg_synth_set_mix_mode: KLG_MIX_LAST_ACTIVE keeps one voice per instance (d_solo[synths])
	bool x2 = true;               // KLG_RENDER_X1=1 in the environment selects the one-voice-per-lane kernel (A/B tests)
	int gsp_vpw = 0;              // a graph bank that runs its sample-parallel form (klg_render_sp.hpp): voices per wave (1 / 8), else 0
	hipFunction_t graph_sp_fn[2] = { nullptr, nullptr };   // ... its kernels [per_voice]
	int grid_gsp = 0;
	bool lanes = false;           // SuperSaw banks that do not fill the chip: an oscillator pair — or one oscillator — per lane (klg_render_lanes.hpp); KLG_SUPERSAW_LANES=0 / 1 / 2 forces the choice
	bool pairs = false; int pairs_p = 1;   // ... the pair form (2) and its sample slots per voice (KLG_SUPERSAW_PAIRS_P forces 1 / 2 / 4)
	bool sp = false;              // ... the sample-parallel form (3; the default: klg_render_supersaw_sp.hpp)
	int sp_vpw = 8;               // ... its voices per wave (8 / 4 / 2: by bank size, so that a SIMD has waves to switch between)
	bool sub_sp = false; int grid_sp = 0;   // sub2a banks of up to KLG_SUB2A_SP_MAX_VOICES voices: one voice per wave, samples side by side (klg_render_sub2a_sp.hpp; KLG_SUB2A_SP=0 / 1 forces the choice)
	int grid_lanes = 0;
	// graph patches (klg_graph.hpp): the render kernels come from a hipRTC code object instead of this library
	const graphrt::Compiled* graph = nullptr;
	hipModule_t module = nullptr;
	hipFunction_t graph_fn[2] = { nullptr, nullptr };
	hipError_t launch_error = hipSuccess;
	// sample tables (klg_table_upload): id -> HBM copy; d_tables mirrors `tables` for the kernels
	struct Table { float* d; std::vector<float> h; uint64_t hash; };
	std::vector<Table> tables;
	TableDesc* d_tables = nullptr; size_t d_tables_cap = 0; bool tables_dirty = false;
	unsigned args_gen = 0;                      // bumped whenever a device pointer that RenderArgs carries may have changed (tables_sync, mix mode): captured spans of an older generation are dropped, not replayed
	float* d_note_rings = nullptr;               // note delays of a graph patch: [stride][ring_rows], each voice's lines contiguous
	// Noise generators of a graph patch (note_prepass): the block's draws [n * draws][rand_cap ranks], every voice's rank, the rank kernels' workspace,
	// the pinned word they report (block number, count) in, and what the host's capacity bound is made of
	int *d_rand = nullptr, *d_rank = nullptr; size_t rand_cap = 0; unsigned* d_rank_groups = nullptr; unsigned long long* h_rank_feedback = nullptr;
	enum { ONS_RING = 64 };
	unsigned rank_seq = 0; unsigned long long ons_total = 0, ons_at[ONS_RING] = {};   // voice records sent to the bank so far; ... when block `seq` was enqueued
	float* d_smoothed = nullptr; bool smoothed_host_newer = false, smoothed_device_newer = false;   // Control::smoothed [S][nctl] on the device (klg_note_smooth)
	// host mirrors
	std::vector<host::ControlH> controls;        // [S][nctl]
	std::vector<float> h_controls;               // [S][KLG_MAX_CTL]
	std::vector<float> smoothed;                 // Control::smoothed of every control [S][nctl]: the host's copy (klg_set / get_control_smoothed; the device's is d_smoothed)
	bool controls_dirty = true;
	std::vector<HostVoice> voices;
	std::vector<unsigned> noteOns;               // [S]
	std::vector<unsigned> noteStart;             // [S][128]
	bool stages_dirty = false;
	// events
	std::vector<Event> events;
	std::vector<uint32_t> payload;
	unsigned seq = 0;
	uint32_t* record_sink = nullptr;             // klg_note_record
	bool scripted = false;                       // a klg_script has played on this bank: note stages live on the device only
	std::vector<struct klg_script*> scripts;     // the event scripts compiled for this bank: invalidated when the bank is destroyed
	void* h_stage = nullptr; size_t h_stage_cap = 0;      // pinned
	void* d_stage = nullptr; size_t d_stage_cap = 0;
	hipEvent_t stage_done = nullptr;
	// pinned readback
	float* h_mix = nullptr; uint32_t* h_flags = nullptr; float* h_per_voice = nullptr;
	// timing
	bool timing = false; std::vector<hipEvent_t> tev, tev_aux; int launches = 0, launches_aux = 0;   // (aux: the block's event kernel and reduce, klg_timing_end_aux)
};

static void synth_free(klg_synth* s) {
	if (!s) return;
	if (s->stream) (void)hipStreamSynchronize(s->stream);
	void* dev[] = { s->d_state, s->d_controls, s->d_partials, s->d_mix, s->d_per_voice, s->d_scratch_rec, s->d_stage, s->d_rand, s->d_rank, s->d_rank_groups, s->d_smoothed, s->d_ticket };
	for (void* p : dev) if (p) (void)hipFree(p);
	if (s->d_note_rings) (void)hipFree(s->d_note_rings);
	if (s->d_solo) (void)hipFree(s->d_solo);
	for (auto& t : s->tables) if (t.d) (void)hipFree(t.d);
	if (s->d_tables) (void)hipFree(s->d_tables);
	void* pinned[] = { s->h_stage, s->h_mix, s->h_flags, s->h_per_voice, s->h_rank_feedback };
	for (void* p : pinned) if (p) (void)hipHostFree(p);
	if (s->stage_done) (void)hipEventDestroy(s->stage_done);
	if (s->module) (void)hipModuleUnload(s->module);
	for (auto e : s->tev) (void)hipEventDestroy(e);
	for (auto e : s->tev_aux) (void)hipEventDestroy(e);
	if (s->stream) (void)hipStreamDestroy(s->stream);
	delete s;
}


// ------------------------------------------------------------------------------------------------
// multi-device banks (SURVEY.md §8e behind the C-ABI): klg_init(ids, n > 1), then every synth bank is sharded over the devices
// ------------------------------------------------------------------------------------------------
// RCCL, loaded on first use (a single-device process never needs it)
struct Rccl {
	void* lib = nullptr; std::string error;
	int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
	int (*CommDestroy)(void* comm) = nullptr;
	int (*AllReduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) = nullptr;
	int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
	int (*CommCount)(void* comm, int* count) = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
	bool load() {
		if (lib) return true;
		const char* names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" };
		for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
		if (!lib) { error = std::string("cannot load librccl.so: ") + dlerror(); return false; }
		bool ok = true;
		auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) { ok = false; error = std::string("librccl.so lacks ") + n; } return p; };
		CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
		AllReduce = (decltype(AllReduce))sym("ncclAllReduce"); GroupStart = (decltype(GroupStart))sym("ncclGroupStart"); GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
		GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
		CommCount = (decltype(CommCount))sym("ncclCommCount");
		if (!ok) { dlclose(lib); lib = nullptr; }
		return ok;
	}
};
static Rccl g_rccl;
enum { KLG_NCCL_FLOAT32 = 7, KLG_NCCL_SUM = 0 };                   // ncclFloat32, ncclSum (rccl.h)
// The library is not LINKED against RCCL (librccl.so is loaded on the first multi-device bank: a one-GPU host needs none), so the two enumerators are spelled
// out above — and checked against the header at build time wherever the header is:
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert((int)ncclFloat32 == (int)KLG_NCCL_FLOAT32 && (int)ncclSum == (int)KLG_NCCL_SUM, "rccl.h's ncclFloat32 / ncclSum are not the values klg_api.hip passes to ncclAllReduce");
#endif

struct Multi {
	std::vector<klg_synth*> shard; std::vector<int> first;          // shard i owns synth instances [first[i], first[i + 1])
	std::vector<void*> comm; bool rccl = false;                     // one communicator per shard (distinct devices), else a device-side add
	std::vector<hipEvent_t> done;                                   // same-device combine: shard i's block is ready
	hipEvent_t combined = nullptr, consumed = nullptr;              // the previous block: shard blocks added into shard 0's / shard 0's block added into the caller's
	bool have_combined = false, have_consumed = false;
};
__global__ void klg_add_block(float* dst, const float* src, int count) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < count) dst[i] += src[i]; }
__global__ void klg_sub_block(float* dst, const float* src, int count) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < count) dst[i] -= src[i]; }

// run `f` with shard i's device current for this thread (the shard's own entry points bind themselves; the raw HIP calls inside `f` need it)
template<class F> static auto on_shard(klg_synth* r, size_t i, F&& f) {
	DeviceGuard bound(r->multi->shard[i]->device);
	return f(r->multi->shard[i]);
}
static int shard_of_synth(const klg_synth* r, int synth, int* local) {
	const Multi& m = *r->multi;
	for (size_t i = 0; i + 1 < m.first.size(); i++) if (synth >= m.first[i] && synth < m.first[i + 1]) { *local = synth - m.first[i]; return (int)i; }
	return -1;
}
static int shard_of_voice(const klg_synth* r, int voice, int* local) { int ls = 0; const int i = shard_of_synth(r, voice / r->P, &ls); if (i >= 0) *local = ls * r->P + voice % r->P; return i; }

static void multi_free(klg_synth* r) {
	if (!r || !r->multi) return;
	Multi* m = r->multi;
	for (size_t i = 0; i < m->shard.size(); i++) on_shard(r, i, [&](klg_synth* sh) { if (m->rccl && i < m->comm.size() && m->comm[i]) g_rccl.CommDestroy(m->comm[i]); klg_synth_destroy(sh); return 0; });
	for (auto e : m->done) if (e) (void)hipEventDestroy(e);
	if (m->combined) (void)hipEventDestroy(m->combined);
	if (m->consumed) (void)hipEventDestroy(m->consumed);
	delete m; r->multi = nullptr;
	delete r;
}
// create: `make(device, synths)` is the ordinary single-device creator, run once per device of the list
template<class MAKE> static klg_synth* multi_create(int synths, int notes_per_synth, int max_block, MAKE&& make) {
	const std::vector<int> devs = g_devices;
	const int n = (int)std::min<size_t>(devs.size(), (size_t)synths);
	klg_synth* r = new klg_synth();
	r->multi = new Multi(); r->S = synths; r->P = notes_per_synth; r->V = synths * notes_per_synth; r->max_block = max_block; r->device = devs[0];
	Multi& m = *r->multi;
	const int base = synths / n, extra = synths % n;                // contiguous ranges; the first `extra` shards own one instance more
	m.first.push_back(0);
	for (int i = 0; i < n; i++) {
		const int count = base + (i < extra ? 1 : 0);
		klg_synth* sh = make(devs[(size_t)i], count);
		if (!sh) { const std::string why = g_err; multi_free(r); fail(KLG_ERR_NOMEM, "multi-device bank: shard %d on device %d: %s", i, devs[(size_t)i], why.c_str()); return nullptr; }
		m.shard.push_back(sh); m.first.push_back(m.first.back() + count);
	}
	r->patch = m.shard[0]->patch; r->W = m.shard[0]->W; r->nctl = m.shard[0]->nctl; r->fs = m.shard[0]->fs; r->note_ch = m.shard[0]->note_ch;
	bool distinct = true;
	for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) if (devs[(size_t)i] == devs[(size_t)j]) distinct = false;
	if (distinct && n > 1) {                                        // ONE communicator clique over the shards' devices (RCCL over xGMI)
		if (!g_rccl.load()) { const std::string why = g_rccl.error; multi_free(r); fail(KLG_ERR_HIP, "multi-device bank: %s", why.c_str()); return nullptr; }
		m.comm.assign((size_t)n, nullptr);
		const int rc = g_rccl.CommInitAll(m.comm.data(), n, devs.data());
		if (rc != 0) { const std::string why = g_rccl.GetErrorString(rc); multi_free(r); fail(KLG_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", n, why.c_str()); return nullptr; }
		m.rccl = true;
	}
	else for (int i = 0; i < n; i++) { hipEvent_t e = nullptr; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); m.done.push_back(e); }
	return r;
}

static int multi_process_host(klg_synth* r, float* per_voice, float* const* out, int channels, int n, float* parameters);
static int multi_process_device(klg_synth* r, float* d_mix, int n, void* hip_stream);

enum { KLG_PATCH_GRAPH = 1000 };     // klg_synth::patch of a graph patch (not a klg_patch id)
#ifndef KLG_GSP_MAX_VOICES
#define KLG_GSP_VPW1_MAX_VOICES 2048   // graph banks up to this size: the sample-parallel form with a voice per wave (two waves per SIMD at 2,048 voices)
#define KLG_GSP_VPW4_MAX_VOICES 8192   // ... up to this size with four voices per wave
#define KLG_GSP_MAX_VOICES 32768       // ... up to this size with eight voices per wave (bodies of four oscillators and more); larger banks: a lane (or half a lane) per voice
#endif

static klg_synth* synth_create_common(int device, int patch_id, const PatchInfo* pi, int synths, int notes_per_synth, float sample_rate, int max_block) {
	if (synths <= 0 || notes_per_synth <= 0 || notes_per_synth > 128) { fail(KLG_ERR_INVALID, "klg_synth_create: synths=%d notes_per_synth=%d (1..128, Array<NOTE*,128>)", synths, notes_per_synth); return nullptr; }
	if (max_block <= 0 || max_block > MAX_BLOCK) { fail(KLG_ERR_INVALID, "klg_synth_create: max_block %d not in 1..%d", max_block, (int)MAX_BLOCK); return nullptr; }
	if (!(sample_rate > 0.f)) { fail(KLG_ERR_INVALID, "klg_synth_create: bad sample rate"); return nullptr; }
	RandGuard rg;
	DeviceGuard bound(device);
	if (!bound.ok) return nullptr;
	klg_synth* s = new klg_synth();
	s->device = device;
	s->patch = patch_id; s->S = synths; s->P = notes_per_synth; s->V = synths * notes_per_synth; s->W = pi->words;
	s->max_block = max_block; s->nctl = pi->ncontrols; s->note_ch = pi->note_channels == 2 ? 2 : 1;
	s->stride = ((size_t)s->V + WG - 1) / WG * WG;
	s->fs = host::Fs(sample_rate);
	s->dfs.f = s->fs.f; s->dfs.w = s->fs.w; s->dfs.timeInc = 1.0f / s->fs.f;
	hipDeviceProp_t prop;
	bool ok = hipGetDeviceProperties(&prop, device) == hipSuccess;
	const int groups = (int)(s->stride / WG);
	s->grid = std::min(groups, (ok ? prop.multiProcessorCount : 256) * 8);   // 2 x the 4 resident workgroups per CU (LDS-limited); more groups are strided
	ok = ok && hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess;
	ok = ok && hipMalloc(&s->d_state, (size_t)s->W * s->stride * 4) == hipSuccess;
	ok = ok && hipMalloc(&s->d_controls, (size_t)s->S * KLG_MAX_CTL * 4) == hipSuccess;
	if (patch_id == KLG_PATCH_SUPERSAW) {
		const char* e = getenv("KLG_SUPERSAW_LANES");
		s->lanes = e ? (e[0] == '1' || e[0] == '2' || e[0] == '3') : true;          // (the sample-parallel form at every size: 16,384 voices 26 us per block against 37 for the pair form, 524,288 voices 0.53 ms against 0.58 for a voice per lane)
		s->sp = s->lanes && !(e && (e[0] == '1' || e[0] == '2'));
		s->pairs = s->lanes && e && e[0] == '2';
#ifndef KLG_AB_KERNELS
		// (round 6) the oscillator-per-lane and pair-per-lane kernels are A/B references of klg_render_supersaw_sp: they are in the library only when it is built with
		// klang_amd/csrc/build.sh -DKLG_AB_KERNELS — the shipped .so holds what runs
		if (s->lanes && !s->sp) { fail(KLG_ERR_INVALID, "KLG_SUPERSAW_LANES=%s asks for an A/B reference kernel; this library was built without -DKLG_AB_KERNELS", e); synth_free(s); return nullptr; }
#endif
		// sample slots per voice: enough waves for four per SIMD (a 16,384-voice bank: 4096); a bank that has them anyway keeps one
		const char* pe = getenv("KLG_SUPERSAW_PAIRS_P");
		s->pairs_p = pe ? atoi(pe) : (s->V <= 16384 ? 4 : 1);     // measured: 16,384 voices 58 / 49 / 49 us with 1 / 2 / 4 slots, 32,768 voices 77 / 82 / 87
		if (s->pairs_p != 1 && s->pairs_p != 2 && s->pairs_p != 4) s->pairs_p = 1;
		const char* ve = getenv("KLG_SUPERSAW_VPW");
		s->sp_vpw = ve ? atoi(ve) : (s->V <= KLG_SP_VPW2_MAX_VOICES ? 2 : s->V <= KLG_SP_VPW4_MAX_VOICES ? 4 : 8);
		if (s->sp_vpw != 2 && s->sp_vpw != 4) s->sp_vpw = 8;
		const int per_wg = s->sp ? s->sp_vpw * WAVES : s->pairs ? 16 / s->pairs_p * WAVES : KLG_LANES_VOICES_PER_WG;
		s->grid_lanes = std::min((s->V + per_wg - 1) / per_wg, (ok ? prop.multiProcessorCount : 256) * 8);
	}
	if (patch_id == KLG_PATCH_SUB2A) {
		const char* e = getenv("KLG_SUB2A_SP");
		s->sub_sp = e ? e[0] == '1' : s->V <= KLG_SUB2A_SP_MAX_VOICES;
		s->grid_sp = std::min((s->V + WAVES - 1) / WAVES, (ok ? prop.multiProcessorCount : 256) * 8);
	}
	ok = ok && hipMalloc(&s->d_partials, (size_t)std::max(std::max(s->grid, s->lanes ? s->grid_lanes : 0), s->sub_sp ? s->grid_sp : 0) * max_block * 4 * s->note_ch) == hipSuccess;
	ok = ok && hipMalloc(&s->d_mix, (size_t)2 * max_block * 4) == hipSuccess;
	ok = ok && hipMalloc((void**)&s->d_ticket, sizeof(unsigned)) == hipSuccess && hipMemset(s->d_ticket, 0, sizeof(unsigned)) == hipSuccess;
	ok = ok && hipMalloc(&s->d_scratch_rec, (size_t)std::max(64, s->W) * 4) == hipSuccess;
	ok = ok && hipHostMalloc(&s->h_mix, (size_t)2 * max_block * 4) == hipSuccess;
	ok = ok && hipHostMalloc(&s->h_flags, s->stride * 4) == hipSuccess;
	ok = ok && hipEventCreateWithFlags(&s->stage_done, hipEventDisableTiming) == hipSuccess;
	if (!ok) { fail(KLG_ERR_NOMEM, "klg_synth_create: device allocation failed (%zu state bytes): %s", (size_t)s->W * s->stride * 4, hipGetErrorString(hipGetLastError())); synth_free(s); return nullptr; }
	// every voice starts Off (NoteBase::stage = Off, klang.h:4286): flags plane = ST_OFF, rest zero
	std::vector<uint32_t> init((size_t)s->W * s->stride, 0u);
	std::fill(init.begin(), init.begin() + s->stride, (uint32_t)ST_OFF);
	if (hipMemcpy(s->d_state, init.data(), init.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { fail(KLG_ERR_HIP, "state init copy failed"); synth_free(s); return nullptr; }
	s->controls.resize((size_t)s->S * std::max(1, s->nctl));
	s->smoothed.assign((size_t)s->S * std::max(1, s->nctl), 0.f);
	s->h_controls.assign((size_t)s->S * KLG_MAX_CTL, 0.f);
	for (int i = 0; i < s->S; i++) for (int c = 0; c < s->nctl; c++) {
		s->controls[(size_t)i * s->nctl + c] = { pi->dials[c].min, pi->dials[c].max, pi->dials[c].initial };
		s->smoothed[(size_t)i * s->nctl + c] = pi->dials[c].initial;
		s->h_controls[(size_t)i * KLG_MAX_CTL + c] = pi->dials[c].initial;
	}
	s->voices.resize(s->V);
	if (patch_id == KLG_PATCH_SUPERSAW) for (auto& v : s->voices) for (auto& o : v.osm) o = host::OsmH(0.f);
	if (const char* e = getenv("KLG_RENDER_X1")) s->x2 = !(e[0] == '1');
	s->noteOns.assign(s->S, 0u);
	s->noteStart.assign((size_t)s->S * 128, 0u);
	return s;
}

extern "C" klg_synth* klg_synth_create(int patch_id, int synths, int notes_per_synth, float sample_rate, int max_block) {
	const PatchInfo* pi = patch_info(patch_id);
	if (!pi || pi->words == 0) { fail(KLG_ERR_INVALID, "klg_synth_create: patch %d is not a synth patch", patch_id); return nullptr; }
	if (default_device() < 0) return nullptr;
	if (g_devices.size() > 1 && synths > 0) return multi_create(synths, notes_per_synth, max_block, [&](int device, int count) { return synth_create_common(device, patch_id, pi, count, notes_per_synth, sample_rate, max_block); });
	return synth_create_common(g_device, patch_id, pi, synths, notes_per_synth, sample_rate, max_block);
}

// replaces: constructing a user Synth whose Note::process() is NOT one of the shipped patch ids: the recorded body
// (include/klang_mi355_graph.h) is compiled for gfx950 with hipRTC and rendered by the same klg_render kernel.
static klg_synth* synth_create_graph_on(int device, const char* program, int synths, int notes_per_synth, float sample_rate, int max_block);
extern "C" klg_synth* klg_synth_create_graph(const char* program, int synths, int notes_per_synth, float sample_rate, int max_block) {
	if (default_device() < 0) return nullptr;
	if (g_devices.size() > 1 && synths > 0) return multi_create(synths, notes_per_synth, max_block, [&](int device, int count) { return synth_create_graph_on(device, program, count, notes_per_synth, sample_rate, max_block); });
	return synth_create_graph_on(g_device, program, synths, notes_per_synth, sample_rate, max_block);
}
static klg_synth* synth_create_graph_on(int device, const char* program, int synths, int notes_per_synth, float sample_rate, int max_block) {
	RandGuard rg;
	DeviceGuard bound(device);
	if (!bound.ok) return nullptr;                 // hipRTC / comgr draw temporary names from libc random(): the caller's klang::random(seed) stream must survive
	const graphrt::Compiled* c = nullptr;
	graph::Program g;
	{ const std::string perr = g.parse(program); if (!perr.empty()) { fail(KLG_ERR_INVALID, "klg_synth_create_graph: %s", perr.c_str()); return nullptr; } }
	// two voices per lane (packed fp32, klg_render_x2<P>) when every node / op of the program has a packed form and the kernel
	// keeps both voices in registers; KLG_GRAPH_X1=1 forces one voice per lane (A/B tests)
	const char* x1env = getenv("KLG_GRAPH_X1");
	// Small banks of a recorded patch run its SAMPLE-PARALLEL form where the body has one (klg_render_sp.hpp): a voice per wave up to KLG_GSP_VPW1_MAX_VOICES, four
	// per wave up to KLG_GSP_VPW4_MAX_VOICES, eight up to KLG_GSP_MAX_VOICES; beyond that a lane (or half a lane) per voice fills the chip.  KLG_GRAPH_SP = 0 / 1 / 4 / 8
	// forces none / a voice per wave / four / eight per wave at any size (the same bits every way).
	const long long V_all = (long long)synths * notes_per_synth;
	// (measured, tools/recorded_small_banks.py -> profiles/r06/recorded_small_banks.jsonl: the form that wins follows the waves it makes — up to ~2 per SIMD.  Eight voices
	//  per wave only pay for bodies whose voice-per-lane kernel is slow: several general oscillators per voice — the recorded SuperSaw.k at 16,384 voices 71 us against 242)
	int n_osc = 0; for (const graph::Op& o : g.ops) if (o.code == graph::OP_OSC) n_osc++;
	int sp_vpw = V_all <= KLG_GSP_VPW1_MAX_VOICES ? 1 : V_all <= KLG_GSP_VPW4_MAX_VOICES ? 4 : (V_all <= KLG_GSP_MAX_VOICES && n_osc >= 4) ? 8 : 0;
	if (const char* e = getenv("KLG_GRAPH_SP")) sp_vpw = e[0] == '1' ? 1 : e[0] == '4' ? 4 : e[0] == '8' ? 8 : e[0] == '0' ? 0 : sp_vpw;
	const bool want_x2 = graphrt::x2_eligible(g) && !(x1env && x1env[0] == '1');
	bool x2 = want_x2 && !sp_vpw;
	for (;;) {
		const std::string err = graphrt::compile(program, &c, x2);
		if (!err.empty()) { fail(KLG_ERR_INVALID, "klg_synth_create_graph: %s", err.c_str()); return nullptr; }
		if (!x2) {
			if (sp_vpw && !c->sp && want_x2) { sp_vpw = 0; x2 = true; continue; }   // (a body without a sample-parallel tile: the packed form after all)
			break;
		}
		// worth it only for the smallest patches (tools/graph_width_bench.py, profiles/r01o_graph_width_bench.jsonl): one saw + biquad + ADSR
		// (127-129 registers) gains 10 %, two saws (160) already lose 3 %, seven (296) lose 17 % — and a saw in its general form (duty != 0)
		// loses 25 % even in the smallest.  Hence <= 130 registers; a patch of seven
		// general OSM oscillators needs 250+ and renders 1.6x SLOWER packed (recorded SuperSaw.k, tools/graph_bench_supersaw.py)
		hipModule_t m = nullptr; hipFunction_t fn = nullptr; int scratch = 1, regs = 1 << 20;
		const bool loaded = hipModuleLoadData(&m, c->code.data()) == hipSuccess && hipModuleGetFunction(&fn, m, c->name[0].c_str()) == hipSuccess
			&& hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, fn) == hipSuccess && hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, fn) == hipSuccess;
		if (m) (void)hipModuleUnload(m);
		const char* force = getenv("KLG_GRAPH_X2");
		const bool keep = loaded && scratch == 0 && (regs <= 130 || (force && force[0] == '1'));
		if (getenv("KLG_GRAPH_DEBUG")) fprintf(stderr, "klang-mi355: graph patch, two voices per lane: %s, %d registers, scratch %d bytes per lane -> %s\n", loaded ? "loaded" : "failed to load", regs, scratch, keep ? "used" : "one voice per lane");
		if (keep) break;
		x2 = false;
	}
	PatchInfo pi = {};
	pi.words = c->words; pi.ncontrols = g.nctl; pi.note_channels = c->note_channels;
	for (int i = 0; i < g.nctl; i++) pi.dials[i] = { g.dials[i].min, g.dials[i].max, g.dials[i].initial };
	klg_synth* s = synth_create_common(device, KLG_PATCH_GRAPH, &pi, synths, notes_per_synth, sample_rate, max_block);
	if (!s) return nullptr;
	s->graph = c;
	bool ok = hipModuleLoadData(&s->module, c->code.data()) == hipSuccess;
	for (int i = 0; i < 2 && ok; i++) ok = hipModuleGetFunction(&s->graph_fn[i], s->module, c->name[i].c_str()) == hipSuccess;
	if (ok && c->sp && sp_vpw) {                                       // small banks: the sample-parallel form of the recorded body (klg_render_sp.hpp)
		bool got = true;
		for (int i = 0; i < 2; i++) got = got && hipModuleGetFunction(&s->graph_sp_fn[i], s->module, graphrt::sp_kernel_name(i != 0, sp_vpw).c_str()) == hipSuccess;
		if (!got) (void)hipGetLastError();
		else {
			hipDeviceProp_t prop; const int cus = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 256;
			s->gsp_vpw = sp_vpw; s->grid_gsp = std::min((s->V + sp_vpw * WAVES - 1) / (sp_vpw * WAVES), cus * 8);
			if (s->grid_gsp > s->grid) {                                    // (the partial rows: one per workgroup of the render launch)
				(void)hipFree(s->d_partials); s->d_partials = nullptr;
				ok = hipMalloc(&s->d_partials, (size_t)s->grid_gsp * max_block * 4 * s->note_ch) == hipSuccess;
			}
		}
	}
	if (!ok) { fail(KLG_ERR_HIP, "klg_synth_create_graph: loading the compiled patch failed: %s", hipGetErrorString(hipGetLastError())); synth_free(s); return nullptr; }
	if (c->ring_rows > 0) {                                       // a delay line per voice and Delay member (zero-filled, like a fresh Delay)
		const size_t bytes = (s->stride + 1) * (size_t)c->ring_rows * sizeof(float);      // + the scratch line dead lanes of a live wave write to (klg_render)
		if (bytes > (200ull << 30) || hipMalloc((void**)&s->d_note_rings, bytes) != hipSuccess || hipMemset(s->d_note_rings, 0, bytes) != hipSuccess) {
			fail(KLG_ERR_HIP, "klg_synth_create_graph: %zu voices x %lld delay samples = %.1f GB of delay lines could not be allocated", s->stride, c->ring_rows, bytes / 1e9);
			synth_free(s); return nullptr;
		}
	}
	return s;
}
// Parse, generate and compile a graph program for gfx950 WITHOUT touching a device (build-time / CI check).
// Returns 0 or KLG_ERR_INVALID; the message (or the generated source when `want_source`) is copied into `out`.
extern "C" int klg_graph_check(const char* program, int want_source, char* out, size_t out_cap) {
	RandGuard rg;
	const graphrt::Compiled* c = nullptr;
	const std::string err = graphrt::compile(program, &c, want_source == 2);
	const std::string& msg = err.empty() ? (want_source ? c->source : err) : err;
	if (out && out_cap) { const size_t n = std::min(out_cap - 1, msg.size()); memcpy(out, msg.data(), n); out[n] = 0; }
	return err.empty() ? 0 : fail(KLG_ERR_INVALID, "klg_graph_check: %s", err.c_str());
}

static void scripts_invalidate(klg_synth* s);
extern "C" void klg_synth_destroy(klg_synth* s) { if (!s) return; if (s->multi) { multi_free(s); return; } DeviceGuard bound(s->device); scripts_invalidate(s); synth_free(s); }
extern "C" int klg_synth_voices_per_lane(const klg_synth* s) { if (!s) return KLG_ERR_INVALID; if (s->multi) return klg_synth_voices_per_lane(s->multi->shard[0]); return ((s->patch == KLG_PATCH_SUB2A && s->x2) || (s->graph && s->graph->x2)) ? 2 : 1; }
// replaces: the voice loop of the MONO Synth::process(float*, int, float*) (klang.h:4450-4457), see include/klang_mi355.h
extern "C" int klg_synth_set_mix_mode(klg_synth* s, int mode) {
	if (!s || (mode != KLG_MIX_SUM && mode != KLG_MIX_LAST_ACTIVE)) return fail(KLG_ERR_INVALID, "klg_synth_set_mix_mode: bad handle or mode %d", mode);
	if (s->multi) { for (size_t i = 0; i < s->multi->shard.size(); i++) if (int rc = on_shard(s, i, [&](klg_synth* sh) { return klg_synth_set_mix_mode(sh, mode); })) return rc; s->mix_mode = mode; return 0; }
	KLG_BIND(s);
	if (mode == KLG_MIX_LAST_ACTIVE && !s->d_solo) {
		RandGuard rg;
		HIP_TRY(hipStreamSynchronize(s->stream));
		HIP_TRY(hipMalloc((void**)&s->d_solo, (size_t)s->S * sizeof(int)));
		if (s->patch == KLG_PATCH_SUB2A) s->x2 = false;             // the packed kernel has no per-voice mask at its tile write: one voice per lane in this mode
	}
	s->mix_mode = mode; s->args_gen++;
	return 0;
}
extern "C" int klg_synth_voices(const klg_synth* s) { return s ? s->V : KLG_ERR_INVALID; }
extern "C" int klg_synth_controls(const klg_synth* s) { return s ? s->nctl : KLG_ERR_INVALID; }
extern "C" int klg_synth_note_channels(const klg_synth* s) { return s ? s->note_ch : KLG_ERR_INVALID; }
extern "C" size_t klg_synth_state_bytes(const klg_synth* s) { return s ? (size_t)s->W * 4 : 0; }

// ------------------------------------------------------------------------------------------------
// kernel dispatch by patch
// ------------------------------------------------------------------------------------------------
template<class P> static void launch_render_t(klg_synth* s, const RenderArgs& a, bool pv, hipStream_t st) {
	if (pv) KLG_LAUNCH((klg_render<P, true>), dim3(s->grid), dim3(WG), render_lds_bytes(a.n), st, a);
	else KLG_LAUNCH((klg_render<P, false>), dim3(s->grid), dim3(WG), render_lds_bytes(a.n), st, a);
}
static int render_grid(const klg_synth* s) {      // workgroups (= partial rows) of the render launch
	if (s->gsp_vpw) return s->grid_gsp;
	if (s->lanes) return s->grid_lanes;
	if (s->sub_sp) return s->grid_sp;
	if ((s->patch == KLG_PATCH_SUB2A && s->x2) || (s->graph && s->graph->x2)) return std::min((int)((s->stride + X2_VOICES_PER_WG - 1) / X2_VOICES_PER_WG), s->grid);
	return s->grid;
}
static void launch_render(klg_synth* s, const RenderArgs& a, bool pv, hipStream_t st) {
	if (s->graph) {                                       // hipRTC code object: klg_render<PatchGen, pv>
		RenderArgs args = a;
		void* params[] = { &args };
		s->launch_error = klg_module_launch(s->gsp_vpw ? s->graph_sp_fn[pv ? 1 : 0] : s->graph_fn[pv ? 1 : 0], (unsigned)render_grid(s), WG, (unsigned)render_lds_bytes(a.n, s->note_ch), st, params);
		return;
	}
	if (s->sub_sp) {                                      // sub2a, small banks: one voice per wave, samples side by side (klg_render_sub2a_sp.hpp)
		const dim3 g(render_grid(s)), b(WG);
		if (pv) KLG_LAUNCH(klg_render_sub2a_sp<true>, g, b, render_lds_bytes(a.n), st, a);
		else KLG_LAUNCH(klg_render_sub2a_sp<false>, g, b, render_lds_bytes(a.n), st, a);
		return;
	}
	if (s->patch == KLG_PATCH_SUB2A && s->x2) {           // two voices per lane, packed fp32 (klg_render_x2.hpp)
		const dim3 g(render_grid(s)), b(WG);
		if (pv) KLG_LAUNCH(klg_render_sub2a_x2<true>, g, b, render_lds_bytes(a.n), st, a);
		else KLG_LAUNCH(klg_render_sub2a_x2<false>, g, b, render_lds_bytes(a.n), st, a);
		return;
	}
	if (s->sp) {                                          // SuperSaw, samples side by side and the rare cases apart (small banks)
		const dim3 g(render_grid(s)), b(WG);
		const size_t lds = render_lds_bytes(a.n);
		if (s->sp_vpw == 2) { if (pv) KLG_LAUNCH((klg_render_supersaw_sp<true, 2>), g, b, lds, st, a); else KLG_LAUNCH((klg_render_supersaw_sp<false, 2>), g, b, lds, st, a); }
		else if (s->sp_vpw == 4) { if (pv) KLG_LAUNCH((klg_render_supersaw_sp<true, 4>), g, b, lds, st, a); else KLG_LAUNCH((klg_render_supersaw_sp<false, 4>), g, b, lds, st, a); }
		else if (pv) KLG_LAUNCH((klg_render_supersaw_sp<true, 8>), g, b, lds, st, a);
		else KLG_LAUNCH((klg_render_supersaw_sp<false, 8>), g, b, lds, st, a);
		return;
	}
#ifdef KLG_AB_KERNELS
	if (s->pairs) {                                       // SuperSaw, an oscillator pair per lane (KLG_SUPERSAW_LANES=2)
		const dim3 g(render_grid(s)), b(WG);
		const size_t lds = render_lds_bytes(a.n);
		switch (s->pairs_p * 2 + (pv ? 1 : 0)) {
		case 2: KLG_LAUNCH((klg_render_supersaw_pairs<1, false>), g, b, lds, st, a); break;
		case 3: KLG_LAUNCH((klg_render_supersaw_pairs<1, true>), g, b, lds, st, a); break;
		case 4: KLG_LAUNCH((klg_render_supersaw_pairs<2, false>), g, b, lds, st, a); break;
		case 5: KLG_LAUNCH((klg_render_supersaw_pairs<2, true>), g, b, lds, st, a); break;
		case 8: KLG_LAUNCH((klg_render_supersaw_pairs<4, false>), g, b, lds, st, a); break;
		default: KLG_LAUNCH((klg_render_supersaw_pairs<4, true>), g, b, lds, st, a); break;
		}
		return;
	}
	if (s->lanes) {                                       // SuperSaw, one oscillator per lane (KLG_SUPERSAW_LANES=1)
		const dim3 g(render_grid(s)), b(WG);
		if (pv) KLG_LAUNCH(klg_render_supersaw_lanes<true>, g, b, render_lds_bytes(a.n), st, a);
		else KLG_LAUNCH(klg_render_supersaw_lanes<false>, g, b, render_lds_bytes(a.n), st, a);
		return;
	}
#endif
	switch (s->patch) {
	case KLG_PATCH_SINE: launch_render_t<PatchSine>(s, a, pv, st); break;
	case KLG_PATCH_BSINE: launch_render_t<PatchBSine>(s, a, pv, st); break;
	case KLG_PATCH_SUB2A: launch_render_t<PatchSub2a>(s, a, pv, st); break;
	case KLG_PATCH_SUB2B: launch_render_t<PatchSub2b>(s, a, pv, st); break;
	case KLG_PATCH_SUPERSAW: launch_render_t<PatchSuperSaw>(s, a, pv, st); break;
	case KLG_PATCH_FM3: launch_render_t<PatchFM<3>>(s, a, pv, st); break;
	case KLG_PATCH_FM4: launch_render_t<PatchFM<4>>(s, a, pv, st); break;
	}
}
static void launch_events(klg_synth* s, const EventArgs& a, hipStream_t st) {
	const dim3 g((a.runs + 63) / 64), b(64);
	TimedAux timed(s);
	if (s->graph) { KLG_LAUNCH(klg_apply_records, g, b, 0, st, a, s->W); return; }
	switch (s->patch) {
	case KLG_PATCH_SINE: KLG_LAUNCH(klg_apply_events<PatchSine>, g, b, 0, st, a); break;
	case KLG_PATCH_BSINE: KLG_LAUNCH(klg_apply_events<PatchBSine>, g, b, 0, st, a); break;
	case KLG_PATCH_SUB2A: KLG_LAUNCH(klg_apply_events<PatchSub2a>, g, b, 0, st, a); break;
	case KLG_PATCH_SUB2B: KLG_LAUNCH(klg_apply_events<PatchSub2b>, g, b, 0, st, a); break;
	case KLG_PATCH_SUPERSAW: KLG_LAUNCH(klg_apply_events<PatchSuperSaw>, g, b, 0, st, a); break;
	case KLG_PATCH_FM3: KLG_LAUNCH(klg_apply_events<PatchFM<3>>, g, b, 0, st, a); break;
	case KLG_PATCH_FM4: KLG_LAUNCH(klg_apply_events<PatchFM<4>>, g, b, 0, st, a); break;
	}
}

// ------------------------------------------------------------------------------------------------
// events: host queue -> one staged H2D copy -> klg_apply_events
// ------------------------------------------------------------------------------------------------
// stages the queued events on the device and fills `a`; launches klg_apply_events unless the caller takes the run list into the render launch (`fused`)
enum { KLG_FUSE_MAX_ROWS = 128, KLG_FUSE_MAX_RUNS = 2048 };     // one launch per block: partial rows the last workgroup still adds up quickly (32 loads under way at a time: ~1.1 us per 32 rows; with 256 rows the separate klg_reduce launch is faster: sub2a at 1,024 voices 19.8 against 13.5 us per block); event runs every workgroup can scan
static int flush_events(klg_synth* s, hipStream_t st, EventArgs* fused = nullptr) {
	if (fused) fused->runs = 0;
	if (s->events.empty()) return 0;
	std::stable_sort(s->events.begin(), s->events.end(), [](const Event& a, const Event& b) { return a.voice != b.voice ? a.voice < b.voice : a.seq < b.seq; });
	const size_t E = s->events.size();
	std::vector<int> run_voice, run_first, run_count;
	for (size_t e = 0; e < E; e++) {
		if (e == 0 || s->events[e].voice != s->events[e - 1].voice) { run_voice.push_back(s->events[e].voice); run_first.push_back((int)e); run_count.push_back(0); }
		run_count.back()++;
	}
	const size_t R = run_voice.size();
	const size_t bytes = (3 * R + 2 * E) * 4 + s->payload.size() * 4;
	if (bytes > s->h_stage_cap) {
		HIP_TRY(hipEventSynchronize(s->stage_done));
		if (s->h_stage) (void)hipHostFree(s->h_stage);
		if (s->d_stage) { HIP_TRY(hipStreamSynchronize(st)); (void)hipFree(s->d_stage); }
		s->h_stage_cap = s->d_stage_cap = bytes * 2;
		HIP_TRY(hipHostMalloc(&s->h_stage, s->h_stage_cap));
		HIP_TRY(hipMalloc(&s->d_stage, s->d_stage_cap));
	}
	HIP_TRY(hipEventSynchronize(s->stage_done));           // previous staged copy has left the pinned buffer
	HIP_TRY(hipStreamSynchronize(st));                     // previous apply kernel has consumed the device copy
	int* h = (int*)s->h_stage;
	int* h_rv = h; int* h_rf = h + R; int* h_rc = h + 2 * R; int* h_et = h + 3 * R; int* h_ep = h + 3 * R + E; uint32_t* h_pl = (uint32_t*)(h + 3 * R + 2 * E);
	std::copy(run_voice.begin(), run_voice.end(), h_rv);
	std::copy(run_first.begin(), run_first.end(), h_rf);
	std::copy(run_count.begin(), run_count.end(), h_rc);
	for (size_t e = 0; e < E; e++) { h_et[e] = s->events[e].type; h_ep[e] = s->events[e].payload; }
	std::copy(s->payload.begin(), s->payload.end(), h_pl);
	HIP_TRY(hipMemcpyAsync(s->d_stage, s->h_stage, bytes, hipMemcpyHostToDevice, st));
	HIP_TRY(hipEventRecord(s->stage_done, st));
	int* d = (int*)s->d_stage;
	EventArgs a;
	a.state = s->d_state; a.stride = s->stride;
	a.run_voice = d; a.run_first = d + R; a.run_count = d + 2 * R; a.runs = (int)R;
	a.ev_type = d + 3 * R; a.ev_payload = d + 3 * R + E; a.payload = (const uint32_t*)(d + 3 * R + 2 * E);
	a.fs = s->fs.f;
	if (fused && a.runs <= KLG_FUSE_MAX_RUNS) *fused = a;
	else { launch_events(s, a, st); HIP_TRY(hipGetLastError()); }
	s->events.clear(); s->payload.clear();
	return 0;
}

static int upload_controls(klg_synth* s, hipStream_t st) {
	if (!s->controls_dirty) return 0;
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipMemcpyAsync(s->d_controls, s->h_controls.data(), s->h_controls.size() * 4, hipMemcpyHostToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));   // h_controls is pageable and may change right after return
	s->controls_dirty = false;
	return 0;
}

// device flags -> host NoteBase::stage mirror (a note the GPU stopped becomes Off: klang.h:4455-4456)
static int refresh_stages(klg_synth* s) {
	if (!s->stages_dirty) return 0;
	if (int rc = flush_events(s, s->stream)) return rc;
	HIP_TRY(hipMemcpyAsync(s->h_flags, s->d_state, (size_t)s->V * 4, hipMemcpyDeviceToHost, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	// events of a klg_script are applied on the device only: for such a bank the lane's stage bits ARE the note's stage
	if (s->scripted) for (int v = 0; v < s->V; v++) s->voices[v].stage = (uint8_t)(s->h_flags[v] & 3u);
	else for (int v = 0; v < s->V; v++) if ((s->h_flags[v] & 3u) == (uint32_t)ST_OFF) s->voices[v].stage = ST_OFF;
	s->stages_dirty = false;
	return 0;
}

// ------------------------------------------------------------------------------------------------
// the patches' on() code (host) -> lane record
// ------------------------------------------------------------------------------------------------
static void push_note_on(klg_synth* s, int voice, const void* rec) {
	const uint32_t* w = (const uint32_t*)rec;
	if (s->record_sink) { std::memcpy(s->record_sink, w, (size_t)s->W * 4); return; }      // klg_note_record: the record goes to the caller, nothing is queued
	const int idx = (int)(s->payload.size() / s->W);
	s->ons_total++;
	s->payload.insert(s->payload.end(), w, w + s->W);
	s->events.push_back({ voice, 0, idx, s->seq++ });
}

static void patch_on(klg_synth* s, int synth, int voice, HostVoice* scratch = nullptr) {
	HostVoice& hv = scratch ? *scratch : s->voices[voice];
	const host::Fs& fs = s->fs;
	const host::ControlH* ctl = s->nctl ? &s->controls[(size_t)synth * s->nctl] : nullptr;
	const float f = host::pitch_to_frequency(hv.pitch);          // const param f = pitch -> Frequency;
	switch (s->patch) {
	case KLG_PATCH_SINE: {                                        // osc(f, 0)
		hv.fs[0].set(f, 0.f, fs);
		PatchSine::Rec r; r.flags = ST_SUSTAIN; r.inc = hv.fs[0].inc; r.pos = hv.fs[0].pos;
		push_note_on(s, voice, &r);
	} break;
	case KLG_PATCH_BSINE: {
		host::BOscH o; o.set(f, 0.f, fs);
		PatchBSine::Rec r; r.flags = ST_SUSTAIN; r.increment = o.increment; r.position = o.position; r.offset = o.offset;
		push_note_on(s, voice, &r);
	} break;
	case KLG_PATCH_SUB2A: {                                       // osc(f,0); lpf.reset(); lpf.set(4*f, 2); adsr(0.01,0.1,0.7,0.25)
		host::OsmH osc(0.f); osc.set(f, 0.f, fs);
		host::BiquadLpfH lpf; lpf.reset(); lpf.set(4.f * f, 2.f, fs);
		host::AdsrH adsr; adsr.set(0.01f, 0.1f, 0.7f, 0.25f, fs);
		PatchSub2a::Rec r;
		osc.pack(r.osc); lpf.pack(r.lpf); adsr.pack(r.adsr);
		r.flags = (uint32_t)ST_SUSTAIN | (adsr.env.bits() << 2) | ((uint32_t)osc.state << 8);
		push_note_on(s, voice, &r);
	} break;
	case KLG_PATCH_SUB2B: {                                       // subtractive.k:14-23
		host::OsmH osc(1.0f); osc.set(f, 0.f, fs);
		host::AdsrH adsr; adsr.set(0.f, 0.f, 1.f, 0.25f, fs);
		host::EnvH env; const float xy[6] = { 0.f, f * 2.f, 0.25f, f * 10.f, 2.f, f * 5.f }; env.set_points(3, xy, fs);
		host::BiquadLpfH filter; filter.reset();
		PatchSub2b::Rec r;
		osc.pack(r.osc); adsr.pack(r.adsr);
		r.env.r_out = env.r_out; r.env.r_target = env.r_target; r.env.r_rate = env.r_rate; r.env.time = env.time;
		for (int k = 0; k < 3; k++) { r.env.px[k] = env.px[k]; r.env.py[k] = env.py[k]; }
		r.filter.f = filter.f; r.filter.Q = filter.Q; filter.pack(r.filter.c);
		r.flags = (uint32_t)ST_SUSTAIN | (adsr.env.bits() << 2) | (env.bits() << 8) | ((uint32_t)osc.state << 14);
		push_note_on(s, voice, &r);
	} break;
	case KLG_PATCH_SUPERSAW: {                                    // SuperSaw.k:12-19
		(void)rng_to_host();                                       // (random() below draws from the C library: the stream comes home if a Noise bank holds it)
		const float detune = (float)(0.01 * (double)ctl[2].value * (double)f);
		PatchSuperSaw::Rec r;
		uint32_t flags = ST_SUSTAIN;
		for (int k = 0; k < 7; k++) {
			const double d = (double)((float)(k - 3) * detune) * host::random_d(0.999, 1.001);
			hv.osm[k].set(f + (float)d, 0.f, ctl[1].value, fs);
			hv.osm[k].pack(r.osc[k]);
			flags |= (uint32_t)hv.osm[k].state << (8 + 2 * k);
		}
		host::AdsrH adsr; adsr.set(ctl[0].value, 0.25f, 1.0f, 0.5f, fs);
		adsr.pack(r.adsr);
		r.flags = flags | (adsr.env.bits() << 2);
		push_note_on(s, voice, &r);
	} break;
	case KLG_PATCH_FM3: case KLG_PATCH_FM4: {                     // FM.k:36-55 / 4-operator variant (oracle/ref/ref_fm.cpp)
		const int NOPS = s->patch == KLG_PATCH_FM3 ? 3 : 4;
		const float fc = f, fd = fc * ctl[0].value;
		static const float e1[4] = { 0, 0, 3, 1 }, e2[4] = { 0, 1.5f, 3, 0.5f }, e3[4] = { 0, 1, 2, 0.25f };
		const float* envs[4] = { e1, e2, NOPS == 4 ? e3 : nullptr, nullptr };
		uint32_t rec_words[64] = { 0 };
		uint32_t flags = ST_SUSTAIN, meta = 0;
		OpRec* ops = (OpRec*)(rec_words + 2);
		for (int k = 0; k < NOPS; k++) {
			hv.fs[k].set(k == NOPS - 1 ? fc : fd, 0.f, fs);
			host::EnvH env;                                      // Operator::env default = Envelope() : one point (0,1)
			if (envs[k]) env.set_points(2, envs[k], fs);
			else { const float one[2] = { 0.f, 1.f }; env.set_points(1, one, fs); }
			OpRec& q = ops[k];
			q.inc = hv.fs[k].inc; q.pos = hv.fs[k].pos;
			q.r_out = env.r_out; q.r_target = env.r_target; q.r_rate = env.r_rate; q.time = env.time;
			q.px[0] = env.px[0]; q.px[1] = env.px[1]; q.py[0] = env.py[0]; q.py[1] = env.py[1];
			flags |= env.bits() << (8 + 6 * k);
			meta |= (uint32_t)env.npoints << (2 * k);
		}
		host::AdsrH adsr; adsr.set(ctl[NOPS].value, 0.1f, 1.f, 1.f, fs);
		adsr.pack(*(AdsrRec*)(ops + NOPS));
		rec_words[0] = flags | (adsr.env.bits() << 2); rec_words[1] = meta;
		push_note_on(s, voice, rec_words);
	} break;
	}
}

// Notes::assign klang.h:4336-4372
static int synth_assign(klg_synth* s, int synth) {
	HostVoice* n = &s->voices[(size_t)synth * s->P];
	unsigned* start = &s->noteStart[(size_t)synth * 128];
	unsigned& ons = s->noteOns[synth];
	for (int i = 0; i < s->P; i++) if (n[i].stage == ST_OFF) { start[i] = ons++; return i; }
	int oldest = -1; unsigned oldest_start = 0;
	for (int i = 0; i < s->P; i++) if (n[i].stage == ST_RELEASE && (oldest == -1 || start[i] < oldest_start)) { oldest = i; oldest_start = start[i]; }
	if (oldest != -1) { start[oldest] = ons++; return oldest; }
	oldest = -1; oldest_start = 0;
	for (int i = 0; i < s->P; i++) if (oldest == -1 || start[i] < oldest_start) { oldest = i; oldest_start = start[i]; }
	start[oldest] = ons++;
	return oldest;
}

static const char* const kGraphEvents = "graph patches keep on()/off() in the caller (the DSL facade): move voice records with klg_voice_download / klg_voice_upload / klg_voices_upload";
extern "C" int klg_note_on(klg_synth* s, int synth, int pitch, float velocity) {
	if (!s || synth < 0 || synth >= s->S) return fail(KLG_ERR_INVALID, "klg_note_on: bad handle or synth index %d", synth);
	if (s->multi) { int ls = 0; const int i = shard_of_synth(s, synth, &ls); return on_shard(s, (size_t)i, [&](klg_synth* sh) { return klg_note_on(sh, ls, pitch, velocity); }); }
	if (s->graph) return fail(KLG_ERR_INVALID, "klg_note_on: %s", kGraphEvents);
	KLG_BIND(s);
	if (int rc = refresh_stages(s)) return rc;
	const int slot = synth_assign(s, synth);
	const int voice = synth * s->P + slot;
	HostVoice& hv = s->voices[voice];
	hv.stage = ST_ONSET;                                         // NoteBase::start klang.h:4257-4263
	hv.pitch = (float)pitch; hv.velocity = velocity;
	patch_on(s, synth, voice);
	hv.stage = ST_SUSTAIN;
	return slot;
}

extern "C" int klg_note_off(klg_synth* s, int synth, int pitch, float velocity) {
	(void)velocity;
	if (!s || synth < 0 || synth >= s->S) return fail(KLG_ERR_INVALID, "klg_note_off: bad handle or synth index %d", synth);
	if (s->multi) { int ls = 0; const int i = shard_of_synth(s, synth, &ls); return on_shard(s, (size_t)i, [&](klg_synth* sh) { return klg_note_off(sh, ls, pitch, velocity); }); }
	if (s->graph) return fail(KLG_ERR_INVALID, "klg_note_off: %s", kGraphEvents);
	for (int i = 0; i < s->P; i++) {
		const int voice = synth * s->P + i;
		HostVoice& hv = s->voices[voice];
		if (hv.pitch == (float)pitch && hv.stage == ST_SUSTAIN) { // klang.h:4432 ; NoteBase::release 4265-4275
			hv.stage = ST_RELEASE;
			if (s->patch == KLG_PATCH_SINE || s->patch == KLG_PATCH_BSINE) hv.stage = ST_OFF;   // off() { stop(); }
			s->events.push_back({ voice, 1, -1, s->seq++ });
		}
	}
	return 0;
}

extern "C" int klg_note_on_many(klg_synth* s, int n, const int* synth, const int* pitch, const float* velocity) {
	if (!s || n < 0 || !synth || !pitch || !velocity) return fail(KLG_ERR_INVALID, "klg_note_on_many: bad arguments");
	for (int i = 0; i < n; i++) { const int rc = klg_note_on(s, synth[i], pitch[i], velocity[i]); if (rc < 0) return rc; }
	return 0;
}
extern "C" int klg_note_off_many(klg_synth* s, int n, const int* synth, const int* pitch, const float* velocity) {
	if (!s || n < 0 || !synth || !pitch || !velocity) return fail(KLG_ERR_INVALID, "klg_note_off_many: bad arguments");
	for (int i = 0; i < n; i++) { const int rc = klg_note_off(s, synth[i], pitch[i], velocity[i]); if (rc < 0) return rc; }
	return 0;
}

extern "C" int klg_set_control(klg_synth* s, int synth, int index, float value) {
	if (!s || synth < 0 || synth >= s->S || index < 0 || index >= s->nctl) return fail(KLG_ERR_INVALID, "klg_set_control: synth %d / control %d out of range", synth, index);
	if (s->multi) { int ls = 0; const int i = shard_of_synth(s, synth, &ls); return klg_set_control(s->multi->shard[(size_t)i], ls, index, value); }
	host::ControlH& c = s->controls[(size_t)synth * s->nctl + index];
	c.set(value);
	s->h_controls[(size_t)synth * KLG_MAX_CTL + index] = c.value;
	s->controls_dirty = true;
	return 0;
}
// Control::smoothed lives on the device once a block has run klg_note_smooth: the host's copy is refreshed when somebody asks
static int smoothed_pull(klg_synth* s) {
	if (!s->smoothed_device_newer || !s->d_smoothed) return 0;
	KLG_BIND(s);
	HIP_TRY(hipStreamSynchronize(s->stream)); HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipMemcpy(s->smoothed.data(), s->d_smoothed, s->smoothed.size() * 4, hipMemcpyDeviceToHost));
	s->smoothed_device_newer = false;
	return 0;
}
extern "C" int klg_set_control_smoothed(klg_synth* s, int synth, int index, float smoothed) {
	if (!s || synth < 0 || synth >= s->S || index < 0 || index >= s->nctl) return fail(KLG_ERR_INVALID, "klg_set_control_smoothed: synth %d / control %d out of range", synth, index);
	if (s->multi) { int ls = 0; const int i = shard_of_synth(s, synth, &ls); return klg_set_control_smoothed(s->multi->shard[(size_t)i], ls, index, smoothed); }
	if (int rc = smoothed_pull(s)) return rc;
	s->smoothed[(size_t)synth * s->nctl + index] = smoothed;
	s->smoothed_host_newer = true;
	return 0;
}
extern "C" int klg_get_control_smoothed(klg_synth* s, int synth, int index, float* smoothed) {
	if (!s || !smoothed || synth < 0 || synth >= s->S || index < 0 || index >= s->nctl) return fail(KLG_ERR_INVALID, "klg_get_control_smoothed: out of range");
	if (s->multi) { int ls = 0; const int i = shard_of_synth(s, synth, &ls); return klg_get_control_smoothed(s->multi->shard[(size_t)i], ls, index, smoothed); }
	if (int rc = smoothed_pull(s)) return rc;
	*smoothed = s->smoothed[(size_t)synth * s->nctl + index];
	return 0;
}
extern "C" int klg_get_control(klg_synth* s, int synth, int index, float* value) {
	if (!s || !value || synth < 0 || synth >= s->S || index < 0 || index >= s->nctl) return fail(KLG_ERR_INVALID, "klg_get_control: out of range");
	if (s->multi) { int ls = 0; const int i = shard_of_synth(s, synth, &ls); return klg_get_control(s->multi->shard[(size_t)i], ls, index, value); }
	*value = s->controls[(size_t)synth * s->nctl + index].value;
	return 0;
}

// ------------------------------------------------------------------------------------------------
// block processing
// ------------------------------------------------------------------------------------------------
static int tables_sync(klg_synth* s);
// What the notes of a reference Synth SHARE, and therefore see in the order Synth::process walks them — synth by synth, note slot by
// note slot, each sounding note through the whole block (klang.h:4842-4848; Note::process(buffer) 4295-4303 runs all n samples, also
// after a stop()).  Both are settled HERE, per block, once the block's events are on the device — by kernels on the block's own stream,
// with no copy to the host and no wait (klg_rand_dev.hpp):
//  * Noise generators (4947-4951, 5357-5366): one libc rand() per generator and sample, from the process's one sequence.  klg_rand_rank
//    numbers the sounding voices in that order, klg_rand_fill produces rank r's n * draws values from the stream's state on the device
//    (stored [index][rank]; voice v reads column rank[v]), klg_rand_advance moves the state past the block.
//  * controls[i].smooth() (1715): the control is the Synth's, every sounding note advances it: klg_note_smooth, a lane per instance.
// The one thing the host decides is how many ranks the draw buffer must hold.  It never asks: the rank kernel leaves (block number, count) in a
// pinned word the host reads whenever it comes by; a later block can need at most that count + every voice record that went to the bank since.
// Only when that bound exceeds the buffer does the host wait for the exact count of THIS block (and grow the buffer if it must).
static int note_prepass(klg_synth* s, RenderArgs& a, int n, hipStream_t st) {
	const int draws = s->graph->noise_calls;
	if (!s->graph->smooths.empty()) {
		const size_t words = (size_t)s->S * (size_t)std::max(1, s->nctl);
		if (!s->d_smoothed) { RandGuard rg; HIP_TRY(hipMalloc((void**)&s->d_smoothed, words * 4)); s->smoothed_host_newer = true; }
		if (s->smoothed_host_newer) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipMemcpy(s->d_smoothed, s->smoothed.data(), words * 4, hipMemcpyHostToDevice)); s->smoothed_host_newer = false; }
		SmoothArgs q;
		q.state = s->d_state; q.stride = s->stride; q.synths = s->S; q.notes_per_synth = s->P; q.n = n; q.nctl = std::max(1, s->nctl);
		q.controls = s->d_controls; q.smoothed = s->d_smoothed; q.count = 0;
		for (const auto& sm : s->graph->smooths) if (q.count < KLG_MAX_CTL) { q.sm[q.count].word = sm.word; q.sm[q.count].ctl = sm.ctl; q.sm[q.count].calls = sm.calls; q.count++; }
		{ TimedAux timed(s); KLG_LAUNCH(klg_note_smooth, dim3((unsigned)((s->S + 63) / 64)), dim3(64), 0, st, q); }
		HIP_TRY(hipGetLastError());
		s->smoothed_device_newer = true;
	}
	if (draws == 0) return 0;
	const int groups = (s->V + RANK_WG - 1) / RANK_WG;
	if (!s->d_rank) {
		RandGuard rg;                                                  // (allocations must not disturb the C library's generator)
		HIP_TRY(hipMalloc((void**)&s->d_rank, (size_t)s->V * sizeof(int)));
		HIP_TRY(hipMalloc((void**)&s->d_rank_groups, ((size_t)groups + 1) * sizeof(unsigned)));
		HIP_TRY(hipHostMalloc((void**)&s->h_rank_feedback, sizeof(unsigned long long)));
		*s->h_rank_feedback = 0ull;
	}
	const unsigned seq = ++s->rank_seq;                                // (block numbers start at 1: a feedback word of 0 = nothing reported yet)
	s->ons_at[seq % klg_synth::ONS_RING] = s->ons_total;
	RankArgs r;
	r.flags = s->d_state; r.voices = s->V; r.rank = s->d_rank; r.block_counts = s->d_rank_groups; r.count = s->d_rank_groups + groups; r.feedback = s->h_rank_feedback; r.seq = seq;
	{
		TimedAux timed(s);
		if (groups == 1) KLG_LAUNCH(klg_rand_rank, dim3(1), dim3(RANK_WG), 0, st, r, 1);
		else {
			KLG_LAUNCH(klg_rand_count, dim3((unsigned)groups), dim3(RANK_WG), 0, st, r);
			hipLaunchKernelGGL(klg_rand_scan, dim3(1), dim3(RANK_WG), 0, st, r, groups);
			hipLaunchKernelGGL(klg_rand_rank, dim3((unsigned)groups), dim3(RANK_WG), 0, st, r, 0);
		}
	}
	HIP_TRY(hipGetLastError());
	// how many ranks can this block have?
	const unsigned long long fb = *(volatile unsigned long long*)s->h_rank_feedback;
	const unsigned seen_seq = (unsigned)(fb >> 32), seen_count = (unsigned)fb;
	size_t bound = (size_t)s->V;
	if (seen_seq != 0 && seq - seen_seq < (unsigned)klg_synth::ONS_RING) bound = std::min<size_t>(bound, (size_t)seen_count + (size_t)(s->ons_total - s->ons_at[seen_seq % klg_synth::ONS_RING]));
	if (bound > s->rand_cap) {
		HIP_TRY(hipStreamSynchronize(st));                             // (rare: the first block, a burst of note-ons beyond everything seen so far)
		const size_t exact = (size_t)(unsigned)*(volatile unsigned long long*)s->h_rank_feedback;
		bound = exact;
		if (exact > s->rand_cap) {
			RandGuard rg;
			if (s->d_rand) (void)hipFree(s->d_rand);
			s->d_rand = nullptr;
			s->rand_cap = std::min<size_t>(((size_t)s->V + 255) / 256 * 256, (std::max<size_t>(exact + exact / 2, 1024) + 255) / 256 * 256);
			const size_t bytes = s->rand_cap * (size_t)(s->max_block + KLG_NZ_GROUP) * (size_t)draws * sizeof(int);   // (+ the rows the last group of a block reads ahead: klg_render nz_stage)
			if (hipMalloc((void**)&s->d_rand, bytes) != hipSuccess) { s->rand_cap = 0; return fail(KLG_ERR_NOMEM, "the Noise generators' draws of %zu sounding voices x %d samples x %d generators (%.2f GB) could not be allocated", exact, s->max_block, draws, bytes / 1e9); }
		}
	}
	if (int rc = rng_fill(s->device, st, s->d_rand, s->rand_cap, r.count, (unsigned)std::min(bound, s->rand_cap), 0u, n * draws)) return rc;
	a.rand = s->d_rand; a.rand_base = s->d_rank; a.rstride = s->rand_cap;
	return 0;
}
// does the render launch of this bank apply events / combine its partial rows itself (RenderArgs::ev / ticket)?  The kernels that can: klg_render<P>
// (hand-written and generated patches, one voice per lane), klg_render_sub2a_x2, klg_render_supersaw_pairs.
static bool fusing_kernel(const klg_synth* s) { return !(s->graph && s->graph->x2) && !(s->lanes && !s->pairs && !s->sp); }
static int enqueue_block(klg_synth* s, float* d_mix, int n, bool per_voice, hipStream_t st, const EventArgs* script_events = nullptr) {
	const bool prepass = s->graph && (s->graph->noise_calls > 0 || !s->graph->smooths.empty());
	const char* fuse_env = getenv("KLG_FUSE"); const bool fuse_off = fuse_env && fuse_env[0] == '0';   // KLG_FUSE=0: always the separate launches (A/B; read per block so that a test can flip it)
	const bool small = !fuse_off && fusing_kernel(s) && render_grid(s) <= KLG_FUSE_MAX_ROWS;
	const bool fuse_events = small && !prepass && s->mix_mode != KLG_MIX_LAST_ACTIVE;   // (those two read the note stages between the events and the render)
	EventArgs ev; ev.runs = 0;
	if (script_events) {                                            // a klg_script's block: anything queued interactively comes first, as a launch of its own
		if (int rc = flush_events(s, st)) return rc;
		s->ons_total += (unsigned long long)script_events->runs;
		if (fuse_events && script_events->runs <= KLG_FUSE_MAX_RUNS) ev = *script_events;
		else if (script_events->runs > 0) { launch_events(s, *script_events, st); HIP_TRY(hipGetLastError()); }
	}
	else if (int rc = flush_events(s, st, fuse_events ? &ev : nullptr)) return rc;
	// Event runs handed to the render launch (`ev`) have left the host queue.  If this block bails out before that launch is issued (a control / table
	// upload failed, the generated kernel could not be launched), they still take effect: applied by a klg_apply_events launch of their own.
	struct FusedEvents { klg_synth* s; const EventArgs* ev; hipStream_t st; bool consumed;
		~FusedEvents() { if (!consumed && ev->runs > 0) { launch_events(s, *ev, st); (void)hipGetLastError(); } } } fused_events = { s, &ev, st, false };
	if (int rc = upload_controls(s, st)) return rc;
	RenderArgs a;
	a.state = s->d_state; a.stride = s->stride; a.voices = s->V; a.notes_per_synth = s->P; a.n = n;
	a.controls = s->d_controls; a.fs = s->dfs; a.partials = s->d_partials; a.per_voice = per_voice ? s->d_per_voice : nullptr;
	if (int rc = tables_sync(s)) return rc;
	a.tables = s->d_tables;
	a.rings = s->d_note_rings; a.ring_rows = s->graph ? (size_t)s->graph->ring_rows : 0;
	a.solo = nullptr;
	a.rand = nullptr; a.rand_base = nullptr; a.rstride = 0;
	a.ev = ev; a.mix = d_mix; a.mix_channels = 2; a.ticket = small ? s->d_ticket : nullptr;
	if (prepass) if (int rc = note_prepass(s, a, n, st)) return rc;
	if (s->mix_mode == KLG_MIX_LAST_ACTIVE) {                      // after this block's events: which voice of each instance is heard
		hipLaunchKernelGGL(klg_select_last_active, dim3((s->S + 255) / 256), dim3(256), 0, st, (const uint32_t*)s->d_state, s->S, s->P, s->d_solo);
		a.solo = s->d_solo;
	}
	{ TimedLaunch timed(s); launch_render(s, a, per_voice, st); }
	if (s->launch_error != hipSuccess) { const hipError_t e = s->launch_error; s->launch_error = hipSuccess; return fail(KLG_ERR_HIP, "launching the compiled graph patch failed: %s", hipGetErrorString(e)); }
	fused_events.consumed = true;
	if (a.ticket) {}                                                 // (the render launch's last workgroup added the rows)
	else if (s->note_ch == 2) { TimedAux timed(s); KLG_LAUNCH(klg_reduce_stereo, dim3((n + 31) / 32, 2), dim3(1024), 0, st, (const float*)s->d_partials, render_grid(s), n, d_mix, 2); }
	else { TimedAux timed(s); KLG_LAUNCH(klg_reduce, dim3((n + 31) / 32), dim3(1024), 0, st, (const float*)s->d_partials, render_grid(s), n, d_mix, 2); }
	HIP_TRY(hipGetLastError());
	s->stages_dirty = true;
	return 0;
}

static int process_host(klg_synth* s, float* per_voice, float* const* out, int channels, int n, float* parameters) {
	if (!s || n <= 0 || n > s->max_block) return fail(KLG_ERR_INVALID, "klg_process: n=%d not in 1..max_block", n);
	if (out && (channels < 1 || channels > 2)) return fail(KLG_ERR_INVALID, "klg_process: channels must be 1 or 2");
	KLG_BIND(s);
	if (parameters && s->nctl)                                   // sync parameters in (klang.h:4836-4839)
		for (int i = 0; i < s->S; i++) for (int c = 0; c < s->nctl; c++) klg_set_control(s, i, c, parameters[(size_t)i * s->nctl + c]);
	hipStream_t st = s->stream;
	if (per_voice && !s->d_per_voice) {
		HIP_TRY(hipMalloc(&s->d_per_voice, (size_t)s->V * s->max_block * 4 * s->note_ch));
		HIP_TRY(hipHostMalloc(&s->h_per_voice, (size_t)s->V * s->max_block * 4 * s->note_ch));
	}
	bool replace = false;                                          // KLG_MIX_LAST_ACTIVE: a sounding note overwrites the caller's samples (klang.h:4299)
	if (s->mix_mode == KLG_MIX_LAST_ACTIVE) {
		if (int rc = refresh_stages(s)) return rc;
		for (int v = 0; v < s->V && !replace; v++) replace = s->voices[v].stage != ST_OFF;
	}
	HIP_TRY(hipMemsetAsync(s->d_mix, 0, (size_t)2 * n * 4, st));
	if (int rc = enqueue_block(s, s->d_mix, n, per_voice != nullptr, st)) return rc;
	HIP_TRY(hipMemcpyAsync(s->h_mix, s->d_mix, (size_t)2 * n * 4, hipMemcpyDeviceToHost, st));
	if (per_voice) HIP_TRY(hipMemcpyAsync(s->h_per_voice, s->d_per_voice, (size_t)s->V * n * 4 * s->note_ch, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	if (out) for (int c = 0; c < channels; c++) {
		float* dst = out[c]; const float* src = s->h_mix + (size_t)c * n;
		if (replace) std::memcpy(dst, src, (size_t)n * 4);
		else if (s->mix_mode != KLG_MIX_LAST_ACTIVE) for (int i = 0; i < n; i++) dst[i] += src[i];
	}
	if (per_voice) std::memcpy(per_voice, s->h_per_voice, (size_t)s->V * n * 4 * s->note_ch);
	// (the note stages come back when somebody asks — klg_note_on's assign(), klg_voice_stages: refresh_stages() — not 4 bytes per voice
	//  over PCIe after every block: a block's host traffic is its 2 KiB of mix)
	if (parameters && s->nctl)                                   // sync update out (klang.h:4854-4857)
		for (int i = 0; i < s->S; i++) for (int c = 0; c < s->nctl; c++) parameters[(size_t)i * s->nctl + c] = s->controls[(size_t)i * s->nctl + c].value;
	return 0;
}


// every shard renders its block into its own [2][n] (cleared first), then ONE all-reduce (RCCL) — or, for shards sharing a GPU, adds
// on shard 0's stream — leaves the global block in shard 0's d_mix
static int multi_render(klg_synth* r, int n, bool per_voice) {
	Multi& m = *r->multi;
	for (size_t i = 0; i < m.shard.size(); i++) {
		const int rc = on_shard(r, i, [&](klg_synth* sh) -> int {
			if (per_voice && !sh->d_per_voice) { HIP_TRY(hipMalloc(&sh->d_per_voice, (size_t)sh->V * sh->max_block * 4 * sh->note_ch)); HIP_TRY(hipHostMalloc(&sh->h_per_voice, (size_t)sh->V * sh->max_block * 4 * sh->note_ch)); }
			// a shard's block buffer is free again once the previous block has been combined (and, for shard 0, handed to the caller)
			if (m.have_combined) HIP_TRY(hipStreamWaitEvent(sh->stream, m.combined, 0));
			if (i == 0 && m.have_consumed) HIP_TRY(hipStreamWaitEvent(sh->stream, m.consumed, 0));
			HIP_TRY(hipMemsetAsync(sh->d_mix, 0, (size_t)2 * n * 4, sh->stream));
			if (int e = enqueue_block(sh, sh->d_mix, n, per_voice, sh->stream)) return e;
			if (!m.rccl) HIP_TRY(hipEventRecord(m.done[i], sh->stream));
			return 0;
		});
		if (rc) return rc;
	}
	if (m.shard.size() == 1) return 0;
	if (m.rccl) {
		int rc = g_rccl.GroupStart();
		for (size_t i = 0; i < m.shard.size() && rc == 0; i++) rc = g_rccl.AllReduce(m.shard[i]->d_mix, m.shard[i]->d_mix, (size_t)2 * n, KLG_NCCL_FLOAT32, KLG_NCCL_SUM, m.comm[i], m.shard[i]->stream);
		const int rc2 = g_rccl.GroupEnd();
		if (rc != 0 || rc2 != 0) return fail(KLG_ERR_HIP, "ncclAllReduce of the [2][%d] block failed: %s", n, g_rccl.GetErrorString(rc ? rc : rc2));
		return 0;
	}
	return on_shard(r, 0, [&](klg_synth* s0) -> int {
		for (size_t i = 1; i < m.shard.size(); i++) {
			HIP_TRY(hipStreamWaitEvent(s0->stream, m.done[i], 0));
			hipLaunchKernelGGL(klg_add_block, dim3((2 * n + 255) / 256), dim3(256), 0, s0->stream, s0->d_mix, (const float*)m.shard[i]->d_mix, 2 * n);
		}
		HIP_TRY(hipGetLastError());
		if (!m.combined) HIP_TRY(hipEventCreateWithFlags(&m.combined, hipEventDisableTiming));
		HIP_TRY(hipEventRecord(m.combined, s0->stream)); m.have_combined = true;
		return 0;
	});
}
static int multi_process_host(klg_synth* r, float* per_voice, float* const* out, int channels, int n, float* parameters) {
	if (n <= 0 || n > r->max_block) return fail(KLG_ERR_INVALID, "klg_process: n=%d not in 1..max_block", n);
	if (out && (channels < 1 || channels > 2)) return fail(KLG_ERR_INVALID, "klg_process: channels must be 1 or 2");
	Multi& m = *r->multi;
	if (parameters && r->nctl) for (int i = 0; i < r->S; i++) for (int c = 0; c < r->nctl; c++) klg_set_control(r, i, c, parameters[(size_t)i * r->nctl + c]);
	bool replace = false;
	if (r->mix_mode == KLG_MIX_LAST_ACTIVE) for (size_t i = 0; i < m.shard.size(); i++) {
		if (int rc = on_shard(r, i, [&](klg_synth* sh) { return refresh_stages(sh); })) return rc;
		for (int v = 0; v < m.shard[i]->V && !replace; v++) replace = m.shard[i]->voices[v].stage != ST_OFF;
	}
	if (int rc = multi_render(r, n, per_voice != nullptr)) return rc;
	size_t v0 = 0;
	for (size_t i = 0; i < m.shard.size(); i++) {
		const int rc = on_shard(r, i, [&](klg_synth* sh) -> int {
			if (i == 0) HIP_TRY(hipMemcpyAsync(sh->h_mix, sh->d_mix, (size_t)2 * n * 4, hipMemcpyDeviceToHost, sh->stream));
			if (per_voice) HIP_TRY(hipMemcpyAsync(sh->h_per_voice, sh->d_per_voice, (size_t)sh->V * n * 4 * sh->note_ch, hipMemcpyDeviceToHost, sh->stream));
			HIP_TRY(hipStreamSynchronize(sh->stream));
			if (per_voice) std::memcpy(per_voice + v0 * (size_t)n * sh->note_ch, sh->h_per_voice, (size_t)sh->V * n * 4 * sh->note_ch);
			return 0;                                                   // (the note stages stay on the device until asked for: refresh_stages)
		});
		if (rc) return rc;
		v0 += (size_t)m.shard[i]->V;
	}
	const float* mix = m.shard[0]->h_mix;
	if (out) for (int c = 0; c < channels; c++) {
		float* dst = out[c]; const float* src = mix + (size_t)c * n;
		if (replace) std::memcpy(dst, src, (size_t)n * 4);
		else if (r->mix_mode != KLG_MIX_LAST_ACTIVE) for (int i = 0; i < n; i++) dst[i] += src[i];
	}
	if (parameters && r->nctl) for (int i = 0; i < r->S; i++) for (int c = 0; c < r->nctl; c++) klg_get_control(r, i, c, &parameters[(size_t)i * r->nctl + c]);
	return 0;
}
// throughput entry: the global block is ADDED to the caller's d_mix (a buffer on the device of shard 0) on shard 0's stream order
static int multi_process_device(klg_synth* r, float* d_mix, int n, void* hip_stream) {
	if (!d_mix || n <= 0 || n > r->max_block) return fail(KLG_ERR_INVALID, "klg_process_device: bad arguments (n=%d)", n);
	if (int rc = multi_render(r, n, false)) return rc;
	return on_shard(r, 0, [&](klg_synth* s0) -> int {
		hipStream_t user = hip_stream ? (hipStream_t)hip_stream : s0->stream;
		Multi& m = *r->multi;
		if (user != s0->stream) {                                    // the caller's stream continues after the combine
			if (m.done.empty()) { hipEvent_t e = nullptr; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); m.done.push_back(e); }
			HIP_TRY(hipEventRecord(m.done[0], s0->stream));
			HIP_TRY(hipStreamWaitEvent(user, m.done[0], 0));
		}
		hipLaunchKernelGGL(klg_add_block, dim3((2 * n + 255) / 256), dim3(256), 0, user, d_mix, (const float*)s0->d_mix, 2 * n);
		HIP_TRY(hipGetLastError());
		if (user != s0->stream) {                                    // shard 0 may not clear its block before the caller's stream has read it
			if (!m.consumed) HIP_TRY(hipEventCreateWithFlags(&m.consumed, hipEventDisableTiming));
			HIP_TRY(hipEventRecord(m.consumed, user)); m.have_consumed = true;
		}
		return 0;
	});
}

extern "C" int klg_process(klg_synth* s, float* const* out, int channels, int n, float* parameters) {
	if (!out) return fail(KLG_ERR_INVALID, "klg_process: out is NULL");
	if (s && s->multi) return multi_process_host(s, nullptr, out, channels, n, parameters);
	return process_host(s, nullptr, out, channels, n, parameters);
}
extern "C" int klg_process_voices(klg_synth* s, float* per_voice, float* const* out, int channels, int n) {
	if (!per_voice) return fail(KLG_ERR_INVALID, "klg_process_voices: per_voice is NULL");
	if (s && s->multi) return multi_process_host(s, per_voice, out, channels, n, nullptr);
	return process_host(s, per_voice, out, channels, n, nullptr);
}
extern "C" int klg_voice_stages(klg_synth* s, uint8_t* stages, int n_voices) {
	if (!s || !stages || n_voices < 0 || n_voices > s->V) return fail(KLG_ERR_INVALID, "klg_voice_stages: bad arguments");
	if (s->multi) {
		int v0 = 0;
		for (size_t i = 0; i < s->multi->shard.size() && v0 < n_voices; i++) {
			const int take = std::min(n_voices - v0, s->multi->shard[i]->V);
			if (int rc = on_shard(s, i, [&](klg_synth* sh) { return klg_voice_stages(sh, stages + v0, take); })) return rc;
			v0 += take;
		}
		return 0;
	}
	KLG_BIND(s);
	if (int rc = refresh_stages(s)) return rc;
	for (int v = 0; v < n_voices; v++) stages[v] = s->voices[v].stage;
	return 0;
}

extern "C" int klg_process_device(klg_synth* s, float* d_mix, int n, void* hip_stream) {
	if (s && s->multi) return multi_process_device(s, d_mix, n, hip_stream);
	if (!s || !d_mix || n <= 0 || n > s->max_block) return fail(KLG_ERR_INVALID, "klg_process_device: bad arguments (n=%d)", n);
	KLG_BIND(s);
	return enqueue_block(s, d_mix, n, false, hip_stream ? (hipStream_t)hip_stream : s->stream);
}
extern "C" int klg_sync(klg_synth* s) {
	if (!s) return fail(KLG_ERR_INVALID, "klg_sync: NULL handle");
	if (s->multi) { for (size_t i = 0; i < s->multi->shard.size(); i++) if (int rc = on_shard(s, i, [&](klg_synth* sh) { return klg_sync(sh); })) return rc; return 0; }
	KLG_BIND(s);
	HIP_TRY(hipStreamSynchronize(s->stream));
	HIP_TRY(hipDeviceSynchronize());
	return 0;
}

extern "C" int klg_voice_download(klg_synth* s, int voice, void* state, size_t bytes) {
	if (!s || !state || voice < 0 || voice >= s->V || bytes != (size_t)s->W * 4) return fail(KLG_ERR_INVALID, "klg_voice_download: bad arguments (record is %d bytes)", s ? s->W * 4 : 0);
	if (s->multi) { int lv = 0; const int i = shard_of_voice(s, voice, &lv); return on_shard(s, (size_t)i, [&](klg_synth* sh) { return klg_voice_download(sh, lv, state, bytes); }); }
	KLG_BIND(s);
	if (int rc = flush_events(s, s->stream)) return rc;
	hipLaunchKernelGGL(klg_copy_record, dim3(1), dim3(128), 0, s->stream, s->d_state, s->stride, voice, s->d_scratch_rec, s->W, 0);
	HIP_TRY(hipMemcpyAsync(state, s->d_scratch_rec, bytes, hipMemcpyDeviceToHost, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	return 0;
}
extern "C" int klg_voice_upload(klg_synth* s, int voice, const void* state, size_t bytes) {
	if (!s || !state || voice < 0 || voice >= s->V || bytes != (size_t)s->W * 4) return fail(KLG_ERR_INVALID, "klg_voice_upload: bad arguments (record is %d bytes)", s ? s->W * 4 : 0);
	if (s->multi) { int lv = 0; const int i = shard_of_voice(s, voice, &lv); return on_shard(s, (size_t)i, [&](klg_synth* sh) { return klg_voice_upload(sh, lv, state, bytes); }); }
	KLG_BIND(s);
	if (int rc = flush_events(s, s->stream)) return rc;
	HIP_TRY(hipMemcpyAsync(s->d_scratch_rec, state, bytes, hipMemcpyHostToDevice, s->stream));
	hipLaunchKernelGGL(klg_copy_record, dim3(1), dim3(128), 0, s->stream, s->d_state, s->stride, voice, s->d_scratch_rec, s->W, 1);
	HIP_TRY(hipStreamSynchronize(s->stream));
	s->voices[voice].stage = (uint8_t)(((const uint32_t*)state)[0] & 3u);
	s->ons_total++;
	return 0;
}

// Bulk form: record i (klg_synth_state_bytes() bytes each, AoS) replaces the state of voices[i]; queued like note events
// and applied by one kernel at the start of the next block.
extern "C" int klg_voices_upload(klg_synth* s, int n, const int* voices, const void* states) {
	if (!s || n < 0 || !voices || !states) return fail(KLG_ERR_INVALID, "klg_voices_upload: bad arguments");
	for (int i = 0; i < n; i++) if (voices[i] < 0 || voices[i] >= s->V) return fail(KLG_ERR_INVALID, "klg_voices_upload: voice %d out of range", voices[i]);
	if (s->multi) {
		for (int i = 0; i < n; i++) { int lv = 0; const int sh = shard_of_voice(s, voices[i], &lv); if (int rc = klg_voices_upload(s->multi->shard[(size_t)sh], 1, &lv, (const uint32_t*)states + (size_t)i * s->W)) return rc; }
		return 0;
	}
	const uint32_t* w = (const uint32_t*)states;
	for (int i = 0; i < n; i++) {
		push_note_on(s, voices[i], w + (size_t)i * s->W);
		s->voices[voices[i]].stage = (uint8_t)(w[(size_t)i * s->W] & 3u);
	}
	return 0;
}

// Delay::clear() of a note's delay line (klang.h:3392-3394), in stream order with the blocks
extern "C" int klg_voice_delay_clear(klg_synth* s, int voice, int delay_index) {
	if (s && s->multi) { int lv = 0; const int i = shard_of_voice(s, voice, &lv); if (i < 0) return fail(KLG_ERR_INVALID, "klg_voice_delay_clear: voice %d out of range", voice); return on_shard(s, (size_t)i, [&](klg_synth* sh) { return klg_voice_delay_clear(sh, lv, delay_index); }); }
	if (!s || !s->graph || !s->d_note_rings) return fail(KLG_ERR_INVALID, "klg_voice_delay_clear: the bank has no note delays (graph %p, %lld ring rows, lines %p)", s ? (const void*)s->graph : nullptr, s && s->graph ? s->graph->ring_rows : -1ll, s ? (const void*)s->d_note_rings : nullptr);
	if (voice < 0 || voice >= s->V || delay_index < 0 || delay_index >= (int)s->graph->delays.size()) return fail(KLG_ERR_INVALID, "klg_voice_delay_clear: voice %d / delay %d out of range", voice, delay_index);
	const long long row0 = s->graph->delays[(size_t)delay_index].first; const int size = s->graph->delays[(size_t)delay_index].second;
	float* line = s->d_note_rings + (size_t)voice * (size_t)s->graph->ring_rows + (size_t)row0;
	KLG_BIND(s);
	HIP_TRY(hipMemsetAsync(line, 0, (size_t)size * sizeof(float), s->stream));
	return 0;
}

// ---- sample tables (include/klang_mi355.h: klg_table_upload) ----
static int table_add(klg_synth* s, const float* samples, int n) {
	klg_synth::Table t; t.d = nullptr; t.h.assign(samples, samples + n);
	uint64_t h = 1469598103934665603ull;
	for (int i = 0; i < n; i++) { uint32_t u; memcpy(&u, &samples[i], 4); h = (h ^ u) * 1099511628211ull; }
	t.hash = h;
	if (hipMalloc((void**)&t.d, (size_t)n * sizeof(float)) != hipSuccess) return fail(KLG_ERR_HIP, "klg_table_upload: hipMalloc of %d floats failed", n);
	if (hipMemcpy(t.d, samples, (size_t)n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(t.d); return fail(KLG_ERR_HIP, "klg_table_upload: copy failed"); }
	s->tables.push_back(std::move(t));
	s->tables_dirty = true;
	return (int)s->tables.size() - 1;
}
extern "C" int klg_table_upload(klg_synth* s, const float* samples, int n, int dedup) {
	if (!s || !samples || n < 2 || n > (1 << 26)) return fail(KLG_ERR_INVALID, "klg_table_upload: bad arguments (2 .. 2^26 samples)");
	if (s->multi) { int id = -1; for (size_t i = 0; i < s->multi->shard.size(); i++) { id = on_shard(s, i, [&](klg_synth* sh) { return klg_table_upload(sh, samples, n, dedup); }); if (id < 0) return id; } return id; }   // the same tables in the same order on every device: one id
	if (s->patch != KLG_PATCH_GRAPH) return fail(KLG_ERR_INVALID, "klg_table_upload: only graph banks (klg_synth_create_graph) read tables");
	RandGuard rg;
	KLG_BIND(s);
	if (s->tables.empty()) { const float zero[2] = { 0.f, 0.f }; const int id0 = table_add(s, zero, 2); if (id0 < 0) return id0; }   // id 0: what an all-zero record reads
	if (dedup) {
		uint64_t h = 1469598103934665603ull;
		for (int i = 0; i < n; i++) { uint32_t u; memcpy(&u, &samples[i], 4); h = (h ^ u) * 1099511628211ull; }
		for (size_t k = 1; k < s->tables.size(); k++) if (s->tables[k].hash == h && (int)s->tables[k].h.size() == n && !memcmp(s->tables[k].h.data(), samples, (size_t)n * sizeof(float))) return (int)k;
	}
	return table_add(s, samples, n);
}
// the descriptor array the kernels index; re-uploaded (on the bank's stream order: synchronously, before the next launch) when a table was added
static int tables_sync(klg_synth* s) {
	if (!s->tables_dirty) return 0;
	HIP_TRY(hipDeviceSynchronize());                       // rare (a new table): nothing in flight may still read the old array
	if (s->tables.size() > s->d_tables_cap) {
		if (s->d_tables) (void)hipFree(s->d_tables);
		s->d_tables_cap = s->tables.size() * 2 + 8;
		HIP_TRY(hipMalloc((void**)&s->d_tables, s->d_tables_cap * sizeof(TableDesc)));
	}
	std::vector<TableDesc> h(s->tables.size());
	for (size_t k = 0; k < h.size(); k++) { h[k].p = s->tables[k].d; h[k].size = (int)s->tables[k].h.size(); h[k].pad_ = 0; }
	HIP_TRY(hipMemcpy(s->d_tables, h.data(), h.size() * sizeof(TableDesc), hipMemcpyHostToDevice));
	s->tables_dirty = false;
	s->args_gen++;                                        // (a captured span holds the old d_tables pointer in its kernel parameters)
	return 0;
}

// ------------------------------------------------------------------------------------------------
// Event scripts resident in HBM (include/klang_mi355.h: klg_script_*): offline / throughput rendering of a known event stream
// ------------------------------------------------------------------------------------------------
extern "C" int klg_note_record(klg_synth* s, int synth, int pitch, float velocity, void* record, size_t bytes) {
	if (s && s->multi) { int ls = 0; const int i = shard_of_synth(s, synth, &ls); if (i < 0) return fail(KLG_ERR_INVALID, "klg_note_record: synth %d out of range", synth); return klg_note_record(s->multi->shard[(size_t)i], ls, pitch, velocity, record, bytes); }
	if (!s || !record || synth < 0 || synth >= s->S || bytes != (size_t)s->W * 4) return fail(KLG_ERR_INVALID, "klg_note_record: bad arguments (record is %d bytes)", s ? s->W * 4 : 0);
	if (s->graph) return fail(KLG_ERR_INVALID, "klg_note_record: %s", kGraphEvents);
	HostVoice hv = s->voices[(size_t)synth * s->P];                // a scratch note of this instance (the oscillators' cached frequencies start as a fresh note's)
	hv = HostVoice();
	if (s->patch == KLG_PATCH_SUPERSAW) for (auto& o : hv.osm) o = host::OsmH(0.f);
	hv.stage = ST_ONSET; hv.pitch = (float)pitch; hv.velocity = velocity;
	s->record_sink = (uint32_t*)record;
	patch_on(s, synth, synth * s->P, &hv);
	s->record_sink = nullptr;
	return 0;
}

struct klg_script {
	klg_synth* s = nullptr; int blocks = 0; bool committed = false;
	std::vector<std::vector<Event>> ev;                            // per block, in call order
	std::vector<uint32_t> pool;                                    // note-on records, W words each (shared by every block that starts that note)
	struct Slice { size_t first; int R, E; };                      // where block b's run / event arrays sit in d_index
	std::vector<Slice> slices;
	// klg_script_render_device: the launches of a span of blocks captured ONCE as a hipGraph and replayed (same span, block length, destination, stream)
	struct Captured { int first, blocks, n; float* d_out; hipStream_t st; hipGraphExec_t exec; unsigned gen; };   // gen: the bank's args_gen at capture time
	std::vector<Captured> graphs;
	int* d_index = nullptr; uint32_t* d_pool = nullptr;
};
extern "C" klg_script* klg_script_create(klg_synth* s, int blocks) {
	if (s && s->multi) { fail(KLG_ERR_INVALID, "klg_script_create: event scripts are per device: create one bank + script per GPU (one process per GPU, as bench.py does)"); return nullptr; }
	if (!s || blocks <= 0) { fail(KLG_ERR_INVALID, "klg_script_create: bad arguments"); return nullptr; }
	klg_script* k = new klg_script(); k->s = s; k->blocks = blocks; k->ev.resize((size_t)blocks);
	s->scripts.push_back(k);
	return k;
}
// a bank that is destroyed takes its scripts' device arrays with it and leaves them INVALID (k->s == NULL): every later klg_script_* call on
// such a handle fails with KLG_ERR_INVALID instead of touching freed state; klg_script_destroy() still releases the handle itself
static void script_drop_graphs(klg_script* k) { for (auto& g : k->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec); k->graphs.clear(); }
static void scripts_invalidate(klg_synth* s) {
	for (klg_script* k : s->scripts) { script_drop_graphs(k); if (k->d_index) (void)hipFree(k->d_index); if (k->d_pool) (void)hipFree(k->d_pool); k->d_index = nullptr; k->d_pool = nullptr; k->s = nullptr; }
	s->scripts.clear();
}
extern "C" void klg_script_destroy(klg_script* k) {
	if (!k) return;
	if (k->s) {
		DeviceGuard bound(k->s->device);
		script_drop_graphs(k);
		if (k->d_index) (void)hipFree(k->d_index);
		if (k->d_pool) (void)hipFree(k->d_pool);
		auto& v = k->s->scripts; v.erase(std::remove(v.begin(), v.end(), k), v.end());
	}
	delete k;
}
extern "C" int klg_script_add_record(klg_script* k, const void* record, size_t bytes) {
	if (!k || !k->s || !record || k->committed || bytes != (size_t)k->s->W * 4) return fail(KLG_ERR_INVALID, "klg_script_add_record: bad arguments (record is %d bytes), script already committed, or its bank was destroyed", k && k->s ? k->s->W * 4 : 0);
	const uint32_t* w = (const uint32_t*)record;
	k->pool.insert(k->pool.end(), w, w + k->s->W);
	return (int)(k->pool.size() / (size_t)k->s->W) - 1;
}
extern "C" int klg_script_note_on(klg_script* k, int block, int voice, int record_index) {
	if (!k || !k->s || k->committed || block < 0 || block >= k->blocks || voice < 0 || voice >= k->s->V || record_index < 0 || (size_t)record_index >= k->pool.size() / (size_t)k->s->W)
		return fail(KLG_ERR_INVALID, "klg_script_note_on: bad arguments or script already committed");
	k->ev[(size_t)block].push_back({ voice, 0, record_index, (unsigned)k->ev[(size_t)block].size() });
	return 0;
}
extern "C" int klg_script_note_off(klg_script* k, int block, int voice) {
	if (!k || !k->s || k->committed || block < 0 || block >= k->blocks || voice < 0 || voice >= k->s->V) return fail(KLG_ERR_INVALID, "klg_script_note_off: bad arguments, script already committed, or its bank was destroyed");
	if (k->s->graph) return fail(KLG_ERR_INVALID, "klg_script_note_off: %s", kGraphEvents);
	k->ev[(size_t)block].push_back({ voice, 1, -1, (unsigned)k->ev[(size_t)block].size() });
	return 0;
}
// bulk forms (one call per array instead of one per event)
extern "C" int klg_note_records(klg_synth* s, int n, const int* synth, const int* pitch, const float* velocity, void* records) {
	if (!s || n < 0 || !synth || !pitch || !velocity || !records) return fail(KLG_ERR_INVALID, "klg_note_records: bad arguments");
	for (int i = 0; i < n; i++) if (int rc = klg_note_record(s, synth[i], pitch[i], velocity[i], (uint32_t*)records + (size_t)i * s->W, (size_t)s->W * 4)) return rc;
	return 0;
}
extern "C" int klg_script_add_records(klg_script* k, int n, const void* records) {
	if (!k || !k->s || n < 0 || !records || k->committed) return fail(KLG_ERR_INVALID, "klg_script_add_records: bad arguments, script already committed, or its bank was destroyed");
	const int first = (int)(k->pool.size() / (size_t)k->s->W);
	const uint32_t* w = (const uint32_t*)records;
	k->pool.insert(k->pool.end(), w, w + (size_t)n * k->s->W);
	return first;
}
extern "C" int klg_script_note_on_many(klg_script* k, int n, const int* block, const int* voice, const int* record_index) {
	if (!k || n < 0 || !block || !voice || !record_index) return fail(KLG_ERR_INVALID, "klg_script_note_on_many: bad arguments");
	for (int i = 0; i < n; i++) if (int rc = klg_script_note_on(k, block[i], voice[i], record_index[i])) return rc;
	return 0;
}
extern "C" int klg_script_note_off_many(klg_script* k, int n, const int* block, const int* voice) {
	if (!k || n < 0 || !block || !voice) return fail(KLG_ERR_INVALID, "klg_script_note_off_many: bad arguments");
	for (int i = 0; i < n; i++) if (int rc = klg_script_note_off(k, block[i], voice[i])) return rc;
	return 0;
}
// sort every block's events into per-voice runs (what flush_events does per block) and move everything to HBM, once
extern "C" int klg_script_commit(klg_script* k) {
	if (!k || !k->s || k->committed) return fail(KLG_ERR_INVALID, "klg_script_commit: bad handle, already committed, or the script's bank was destroyed");
	RandGuard rg;
	KLG_BIND(k->s);
	std::vector<int> index;
	k->slices.resize((size_t)k->blocks);
	for (int b = 0; b < k->blocks; b++) {
		std::vector<Event>& ev = k->ev[(size_t)b];
		std::stable_sort(ev.begin(), ev.end(), [](const Event& a, const Event& c) { return a.voice != c.voice ? a.voice < c.voice : a.seq < c.seq; });
		std::vector<int> rv, rf, rc;
		for (size_t e = 0; e < ev.size(); e++) {
			if (e == 0 || ev[e].voice != ev[e - 1].voice) { rv.push_back(ev[e].voice); rf.push_back((int)e); rc.push_back(0); }
			rc.back()++;
		}
		k->slices[(size_t)b] = { index.size(), (int)rv.size(), (int)ev.size() };
		index.insert(index.end(), rv.begin(), rv.end()); index.insert(index.end(), rf.begin(), rf.end()); index.insert(index.end(), rc.begin(), rc.end());
		for (const Event& e : ev) index.push_back(e.type);
		for (const Event& e : ev) index.push_back(e.payload);
		std::vector<Event>().swap(ev);
	}
	if (!index.empty()) { HIP_TRY(hipMalloc((void**)&k->d_index, index.size() * 4)); HIP_TRY(hipMemcpy(k->d_index, index.data(), index.size() * 4, hipMemcpyHostToDevice)); }
	if (!k->pool.empty()) { HIP_TRY(hipMalloc((void**)&k->d_pool, k->pool.size() * 4)); HIP_TRY(hipMemcpy(k->d_pool, k->pool.data(), k->pool.size() * 4, hipMemcpyHostToDevice)); }
	std::vector<uint32_t>().swap(k->pool);
	k->committed = true;
	return 0;
}
// replaces: the host's per-block loop "pass this block's MIDI to the synth, then render" (templates/juce/synth/Source/PluginProcessor.cpp:170-177)
// for a stream known in advance: block `block`'s events are applied from HBM (no host work, no transfer), then the block is rendered
extern "C" int klg_script_play_device(klg_script* k, int block, float* d_mix, int n, void* hip_stream) {
	if (!k || !k->s || !k->committed || block < 0 || block >= k->blocks || !d_mix || n <= 0 || n > k->s->max_block) return fail(KLG_ERR_INVALID, "klg_script_play_device: bad arguments, script not committed, or its bank was destroyed");
	KLG_BIND(k->s);
	klg_synth* s = k->s;
	hipStream_t st = hip_stream ? (hipStream_t)hip_stream : s->stream;
	const klg_script::Slice& sl = k->slices[(size_t)block];
	EventArgs a; a.runs = 0;
	if (sl.E > 0) {
		int* d = k->d_index + sl.first;
		a.state = s->d_state; a.stride = s->stride;
		a.run_voice = d; a.run_first = d + sl.R; a.run_count = d + 2 * sl.R; a.runs = sl.R;
		a.ev_type = d + 3 * sl.R; a.ev_payload = d + 3 * sl.R + sl.E; a.payload = k->d_pool;
		a.fs = s->fs.f;
		s->stages_dirty = true; s->scripted = true;
	}
	return enqueue_block(s, d_mix, n, false, st, &a);               // (queued interactive events first, then this block's: as launches, or inside the render launch for small banks)
}

static int script_render(klg_script* k, int first_block, int blocks, float* d_out, int n, void* hip_stream, bool capture_only) {
	if (!k || !k->s || !k->committed || first_block < 0 || blocks < 0 || first_block + blocks > k->blocks || !d_out || n <= 0 || n > k->s->max_block)
		return fail(KLG_ERR_INVALID, "klg_script_render_device: bad arguments, script not committed, or its bank was destroyed");
	KLG_BIND(k->s);
	klg_synth* s = k->s;
	hipStream_t st = hip_stream ? (hipStream_t)hip_stream : s->stream;
	auto enqueue_span = [&]() -> int {
		HIP_TRY(hipMemsetAsync(d_out, 0, (size_t)blocks * 2 * n * sizeof(float), st));
		for (int b = 0; b < blocks; b++) if (int rc = klg_script_play_device(k, first_block + b, d_out + (size_t)b * 2 * n, n, (void*)st)) return rc;
		return 0;
	};
	// A span of a committed script is a fixed sequence of launches (the events come from HBM): captured once as a hipGraph, replayed afterwards — the
	// per-launch submission cost and most of the gap between dependent launches go (what a bank of a few workgroups spends a fifth of its block on).
	// Only when nothing the capture may not contain is pending: queued interactive events / control or table uploads (they synchronise), the
	// per-block host pass of Noise / smooth() banks, kernel timing.  KLG_GRAPH=0: never.
	const char* genv = getenv("KLG_GRAPH");
	const bool prepass = s->graph && (s->graph->noise_calls > 0 || !s->graph->smooths.empty());
	const bool can_graph = !(genv && genv[0] == '0') && blocks >= 2 && !s->timing && s->events.empty() && !s->controls_dirty && !s->tables_dirty && !prepass && s->mix_mode != KLG_MIX_LAST_ACTIVE;
	if (!can_graph) return capture_only ? fail(KLG_ERR_INVALID, "klg_script_capture_span: nothing captured (pending events / uploads, kernel timing on, KLG_GRAPH=0, or a bank whose blocks need a host pass)") : enqueue_span();
	if (!k->graphs.empty() && k->graphs.front().gen != s->args_gen) script_drop_graphs(k);   // the bank's device pointers moved since these were captured (a table upload re-allocated d_tables, the mix mode changed): never replay them
	for (const auto& g : k->graphs) if (g.first == first_block && g.blocks == blocks && g.n == n && g.d_out == d_out && g.st == st) {
		if (capture_only) return 0;
		HIP_TRY(hipGraphLaunch(g.exec, st)); s->stages_dirty = true; s->scripted = true; return 0;
	}
	const bool was_dirty = s->stages_dirty, was_scripted = s->scripted;
	if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); return capture_only ? fail(KLG_ERR_HIP, "klg_script_capture_span: the stream cannot be captured") : enqueue_span(); }
	const int rc = enqueue_span();
	hipGraph_t graph = nullptr;
	const hipError_t ce = hipStreamEndCapture(st, &graph);
	hipGraphExec_t exec = nullptr;
	const bool ok = !rc && ce == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess && exec;
	if (graph) (void)hipGraphDestroy(graph);
	if (!ok) { (void)hipGetLastError(); if (capture_only) { s->stages_dirty = was_dirty; s->scripted = was_scripted; return fail(KLG_ERR_HIP, "klg_script_capture_span: capture failed"); } return rc ? rc : enqueue_span(); }
	if (k->graphs.size() >= 8) script_drop_graphs(k);                // (a handful of spans at most: a render loop replays the same few)
	k->graphs.push_back({ first_block, blocks, n, d_out, st, exec, s->args_gen });
	if (capture_only) { s->stages_dirty = was_dirty; s->scripted = was_scripted; return 0; }      // (nothing has run)
	HIP_TRY(hipGraphLaunch(exec, st));
	return 0;
}
extern "C" int klg_script_render_device(klg_script* k, int first_block, int blocks, float* d_out, int n, void* hip_stream) { return script_render(k, first_block, blocks, d_out, n, hip_stream, false); }
// the capture without the run: a later klg_script_render_device of the same span on the same stream only replays (a host that wants its first timed
// call free of the one-off capture / instantiation cost)
extern "C" int klg_script_capture_span(klg_script* k, int first_block, int blocks, float* d_out, int n, void* hip_stream) { return script_render(k, first_block, blocks, d_out, n, hip_stream, true); }

extern "C" int klg_timing_begin(klg_synth* s) { if (!s) return fail(KLG_ERR_INVALID, "NULL handle");
 if (s->multi) { for (klg_synth* sh : s->multi->shard) klg_timing_begin(sh); return 0; } s->timing = true; s->launches = 0; s->launches_aux = 0; return 0; }
// the block's OTHER kernels since klg_timing_begin (the event kernel, the voice-mix reduce): launches and summed duration; call before klg_timing_end
extern "C" int klg_timing_end_aux(klg_synth* s, int* launches, float* total_ms) {
	if (!s || !launches || !total_ms) return fail(KLG_ERR_INVALID, "klg_timing_end_aux: bad arguments");
	if (s->multi) {                                                     // the shard klg_timing_end reports: the one whose render kernels took longest (they run concurrently)
		int best = 0; float best_ms = -1.f;
		for (size_t i = 0; i < s->multi->shard.size(); i++) {
			klg_synth* sh = s->multi->shard[i]; DeviceGuard bound(sh->device); if (!bound.ok) return KLG_ERR_NO_DEVICE;
			HIP_TRY(hipDeviceSynchronize());
			float total = 0.f;
			for (int k = 0; k < sh->launches; k++) { float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, sh->tev[2 * k], sh->tev[2 * k + 1])); total += ms; }
			if (total > best_ms) { best_ms = total; best = (int)i; }
		}
		return on_shard(s, (size_t)best, [&](klg_synth* sh) { return klg_timing_end_aux(sh, launches, total_ms); });
	}
	KLG_BIND(s);
	HIP_TRY(hipDeviceSynchronize());
	float total = 0.f;
	for (int i = 0; i < s->launches_aux; i++) { float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, s->tev_aux[2 * i], s->tev_aux[2 * i + 1])); total += ms; }
	*launches = s->launches_aux; *total_ms = total;
	return 0;
}
extern "C" int klg_timing_end(klg_synth* s, int* launches, float* total_ms) {
	if (!s || !launches || !total_ms) return fail(KLG_ERR_INVALID, "klg_timing_end: bad arguments");
	if (s->multi) {                                                     // the slowest shard's render time (they run concurrently)
		int l = 0; float ms = 0.f; *launches = 0; *total_ms = 0.f;
		for (size_t i = 0; i < s->multi->shard.size(); i++) { if (int rc = on_shard(s, i, [&](klg_synth* sh) { return klg_timing_end(sh, &l, &ms); })) return rc; if (ms > *total_ms) { *total_ms = ms; *launches = l; } }
		return 0;
	}
	KLG_BIND(s);
	HIP_TRY(hipDeviceSynchronize());
	float total = 0.f;
	for (int i = 0; i < s->launches; i++) { float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, s->tev[2 * i], s->tev[2 * i + 1])); total += ms; }
	*launches = s->launches; *total_ms = total;
	s->timing = false;
	return 0;
}

extern "C" int klg_synth_multi_info(klg_synth* s, int n, int probe_reps, int* shards, int* rccl_ranks, int* distinct_devices, float* per_shard_kernel_ms, int cap, float* allreduce_us) {
	if (!s || !shards || !rccl_ranks || !distinct_devices || !allreduce_us) return fail(KLG_ERR_INVALID, "klg_synth_multi_info: bad arguments");
	*shards = 1; *rccl_ranks = 0; *distinct_devices = 1; *allreduce_us = 0.f;
	auto shard_ms = [&](klg_synth* sh, float* out) -> int {
		DeviceGuard bound(sh->device); if (!bound.ok) return KLG_ERR_NO_DEVICE;
		HIP_TRY(hipDeviceSynchronize());
		float total = 0.f;
		for (int k = 0; k < sh->launches; k++) { float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, sh->tev[2 * k], sh->tev[2 * k + 1])); total += ms; }
		*out = total; return 0;
	};
	if (!s->multi) { if (per_shard_kernel_ms && cap > 0) return shard_ms(s, per_shard_kernel_ms); return 0; }
	Multi& m = *s->multi;
	*shards = (int)m.shard.size();
	{ std::vector<int> seen; for (klg_synth* sh : m.shard) if (std::find(seen.begin(), seen.end(), sh->device) == seen.end()) seen.push_back(sh->device); *distinct_devices = (int)seen.size(); }
	for (int i = 0; per_shard_kernel_ms && i < *shards && i < cap; i++) if (int rc = shard_ms(m.shard[(size_t)i], per_shard_kernel_ms + i)) return rc;
	if (!m.rccl || m.comm.empty()) return 0;
	if (g_rccl.CommCount) { int c = 0; if (g_rccl.CommCount(m.comm[0], &c) == 0) *rccl_ranks = c; }
	if (probe_reps <= 0 || n <= 0 || n > s->max_block) return 0;
	klg_synth* s0 = m.shard[0];
	DeviceGuard bound(s0->device); if (!bound.ok) return KLG_ERR_NO_DEVICE;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
	struct Free { hipEvent_t a, b; ~Free() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } release = { e0, e1 };
	for (klg_synth* sh : m.shard) { DeviceGuard b2(sh->device); HIP_TRY(hipStreamSynchronize(sh->stream)); }
	double total_ms = 0.0;
	for (int r = 0; r < probe_reps + 3; r++) {                                 // (three untimed: the communicator's first collectives set up its channels)
		HIP_TRY(hipEventRecord(e0, s0->stream));
		int rc = g_rccl.GroupStart();
		for (size_t i = 0; i < m.shard.size() && rc == 0; i++) rc = g_rccl.AllReduce(m.shard[i]->d_mix, m.shard[i]->d_mix, (size_t)2 * n, KLG_NCCL_FLOAT32, KLG_NCCL_SUM, m.comm[i], m.shard[i]->stream);
		const int rc2 = g_rccl.GroupEnd();
		if (rc != 0 || rc2 != 0) return fail(KLG_ERR_HIP, "klg_synth_multi_info: ncclAllReduce failed: %s", g_rccl.GetErrorString(rc ? rc : rc2));
		HIP_TRY(hipEventRecord(e1, s0->stream));
		for (klg_synth* sh : m.shard) { DeviceGuard b2(sh->device); HIP_TRY(hipStreamSynchronize(sh->stream)); }
		float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
		if (r >= 3) total_ms += ms;
	}
	*allreduce_us = (float)(1e3 * total_ms / probe_reps);
	return 0;
}

// ------------------------------------------------------------------------------------------------
// effect banks: see klg_fx_api.hpp
// ------------------------------------------------------------------------------------------------
#include "klg_fx_api.hpp"
#include "klg_selftest.hpp"
