// klang_amd/host/klang_deadline.cpp — the real-time deadline, measured from a host that is not an interpreter.
// replaces (as a measurement host): the audio callback of templates/juce/synth/Source/PluginProcessor.cpp:153-182 — one block per callback, the host
// waits for the block before it returns.  V voices of a hand-written patch (default sub2a) all started in block 0 from an HBM-resident script
// (klg_script_*), then `blocks` consecutive blocks of n samples, each: clear the [2][n] mix, klg_process_device, hipStreamSynchronize — timed with
// CLOCK_MONOTONIC around the three; then every voice released and 44 more blocks (all voices in their release ramp: the worst case).  The process
// is pinned to one core and its memory locked; nothing allocates inside the loop.  Prints ONE JSON line: p50 / p99 / max, the index of the worst
// block and the ten largest block times with their indices (so that a spike can be told from a pattern).  Every block is also bracketed by two HIP events on its
// stream: the ten largest come with the time the DEVICE spent between them, so that a late block is either the device's (both large) or the wake-up of the waiting
// host thread's (wall clock large, device time ordinary).  --spin 1 waits by polling (hipDeviceScheduleSpin) instead of sleeping on the interrupt; --rt 1 asks for SCHED_FIFO (and says whether it got it).  The wall clock of a
// block is also split where the last call that queues work returns: a late block's host time is either before that point (queueing) or after it (waiting).
// --paced 1: the loop runs against the audio clock instead of back to back — block k is not started before T0 + k * (n / fs) (the callback's moment), and its lateness is its end
// minus that moment: a host that plays with L blocks of buffering hears a gap when a block's lateness exceeds L block times (a late block delays the ones behind it until the
// loop has caught up).  Reported: the largest lateness and the number of blocks beyond 1, 2 and 3 block times.
// Build: hipcc -O2 -std=c++17 klang_deadline.cpp -I../../include -L.. -lklang_mi355 -Wl,-rpath,'$ORIGIN/..' -o klang_deadline   (klang_amd/csrc/build.sh does)
// Run:   klang_deadline [--voices V] [--blocks B] [--n N] [--patch id] [--notes P] [--cpu c] [--spin 0|1] [--rt 0|1] [--paced 0|1]
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <time.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <klang_mi355.h>

static double now_ms() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec; }
#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)
#define KLG(call) do { if ((call) < 0) DIE("%s failed: %s", #call, klg_last_error()); } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) DIE("%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

int main(int argc, char** argv) {
	long long V = 1 << 20; int blocks = 2000, n = 256, patch = KLG_PATCH_SUB2A, notes = 32, cpu = -1, spin = 0, rt = 0, paced = 0;
	for (int i = 1; i + 1 < argc; i += 2) {
		if (!strcmp(argv[i], "--voices")) V = atoll(argv[i + 1]); else if (!strcmp(argv[i], "--blocks")) blocks = atoi(argv[i + 1]);
		else if (!strcmp(argv[i], "--n")) n = atoi(argv[i + 1]); else if (!strcmp(argv[i], "--patch")) patch = atoi(argv[i + 1]);
		else if (!strcmp(argv[i], "--notes")) notes = atoi(argv[i + 1]); else if (!strcmp(argv[i], "--cpu")) cpu = atoi(argv[i + 1]);
		else if (!strcmp(argv[i], "--spin")) spin = atoi(argv[i + 1]); else if (!strcmp(argv[i], "--rt")) rt = atoi(argv[i + 1]);
		else if (!strcmp(argv[i], "--paced")) paced = atoi(argv[i + 1]);
		else DIE("unknown option %s", argv[i]);
	}
	if (cpu < 0) cpu = sched_getcpu();
	cpu_set_t set; CPU_ZERO(&set); CPU_SET(cpu, &set);
	const bool pinned = sched_setaffinity(0, sizeof set, &set) == 0;
	bool fifo = false; if (rt) { sched_param sp; memset(&sp, 0, sizeof sp); sp.sched_priority = 50; fifo = sched_setscheduler(0, SCHED_FIFO, &sp) == 0; }
	if (spin) HIP(hipSetDeviceFlags(hipDeviceScheduleSpin));                 // (before the first call that makes the context)
	const int base = 1 << 20;                                                // distinct note records (pitch by voice); voices beyond share them
	klg_synth* bank = klg_synth_create(patch, (int)(V / notes), notes, 48000.f, n);
	if (!bank) DIE("klg_synth_create: %s", klg_last_error());
	V = klg_synth_voices(bank);
	const size_t W = klg_synth_state_bytes(bank) / 4;
	klg_script* script = klg_script_create(bank, 2);
	if (!script) DIE("klg_script_create: %s", klg_last_error());
	{
		const int R = (int)std::min<long long>(base, V);
		std::vector<int> sy((size_t)R), pi((size_t)R); std::vector<float> ve((size_t)R, 0.8f); std::vector<uint32_t> rec((size_t)R * W);
		unsigned x = 2463534242u;
		for (int i = 0; i < R; i++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; sy[(size_t)i] = (int)((i / notes) % (V / notes)); pi[(size_t)i] = 36 + (int)(x % 61u); }
		KLG(klg_note_records(bank, R, sy.data(), pi.data(), ve.data(), rec.data()));
		const int first = klg_script_add_records(script, R, rec.data());
		if (first < 0) DIE("klg_script_add_records: %s", klg_last_error());
		std::vector<int> blk((size_t)V, 0), vo((size_t)V), ri((size_t)V);
		for (long long v = 0; v < V; v++) { vo[(size_t)v] = (int)v; ri[(size_t)v] = first + (int)(v % R); }
		KLG(klg_script_note_on_many(script, (int)V, blk.data(), vo.data(), ri.data()));
		std::fill(blk.begin(), blk.end(), 1);
		KLG(klg_script_note_off_many(script, (int)V, blk.data(), vo.data()));
		KLG(klg_script_commit(script));
	}
	float* d_mix = nullptr; hipStream_t st = nullptr;
	HIP(hipMalloc((void**)&d_mix, (size_t)2 * n * sizeof(float)));
	HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	std::vector<double> t((size_t)blocks), r(44); std::vector<float> g((size_t)blocks, 0.f); std::vector<double> q((size_t)blocks, 0.0); double queued_at = 0.0;
	hipEvent_t e0, e1; HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
	const bool locked = mlockall(MCL_CURRENT | MCL_FUTURE) == 0;
	auto block = [&](int script_block) -> int {
		HIP(hipEventRecord(e0, st));
		HIP(hipMemsetAsync(d_mix, 0, (size_t)2 * n * sizeof(float), st));
		if (script_block >= 0) KLG(klg_script_play_device(script, script_block, d_mix, n, st)); else KLG(klg_process_device(bank, d_mix, n, st));
		HIP(hipEventRecord(e1, st));
		queued_at = now_ms();
		HIP(hipStreamSynchronize(st));
		return 0;
	};
	if (block(0)) return 1;
	const double period = 1e3 * n / 48000.0; std::vector<double> lateness((size_t)(paced ? blocks : 0));
	const double T0 = now_ms();
	for (int b = 0; b < blocks; b++) { if (paced) while (now_ms() < T0 + b * period) {}
		const double t0 = now_ms(); if (block(-1)) return 1; t[(size_t)b] = now_ms() - t0; q[(size_t)b] = queued_at - t0; HIP(hipEventElapsedTime(&g[(size_t)b], e0, e1)); if (paced) lateness[(size_t)b] = t0 + t[(size_t)b] - (T0 + b * period); }   // (the clock stops before the events are read)
	if (block(1)) return 1;
	for (int b = 0; b < 44; b++) { const double t0 = now_ms(); if (block(-1)) return 1; r[(size_t)b] = now_ms() - t0; }
	std::vector<float> mix((size_t)2 * n);
	HIP(hipMemcpy(mix.data(), d_mix, mix.size() * sizeof(float), hipMemcpyDeviceToHost));
	bool finite = true; for (float f : mix) finite = finite && f == f && f - f == 0.f;
	std::vector<int> order((size_t)blocks);
	for (int i = 0; i < blocks; i++) order[(size_t)i] = i;
	std::sort(order.begin(), order.end(), [&](int a, int b) { return t[(size_t)a] > t[(size_t)b]; });
	std::vector<double> s = t; std::sort(s.begin(), s.end());
	const double deadline = 1e3 * n / 48000.0, p50 = s[s.size() / 2], p99 = s[std::min(s.size() - 1, (size_t)(0.99 * (double)s.size()))], mx = s.back();
	const double rmax = *std::max_element(r.begin(), r.end());
	printf("{\"host\": \"klang_deadline (C++, pinned to cpu %d: %s, memory locked: %s, waits by %s, %s)\", \"voices\": %lld, \"blocks\": %d, \"n\": %d, \"deadline_ms\": %.4f, \"p50_ms\": %.4f, \"p99_ms\": %.4f, \"max_ms\": %.4f, \"worst_block\": %d, "
	       "\"release_max_ms\": %.4f, \"every_block_within_the_deadline\": %s, \"every_block_within_90_percent_of_the_deadline\": %s, \"finite\": %s, \"ten_largest\": [",
	       cpu, pinned ? "yes" : "no", locked ? "yes" : "no", spin ? "polling" : "the interrupt", !rt ? "SCHED_OTHER" : fifo ? "SCHED_FIFO 50" : "SCHED_FIFO refused", V, blocks, n, deadline, p50, p99, mx, order[0], rmax, std::max(mx, rmax) <= deadline ? "true" : "false", std::max(mx, rmax) <= 0.9 * deadline ? "true" : "false", finite ? "true" : "false");
	for (int i = 0; i < 10 && i < blocks; i++) printf("%s[%d, %.4f]", i ? ", " : "", order[(size_t)i], t[(size_t)order[(size_t)i]]);
	std::vector<float> gs = g; std::sort(gs.begin(), gs.end());
	printf("], \"ten_largest_device_ms\": [");                               // the same ten blocks: what the device spent between the block's two events
	for (int i = 0; i < 10 && i < blocks; i++) printf("%s%.4f", i ? ", " : "", g[(size_t)order[(size_t)i]]);
	printf("], \"ten_largest_queueing_ms\": [");                             // the same ten blocks: host time until the last queueing call returned
	for (int i = 0; i < 10 && i < blocks; i++) printf("%s%.4f", i ? ", " : "", q[(size_t)order[(size_t)i]]);
	int late = 0, late_dev = 0; for (int i = 0; i < blocks; i++) if (t[(size_t)i] > 0.9 * deadline) { late++; if (g[(size_t)i] > 0.9 * deadline) late_dev++; }
	printf("], \"blocks_over_90_percent\": %d, \"of_them_the_devices\": %d", late, late_dev);
	if (paced) { int over[3] = { 0, 0, 0 }; double worst = 0.0; int at = 0;
		for (int i = 0; i < blocks; i++) { for (int L = 1; L <= 3; L++) if (lateness[(size_t)i] > L * period) over[L - 1]++; if (lateness[(size_t)i] > worst) { worst = lateness[(size_t)i]; at = i; } }
		printf(", \"paced\": {\"period_ms\": %.4f, \"largest_lateness_ms\": %.4f, \"at_block\": %d, \"blocks_later_than_1_2_3_periods\": [%d, %d, %d], \"buffering_with_no_gap_in_this_run_blocks\": %d}",
		       period, worst, at, over[0], over[1], over[2], !over[0] ? 1 : !over[1] ? 2 : !over[2] ? 3 : 4); }
	printf(", \"device_p50_ms\": %.4f, \"device_max_ms\": %.4f, \"worst_block_is\": \"%s\"}\n", gs[gs.size() / 2], gs.back(),
	       g[(size_t)order[0]] > 0.8 * t[(size_t)order[0]] ? "the device's (its own time between the block's events is as long)" : "the waiting host thread's (the device finished on time)");
	klg_script_destroy(script); klg_synth_destroy(bank);
	return 0;
}
