// klang_amd/csrc/klg_fx.hpp — effect banks (BASELINE config 4): the shipped PingPong.k and Reverb.k as gfx950 kernels.
//
// One lane = one Stereo::Effect instance; one wave (64 instances) per workgroup so a bank of K instances
// spreads over K/64 CUs.
//
// Data layout in HBM
//   state   [word][Kpad]            per-instance scalars (controls, smoothing, filter state, tap tables), SoA
//   rings   [wave][line][pos][64]   delay lines, tiled per wave of 64 instances and, inside a tile, POSITION-major /
//                                   instance-minor: all instances advance their write cursor in lock-step (it only
//                                   counts samples), so the 64 lanes of a wave write one contiguous 256-byte row per
//                                   line per sample, rows of consecutive positions are adjacent (a wave streams
//                                   through its own contiguous tile: DRAM-page and TLB friendly at 100+ GB of rings),
//                                   and read taps coalesce whenever instances share a delay time (and degrade to a
//                                   gather inside the tile, never to a 768-KB-strided walk, when they do not).
//   io      [K][2][n]               caller layout (per-instance channel buffers, as the reference host owns them);
//                                   staged through a padded LDS tile in 32-sample chunks so global accesses are
//                                   128-byte row segments and the per-sample loop reads/writes LDS only.
//
// Arithmetic order follows examples/PingPong.k / examples/Reverb.k exactly as restated and pinned in
// oracle/... (test infrastructure) — citations are into the reference files.
#pragma once
#include "klg_device.hpp"
#include "klg_kernels.hpp"
#include "klg_device_x2.hpp"     // the width-generic helpers a generated body uses (to_i, kf, f2u, ...)
#include "klg_delay.hpp"

#pragma clang fp contract(off)

namespace klg {

enum { FX_WG = 64, FX_CHUNK = 32, FX_LD = 65 };

struct BiquadCoef { float b0, b1, b2, a1, a2; };

// Stereo::Delay::tap(float) klang.h:4668-4681: both channels read at the LEFT line's cursor, a*(1-frac) + b*frac form
__device__ __forceinline__ void stereo_delay_tap(const Ring& l, const Ring& r, int position, float delay, float& outl, float& outr) {
	float read = (float)(position - 1) - delay;
	if (read < 0.f) read += l.size;
	const float f = (float)floor((double)read);
	const float frac = read - f;
	const int i = (int)read;
	const int j = (i == l.size - 1) ? 0 : (i + 1);
	// `read` within half an ulp below zero rounds to exactly SIZE (klg_delay.hpp, THE PAD ELEMENT).  The reference then reads buffer[SIZE] — the pad, 0 —
	// and buffer[SIZE + 1], which lies beyond the buffer's `size` (inside its power-of-two allocation, never initialised): indeterminate there.
	// Here that tap is 0 (both elements read as zeros), and no neighbouring line is touched.
	const bool pad = i >= l.size;
	const int ic = pad ? 0 : i, jc = pad ? 0 : j;
	const float la = pad ? 0.f : l.rd(ic), lb = pad ? 0.f : l.rd(jc), ra = pad ? 0.f : r.rd(ic), rb = pad ? 0.f : r.rd(jc);
	outl = la * (1.f - frac) + lb * frac;
	outr = ra * (1.f - frac) + rb * frac;
}

// LDS staging of the [K][2][n] io block: tile[ch][sample][FX_LD]
__device__ __forceinline__ void io_load_chunk(float* tile, const float* io, int k0, int K, int n, int s0, int cl, int lane) {
	const int col = lane & 31, half = lane >> 5;
	for (int it0 = 0; it0 < 64; it0 += 16) {             // 128 rows (64 instances x 2 channels), 2 rows per wave access,
		float v[16];                                     // 16 accesses in flight before the first LDS write
#pragma unroll
		for (int j = 0; j < 16; j++) {
			const int row = 2 * (it0 + j) + half, inst = row >> 1, ch = row & 1;
			v[j] = (col < cl && k0 + inst < K) ? io[((size_t)(k0 + inst) * 2 + ch) * n + s0 + col] : 0.f;
		}
#pragma unroll
		for (int j = 0; j < 16; j++) {
			const int row = 2 * (it0 + j) + half, inst = row >> 1, ch = row & 1;
			tile[(ch * FX_CHUNK + col) * FX_LD + inst] = v[j];
		}
	}
}
__device__ __forceinline__ void io_store_chunk(const float* tile, float* io, int k0, int K, int n, int s0, int cl, int lane) {
	const int col = lane & 31, half = lane >> 5;
#pragma unroll 16
	for (int it = 0; it < 64; it++) {
		const int row = 2 * it + half, inst = row >> 1, ch = row & 1;
		if (col < cl && k0 + inst < K) io[((size_t)(k0 + inst) * 2 + ch) * n + s0 + col] = tile[(ch * FX_CHUNK + col) * FX_LD + inst];
	}
}

// =================================================================================================
// PingPong.k
// =================================================================================================
enum { PP_C0 = 0, PP_SM1 = 6, PP_SM5 = 7, PP_DELAY = 8, PP_LFO_POS = 9, PP_LFO_INC = 10, PP_Z = 11, PP_WORDS = 15 };
enum { PP_SIZE = 192000, PP_ROWS = PP_SIZE + 1 };   // Delay<192000>: rows per line in HBM = SIZE + the pad row (klg_delay.hpp)

struct PingPongArgs {
	float* state; size_t kpad; int K;
	float* rings;               // [kpad/64][2][192000][64]
	int position;               // write cursor of both lines at block start (samples processed % 192000)
	float* io; int n;
	// a SPAN of blocks in one launch (klg_fx_render_device): n = blocks * nb samples, the caller's buffer is [blocks][K][2][nb] — block b's rows start block_stride
	// floats after block b - 1's.  prepare() (PingPong.k:37-41) only sets the DC filters, and no dial can move inside a span, so a span IS one long block whose
	// rows are cut into pieces: the pipeline runs across the block boundaries.  nb == 0: one block of n samples (klg_fx_pingpong_x only; nb a multiple of a chunk)
	int nb; size_t block_stride;
	SampleRate fs;
	BiquadCoef dc;              // dcfilter[k].set(50, 1) — PingPong.k:39-40, computed on the host
	float c1_min, c1_max;
	int ablate;                 // measurement only (KLG_FX_ABLATE, the one-wave kernel): 1 = no ring reads, 2 = no ring writes, 4 = no io staging
	// a long span as TWO launches (klg_fx_pingpong_x, round 6): pass 1 = the kernel compiled without the moving-dials pipeline renders the workgroups whose span
	// is stationary (decided on the device, as ever) and says so in done[workgroup]; pass 2 = the full kernel renders the others.  0: one launch renders everything.
	int pass; int* done;
};

// The kernel works in sub-chunks of PP_SUB samples:
//   1. CONTROL PRE-PASS.  Everything that decides WHERE the delay lines are read — the two Control::smooth()
//      one-poles, the scratch detector, the LFO, controls[1].set() — depends on control state only, never on audio.
//      It is run PP_SUB samples ahead and yields the read taps (position, fraction) of both lines per sample.
//   2. PREFETCH.  If every tap of the sub-chunk lies more than PP_SUB + 2 samples behind the write cursor (true for
//      any delay longer than ~0.2 ms) none of this sub-chunk's writes can alias them, so all 6 x PP_SUB ring reads are
//      issued back to back — one exposed HBM latency per sub-chunk instead of three to four per sample.
//      Otherwise (very short delays) the sub-chunk falls back to in-order loads.
//   3. AUDIO PASS.  Per sample: interpolate, cross-feed, write both rings (stores are fire-and-forget), DC filters.
// Arithmetic and its order are identical in both paths (bit-exact vs the reference, tests/test_gpu_fx.py).
enum { PP_SUB = 8 };

#ifdef KLG_AB_KERNELS   // (the one-wave reference of klg_fx_pingpong_x: in the library only when built with -DKLG_AB_KERNELS)
__global__ __launch_bounds__(FX_WG) void klg_fx_pingpong(const PingPongArgs a) {
	__shared__ float tile[2 * FX_CHUNK * FX_LD];
	const int lane = threadIdx.x, k0 = blockIdx.x * FX_WG, k = k0 + lane;
	const int SIZE = 192000;
	float st[PP_WORDS];
#pragma unroll
	for (int w = 0; w < PP_WORDS; w++) st[w] = a.state[(size_t)w * a.kpad + k];
	const float c0 = st[0], c2 = st[2], c3 = st[3], c4 = st[4], c5 = st[5];
	float c1 = st[1];
	float sm1 = st[PP_SM1], sm5 = st[PP_SM5], mdelay = st[PP_DELAY];
	BOsc lfo; lfo.position = st[PP_LFO_POS]; lfo.increment = st[PP_LFO_INC]; lfo.offset = 0.f;
	Biquad dcl = { a.dc.b0, a.dc.b1, a.dc.b2, a.dc.a1, a.dc.a2, st[PP_Z + 0], st[PP_Z + 1] };
	Biquad dcr = { a.dc.b0, a.dc.b1, a.dc.b2, a.dc.a1, a.dc.a2, st[PP_Z + 2], st[PP_Z + 3] };
	float* tile0 = a.rings + (size_t)blockIdx.x * 2 * PP_ROWS * FX_WG + lane;            // this wave's ring tile: [2][SIZE + 1][64] (row SIZE: the pad, klg_delay.hpp)
	Ring left = { tile0, FX_WG, SIZE }, right = { tile0 + (size_t)PP_ROWS * FX_WG, FX_WG, SIZE };
	int position = a.position;
	// loop invariants of process() PingPong.k:44-60 (controls 0, 2, 3, 4, 5 only change between blocks)
	const float rate = (c3 * c3) * 100.f;                                       // sqr(controls[3]) * 100.f
	const float lfo_inc = rate * 2.f * KLG_PI_F / a.fs.f;                       // Oscillator::set(rate)  klang.h:2862-2865
	const float gain = c0, dry = c4;
	const float vibrato = (c2 * c2) * rate * 1.41421354f;                       // sqr(controls[2]) * rate * root2
	const bool any_vibrato = __ballot(vibrato != 0.f) != 0ull;

	for (int s0 = 0; s0 < a.n; s0 += FX_CHUNK) {
		const int cl = (a.n - s0 < FX_CHUNK) ? (a.n - s0) : FX_CHUNK;
		if (!(a.ablate & 4)) io_load_chunk(tile, a.io, k0, a.K, a.n, s0, cl, lane);
		wave_sync();
		for (int u0 = 0; u0 < cl; u0 += PP_SUB) {
			const int ul = (cl - u0 < PP_SUB) ? (cl - u0) : PP_SUB;
			// ---- 1. control pre-pass ----
			Tap tl[PP_SUB], tr[PP_SUB];
			bool far = true;
#pragma unroll
			for (int u = 0; u < PP_SUB; u++) {
				if (u < ul) {
					sm5 = sm5 * 0.999f + (1.f - 0.999f) * c5;                       // controls[5].smooth()  klang.h:1715
					const float new_delay = sm5;
					if ((double)fabsf(mdelay - new_delay) > 0.001) {
						mdelay = new_delay;
						c1 = (new_delay < a.c1_min) ? a.c1_min : (a.c1_max < new_delay) ? a.c1_max : new_delay;   // controls[1].set()
						lfo.position = KLG_PI_F;                                    // lfo.set(rate, pi)
					}
					else mdelay = c5;
					lfo.increment = lfo_inc;
					sm1 = sm1 * 0.999f + (1.f - 0.999f) * c1;                       // controls[1].smooth()
					const float delay = sm1;
					float lfo_out = 0.f;
					if (any_vibrato) lfo_out = basic_sine(lfo);                     // fp64 sin only when some instance uses it;
					else phase_advance(lfo.position, lfo.increment);                // x * 0 * 5e-5 == +-0 and c1 + (+-0) == c1 exactly
					const float nc1 = c1 + lfo_out * vibrato * 0.00005f;
					c1 = (nc1 < a.c1_min) ? a.c1_min : (a.c1_max < nc1) ? a.c1_max : nc1;
					const int pos = (position + u >= SIZE) ? position + u - SIZE : position + u;
					const float dl = delay * a.fs.f, dr = 0.5f * delay * a.fs.f;
					tl[u] = delay_set(pos, SIZE, dl);                               // left.set(delay * fs)
					tr[u] = delay_set(pos, SIZE, dr);                               // right.set(0.5f * delay * fs)
					far = far && dr >= (float)(PP_SUB + 3) && dl <= (float)(SIZE - PP_SUB - 4);
				}
			}
			// ---- 2. prefetch ----
			float pl[PP_SUB][3], pr[PP_SUB][3];
			const bool prefetch = __ballot(k < a.K && !far) == 0ull;           // padding lanes (zero state, zero delay) do not veto
			if (prefetch) {
#pragma unroll
				for (int u = 0; u < PP_SUB; u++) {
					if (u < ul) {
						const int i0 = tl[u].position, i1 = ring_succ(i0, SIZE), i2 = ring_succ(i1, SIZE);
						const int j0 = tr[u].position, j1 = ring_succ(j0, SIZE), j2 = ring_succ(j1, SIZE);
						if (a.ablate & 1) { pl[u][0] = pl[u][1] = pl[u][2] = 0.f; pr[u][0] = pr[u][1] = pr[u][2] = 0.f; }
						else {
							pl[u][0] = left.rd(i0); pl[u][1] = left.rd(i1); pl[u][2] = left.rd(i2);
							pr[u][0] = right.rd(j0); pr[u][1] = right.rd(j1); pr[u][2] = right.rd(j2);
						}
					}
				}
			}
			// ---- 3. audio pass ----
#pragma unroll
			for (int u = 0; u < PP_SUB; u++) {
				if (u < ul) {
					const int s = u0 + u;
					const float in_l = tile[(0 * FX_CHUNK + s) * FX_LD + lane], in_r = tile[(1 * FX_CHUNK + s) * FX_LD + lane];
					float r1, l1, l2, r2;
					// dry * in.l + (1.f - dry) * ((in.l + right * gain) >> left) >> out.l;   PingPong.k:66
					if (prefetch) r1 = pr[u][0] + tr[u].fraction * (pr[u][1] - pr[u][0]);
					else r1 = delay_process(right, tr[u]);
					if (!(a.ablate & 2)) left.wr(position, in_l + r1 * gain);
					if (prefetch) {
						l1 = pl[u][0] + tl[u].fraction * (pl[u][1] - pl[u][0]);
						l2 = pl[u][1] + tl[u].fraction * (pl[u][2] - pl[u][1]);
					}
					else { l1 = delay_process(left, tl[u]); l2 = delay_process(left, tl[u]); }
					float out_l = dry * in_l + l1 * (1.f - dry);
					// dry * in.r + (1.f - dry) * ((in.r + left * gain) >> right) >> out.r;   PingPong.k:67
					if (!(a.ablate & 2)) right.wr(position, in_r + l2 * gain);
					if (prefetch) r2 = pr[u][1] + tr[u].fraction * (pr[u][2] - pr[u][1]);
					else r2 = delay_process(right, tr[u]);
					float out_r = dry * in_r + r2 * (1.f - dry);
					position = (position + 1 == SIZE) ? 0 : position + 1;
					out_l = biquad_process(dcl, out_l);                             // out.l >> dcfilter[0] >> out.l
					out_r = biquad_process(dcr, out_r);
					tile[(0 * FX_CHUNK + s) * FX_LD + lane] = out_l;
					tile[(1 * FX_CHUNK + s) * FX_LD + lane] = out_r;
				}
			}
		}
		wave_sync();
		if (!(a.ablate & 4)) io_store_chunk(tile, a.io, k0, a.K, a.n, s0, cl, lane);
		wave_sync();
	}
	if (k < a.K) {
		a.state[(size_t)1 * a.kpad + k] = c1;
		a.state[(size_t)PP_SM1 * a.kpad + k] = sm1;
		a.state[(size_t)PP_SM5 * a.kpad + k] = sm5;
		a.state[(size_t)PP_DELAY * a.kpad + k] = mdelay;
		a.state[(size_t)PP_LFO_POS * a.kpad + k] = lfo.position;
		a.state[(size_t)PP_LFO_INC * a.kpad + k] = lfo.increment;
		a.state[(size_t)(PP_Z + 0) * a.kpad + k] = dcl.z0; a.state[(size_t)(PP_Z + 1) * a.kpad + k] = dcl.z1;
		a.state[(size_t)(PP_Z + 2) * a.kpad + k] = dcr.z0; a.state[(size_t)(PP_Z + 3) * a.kpad + k] = dcr.z1;
	}
}

#endif

template<bool B> struct BoolTag { static constexpr bool value = B; };   // (std::true_type without <type_traits>: this header is also compiled by hipRTC)
template<int I> struct IntTag { static constexpr int value = I; };
// -------------------------------------------------------------------------------------------------
// PingPong.k, twelve waves per 64 instances (the production kernel).
//
// Once every tap of a chunk lies further behind the write cursor than the chunk is long, the samples of the chunk no
// longer depend on each other through the delay lines: only three short recurrences are sequential in time — the control
// smoothing, and the two DC filters.  The kernel therefore cuts the block into chunks of PPX_CHUNK samples and runs
// them through a three-stage pipeline, one stage per group of waves, one __syncthreads() per chunk:
//   wave 0       CONTROL  of chunk j+1: Control::smooth x2, scratch detector, LFO, controls[1].set() -> delay time per sample
//   waves 1..8   AUDIO    of chunk j:   wave w owns 4 consecutive samples: Delay::set, 24 ring rows in flight, interpolate,
//                                       cross-feed, both ring writes; also fetches the io rows of chunk j+1
//   waves 9, 10  FILTER   of chunk j-1: out.l / out.r >> dcfilter over the chunk; the io rows of chunk j-2 are stored
// A chunk with a near tap (delay < ~0.8 ms, or within a chunk of the full line) is walked in order by wave 1 alone.
//
// REQUEST-AHEAD (round 3).  In the pipeline above the audio stage asks for its ring rows and waits for them inside its step: a memory round
// trip — a microsecond and more — in every one of a block's ten steps, on a chip that a bank of 4,096 instances does not load enough to hide
// it (one workgroup per CU).  Requesting a step earlier changes nothing (measured): a step's work is a quarter of the round trip.  So when the
// whole block is known up front — the controls are STATIONARY (below), every tap lies more than P + 1 chunks behind the write cursor, the
// block is whole chunks — the audio waves run P chunks ahead of themselves: in step j they use the rows of chunk j (requested in step j - P),
// then request the rows of chunk j + P and the caller's rows of chunk j + P + 1; nothing requested in a step is waited for in it (the barrier
// waits for LDS only, and the requests come back in the order they are used in).  P per workgroup width: what the registers hold.
// Arithmetic and its order are those of klg_fx_pingpong (KLG_FX_PINGPONG1=1) and the reference, bit for bit.
#ifndef KLG_PPX_VARIANT
#define KLG_PPX_VARIANT 0         // measurement builds only: 1 the filter waves store their chunk (one step fewer), 2 the first requests wait for the decision, 4 workgroups take their instances in blockIdx order, 8 vibrato: no sines ahead
#endif
#ifndef KLG_PPX_ABLATE
#define KLG_PPX_ABLATE 0          // measurement builds only (tools/ppx_ablate.sh): 1 no DC filter chain, 2 no output stores, 4 no ring requests, 8 no audio stage, 16 / 64 moving dials: no first / second half of the control chain, 32 no LFO phase walk in the second
#endif
#ifndef KLG_PPX_MOVING_MIN
#define KLG_PPX_MOVING_MIN 32
#endif
enum { PPX_MOVING_MIN = KLG_PPX_MOVING_MIN };       // the shortest span (chunks) that runs the request-ahead pipeline with moving dials
enum { PPX_CHUNK = 32, PPX_AUDIO = 8, PPX_PER = PPX_CHUNK / PPX_AUDIO, PPX_WAVES = 1 + PPX_AUDIO + 2 + 1, PPX_THREADS = PPX_WAVES * 64 };

template<int G> struct PpxLds {
	float tile[4][2][PPX_CHUNK][G + 1];         // [chunk & 3][channel][sample][instance]: loaded, audio, filter, store
	float C1[2][PPX_CHUNK][G];                  // moving dials, running ahead: controls[1] per sample, from the first control wave to the second, [chunk & 1]
	int lastu[2][G];                            // ... and the chunk's last sample at which the scratch detector fired (-1: none)
	float D[G <= 32 ? 8 : 4][PPX_CHUNK][G];     // delay time (smoothed controls[1]) per sample, [chunk & 1]; the request-ahead pipeline with MOVING dials: a ring of P + 2 <= ND chunks, [chunk & (ND - 1)] (G = 64: P = 2, four buffers — what its LDS has room for)
	int far[2];                                 // 1: every tap of the chunk is far from the write cursor
	int deep;                                   // the block runs the request-ahead pipeline (see PPX_DEEP below): 1 with stationary dials, 2 with moving ones (the control chain P + 1 chunks ahead)
	int vibfast;                                // vibrato: the LFO's sines are taken by the audio waves, two chunks ahead of the chain
	float ph[2][PPX_CHUNK + 1][G], sn[2][PPX_CHUNK][G];   // [chunk & 1]: the LFO's phase at every sample of the chunk (row PPX_CHUNK: after it), and their sines
	// a chunk with a NEAR tap (G <= 32; G = 64 has no room and keeps the walk through memory): the six ring values of every sample as they stood before the
	// chunk, and what the chunk itself has written so far — the walk in sample order then waits for LDS, not for memory
	float nv[G <= 32 ? PPX_CHUNK : 1][6][G], nw[2][G <= 32 ? PPX_CHUNK : 1][G];
};

// G = instances per workgroup: 64 (a whole ring group: banks that fill the chip on their own), or 32 / 16 — a half / a quarter of a
// ring group per workgroup.  The pipeline's time is the instruction count of a chunk on the ONE CU a workgroup runs on (DESIGN.md §3:
// ~7.7 k instructions per 32-sample chunk at G = 64, most of them the audio stage's), and a bank of 4,096 instances is 64 such
// workgroups on a chip of 256 CUs.  With G = 16 an audio wave's 64 lanes are 4 samples x 16 instances — its four samples side by
// side instead of one after the other — so the audio stage is a quarter of the instructions and the bank covers every CU; the control
// and filter waves run their recurrences on 16 lanes (a lane costs nothing, an instruction does).  The ring layout is the same.
// MODE (round 6): three compilations of the one kernel.  The per-sample delay-time arithmetic of the moving-dials steps — and the general loop's near-tap / vibrato
// forms — share the register allocation of everything around them: the full kernel keeps 36 / 160 / 16 bytes of scratch at G = 16 / 32 / 64.
//   PPX_FULL        everything (a long span whose dials may move)
//   PPX_NO_MOVING   without the moving-dials pipeline, which only spans of PPX_MOVING_MIN chunks and more ever run: what a shorter launch — a real-time host's one
//                   block per call — takes; the same paths otherwise
//   PPX_STATIONARY  the request-ahead pipeline with dials at rest and nothing else: the FIRST of a long span's two launches (PingPongArgs::pass).  A workgroup whose
//                   span is not stationary — decided on the device, on the dials as they are — leaves it to the second launch (the full kernel)
// Same bits whichever compilation renders a block (tests/test_gpu_fx_spans.py).
enum { PPX_FULL = 0, PPX_NO_MOVING = 1, PPX_STATIONARY = 2 };
template<int G, int MODE>
__global__ __launch_bounds__(PPX_THREADS) void klg_fx_pingpong_x(const PingPongArgs a) {
	static_assert(G == 64 || G == 32 || G == 16, "instances per workgroup");
	constexpr bool MV = MODE == PPX_FULL, DEEP_ONLY = MODE == PPX_STATIONARY;
	constexpr int SPW = 64 / G, PASSES = PPX_PER / SPW;                           // samples a wave holds side by side; passes over its PPX_PER samples
	__shared__ PpxLds<G> S;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int li = lane & (G - 1), lq = lane / G;                                  // this lane's instance of the workgroup; its sample slot in an audio pass
	// Which instances this workgroup takes.  The delay lines are rows of 64 instances (256 bytes); at G = 16 / 32 a workgroup reads and writes a quarter / half of
	// every row, and a 128-byte line of the memory system then belongs to TWO workgroups.  Dealt out in blockIdx order the parts of a ring group land on different
	// XCDs (workgroups go to the eight XCDs in turn), each with its own L2, and every XCD fetches whole lines for the half it uses: 42.1 MB moved per block against
	// 33.6 MB algorithmic at 4,096 instances (`FETCH_SIZE`, profiles/r04_pmc/pmc_pingpong_4096_spans64.json: exactly the ring reads counted twice).  So the 64 / G
	// parts of a group go to workgroups w, w + 8, w + 16, ... — the same XCD: the second reader of a line finds it in that XCD's L2.
	constexpr int PARTS = 64 / G;
	const int nwg = (int)(a.kpad / G), wg = (int)blockIdx.x;
	const bool remap = PARTS > 1 && (nwg % (8 * PARTS)) == 0 && !(KLG_PPX_VARIANT & 4);
	const int k0 = remap ? (((wg / (8 * PARTS)) * 8 + (wg % 8)) * PARTS + (wg / 8) % PARTS) * G : wg * G, k = k0 + li;
	if (a.pass == 2 && a.done[wg]) return;                                         // rendered by the first launch
	const int SIZE = 192000, n = a.n;
	const int nb = a.nb > 0 ? a.nb : n;                                            // the length of a row of the caller's buffer (a span: one block's)
	// where the caller's rows of this workgroup's first instance stand for sample s of the span (s a multiple of the chunk: a chunk never straddles two blocks)
	auto io_rows = [&](const int s) { const int b = s / nb; return (char*)(a.io + (size_t)b * a.block_stride + (size_t)k0 * 2 * nb + (s - b * nb)); };
	const int nchunks = (n + PPX_CHUNK - 1) / PPX_CHUNK;
	const bool w_control = wv == 0, w_audio = wv >= 1 && wv <= PPX_AUDIO, w_filter = wv > PPX_AUDIO && wv <= PPX_AUDIO + 2;
	const bool w_control2 = MV && wv == PPX_AUDIO + 3;                                   // the second half of the control chain when the dials move and the pipeline runs ahead (see `moving`)
	// the control and filter waves are dependent chains that pace the pipeline; the audio waves share their SIMDs and mostly wait for
	// memory: when both are ready, the chain issues first
	if (!w_audio) __builtin_amdgcn_s_setprio(3);
	const float* W = a.state + k;
#define PPW(w) W[(size_t)(w) * a.kpad]
	// ---- control wave state (PingPong.k:44-60) ----
	float c1 = 0.f, c5 = 0.f, sm1 = 0.f, sm5 = 0.f, mdelay = 0.f, lfo_inc = 0.f, vibrato = 0.f;
	BOsc lfo; lfo.position = 0.f; lfo.increment = 0.f; lfo.offset = 0.f;
	bool any_vibrato = false;
	if (w_control) {
		const float c2 = PPW(2), c3 = PPW(3);
		c1 = PPW(1); c5 = PPW(5); sm1 = PPW(PP_SM1); sm5 = PPW(PP_SM5); mdelay = PPW(PP_DELAY);
		lfo.position = PPW(PP_LFO_POS); lfo.increment = PPW(PP_LFO_INC);
		const float rate = (c3 * c3) * 100.f;                                      // sqr(controls[3]) * 100.f
		lfo_inc = rate * 2.f * KLG_PI_F / a.fs.f;                                  // Oscillator::set(rate)  klang.h:2862-2865
		vibrato = (c2 * c2) * rate * 1.41421354f;                                  // sqr(controls[2]) * rate * root2
		any_vibrato = __ballot(vibrato != 0.f) != 0ull;
	}
	if (w_control2) {
		const float c3 = PPW(3);
		sm1 = PPW(PP_SM1); lfo.position = PPW(PP_LFO_POS);
		lfo_inc = ((c3 * c3) * 100.f) * 2.f * KLG_PI_F / a.fs.f;
	}
	bool moving = false, rest5_all = false;                                        // (set where the block's pipeline is chosen)
	// STATIONARY controls: once both smoothers sit at their fp32 fixed points (x * 0.999f + (1.f - 0.999f) * v == x: some ten thousand samples
	// after a dial last moved), no scratch is detected and there is no vibrato, every sample of the block leaves sm5, mdelay (= controls[5]),
	// controls[1] and sm1 exactly as it found them: the delay time of every sample IS sm1.  The serial chain that otherwise paces the pipeline
	// (~16 dependent operations per sample on one wave: 2.3 k of a step's 4.0 k cycles) is then not run at all — the control wave fills both
	// delay-time buffers once and only walks the LFO's phase (which nothing waits for).  Checked per block, wave-uniform, on the values themselves.
	bool stationary = false;
	if (w_control && !any_vibrato) {
		const float k5 = (1.f - 0.999f) * c5;
		const bool fixed = (sm5 * 0.999f + k5 == sm5) && !(fabsf(mdelay - sm5) >= 0.001f) && !(fabsf(c5 - sm5) >= 0.001f) && (sm1 * 0.999f + (1.f - 0.999f) * c1 == sm1);
		stationary = __ballot(!fixed) == 0ull;
	}
	// ---- vibrato: the LFO's phases walked ahead (control wave), their sines (audio waves); see the control stage ----
	float spec_pos = 0.f; int plain_chunk = -1;
	auto prescan = [&](const int chunk) {
		const bool inc_ok = !(lfo_inc >= KLG_TWO_PI);                                  // Phase::operator+= klang.h:1518-1525
		float (*PH)[G] = S.ph[chunk & 1];
		float pos = spec_pos;
		for (int u = 0; u < PPX_CHUNK; u++) {
			PH[u][li] = pos;                                                            // (the lanes of an instance hold the same state: they store the same value)
			const float p1 = pos + lfo_inc, p2 = (p1 > KLG_TWO_PI) ? p1 - KLG_TWO_PI : p1;
			pos = inc_ok ? p2 : pos;
		}
		PH[PPX_CHUNK][li] = pos;
		spec_pos = pos;
	};
	auto sines = [&](const int chunk) {                                             // Basic::Sine klang.h:4902 on every phase of the chunk: 32 x G values over the 512 audio lanes
		const float (*PH)[G] = S.ph[chunk & 1]; float (*SN)[G] = S.sn[chunk & 1];
		constexpr int PER = PPX_CHUNK * G / (PPX_AUDIO * 64);
		const int at = tid - 64;
		float y[PER];
#pragma unroll
		for (int i = 0; i < PER; i++) { const int idx = at + i * PPX_AUDIO * 64; y[i] = (float)sin_f64_core((double)(PH[idx / G][idx % G] + 0.f)); }   // (position + offset; the control wave's LFO has none.  Phases are in [0, 2 pi]: no range test)
#pragma unroll
		for (int i = 0; i < PER; i++) { const int idx = at + i * PPX_AUDIO * 64; SN[idx / G][idx % G] = y[i]; }
	};
	// ---- audio wave state ----
	float gain = 0.f, dry = 0.f;
	if (w_audio) { gain = PPW(0); dry = PPW(4); }
	float* ring0 = a.rings + (size_t)(k0 / FX_WG) * 2 * PP_ROWS * FX_WG;           // this workgroup's ring tile: [2][SIZE + 1][64] (G < 64: its 64-instance group's; row SIZE: the pad, klg_delay.hpp)
	const unsigned lane4 = (unsigned)((k0 & (FX_WG - 1)) + li) * 4u, ROW = FX_WG * 4u;
	auto ring_rd = [&](int line, int i) { return *(const float*)((const char*)(ring0 + (size_t)line * PP_ROWS * FX_WG) + ((unsigned)i * ROW + lane4)); };
	auto ring_wr = [&](int line, int i, float v) { *(float*)((char*)(ring0 + (size_t)line * PP_ROWS * FX_WG) + ((unsigned)i * ROW + lane4)) = v; };
	auto wrap = [&](int i) { return i >= SIZE ? i - SIZE : i; };
	// ---- filter wave state: dcfilter[ch].set(50, 1)  PingPong.k:39-40 ----
	const int fch = wv - (PPX_AUDIO + 1);
	Biquad dc = { a.dc.b0, a.dc.b1, a.dc.b2, a.dc.a1, a.dc.a2, 0.f, 0.f };
	if (w_filter) { dc.z0 = PPW(PP_Z + 2 * fch); dc.z1 = PPW(PP_Z + 2 * fch + 1); }

	// FILTER of chunk jf, then its store (the filter wave's own rows: wave-level ordering is enough)
	auto filter_stage = [&](const int jf, auto store_c) __attribute__((always_inline)) {
		const int s0 = jf * PPX_CHUNK, cl = (n - s0 < PPX_CHUNK) ? (n - s0) : PPX_CHUNK;
		float (*T)[G + 1] = S.tile[jf & 3][fch];
		auto filter_chunk = [&](auto full_c) {                                    // (FULL: see far_chunk — a bounds test per sample is a join per sample)
			constexpr bool FULL = decltype(full_c)::value;
			for (int b = 0; b < PPX_CHUNK; b += 8) {                              // eight LDS reads in flight, then the (sequential) filter
				float x[8];
#pragma unroll
				for (int u = 0; u < 8; u++) x[u] = T[b + u][li];
				// (packed fp32 operations for the independent products were tried — 5.5 instead of 9 VALU instructions per sample, bit-identical —
				//  and were 2.3 x SLOWER here when measured (round 3).  Not by the packed operations' own latency — a dependent v_pk_mul / add / fma_f32 waits 8 cycles like a
				//  plain one, tools/calib/issue_latency.hip —: what builds and splits the pairs are instructions too, and this wave pays an issue slot for each)
#pragma unroll
				for (int u = 0; u < 8; u++) if ((FULL || b + u < cl) && !(KLG_PPX_ABLATE & 1)) x[u] = biquad_process(dc, x[u]);     // out >> dcfilter[ch] >> out
				if (lane < G) {
#pragma unroll
					for (int u = 0; u < 8; u++) T[b + u][li] = x[u];
				}
			}
		};
		if (cl == PPX_CHUNK) filter_chunk(BoolTag<true>{}); else filter_chunk(BoolTag<false>{});
		if constexpr (!decltype(store_c)::value) return;                          // (the request-ahead pipeline: the audio waves store the chunk a step later)
		wave_sync();
		int sl = lane; asm volatile("" : "+v"(sl));
		const int col = sl & 31, half = sl >> 5;
		char* dst = io_rows(s0);
		if (KLG_PPX_ABLATE & 2) {}
		else if (cl == PPX_CHUNK && k0 + G <= a.K) {
#pragma unroll 8
			for (int it = 0; it < G / 2; it++) { const int inst = 2 * it + half; *(float*)(dst + (unsigned)((inst * 2 + fch) * nb + col) * 4u) = T[col][inst]; }
		}
		else {
#pragma unroll 8
			for (int it = 0; it < G / 2; it++) {
				const int inst = 2 * it + half;
				if (col < cl && k0 + inst < a.K) *(float*)(dst + (unsigned)((inst * 2 + fch) * nb + col) * 4u) = T[col][inst];
			}
		}
	};

	// ---------------- the request-ahead pipeline (header comment) ----------------
#ifndef KLG_PPX_DEEP16_LONG
#define KLG_PPX_DEEP16_LONG 4
#endif
#ifndef KLG_PPX_DEEP16_SHORT
#define KLG_PPX_DEEP16_SHORT 2          // (measured, 4,096 instances, one block per call: 14.0 / 13.4 / 12.9 us with a lead of 4 / 3 / 2 chunks)
#endif
	// (the compilation that short launches take — a real-time host's one block per call: 8 chunks — may run less far ahead: every chunk of lead is a step the block spends filling and draining the pipeline)
	constexpr int PPX_DEEP = G == 16 ? (MODE == PPX_NO_MOVING ? KLG_PPX_DEEP16_SHORT : KLG_PPX_DEEP16_LONG) : G == 32 ? 3 : 2;                   // chunks an audio wave runs ahead of itself
	constexpr int P = PPX_DEEP;
	constexpr int DIOV = G * 2 * PPX_CHUNK / (PPX_AUDIO * 64);                  // values of the caller's chunk per audio thread
	float iov[P][DIOV];                                                         // the caller's rows of a chunk, requested P steps before they go to LDS
	float rwl[P][PASSES][3], rwr[P][PASSES][3];                                 // a chunk's ring rows (left / right line), requested P steps before they are used
	const bool whole_chunks = (n % PPX_CHUNK) == 0, whole_group = k0 + G <= a.K;
	const float dly = (w_audio && whole_chunks) ? PPW(PP_SM1) : 0.f;            // this instance's delay time: every sample's, if the block turns out stationary
		// the audio waves' part of step j; SLOT = j mod P (compile-time: the register arrays are indexed by constants only); nch = the block's chunks
		// STEADY (a span's inner steps, 2 <= j <= nch - P - 2: every chunk the step touches exists): no test of j at all — through run-time guards every part of
		// a step ends in a join, and behind a join the compiler waits for everything that is under way
		// MOVING: the delay time of every sample comes from the control chain's buffers (S.D[chunk & 7]) instead of being the one stationary value
		// Where step j stands, kept from step to step instead of being derived from j (round 6): a step's two write-cursor positions were 64-bit remainders and its
		// two row addresses divisions by the block length — on the scalar unit, but ~100 of the ~235 instructions an audio wave issues per step, and a wave issues its
		// instructions one after the other.  Steps come in order (first_requests starts at j = -P - 1, every later step is the next): pos_j = the write cursor at chunk
		// j; st_* / ld_*: the caller's rows of chunk j - 2 (to be stored) and of chunk j + P + 1 (to be requested): pointer, samples into the block, chunk number.
		int pos_j = 0, st_off = 0, st_c = 0, ld_off = 0;
		char* st_ptr = nullptr; const char* ld_ptr = nullptr;
		auto cursors_begin = [&]() __attribute__((always_inline)) {
			const int j0 = -P - 1;
			int p0 = (int)(((long long)a.position + (long long)j0 * PPX_CHUNK) % SIZE);
			pos_j = p0 < 0 ? p0 + SIZE : p0;
			st_c = j0 - 2; st_off = 0; st_ptr = io_rows(0);                 // (chunks before 0 do not exist: the cursor waits at chunk 0 until its step comes)
			ld_off = 0; ld_ptr = io_rows(0);                                // chunk j0 + P + 1 = 0
		};
		auto cursors_next = [&]() __attribute__((always_inline)) {
			pos_j += PPX_CHUNK; pos_j = pos_j >= SIZE ? pos_j - SIZE : pos_j;
			const long long skip = ((long long)a.block_stride - nb) * 4;    // from the end of a block's row to the same row of the next block
			if (st_c >= 0) { st_off += PPX_CHUNK; st_ptr += PPX_CHUNK * 4; if (st_off == nb) { st_off = 0; st_ptr += skip; } }
			st_c++;
			ld_off += PPX_CHUNK; ld_ptr += PPX_CHUNK * 4; if (ld_off == nb) { ld_off = 0; ld_ptr += skip; }
		};
		// (steady_c: false = guarded, true = steady, IntTag<2> = steady AND the workgroup's instances all exist — no access is predicated: what the compiler needs to
		//  count the requests under way instead of waiting for all of them, `s_waitcnt vmcnt(0)`, once per step)
		auto audio_part = [&](auto slot_c, auto steady_c, auto moving_c, const int j, const int nch) __attribute__((always_inline)) {
			constexpr int RS = decltype(slot_c)::value, IS = (RS + 1) % P;
			constexpr int SV = (int)decltype(steady_c)::value;             // 0 guarded, 1 steady, 2 steady + whole group, 3 guarded + whole group
			constexpr bool ST = SV == 1 || SV == 2, MOVING = decltype(moving_c)::value;
			constexpr bool WHOLE = SV == 2 || SV == 3;
			const bool whole_group = WHOLE || k0 + G <= a.K;
			constexpr int ND = G <= 32 ? 8 : 4;
			int at = tid - 64; asm volatile("" : "+v"(at));                    // (see the general loop: keeps per-thread addresses out of loop-invariant registers)
			// The caller's rows move as VECTORS: a thread takes VW consecutive samples of a row (a 32-sample row of the chunk is 128 contiguous bytes): 4 / 2 x 4 samples
			// at G = 32 / 64 — a chunk's 2 G rows are 8 / 16 wave-instructions each way instead of 32 / 64.  A step is paced by how many vector-memory
			// instructions the CU's one address path takes as much as by their bytes: 8,192 instances 18.0 -> 15.8 us per block in 64-block spans, 65,536: 99.8 -> 94
			// (G = 16: two samples a thread changed nothing in spans; four samples on half the threads: not faster in spans, slower in single blocks
			//  — and the two-sample form costs the <16> kernel's OTHER loops 7 % through its register allocation (the vibrato leg 19.9 -> 21.5 us): G = 16 keeps one sample a thread)
			constexpr int VW = DIOV < 4 ? 1 : 4, NV = DIOV / VW, TPR = PPX_CHUNK / VW, RPP = PPX_AUDIO * 64 / TPR;   // samples per vector, vectors per thread, threads per row, rows per pass
			typedef float fvec __attribute__((ext_vector_type(VW), aligned(4)));
			const int vcol = (at % TPR) * VW, vrow = at / TPR;
			const int jn = j + 1, js = j - 2;
			if ((ST || (js >= 0 && js < nch)) && !(KLG_PPX_ABLATE & 2) && !(KLG_PPX_VARIANT & 1)) {       // the caller's rows of chunk j - 2 (filtered in the step before): to memory
				char* dst = st_ptr;
#pragma unroll
				for (int i = 0; i < NV; i++) {
					const int row = vrow + RPP * i;
					fvec v;
#pragma unroll
					for (int q = 0; q < VW; q++) v[q] = S.tile[js & 3][row & 1][vcol + q][row >> 1];
					if (whole_group || k0 + (row >> 1) < a.K) *(fvec*)(dst + (unsigned)(row * nb + vcol) * 4u) = v;
				}
			}
			if (ST || (jn >= 0 && jn < nch)) {                              // the caller's rows of chunk j + 1 (requested in step j - P): into LDS
#pragma unroll
				for (int i = 0; i < NV; i++) {
					const int row = vrow + RPP * i;
#pragma unroll
					for (int q = 0; q < VW; q++) S.tile[jn & 3][row & 1][vcol + q][row >> 1] = iov[IS][i * VW + q];
				}
			}
			const int u0 = (wv - 1) * PPX_PER + lq;
			if ((ST || (j >= 0 && j < nch)) && !(KLG_PPX_ABLATE & 8)) {                              // AUDIO of chunk j: its rows were requested in step j - P
				const int pos0 = pos_j;
				float (*T)[PPX_CHUNK][G + 1] = S.tile[j & 3];
#pragma unroll
				for (int q = 0; q < PASSES; q++) {
					const int u = u0 + q * SPW, pos = wrap(pos0 + u);
					const float in_l = T[0][u][li], in_r = T[1][u][li];
					const float dl = MOVING ? S.D[j & (ND - 1)][u][li] : dly;
					const float fl = delay_set(pos, SIZE, dl * a.fs.f).fraction, fr = delay_set(pos, SIZE, 0.5f * dl * a.fs.f).fraction;   // (as at the request: a dozen operations, not registers held for P steps)
					// dry * in.l + (1.f - dry) * ((in.l + right * gain) >> left) >> out.l;   PingPong.k:66
					const float r1 = rwr[RS][q][0] + fr * (rwr[RS][q][1] - rwr[RS][q][0]);
					ring_wr(0, pos, in_l + r1 * gain);
					const float l1 = rwl[RS][q][0] + fl * (rwl[RS][q][1] - rwl[RS][q][0]);
					const float l2 = rwl[RS][q][1] + fl * (rwl[RS][q][2] - rwl[RS][q][1]);
					T[0][u][li] = dry * in_l + l1 * (1.f - dry);
					// dry * in.r + (1.f - dry) * ((in.r + left * gain) >> right) >> out.r;   PingPong.k:67
					ring_wr(1, pos, in_r + l2 * gain);
					const float r2 = rwr[RS][q][1] + fr * (rwr[RS][q][2] - rwr[RS][q][1]);
					T[1][u][li] = dry * in_r + r2 * (1.f - dry);
				}
			}
			const int c = j + P;
			if ((ST || (c >= 0 && c < nch)) && !(KLG_PPX_ABLATE & 4)) {                              // the ring rows of chunk j + P
				const int pos1 = wrap(pos_j + P * PPX_CHUNK);
#pragma unroll
				for (int q = 0; q < PASSES; q++) {
					const int u = u0 + q * SPW, pos = wrap(pos1 + u);
					const float dl = MOVING ? S.D[c & (ND - 1)][u][li] : dly;
					const Tap tl = delay_set(pos, SIZE, dl * a.fs.f);               // left.set(delay * fs)
					const Tap tr = delay_set(pos, SIZE, 0.5f * dl * a.fs.f);        // right.set(0.5f * delay * fs)
					const int i0 = tl.position, i1 = ring_succ(i0, SIZE), i2 = ring_succ(i1, SIZE);   // (a tap may sit on the pad row SIZE: klg_delay.hpp)
					const int j0 = tr.position, j1 = ring_succ(j0, SIZE), j2 = ring_succ(j1, SIZE);
					rwl[RS][q][0] = ring_rd(0, i0); rwl[RS][q][1] = ring_rd(0, i1); rwl[RS][q][2] = ring_rd(0, i2);
					rwr[RS][q][0] = ring_rd(1, j0); rwr[RS][q][1] = ring_rd(1, j1); rwr[RS][q][2] = ring_rd(1, j2);
				}
			}
			const int c2 = j + P + 1;
			if (ST || (c2 >= 0 && c2 < nch)) {                              // the caller's rows of chunk j + P + 1
				const char* src = ld_ptr;
#pragma unroll
				for (int i = 0; i < NV; i++) {
					const int row = vrow + RPP * i;
					fvec v;
#pragma unroll
					for (int q = 0; q < VW; q++) v[q] = 0.f;
					if (whole_group || k0 + (row >> 1) < a.K) v = *(const fvec*)(src + (unsigned)(row * nb + vcol) * 4u);
#pragma unroll
					for (int q = 0; q < VW; q++) iov[IS][i * VW + q] = v[q];
				}
			}
			cursors_next();
		};
		// The first P steps only request — chunks 0 .. P - 1's ring rows, chunks 0 .. P's caller rows — and need nothing but this lane's own delay time:
		// they are issued BEFORE the workgroup knows whether the block qualifies (the control wave's words are still on their way), so the
		// decision costs no round trip of its own.  A block that does not qualify ignores what arrives (every address is a valid one).
		auto first_requests = [&](auto moving_c) __attribute__((always_inline)) {
			cursors_begin();
			auto prologue = [&](auto self, auto t_c) __attribute__((always_inline)) {
				constexpr int T = decltype(t_c)::value, J = T - P - 1;
				if constexpr (T < P) { audio_part(IntTag<((J % P) + P) % P>{}, BoolTag<false>{}, moving_c, J, nchunks); self(self, IntTag<T + 1>{}); }
			};
			prologue(prologue, IntTag<0>{});
		};
		// (not with vibrato — a block with an LFO on the delay time is never stationary, and its requests would only be in the way: the audio waves
		//  see that in two words of their own instances)
		bool may_qualify = w_audio && whole_chunks;
		if (may_qualify) { const float c2 = PPW(2), c3 = PPW(3); may_qualify = __ballot((c2 * c2) * ((c3 * c3) * 100.f) * 1.41421354f != 0.f) == 0ull; }
		if (may_qualify && !(KLG_PPX_VARIANT & 2)) first_requests(BoolTag<false>{});
		if (w_control) {
			// (stationary: every sample's delay time is sm1.)  Rows of chunk c are requested while chunks c - P .. c - 1 are not written yet.
			const bool far_deep = 0.5f * sm1 * a.fs.f >= (float)((P + 1) * PPX_CHUNK + 3) && sm1 * a.fs.f <= (float)(SIZE - PPX_CHUNK - 4);
			const bool ok = stationary && whole_chunks && __ballot(k < a.K && !far_deep) == 0ull;
			// MOVING dials without vibrato, every width (a dial being turned, a scratch, the smoothers still converging after either): every delay time of the block lies between the
			// smallest and the largest of where the chain's values are and where they are heading — controls[1].smooth() moves towards controls[1], which only ever
			// takes clamped values of controls[5].smooth(), which moves towards controls[5] — so "far" is decided for the whole block here.  The chain then runs
			// P + 1 chunks ahead of the audio; its head start (P chunks before the first request) is repaid over a span, not inside one block.
			bool ok2 = false;
			rest5_all = false;
			if constexpr (MV) {
				const float t5 = __builtin_amdgcn_fmed3f(c5, a.c1_min, a.c1_max), s5 = __builtin_amdgcn_fmed3f(sm5, a.c1_min, a.c1_max);
				// (controls[5] at rest — smoother at its fixed point, detector quiet, as in `stationary` — sets nothing: only controls[1].smooth() still moves, towards controls[1])
				const bool rest5 = (sm5 * 0.999f + (1.f - 0.999f) * c5 == sm5) && !(fabsf(mdelay - sm5) >= 0.001f) && !(fabsf(c5 - sm5) >= 0.001f);
				const float lo = rest5 ? __builtin_fminf(sm1, c1) : __builtin_fminf(__builtin_fminf(sm1, c1), __builtin_fminf(t5, s5));
				const float hi = rest5 ? __builtin_fmaxf(sm1, c1) : __builtin_fmaxf(__builtin_fmaxf(sm1, c1), __builtin_fmaxf(t5, s5));
				const bool far2 = 0.5f * lo * a.fs.f >= (float)((P + 1) * PPX_CHUNK + 4) && hi * a.fs.f <= (float)(SIZE - PPX_CHUNK - 5);
				rest5_all = __ballot(!rest5) == 0ull;
				ok2 = !stationary && !any_vibrato && whole_chunks && nchunks >= PPX_MOVING_MIN && !(KLG_PPX_VARIANT & 16) && __ballot(k < a.K && !far2) == 0ull;
			}
			if (lane == 0) S.deep = ok ? 1 : ok2 ? 2 : 0;
			// vibrato (never stationary): a phase no LFO walks to (an uploaded record) keeps the plain chain, with sin's range test
			const bool vf = any_vibrato && __ballot(!(fabsf(lfo.position) < 1.0e4f)) == 0ull && !(KLG_PPX_VARIANT & 8);
			if (lane == 0) S.vibfast = vf ? 1 : 0;
			if (vf) { spec_pos = lfo.position; prescan(0); prescan(1); }
		}
		__syncthreads();
	if (DEEP_ONLY || a.pass == 1) {                                                  // (wave-uniform: S.deep is the workgroup's)
		if (tid == 0) a.done[wg] = S.deep == 1;
		if (S.deep != 1) return;                                                     // not stationary: nothing was written yet — the second launch renders this workgroup
	}
	if (w_control2 && S.deep != 2) return;                                         // (the twelfth wave has a part only when dials move and the pipeline runs ahead: a wave that has ended is not waited for at a barrier)
	const bool vibfast = S.vibfast != 0;
	if (!DEEP_ONLY && vibfast) { if (w_audio) sines(0); __syncthreads(); }                        // (chunk 1's are taken in step -1, chunk j + 2's in step j)
	if (S.deep) {
		moving = MV && S.deep == 2 && !(KLG_PPX_VARIANT & 16);
		// The control chain with moving dials, a chunk at a time.  One wave running all of it (the general loop: ~17 instructions a sample, each waiting for the one
		// before: ~128 cycles a sample, 1.95 us a chunk) is slower than the memory pipeline it feeds (1.3 us a chunk), however far ahead it runs.  So it is cut where it
		// only flows one way: the FIRST control wave smooths controls[5], runs the scratch detector and sets controls[1] (PingPong.k:47-56) and leaves controls[1] per sample
		// and the chunk's last detection in LDS; the SECOND, a step behind, smooths controls[1] into the delay times (PingPong.k:58) and walks the LFO's phase — which, with no
		// vibrato in the wave, only a detection touches (lfo.set(rate, pi)): from the last one of the chunk on, or all the way when there was none.
		auto chain_first = [&](const int cc) __attribute__((always_inline)) {
			const float k5 = (1.f - 0.999f) * c5;
			float (*C)[G] = S.C1[cc & 1];
			int lu = -1;
			if (KLG_PPX_ABLATE & 16) return;
			if (rest5_all) {                                                        // controls[5] at rest in every instance: controls[1] stays what it is — both buffers, once
				if (cc < 2) {
#pragma unroll 8
					for (int u = 0; u < PPX_CHUNK; u++) C[u][li] = c1;
					S.lastu[cc & 1][li] = -1;
				}
				mdelay = c5;                                                        // (no scratch: `else delay = controls[5]`)
				return;
			}
#pragma unroll 8
			for (int u = 0; u < PPX_CHUNK; u++) {
				sm5 = sm5 * 0.999f + k5;                                            // controls[5].smooth()  klang.h:1715
				const bool trig = fabsf(mdelay - sm5) >= 0.001f;                    // (double)fabsf(d) > 0.001
				mdelay = trig ? sm5 : c5;
				c1 = trig ? __builtin_amdgcn_fmed3f(sm5, a.c1_min, a.c1_max) : c1;  // controls[1].set(new_delay)
				lu = trig ? u : lu;
				C[u][li] = c1;
			}
			S.lastu[cc & 1][li] = lu;
		};
		auto chain_second = [&](const int cc) __attribute__((always_inline)) {
			const float (*C)[G] = S.C1[cc & 1];
			float (*D)[G] = S.D[cc & ((G <= 32 ? 8 : 4) - 1)];
			const int lu = S.lastu[cc & 1][li];
			if (KLG_PPX_ABLATE & 64) { for (int u = 0; u < PPX_CHUNK; u++) D[u][li] = sm1; return; }
			const bool inc_ok = !(lfo_inc >= KLG_TWO_PI);
			float pos = lu >= 0 ? KLG_PI_F : lfo.position;                          // lfo.set(rate, pi)
			const bool plain = __ballot(lu >= 0 || !inc_ok) == 0ull && !(KLG_PPX_ABLATE & 32);   // no detection in the chunk, no LFO at a standstill: both recurrences side by side, nothing to test
			int first = 0;                                                          // else the phase from the earliest "last detection" of the wave's instances on (while a scratch converges the detector fires every
#pragma unroll                                                                  // other sample: the last two samples of the chunk), every instance from its own
			for (int bit = PPX_CHUNK / 2; bit > 0; bit >>= 1) if (__ballot(lu < first + bit) == 0ull) first += bit;
			for (int u0 = 0; u0 < PPX_CHUNK; u0 += 8) {
				float cv[8];                                                        // (eight values requested at once: a round trip to LDS per eight samples, not one a sample)
#pragma unroll
				for (int i = 0; i < 8; i++) cv[i] = (1.f - 0.999f) * C[u0 + i][li];
				if (plain) {
#pragma unroll
					for (int i = 0; i < 8; i++) {
						sm1 = sm1 * 0.999f + cv[i];                                 // controls[1].smooth()
						D[u0 + i][li] = sm1;
						const float p1 = pos + lfo_inc;                             // Phase::operator+= klang.h:1518-1525
						pos = (p1 > KLG_TWO_PI) ? p1 - KLG_TWO_PI : p1;
					}
				}
				else {
#pragma unroll
					for (int i = 0; i < 8; i++) { sm1 = sm1 * 0.999f + cv[i]; D[u0 + i][li] = sm1; }
				}
			}
			if (!plain && !(KLG_PPX_ABLATE & 32))
				for (int u = first; u < PPX_CHUNK; u++) {
					const float p1 = pos + lfo_inc, p2 = (p1 > KLG_TWO_PI) ? p1 - KLG_TWO_PI : p1;
					pos = (inc_ok && u >= lu) ? p2 : pos;
				}
			lfo.position = pos;
		};
		if (MV && moving) {
			// the chain's head start: delay times of chunks 0 .. P - 1 (chunk j + P's are computed in step j - 1), controls[1] of chunks 0 .. P; only then the first requests
			for (int t = 0; t <= P; t++) {
				if (w_control) { if (t < nchunks) chain_first(t); }
				else if (w_control2 && t >= 1) chain_second(t - 1);
				__syncthreads();
			}
			if (w_control) lfo.increment = lfo_inc;
			if (w_audio) first_requests(BoolTag<true>{});
		}
		else {
			if (w_control) mdelay = c5;                                         // (no scratch: `else delay = controls[5]`)
			if (w_audio && (KLG_PPX_VARIANT & 2)) first_requests(BoolTag<false>{});
		}
		// the other waves' part: the LFO keeps running, a chunk per step (control: beside the filter waves, which are slower); FILTER of chunk j - 1
		auto other_part = [&](const int j) __attribute__((always_inline)) {
			if (MV && moving && !w_filter) {
				if (w_control) { const int cc = j + P + 2; if (cc < nchunks) chain_first(cc); }
				else if (w_control2) { const int cc = j + P + 1; if (cc < nchunks) chain_second(cc); }
			}
			else if (w_control) {
				const int cc = j - 1;
				if (cc >= 0 && cc < nchunks) {
					lfo.increment = lfo_inc;
					const bool inc_ok = !(lfo_inc >= KLG_TWO_PI);
					float pos = lfo.position;
					for (int u = 0; u < PPX_CHUNK; u++) { const float p1 = pos + lfo_inc, p2 = (p1 > KLG_TWO_PI) ? p1 - KLG_TWO_PI : p1; pos = inc_ok ? p2 : pos; }
					lfo.position = pos;
				}
			}
			else if (w_filter && j >= 1 && j <= nchunks) filter_stage(j - 1, BoolTag<(KLG_PPX_VARIANT & 1) != 0>{});
		};
		// Blocks of 4 / 8 / 16 chunks (128 / 256 / 512 samples): the audio waves' steps are written out one after the other — in straight-line code
		// the compiler's wait before a use is exactly "everything requested since may still be under way"; through the loop below, with its
		// guards, it waits for more than it has to (measured at 4,096 instances: 19.4 us with the loop).
		auto audio_unrolled = [&](auto nch_c, auto whole_c) __attribute__((always_inline)) {
			constexpr int NCH = decltype(nch_c)::value;
			auto run = [&](auto self, auto t_c) __attribute__((always_inline)) {
				constexpr int T = decltype(t_c)::value, J = T - 1;
				if constexpr (J <= NCH + ((KLG_PPX_VARIANT & 1) ? 0 : 1)) {
					audio_part(IntTag<((J % P) + P) % P>{}, whole_c, BoolTag<false>{}, J, NCH);
					__syncthreads();
					self(self, IntTag<T + 1>{});
				}
			};
			run(run, IntTag<0>{});
		};
		const bool unrolled = nchunks == 4 || nchunks == 8 || nchunks == 16;
		if (unrolled && w_audio) {
			// (a workgroup whose instances all exist — every one but a bank's last — runs the form in which no access is predicated: the compiler then counts what is under way)
			if (nchunks == 8) { if (whole_group) audio_unrolled(IntTag<8>{}, IntTag<3>{}); else audio_unrolled(IntTag<8>{}, IntTag<0>{}); }
			else if (nchunks == 4) audio_unrolled(IntTag<4>{}, IntTag<0>{}); else audio_unrolled(IntTag<16>{}, IntTag<0>{});
		}
		else if (unrolled) {
			for (int j = -1; j <= nchunks + ((KLG_PPX_VARIANT & 1) ? 0 : 1); j++) { other_part(j); __syncthreads(); }
		}
		else {
		// (audio_c: the audio waves and the others walk the SAME steps — the same barriers — in loops of their own (round 6).  In one loop with the role tested inside
		//  every step, the compiler's count of the requests under way, which follows no particular path, allowed for a wave that is an audio wave in one step and not
		//  in the next three: it made every audio step wait for all but the last two or three requests, and the pipeline ran one step deep instead of P.)
		auto steps = [&](auto moving_c, auto audio_c) __attribute__((always_inline)) {
			constexpr bool AUDIO = decltype(audio_c)::value;
			auto deep_step = [&](auto slot_c, auto steady_c, const int j) __attribute__((always_inline)) {
				if constexpr (AUDIO) audio_part(slot_c, steady_c, moving_c, j, nchunks); else other_part(j);
				__syncthreads();
			};
			const int jend = nchunks + ((KLG_PPX_VARIANT & 1) ? 0 : 1);
			int j = -1;
			deep_step(IntTag<P - 1>{}, BoolTag<false>{}, j); ++j;                   // (-1 = P - 1 mod P)
			// a group of P steps, slots 0 .. P - 1 (j is a multiple of P at its head); guarded: leaves where the block ends
			auto group = [&](auto steady_c) __attribute__((always_inline)) -> bool {
				constexpr bool ST = decltype(steady_c)::value != 0;
				if (!ST && j > jend) return false;
				deep_step(IntTag<0>{}, steady_c, j); ++j;
				if (!ST && j > jend) return false;
				deep_step(IntTag<1 % P>{}, steady_c, j); ++j;
				if constexpr (P > 2) { if (!ST && j > jend) return false; deep_step(IntTag<2 % P>{}, steady_c, j); ++j; }
				if constexpr (P > 3) { if (!ST && j > jend) return false; deep_step(IntTag<3 % P>{}, steady_c, j); ++j; }
				return true;
			};
			// the head of the span with its guards (steps 0 .. P - 1), the inner steps in groups without any (every chunk a step of theirs touches
			// exists: 2 <= j, j + P - 1 <= nchunks - P - 2), the tail with guards again
			if (group(BoolTag<false>{})) {
				// (everything requested so far is waited for ONCE here, visibly to the compiler: what is under way when the loop is entered came from guarded steps, and merged
				//  into the loop's own state it made every step of the loop wait for all but two or three of its requests — the request-ahead pipeline ran one step deep)
				if (whole_group) { if constexpr (AUDIO) __builtin_amdgcn_s_waitcnt(0x0F70); while (j + P - 1 <= nchunks - P - 2) group(IntTag<2>{}); }
				else while (j + P - 1 <= nchunks - P - 2) group(BoolTag<true>{});
				while (group(BoolTag<false>{})) {}
			}
		};
			auto by_role = [&](auto moving_c) __attribute__((always_inline)) { if (w_audio) steps(moving_c, BoolTag<true>{}); else steps(moving_c, BoolTag<false>{}); };
			if constexpr (MV) { if (moving) by_role(BoolTag<true>{}); else by_role(BoolTag<false>{}); }
			else by_role(BoolTag<false>{});
		}
	}
	else if constexpr (!DEEP_ONLY)
	for (int j = -1; j <= nchunks; j++) {
		// ---------------- io rows of chunk j+1 (audio waves; landed in LDS at the end of the step) ----------------
		const int jn = j + 1;
		const bool load_next = w_audio && jn < nchunks;
		constexpr int IOV = G * 2 * PPX_CHUNK / (PPX_AUDIO * 64);               // values of the caller's chunk per audio thread
		float iov[IOV];
		// (the thread index is laundered through an empty asm once per step: otherwise every per-thread address and bounds
		//  predicate below is loop-invariant, gets hoisted out of the chunk loop, and ~100 VGPRs stay live for the whole kernel)
		int at = tid - 64; asm volatile("" : "+v"(at));
		const int acol = at & 31, arow = at >> 5;                                  // 512 audio threads: 16 rows x 32 samples per pass, 8 passes
		const int ns0 = jn * PPX_CHUNK, ncl = (n - ns0 < PPX_CHUNK) ? (n - ns0) : PPX_CHUNK;
		if (load_next) {
			const char* src = io_rows(ns0);
			if (ncl == PPX_CHUNK && k0 + G <= a.K) {                                  // a whole chunk of a whole group: no bounds to test
#pragma unroll
				for (int i = 0; i < IOV; i++) iov[i] = *(const float*)(src + (unsigned)((arow + 16 * i) * nb + acol) * 4u);
			}
			else {
#pragma unroll
				for (int i = 0; i < IOV; i++) {
					const int row = arow + 16 * i, inst = row >> 1;
					iov[i] = (acol < ncl && k0 + inst < a.K) ? *(const float*)(src + (unsigned)(row * nb + acol) * 4u) : 0.f;
				}
			}
		}
		// ---------------- CONTROL of chunk j+1 ----------------
		if (w_control && jn < nchunks) {
			float dmin = 3.0e38f, dmax = 0.f;                                         // range of the delay time over the chunk
			lfo.increment = lfo_inc;
			float (*D)[G] = S.D[jn & 1];
			if (stationary) {
				if (jn < 2) {                                                            // both buffers, once: every later chunk finds them as they are
#pragma unroll 8
					for (int u = 0; u < PPX_CHUNK; u++) D[u][li] = sm1;
					dmin = dmax = sm1;
				}
				mdelay = c5;                                                             // (no scratch: `else delay = controls[5]`)
				const bool inc_ok = !(lfo_inc >= KLG_TWO_PI);                            // the LFO keeps running: Phase::operator+= klang.h:1518-1525
				float pos = lfo.position;
				for (int u = 0; u < ncl; u++) { const float p1 = pos + lfo_inc, p2 = (p1 > KLG_TWO_PI) ? p1 - KLG_TWO_PI : p1; pos = inc_ok ? p2 : pos; }
				lfo.position = pos;
			}
			else if (!any_vibrato) {
				// No instance of the wave has vibrato (controls[2] == 0: `lfo * vibrato * 0.00005` is +-0, `controls[1] + (+-0)` is controls[1],
				// which is already clamped): the LFO only advances its phase.  This serial chain — one wave, one dependent instruction after
				// the other — paces the whole pipeline, so it is written without branches: ~16 operations per sample (it was ~45 with them).
				const float k5 = (1.f - 0.999f) * c5;                              // (1.f - 0.999f) * value: the control does not change inside a block
				const bool inc_ok = !(lfo_inc >= KLG_TWO_PI);                      // Phase::operator+= klang.h:1518-1525
				float pos = lfo.position;
				auto chain = [&](const int count) {
#pragma unroll 8
				for (int u = 0; u < count; u++) {
					sm5 = sm5 * 0.999f + k5;                                        // controls[5].smooth()  klang.h:1715
					// (double)fabsf(d) > 0.001  <=>  fabsf(d) >= 0.001f: 0.001f is the smallest float above the double 0.001
					const bool trig = fabsf(mdelay - sm5) >= 0.001f;
					mdelay = trig ? sm5 : c5;
					c1 = trig ? __builtin_amdgcn_fmed3f(sm5, a.c1_min, a.c1_max) : c1;   // controls[1].set(new_delay): the clamp is the median of (x, min, max)
					pos = trig ? KLG_PI_F : pos;                                    // lfo.set(rate, pi)
					sm1 = sm1 * 0.999f + (1.f - 0.999f) * c1;                       // controls[1].smooth()
					D[u][li] = sm1;
					const float p1 = pos + lfo_inc, p2 = (p1 > KLG_TWO_PI) ? p1 - KLG_TWO_PI : p1;
					pos = inc_ok ? p2 : pos;
					dmin = __builtin_fminf(dmin, sm1); dmax = __builtin_fmaxf(dmax, sm1);
				}
				};
				if (ncl == PPX_CHUNK) chain(PPX_CHUNK); else chain(ncl);              // (a whole chunk: a constant trip count)
				lfo.position = pos;
			}
			else {
				auto vibrato_serial = [&](int u) {                                     // -> a scratch detector fired (any instance of the wave)
					bool fired = false;
#pragma unroll 4
					for (; u < ncl; u++) {
						sm5 = sm5 * 0.999f + (1.f - 0.999f) * c5;                           // controls[5].smooth()  klang.h:1715
						const float new_delay = sm5;
						if (fabsf(mdelay - new_delay) >= 0.001f) {
							mdelay = new_delay;
							c1 = (new_delay < a.c1_min) ? a.c1_min : (a.c1_max < new_delay) ? a.c1_max : new_delay;   // controls[1].set()
							lfo.position = KLG_PI_F;                                        // lfo.set(rate, pi)
							fired = true;
						}
						else mdelay = c5;
						sm1 = sm1 * 0.999f + (1.f - 0.999f) * c1;                           // controls[1].smooth()
						const float delay = sm1;
						const float lfo_out = basic_sine(lfo);                              // fp64 sin
						const float nc1 = c1 + lfo_out * vibrato * 0.00005f;
						c1 = (nc1 < a.c1_min) ? a.c1_min : (a.c1_max < nc1) ? a.c1_max : nc1;
						D[u][li] = delay;
						dmin = fminf(dmin, delay); dmax = fmaxf(dmax, delay);
					}
					return __ballot(fired) != 0ull;
				};
				// The LFO's fp64 sine is most of this chain (~450 of a sample's ~640 cycles) and depends on nothing but the LFO's phase, which only the scratch
				// detector disturbs.  So the phases are walked ahead on their own (prescan: three operations a sample, this wave, three chunks before the chain
				// gets there), the AUDIO waves — 512 lanes that mostly wait for memory — take the sines of a whole chunk at once a step later (`sines`), and
				// the chain reads them from LDS, eight samples at a time without a branch.  A detector that fires ends the speculation: the rest of the chunk
				// and the next one (whose sines are already under way from the wrong phases) are walked the plain way, the phases restart from the true one.
				auto respec = [&]() {                                                  // (true phase at the end of chunk jn -> the start of chunk jn + 2)
					const bool inc_ok = !(lfo_inc >= KLG_TWO_PI);
					float pos = lfo.position;
					for (int u = 0; u < PPX_CHUNK; u++) { const float p1 = pos + lfo_inc, p2 = (p1 > KLG_TWO_PI) ? p1 - KLG_TWO_PI : p1; pos = inc_ok ? p2 : pos; }
					spec_pos = pos; plain_chunk = jn + 1;
				};
				if (vibfast && jn != plain_chunk) {
					float (*PH)[G] = S.ph[jn & 1]; float (*SN)[G] = S.sn[jn & 1];
					int u = 0;
					for (; u + 8 <= ncl; u += 8) {
						float sv[8];
#pragma unroll
						for (int i = 0; i < 8; i++) sv[i] = SN[u + i][li] * vibrato * 0.00005f;
						const float sm5_0 = sm5, mdelay_0 = mdelay, sm1_0 = sm1, c1_0 = c1, dmin_0 = dmin, dmax_0 = dmax;
						bool fired = false;
#pragma unroll
						for (int i = 0; i < 8; i++) {
							sm5 = sm5 * 0.999f + (1.f - 0.999f) * c5;                       // controls[5].smooth()  klang.h:1715
							fired = fired || (fabsf(mdelay - sm5) >= 0.001f);              // the scratch detector
							mdelay = c5;
							sm1 = sm1 * 0.999f + (1.f - 0.999f) * c1;                       // controls[1].smooth()
							const float nc1 = c1 + sv[i];
							c1 = __builtin_amdgcn_fmed3f(nc1, a.c1_min, a.c1_max);          // controls[1].set(): the clamp is the median of (x, min, max)
							D[u + i][li] = sm1;
							dmin = fminf(dmin, sm1); dmax = fmaxf(dmax, sm1);
						}
						if (__ballot(fired) != 0ull) { sm5 = sm5_0; mdelay = mdelay_0; sm1 = sm1_0; c1 = c1_0; dmin = dmin_0; dmax = dmax_0; break; }
					}
					lfo.position = PH[u][li];                                           // (row u: the phase sample u starts from; u == ncl: where the chunk ends)
					if (u < ncl && vibrato_serial(u)) respec();                        // (a ragged last chunk's tail comes here too, without a firing)
				}
				else if (vibrato_serial(0) && vibfast) respec();
				if (vibfast && jn + 2 < nchunks) prescan(jn + 2);
			}
			// x -> 0.5f * x * fs and x -> x * fs are monotonic, so the extreme delays decide for the whole chunk
			if (!stationary || jn < 2) {
				const bool far = 0.5f * dmin * a.fs.f >= (float)(PPX_CHUNK + 3) && dmax * a.fs.f <= (float)(SIZE - PPX_CHUNK - 4);
				const bool all_far = __ballot(k < a.K && !far) == 0ull;             // padding lanes (zero state, zero delay) do not veto
				// HALF-far: every tap further behind the cursor than HALF a chunk is long (+ 3) — the shortest delay the dial allows, 1 ms, is there at 44.1 / 48 kHz
				// (right tap: 22 / 24 samples; PingPong.k's preset "Doctor Who?").  The chunk is then two half-chunks that are each far: audio waves 1 - 4 take
				// samples 0 - 15, a barrier, waves 5 - 8 samples 16 - 31 (flag 2)
				const bool half = 0.5f * dmin * a.fs.f >= (float)(PPX_CHUNK / 2 + 3) && dmax * a.fs.f <= (float)(SIZE - PPX_CHUNK / 2 - 4);
				const bool all_half = __ballot(k < a.K && !half) == 0ull;
				if (lane == 0) S.far[jn & 1] = all_far ? 1 : all_half ? 2 : 0;
			}
		}
		// ---------------- AUDIO of chunk j ----------------
		const bool audio_now = j >= 0 && j < nchunks;
		const int farj = audio_now ? S.far[j & 1] : 1;                             // (workgroup-uniform: written a step ago, behind that step's barrier)
		auto audio_stage = [&]() __attribute__((always_inline)) {
			const int s0 = j * PPX_CHUNK, cl = (n - s0 < PPX_CHUNK) ? (n - s0) : PPX_CHUNK;
			const int pos0 = (int)(((long long)a.position + s0) % SIZE);
			float (*T)[PPX_CHUNK][G + 1] = S.tile[j & 3];
			if (farj != 0) {
				// a wave's PPX_PER samples: SPW of them side by side in its lanes (lane = sample slot x instance), PASSES passes.  FULL (a whole chunk: every
				// chunk but a block's ragged last) is a compile-time variant: with the per-pass bounds tests in place every pass ends in a join, and the
				// joins cost a dozen 64-bit register copies per pass (the rows in flight are live across them)
				auto far_chunk = [&](auto full_c) {
					constexpr bool FULL = decltype(full_c)::value;
				const int u0 = (wv - 1) * PPX_PER + lq;
					Tap tl[PASSES], tr[PASSES];
					float pl[PASSES][3], pr[PASSES][3];
#pragma unroll
					for (int q = 0; q < PASSES; q++) if (FULL || u0 + q * SPW < cl) {
						const int u = u0 + q * SPW;
						const float delay = S.D[j & 1][u][li];
						const int pos = wrap(pos0 + u);
						tl[q] = delay_set(pos, SIZE, delay * a.fs.f);                   // left.set(delay * fs)
						tr[q] = delay_set(pos, SIZE, 0.5f * delay * a.fs.f);            // right.set(0.5f * delay * fs)
						const int i0 = tl[q].position, i1 = ring_succ(i0, SIZE), i2 = ring_succ(i1, SIZE);   // (a tap may sit on the pad row SIZE: klg_delay.hpp)
						const int j0 = tr[q].position, j1 = ring_succ(j0, SIZE), j2 = ring_succ(j1, SIZE);
						pl[q][0] = ring_rd(0, i0); pl[q][1] = ring_rd(0, i1); pl[q][2] = ring_rd(0, i2);
						pr[q][0] = ring_rd(1, j0); pr[q][1] = ring_rd(1, j1); pr[q][2] = ring_rd(1, j2);
					}
#pragma unroll
					for (int q = 0; q < PASSES; q++) if (FULL || u0 + q * SPW < cl) {
						const int u = u0 + q * SPW, pos = wrap(pos0 + u);
						const float in_l = T[0][u][li], in_r = T[1][u][li];
						// dry * in.l + (1.f - dry) * ((in.l + right * gain) >> left) >> out.l;   PingPong.k:66
						const float r1 = pr[q][0] + tr[q].fraction * (pr[q][1] - pr[q][0]);
						ring_wr(0, pos, in_l + r1 * gain);
						const float l1 = pl[q][0] + tl[q].fraction * (pl[q][1] - pl[q][0]);
						const float l2 = pl[q][1] + tl[q].fraction * (pl[q][2] - pl[q][1]);
						T[0][u][li] = dry * in_l + l1 * (1.f - dry);
						// dry * in.r + (1.f - dry) * ((in.r + left * gain) >> right) >> out.r;   PingPong.k:67
						ring_wr(1, pos, in_r + l2 * gain);
						const float r2 = pr[q][1] + tr[q].fraction * (pr[q][2] - pr[q][1]);
						T[1][u][li] = dry * in_r + r2 * (1.f - dry);
					}
							};
				if (cl == PPX_CHUNK) far_chunk(BoolTag<true>{}); else far_chunk(BoolTag<false>{});
			}
			else if (G <= 32 && wv == 1) {
				// A NEAR TAP (a delay under ~0.75 ms — PingPong.k's preset "Doctor Who?" sits there — or one within a chunk of the whole line): the chunk is
				// walked in sample order by one wave, lane = instance.  Through memory that walk is three dependent round trips per sample (a tap may read what
				// the sample before wrote): 40 us per chunk.  Here the wave first fetches, for every sample of the chunk, its six ring values AS THEY STAND
				// BEFORE THE CHUNK (all in flight together, 64 / G samples per pass); the walk then takes a value from what the chunk has written so far
				// (LDS) whenever its row is one of those, else the fetched one — the same values the reference reads — and waits for LDS only.
				{
					constexpr int NSPW = 64 / G;
#pragma unroll
					for (int p0 = 0; p0 < PPX_CHUNK; p0 += NSPW) {
						const int u = p0 + lq;
						if (u < cl) {
							const float delay = S.D[j & 1][u][li];
							const int pos = wrap(pos0 + u);
							const Tap tl = delay_set(pos, SIZE, delay * a.fs.f), tr = delay_set(pos, SIZE, 0.5f * delay * a.fs.f);
							const int i0 = tl.position, i1 = ring_succ(i0, SIZE), i2 = ring_succ(i1, SIZE);
							const int j0 = tr.position, j1 = ring_succ(j0, SIZE), j2 = ring_succ(j1, SIZE);
							S.nv[u][0][li] = ring_rd(0, i0); S.nv[u][1][li] = ring_rd(0, i1); S.nv[u][2][li] = ring_rd(0, i2);
							S.nv[u][3][li] = ring_rd(1, j0); S.nv[u][4][li] = ring_rd(1, j1); S.nv[u][5][li] = ring_rd(1, j2);
						}
					}
				}
				wave_sync();
				if (lane < G) {
					// row `row` of `line` as sample u finds it: written by this chunk (its first `upto` samples) or as fetched
					auto seen = [&](const int line, const int row, const float fetched, const int upto) {
						int d = row - pos0; d = d < 0 ? d + SIZE : d;
						const bool mine = row < SIZE && d < upto;
						const float w = S.nw[line][mine ? d : 0][li];
						return mine ? w : fetched;
					};
					for (int u = 0; u < cl; u++) {
						const float delay = S.D[j & 1][u][li];
						const int pos = wrap(pos0 + u);
						const Tap tl = delay_set(pos, SIZE, delay * a.fs.f), tr = delay_set(pos, SIZE, 0.5f * delay * a.fs.f);
						const int i0 = tl.position, i1 = ring_succ(i0, SIZE), i2 = ring_succ(i1, SIZE);
						const int j0 = tr.position, j1 = ring_succ(j0, SIZE), j2 = ring_succ(j1, SIZE);
						const float in_l = T[0][u][li], in_r = T[1][u][li];
						const float ra = seen(1, j0, S.nv[u][3][li], u), rb = seen(1, j1, S.nv[u][4][li], u);
						const float r1 = ra + tr.fraction * (rb - ra);                          // right * gain: Delay::process under the head
						const float wl = in_l + r1 * gain;
						S.nw[0][u][li] = wl; ring_wr(0, pos, wl);                             // (in.l + right * gain) >> left
						const float la = seen(0, i0, S.nv[u][0][li], u + 1), lb = seen(0, i1, S.nv[u][1][li], u + 1), lc = seen(0, i2, S.nv[u][2][li], u + 1);
						const float l1 = la + tl.fraction * (lb - la), l2 = lb + tl.fraction * (lc - lb);
						T[0][u][li] = dry * in_l + l1 * (1.f - dry);
						const float wr = in_r + l2 * gain;
						S.nw[1][u][li] = wr; ring_wr(1, pos, wr);                             // (in.r + left * gain) >> right
						const float rb2 = seen(1, j1, S.nv[u][4][li], u + 1), rc = seen(1, j2, S.nv[u][5][li], u + 1);
						const float r2 = rb2 + tr.fraction * (rc - rb2);
						T[1][u][li] = dry * in_r + r2 * (1.f - dry);
					}
				}
			}
			else if (wv == 1 && lane < G) {                                         // a near tap: the chunk is walked in order by one wave, lane = instance
				const int col = (k0 & (FX_WG - 1)) + li;
				Ring left = { ring0 + col, FX_WG, SIZE }, right = { ring0 + (size_t)PP_ROWS * FX_WG + col, FX_WG, SIZE };
				for (int u = 0; u < cl; u++) {
					const float delay = S.D[j & 1][u][li];
					const int pos = wrap(pos0 + u);
					Tap tl = delay_set(pos, SIZE, delay * a.fs.f), tr = delay_set(pos, SIZE, 0.5f * delay * a.fs.f);
					const float in_l = T[0][u][li], in_r = T[1][u][li];
					const float r1 = delay_process(right, tr);
					left.wr(pos, in_l + r1 * gain);
					const float l1 = delay_process(left, tl), l2 = delay_process(left, tl);
					T[0][u][li] = dry * in_l + l1 * (1.f - dry);
					right.wr(pos, in_r + l2 * gain);
					const float r2 = delay_process(right, tr);
					T[1][u][li] = dry * in_r + r2 * (1.f - dry);
				}
			}
		};
		// far: every audio wave its samples at once.  Half-far (farj == 2): waves 1 .. 4 (samples 0 - 15), then — behind a barrier of the whole workgroup, so that
		// what they wrote is there — waves 5 .. 8 (samples 16 - 31).  Near (0): wave 1 walks the chunk
		if (w_audio && audio_now && !(farj == 2 && wv > PPX_AUDIO / 2)) audio_stage();
		if (farj == 2) { __syncthreads(); if (w_audio && wv > PPX_AUDIO / 2) audio_stage(); }
		if (vibfast && w_audio && j + 2 < nchunks) sines(j + 2);                    // vibrato: the sines the control chain reads two steps from now
		// ---------------- FILTER of chunk j-1, then its store ----------------
		if (w_filter && j >= 1 && j <= nchunks) filter_stage(j - 1, BoolTag<true>{});
		if (load_next) {
#pragma unroll
			for (int i = 0; i < IOV; i++) { const int row = arow + 16 * i; S.tile[jn & 3][row & 1][acol][row >> 1] = iov[i]; }
		}
		__syncthreads();
	}
	if (k < a.K && lane < G) {
		float* Wr = a.state + k;
		if (w_control) {
			Wr[(size_t)1 * a.kpad] = c1; Wr[(size_t)PP_SM5 * a.kpad] = sm5; Wr[(size_t)PP_DELAY * a.kpad] = mdelay; Wr[(size_t)PP_LFO_INC * a.kpad] = lfo.increment;
			if (!moving) { Wr[(size_t)PP_SM1 * a.kpad] = sm1; Wr[(size_t)PP_LFO_POS * a.kpad] = lfo.position; }
		}
		if (w_control2 && moving) { Wr[(size_t)PP_SM1 * a.kpad] = sm1; Wr[(size_t)PP_LFO_POS * a.kpad] = lfo.position; }   // (the second control wave's part of the chain)
		if (w_filter) { Wr[(size_t)(PP_Z + 2 * fch) * a.kpad] = dc.z0; Wr[(size_t)(PP_Z + 2 * fch + 1) * a.kpad] = dc.z1; }
	}
#undef PPW
}

// =================================================================================================
// Reverb.k
// =================================================================================================
// per-instance words
enum {
	RV_CTL = 0,                 // dry(c0) c1 c2 c3 wet(c4)
	RV_EZ = 5,                  // early: lpf z0,z1 (L), lpf z0,z1 (R), hpf z0,z1 (L), hpf z0,z1 (R)
	RV_ELPF = 13, RV_EHPF = 18, // early filter coefficients (b0 b1 b2 a1 a2)
	RV_ECOUNT = 23,
	RV_ETIMES = 24, RV_EGL = 44, RV_EGR = 64,
	RV_FD = 84,                 // 16 x FilteredDelay
	FD_Z0 = 0, FD_Z1 = 1, FD_IN = 2, FD_LASTP = 3, FD_LASTF = 4, FD_GAIN = 5, FD_COEF = 6, FD_WORDS = 11,
	RV_WORDS = RV_FD + 16 * FD_WORDS
};
enum { RV_ESIZE = 21600, RV_FSIZE = 192000 };
// ring layout 1 (a contiguous ring per (instance, line): klg_fx_reverb_q, and klg_fx_reverb when it stands in for it): every line is followed by a tail — see klg_fx_reverb_q
enum { RV_FPAD = 32, RV_EMIRROR = 16, RV_EZERO = RV_ESIZE + RV_EMIRROR, RV_EPAD = RV_EMIRROR + 16, RV_FSTRIDE = RV_FSIZE + RV_FPAD, RV_ESTRIDE = RV_ESIZE + RV_EPAD };

struct ReverbArgs {
	float* state; size_t kpad; int K;
	float* early_rings;         // [kpad/64][2][21600][64]
	float* fd_rings;            // [kpad/64][16][192000][64]
	int epos;                   // early write cursor at block start
	int fpos;                   // FilteredDelay write cursor at block start (advances 2 per sample)
	float* io; int n;
	int layout;                 // 1: every (instance, line) its own contiguous ring (klg_fx_reverb_q; the one layout since round 5).  0 — rings tiled per 64 instances, position-major rows — was klg_fx_reverb16's; klg_fx_reverb still reads both
	float* early_sums;          // layout 1: [kpad][2][n] the block's early-reflection sums (klg_fx_reverb_early writes them, klg_fx_reverb_q<true> reads them), else null
};

struct FDelay { Biquad f; float in, gain; Tap last; Ring ring; };

// FilteredDelay::process Reverb.k:130-132 : (in >> delay >> filter) * gain >> out
__device__ __forceinline__ float fd_process(FDelay& d, int& wpos) {
	d.ring.wr(wpos, d.in);
	if (d.ring.stride == 1 && wpos < 32) d.ring.wr(wpos + RV_FSIZE, d.in);  // layout 1: the mirror tail of klg_fx_reverb_q
	const float t = delay_process(d.ring, d.last);
	return biquad_process(d.f, t) * d.gain;
}

__device__ __forceinline__ void fd_load(FDelay& d, const ReverbArgs& a, int idx, int k) {
	const float* s = a.state + (size_t)(RV_FD + idx * FD_WORDS) * a.kpad + k;
	d.f.z0 = s[(size_t)FD_Z0 * a.kpad]; d.f.z1 = s[(size_t)FD_Z1 * a.kpad]; d.in = s[(size_t)FD_IN * a.kpad];
	d.last.position = __float_as_int(s[(size_t)FD_LASTP * a.kpad]); d.last.fraction = s[(size_t)FD_LASTF * a.kpad];
	d.gain = s[(size_t)FD_GAIN * a.kpad];
	d.f.b0 = s[(size_t)(FD_COEF + 0) * a.kpad]; d.f.b1 = s[(size_t)(FD_COEF + 1) * a.kpad]; d.f.b2 = s[(size_t)(FD_COEF + 2) * a.kpad];
	d.f.a1 = s[(size_t)(FD_COEF + 3) * a.kpad]; d.f.a2 = s[(size_t)(FD_COEF + 4) * a.kpad];
	if (a.layout) { d.ring.base = a.fd_rings + ((size_t)k * 16 + idx) * RV_FSTRIDE; d.ring.stride = 1; }   // the line + its mirror tail (klg_fx_reverb_q)
	else { d.ring.base = a.fd_rings + ((size_t)blockIdx.x * 16 + idx) * RV_FSIZE * FX_WG + threadIdx.x; d.ring.stride = FX_WG; }
	d.ring.size = RV_FSIZE;
}
__device__ __forceinline__ void fd_store(const FDelay& d, const ReverbArgs& a, int idx, int k) {
	float* s = a.state + (size_t)(RV_FD + idx * FD_WORDS) * a.kpad + k;
	s[(size_t)FD_Z0 * a.kpad] = d.f.z0; s[(size_t)FD_Z1 * a.kpad] = d.f.z1; s[(size_t)FD_IN * a.kpad] = d.in;
	s[(size_t)FD_LASTP * a.kpad] = __int_as_float(d.last.position);
}

// LateReflections::process Reverb.k:153-168 (each FilteredDelay is processed twice per sample, see the oracle notes)
__device__ __forceinline__ float late_process(FDelay (&d)[4], float in, int wpos0) {
	int w0 = wpos0, w1 = (wpos0 + 1 == RV_FSIZE) ? 0 : wpos0 + 1;
	float dl[4];
#pragma unroll
	for (int k = 0; k < 4; k++) dl[k] = fd_process(d[k], w0);
	// signals<4> >> Matrix klang.h:1462-1467 with the FDN matrix of Reverb.k:158-161, row-major, products summed left to right
	float fb[4];
	fb[0] = 0.f * dl[0] + 1.f * dl[1] + 1.f * dl[2] + -1.f * dl[3];
	fb[1] = -1.f * dl[0] + 0.f * dl[1] + -1.f * dl[2] + 1.f * dl[3];
	fb[2] = -1.f * dl[0] + 1.f * dl[1] + 0.f * dl[2] + -1.f * dl[3];
	fb[3] = 1.f * dl[0] + -1.f * dl[1] + 1.f * dl[2] + 0.f * dl[3];
#pragma unroll
	for (int k = 0; k < 4; k++) d[k].in = fb[k] + in;
	const float o0 = fd_process(d[0], w1);
	const float o1 = fd_process(d[1], w1);
	const float s01 = o0 + o1;
	const float s012 = fd_process(d[2], w1) + s01;
	return fd_process(d[3], w1) + s012;
}

__global__ __launch_bounds__(FX_WG) void klg_fx_reverb(const ReverbArgs a) {
	__shared__ float tile[2 * FX_CHUNK * FX_LD];
	const int lane = threadIdx.x, k0 = blockIdx.x * FX_WG, k = k0 + lane;
	const size_t KP = a.kpad;
	const float* S = a.state + k;
#define RVW(w) S[(size_t)(w) * KP]
	const float dry = RVW(RV_CTL + 0), c1 = RVW(RV_CTL + 1), c2 = RVW(RV_CTL + 2), c3 = RVW(RV_CTL + 3), wet = RVW(RV_CTL + 4);
	Biquad elpf[2], ehpf[2];
#pragma unroll
	for (int c = 0; c < 2; c++) {
		elpf[c] = { RVW(RV_ELPF + 0), RVW(RV_ELPF + 1), RVW(RV_ELPF + 2), RVW(RV_ELPF + 3), RVW(RV_ELPF + 4), RVW(RV_EZ + 2 * c), RVW(RV_EZ + 2 * c + 1) };
		ehpf[c] = { RVW(RV_EHPF + 0), RVW(RV_EHPF + 1), RVW(RV_EHPF + 2), RVW(RV_EHPF + 3), RVW(RV_EHPF + 4), RVW(RV_EZ + 4 + 2 * c), RVW(RV_EZ + 4 + 2 * c + 1) };
	}
	const int ecount = __float_as_int(RVW(RV_ECOUNT));
	float* etile = a.early_rings + (size_t)blockIdx.x * 2 * RV_ESIZE * FX_WG + lane;
	Ring el = { etile, FX_WG, RV_ESIZE }, er = { etile + (size_t)RV_ESIZE * FX_WG, FX_WG, RV_ESIZE };
	if (a.layout) { el.base = a.early_rings + (size_t)k * 2 * RV_ESTRIDE; er.base = el.base + RV_ESTRIDE; el.stride = er.stride = 1; }
	FDelay mid0[4], mid1[4], late0[4], late1[4];
#pragma unroll
	for (int j = 0; j < 4; j++) { fd_load(mid0[j], a, 0 + j, k); fd_load(mid1[j], a, 4 + j, k); fd_load(late0[j], a, 8 + j, k); fd_load(late1[j], a, 12 + j, k); }
	int epos = a.epos, fpos = a.fpos;

	for (int s0 = 0; s0 < a.n; s0 += FX_CHUNK) {
		const int cl = (a.n - s0 < FX_CHUNK) ? (a.n - s0) : FX_CHUNK;
		io_load_chunk(tile, a.io, k0, a.K, a.n, s0, cl, lane);
		wave_sync();
		for (int s = 0; s < cl; s++) {
			const float in_l = tile[(0 * FX_CHUNK + s) * FX_LD + lane], in_r = tile[(1 * FX_CHUNK + s) * FX_LD + lane];
			// EarlyReflections::process Reverb.k:87-93 : in >> lpf >> hpf >> delay; out = sum delay(times[d]) * gains[d]
			const float fl = biquad_process(ehpf[0], biquad_process(elpf[0], in_l));
			const float fr = biquad_process(ehpf[1], biquad_process(elpf[1], in_r));
			el.wr(epos, fl); er.wr(epos, fr);
			if (a.layout && epos < 16) { el.wr(epos + RV_ESIZE, fl); er.wr(epos + RV_ESIZE, fr); }   // layout 1: mirror tail
			epos = (epos + 1 == RV_ESIZE) ? 0 : epos + 1;
			float r1l = 0.f, r1r = 0.f;
			for (int d = 0; d < ecount; d++) {                                          // Stereo::Delay::tap(float) klang.h:4668-4681
				float tl, tr;
				stereo_delay_tap(el, er, epos, RVW(RV_ETIMES + d), tl, tr);
				r1l += tl * RVW(RV_EGL + d);
				r1r += tr * RVW(RV_EGR + d);
			}
			// Reflections::process Reverb.k:223-245
			const float r2l = late_process(mid0, r1l, fpos);
			const float r2r = late_process(mid1, r1r, fpos);
			const float r3l = late_process(late0, r2l, fpos);
			const float r3r = late_process(late1, r2r, fpos);
			fpos += 2; if (fpos >= RV_FSIZE) fpos -= RV_FSIZE;
			const float refl_l = (r1l * c1 + r2l * c2) + r3l * c3;
			const float refl_r = (r1r * c1 + r2r * c2) + r3r * c3;
			// (in * dry + (in >> reflections) * wet) >> out  Reverb.k:271 — `reflections * wet` multiplies by signals<2>{ wet, 0 }
			tile[(0 * FX_CHUNK + s) * FX_LD + lane] = in_l * dry + refl_l * wet;
			tile[(1 * FX_CHUNK + s) * FX_LD + lane] = in_r * dry + refl_r * 0.f;
		}
		wave_sync();
		io_store_chunk(tile, a.io, k0, a.K, a.n, s0, cl, lane);
		wave_sync();
	}
	if (k < a.K) {
		float* W = a.state + k;
#pragma unroll
		for (int c = 0; c < 2; c++) {
			W[(size_t)(RV_EZ + 2 * c) * KP] = elpf[c].z0; W[(size_t)(RV_EZ + 2 * c + 1) * KP] = elpf[c].z1;
			W[(size_t)(RV_EZ + 4 + 2 * c) * KP] = ehpf[c].z0; W[(size_t)(RV_EZ + 4 + 2 * c + 1) * KP] = ehpf[c].z1;
		}
#pragma unroll
		for (int j = 0; j < 4; j++) { fd_store(mid0[j], a, 0 + j, k); fd_store(mid1[j], a, 4 + j, k); fd_store(late0[j], a, 8 + j, k); fd_store(late1[j], a, 12 + j, k); }
	}
#undef RVW
}

// =================================================================================================
// Reverb.k, one wave per FOUR instances (the production kernel)
// =================================================================================================
// (Round 2's klg_fx_reverb16 — retired in round 5: this kernel serves every bank that fits in memory — spread the graph of 64 instances over sixteen WAVES: the 16 FilteredDelays met twice per sample through LDS and
// two workgroup barriers, a 64-instance group is the unit of work, and a bank of 4096 instances is 64 workgroups on 64 of the 256 CUs —
// 380 us per 256-sample block whatever the bank size below 16k, bound by barriers and by one CU's instruction issue.
// This kernel gives a wave four instances and runs a block in two phases:
//
//   1. The early reflections of the WHOLE block, lane = sample.  `out = sum delay(times[d]) * gains[d]` (Reverb.k:90-92) reads the early
//      line `times[d]` samples back — further than a block is long (checked by the host: the shortest tap must be > n + 2 samples, else
//      the single-lane kernel runs) — so none of the block's taps sees a sample the block itself writes: the sum of sample e depends on
//      history only, and the 64 lanes of the wave compute 64 consecutive samples at once.  A lane walks the taps in the reference's order
//      (same products, same left-to-right sum); the 64 lanes read 64 consecutive ring positions per tap: every cache line the load touches
//      is used completely by that one instruction.  (The previous form of this kernel — tap = lane, a running DPP sum per sample — fetched
//      a 128-byte line per tap for every eight samples: 3.3 x the ring bytes at 4096 instances, measured with FETCH_SIZE; that, not
//      instruction issue, bounded it.)  The sums wait in LDS.
//   2. The recursive part, lane = (instance i of 4) x (slot r of 16), one sample per iteration:
//        slot r = FilteredDelay r (LateReflections r / 4: mid[0], mid[1], late[0], late[1]; line r % 4)
//               + (r == 0 / 8) the early LPF >> HPF of the left / right channel (its output goes to the early line for LATER blocks),
//               + (r == 8 / 12) the left / right output sample.
//      The four delays of a LateReflections are the four lanes of a quad: the 4 x 4 feedback matrix and the output sum read their
//      neighbours with DPP quad_perm; mid[]'s sum reaches late[] by ds_bpermute.  The stages run skewed (early filter: sample t + 2,
//      mid: t + 1, late: t, output: t - 1) so that one iteration holds independent dependency chains; no barrier anywhere.
// Delay lines: every (instance, line) has its own CONTIGUOUS ring (ReverbArgs::layout 1).  FilteredDelay rows come in batches of RVQ_B
// samples per lane (16 consecutive positions: four 16-byte loads), requested a batch ahead; what a batch writes is collected in
// registers and stored as whole 64-byte pieces.  So that a window never straddles the end of a ring, every line carries a MIRROR of
// its first positions behind its last one (RV_FPAD / RV_EPAD floats, written together with the original).
// Arithmetic, operand order and summation order are exactly those of klg_fx_reverb / the reference (the two kernels are compared bit
// for bit in tests/test_gpu_fx.py: KLG_FX_REVERB1=1 selects the single-lane kernel).
enum { RVQ_MAX_INSTANCES = 65536 };      // banks up to this size run klg_fx_reverb_q: every bank that fits (12.5 MB of rings per instance: ~22 k in 288 GB).  Round 2 drew the line at 8,192
                                         // (profiles/r02_fx_sizes.md); since round 3's write-side work the kernel also wins above it — 16,384 instances: 0.467 against klg_fx_reverb16's 0.490 ms
enum { RVQ_WG = 64 };
#ifndef KLG_RVQ_ABLATE
#define KLG_RVQ_ABLATE 0          // measurement builds only (tools/rvq_ablate.sh): 1 no FilteredDelay piece stores, 2 no early-line stores, 4 no output block, 8 no record write-back
#endif
// an early line in layout 1: [0, RV_ESIZE) the ring, then RV_EMIRROR floats mirroring positions 0 .. 15, then sixteen floats that are ALWAYS ZERO (RV_EZERO; sixteen, not two: a line's length stays a multiple of 128 bytes, so every line's 32-byte store pieces are whole sectors): where a
// tap whose read position rounded up to exactly RV_ESIZE is pointed (stereo_delay_tap: that tap reads the pad — zeros —; one compare and one select per tap,
// no branch: a rarely-taken branch cost 9 VALU per tap in register copies at its join, profiles/r03_pmc/pmc_reverb_q_4096_padbranch.json)
enum { RVQ_B = 8 };
static_assert(RV_ESTRIDE % 4 == 0, "16-byte stores into an early line need 16-byte aligned line starts");
enum { RVQ_XQ_LD = 36, RVQ_XQ_FLOATS = 64 * RVQ_XQ_LD };  // the quarter exchange of the ring stores: 64 rows of 32 floats (two batches' pieces: 128 bytes per line), padded
enum { RVQ_TILE_ROWS = 17 };             // LDS per wave, rows of n floats: 0..7 the caller's block (instance * 2 + channel), 8 scrap, 9..16 the early sums

__device__ __forceinline__ float lane_get(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v))); }
template<int Q> __device__ __forceinline__ float quad_bcast(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), Q * 0x55, 0xF, 0xF, true)); }

typedef float rvq_f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float rvq_f2 __attribute__((ext_vector_type(2), aligned(8)));
typedef float rvq_v4 __attribute__((ext_vector_type(4)));
// FilteredDelay rows J + 1 .. J + 16 of a batch (J = 2 x its first sample; row J is carried over).  They are requested two batches ahead —
// into ACCUMULATION registers, by hand: a load the compiler knows about lands in registers it manages, and under this kernel's
// register pressure it splits such a live range with copies wherever it likes — also between the request and the wait, where they copy
// what has not arrived.  a0 .. a47 (three sets of 16) are outside its allocation (it uses no AGPR here: checked in the build remarks);
// rvq_await waits and moves a set into ordinary registers, which the compiler sees defined only there.
// vmcnt: loads retire in order, so "at most N outstanding" with N = the loads requested since (two later batches = 8) is exact for the
// rows whatever the ring stores in between do.  "memory" clobbers pin every other memory instruction on its side of both statements.
template<int FD, int E, int O> struct RvqMode { static constexpr int fd = FD, e = E, o = O; };   // see `step` in klg_fx_reverb_q
struct RvqRows {
	float f[2 * RVQ_B];
	template<int I> __device__ __forceinline__ float F() const { return f[I]; }
};
template<int SET> __device__ __forceinline__ void rvq_request(const float* p) {
	if constexpr (SET == 0) {
		asm volatile("global_load_dwordx4 a[0:3], %0, off\n\tglobal_load_dwordx4 a[4:7], %0, off offset:16\n\tglobal_load_dwordx4 a[8:11], %0, off offset:32\n\tglobal_load_dwordx4 a[12:15], %0, off offset:48" :: "v"(p) : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
	}
	else if constexpr (SET == 1) {
		asm volatile("global_load_dwordx4 a[16:19], %0, off\n\tglobal_load_dwordx4 a[20:23], %0, off offset:16\n\tglobal_load_dwordx4 a[24:27], %0, off offset:32\n\tglobal_load_dwordx4 a[28:31], %0, off offset:48" :: "v"(p) : "memory", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
	}
	else {
		asm volatile("global_load_dwordx4 a[32:35], %0, off\n\tglobal_load_dwordx4 a[36:39], %0, off offset:16\n\tglobal_load_dwordx4 a[40:43], %0, off offset:32\n\tglobal_load_dwordx4 a[44:47], %0, off offset:48" :: "v"(p) : "memory", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
	}
}
template<int SET, int N> __device__ __forceinline__ void rvq_await(RvqRows& X) {
	if constexpr (SET == 0) {
		asm volatile("s_waitcnt vmcnt(%16)\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\tv_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7\n\tv_accvgpr_read_b32 %8, a8\n\tv_accvgpr_read_b32 %9, a9\n\tv_accvgpr_read_b32 %10, a10\n\tv_accvgpr_read_b32 %11, a11\n\tv_accvgpr_read_b32 %12, a12\n\tv_accvgpr_read_b32 %13, a13\n\tv_accvgpr_read_b32 %14, a14\n\tv_accvgpr_read_b32 %15, a15" : "=v"(X.f[0]), "=v"(X.f[1]), "=v"(X.f[2]), "=v"(X.f[3]), "=v"(X.f[4]), "=v"(X.f[5]), "=v"(X.f[6]), "=v"(X.f[7]), "=v"(X.f[8]), "=v"(X.f[9]), "=v"(X.f[10]), "=v"(X.f[11]), "=v"(X.f[12]), "=v"(X.f[13]), "=v"(X.f[14]), "=v"(X.f[15]) : "n"(N) : "memory");
	}
	else if constexpr (SET == 1) {
		asm volatile("s_waitcnt vmcnt(%16)\n\tv_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19\n\tv_accvgpr_read_b32 %4, a20\n\tv_accvgpr_read_b32 %5, a21\n\tv_accvgpr_read_b32 %6, a22\n\tv_accvgpr_read_b32 %7, a23\n\tv_accvgpr_read_b32 %8, a24\n\tv_accvgpr_read_b32 %9, a25\n\tv_accvgpr_read_b32 %10, a26\n\tv_accvgpr_read_b32 %11, a27\n\tv_accvgpr_read_b32 %12, a28\n\tv_accvgpr_read_b32 %13, a29\n\tv_accvgpr_read_b32 %14, a30\n\tv_accvgpr_read_b32 %15, a31" : "=v"(X.f[0]), "=v"(X.f[1]), "=v"(X.f[2]), "=v"(X.f[3]), "=v"(X.f[4]), "=v"(X.f[5]), "=v"(X.f[6]), "=v"(X.f[7]), "=v"(X.f[8]), "=v"(X.f[9]), "=v"(X.f[10]), "=v"(X.f[11]), "=v"(X.f[12]), "=v"(X.f[13]), "=v"(X.f[14]), "=v"(X.f[15]) : "n"(N) : "memory");
	}
	else {
		asm volatile("s_waitcnt vmcnt(%16)\n\tv_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35\n\tv_accvgpr_read_b32 %4, a36\n\tv_accvgpr_read_b32 %5, a37\n\tv_accvgpr_read_b32 %6, a38\n\tv_accvgpr_read_b32 %7, a39\n\tv_accvgpr_read_b32 %8, a40\n\tv_accvgpr_read_b32 %9, a41\n\tv_accvgpr_read_b32 %10, a42\n\tv_accvgpr_read_b32 %11, a43\n\tv_accvgpr_read_b32 %12, a44\n\tv_accvgpr_read_b32 %13, a45\n\tv_accvgpr_read_b32 %14, a46\n\tv_accvgpr_read_b32 %15, a47" : "=v"(X.f[0]), "=v"(X.f[1]), "=v"(X.f[2]), "=v"(X.f[3]), "=v"(X.f[4]), "=v"(X.f[5]), "=v"(X.f[6]), "=v"(X.f[7]), "=v"(X.f[8]), "=v"(X.f[9]), "=v"(X.f[10]), "=v"(X.f[11]), "=v"(X.f[12]), "=v"(X.f[13]), "=v"(X.f[14]), "=v"(X.f[15]) : "n"(N) : "memory");
	}
}

// =================================================================================================
// Reverb.k, the early-reflection sums of a block as a kernel of their own (lane = sample)
// =================================================================================================
// Phase 1 of klg_fx_reverb_q depends on HISTORY only (every tap reads the early line further back than a block is long), i.e. on nothing the
// recursive part of the same block computes.  Inside that kernel it runs on a wave that is alone on its SIMD (304 registers), with nothing to
// hide its load and issue latencies behind: a third of the kernel's instructions at half the issue rate.  Here the same sums are a plain
// data-parallel launch — one lane per (instance, sample), sixteen waves per SIMD at 4,096 instances, tap words in scalar registers (the instance
// is uniform per workgroup) — and the recursive kernel starts from them (klg_fx_reverb_q<true> copies them into its LDS rows).  Same
// operations in the same order as phase 1 / EarlyReflections::process (Reverb.k:90-92, klang.h:4668-4681): bit-identical sums.
// Because the sums of block b depend on samples >= 45 ms old only, this launch may also run BESIDE the recursive kernel of block b - 1 (the host
// puts it on a second stream): the waves of this kernel fill the issue slots the lone recursive wave of a SIMD leaves empty.
enum { RVE_WG = 256 };
__global__ __launch_bounds__(RVE_WG) void klg_fx_reverb_early(const ReverbArgs a) {
	const int k = blockIdx.y, e = blockIdx.x * RVE_WG + threadIdx.x, n = a.n;
	if (k >= a.K) return;
	const size_t KP = a.kpad;
	const float* W = a.state + k;                                           // (uniform: the tap words are scalar loads)
	const int cnt = __float_as_int(W[(size_t)RV_ECOUNT * KP]);
	const int epos0 = a.epos % RV_ESIZE;
	const float* el = a.early_rings + (size_t)k * 2 * RV_ESTRIDE;
	const float* er = el + RV_ESTRIDE;
	int wpos = epos0 + e; if (wpos >= RV_ESIZE) wpos -= RV_ESIZE;           // the early write cursor of sample e (lanes past the block's end compute a valid position and drop the result)
	const int pos = (wpos + 1 == RV_ESIZE) ? 0 : wpos + 1;                  // ... after Delay::input()
	const float at = (float)(pos - 1);
	typedef float rve_f2u __attribute__((ext_vector_type(2), aligned(4)));
	rve_f2u l[20], r[20]; float fr[20];
#pragma unroll
	for (int d = 0; d < 20; d++) {                                          // every load in flight before the first product (a tap past `cnt` repeats tap 0's address and is not summed)
		const float t = W[(size_t)(RV_ETIMES + (d < cnt ? d : 0)) * KP];
		float read = at - t;
		if (read < 0.f) read += RV_ESIZE;
		fr[d] = read - floorf(read);
		int i0 = (int)read;                                                 // i0 + 1 may be RV_ESIZE: the mirror tail holds position 0 there
		i0 = (i0 == RV_ESIZE) ? (int)RV_EZERO : i0;                         // read rounded up to SIZE (stereo_delay_tap): the tap reads zeros, not the mirror of positions 0 / 1
		l[d] = *reinterpret_cast<const rve_f2u*>(el + i0); r[d] = *reinterpret_cast<const rve_f2u*>(er + i0);
	}
	float accl = 0.f, accr = 0.f;
#pragma unroll
	for (int d = 0; d < 20; d++) {
		const float omf = 1.f - fr[d];
		const float pl = (l[d].x * omf + l[d].y * fr[d]) * W[(size_t)(RV_EGL + d) * KP];
		const float pr = (r[d].x * omf + r[d].y * fr[d]) * W[(size_t)(RV_EGR + d) * KP];
		accl = d < cnt ? accl + pl : accl;                                  // out += delay(times[d]) * gains[d], d < count
		accr = d < cnt ? accr + pr : accr;
	}
	if (e < n) { a.early_sums[((size_t)k * 2) * n + e] = accl; a.early_sums[((size_t)k * 2 + 1) * n + e] = accr; }
}

// Launch shape: a workgroup is RVQ_WG / 64 independent waves that share nothing (no barrier, a private slice of the LDS tile each).
// PRE: the early sums were computed by klg_fx_reverb_early (a.early_sums) — phase 1 is a copy into the LDS rows.
template<bool PRE>
__global__ __launch_bounds__(RVQ_WG) void klg_fx_reverb_q(const ReverbArgs a) {
	extern __shared__ float rvq_tiles[];
	const int lane = threadIdx.x & 63, inst = lane >> 4, r = lane & 15;
	const int ns = ((a.n + 3) & ~3) + 4;                                    // LDS row stride: a block length of 64 k would put every row in the same banks (the eight rows are read side by side)
	float* const rvq_tile = rvq_tiles + (threadIdx.x >> 6) * (RVQ_TILE_ROWS * ns + RVQ_XQ_FLOATS);
	const int k0 = (blockIdx.x * (RVQ_WG / 64) + (threadIdx.x >> 6)) * 4, k = k0 + inst;
	if (k0 >= (int)a.kpad) return;
	const size_t KP = a.kpad;
	const float* W = a.state + k;
#define RVW(w) W[(size_t)(w) * KP]
	const int n = a.n;
	// the caller's block of these four instances is one contiguous span of 8 n floats: 16-byte pieces, all requested before the first lands
	const bool whole = k0 + 4 <= a.K && (n & 3) == 0;                       // (else: the bank's last wave, or an odd block length — element by element)
	if (whole) {
		const rvq_v4* src = reinterpret_cast<const rvq_v4*>(a.io + (size_t)k0 * 2 * n);
		rvq_v4* dst = reinterpret_cast<rvq_v4*>(rvq_tile);
		const int q4 = n >> 2;                                                  // 16-byte pieces per row
		for (int c0 = 0; c0 < 2 * n; c0 += 8 * 64) {
			rvq_v4 x[8];
#pragma unroll
			for (int j = 0; j < 8; j++) { const int c = c0 + j * 64 + lane; x[j] = src[c < 2 * n ? c : 0]; }
#pragma unroll
			for (int j = 0; j < 8; j++) { const int c = c0 + j * 64 + lane; if (c < 2 * n) { const int R = c / q4; dst[R * (ns >> 2) + (c - R * q4)] = x[j]; } }
		}
	}
	else for (int R = 0; R < 8; R++) {
		const int ki = k0 + (R >> 1);
		for (int c = lane; c < n; c += 64) rvq_tile[R * ns + c] = (ki < a.K) ? a.io[((size_t)ki * 2 + (R & 1)) * n + c] : 0.f;
	}
	// wave-uniform cursors of the block: every instance has processed the same number of samples
	const int epos0 = __builtin_amdgcn_readfirstlane(a.epos % RV_ESIZE), fpos0 = __builtin_amdgcn_readfirstlane(a.fpos % RV_FSIZE);

	// =========== phase 1: the early sums of the block, lane = sample ===========
	// EarlyReflections::process Reverb.k:90-92 with Stereo::Delay::tap(float) klang.h:4668-4681 (both channels read at one cursor).
	float* const r1_rows = rvq_tile + 9 * ns;
	if constexpr (PRE) {
		if (whole) {                                                            // [k0 .. k0 + 3][2][n]: one contiguous span, like the caller's block
			const rvq_v4* src = reinterpret_cast<const rvq_v4*>(a.early_sums + (size_t)k0 * 2 * n);
			rvq_v4* dst = reinterpret_cast<rvq_v4*>(r1_rows);
			const int q4 = n >> 2;
			for (int c0 = 0; c0 < 2 * n; c0 += 8 * 64) {
				rvq_v4 x[8];
#pragma unroll
				for (int j = 0; j < 8; j++) { const int c = c0 + j * 64 + lane; x[j] = src[c < 2 * n ? c : 0]; }
#pragma unroll
				for (int j = 0; j < 8; j++) { const int c = c0 + j * 64 + lane; if (c < 2 * n) { const int R = c / q4; dst[R * (ns >> 2) + (c - R * q4)] = x[j]; } }
			}
		}
		else for (int R = 0; R < 8; R++) {
			const int ki = k0 + (R >> 1);
			for (int c = lane; c < n; c += 64) r1_rows[R * ns + c] = (ki < a.K) ? a.early_sums[((size_t)ki * 2 + (R & 1)) * n + c] : 0.f;
		}
	}
	else {
		typedef float rvq_f2u __attribute__((ext_vector_type(2), aligned(4)));
		struct Taps { rvq_f2u l[20], r[20]; float fr[20]; };                    // per tap: positions i0, i0 + 1 of both lines (one 8-byte load each) and the fraction
		const int chunks = (n + 63) >> 6, groups = 4 * chunks;                  // a group = 64 consecutive samples of one instance
		// the taps' words (time, left gain, right gain; count) of the four instances wait in lanes: tap d of instance i in lane 16 i + d (d < 16)
		// of the first register, lane 16 i + d - 16 of the second; v_readlane hands one to the whole wave (no memory in the loop)
		const int pd = lane & 15;
		const float* Wp = a.state + (k0 + inst);
		const float tA = Wp[(size_t)(RV_ETIMES + pd) * KP], glA = Wp[(size_t)(RV_EGL + pd) * KP], grA = Wp[(size_t)(RV_EGR + pd) * KP];
		const float tB = Wp[(size_t)(RV_ETIMES + 16 + (pd & 3)) * KP], glB = Wp[(size_t)(RV_EGL + 16 + (pd & 3)) * KP], grB = Wp[(size_t)(RV_EGR + 16 + (pd & 3)) * KP];
		const int cntv = __float_as_int(Wp[(size_t)RV_ECOUNT * KP]);
		auto word = [&](float A, float B, int i, int d) __attribute__((always_inline)) {
			return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d < 16 ? A : B), 16 * i + (d & 15)));
		};
		// the loads of a group are all in flight before its first sum, and the next group's loads are issued before this group's sums
		auto issue = [&](int g, Taps& T) __attribute__((always_inline)) {
			const int i = g / chunks, e = (g - i * chunks) * 64 + lane;         // instance k0 + i (uniform)
			const int cnt = __builtin_amdgcn_readlane(cntv, 16 * i);
			const float* el = a.early_rings + (size_t)(k0 + i) * 2 * RV_ESTRIDE;
			const float* er = el + RV_ESTRIDE;
			int wpos = epos0 + e; if (wpos >= RV_ESIZE) wpos -= RV_ESIZE;       // the early write cursor of sample e (lanes past the block's end compute a valid position and drop the result)
			const int pos = (wpos + 1 == RV_ESIZE) ? 0 : wpos + 1;              // ... after Delay::input()
			const float at = (float)(pos - 1);
#pragma unroll
			for (int d = 0; d < 20; d++) if (d < 10 || d < cnt) {               // (Reverb.k:26: ten to twenty taps by the room-size dial; a tap the instance does not have is neither fetched nor summed — the test is wave-uniform)
				float read = at - word(tA, tB, i, d);
				if (read < 0.f) read += RV_ESIZE;
				T.fr[d] = read - floorf(read);
				int i0 = (int)read;                                             // i0 + 1 may be RV_ESIZE: the mirror tail holds position 0 there
				i0 = (i0 == RV_ESIZE) ? (int)RV_EZERO : i0;                     // read rounded up to SIZE (stereo_delay_tap): the tap reads zeros
				T.l[d] = *reinterpret_cast<const rvq_f2u*>(el + i0); T.r[d] = *reinterpret_cast<const rvq_f2u*>(er + i0);
			}
		};
		auto finish = [&](int g, const Taps& T) __attribute__((always_inline)) {
			const int i = g / chunks, e = (g - i * chunks) * 64 + lane;
			const int cnt = __builtin_amdgcn_readlane(cntv, 16 * i);
			float accl = 0.f, accr = 0.f;
#pragma unroll
			for (int d = 0; d < 20; d++) {
				const float omf = 1.f - T.fr[d];
				const float pl = (T.l[d].x * omf + T.l[d].y * T.fr[d]) * word(glA, glB, i, d);
				const float pr = (T.r[d].x * omf + T.r[d].y * T.fr[d]) * word(grA, grB, i, d);
				accl = d < cnt ? accl + pl : accl;                              // out += delay(times[d]) * gains[d], d < count
				accr = d < cnt ? accr + pr : accr;
			}
			if (e < n) { r1_rows[(i * 2) * ns + e] = accl; r1_rows[(i * 2 + 1) * ns + e] = accr; }
		};
		Taps T0, T1;
		issue(0, T0);
		int g = 0;
		for (; g + 2 < groups; g += 2) {                                        // (groups is a multiple of four)
			issue(g + 1, T1); __builtin_amdgcn_sched_barrier(0); finish(g, T0);
			issue(g + 2, T0); __builtin_amdgcn_sched_barrier(0); finish(g + 1, T1);
		}
		issue(g + 1, T1); __builtin_amdgcn_sched_barrier(0); finish(g, T0); finish(g + 1, T1);
	}

	// =========== phase 2: the recursive part, lane = (instance, slot) ===========
	// ---- this lane's FilteredDelay ----
	const int grp = r >> 2, kk = r & 3, cf = grp < 2 ? 1 : 0;               // mid[] works on sample t + 1, late[] on t
	const int fw = RV_FD + r * FD_WORDS;
	Biquad ff = { RVW(fw + FD_COEF + 0), RVW(fw + FD_COEF + 1), RVW(fw + FD_COEF + 2), RVW(fw + FD_COEF + 3), RVW(fw + FD_COEF + 4), RVW(fw + FD_Z0), RVW(fw + FD_Z1) };
	float fin = RVW(fw + FD_IN);
	const float fgain = RVW(fw + FD_GAIN), ffrac = RVW(fw + FD_LASTF);
	const int flast = __float_as_int(RVW(fw + FD_LASTP));
	float* const fline = a.fd_rings + ((size_t)k * 16 + r) * RV_FSTRIDE;    // this (instance, line)'s own ring (+ mirror tail)
	float* const fquad = a.fd_rings + ((size_t)k * 16 + (r & 12)) * RV_FSTRIDE + 4 * (r & 3);   // quarter r % 4 of a 64-byte piece of the quad's first line (see the steady batch's stores)
	rvq_v4* const xq_mine = reinterpret_cast<rvq_v4*>(rvq_tile + RVQ_TILE_ROWS * ns + lane * RVQ_XQ_LD);                         // the quarter exchange: a lane's 16 floats, rows padded against bank conflicts
	const rvq_v4* const xq_quad = reinterpret_cast<const rvq_v4*>(rvq_tile + RVQ_TILE_ROWS * ns + (lane & 60) * RVQ_XQ_LD + 4 * (r & 3));   // ... quarter r % 4 of the quad's lane 0 (lane v: + v rows)
	// row kk of the FDN matrix (Reverb.k:158-161): products are summed left to right
	const float m0 = kk == 0 ? 0.f : kk == 3 ? 1.f : -1.f, m1 = kk == 1 ? 0.f : kk == 3 ? -1.f : 1.f, m2 = kk == 2 ? 0.f : kk == 1 ? -1.f : 1.f, m3 = kk == 3 ? 0.f : kk == 1 ? 1.f : -1.f;
	// ---- the early filter of channel ech (every lane of the channel runs it on the same input: eight copies of one state; lane r % 8 == 0 stores) ----
	const int ech = r >> 3;
	float* const eline = a.early_rings + ((size_t)k * 2 + ech) * RV_ESTRIDE;  // this (instance, channel)'s own early ring (+ mirror tail)
	const bool efilter = (r & 7) == 0;                                      // in >> lpf >> hpf >> delay for channel ech (Reverb.k:88)
	const float lb0 = RVW(RV_ELPF + 0), lb1 = RVW(RV_ELPF + 1), lb2 = RVW(RV_ELPF + 2), la1 = RVW(RV_ELPF + 3), la2 = RVW(RV_ELPF + 4);
	const float hb0 = RVW(RV_EHPF + 0), hb1 = RVW(RV_EHPF + 1), hb2 = RVW(RV_EHPF + 2), ha1 = RVW(RV_EHPF + 3), ha2 = RVW(RV_EHPF + 4);
	float elz0 = RVW(RV_EZ + 2 * ech), elz1 = RVW(RV_EZ + 2 * ech + 1), ehz0 = RVW(RV_EZ + 4 + 2 * ech), ehz1 = RVW(RV_EZ + 4 + 2 * ech + 1);
	const bool outlane = r == 8 || r == 12;                                 // the output sample of channel och (a lane of late[och]'s quad)
	const int och = r == 12 ? 1 : 0;
	const float dry = RVW(RV_CTL + 0), c1 = RVW(RV_CTL + 1), c2 = RVW(RV_CTL + 2), c3 = RVW(RV_CTL + 3), wet = RVW(RV_CTL + 4);
	const float wet_o = och ? 0.f : wet;
	float* const in_e = rvq_tile + (inst * 2 + ech) * ns;                     // the filter lane's input row
	float* const io_o = rvq_tile + (inst * 2 + och) * ns;                     // the output lane's row
	float* const io_w = outlane ? io_o : rvq_tile + 8 * ns;                  // ... and where a lane stores "its" output sample: an unconditional ds_write, no exec-mask branch in the sample loop
	// the early sum a lane wants: mid[c] (lanes 4c .. 4c + 3) of its own sample t + 1, the output lane of channel c (8 / 12) of sample t - 1
	const float* const r1_row = r1_rows + (inst * 2 + ((r < 4 || r == 8) ? 0 : 1)) * ns;
	const int r1_at = r < 8 ? 1 : -1;

	// ---- row requests, one batch ahead ----
	// The batches sit on a grid: a steady batch's ring stores are whole aligned 64-byte (FilteredDelay: 16 positions) / 32-byte (early line:
	// 8 positions) pieces — an unaligned piece is two partial sectors to the memory system, twice the write traffic (measured: WRITE_SIZE
	// 2 x the bytes, and 60 of 160 us at 4096 instances).  Both cursors advance with the sample count (2 per sample / 1 per sample) and both
	// ring sizes are multiples of the piece: a batch that starts at an iteration t with (fpos0 / 2 + t) % 8 == 0 has late[]'s first pair at
	// a multiple of 16 and the early cursor at a multiple of 8.  t_s = the first such iteration >= 1 (steady batches need every stage
	// active); two guarded batches before it cover the ramp-up t = -2 .. t_s - 1 (the first of them may have nothing to do).
	const int t_s = ((8 - ((fpos0 >> 1) & 7)) & 7) ? ((8 - ((fpos0 >> 1) & 7)) & 7) : 8;
	const int t_first = t_s - 2 * RVQ_B;
	int fnext = flast + 2 * (t_first + cf) + 1; if (fnext < 0) fnext += RV_FSIZE;   // row J + 1 of the first batch (its first sample: s = t_first + cf)
	auto request = [&](auto set) __attribute__((always_inline)) {
		rvq_request<decltype(set)::value>(fline + fnext);                    // (never past the mirror tail: fnext < RV_FSIZE, 16 rows)
		fnext += 2 * RVQ_B; if (fnext >= RV_FSIZE) fnext -= RV_FSIZE;
	};
	const IntTag<0> A; const IntTag<1> Bn; const IntTag<2> Cn;              // three sets of accumulation registers in rotation: a batch's rows are requested TWO batches before it runs
	float fr0;                                                              // row `last` of the FilteredDelay's current sample ( = row last + 2 of the previous one)
	{ int p0 = flast + 2 * (t_first + cf); if (p0 < 0) p0 += RV_FSIZE; fr0 = fline[p0]; }
	request(A);
	wave_sync();                                                            // the io tile and the early sums are in LDS

	auto inside = [&](int s) { return s < 0 ? 0 : s >= n ? n - 1 : s; };
	float ssum_prev = 0.f, lr_prev = 0.f;
	// What a batch reads from LDS — the filter lane's input samples, the output lane's dry samples, the early sums — is known for the whole
	// block: a batch's 24 values are read while the batch before it runs (three sets in rotation, like the rows), so no iteration waits for LDS.
	struct RvqLds { float xin[RVQ_B], xout[RVQ_B], r1[RVQ_B]; };
	auto fetch = [&](auto guarded, RvqLds& L, const int tb) __attribute__((always_inline)) {   // for the batch of iterations tb .. tb + 7
		constexpr bool G = decltype(guarded)::value;
#pragma unroll
		for (int u = 0; u < RVQ_B; u++) {
			const int t = tb + u;
			L.xin[u] = in_e[G ? inside(t + 2) : t + 2];                         // sample e = t + 2 (read ahead of the output samples written in place: o = t - 1)
			L.xout[u] = io_o[G ? inside(t - 1) : t - 1];
			L.r1[u] = r1_row[G ? inside(t + r1_at) : t + r1_at];
		}
	};
	RvqLds LA, LB, LC;
	// What a steady batch WRITES is collected in registers and stored once per batch — 64 bytes per FilteredDelay line, 32 per early line:
	// whole sectors instead of eight 8-byte (4-byte) pieces of one, each of which the memory system would otherwise merge on its own.
	// late[] (cf = 0) collects the pairs of iterations u = 0 .. 7 — positions fbase(t0) .. + 15; mid[] writes one sample ahead, so the SAME
	// positions hold ITS pairs of iterations u = -1 .. 6: it stores the previous iteration's pair (pp0, pp1) in slot u.  The early line's
	// aligned eight are the samples of iterations u = -2 .. 5 (slot (u + 2) % 8, stored after u = 5).
	float Wf[2 * RVQ_B], We[RVQ_B], pp0 = 0.f, pp1 = 0.f;
#pragma unroll
	for (int j = 0; j < 2 * RVQ_B; j++) Wf[j] = 0.f;
#pragma unroll
	for (int j = 0; j < RVQ_B; j++) We[j] = 0.f;

	// One iteration: early filter of sample t + 2, mid[] of t + 1, late[] of t, output of t - 1; u = its place in the batch (compile-time: every
	// row is a named register).  G = guarded (the batches at the edges of the block test which stages are active).
	// A wave is alone on its SIMD at the bank sizes that matter, so nothing hides a wait: every cross-lane / LDS value is requested at the
	// top of the iteration before the one that uses it, or at the top of this one with a long computation in between.
	// A steady batch is ONE basic block of eight samples (no lane-predicated branch: what only some lanes need is computed by all of
	// them — a masked-off lane costs the same issue slot), so the scheduler can fill the latency of one sample's LDS / bpermute answers with
	// the arithmetic of its neighbours.
	// A batch's MODE says, per stage, whether its lanes test if their sample lies inside the block and how what they write reaches the rings:
	//   fd (FilteredDelays): 0 every lane runs, pairs collected, the 64-byte piece flushed at u = 7 (steady) | 1 tested, pairs stored one by one (an edge off the grid)
	//                        | 2 tested, collected and flushed (an edge ON the grid: the piece is complete) | 3 tested, nothing stored (the iterations before an aligned block: only
	//                        mid[]'s first pair exists, and it travels in pp to the next batch's slot 0)
	//                        | 4 / 5 every lane runs, 6 / 7 tested: the FIRST / SECOND batch of a PAIR whose two 64-byte pieces leave together as one aligned 128-byte piece (below)
	//   e (early filter):    0 runs, collected, flushed at u = 5 | 1 tested, stored one by one | 2 tested, collected and flushed | 3 tested, collected only
	//   o (output):          0 runs | 1 tested
	// (8-byte pair stores are not merged on their way to memory: every one is a 32-byte write — 11 - 14 KB per instance and block at the edges,
	//  profiles/r03_pmc/reverb_q_traffic_vs_block_length.jsonl; on the grid the edges are whole pieces too.)
	auto step = [&](auto mode, auto place, const int t, const RvqRows& X, const RvqLds& L) __attribute__((always_inline)) {
		constexpr int MFD = decltype(mode)::fd, ME = decltype(mode)::e, MO = decltype(mode)::o;
		constexpr int u = decltype(place)::value;
		const int e = t + 2, sfd = t + cf, o = t - 1;
		const bool e_on = ME == 0 || (e >= 0 && e < n), fd_on = MFD == 0 || MFD == 4 || MFD == 5 || (sfd >= 0 && sfd < n), o_on = MO == 0 || (o >= 0 && o < n);
		int ewpos = epos0 + e; if (ewpos >= RV_ESIZE) ewpos -= RV_ESIZE; if (ewpos < 0) ewpos += RV_ESIZE;   // uniform: the early write cursor of sample e
		// ---- requests whose answers are needed later in this iteration / in the next one ----
		const float from_mid = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ssum_prev), 0x118, 0xF, 0xF, true));   // late[]'s input: mid[]'s sum of the previous iteration (DPP row_shr:8 — lane L takes lane L - 8 of its row of 16)
		const float x_in_now = L.xin[u], x_out_now = L.xout[u], r1_now = L.r1[u];
		// ---- the FilteredDelay: mid[] on sample t + 1 (input: the early reflections of its channel), late[] on sample t (input: mid[]'s sum) ----
		float r0 = fr0; if constexpr (u > 0) r0 = X.template F<(u > 0 ? 2 * u - 1 : 0)>();
		const float r1v = X.template F<2 * u>(), r2v = X.template F<2 * u + 1>();
		const float fdt1 = r0 + ffrac * (r1v - r0), fdt2 = r1v + ffrac * (r2v - r1v);     // the two delay reads of this sample (Delay::operator>> klang.h:3491-3500)
		float ssum = 0.f, lr_in = 0.f;
		const int fbase = (fpos0 + 2 * t >= RV_FSIZE) ? fpos0 + 2 * t - RV_FSIZE : fpos0 + 2 * t;   // uniform: late[]'s write cursor (mid[] is one sample = 2 ahead)
		float wa = pp0, wb = pp1;                                                // this slot's pair: mid[] (one sample ahead) stores the PREVIOUS iteration's, late[] this one's
		if (fd_on) {
			const float dl = biquad_process(ff, fdt1) * fgain;                // signals<4> delays = { delay[0..3] }: first process() — (in >> delay >> filter) * gain, Reverb.k:130-132
			const float d0 = quad_bcast<0>(dl), d1 = quad_bcast<1>(dl), d2 = quad_bcast<2>(dl), d3 = quad_bcast<3>(dl);
			const float fb = m0 * d0 + m1 * d1 + m2 * d2 + m3 * d3;           // (delays >> matrix): row kk, products summed left to right (klang.h:1462-1467)
			lr_in = r < 8 ? r1_now : from_mid;
			const float fin_new = fb + lr_in;                                 // fb[k] = ... + in;  fb[k] >> delay[k]
			if constexpr (MFD == 1) {
				int fwpos = fbase + 2 * cf; if (fwpos >= RV_FSIZE) fwpos -= RV_FSIZE; if (fwpos < 0) fwpos += RV_FSIZE;
				const rvq_f2 pair = { fin, fin_new };                         // the two inputs of this sample (fwpos is even: both in one 8-byte store)
				*reinterpret_cast<rvq_f2*>(fline + fwpos) = pair;
				if (fbase < RV_FPAD || fbase + 2 >= RV_FSIZE || fbase < 0) { if (fwpos < RV_FPAD) *reinterpret_cast<rvq_f2*>(fline + fwpos + RV_FSIZE) = pair; }   // mirror (the outer test is uniform and almost never true)
			}
			wa = cf ? pp0 : fin; wb = cf ? pp1 : fin_new;
			pp0 = fin; pp1 = fin_new;
			fin = fin_new;
			const float o2 = biquad_process(ff, fdt2) * fgain;                // the `+` chain processes each FilteredDelay a second time
			const float q0 = quad_bcast<0>(o2), q1 = quad_bcast<1>(o2), q2 = quad_bcast<2>(o2), q3 = quad_bcast<3>(o2);
			ssum = q3 + (q2 + (q0 + q1));                                     // ((o0 + o1) + o2) + o3 as the reference's `+` chain associates
		}
		if constexpr (MFD == 0 || MFD == 2) {                                 // every lane, also one whose own sample lies outside the block (mid[] in a block's last iteration: its slot is pp)
			Wf[2 * u] = wa; Wf[2 * u + 1] = wb;                               // stored with the rest of the batch (flush below)
			if constexpr (u == RVQ_B - 1) {
				const int w0 = fbase - 2 * (RVQ_B - 1);                   // uniform and a multiple of 16 (the grid): the batch's piece of every line, never across the ring's end
				// A lane holds the 64 bytes of ITS line; stored as they are, every 16-byte quarter is a write request of its own to the L2
				// (64 lanes, 64 lines: nothing to merge — 10.7 M requests per block at 4096 instances, the kernel's bound).  The four
				// lanes of a quad swap quarters through LDS instead: store v of lane i is quarter i of the quad's line v, so a quad
				// writes 64 contiguous bytes per instruction — one request.
				// The stores are NONTEMPORAL: a piece is half of a 128-byte L2 line whose other half comes a batch later, and nobody
				// reads either for milliseconds.  Kept as ordinary dirty lines under this kernel's read stream, a third of them went to memory twice
				// (WRITE_SIZE 200 MB for 150 MB of distinct bytes at 4,096 instances; 154 MB with the hint — tools/rvq_ablate.sh,
				// profiles/r03_pmc/reverb_q_writes.jsonl).
#pragma unroll
				for (int v = 0; v < 4; v++) { const rvq_v4 x = { Wf[4 * v], Wf[4 * v + 1], Wf[4 * v + 2], Wf[4 * v + 3] }; xq_mine[v] = x; }
				wave_sync();
				rvq_v4 quarter[4];
#pragma unroll
				for (int v = 0; v < 4; v++) quarter[v] = xq_quad[v * (RVQ_XQ_LD / 4)];
				wave_sync();                                              // (the next batch's writes come after these reads)
#pragma unroll
				for (int v = 0; v < 4; v++) if (!(KLG_RVQ_ABLATE & 1)) __builtin_nontemporal_store(quarter[v], reinterpret_cast<rvq_v4*>(fquad + (size_t)v * RV_FSTRIDE + w0));
				if (w0 < RV_FPAD) {                                       // the mirrored head (two batches per lap of the ring)
#pragma unroll
					for (int v = 0; v < 4; v++) *reinterpret_cast<rvq_v4*>(fquad + (size_t)v * RV_FSTRIDE + w0 + RV_FSIZE) = quarter[v];
				}
			}
		}
		if constexpr (MFD >= 4) {
			// PAIRS: a 64-byte piece is half of a 128-byte line, and the memory system moves this kernel's pattern — 65,536 lines, each touched one piece at a
			// time — at 4.0 TB/s with 64-byte pieces and 5.0 TB/s with 128-byte ones (tools/calib/piece_bw.hip, profiles/r03_pmc/reverb_q_piece_size.jsonl).  The first
			// batch of a pair leaves its lanes' sixteen floats in LDS; the second adds its own and the quad writes both halves of every line back to back.
			Wf[2 * u] = wa; Wf[2 * u + 1] = wb;
			if constexpr (u == RVQ_B - 1) {
				constexpr int H = (MFD == 5 || MFD == 7) ? 1 : 0;
#pragma unroll
				for (int v = 0; v < 4; v++) { const rvq_v4 x = { Wf[4 * v], Wf[4 * v + 1], Wf[4 * v + 2], Wf[4 * v + 3] }; xq_mine[4 * H + v] = x; }
				if constexpr (H == 1) {
					const int w0 = fbase - 2 * (RVQ_B - 1) - 2 * RVQ_B;       // uniform and a multiple of 32: the pair's piece of every line
					wave_sync();
					rvq_v4 quarter[4][2];
#pragma unroll
					for (int v = 0; v < 4; v++) { quarter[v][0] = xq_quad[v * (RVQ_XQ_LD / 4)]; quarter[v][1] = xq_quad[v * (RVQ_XQ_LD / 4) + 4]; }
					wave_sync();                                              // (the next pair's writes come after these reads)
#pragma unroll
					// (ordinary stores: the 128-byte line is dirty as a whole, so it leaves the L2 once — WRITE_SIZE 151 MB for 150 MB of distinct bytes at 4,096
					//  instances; with the nontemporal hint the single pieces need, a pair counted 171 MB at the same speed)
					for (int v = 0; v < 4; v++) if (!(KLG_RVQ_ABLATE & 1)) {
						*reinterpret_cast<rvq_v4*>(fquad + (size_t)v * RV_FSTRIDE + w0) = quarter[v][0];
						*reinterpret_cast<rvq_v4*>(fquad + (size_t)v * RV_FSTRIDE + w0 + 2 * RVQ_B) = quarter[v][1];
					}
					if (w0 < RV_FPAD) {                                       // the mirrored head (once per lap of the ring)
#pragma unroll
						for (int v = 0; v < 4; v++) { *reinterpret_cast<rvq_v4*>(fquad + (size_t)v * RV_FSTRIDE + w0 + RV_FSIZE) = quarter[v][0]; *reinterpret_cast<rvq_v4*>(fquad + (size_t)v * RV_FSTRIDE + w0 + 2 * RVQ_B + RV_FSIZE) = quarter[v][1]; }
					}
				}
			}
		}
		if constexpr (u == RVQ_B - 1) fr0 = r2v;                             // row last + 2 of the batch's last sample is row `last` of the next batch's first
		// ---- early stage, sample e: in >> lpf >> hpf >> delay  Reverb.k:88 ----
		if (e_on) {
			Biquad elpf = { lb0, lb1, lb2, la1, la2, elz0, elz1 }, ehpf = { hb0, hb1, hb2, ha1, ha2, ehz0, ehz1 };
			const float y = biquad_process(ehpf, biquad_process(elpf, x_in_now));
			elz0 = elpf.z0; elz1 = elpf.z1; ehz0 = ehpf.z0; ehz1 = ehpf.z1;
			We[(u + 2) & (RVQ_B - 1)] = y;
			if constexpr (ME == 1) {
				if (efilter) {
					eline[ewpos] = y;
					if (ewpos < RV_EMIRROR) eline[ewpos + RV_ESIZE] = y;      // mirror (a uniform test)
				}
			}
			else if constexpr (ME != 3 && u == RVQ_B - 3) {
				if (efilter) {
					const int w0 = ewpos - (RVQ_B - 1);                       // uniform and a multiple of 8 (the grid)
					typedef float rvq_v4e __attribute__((ext_vector_type(4), aligned(16)));
					rvq_v4e* const dst = reinterpret_cast<rvq_v4e*>(eline + w0);
#pragma unroll
					for (int v = 0; v < RVQ_B / 4; v++) { const rvq_v4e x = { We[4 * v], We[4 * v + 1], We[4 * v + 2], We[4 * v + 3] }; if (!(KLG_RVQ_ABLATE & 2)) __builtin_nontemporal_store(x, dst + v); }
					if (w0 < RV_EMIRROR) {
#pragma unroll
						for (int j = 0; j < RVQ_B; j++) eline[w0 + j + RV_ESIZE] = We[j];
					}
				}
			}
		}
		// ---- output, sample o: Reflections::process + Reverb::process ----
		if (o_on) {
			const float refl = (r1_now * c1 + lr_prev * c2) + ssum_prev * c3;     // r1 of sample o, r2 = this late[]'s input and r3 = its sum of the previous iteration
			io_w[o] = x_out_now * dry + refl * wet_o;                             // wet side is signals<2>{ wet, 0 }
		}
		ssum_prev = ssum; lr_prev = lr_in;
	};
	// a batch: request the rows of the batch after next (into the set the previous batch has used), read the next batch's LDS values, run
	auto batch = [&](auto mode, auto guarded_next, const int t0, auto set, const RvqLds& L, auto set_req, RvqLds& Lnext) __attribute__((always_inline)) {
		request(set_req);
		fetch(guarded_next, Lnext, t0 + RVQ_B);
		RvqRows X;
		rvq_await<decltype(set)::value, 8>(X);
		step(mode, IntTag<0>(), t0 + 0, X, L); step(mode, IntTag<1>(), t0 + 1, X, L); step(mode, IntTag<2>(), t0 + 2, X, L); step(mode, IntTag<3>(), t0 + 3, X, L);
		step(mode, IntTag<4>(), t0 + 4, X, L); step(mode, IntTag<5>(), t0 + 5, X, L); step(mode, IntTag<6>(), t0 + 6, X, L); step(mode, IntTag<7>(), t0 + 7, X, L);
	};
	const BoolTag<true> gfetch; const BoolTag<false> sfetch;                    // the next batch's LDS values: indices clamped to the block / as they are
	const RvqMode<1, 1, 1> ramp; const RvqMode<0, 0, 0> steady;
	const RvqMode<3, 3, 1> lead_in; const RvqMode<2, 2, 1> edge;               // the two kinds of edge batch of a block that starts and ends on the grid
	const RvqMode<4, 0, 0> steady_1st; const RvqMode<5, 0, 0> steady_2nd; const RvqMode<6, 2, 1> edge_1st; const RvqMode<7, 2, 1> edge_2nd;   // ... whose pieces leave in pairs (128 bytes per line)
	// iterations t = -2 .. n in batches of eight; the rows of a batch are requested two batches ahead (a wave alone on its SIMD has nothing
	// but distance to hide HBM latency with; a batch is about a microsecond).  The sched_barrier: the rows are REQUESTED there, not where used.
	// (A request past the block's end reads rows that exist and are not used.)
	int t0 = t_first;
	request(Bn); fetch(gfetch, LA, t0);
	// A block that starts and ends ON the grid (both cursors at a multiple of the piece, a whole number of batches — the usual case: blocks of 64 / 256 /
	// 1024 samples from a host that always asks for the same length) has no ragged edge: the batch of iterations -8 .. -1 only produces mid[]'s first pair
	// and the early filter's first two samples (carried over in registers), the batches 0 .. 7 and n - 8 .. n - 1 hold complete pieces and store them whole.
	const int steady_batches = n / RVQ_B - 2;
	if (t_s == RVQ_B && ((fpos0 >> 1) & (2 * RVQ_B - 1)) == 0 && n % (2 * RVQ_B) == 0 && steady_batches >= 2) {
		// ... and on the grid of PAIRS (any block of a multiple of sixteen samples that starts on it): batch 0 (an edge) is the first of a pair, the steady batches
		// alternate second / first — with the three register sets in rotation a turn is six batches, and what is left of them (none, two or four) is written out
		// with the sets it finds —, the closing edge batch is a second.
		const int turns = steady_batches / 6, rest = steady_batches - 6 * turns;        // rest: 0, 2 or 4 (an even number of batches)
		batch(lead_in, gfetch, t0, A, LA, Cn, LB); t0 += RVQ_B;                 // t = -8 .. -1
		batch(edge_1st, sfetch, t0, Bn, LB, A, LC); t0 += RVQ_B;                // t = 0 .. 7
		for (int turn = turns; turn > 0; turn--, t0 += 6 * RVQ_B) {
			batch(steady_2nd, sfetch, t0, Cn, LC, Bn, LA);
			batch(steady_1st, sfetch, t0 + RVQ_B, A, LA, Cn, LB);
			batch(steady_2nd, sfetch, t0 + 2 * RVQ_B, Bn, LB, A, LC);
			batch(steady_1st, sfetch, t0 + 3 * RVQ_B, Cn, LC, Bn, LA);
			batch(steady_2nd, sfetch, t0 + 4 * RVQ_B, A, LA, Cn, LB);
			if (turn > 1 || rest) batch(steady_1st, sfetch, t0 + 5 * RVQ_B, Bn, LB, A, LC);
			else batch(steady_1st, gfetch, t0 + 5 * RVQ_B, Bn, LB, A, LC);     // (what follows the last steady batch is guarded)
		}
		if (rest == 0) {
			batch(edge_2nd, gfetch, t0, Cn, LC, Bn, LA);                        // t = n - 8 .. n - 1
			batch(ramp, gfetch, t0 + RVQ_B, A, LA, Cn, LB);                     // t = n: the last output sample
		}
		else if (rest == 2) {
			batch(steady_2nd, sfetch, t0, Cn, LC, Bn, LA);
			batch(steady_1st, gfetch, t0 + RVQ_B, A, LA, Cn, LB);
			batch(edge_2nd, gfetch, t0 + 2 * RVQ_B, Bn, LB, A, LC);
			batch(ramp, gfetch, t0 + 3 * RVQ_B, Cn, LC, Bn, LA);
		}
		else {
			batch(steady_2nd, sfetch, t0, Cn, LC, Bn, LA);
			batch(steady_1st, sfetch, t0 + RVQ_B, A, LA, Cn, LB);
			batch(steady_2nd, sfetch, t0 + 2 * RVQ_B, Bn, LB, A, LC);
			batch(steady_1st, gfetch, t0 + 3 * RVQ_B, Cn, LC, Bn, LA);
			batch(edge_2nd, gfetch, t0 + 4 * RVQ_B, A, LA, Cn, LB);
			batch(ramp, gfetch, t0 + 5 * RVQ_B, Bn, LB, A, LC);
		}
		t0 = n + 1;
	}
	else if (t_s == RVQ_B && n % RVQ_B == 0 && steady_batches >= 3 && steady_batches % 3 == 0) {
		batch(lead_in, gfetch, t0, A, LA, Cn, LB); t0 += RVQ_B;                 // t = -8 .. -1
		batch(edge, sfetch, t0, Bn, LB, A, LC); t0 += RVQ_B;                    // t = 0 .. 7: every FilteredDelay and early sample of the piece is inside the block; only the output of t = 0 is not
		for (int turn = steady_batches / 3; turn > 1; turn--, t0 += 3 * RVQ_B) {
			batch(steady, sfetch, t0, Cn, LC, Bn, LA);
			batch(steady, sfetch, t0 + RVQ_B, A, LA, Cn, LB);
			batch(steady, sfetch, t0 + 2 * RVQ_B, Bn, LB, A, LC);
		}
		batch(steady, sfetch, t0, Cn, LC, Bn, LA);
		batch(steady, sfetch, t0 + RVQ_B, A, LA, Cn, LB);
		batch(steady, gfetch, t0 + 2 * RVQ_B, Bn, LB, A, LC); t0 += 3 * RVQ_B;
		batch(edge, gfetch, t0, Cn, LC, Bn, LA); t0 += RVQ_B;                   // t = n - 8 .. n - 1: mid[]'s last slot is the pair of t = n - 2, the early piece is complete at t = n - 3
		batch(ramp, gfetch, t0, A, LA, Cn, LB); t0 += RVQ_B;                    // t = n: the last output sample
		t0 = n + 1;
	}
	else {
	batch(ramp, gfetch, t0, A, LA, Cn, LB); t0 += RVQ_B;                        // (iterations before t = -2 find every stage off)
	batch(ramp, gfetch, t0, Bn, LB, A, LC); t0 += RVQ_B;                        // ... up to t_s - 1
	if (t0 + 3 * RVQ_B - 1 <= n - 3) {                                          // three steady batches per turn (t >= 1 and t + 2 < n throughout): the sets swap roles
		for (; t0 + 6 * RVQ_B - 1 <= n - 3; t0 += 3 * RVQ_B) {                  // ... while a whole steady turn follows
			batch(steady, sfetch, t0, Cn, LC, Bn, LA);
			batch(steady, sfetch, t0 + RVQ_B, A, LA, Cn, LB);
			batch(steady, sfetch, t0 + 2 * RVQ_B, Bn, LB, A, LC);
		}
		batch(steady, sfetch, t0, Cn, LC, Bn, LA);                              // the last steady turn: what follows it is guarded
		batch(steady, sfetch, t0 + RVQ_B, A, LA, Cn, LB);
		batch(steady, gfetch, t0 + 2 * RVQ_B, Bn, LB, A, LC);
		t0 += 3 * RVQ_B;
		// what the steady batches still hold: mid[]'s pair of the last iteration, the early samples of the last two
		{
			int fb = fpos0 + 2 * (t0 - 1); while (fb >= RV_FSIZE) fb -= RV_FSIZE;                   // late[]'s cursor of iteration t0 - 1; mid[]'s is one sample ahead
			int w = fb + 2; if (w >= RV_FSIZE) w -= RV_FSIZE;
			if (cf) {
				const rvq_f2 pair = { pp0, pp1 };
				*reinterpret_cast<rvq_f2*>(fline + w) = pair;
				if (w < RV_FPAD) *reinterpret_cast<rvq_f2*>(fline + w + RV_FSIZE) = pair;
			}
			int ew = epos0 + t0; while (ew >= RV_ESIZE) ew -= RV_ESIZE;                             // the early cursor of iteration t0 - 2 (sample t0), a multiple of 8
			if (efilter) {
				eline[ew] = We[0]; eline[ew + 1] = We[1];
				if (ew < RV_EMIRROR) { eline[ew + RV_ESIZE] = We[0]; eline[ew + 1 + RV_ESIZE] = We[1]; }
			}
		}
	}
	// the last batches, guarded (the roles of the sets stay compile-time: no array ever lives in memory); at most two of them could have been steady
	if (t0 <= n) { batch(ramp, gfetch, t0, Cn, LC, Bn, LA); t0 += RVQ_B; }
	if (t0 <= n) { batch(ramp, gfetch, t0, A, LA, Cn, LB); t0 += RVQ_B; }
	if (t0 <= n) { batch(ramp, gfetch, t0, Bn, LB, A, LC); t0 += RVQ_B; }
	if (t0 <= n) { batch(ramp, gfetch, t0, Cn, LC, Bn, LA); t0 += RVQ_B; }
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // (requests past the block's end land in accumulation registers nobody reads)
	wave_sync();
	if (KLG_RVQ_ABLATE & 4) {}
	else if (whole) {
		rvq_v4* dst = reinterpret_cast<rvq_v4*>(a.io + (size_t)k0 * 2 * n);
		const rvq_v4* src = reinterpret_cast<const rvq_v4*>(rvq_tile);
		const int q4 = n >> 2;
		for (int c = lane; c < 2 * n; c += 64) { const int R = c / q4; dst[c] = src[R * (ns >> 2) + (c - R * q4)]; }
	}
	else for (int R = 0; R < 8; R++) {
		const int ki = k0 + (R >> 1);
		if (ki < a.K) for (int c = lane; c < n; c += 64) a.io[((size_t)ki * 2 + (R & 1)) * n + c] = rvq_tile[R * ns + c];
	}
	// ---- write back what changed ----
	if (k < a.K && !(KLG_RVQ_ABLATE & 8)) {
		float* Wr = a.state + k;
		Wr[(size_t)(fw + FD_Z0) * KP] = ff.z0; Wr[(size_t)(fw + FD_Z1) * KP] = ff.z1; Wr[(size_t)(fw + FD_IN) * KP] = fin;
		Wr[(size_t)(fw + FD_LASTP) * KP] = __int_as_float((int)(((long long)flast + 2ll * n) % RV_FSIZE));
		if (efilter) {
			Wr[(size_t)(RV_EZ + 2 * ech) * KP] = elz0; Wr[(size_t)(RV_EZ + 2 * ech + 1) * KP] = elz1;
			Wr[(size_t)(RV_EZ + 4 + 2 * ech) * KP] = ehz0; Wr[(size_t)(RV_EZ + 4 + 2 * ech + 1) * KP] = ehz1;
		}
	}
#undef RVW
}

// =================================================================================================
// Graph effects (include/klang_mi355_graph.h, `kind effect`): the recorded body of a user Effect::process()
// =================================================================================================
// One lane per effect instance, one wave per 64 instances (the shape of klg_fx_pingpong): record words in registers for
// the block, the caller's [K][CH][n] block staged through the padded LDS tile, Delay<SIZE> members as position-major rings
// in this group's ring tile.  P is generated by klg_graph.hpp: Rec / kStoreMask(2) / Live / begin / sample / end, kChannels.
struct FxGraphArgs {
	uint32_t* state; size_t kpad; int K;
	float* rings; size_t ring_rows;          // rows (of 64 floats) per group of 64 instances: the sum of the Delay SIZEs
	float* io; int n;
	const float* controls;                   // [kpad][KLG_MAX_CTL]
	SampleRate fs;
	unsigned long long samples;              // samples processed before this block (every delay's write cursor derives from it)
	const int* rand; size_t rstride;         // Noise: the span's rand() values [n * draws per sample][rstride], column = block * K + instance (or null): klg_rand_fill
	int blocks; size_t block_stride;         // klg_fx_staged only (klg_fx_render_device): a span of `blocks` blocks of n samples in one launch, block b's [K][CH][n] rows
	                                         // block_stride floats after block b - 1's; prepare() at the head of every block as in Effect::process(buffer).  0 / 1: one block
};
struct FxCtx { SampleRate fs; const float* ctl; unsigned long long samples; float* ring; const int* rand; size_t rstride; const uint32_t* rec; size_t stride; };   // rec / stride: the instance's record in HBM (word w at rec[w * stride]): what an Envelope with more than four point slots reads when a segment ends   // ring: this lane's column of the group's tile; rand: this instance's column of the block's draws
__device__ __forceinline__ float ctl_read(const FxCtx& c, unsigned i) { return c.ctl[i]; }

template<class P>
__global__ __launch_bounds__(FX_WG) void klg_fx_graph(const FxGraphArgs a) {
	using Rec = typename P::Rec;
	constexpr int W = sizeof(Rec) / 4, CH = P::kChannels;
	__shared__ float tile[2 * FX_CHUNK * FX_LD];
	const int lane = threadIdx.x, k0 = blockIdx.x * FX_WG, k = k0 + lane;
	RecWords<Rec> rw;
#pragma unroll
	for (int w = 0; w < W; w++) rw.w[w] = a.state[(size_t)w * a.kpad + k];
	Rec rec; rw.to(rec);
	typename P::Live L;
	FxCtx c;
	c.fs = a.fs; c.ctl = a.controls + (size_t)k * KLG_MAX_CTL; c.samples = a.samples;
	c.ring = a.rings + (size_t)(k / P::kRingRow) * a.ring_rows * P::kRingRow + (k % P::kRingRow);     // (rows of 64 instances: blockIdx.x * ring_rows * 64 + lane)
	c.rand = a.rand ? a.rand + (size_t)(k < a.K ? k : 0) : nullptr; c.rstride = a.rstride;
	c.rec = a.state + k; c.stride = a.kpad;
	P::begin(L, rec, c);
	const int col = lane & 31, half = lane >> 5;
	for (int s0 = 0; s0 < a.n; s0 += FX_CHUNK) {
		const int cl = (a.n - s0 < FX_CHUNK) ? (a.n - s0) : FX_CHUNK;
		for (int it = 0; it < 32 * CH; it++) {                       // 64 * CH rows of the [K][CH][n] block, two rows per access
			const int row = 2 * it + half, inst = row / CH, ch = row % CH;
			tile[(ch * FX_CHUNK + col) * FX_LD + inst] = (col < cl && k0 + inst < a.K) ? a.io[((size_t)(k0 + inst) * CH + ch) * a.n + s0 + col] : 0.f;
		}
		wave_sync();
		for (int s = 0; s < cl; s++) {
			const float in0 = tile[(0 * FX_CHUNK + s) * FX_LD + lane], in1 = CH > 1 ? tile[(1 * FX_CHUNK + s) * FX_LD + lane] : 0.f;
			float out0 = 0.f, out1 = 0.f;
			P::sample(L, c, in0, in1, out0, out1);
			tile[(0 * FX_CHUNK + s) * FX_LD + lane] = out0;
			if (CH > 1) tile[(1 * FX_CHUNK + s) * FX_LD + lane] = out1;
		}
		wave_sync();
		for (int it = 0; it < 32 * CH; it++) {
			const int row = 2 * it + half, inst = row / CH, ch = row % CH;
			if (col < cl && k0 + inst < a.K) a.io[((size_t)(k0 + inst) * CH + ch) * a.n + s0 + col] = tile[(ch * FX_CHUNK + col) * FX_LD + inst];
		}
		wave_sync();
	}
	if (k < a.K) {
		P::end(L, rec);
		rw.from(rec);
#pragma unroll
		for (int w = 0; w < W; w++) if (patch_stores<P>(w)) a.state[(size_t)w * a.kpad + k] = rw.w[w];
	}
}

// scatter host-side updates into the SoA state: upd = { k, word, value_bits } triples
__global__ void klg_fx_apply_updates(float* state, size_t kpad, const int* upd, int count) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < count) state[(size_t)upd[3 * i + 1] * kpad + upd[3 * i + 0]] = __int_as_float(upd[3 * i + 2]);
}

} // namespace klg
