// klang_amd/csrc/klg_delay.hpp — Delay<SIZE> (klang.h:3381-3512) on position-major rings in HBM: element i of a line sits at
// base[i * stride], `stride` apart from its neighbours in time and next to the same element of the other 63 lanes of the wave
// (one 256-byte row per ring position per wave: a wave's read or write of "position i" is one coalesced access when the lanes
// agree on i, and 64 sectors when they do not).  Used by the effect kernels (klg_fx.hpp) and by note delays of generated
// synth patches (physical models: a delay line per voice).
#pragma once
#include "klg_device.hpp"

#pragma clang fp contract(off)

namespace klg {

// ---- Delay<SIZE> on an interleaved ring (klang.h:3381-3512) ----
// THE PAD ELEMENT.  klang's Delay<SIZE> owns SIZE + 1 floats (`buffer(SIZE + 1, 0)`, klang.h:3391) and writes SIZE of them: element SIZE stays
// 0 for ever.  It IS read, rarely: `read = (position - 1) - time; if (read < 0) read += SIZE;` is float arithmetic, and a `read` within half an
// ulp below zero rounds to exactly SIZE (at SIZE = 192000 that is one cursor position in ~130 per crossing of the tap over the ring's start) —
// the tap then takes buffer[SIZE] = 0 and buffer[(SIZE + 1) % SIZE] = buffer[1], and a walking read head continues at 1.  Every line here
// therefore has SIZE + 1 elements as well (the pad is never written: zero-filled at creation), a Tap::position may be SIZE, and the successor
// of index i is (i + 1) mod SIZE for 0 <= i <= SIZE — ring_succ below (two operations: the unsigned minimum of j and j - SIZE).
// (Found by tests/test_gpu_fx.py::test_pingpong_with_stationary_controls: one of twenty instances hit that cursor position.)
__device__ __forceinline__ int ring_succ(int i, int size) { const unsigned j = (unsigned)i + 1u; const unsigned w = j - (unsigned)size; return (int)(j < w ? j : w); }
struct Ring {
	float* base;          // this wave's column: &rings[line][0][k]
	size_t stride;        // Kpad
	int size;
	__device__ __forceinline__ float rd(int i) const { return base[(size_t)i * stride]; }
	__device__ __forceinline__ void wr(int i, float v) const { base[(size_t)i * stride] = v; }
};
struct Tap { int position; float fraction; };
__device__ __forceinline__ Tap delay_set(int position, int size, float samples) {      // Delay::set 3480-3489
	const float time = samples < size ? samples : (float)size;
	float read = (float)(position - 1) - time;
	if (read < 0.f) read += size;
	Tap t; t.position = (int)read; t.fraction = read - t.position;
	return t;
}
__device__ __forceinline__ float delay_process(const Ring& r, Tap& t) {               // tap() 3461-3468 + process 3470-3473
	const int i = t.position;
	const int j = ring_succ(i, r.size);                                               // (i + 1) % SIZE for 0 <= i <= SIZE (i == SIZE: the pad element)
	const float a = r.rd(i), b = r.rd(j);
	const float out = a + t.fraction * (b - a);
	t.position = j;
	return out;
}

// Delay::tap(int) 3405-3410, tap(float) 3412-3427, lagrange(float) 3429-3458 — `position` is the write cursor
__device__ __forceinline__ float delay_tap_int(const Ring& r, int position, int delay) {
	int read = (position - 1) - delay;
	if (read < 0) read += r.size;
	read = read < 0 ? r.size : (read > r.size ? r.size : read);                       // a tap outside the line (delay < 0 or > SIZE: undefined in the reference) reads the pad, not a neighbour's line
	return r.rd(read);
}
__device__ __forceinline__ float delay_tap_float(const Ring& r, int position, float delay) {
	float read = (float)(position - 1) - delay;
	if (read < 0.f) read += r.size;
	const int i = (int)read < 0 ? 0 : (int)read;                               // (a delay beyond SIZE leaves `read` negative: undefined in the reference, which indexes its buffer with it; here the tap stays inside this line)
	const float fraction = read - i;
	const int j = (i + 1) % r.size;
	const float a = r.rd(i), b = r.rd(j);
	return a + fraction * (b - a);
}
// HOISTED TAPS (generated effects, klg_graph.hpp).  In a recorded process() a tap's rows are loaded where the reference reads them — after the line's
// input() of the sample, which waits for the OTHER line's tap: two or three dependent round trips to memory per sample (PingPong.k), eleven (Reverb.k).
// The rows are known long before: the code generator requests them where their address is first computable (`h`: rows and values), and the tap at its own
// place takes them if they are the rows it needs and no input() of the line has written one of them since (`hazard`: compared row by row) — else it reads
// memory as before.  Same arithmetic on the same values.
struct Rows2 { int i, j; };
__device__ __forceinline__ float delay_process_h(const Ring& r, Tap& t, Rows2 h, float ha, float hb, bool hazard) {
	const int i = t.position, j = ring_succ(i, r.size);
	float a = ha, b = hb;
	if (hazard || i != h.i || j != h.j) { a = r.rd(i); b = r.rd(j); }
	const float out = a + t.fraction * (b - a);
	t.position = j;
	return out;
}
__device__ __forceinline__ Rows2 delay_tap_float_rows(int size, int position, float delay) {
	float read = (float)(position - 1) - delay;
	if (read < 0.f) read += size;
	Rows2 h; h.i = (int)read < 0 ? 0 : (int)read; h.j = (h.i + 1) % size;
	return h;
}
__device__ __forceinline__ float delay_tap_float_h(const Ring& r, int position, float delay, Rows2 h, float ha, float hb, bool hazard) {
	float read = (float)(position - 1) - delay;
	if (read < 0.f) read += r.size;
	const int i = (int)read < 0 ? 0 : (int)read;
	const float fraction = read - i;
	const int j = (i + 1) % r.size;
	float a = ha, b = hb;
	if (hazard || i != h.i || j != h.j) { a = r.rd(i); b = r.rd(j); }
	return a + fraction * (b - a);
}
__device__ __forceinline__ Rows2 delay_tap_stereo_rows(int size, int position, float delay) {
	float read = (float)(position - 1) - delay;
	if (read < 0.f) read += size;
	const int i = (int)read, j = (i == size - 1) ? 0 : (i + 1);
	const bool pad = i >= size || i < 0;
	Rows2 h; h.i = pad ? 0 : i; h.j = pad ? 0 : j;
	return h;
}
__device__ __forceinline__ float delay_tap_stereo_h(const Ring& r, int position, float delay, Rows2 h, float ha, float hb, bool hazard) {
	float read = (float)(position - 1) - delay;
	if (read < 0.f) read += r.size;
	const float f = (float)floor((double)read);
	const float frac = read - f;
	const int i = (int)read;
	const int j = (i == r.size - 1) ? 0 : (i + 1);
	const bool pad = i >= r.size || i < 0;
	const int ri = pad ? 0 : i, rj = pad ? 0 : j;
	float va = ha, vb = hb;
	if (hazard || ri != h.i || rj != h.j) { va = r.rd(ri); vb = r.rd(rj); }
	const float a = pad ? 0.f : va, b = pad ? 0.f : vb;
	return a * (1.f - frac) + b * frac;
}
// one channel of Stereo::Delay::tap(float) klang.h:4668-4681 — not the mono form: a * (1 - frac) + b * frac, frac against floor(read), the successor wraps at
// SIZE - 1 (both lines of a Stereo::Delay stand at the same cursor).  `read` rounded up to exactly SIZE: the reference reads the pad and one element
// past it (indeterminate there); that tap is 0 here (klg_fx.hpp stereo_delay_tap, oracle ko_stereo_delay_tap_float).
__device__ __forceinline__ float delay_tap_stereo(const Ring& r, int position, float delay) {
	float read = (float)(position - 1) - delay;
	if (read < 0.f) read += r.size;
	const float f = (float)floor((double)read);
	const float frac = read - f;
	const int i = (int)read;
	const int j = (i == r.size - 1) ? 0 : (i + 1);
	const bool pad = i >= r.size || i < 0;                                        // (i < 0: a delay beyond SIZE — undefined in the reference; the tap is 0 here and no other line is touched)
	const float a = pad ? 0.f : r.rd(pad ? 0 : i), b = pad ? 0.f : r.rd(pad ? 0 : j);
	return a * (1.f - frac) + b * frac;
}
// STAGED EFFECTS (klg_graph_staged.hpp): a chunk of C samples of an instance is computed with samples side by side, every ring read of the chunk before any of
// its ring writes.  That is the reference's result exactly when no read of the chunk touches a row the chunk itself writes — rows w0 .. w0 + wn - 1 (mod SIZE) of
// the line: a tap that close behind the cursor would have to see this chunk's input()s, one almost SIZE behind would have to see the OLD value of a row a later
// sample overwrites.  Each read reports whether it did (`bad`); the chunk is then walked again in sample order by the plain body, before anything was written.
// The arithmetic is that of delay_process / delay_tap_float / delay_tap_stereo / delay_tap_int above, on the same rows.
struct RingWindow { int w0, wn; };
__device__ __forceinline__ bool ring_in_window(int row, int size, RingWindow w) { int d = row - w.w0; d = d < 0 ? d + size : d; return row < size && d < w.wn; }
__device__ __forceinline__ int ring_walk(int p, int m, int size) {                  // the read head after m process() calls: ring_succ applied m times (m < SIZE)
	if (m == 0) return p;
	int q = (p == size ? 0 : p) + m;
	return q >= size ? q - size : q;
}
// Every read comes in two halves — `_fetch` finds the rows, checks them and ISSUES the loads, `_finish` is the arithmetic on what they return — so that a
// level of the staged kernel can issue all its taps' loads before it waits for the first of them (klg_graph_staged.hpp: one memory round trip per level
// instead of one per tap).  The whole functions are finish(fetch()): the same operations in the same order either way.
struct TapFetch { float a, b, c, d, f; bool pad; float la, lb; bool na, nb; };   // la / lb, na / nb: a row the chunk itself has written by then, from the chunk's own copy (staged_tap_float_fetch_near)
// a line of the staged kernel's ring tile: rows of `stride4` bytes from a wave-uniform base, this lane's column `col4` bytes into a row — 32-bit offsets
// (a workgroup's tile is < 4 GB): one operation per address where a per-lane 64-bit pointer takes three
struct RingS {
	const char* base; unsigned col4, stride4; int size;
	__device__ __forceinline__ float rd(int i) const { return *(const float*)(base + ((unsigned)i * stride4 + col4)); }
	__device__ __forceinline__ void wr(int i, float v) const { *(float*)(const_cast<char*>(base) + ((unsigned)i * stride4 + col4)) = v; }
};
// rows i and ring_succ(i) of a read, checked together: i in [w0 - 1, w0 + wn) (mod SIZE) is exactly "i or its successor lies in the window"; the pad row
// (i == SIZE: zeros, its successor row 0 or 1) counts as row 0 — a check that says "maybe" where it need not only sends the chunk to the plain body
__device__ __forceinline__ bool ring_pair_in_window(int i, int size, RingWindow w) {
	const int ii = i >= size ? 0 : i;
	int d = ii - w.w0 + 1;
	d = d < 0 ? d + size : (d >= size ? d - size : d);
	return d < w.wn + 1;
}
template<class RG> __device__ __forceinline__ TapFetch staged_process_fetch(const RG& r, int position, float fraction, RingWindow w, int& bad) {
	const int i = position, j = ring_succ(i, r.size);
	bad |= (int)ring_pair_in_window(i, r.size, w);
	TapFetch t; t.a = r.rd(i); t.b = r.rd(j); t.c = t.d = 0.f; t.f = fraction; t.pad = false;
	return t;
}
__device__ __forceinline__ float staged_process_finish(const TapFetch& t) { return t.a + t.f * (t.b - t.a); }
template<class RG> __device__ __forceinline__ float staged_process(const RG& r, int position, float fraction, RingWindow w, int& bad) { return staged_process_finish(staged_process_fetch(r, position, fraction, w, bad)); }
template<class RG> __device__ __forceinline__ TapFetch staged_tap_int_fetch(const RG& r, int position, int delay, RingWindow w, int& bad) {
	int read = (position - 1) - delay;
	if (read < 0) read += r.size;
	read = read < 0 ? r.size : (read > r.size ? r.size : read);
	bad |= (int)ring_in_window(read, r.size, w);
	TapFetch t; t.a = r.rd(read); t.b = t.c = t.d = t.f = 0.f; t.pad = false;
	return t;
}
__device__ __forceinline__ float staged_tap_int_finish(const TapFetch& t) { return t.a; }
template<class RG> __device__ __forceinline__ float staged_tap_int(const RG& r, int position, int delay, RingWindow w, int& bad) { return staged_tap_int_finish(staged_tap_int_fetch(r, position, delay, w, bad)); }
template<class RG> __device__ __forceinline__ TapFetch staged_tap_float_fetch(const RG& r, int position, float delay, RingWindow w, int& bad) {
	float read = (float)(position - 1) - delay;
	if (read < 0.f) read += r.size;
	const int i = (int)read < 0 ? 0 : (int)read;
	const float fraction = read - i;
	const int j = (i + 1 >= r.size) ? i + 1 - r.size : i + 1;                         // (i + 1) % SIZE for 0 <= i <= SIZE
	bad |= (int)ring_pair_in_window(i, r.size, w);                                   // (i == SIZE: the pad row and row 1 — checked as rows 0 and 1)
	TapFetch t; t.a = r.rd(i); t.b = r.rd(j); t.c = t.d = 0.f; t.f = fraction; t.pad = false;
	return t;
}
__device__ __forceinline__ float staged_tap_float_finish(const TapFetch& t) { return t.a + t.f * (t.b - t.a); }
// A line whose input() takes the effect's `in` itself (`in >> delay; ... delay(t)`: Flanger, Chorus, ModDelay) needs no check at all.  What a tap may read of
// this chunk is known before the chunk starts: row w0 + m holds sample m's input — the chunk's copy of `in`, `mine[m * ld]` —, visible to the tap of sample s
// when m < `vis` (= s, + 1 when the input() stands before the tap in the sample); every other row is read from the ring, which holds what it held before the
// chunk until the chunk's last level (also rows a LATER sample overwrites: the tap wants their old value).  One input() per sample.
template<class RG> __device__ __forceinline__ TapFetch staged_tap_float_fetch_near(const RG& r, int position, float delay, int w0, int vis, const float* mine, int ld) {
	float read = (float)(position - 1) - delay;
	if (read < 0.f) read += r.size;
	const int i = (int)read < 0 ? 0 : (int)read;
	const float fraction = read - i;
	const int j = (i + 1 >= r.size) ? i + 1 - r.size : i + 1;
	int di = i - w0; di = di < 0 ? di + r.size : di;
	int dj = j - w0; dj = dj < 0 ? dj + r.size : dj;
	TapFetch t; t.c = t.d = 0.f; t.f = fraction; t.pad = false;
	t.na = i < r.size && di < vis; t.nb = dj < vis;
	t.a = r.rd(i); t.b = r.rd(j);
	t.la = mine[(t.na ? di : 0) * ld]; t.lb = mine[(t.nb ? dj : 0) * ld];
	return t;
}
__device__ __forceinline__ float staged_tap_float_finish_near(const TapFetch& t) { const float a = t.na ? t.la : t.a, b = t.nb ? t.lb : t.b; return a + t.f * (b - a); }
template<class RG> __device__ __forceinline__ float staged_tap_float_near(const RG& r, int position, float delay, int w0, int vis, const float* mine, int ld) { return staged_tap_float_finish_near(staged_tap_float_fetch_near(r, position, delay, w0, vis, mine, ld)); }
template<class RG> __device__ __forceinline__ float staged_tap_float(const RG& r, int position, float delay, RingWindow w, int& bad) { return staged_tap_float_finish(staged_tap_float_fetch(r, position, delay, w, bad)); }
template<class RG> __device__ __forceinline__ TapFetch staged_tap_stereo_fetch(const RG& r, int position, float delay, RingWindow w, int& bad) {
	float read = (float)(position - 1) - delay;
	if (read < 0.f) read += r.size;
	const float f = __builtin_floorf(read);                                         // == (float)floor((double)read): the floor of a float is a float
	const float frac = read - f;
	const int i = (int)read;
	const int j = (i == r.size - 1) ? 0 : (i + 1);
	const bool pad = i >= r.size || i < 0;
	const int ri = pad ? 0 : i, rj = pad ? 0 : j;
	bad |= (int)(!pad && ring_pair_in_window(ri, r.size, w));
	TapFetch t; t.a = r.rd(ri); t.b = r.rd(rj); t.c = t.d = 0.f; t.f = frac; t.pad = pad;      // (a pad tap reads row 0 and drops it: no branch around the loads)
	return t;
}
__device__ __forceinline__ float staged_tap_stereo_finish(const TapFetch& t) { const float a = t.pad ? 0.f : t.a, b = t.pad ? 0.f : t.b; return a * (1.f - t.f) + b * t.f; }
template<class RG> __device__ __forceinline__ float staged_tap_stereo(const RG& r, int position, float delay, RingWindow w, int& bad) { return staged_tap_stereo_finish(staged_tap_stereo_fetch(r, position, delay, w, bad)); }
template<class RG> __device__ __forceinline__ TapFetch staged_lagrange_fetch(const RG& r, int position, float delay, RingWindow w, int& bad) {   // delay_lagrange below + the check of its four rows
	const int SIZE = r.size;
	float read = (float)(position - 1) - delay;
	if (read < 0.f) read += SIZE;
	int i = (int)read; i = i < 0 ? 0 : (i > SIZE ? SIZE : i);
	const float x = read - i;
	const int i0 = (i - 1 + SIZE) % SIZE, i2 = (i + 1) % SIZE, i3 = (i + 2) % SIZE;
	bad |= (int)(ring_in_window(i0, SIZE, w) || ring_in_window(i, SIZE, w) || ring_in_window(i2, SIZE, w) || ring_in_window(i3, SIZE, w));
	TapFetch t; t.a = r.rd(i0); t.b = r.rd(i); t.c = r.rd(i2); t.d = r.rd(i3); t.f = x; t.pad = false;
	return t;
}
__device__ __forceinline__ float staged_lagrange_finish(const TapFetch& t) {
	const float x = t.f, y0 = t.a, y1 = t.b, y2 = t.c, y3 = t.d;
	const float c0 = (-x * (x - 1) * (x - 2)) / 6.0f;
	const float c1 = ((x + 1) * (x - 1) * (x - 2)) / 2.0f;
	const float c2 = (-x * (x + 1) * (x - 2)) / 2.0f;
	const float c3 = (x * (x + 1) * (x - 1)) / 6.0f;
	return c0 * y0 + c1 * y1 + c2 * y2 + c3 * y3;
}
template<class RG> __device__ __forceinline__ float staged_lagrange(const RG& r, int position, float delay, RingWindow w, int& bad) { return staged_lagrange_finish(staged_lagrange_fetch(r, position, delay, w, bad)); }
__device__ __forceinline__ float delay_lagrange(const Ring& r, int position, float delay) {
	const int SIZE = r.size;
	float read = (float)(position - 1) - delay;
	if (read < 0.f) read += SIZE;
	int i = (int)read; i = i < 0 ? 0 : (i > SIZE ? SIZE : i);                      // (a delay beyond the line is undefined in the reference; here the tap stays inside this line; i == SIZE: the pad element, as in tap(float))
	const float x = read - i;
	const float y0 = r.rd((i - 1 + SIZE) % SIZE), y1 = r.rd(i), y2 = r.rd((i + 1) % SIZE), y3 = r.rd((i + 2) % SIZE);
	const float c0 = (-x * (x - 1) * (x - 2)) / 6.0f;
	const float c1 = ((x + 1) * (x - 1) * (x - 2)) / 2.0f;
	const float c2 = (-x * (x + 1) * (x - 2)) / 2.0f;
	const float c3 = (x * (x + 1) * (x - 1)) / 6.0f;
	return c0 * y0 + c1 * y1 + c2 * y2 + c3 * y3;
}

} // namespace klg
