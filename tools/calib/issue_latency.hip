// tools/calib/issue_latency.hip — what ONE wave alone on its SIMD pays per instruction on gfx950 (MI355X): cycles (s_memtime) per operation of a chain of
// dependent v_add_f32 / v_mul_f32 / v_pk_add_f32, of independent v_add_f32, and of independent v_add_f32 with a scalar operation in between.
// The numbers behind DESIGN.md's "a wave alone on its SIMD" statements (klg_render_sub2a_sp, the generated serial loops, PingPong's filter waves).
//   hipcc --offload-arch=gfx950 -O3 tools/calib/issue_latency.hip -o /tmp/issue_latency && /tmp/issue_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
__global__ void probe(float* out, long long* cyc, float a, float b) {
	float x = a, y0 = a, y1 = b, y2 = a + 1.f, y3 = b + 1.f;
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 p = { a, b }; const f2 q = { b, a };
	long long t0, t1;
	// 1. dependent v_add_f32
	t0 = clock64();
#pragma unroll
	for (int i = 0; i < REP; i++) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x) : "v"(b));
	t1 = clock64(); cyc[0] = t1 - t0;
	// 2. dependent v_mul_f32
	t0 = clock64();
#pragma unroll
	for (int i = 0; i < REP; i++) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x) : "v"(b));
	t1 = clock64(); cyc[1] = t1 - t0;
	// 3. dependent v_pk_add_f32
	t0 = clock64();
#pragma unroll
	for (int i = 0; i < REP; i++) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p) : "v"(q));
	t1 = clock64(); cyc[2] = t1 - t0;
	// 4. four independent chains of v_add_f32 (REP instructions in all)
	t0 = clock64();
#pragma unroll
	for (int i = 0; i < REP / 4; i++) { asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(y0) : "v"(b)); asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(y1) : "v"(b));
		asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(y2) : "v"(b)); asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(y3) : "v"(b)); }
	t1 = clock64(); cyc[3] = t1 - t0;
	// 5. the same with a scalar operation behind every vector one (2 x REP instructions)
	t0 = clock64();
#pragma unroll
	for (int i = 0; i < REP / 4; i++) { asm volatile("v_add_f32_e32 %0, %1, %0\n\ts_nop 0" : "+v"(y0) : "v"(b)); asm volatile("v_add_f32_e32 %0, %1, %0\n\ts_nop 0" : "+v"(y1) : "v"(b));
		asm volatile("v_add_f32_e32 %0, %1, %0\n\ts_nop 0" : "+v"(y2) : "v"(b)); asm volatile("v_add_f32_e32 %0, %1, %0\n\ts_nop 0" : "+v"(y3) : "v"(b)); }
	t1 = clock64(); cyc[4] = t1 - t0;
	// 6. dependent v_add_f32 with a DPP move (wave_shl:1) of an unrelated register behind each
	float h = a;
	t0 = clock64();
#pragma unroll
	for (int i = 0; i < REP; i++) { asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x) : "v"(b)); asm volatile("v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(h)); }
	t1 = clock64(); cyc[5] = t1 - t0;
	// 7. dependent v_pk_mul_f32, 8. dependent v_pk_fma_f32, 9. a packed result consumed by a PLAIN operation on one of its halves and fed back (what a filter packed by hand does)
	t0 = clock64();
#pragma unroll
	for (int i = 0; i < REP; i++) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p) : "v"(q));
	t1 = clock64(); cyc[6] = t1 - t0;
	t0 = clock64();
#pragma unroll
	for (int i = 0; i < REP; i++) asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(p) : "v"(q));
	t1 = clock64(); cyc[7] = t1 - t0;
	t0 = clock64();
#pragma unroll
	for (int i = 0; i < REP / 2; i++) { asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p) : "v"(q)); asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(p.x) : "v"(b)); }
	t1 = clock64(); cyc[8] = t1 - t0;
	out[threadIdx.x] = x + y0 + y1 + y2 + y3 + p.x + p.y + h;
}
int main() {
	float* out; long long* cyc; hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 16 * 8);
	for (int r = 0; r < 3; r++) probe<<<1, 64>>>(out, cyc, 1.0f, 1e-3f);
	long long h[9]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
	const char* names[9] = { "dependent v_add_f32", "dependent v_mul_f32", "dependent v_pk_add_f32", "independent v_add_f32 (4 chains)", "independent v_add_f32 + s_nop each", "dependent v_add_f32 + a DPP move each", "dependent v_pk_mul_f32", "dependent v_pk_fma_f32", "v_pk_mul_f32 <-> v_add_f32 on one half, alternating" };
	for (int i = 0; i < 9; i++) printf("%-40s %6.2f cycles per vector operation (%lld for %d)\n", names[i], (double)h[i] / REP, h[i], REP);
	return 0;
}
