// tools/calib/piece_bw.hip — what the memory system delivers for klg_fx_reverb_q's access pattern as a function of the PIECE a line is touched with:
// 65,536 lines 768 KB apart (16 FilteredDelay lines x 4,096 instances), every line read PIECE bytes at a time at one place and written PIECE bytes at a time
// at another, 2 KB per line and launch (a 256-sample block), waves x lanes = lines.  Reads as the kernel does them (a lane reads ITS line: 16-byte loads at
// consecutive addresses), writes as it does them (the four lanes of a quad write 64 contiguous bytes of one line per store).  No arithmetic: enough waves
// per SIMD that latency is hidden (SPLIT workgroups share a group of 64 lines, each taking a part of the visits: SPLIT waves per SIMD) — the number is what the
// access pattern costs the memory system, not what a kernel's dependent chain costs.  With SPLIT = 1 (one wave per SIMD, a visit waits for its loads) it is the
// round trip instead: 32 visits of 64 bytes in 66 us = 2.1 us each under that load.
// Build: hipcc --offload-arch=gfx950 -O3 tools/calib/piece_bw.hip -o tools/calib/piece_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
template<int PIECE>                                                          // bytes per line and visit: 64, 128, 256
__global__ __launch_bounds__(64) void piece_rw(float* rings, size_t stride, int visits, int rd0, int wr0) {
	const int lane = threadIdx.x, quad = lane >> 2, qi = lane & 3;
	const size_t wave = blockIdx.x;
	rd0 += blockIdx.y * visits * (PIECE / 4); wr0 += blockIdx.y * visits * (PIECE / 4);   // this workgroup's part of the line's 2 KB
	const float* rline = rings + (wave * 64 + lane) * stride + rd0;            // this lane's own line
	float* wbase = rings + (wave * 64 + quad * 4) * stride + wr0 + 4 * qi;     // the quad's four lines, this lane's 16-byte column
	v4 acc = { 0.f, 0.f, 0.f, 0.f };
	constexpr int Q = PIECE / 16;                                              // 16-byte loads per lane and visit
	for (int it = 0; it < visits; it++) {
		v4 x[Q];
#pragma unroll
		for (int q = 0; q < Q; q++) x[q] = *reinterpret_cast<const v4*>(rline + (size_t)it * (PIECE / 4) + 4 * q);
#pragma unroll
		for (int q = 0; q < Q; q++) acc += x[q];
#pragma unroll
		for (int v = 0; v < 4; v++)
#pragma unroll
			for (int h = 0; h < PIECE / 64; h++) __builtin_nontemporal_store(acc, reinterpret_cast<v4*>(wbase + (size_t)v * stride + (size_t)it * (PIECE / 4) + 16 * h));
	}
}
int main() {
	const size_t stride = 192032; const int lines = 65536;
	float* rings;
	if (hipMalloc(&rings, (size_t)lines * stride * 4) != hipSuccess) { std::printf("no memory\n"); return 1; }
	hipMemset(rings, 0, (size_t)lines * stride * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	auto run = [&](auto kernel, int piece, const char* name, int split) {
		const int visits = 2048 / piece / split;
		for (int rep = 0; rep < 3; rep++) kernel<<<dim3(lines / 64, split), 64>>>(rings, stride, visits, 96000 + rep * 512, rep * 512);
		hipEventRecord(e0);
		for (int rep = 3; rep < 23; rep++) kernel<<<dim3(lines / 64, split), 64>>>(rings, stride, visits, 96000 + rep * 512, rep * 512);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		const double bytes = 2.0 * lines * 2048.0;
		std::printf("{\"piece_bytes\": %d, \"waves_per_simd\": %d, \"us_per_launch\": %.1f, \"TB_per_s_read_plus_write\": %.2f}\n", piece, split, 1e3 * ms / 20, bytes / (1e-3 * ms / 20) / 1e12);
	};
	for (int split : { 1, 4, 8 }) { run(piece_rw<64>, 64, "piece_rw<64>", split); run(piece_rw<128>, 128, "piece_rw<128>", split); if (split <= 4) run(piece_rw<256>, 256, "piece_rw<256>", split); }
	return 0;
}
