// tools/calib/write_calib.hip — what does rocprofv3's WRITE_SIZE report for a known number of written bytes?  (MI355X_MICROARCH.md: "WRITE_SIZE is uncalibrated".)
// Three kernels that each write exactly 128 MiB, every byte once:
//   calib_wide     a wave writes 1 KB contiguous per store instruction (16 B per lane)
//   calib_piece64  the Reverb kernel's FilteredDelay flush: a quad of lanes writes one 64-byte piece of ITS line per store, 16 lines per wave, a line advances
//                  by 64 bytes per iteration (32 iterations = 2 KB per line and launch), lines 768 KB apart (the ring stride)
//   calib_piece32  a lane pair writes 32 bytes of its line per store (the early line's flush), same walk
// Build: hipcc --offload-arch=gfx950 -O3 tools/calib/write_calib.hip -o tools/calib/write_calib ; run under tools/pmc_any.py.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ void calib_wide(float* out, int iters) {
	v4* p = reinterpret_cast<v4*>(out) + (size_t)blockIdx.x * iters * 64 + threadIdx.x;
	for (int i = 0; i < iters; i++) { const v4 x = { (float)i, 1.f, 2.f, 3.f }; p[(size_t)i * 64] = x; }
}
__global__ void calib_piece64(float* out, size_t line_stride, int iters, int first) {     // 64 lanes = 16 lines x 4 quarters
	const int lane = threadIdx.x, line = blockIdx.x * 16 + (lane >> 2);
	float* p = out + (size_t)line * line_stride + first + 4 * (lane & 3);
	for (int i = 0; i < iters; i++) { const v4 x = { (float)i, 1.f, 2.f, 3.f }; *reinterpret_cast<v4*>(p + 16 * i) = x; asm volatile("s_sleep 20"); }
}
__global__ void calib_piece32(float* out, size_t line_stride, int iters, int first) {     // 64 lanes = 32 lines x 2 halves
	const int lane = threadIdx.x, line = blockIdx.x * 32 + (lane >> 1);
	float* p = out + (size_t)line * line_stride + first + 4 * (lane & 1);
	for (int i = 0; i < iters; i++) { const v4 x = { (float)i, 1.f, 2.f, 3.f }; *reinterpret_cast<v4*>(p + 8 * i) = x; asm volatile("s_sleep 20"); }
}
int main() {
	const size_t total = 128u << 20, stride = 192032;                      // floats per line (RV_FSTRIDE)
	const int lines64 = (int)(total / 2048), lines32 = (int)(total / 1024);
	float* wide; float* rings;
	const size_t estride = 21636;                                           // RV_ESTRIDE
	if (hipMalloc(&wide, total) != hipSuccess || hipMalloc(&rings, (size_t)lines64 * stride * 4) != hipSuccess) { std::printf("no memory\n"); return 1; }
	for (int rep = 0; rep < 12; rep++) {
		calib_wide<<<(int)(total / (64 * 16 * 32)), 64>>>(wide, 32);
		calib_piece64<<<lines64 / 16, 64>>>(rings, stride, 32, rep * 512);
		calib_piece32<<<lines32 / 32, 64>>>(rings, estride, 32, rep * 256);
	}
	if (hipDeviceSynchronize() != hipSuccess) return 1;
	std::printf("bytes per launch: %zu\n", total);
	return 0;
}
