// stream_read.hip -- what read bandwidth does the K_e access pattern of mp_fused reach on its own?  Every wave streams a private
// contiguous region with 16-byte-per-lane loads (1 KiB per instruction), DEPTH instructions in flight, WAVES waves per CU,
// non-temporal or default policy.  hipcc --offload-arch=gfx950 -O3 stream_read.hip -o stream_read && ./stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ src, float* __restrict__ out, size_t per_wave_vec) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const f32x4* p = src + wave * per_wave_vec + lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 q[DEPTH];
    const size_t n = per_wave_vec / 64;                       // instructions of this wave
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) q[d] = NT ? __builtin_nontemporal_load(p + (size_t)d * 64) : p[(size_t)d * 64];
    for (size_t i = DEPTH; i < n; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            acc += q[d];
            q[d] = NT ? __builtin_nontemporal_load(p + (i + d) * 64) : p[(i + d) * 64];
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += q[d];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = acc[0];
}

template <int DEPTH, bool NT>
static void run(const f32x4* src, float* out, size_t bytes, int wgs) {
    const size_t per_wave_vec = bytes / 16 / ((size_t)wgs * 4) / 64 / DEPTH * 64 * DEPTH;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((stream_kernel<DEPTH, NT>), dim3(wgs), dim3(256), 0, 0, src, out, per_wave_vec);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double moved = (double)per_wave_vec * 16 * wgs * 4;
    printf("depth %2d  %-3s  %5d workgroups (%4.1f waves per SIMD)  %7.1f MB in %7.1f us  = %5.2f TB/s\n", DEPTH, NT ? "nt" : "def", wgs,
           wgs * 4 / 1024.0, moved / 1e6, ms * 1e3, moved / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = (size_t)1 << 30;                     // 1 GiB: four times the Infinity Cache
    f32x4* src; float* out;
    hipMalloc(&src, bytes); hipMalloc(&out, 64);
    hipMemset(src, 0, bytes);
    for (int wgs : {256, 512, 1024, 2048}) {
        run<2, true>(src, out, bytes, wgs);
        run<4, true>(src, out, bytes, wgs);
        run<8, true>(src, out, bytes, wgs);
        run<16, true>(src, out, bytes, wgs);
        run<8, false>(src, out, bytes, wgs);
    }
    // 256 MB footprint (the size of one K_e array at the bf16 shapes): does the Infinity Cache help a re-read?
    for (int wgs : {512, 1024}) {
        run<8, true>(src, out, (size_t)256 << 20, wgs);
        run<8, false>(src, out, (size_t)256 << 20, wgs);
    }
    return 0;
}
