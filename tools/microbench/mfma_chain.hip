// Microbenchmark: how fast does a SIMD issue v_mfma_f32_32x32x2_f32 when every MFMA depends on the
// previous one (one accumulator chain per wave) vs. two independent chains, at 1 / 2 / 3 waves per SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS, int VALU>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = threadIdx.x * 0.001f + c;
    float a = a0 + threadIdx.x, b = a0 * 0.5f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
            if (VALU) {
#pragma unroll
                for (int i = 0; i < VALU; ++i) v[i & 7] = fmaf(v[i & 7], 1.0001f, 0.5f);
            }
        }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS, int VALU>
void run(const char* name, int wg_per_cu, float* d) {
    const int iters = 2000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS, VALU><<<grid, 256>>>(d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CHAINS, VALU><<<grid, 256>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * iters * 16 * CHAINS;      // wave-level MFMA instructions
    const double tf = mfma * 2.0 * 32 * 32 * 2 / (ms * 1e-3) / 1e12;
    const double cyc_per_mfma_per_simd = (ms * 1e-3 * 2.4e9) / (mfma / (256.0 * 4));
    printf("%-34s wg/CU %d (waves/SIMD %d): %8.3f ms  %7.1f TF/s  %.1f cycles@2.4GHz per MFMA per SIMD\n", name, wg_per_cu, wg_per_cu,
           ms, tf, cyc_per_mfma_per_simd);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w = 1; w <= 3; ++w) {
        run<1, 0>("1 dependent chain, no VALU", w, d);
        run<2, 0>("2 independent chains, no VALU", w, d);
        run<1, 2>("1 chain + 2 VALU per MFMA", w, d);
        run<1, 6>("1 chain + 6 VALU per MFMA", w, d);
        run<2, 6>("2 chains + 6 VALU per MFMA pair", w, d);
    }
    return 0;
}
