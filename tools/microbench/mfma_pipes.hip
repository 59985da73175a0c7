// Microbenchmark: which MFMA flavours share an execution pipe with what?
//  - fp32 MFMA (v_mfma_f32_32x32x2_f32), bf16 MFMA (v_mfma_f32_32x32x16_bf16), fp64 MFMA (v_mfma_f64_16x16x4_f64)
//  - each alone, each mixed with plain VALU fmas in the same wave, and pairs of flavours run by different
//    waves of the same SIMD (even waves flavour A, odd waves flavour B).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0 fp32 mfma, 1 bf16 mfma, 2 fp64 mfma, 3 VALU only
__device__ __forceinline__ float body(int iters, float seed) {
    float s = 0;
    if constexpr (KIND == 0) {
        f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = seed + r;
        float a = seed, b = seed * 0.5f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        for (int r = 0; r < 16; ++r) s += acc[r];
    } else if constexpr (KIND == 1) {
        f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = seed + r;
        bf16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed * 0.5f + i); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        for (int r = 0; r < 16; ++r) s += acc[r];
    } else if constexpr (KIND == 2) {
        f64x4 acc; for (int r = 0; r < 4; ++r) acc[r] = seed + r;
        double a = seed, b = seed * 0.5;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        for (int r = 0; r < 4; ++r) s += (float)acc[r];
    } else {
        float v[8]; for (int i = 0; i < 8; ++i) v[i] = seed + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    return s;
}

template <int KA, int KB>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    float s;
    if ((wave >> 2) & 1) s = body<KB>(iters, 1.0f + threadIdx.x);      // waves 4-7: second wave of every SIMD
    else s = body<KA>(iters, 2.0f + threadIdx.x);                       // waves 0-3: first wave of every SIMD
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KA, int KB>
void run(const char* name, float* d) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KA, KB><<<256, 512>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KA, KB><<<256, 512>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: one wave of kind A and one of kind B, each issuing iters*16 instructions (or 16*8 fmas for VALU)
    printf("%-46s %8.3f ms   cycles@2.4GHz per (A-instr + B-instr) pair per SIMD: %.1f\n", name, ms, ms * 1e-3 * 2.4e9 / (iters * 16.0));
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * sizeof(float));
    run<0, 0>("fp32 mfma | fp32 mfma", d);
    run<1, 1>("bf16 mfma | bf16 mfma", d);
    run<2, 2>("fp64 mfma | fp64 mfma", d);
    run<3, 3>("8 VALU fma | 8 VALU fma", d);
    run<0, 1>("fp32 mfma | bf16 mfma", d);
    run<0, 2>("fp32 mfma | fp64 mfma", d);
    run<1, 2>("bf16 mfma | fp64 mfma", d);
    run<0, 3>("fp32 mfma | 8 VALU fma", d);
    run<1, 3>("bf16 mfma | 8 VALU fma", d);
    run<2, 3>("fp64 mfma | 8 VALU fma", d);
    return 0;
}
