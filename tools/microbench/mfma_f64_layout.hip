// Layout probe for v_mfma_f64_16x16x4_f64 and v_mfma_f32_16x16x4_f32 on gfx950 (build: hipcc --offload-arch=gfx950 -O2).
// Assumed (and checked here against a host product):  A[i][k]: lane = i + 16 k;  B[k][j]: lane = j + 16 k;
// D[i][j]: lane = j + 16 (i / 4), register i % 4.  Also times a dependent chain of each to get cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void probe(const double* A, const double* B, double* D, float* Df) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
    f4 accf = {0, 0, 0, 0};
    accf = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a, (float)b, accf, 0, 0, 0);
    for (int r = 0; r < 4; ++r) Df[(4 * (l >> 4) + r) * 16 + (l & 15)] = accf[r];
}

template <int WHICH>
__global__ void chain(double* out, int n, long long* cyc) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    d4 acc = {0, 0, 0, 0};
    f4 accf = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (WHICH == 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        else accf = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a, (float)b, accf, 0, 0, 0);
    }
    const long long t1 = clock64();
    out[threadIdx.x] = acc[0] + accf[0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    std::vector<float> Df(256);
    for (int i = 0; i < 64; ++i) { A[i] = std::sin(i * 1.37) + 1e-9 * i; B[i] = std::cos(i * 0.73) - 1e-9 * i; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD; float* dDf; long long* dc;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048); hipMalloc(&dDf, 1024); hipMalloc(&dc, 8 * 1024);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD, dDf);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost); hipMemcpy(Df.data(), dDf, 1024, hipMemcpyDeviceToHost);
    double e = 0, ef = 0;
    for (int i = 0; i < 256; ++i) { e = std::fmax(e, std::fabs(D[i] - R[i])); ef = std::fmax(ef, std::fabs(Df[i] - R[i])); }
    printf("f64 16x16x4 layout: max err %.3e (%s)   f32 16x16x4 layout: max err %.3e (%s)\n", e, e < 1e-14 ? "OK" : "MISMATCH", ef,
           ef < 1e-5 ? "OK" : "MISMATCH");
    const int n = 4096;
    for (int which = 0; which < 2; ++which)
        for (int waves = 1; waves <= 2; ++waves) {     // waves per SIMD: block of 256*waves threads on one CU
            long long c = 0;
            if (which == 0) chain<0><<<1, 256 * waves>>>(dD, n, dc); else chain<1><<<1, 256 * waves>>>(dD, n, dc);
            hipDeviceSynchronize();
            hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
            printf("%s dependent chain, %d wave(s) per SIMD: %.1f cycles per MFMA per wave\n", which ? "f32 16x16x4" : "f64 16x16x4", waves,
                   (double)c / n);
        }
    return e < 1e-14 ? 0 : 1;
}
