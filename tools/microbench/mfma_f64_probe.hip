// Finds the operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950 by brute force over a few hypotheses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void raw(const double* a_lane, const double* b_lane, double* out) {
    const int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_lane[l], b_lane[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}
int main() {
    std::vector<double> A(64), B(64), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = std::sin(i * 1.37) + 2; B[i] = std::cos(i * 0.73) + 3; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *da, *db, *dout;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dout, 2048);
    const char* an[2] = {"A lane = i + 16k", "A lane = 4i + k"};
    const char* bn[2] = {"B lane = j + 16k", "B lane = 4j + k"};
    for (int ha = 0; ha < 2; ++ha) for (int hb = 0; hb < 2; ++hb) {
        std::vector<double> al(64), bl(64), out(256);
        for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) {
            al[ha == 0 ? i + 16 * k : 4 * i + k] = A[i * 4 + k];
            bl[hb == 0 ? i + 16 * k : 4 * i + k] = B[k * 16 + i];
        }
        hipMemcpy(da, al.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, bl.data(), 512, hipMemcpyHostToDevice);
        raw<<<1, 64>>>(da, db, dout);
        hipMemcpy(out.data(), dout, 2048, hipMemcpyDeviceToHost);
        // match every (lane, reg) to an (i, j)
        int matched = 0; std::vector<int> mi(256, -1), mj(256, -1);
        for (int x = 0; x < 256; ++x) for (int y = 0; y < 256; ++y) if (std::fabs(out[x] - R[y]) < 1e-12) { mi[x] = y / 16; mj[x] = y % 16; ++matched; break; }
        printf("%s, %s: %d / 256 results match some element\n", an[ha], bn[hb], matched);
        if (matched == 256) {
            printf("  lane: (i,j) of registers 0..3\n");
            for (int l = 0; l < 64; l += 1) if (l < 4 || l % 16 == 0 || l == 17) {
                printf("  lane %2d:", l);
                for (int r = 0; r < 4; ++r) printf(" (%2d,%2d)", mi[l * 4 + r], mj[l * 4 + r]);
                printf("\n");
            }
        }
    }
    return 0;
}
