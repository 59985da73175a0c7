// train_kernels.hip -- training-mode forward (activations kept) and backward of the explorer's TRAINABLE path
// (train_explorer.py:156-186).
//
// What trains under the reference's loss: model.py:141,142,146 detach node_free_code / edge_free_code before every
// use, so the policy loss reaches node_code, edge_code, goal_encoder, encoder, process.lin_0 / lin_1, decoder and
// policy, and NOTHING of the obstacle-attention stack (node/edge_free_code, obs_*_code, *_attentions).  The frozen
// inputs NF [N, d] and EF [E, d] therefore come from the inference kernels (pre_kernel / pre_resident_kernel store them
// on request); this file restates the rest in the reference's own formulation (materialised concatenations), one
// plain kernel per operator, in the padded CSR index space of explorer_kernels.hip:
//
//   rows_linear      Y = act(X W^T + b)                       W row-major [O, K] exactly as in the state_dict; the three GEMM
//                    shapes run on v_mfma_f32_32x32x2_f32 with the operands read at their natural strides
//   rows_linear_dx   dX (+)= dY W
//   rows_linear_dw   dW += dY^T X, db += sum_rows dY          (per-block partial sums, then an ordered second stage)
//   segment_max      A[n] = max over the node's CSR segment, with the arg-max slot per (node, feature); 0 if empty
//   gather / scatter kernels for the concatenations [v, g, (v-g)^2, v-g], [v_s, v_t], [NC, NF, H0, H],
//                    [X_s - X_t, X_s, X_t, EF, EC], [NC, H], [D_s, D_s - D_t, EF] and their adjoints
//
// The register-resident MFMA chains of the inference path are not reused: their packed, pre-combined weights (W_a + W_b,
// ...) are the wrong parameterisation for gradients, and the weights change every optimizer step.  Everything is
// deterministic: no float atomics, fixed summation orders (CSR segments sorted by caller column first).
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.hpp"

namespace gnnmp {

#define TRAIN_LAUNCH_CHECK()                 \
    do {                                     \
        hipError_t _e = hipGetLastError();   \
        if (_e != hipSuccess) return _e;     \
    } while (0)

// Y[r, o] = act(sum_k X[r, k] W[o, k] + b[o]);  rows with valid[r] == 0 (padding slots) produce 0
__global__ void rows_linear_kernel(int R, int K, int O, const float* __restrict__ X, const float* __restrict__ W,
                                   const float* __restrict__ b, float* __restrict__ Y, int relu) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * O) return;
    const int r = (int)(i / O), o = (int)(i % O);
    const float* x = X + (size_t)r * K;
    const float* w = W + (size_t)o * K;
    float acc = b ? b[o] : 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(x[k], w[k], acc);
    Y[i] = (relu && acc < 0.f) ? 0.f : acc;
}

// dX[r, k] (+)= sum_o dY[r, o] W[o, k]
__global__ void rows_linear_dx_kernel(int R, int K, int O, const float* __restrict__ dY, const float* __restrict__ W,
                                      float* __restrict__ dX, int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * K) return;
    const int r = (int)(i / K), k = (int)(i % K);
    const float* dy = dY + (size_t)r * O;
    float acc = 0.f;
    for (int o = 0; o < O; ++o) acc = fmaf(dy[o], W[(size_t)o * K + k], acc);
    dX[i] = accumulate ? dX[i] + acc : acc;
}

// dW[o, k] += sum_r dY[r, o] X[r, k];  db[o] += sum_r dY[r, o] in two DETERMINISTIC stages: one block per chunk of kDwRows
// rows writes its partial sums (fixed row order) to part[block][O K + O]; a second kernel adds the partials block by block
// in ascending order.  No float atomics: the same inputs give the same gradient bits on every run.
constexpr int kDwRows = 128;
__global__ __launch_bounds__(256) void rows_linear_dw_partial_kernel(int R, int K, int O, const float* __restrict__ dY,
                                                                     const float* __restrict__ X, float* __restrict__ part) {
    const int r0 = blockIdx.x * kDwRows, r1 = min(R, r0 + kDwRows);
    float* out = part + (size_t)blockIdx.x * (O * K + O);
    for (int i = threadIdx.x; i < O * K; i += 256) {
        const int o = i / K, k = i % K;
        float acc = 0.f;
        for (int r = r0; r < r1; ++r) acc = fmaf(dY[(size_t)r * O + o], X[(size_t)r * K + k], acc);
        out[i] = acc;
    }
    for (int o = threadIdx.x; o < O; o += 256) {
        float acc = 0.f;
        for (int r = r0; r < r1; ++r) acc += dY[(size_t)r * O + o];
        out[O * K + o] = acc;
    }
}
// second stage: element i of dW / db += the nblk partials in a FIXED order -- eight contiguous slices of the block range summed side by
// side (ascending inside a slice), the eight slice sums added in ascending order.  (One serial loop over all blocks per element was
// 26 us per layer at 88 blocks and 170 us at 700: the second largest item of a training step.)
__global__ __launch_bounds__(256) void dw_reduce_kernel(int nblk, int OK, int O, const float* __restrict__ part, float* __restrict__ dW, float* __restrict__ db) {
    __shared__ float sm[8][32];
    const int l = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + l;
    const int per = (nblk + 7) >> 3, b0 = sl * per, b1 = min(nblk, b0 + per);
    float acc = 0.f;
    if (i < OK + O)
        for (int b = b0; b < b1; ++b) acc += part[(size_t)b * (OK + O) + i];
    sm[sl][l] = acc;
    __syncthreads();
    if (sl == 0 && i < OK + O) {
        float t = sm[0][l];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += sm[q][l];
        if (i < OK) dW[i] += t;
        else if (db) db[i - OK] += t;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The three GEMM shapes of the training path on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32: an exact k-ordered fmaf
// chain, so results do not depend on scheduling).  Weights stay in the state_dict's row-major layout -- they change every
// optimizer step, so nothing is pre-packed: the A / B operands are read with their natural strides.
//   D[i][j] += sum_k A[i][k] B[k][j];  A[i][k] in lane i + 32 k', B[k][j] in lane j + 32 k' (k = 2 s + k'),
//   D[i][j] in lane j + 32 h, register r with i = phi(r, h).
// ---------------------------------------------------------------------------------------------------------------------
typedef float t_f32x16 __attribute__((ext_vector_type(16)));
typedef float t_f32x4 __attribute__((ext_vector_type(4)));

// Y[r, o] = act(sum_k X[r, k] W[o, k] + b[o]): one wave per (32-row tile, 32-output tile); A = W rows, B = X rows.
__global__ __launch_bounds__(256) void rows_linear_mfma_kernel(int R, int K, int O, const float* __restrict__ X, const float* __restrict__ W,
                                                               const float* __restrict__ b, float* __restrict__ Y, int relu) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int otiles = (O + 31) >> 5;
    const int job = blockIdx.x * 4 + wave;
    const int rt = job / otiles, ot = job % otiles;
    if (rt * 32 >= R) return;
    const int row = rt * 32 + j, o = ot * 32 + j;
    const float* xr = X + (size_t)(row < R ? row : 0) * K;
    const float* wr = W + (size_t)(o < O ? o : 0) * K;
    const float xm = row < R ? 1.f : 0.f, wm = o < O ? 1.f : 0.f;
    t_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if ((K & 3) == 0) {
        int k4 = 0;
        for (; k4 + 16 <= K; k4 += 16) {                    // four float4 pairs in flight per round
            t_f32x4 xv[4], wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xv[u] = *reinterpret_cast<const t_f32x4*>(xr + k4 + 4 * u);
                wv[u] = *reinterpret_cast<const t_f32x4*>(wr + k4 + 4 * u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u][h] * wm, xv[u][h] * xm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u][2 + h] * wm, xv[u][2 + h] * xm, acc, 0, 0, 0);
            }
        }
        for (; k4 < K; k4 += 4) {
            const t_f32x4 xv = *reinterpret_cast<const t_f32x4*>(xr + k4), wv = *reinterpret_cast<const t_f32x4*>(wr + k4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[h] * wm, xv[h] * xm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[2 + h] * wm, xv[2 + h] * xm, acc, 0, 0, 0);
        }
    } else {
        for (int k = 0; k < K; k += 2) {
            const bool ok = k + h < K;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? wr[k + h] * wm : 0.f, ok ? xr[k + h] * xm : 0.f, acc, 0, 0, 0);
        }
    }
    if (row >= R) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int oo = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (oo < O) {
            const float val = acc[r] + (b ? b[oo] : 0.f);
            Y[(size_t)row * O + oo] = (relu && val < 0.f) ? 0.f : val;
        }
    }
}

// dX[r, k] (+)= sum_o dY[r, o] W[o, k]: one wave per (32-row tile, 32-column tile of k); A[i = k][o] = W[o, k], B[o][j = row] = dY
__global__ __launch_bounds__(256) void rows_linear_dx_mfma_kernel(int R, int K, int O, const float* __restrict__ dY, const float* __restrict__ W,
                                                                  float* __restrict__ dX, int accumulate) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int ktiles = (K + 31) >> 5;
    const int job = blockIdx.x * 4 + wave;
    const int rt = job / ktiles, kt = job % ktiles;
    if (rt * 32 >= R) return;
    const int row = rt * 32 + j, kc = kt * 32 + j;
    const float* dyr = dY + (size_t)(row < R ? row : 0) * O;
    const float ym = row < R ? 1.f : 0.f, wm = kc < K ? 1.f : 0.f;
    const int kcc = kc < K ? kc : 0;
    t_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int o = 0; o < O; o += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = o + 2 * u + h < O;
            const int oo = ok ? o + 2 * u + h : 0;
            av[u] = W[(size_t)oo * K + kcc] * (ok ? wm : 0.f);
            bv[u] = dyr[oo] * (ok ? ym : 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    if (row >= R) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int kk = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (kk < K) {
            const size_t idx = (size_t)row * K + kk;
            dX[idx] = accumulate ? dX[idx] + acc[r] : acc[r];
        }
    }
}

// partial dW of one chunk of kDwRowsM rows: part[blk][o, k] = sum_{r in chunk} dY[r, o] X[r, k], part[blk][O K + o] = sum_r dY[r, o].
// A[i = o][r] = dY[r, o], B[r][j = k] = X[r, k].  Round 5 (the two dW kernels were 52 % of a training step's GPU time,
// profiles/r05_train_step.txt): wave w owns ROWS r0 + 32 w .. + 32 of the chunk for every (o-tile, k-tile) pair -- all 32 operand loads
// of a pair are in flight before its 16 MFMAs, a 32 x 32 layer keeps four waves busy instead of one -- and the four partial tiles are
// added through LDS in ascending wave order; the bias gradient is column K of the same product (X extended by a column of ones) instead
// of a serial 128-row loop per output.  Still no float atomics and a fixed summation order: the same inputs give the same bits.
constexpr int kDwRowsM = 128;
constexpr int kDwPairs = 4;             // (o-tile, k-tile) pairs per workgroup: blockIdx.y walks the pair groups (the smoother's 128 x 385 layer has 52 pairs
                                        // over ~8 row chunks: one workgroup per chunk left the device empty)
__global__ __launch_bounds__(256) void rows_linear_dw_mfma_kernel(int R, int K, int O, const float* __restrict__ dY,
                                                                  const float* __restrict__ X, float* __restrict__ part) {
    __shared__ float red[4][1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int r0 = blockIdx.x * kDwRowsM, r1 = min(R, r0 + kDwRowsM);
    const int rw0 = r0 + 32 * wave;
    float* out = part + (size_t)blockIdx.x * ((size_t)O * K + O);
    const int otiles = (O + 31) >> 5, ktiles = (K + 1 + 31) >> 5;          // column K: ones (bias gradient)
    const int t_begin = blockIdx.y * kDwPairs, t_end = min(otiles * ktiles, t_begin + kDwPairs);
    for (int t = t_begin; t < t_end; ++t) {
        const int ot = t / ktiles, kt = t % ktiles;
        const int o = ot * 32 + j, kc = kt * 32 + j;
        const int oc = o < O ? o : 0, kcc = kc < K ? kc : 0;
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = rw0 + 2 * u + h;
            const bool ok = r < r1;
            const size_t rr = ok ? r : r0;
            const float a = dY[rr * O + oc], x = X[rr * K + kcc];
            av[u] = (ok && o < O) ? a : 0.f;
            bv[u] = !ok ? 0.f : (kc < K ? x : (kc == K ? 1.f : 0.f));
        }
        t_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        if (t > t_begin) __syncthreads();                                   // the previous pair's partial tiles have been read
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][r * 64 + lane] = acc[r];
        __syncthreads();
        for (int e = threadIdx.x; e < 1024; e += 256) {
            const float v = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
            const int r = e >> 6, l = e & 63;
            const int oo = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), kk = kt * 32 + (l & 31);
            if (oo < O) {
                if (kk < K) out[(size_t)oo * K + kk] = v;
                else if (kk == K) out[(size_t)O * K + oo] = v;
            }
        }
    }
}

__global__ void relu_bwd_kernel(size_t n, const float* __restrict__ y, float* __restrict__ dy) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(y[i] > 0.f)) dy[i] = 0.f;
}

__global__ void fill_kernel(size_t n, float* __restrict__ x, float val) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = val;
}

// ---- node-level inputs.  padded node row n of graph g: caller row node_ptr[g] + (n - node_ptr_pad[g]) if inside
__device__ __forceinline__ int caller_row(const TrainGeom& q, int n, int& g) {
    g = q.ntile_graph[n >> 5];
    if (g < 0) return -1;
    const int local = n - q.node_ptr_pad[g];
    return local < q.node_ptr[g + 1] - q.node_ptr[g] ? q.node_ptr[g] + local : -1;
}

// NCin[n] = [v, g, (v-g)^2, v-g]  (model.py:119), zeros for padding rows
__global__ void node_in_kernel(TrainGeom q, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int C = q.C;
    if (i >= (size_t)q.Npad * 4 * C) return;
    const int n = (int)(i / (4 * C)), k = (int)(i % (4 * C));
    int g;
    const int row = caller_row(q, n, g);
    float val = 0.f;
    if (row >= 0) {
        const int part = k / C, c = k % C;
        const float x = q.v[(size_t)row * C + c], gg = q.goal[(size_t)g * C + c];
        const float dlt = x - gg;
        val = part == 0 ? x : (part == 1 ? gg : (part == 2 ? dlt * dlt : dlt));
    }
    out[i] = val;
}

// ECin[e] = [v_src, v_dst]  (model.py:120), zeros for padding slots
__global__ void edge_in_kernel(TrainGeom q, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int C = q.C;
    if (i >= (size_t)q.Epad * 2 * C) return;
    const int e = (int)(i / (2 * C)), k = (int)(i % (2 * C));
    const int4 rec = q.csr[e];
    float val = 0.f;
    if (rec.x >= 0) {
        int g;
        const int row = caller_row(q, k < C ? rec.x : rec.y, g);
        val = q.v[(size_t)row * C + (k % C)];
    }
    out[i] = val;
}

// H0[n] = goal_encoder on the goal node, else 0  (model.py:133-134)
__global__ void h0_kernel(TrainGeom q, int D, const float* __restrict__ goal_encoder, float* __restrict__ H0) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)q.Npad * D) return;
    const int n = (int)(i / D), f = (int)(i % D);
    const int g = q.ntile_graph[n >> 5];
    H0[i] = (g >= 0 && q.goal_node[g] == n) ? goal_encoder[f] : 0.f;
}
__global__ void h0_bwd_kernel(TrainGeom q, int D, const float* __restrict__ dH0, float* __restrict__ d_goal_encoder) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= D) return;
    float acc = 0.f;                                        // graphs in order: deterministic
    for (int g = 0; g < q.G; ++g)
        if (q.goal_node[g] >= 0) acc += dH0[(size_t)q.goal_node[g] * D + f];
    d_goal_encoder[f] += acc;
}

// out[n] = [a0[n], a1[n], a2[n], a3[n]] (row-major concat of up to four [Npad, D] arrays; nullptr parts are skipped)
__global__ void concat_rows_kernel(int R, int D, int parts, const float* a0, const float* a1, const float* a2, const float* a3,
                                   float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * parts * D) return;
    const int r = (int)(i / (parts * D)), k = (int)(i % (parts * D));
    const float* src = k / D == 0 ? a0 : (k / D == 1 ? a1 : (k / D == 2 ? a2 : a3));
    out[i] = src[(size_t)r * D + k % D];
}
// dst[r] (+)= d_in[r, part*D : (part+1)*D]
__global__ void split_rows_kernel(int R, int D, int parts, int part, const float* __restrict__ d_in, float* __restrict__ dst,
                                  int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * D) return;
    const int r = (int)(i / D), f = (int)(i % D);
    const float val = d_in[(size_t)r * parts * D + part * D + f];
    dst[i] = accumulate ? dst[i] + val : val;
}

// Zin[e] = [X_s - X_t, X_s, X_t, EF_e, EC_e]  (model.py:38-39,142); zeros for padding slots
__global__ void msg_in_kernel(TrainGeom q, int D, const float* __restrict__ X, const float* __restrict__ EF,
                              const float* __restrict__ EC, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)q.Epad * 5 * D) return;
    const int e = (int)(i / (5 * D)), k = (int)(i % (5 * D));
    const int4 rec = q.csr[e];
    float val = 0.f;
    if (rec.x >= 0) {
        const int part = k / D, f = k % D;
        const float xs = X[(size_t)rec.x * D + f], xt = X[(size_t)rec.y * D + f];
        val = part == 0 ? xs - xt : (part == 1 ? xs : (part == 2 ? xt : (part == 3 ? EF[(size_t)e * D + f] : EC[(size_t)e * D + f])));
    }
    out[i] = val;
}
// The prep stage ranks the edges of a target by ARRIVAL (LDS atomics): fine for inference (max aggregation is order-free,
// scores go back to caller order), but the training path sums over CSR slots, so the slot order must not depend on timing.
// One thread per node sorts its segment by caller column before anything reads the CSR.
__global__ void csr_sort_segments_kernel(int Npad, int4* __restrict__ csr, const int* __restrict__ row_beg, const int* __restrict__ deg) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Npad) return;
    int4* a = csr + row_beg[n];
    const int m = deg[n];
    for (int i = 1; i < m; ++i) {
        const int4 key = a[i];
        int j = i - 1;
        while (j >= 0 && a[j].z > key.z) { a[j + 1] = a[j]; --j; }
        a[j + 1] = key;
    }
}

// ---- edges grouped by SOURCE (the CSR is grouped by target): out_beg / out_cnt per padded node, out_slot = the CSR slots
// of the node's outgoing edges in ascending slot order.  Built once per training forward (integer atomics give the counts
// and an arbitrary fill order; every node's short list is then sorted), it lets the adjoints of the row gathers run
// node-centric -- a fixed summation order instead of float atomics.
__global__ void out_count_kernel(TrainGeom q) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= q.Epad) return;
    const int4 rec = q.csr[e];
    if (rec.x >= 0) atomicAdd(&q.out_cnt[rec.x], 1);
}
__global__ __launch_bounds__(1024) void out_scan_kernel(TrainGeom q) {          // one block: exclusive scan of out_cnt over Npad
    __shared__ int sh[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < q.Npad; base += 1024) {
        const int n = base + threadIdx.x;
        const int c = n < q.Npad ? q.out_cnt[n] : 0;
        sh[threadIdx.x] = c;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (n < q.Npad) { q.out_beg[n] = carry + sh[threadIdx.x] - c; q.out_cur[n] = 0; }
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
}
__global__ void out_fill_kernel(TrainGeom q) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= q.Epad) return;
    const int4 rec = q.csr[e];
    if (rec.x >= 0) q.out_slot[q.out_beg[rec.x] + atomicAdd(&q.out_cur[rec.x], 1)] = e;
}
__global__ void out_sort_kernel(TrainGeom q) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= q.Npad) return;
    int* a = q.out_slot + q.out_beg[n];
    const int m = q.out_cnt[n];
    for (int i = 1; i < m; ++i) {                           // insertion sort: out-degrees are tens
        const int key = a[i];
        int j = i - 1;
        while (j >= 0 && a[j] > key) { a[j + 1] = a[j]; --j; }
        a[j + 1] = key;
    }
}

// adjoint of msg_in: dX[s] += dZ[:, 0:D] + dZ[:, D:2D];  dX[t] += -dZ[:, 0:D] + dZ[:, 2D:3D];  dEC[e] += dZ[:, 4D:5D]  (EF detached)
// node n collects its incoming slots (its CSR segment, as target) and then its outgoing slots (out list, as source)
__global__ void msg_in_bwd_nodes_kernel(TrainGeom q, int D, const float* __restrict__ dZ, float* __restrict__ dX) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)q.Npad * D) return;
    const int n = (int)(i / D), f = (int)(i % D);
    float acc = 0.f;
    const int b = q.row_beg[n], dg = q.deg[n];
    for (int sl = b; sl < b + dg; ++sl) { const float* z = dZ + (size_t)sl * 5 * D; acc += z[2 * D + f] - z[f]; }
    const int* os = q.out_slot + q.out_beg[n];
    for (int k = 0; k < q.out_cnt[n]; ++k) { const float* z = dZ + (size_t)os[k] * 5 * D; acc += z[f] + z[D + f]; }
    dX[i] += acc;
}
__global__ void msg_in_bwd_edges_kernel(TrainGeom q, int D, const float* __restrict__ dZ, float* __restrict__ dEC) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)q.Epad * D) return;
    const int e = (int)(i / D), f = (int)(i % D);
    if (q.csr[e].x >= 0) dEC[i] += dZ[(size_t)e * 5 * D + 4 * D + f];
}

// Pin[e] = [D_s, D_s - D_t, EF_e]  (model.py:145)
__global__ void pol_in_kernel(TrainGeom q, int D, const float* __restrict__ Dn, const float* __restrict__ EF, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)q.Epad * 3 * D) return;
    const int e = (int)(i / (3 * D)), k = (int)(i % (3 * D));
    const int4 rec = q.csr[e];
    float val = 0.f;
    if (rec.x >= 0) {
        const int part = k / D, f = k % D;
        const float ds = Dn[(size_t)rec.x * D + f], dt = Dn[(size_t)rec.y * D + f];
        val = part == 0 ? ds : (part == 1 ? ds - dt : EF[(size_t)e * D + f]);
    }
    out[i] = val;
}
// adjoint of pol_in: dDn[s] += dP[:, 0:D] + dP[:, D:2D];  dDn[t] -= dP[:, D:2D]  -- node-centric like msg_in_bwd_nodes_kernel
__global__ void pol_in_bwd_kernel(TrainGeom q, int D, const float* __restrict__ dP, float* __restrict__ dDn) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)q.Npad * D) return;
    const int n = (int)(i / D), f = (int)(i % D);
    float acc = 0.f;
    const int b = q.row_beg[n], dg = q.deg[n];
    for (int sl = b; sl < b + dg; ++sl) acc -= dP[(size_t)sl * 3 * D + D + f];
    const int* os = q.out_slot + q.out_beg[n];
    for (int k = 0; k < q.out_cnt[n]; ++k) { const float* z = dP + (size_t)os[k] * 3 * D; acc += z[f] + z[D + f]; }
    dDn[i] += acc;
}

// A[n, f] = max over the node's CSR segment of M[slot, f]; arg = that slot (first maximum); 0 / -1 without incoming edges
__global__ void segment_max_kernel(TrainGeom q, int D, const float* __restrict__ M, float* __restrict__ A, int* __restrict__ arg) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)q.Npad * D) return;
    const int n = (int)(i / D), f = (int)(i % D);
    const int b = q.row_beg[n], dg = q.deg[n];
    float best = 0.f;
    int bi = -1;
    for (int s = b; s < b + dg; ++s) {
        const float val = M[(size_t)s * D + f];
        if (bi < 0 || val > best) { best = val; bi = s; }
    }
    A[i] = best;
    arg[i] = bi;
}
// dM[arg[n, f], f] = dA[n, f] (dM zero-filled by the caller; every (slot, f) belongs to one node)
__global__ void segment_max_bwd_kernel(int Npad, int D, const float* __restrict__ dA, const int* __restrict__ arg, float* __restrict__ dM) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Npad * D) return;
    const int s = arg[i];
    if (s >= 0) dM[(size_t)s * D + i % D] = dA[i];
}

// scores in caller column order <-> CSR slot order
__global__ void scores_out_kernel(TrainGeom q, const float* __restrict__ slot_scores, float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= q.Epad) return;
    const int4 rec = q.csr[e];
    if (rec.x >= 0) out[rec.z] = slot_scores[e];
}
__global__ void scores_in_kernel(TrainGeom q, const float* __restrict__ d_out, float* __restrict__ d_slot) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= q.Epad) return;
    const int4 rec = q.csr[e];
    d_slot[e] = rec.x >= 0 ? d_out[rec.z] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
static inline unsigned blocks(size_t n) { return (unsigned)((n + 255) / 256); }

hipError_t t_linear(int R, int K, int O, const float* X, const float* W, const float* b, float* Y, bool relu, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    if (O >= 8) {       // narrow outputs (policy.4: one column) stay on the plain kernel
        const size_t jobs = (size_t)((R + 31) / 32) * ((O + 31) / 32);
        hipLaunchKernelGGL(rows_linear_mfma_kernel, dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0, st, R, K, O, X, W, b, Y, relu ? 1 : 0);
        TRAIN_LAUNCH_CHECK();
        return hipSuccess;
    }
    hipLaunchKernelGGL(rows_linear_kernel, dim3(blocks((size_t)R * O)), dim3(256), 0, st, R, K, O, X, W, b, Y, relu ? 1 : 0);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_linear_dx(int R, int K, int O, const float* dY, const float* W, float* dX, bool accumulate, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    if (O >= 8 && K >= 8) {
        const size_t jobs = (size_t)((R + 31) / 32) * ((K + 31) / 32);
        hipLaunchKernelGGL(rows_linear_dx_mfma_kernel, dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0, st, R, K, O, dY, W, dX, accumulate ? 1 : 0);
        TRAIN_LAUNCH_CHECK();
        return hipSuccess;
    }
    hipLaunchKernelGGL(rows_linear_dx_kernel, dim3(blocks((size_t)R * K)), dim3(256), 0, st, R, K, O, dY, W, dX, accumulate ? 1 : 0);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
size_t t_linear_dw_scratch_floats(int R, int K, int O) { return (size_t)((R + kDwRows - 1) / kDwRows) * ((size_t)O * K + O); }
hipError_t t_linear_dw(int R, int K, int O, const float* dY, const float* X, float* dW, float* db, float* scratch, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    int nblk = (R + kDwRows - 1) / kDwRows;
    if (O >= 8 && K >= 4) {
        nblk = (R + kDwRowsM - 1) / kDwRowsM;             // fewer, larger chunks: the scratch sized for kDwRows covers them
        const int pairs = ((O + 31) / 32) * ((K + 1 + 31) / 32);
        hipLaunchKernelGGL(rows_linear_dw_mfma_kernel, dim3(nblk, (pairs + kDwPairs - 1) / kDwPairs), dim3(256), 0, st, R, K, O, dY, X, scratch);
    } else {
        hipLaunchKernelGGL(rows_linear_dw_partial_kernel, dim3(nblk), dim3(256), 0, st, R, K, O, dY, X, scratch);
    }
    TRAIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)(((size_t)O * K + O + 31) / 32)), dim3(256), 0, st, nblk, O * K, O, scratch, dW, db);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_sort_csr(int Npad, int4* csr, const int* row_beg, const int* deg, hipStream_t st) {
    hipLaunchKernelGGL(csr_sort_segments_kernel, dim3(blocks((size_t)Npad)), dim3(256), 0, st, Npad, csr, row_beg, deg);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_out_csr(const TrainGeom& q, hipStream_t st) {
    hipError_t e = hipMemsetAsync(q.out_cnt, 0, sizeof(int) * (size_t)q.Npad, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(out_count_kernel, dim3(blocks((size_t)q.Epad)), dim3(256), 0, st, q);
    TRAIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(out_scan_kernel, dim3(1), dim3(1024), 0, st, q);
    TRAIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(out_fill_kernel, dim3(blocks((size_t)q.Epad)), dim3(256), 0, st, q);
    TRAIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(out_sort_kernel, dim3(blocks((size_t)q.Npad)), dim3(256), 0, st, q);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_relu_bwd(size_t n, const float* y, float* dy, hipStream_t st) {
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks(n)), dim3(256), 0, st, n, y, dy);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_fill(size_t n, float* x, float val, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_kernel, dim3(blocks(n)), dim3(256), 0, st, n, x, val);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_node_in(const TrainGeom& q, float* out, hipStream_t st) {
    hipLaunchKernelGGL(node_in_kernel, dim3(blocks((size_t)q.Npad * 4 * q.C)), dim3(256), 0, st, q, out);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_edge_in(const TrainGeom& q, float* out, hipStream_t st) {
    hipLaunchKernelGGL(edge_in_kernel, dim3(blocks((size_t)q.Epad * 2 * q.C)), dim3(256), 0, st, q, out);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_h0(const TrainGeom& q, int D, const float* ge, float* H0, hipStream_t st) {
    hipLaunchKernelGGL(h0_kernel, dim3(blocks((size_t)q.Npad * D)), dim3(256), 0, st, q, D, ge, H0);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_h0_bwd(const TrainGeom& q, int D, const float* dH0, float* dge, hipStream_t st) {
    hipLaunchKernelGGL(h0_bwd_kernel, dim3(blocks((size_t)D)), dim3(256), 0, st, q, D, dH0, dge);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_concat(int R, int D, int parts, const float* a0, const float* a1, const float* a2, const float* a3, float* out,
                    hipStream_t st) {
    hipLaunchKernelGGL(concat_rows_kernel, dim3(blocks((size_t)R * parts * D)), dim3(256), 0, st, R, D, parts, a0, a1, a2, a3, out);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_split(int R, int D, int parts, int part, const float* d_in, float* dst, bool accumulate, hipStream_t st) {
    hipLaunchKernelGGL(split_rows_kernel, dim3(blocks((size_t)R * D)), dim3(256), 0, st, R, D, parts, part, d_in, dst, accumulate ? 1 : 0);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_msg_in(const TrainGeom& q, int D, const float* X, const float* EF, const float* EC, float* out, hipStream_t st) {
    hipLaunchKernelGGL(msg_in_kernel, dim3(blocks((size_t)q.Epad * 5 * D)), dim3(256), 0, st, q, D, X, EF, EC, out);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_msg_in_bwd(const TrainGeom& q, int D, const float* dZ, float* dX, float* dEC, hipStream_t st) {
    hipLaunchKernelGGL(msg_in_bwd_nodes_kernel, dim3(blocks((size_t)q.Npad * D)), dim3(256), 0, st, q, D, dZ, dX);
    TRAIN_LAUNCH_CHECK();
    hipLaunchKernelGGL(msg_in_bwd_edges_kernel, dim3(blocks((size_t)q.Epad * D)), dim3(256), 0, st, q, D, dZ, dEC);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_pol_in(const TrainGeom& q, int D, const float* Dn, const float* EF, float* out, hipStream_t st) {
    hipLaunchKernelGGL(pol_in_kernel, dim3(blocks((size_t)q.Epad * 3 * D)), dim3(256), 0, st, q, D, Dn, EF, out);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_pol_in_bwd(const TrainGeom& q, int D, const float* dP, float* dDn, hipStream_t st) {
    hipLaunchKernelGGL(pol_in_bwd_kernel, dim3(blocks((size_t)q.Npad * D)), dim3(256), 0, st, q, D, dP, dDn);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_segment_max(const TrainGeom& q, int D, const float* M, float* A, int* arg, hipStream_t st) {
    hipLaunchKernelGGL(segment_max_kernel, dim3(blocks((size_t)q.Npad * D)), dim3(256), 0, st, q, D, M, A, arg);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_segment_max_bwd(int Npad, int D, const float* dA, const int* arg, float* dM, hipStream_t st) {
    hipLaunchKernelGGL(segment_max_bwd_kernel, dim3(blocks((size_t)Npad * D)), dim3(256), 0, st, Npad, D, dA, arg, dM);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_scores_out(const TrainGeom& q, const float* slot_scores, float* out, hipStream_t st) {
    hipLaunchKernelGGL(scores_out_kernel, dim3(blocks((size_t)q.Epad)), dim3(256), 0, st, q, slot_scores, out);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_scores_in(const TrainGeom& q, const float* d_out, float* d_slot, hipStream_t st) {
    hipLaunchKernelGGL(scores_in_kernel, dim3(blocks((size_t)q.Epad)), dim3(256), 0, st, q, d_out, d_slot);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace gnnmp

// =====================================================================================================================
// smoother training path (train_smoother.py:33-61 through model_smoother.py:104-142 with model.train(): BatchNorm in
// batch-statistics mode).  One problem per call, like the reference's training loop.
// =====================================================================================================================
namespace gnnmp {

// Xin[n] = [coords of node n (path rows: current scaled path; samples: x / scale), one-hot(path, free, collided)]
__global__ void sm_nodes_in_kernel(int P, int F, int Co, int C, float scale, const float* __restrict__ cur,
                                   const float* __restrict__ free_pts, const float* __restrict__ coll, float* __restrict__ out) {
    const int K = C + 3, Nn = P + F + Co;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nn * K) return;
    const int n = i / K, k = i % K;
    const int kind = n < P ? 0 : (n < P + F ? 1 : 2);
    float val;
    if (k < C) val = kind == 0 ? cur[n * C + k] : (kind == 1 ? free_pts[(n - P) * C + k] / scale : coll[(n - P - F) * C + k] / scale);
    else val = (k - C == kind) ? 1.f : 0.f;
    out[i] = val;
}

// BatchNorm1d, training mode, one block per feature: y = (x - mean) * invstd * gamma + beta with the batch's biased
// variance; stats[f] = mean, stats[D + f] = invstd, stats[2 D + f] = unbiased variance (for the running-stat update)
__global__ __launch_bounds__(256) void bn_train_fwd_kernel(int N, int D, const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float* __restrict__ y,
                                                           float* __restrict__ stats, int relu) {
    __shared__ float red[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int n = tid; n < N; n += 256) s += x[(size_t)n * D + f];
    red[tid] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
    const float mean = red[0] / N;
    __syncthreads();
    float q = 0.f;
    for (int n = tid; n < N; n += 256) { const float dlt = x[(size_t)n * D + f] - mean; q = fmaf(dlt, dlt, q); }
    red[tid] = q;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
    const float var = red[0] / N;
    const float invstd = 1.0f / sqrtf(var + eps);
    if (tid == 0) { stats[f] = mean; stats[D + f] = invstd; stats[2 * D + f] = N > 1 ? red[0] / (N - 1) : var; }
    const float g = gamma[f], b = beta[f];
    for (int n = tid; n < N; n += 256) {
        const float val = (x[(size_t)n * D + f] - mean) * invstd * g + b;
        y[(size_t)n * D + f] = (relu && val < 0.f) ? 0.f : val;
    }
}
// dx = gamma * invstd / N * (N dy - sum(dy) - xhat * sum(dy * xhat));  dgamma += sum(dy * xhat);  dbeta += sum(dy)
__global__ __launch_bounds__(256) void bn_train_bwd_kernel(int N, int D, const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ gamma, const float* __restrict__ stats,
                                                           float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float r0[256], r1[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    const float mean = stats[f], invstd = stats[D + f];
    float s0 = 0.f, s1 = 0.f;
    for (int n = tid; n < N; n += 256) {
        const float g = dy[(size_t)n * D + f];
        s0 += g;
        s1 = fmaf(g, (x[(size_t)n * D + f] - mean) * invstd, s1);
    }
    r0[tid] = s0; r1[tid] = s1;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (tid < off) { r0[tid] += r0[tid + off]; r1[tid] += r1[tid + off]; } __syncthreads(); }
    const float sum_dy = r0[0], sum_dyx = r1[0];
    if (tid == 0) { dgamma[f] += sum_dyx; dbeta[f] += sum_dy; }        // one block per feature: the only writer of slot f
    const float k = gamma[f] * invstd / N;
    for (int n = tid; n < N; n += 256) {
        const float xhat = (x[(size_t)n * D + f] - mean) * invstd;
        dx[(size_t)n * D + f] = k * (N * dy[(size_t)n * D + f] - sum_dy - xhat * sum_dyx);
    }
}

// Zin[e] = [X_s - X_t, X_s, X_t]  (model_smoother.py:36-37); edges [0, *n_edges)
__global__ void sm_msg_in_kernel(const int* __restrict__ n_edges, int D, const int* __restrict__ e_src, const int* __restrict__ e_dst,
                                 const float* __restrict__ X, float* __restrict__ out, int cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap * 3 * D) return;
    const int e = i / (3 * D), k = i % (3 * D);
    float val = 0.f;
    if (e < *n_edges) {
        const int part = k / D, f = k % D;
        const float xs = X[(size_t)e_src[e] * D + f], xt = X[(size_t)e_dst[e] * D + f];
        val = part == 0 ? xs - xt : (part == 1 ? xs : xt);
    }
    out[i] = val;
}
// adjoints / aggregation of the smoother, node-centric: a problem has a few hundred edges, so every (node, feature) thread
// simply walks the edge list in order -- a fixed summation order, no float atomics
__global__ void sm_msg_in_bwd_kernel(const int* __restrict__ n_edges, int D, const int* __restrict__ e_src, const int* __restrict__ e_dst,
                                     const float* __restrict__ dZ, float* __restrict__ dX, int n_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * D) return;
    const int n = i / D, f = i % D, ne = *n_edges;
    float acc = 0.f;
    for (int e = 0; e < ne; ++e) {
        const float* z = dZ + (size_t)e * 3 * D;
        if (e_src[e] == n) acc += z[f] + z[D + f];
        if (e_dst[e] == n) acc += z[2 * D + f] - z[f];
    }
    dX[i] += acc;
}
// S[dst] += M[e] (aggr = 'add'); adjoint dM[e] = dS[dst]
__global__ void sm_scatter_add_kernel(const int* __restrict__ n_edges, int D, const int* __restrict__ e_dst, const float* __restrict__ M,
                                      float* __restrict__ S, int n_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * D) return;
    const int n = i / D, f = i % D, ne = *n_edges;
    float acc = 0.f;
    for (int e = 0; e < ne; ++e)
        if (e_dst[e] == n) acc += M[(size_t)e * D + f];
    S[i] += acc;
}
__global__ void sm_scatter_add_bwd_kernel(const int* __restrict__ n_edges, int D, const int* __restrict__ e_dst, const float* __restrict__ dS,
                                          float* __restrict__ dM, int cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap * D) return;
    const int e = i / D;
    dM[i] = e < *n_edges ? dS[(size_t)e_dst[e] * D + i % D] : 0.f;
}
// z = x + y (rows R x D)
__global__ void add_rows_kernel(size_t n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
// path_next = path_prev with rows 1 .. P-2 replaced by `proposal` (model_smoother.py:139)
__global__ void sm_path_update_kernel(int P, int C, const float* __restrict__ prev, const float* __restrict__ proposal, float* __restrict__ next) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * C) return;
    const int n = i / C;
    next[i] = (n >= 1 && n <= P - 2) ? proposal[i] : prev[i];
}
// adjoint: d_proposal = d_next on interior rows (0 elsewhere); d_prev = d_next on the two end rows (0 elsewhere)
__global__ void sm_path_update_bwd_kernel(int P, int C, const float* __restrict__ d_next, float* __restrict__ d_proposal, float* __restrict__ d_prev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * C) return;
    const int n = i / C;
    const bool inner = n >= 1 && n <= P - 2;
    d_proposal[i] = inner ? d_next[i] : 0.f;
    d_prev[i] = inner ? 0.f : d_next[i];
}
// d_prev[n, c] += dXin[n, c] for path rows (the coordinates of the next iteration's path nodes ARE this path)
__global__ void sm_coords_bwd_kernel(int P, int C, const float* __restrict__ dXin, float* __restrict__ d_prev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * C) return;
    d_prev[i] += dXin[(size_t)(i / C) * (C + 3) + i % C];
}
__global__ void scale_rows_kernel(int n, float s, const float* __restrict__ x, float* __restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * s;
}

hipError_t t_sm_nodes_in(int P, int F, int Co, int C, float scale, const float* cur, const float* free_pts, const float* coll, float* out,
                         hipStream_t st) {
    hipLaunchKernelGGL(sm_nodes_in_kernel, dim3(blocks((size_t)(P + F + Co) * (C + 3))), dim3(256), 0, st, P, F, Co, C, scale, cur, free_pts, coll, out);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_bn_fwd(int N, int D, const float* x, const float* gamma, const float* beta, float* y, float* stats, bool relu, hipStream_t st) {
    hipLaunchKernelGGL(bn_train_fwd_kernel, dim3(D), dim3(256), 0, st, N, D, x, gamma, beta, 1e-5f, y, stats, relu ? 1 : 0);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_bn_bwd(int N, int D, const float* x, const float* dy, const float* gamma, const float* stats, float* dx, float* dgamma,
                    float* dbeta, hipStream_t st) {
    hipLaunchKernelGGL(bn_train_bwd_kernel, dim3(D), dim3(256), 0, st, N, D, x, dy, gamma, stats, dx, dgamma, dbeta);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_sm_msg_in(const int* n_edges, int D, const int* e_src, const int* e_dst, const float* X, float* out, int cap, hipStream_t st) {
    hipLaunchKernelGGL(sm_msg_in_kernel, dim3(blocks((size_t)cap * 3 * D)), dim3(256), 0, st, n_edges, D, e_src, e_dst, X, out, cap);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_sm_msg_in_bwd(const int* n_edges, int D, const int* e_src, const int* e_dst, const float* dZ, float* dX, int n_rows, hipStream_t st) {
    hipLaunchKernelGGL(sm_msg_in_bwd_kernel, dim3(blocks((size_t)n_rows * D)), dim3(256), 0, st, n_edges, D, e_src, e_dst, dZ, dX, n_rows);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_sm_scatter_add(const int* n_edges, int D, const int* e_dst, const float* M, float* S, int n_rows, hipStream_t st) {
    hipLaunchKernelGGL(sm_scatter_add_kernel, dim3(blocks((size_t)n_rows * D)), dim3(256), 0, st, n_edges, D, e_dst, M, S, n_rows);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_sm_scatter_add_bwd(const int* n_edges, int D, const int* e_dst, const float* dS, float* dM, int cap, hipStream_t st) {
    hipLaunchKernelGGL(sm_scatter_add_bwd_kernel, dim3(blocks((size_t)cap * D)), dim3(256), 0, st, n_edges, D, e_dst, dS, dM, cap);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_add_rows(size_t n, const float* a, const float* b, float* out, hipStream_t st) {
    hipLaunchKernelGGL(add_rows_kernel, dim3(blocks(n)), dim3(256), 0, st, n, a, b, out);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_sm_path_update(int P, int C, const float* prev, const float* proposal, float* next, hipStream_t st) {
    hipLaunchKernelGGL(sm_path_update_kernel, dim3(blocks((size_t)P * C)), dim3(256), 0, st, P, C, prev, proposal, next);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_sm_path_update_bwd(int P, int C, const float* d_next, float* d_proposal, float* d_prev, hipStream_t st) {
    hipLaunchKernelGGL(sm_path_update_bwd_kernel, dim3(blocks((size_t)P * C)), dim3(256), 0, st, P, C, d_next, d_proposal, d_prev);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_sm_coords_bwd(int P, int C, const float* dXin, float* d_prev, hipStream_t st) {
    hipLaunchKernelGGL(sm_coords_bwd_kernel, dim3(blocks((size_t)P * C)), dim3(256), 0, st, P, C, dXin, d_prev);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t t_scale(int n, float s, const float* x, float* y, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks((size_t)n)), dim3(256), 0, st, n, s, x, y);
    TRAIN_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace gnnmp
