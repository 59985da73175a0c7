// layout.hpp -- packed-weight and workspace layouts shared by host packing code and kernels.
// All offsets / sizes are in floats and multiples of 4 (16-byte aligned sections).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace gnnmp {

constexpr int kPrepMaxParts = 16;   // workgroups per graph of the prep stage's two-launch form
constexpr int kPad = 256;          // per-graph padding granularity of the node and edge index spaces
constexpr int kRowsPerWave = 32;

// feature (within a 32-feature tile) held by accumulator register r of a lane in half-wave h
__host__ __device__ constexpr int phi(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// P = operand precision of the packed matrices: 0 = fp32 (1024 floats per 32x32 tile), 1 = bf16 (512 floats
// of storage per tile: 2 MFMAs x 64 lanes x 8 bf16), 2 = bf16x3 (three bf16 pieces per fp32 weight, 1536 floats:
// [piece][K half][lane][8]).  Vectors (bias, LayerNorm) are always fp32.
__host__ __device__ constexpr int tile_unit(int P) { return P == 0 ? 1024 : (P == 1 ? 512 : 1536); }
__host__ __device__ constexpr int tile_floats(int D, int P = 0) { return (D / 32) * (D / 32) * tile_unit(P); }   // one DxD matrix
__host__ __device__ constexpr int vec_floats(int D) { return (D / 32) * 32; }                 // one D-vector
// first layers on K raw inputs: fp32 -> ceil(K/2) steps of 64 floats; bf16 -> ceil(K/16) steps of 64 x 8 bf16
__host__ __device__ constexpr int small_steps(int K, int P) { return P ? (K + 15) / 16 : (K + 1) / 2; }
__host__ __device__ constexpr int small_floats(int D, int ksteps, int P = 0) {
    return (D / 32) * ksteps * (P == 0 ? 64 : (P == 1 ? 256 : 768));
}

// One attention Block (model.py:204-218).  The query and key projections only ever meet in inner products
// (model.py:173-174: q . k_map and q . k_obstacle), so they are stored as ONE matrix Wqk = Wq^T Wk (product in double,
// rounded once): logits = m . (Wqk code_o) against obstacle keys that obs_kernel premultiplies with the same matrix, self
// logit = m . (Wqk m) -- one d x d product per row instead of two, with fewer roundings than the reference's
// (Wq m) . (Wk m).  The four DxD matrices come first (`staged` floats), the six D-vectors behind them.
template <int D, int P = 0>
struct AttBlob {
    static constexpr int T = tile_floats(D, P), V = vec_floats(D);
    static constexpr int wqk = 0, wv = T, w1 = 2 * T, w2 = 3 * T, staged = 4 * T;
    static constexpr int ln1g = 4 * T, ln1b = 4 * T + V, b1 = 4 * T + 2 * V, b2 = 4 * T + 3 * V, ln2g = 4 * T + 4 * V,
                         ln2b = 4 * T + 5 * V, size = 4 * T + 6 * V;
};

// Two Seq(Lin, ReLU, Lin) encoders on raw inputs (node_code + node_free_code, or edge_code +
// edge_free_code): [As0][b0][A0][c0][As1][b1][A1][c1]
struct EncBlob {
    int as0, b0, a0, c0, as1, b1, a1, c1, size, ks0, ks1;
    // K0 / K1: raw input widths of the two encoders
    __host__ __device__ static EncBlob make(int P, int D, int K0, int K1) {
        EncBlob e;
        const int T = tile_floats(D, P), V = vec_floats(D);
        const int ks0 = small_steps(K0, P), ks1 = small_steps(K1, P);
        int o = 0;
        e.ks0 = ks0; e.ks1 = ks1;
        e.as0 = o; o += small_floats(D, ks0, P);
        e.b0 = o; o += V;
        e.a0 = o; o += T;
        e.c0 = o; o += V;
        e.as1 = o; o += small_floats(D, ks1, P);
        e.b1 = o; o += V;
        e.a1 = o; o += T;
        e.c1 = o; o += V;
        e.size = o;
        return e;
    }
};

// Edge epilogue: K_e = W1d.EF + W1e.EC + b1 (message first-layer edge constant),
//                PE  = Wpc.EF + bp0        (policy first-layer edge constant)
template <int D, int P = 0>
struct OutEBlob {
    static constexpr int T = tile_floats(D, P), V = vec_floats(D);
    static constexpr int w1d = 0, w1e = T, b1 = 2 * T, wpc = 2 * T + V, bp0 = 3 * T + V, size = 3 * T + 2 * V;
};

// Node epilogue: XI = We_nc.NC + We_nf.NF + be (+ weg on the goal row); X0 = XI (+ wehg on the goal
// row); A0 = Wsrc.X0; B0 = Wdst.X0; DN = Wd_nc.NC + bd
template <int D, int P = 0>
struct OutNBlob {
    static constexpr int T = tile_floats(D, P), V = vec_floats(D);
    static constexpr int we_nc = 0, we_nf = T, be = 2 * T, weg = 2 * T + V, wehg = 2 * T + 2 * V, wsrc = 2 * T + 3 * V,
                         wdst = 3 * T + 3 * V, wd_nc = 4 * T + 3 * V, bd = 5 * T + 3 * V, size = 5 * T + 4 * V;
};

// Node update of one message-passing iteration: H = Wlx.X + Wla.agg + bl; Y = R + M1.H;
// A' = M2.Y; B' = M3.Y.   (loop body: R = XI, M1 = Weh, M2 = Wsrc, M3 = Wdst;
//                          after the last iteration: R = DN, M1 = Wdh, M2 = Wpa+Wpb, M3 = Wpb)
template <int D, int P = 0>
struct MpNBlob {
    static constexpr int T = tile_floats(D, P), V = vec_floats(D);
    static constexpr int wlx = 0, wla = T, bl = 2 * T, m1 = 2 * T + V, m2 = 3 * T + V, m3 = 4 * T + V, size = 5 * T + V;
};

template <int D, int P = 0>
struct MpEBlob {   // message second layer
    static constexpr int T = tile_floats(D, P), V = vec_floats(D);
    static constexpr int w2 = 0, b2 = T, size = T + V;
};

template <int D, int P = 0>
struct PolBlob {   // policy.2 and policy.4
    static constexpr int T = tile_floats(D, P), V = vec_floats(D);
    static constexpr int w2 = 0, b2 = T, w3 = T + V, size = T + 2 * V;
};

// Obstacle side (node or edge): encoder + per block {Wk, Wv, obs_feed}
struct ObsBlob {
    int as0, b0, a0, c0, blk0, blk_stride, size, ks;
    // inside a block:
    int wk, wv, fw1, fb1, fw2, fb2, lng, lnb;
    __host__ __device__ static ObsBlob make(int P, int D, int K) {
        ObsBlob e;
        const int T = tile_floats(D, P), V = vec_floats(D);
        const int ks = small_steps(K, P);
        int o = 0;
        e.ks = ks;
        e.as0 = o; o += small_floats(D, ks, P);
        e.b0 = o; o += V;
        e.a0 = o; o += T;
        e.c0 = o; o += V;
        e.blk0 = o;
        e.wk = 0; e.wv = T; e.fw1 = 2 * T; e.fb1 = 3 * T; e.fw2 = 3 * T + V; e.fb2 = 4 * T + V;
        e.lng = 4 * T + 2 * V; e.lnb = 4 * T + 3 * V;
        e.blk_stride = 4 * T + 4 * V;
        e.size = o + 3 * e.blk_stride;
        return e;
    }
};

// Node side, block 0, in fp64 (node_f64_body, explorer_kernels.hip): node_free_code encoder + the attention sub-block
// through its LayerNorm, computed with v_mfma_f64_16x16x4_f64.  Matrices are stored as that instruction's A operands,
// fp32 values (converted on load): [out/16][ceil(in/4)][64 lanes], lane l of k-step s holding W[16 ob + l%16][4 s + l/16]
// (0 beyond the input width); vectors in plain feature order.  gnnmp_pack_f64_ops in gnnmp.h.
struct F64Blob {
    int w1, b1, w2, b2, wqk, wv, lng, lnb, size, ks;
    __host__ __device__ static F64Blob make(int D, int C) {
        F64Blob e;
        int o = 0;
        e.ks = (C + 3) / 4;
        e.w1 = o; o += (D / 16) * e.ks * 64;
        e.b1 = o; o += D;
        e.w2 = o; o += D * D;
        e.b2 = o; o += D;
        e.wqk = o; o += D * D;
        e.wv = o; o += D * D;
        e.lng = o; o += D;
        e.lnb = o; o += D;
        e.size = o;
        return e;
    }
};
// obstacles whose K'/V operands sit in LDS -- and whose logits sit in registers -- at a time (node_f64_body): 128 at d = 32;
// 32 at d = 64, where the register file is the limit (and the d = 64 checkpoints are the robot arms with five boxes)
__host__ __device__ constexpr int f64_obs_chunk(int D) { return D == 32 ? 128 : 32; }

// ---- smoother (ModelSmoother, model_smoother.py:46-142)
constexpr int kSmK = 10;            // knn(..., k=10) at model_smoother.py:125

struct SmLayout {
    int ks;                          // K steps of node_code.0 (inputs: C coords + 3 one-hot)
    int as0, b0, w3, b3;             // node_code.0 (BatchNorm folded) and node_code.3
    int wsrc, wdst, b00, w02, b02;   // process.lin_0: (W_a + W_b), (W_c - W_a), bias; second layer
    int w10, b10, w12, b12;          // process.lin_1
    int ws, bs;                      // smooth_node, out features padded to 32
    int total;
    __host__ __device__ static SmLayout make(int P, int D, int C) {
        SmLayout L;
        const int T = tile_floats(D, P), V = vec_floats(D);
        int o = 0;
        L.ks = small_steps(C + 3, P);
        L.as0 = o; o += small_floats(D, L.ks, P);
        L.b0 = o; o += V;
        L.w3 = o; o += T;
        L.b3 = o; o += V;
        L.wsrc = o; o += T;
        L.wdst = o; o += T;
        L.b00 = o; o += V;
        L.w02 = o; o += T;
        L.b02 = o; o += V;
        L.w10 = o; o += T;
        L.b10 = o; o += V;
        L.w12 = o; o += T;
        L.b12 = o; o += V;
        L.ws = o; o += (D / 32) * tile_unit(P);
        L.bs = o; o += 32;
        L.total = o;
        return L;
    }
};

// Offsets of every blob inside the device weight buffer of an explorer handle.
struct ExplorerOffsets {
    int enc_e, enc_n, att_e, att_n, out_e, out_n, mpn, mpn_last, mpe, pol, obs_e, obs_n, f64, total;
};

}  // namespace gnnmp
