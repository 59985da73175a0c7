// chain.hpp -- register-resident MLP chains on the fp32 MFMA of gfx950 (CDNA4).
//
// One wavefront (64 lanes) owns a group of 32 rows (edges, nodes or obstacles).  Every linear
// layer is computed TRANSPOSED, Y^T = W . X^T, with v_mfma_f32_32x32x2_f32:
//     A operand = weights   A[i = out feature][k]        (one f32 per lane: i = lane&31, k = lane>>5)
//     B operand = X^T       B[k][j = row]                (one f32 per lane: j = lane&31, k = lane>>5)
//     D         = Y^T       D[i][j]: lane holds column j = lane&31 and the 16 out features
//                           i = phi(r, h) = (r&3) + 8*(r>>2) + 4*h,   r = register, h = lane>>5.
// So after a layer, lane (j, h) holds 16 features of row j -- which is exactly what the NEXT
// layer's B operand needs if its K loop visits, at step r, feature phi(r,0) in the lower half
// wave and phi(r,1) in the upper one.  Sums over k are order-free, so we simply permute the
// weight columns to that order when packing ("A-tile format", gnnmp_pack_a_tiles in gnnmp.h).
// Result: arbitrarily long chains of Linear / bias / ReLU / residual / LayerNorm / softmax run
// entirely in registers, with no LDS transpose between layers; only the (pre-permuted) weights
// stream in from LDS, 256 B per MFMA per wave.  A row's features live in two lanes (l, l^32),
// so row reductions (LayerNorm, softmax, dot products) are 16 in-lane ops + one cross-half swap.
//
// fp32 MFMA is a k-ordered fmaf chain (exact fp32, 64 FLOP/clk/SIMD = the fp32 peak of the chip).
#pragma once
#include <hip/hip_runtime.h>
#include "layout.hpp"

namespace gnnmp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kTile = 32;             // rows per wave, features per tile
constexpr int kATile = 1024;          // floats in one 32x32 A tile


// A row lives in lanes l and l ^ 32: x(l) (+ | max) x(l ^ 32), the same bits in both lanes.  The exchange goes through
// the LDS crossbar (ds_bpermute).  gfx950's v_permlane32_swap does the same on the VALU; measured on the two pre
// kernels it is a wash for the edge kernel (2.223 -> 2.219 ms) and a loss for the register-tight node kernel
// (0.233 -> 0.250 ms), so the VALU -- which shares its issue time with the fp32 MFMAs -- is left alone here.
// (Round 4 measured the v_permlane32_swap form again, within noise on all three BASELINE shapes, bf16 kernels included.)
__device__ __forceinline__ float xsum(float x) {
    return x + __shfl_xor(x, 32, 64);
}
__device__ __forceinline__ float xmax(float x) {
    return fmaxf(x, __shfl_xor(x, 32, 64));
}

__device__ __forceinline__ f32x16 splat16(float x) {
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = x;
    return v;
}

// ---- precision of the MFMA operands.  P = 0: exact fp32 (v_mfma_f32_32x32x2_f32, 16 instructions per
// 32x32x32 block).  P = 1: operands rounded to bf16 (RNE, v_cvt_pk_bf16_f32), fp32 accumulate
// (v_mfma_f32_32x32x16_bf16, 2 instructions per block); everything outside the MFMA (bias, ReLU,
// LayerNorm, softmax, residuals, max) stays fp32.  The accumulator layout is the same, so the chain
// trick carries over: MFMA m of a block consumes the lane's registers 8m..8m+7 (8 bf16 per lane), and
// the packed weights put W[i][phi(8m + t, lane>>5)] in slot t of lane (i, lane>>5).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

template <int P> struct Prec;
template <> struct Prec<0> { static constexpr int TF = 1024; };      // floats of storage per 32x32 A tile
template <> struct Prec<1> { static constexpr int TF = 512; };
// P = 2 ("bf16x3"): fp32-class results from the bf16 matrix pipe.  Every fp32 operand is split exactly into
// three bf16 pieces x = x0 + x1 + x2 (8 mantissa bits each); a product keeps the six piece pairs with
// i + j <= 2 (dropped terms are <= 2^-24 of the leading one) and all of them accumulate in the same fp32
// accumulator, smallest first.  A 32x32x32 block costs 12 bf16 MFMAs (384 matrix-pipe cycles) instead of 16
// fp32 MFMAs (1024 cycles on the VECTOR pipe), and the matrix pipe overlaps with VALU work.
template <> struct Prec<2> { static constexpr int TF = 1536; };     // 3 pieces x 512

// fp32 -> packed bf16 (round to nearest even), two values per instruction.  Spelled as the instruction: inside the big kernels
// __builtin_convertvector(f32x8 -> bf16x8) was scalarised to one v_cvt_pk_bf16_f32 PER ELEMENT (second source unused) plus a v_perm_b32
// per pair -- 3 instead of 1 VALU instructions per pair, 48 instead of 16 per 32-edge chunk of the bf16 message kernel, in
// kernels that are bound by instruction issue (round 5; isolated test kernels get the packed form, the big ones did not).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
// (NOT as inline asm: the hazard recognizer does not look into asm statements, and a conversion reading an MFMA's result registers
// without the wait states the compiler would have inserted returned garbage.)  The pairwise <2 x float> -> <2 x bfloat> fptrunc is the
// pattern the instruction is selected from.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    const f32x2 p = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf16x2_t));
}
__device__ __forceinline__ bf16x8 to_bf16x8(const f32x8& a) {
    u32x4_t r;
#pragma unroll
    for (int w = 0; w < 4; ++w) r[w] = cvt_pk_bf16(a[2 * w], a[2 * w + 1]);
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ bf16x4_t to_bf16x4(const f32x4& a) {
    u32x2_t r;
    r[0] = cvt_pk_bf16(a[0], a[1]);
    r[1] = cvt_pk_bf16(a[2], a[3]);
    return __builtin_bit_cast(bf16x4_t, r);
}

template <int P> struct BOp;                    // one 32-feature activation tile as MFMA B operand
template <> struct BOp<0> {
    f32x16 v;
    __device__ __forceinline__ BOp() {}
    __device__ __forceinline__ explicit BOp(const f32x16& x) : v(x) {}
};
template <> struct BOp<1> {
    bf16x8 lo, hi;
    __device__ __forceinline__ BOp() {}
    __device__ __forceinline__ explicit BOp(const f32x16& x) {
        f32x8 a, b;
#pragma unroll
        for (int r = 0; r < 8; ++r) { a[r] = x[r]; b[r] = x[8 + r]; }
        lo = to_bf16x8(a);
        hi = to_bf16x8(b);
    }
};

template <> struct BOp<2> {
    bf16x8 lo[3], hi[3];
    __device__ __forceinline__ BOp() {}
    __device__ __forceinline__ explicit BOp(const f32x16& x) {
        f32x8 a, b;
#pragma unroll
        for (int r = 0; r < 8; ++r) { a[r] = x[r]; b[r] = x[8 + r]; }
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            lo[pc] = to_bf16x8(a);
            hi[pc] = to_bf16x8(b);
            if (pc < 2) {                                  // exact residual: a - float(bf16(a)) is representable
                a -= __builtin_convertvector(lo[pc], f32x8);
                b -= __builtin_convertvector(hi[pc], f32x8);
            }
        }
    }
};

// SWAP = false: D = W . X^T (A operand = weight tile, B operand = activations): the lane keeps a ROW (edge / node) and
//                16 output features -- the chain layout.
// SWAP = true:  D = X . W^T (A operand = activations, B operand = the same packed weight tile): the lane keeps an output
//                FEATURE (lane & 31) and 16 ROWS phi(r, h) -- the layout for reductions over rows (max aggregation).
//                Products are commutative and the k order is the same, so every D element is bit-identical to the
//                unswapped form's.
template <int P, bool SWAP = false>
__device__ __forceinline__ void mfma_tile_p(const float* a, const BOp<P>& x, f32x16& acc, int lane) {
    if constexpr (P == 0) {
        // all four 16-byte operand reads of the tile are issued before the first MFMA, so only the first
        // MFMA of a 16-instruction block can wait on LDS
        f32x4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = *reinterpret_cast<const f32x4*>(a + (q * 64 + lane) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (SWAP) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.v[q * 4 + c], w[q][c], acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[q][c], x.v[q * 4 + c], acc, 0, 0, 0);
            }
    } else if constexpr (P == 1) {
        const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(a + lane * 4);
        const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(a + (64 + lane) * 4);
        if constexpr (SWAP) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.lo, w0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.hi, w1, acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x.lo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x.hi, acc, 0, 0, 0);
        }
    } else {
        // weight piece pw of K half m sits at a + ((pw*2 + m)*64 + lane)*4; pairs (weight piece, x piece), smallest first
        bf16x8 w[3][2];
#pragma unroll
        for (int pw = 0; pw < 3; ++pw)
#pragma unroll
            for (int m = 0; m < 2; ++m) w[pw][m] = *reinterpret_cast<const bf16x8*>(a + ((pw * 2 + m) * 64 + lane) * 4);
        constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            if constexpr (SWAP) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.lo[PX[t]], w[PW[t]][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.hi[PX[t]], w[PW[t]][1], acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[PW[t]][0], x.lo[PX[t]], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[PW[t]][1], x.hi[PX[t]], acc, 0, 0, 0);
            }
        }
    }
}

// only K groups 0..nq-1 (8 obstacles each, 4 groups per tile) can be non-zero in x
template <int P>
__device__ __forceinline__ void mfma_tile_q_p(const float* a, const BOp<P>& x, f32x16& acc, int lane, int nq) {
    if (nq >= 4) { mfma_tile_p<P>(a, x, acc, lane); return; }
    if constexpr (P == 0) {
        for (int q = 0; q < nq; ++q) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(a + (q * 64 + lane) * 4);
            f32x4 xb;
            switch (q) {
                case 0: xb = f32x4{x.v[0], x.v[1], x.v[2], x.v[3]}; break;
                case 1: xb = f32x4{x.v[4], x.v[5], x.v[6], x.v[7]}; break;
                default: xb = f32x4{x.v[8], x.v[9], x.v[10], x.v[11]}; break;
            }
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[cidx], xb[cidx], acc, 0, 0, 0);
        }
    } else if constexpr (P == 1) {
        const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(a + lane * 4);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x.lo, acc, 0, 0, 0);            // groups 0, 1
        if (nq > 2) {
            const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(a + (64 + lane) * 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x.hi, acc, 0, 0, 0);        // groups 2, 3
        }
    } else {
        constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(a + ((PW[t] * 2) * 64 + lane) * 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x.lo[PX[t]], acc, 0, 0, 0);
            if (nq > 2) {
                const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(a + ((PW[t] * 2 + 1) * 64 + lane) * 4);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x.hi[PX[t]], acc, 0, 0, 0);
            }
        }
    }
}

// y[ot] += sum_it A[ot][it] . x[it]      (A: [NTO][NTI][Prec<P>::TF] floats of storage)
// operand form: the B operands were prepared once (conversion / 3-way split) and feed several matrices
template <int P, int NTO, int NTI>
__device__ __forceinline__ void linear_acc_ops(const float* A, const BOp<P> (&xb)[NTI], f32x16 (&y)[NTO], int lane) {
#pragma unroll
    for (int it = 0; it < NTI; ++it)
#pragma unroll
        for (int ot = 0; ot < NTO; ++ot)
            mfma_tile_p<P>(A + (ot * NTI + it) * Prec<P>::TF, xb[it], y[ot], lane);
}

template <int P, int NT>
__device__ __forceinline__ void make_ops(const f32x16 (&x)[NT], BOp<P> (&xb)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) xb[t] = BOp<P>(x[t]);
}

template <int P, int NTO, int NTI>
__device__ __forceinline__ void linear_acc_p(const float* A, const f32x16 (&x)[NTI], f32x16 (&y)[NTO], int lane) {
    BOp<P> xb[NTI];
    make_ops<P, NTI>(x, xb);
    linear_acc_ops<P, NTO, NTI>(A, xb, y, lane);
}

// first layer on raw inputs.  P = 0: Asmall [NTO][ceil(K/2)][64] floats, lane supplies in[2 st + h].
// P = 1: Asmall [NTO][ceil(K/16)][64][8] bf16, lane supplies in[16 st + 8 h + t], t = 0..7.
// `ksteps` is the step count of the chosen precision (EncBlob / ObsBlob carry it).
template <int P, int NTO, class GetIn>
__device__ __forceinline__ void linear_in_p(const float* Asmall, int ksteps, GetIn getin, f32x16 (&y)[NTO], int lane);

// per-feature vector in register order: vec[(t*2 + h)*16 + r]
template <int NT>
__device__ __forceinline__ void load_vec(const float* vec, f32x16 (&y)[NT], int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(vec + (t * 2 + h) * 16 + q * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) y[t][q * 4 + c] = b[c];
        }
}

// First layer on raw inputs: y[ot] += sum_k W[.., k] in[k]; lane supplies in[2*st + h] at step st.
// Asmall: [NTO][ksteps][64] floats.
template <int NTO, class GetIn>
__device__ __forceinline__ void linear_in(const float* Asmall, int ksteps, GetIn getin, f32x16 (&y)[NTO], int lane) {
    const int h = lane >> 5;
    // inputs are gathered eight steps at a time so the (dependent, possibly global) loads behind getin
    // are all in flight before the first MFMA needs one
    for (int st0 = 0; st0 < ksteps; st0 += 8) {
        float b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            b[u] = 0.f;
            if (st0 + u < ksteps) b[u] = getin(2 * (st0 + u) + h);        // wave-uniform guard; getin is 0 beyond the width
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (st0 + u < ksteps) {
#pragma unroll
                for (int ot = 0; ot < NTO; ++ot)
                    y[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(Asmall[(ot * ksteps + st0 + u) * 64 + lane], b[u], y[ot], 0, 0, 0);
            }
        }
    }
}

template <int P, int NTO, class GetIn>
__device__ __forceinline__ void linear_in_p(const float* Asmall, int ksteps, GetIn getin, f32x16 (&y)[NTO], int lane) {
    if constexpr (P == 0) {
        linear_in<NTO>(Asmall, ksteps, getin, y, lane);
    } else if constexpr (P == 1) {
        const int h = lane >> 5;
        for (int st = 0; st < ksteps; ++st) {
            f32x8 b;
#pragma unroll
            for (int t = 0; t < 8; ++t) b[t] = getin(16 * st + 8 * h + t);      // getin returns 0 beyond the input width
            const bf16x8 bb = to_bf16x8(b);
#pragma unroll
            for (int ot = 0; ot < NTO; ++ot) {
                const bf16x8 w = *reinterpret_cast<const bf16x8*>(Asmall + ((ot * ksteps + st) * 64 + lane) * 4);
                y[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, bb, y[ot], 0, 0, 0);
            }
        }
    } else {
        // Asmall: [NTO][ksteps][3 pieces][64][8] bf16
        const int h = lane >> 5;
        for (int st = 0; st < ksteps; ++st) {
            f32x8 b;
#pragma unroll
            for (int t = 0; t < 8; ++t) b[t] = getin(16 * st + 8 * h + t);
            bf16x8 xb[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                xb[pc] = to_bf16x8(b);
                if (pc < 2) b -= __builtin_convertvector(xb[pc], f32x8);
            }
            constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int ot = 0; ot < NTO; ++ot)
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const bf16x8 w = *reinterpret_cast<const bf16x8*>(Asmall + (((ot * ksteps + st) * 3 + PW[t]) * 64 + lane) * 4);
                    y[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, xb[PX[t]], y[ot], 0, 0, 0);
                }
        }
    }
}

// y += A . relu(x).  fp32 / bf16x3 operands: the ReLU runs on the fp32 registers.  bf16 operands: the conversion comes first and
// the ReLU runs on the PACKED pairs as a signed 16-bit integer maximum with 0 (negative floats are negative integers, -0 the
// most negative one; rounding to bf16 is monotone and keeps the sign, so this equals rounding the ReLU's output bit for bit):
// 8 instead of 16 VALU instructions per 32-feature tile.  Used by the policy head (bf16: 0.102 -> 0.098 ms at the configs[2] shape);
// in the attention chains (FeedForward, encoders) it measured slightly slower (edge stage 0.429 -> 0.438 ms) and is not used.
typedef short s16x8_t __attribute__((ext_vector_type(8)));
template <int P, int NTO, int NTI>
__device__ __forceinline__ void linear_acc_relu_p(const float* A, const f32x16 (&x)[NTI], f32x16 (&y)[NTO], int lane) {
    BOp<P> xb[NTI];
    if constexpr (P == 1) {
        make_ops<P, NTI>(x, xb);
        const s16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < NTI; ++t) {
            xb[t].lo = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8_t, xb[t].lo), zero));
            xb[t].hi = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8_t, xb[t].hi), zero));
        }
    } else {
        f32x16 r[NTI];
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) r[t][q] = fmaxf(x[t][q], 0.0f);
        make_ops<P, NTI>(r, xb);
    }
    linear_acc_ops<P, NTO, NTI>(A, xb, y, lane);
}

template <int NT>
__device__ __forceinline__ void relu_(f32x16 (&x)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[t][r] = fmaxf(x[t][r], 0.0f);
}

// sum of the 16 registers as a pairwise tree of whole-vector adds (v_pk_add_f32: two lanes' worth per instruction,
// 8 + 1 instructions instead of a 16-deep serial chain)
__device__ __forceinline__ float tree_sum(const f32x16& a) {
    const f32x8 a8 = a.lo + a.hi;
    const f32x4 a4 = a8.lo + a8.hi;
    const f32x2 a2 = a4.lo + a4.hi;
    return a2.x + a2.y;
}

// LayerNorm over the D = 32*NT features of each row (biased variance, eps inside the sqrt).  Written with
// whole-vector arithmetic so that the adds / multiplies / fmas issue as packed fp32 pairs: VALU instructions cost
// MFMA issue time in the kernels that use this.
template <int NT>
__device__ __forceinline__ void layer_norm_(f32x16 (&x)[NT], const float* gamma, const float* beta, float eps, int lane) {
    constexpr float inv_d = 1.0f / (32 * NT);
    f32x16 acc = x[0];
#pragma unroll
    for (int t = 1; t < NT; ++t) acc += x[t];
    float mean = xsum(tree_sum(acc)) * inv_d;
    // keep the mean a value of its own: fused into the subtraction (x - sum * (1/d)) it becomes 16 unpacked fmas per tile
    // instead of 8 packed adds
    asm volatile("" : "+v"(mean));
    f32x16 sq = splat16(0.f);
    const float nmean = -mean;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        x[t] += nmean;                                  // packed adds with the negated mean broadcast
        sq += x[t] * x[t];
    }
    const float v = xsum(tree_sum(sq));
    // 1 / sqrt: hardware estimate + one Newton step (full fp32 accuracy, 5 instructions instead of the ~20 of the IEEE
    // square root and division sequences)
    const float var = v * inv_d + eps;
    float rstd = __builtin_amdgcn_rsqf(var);
    rstd = rstd * (1.5f - 0.5f * var * rstd * rstd);
    f32x16 g[NT], b[NT];
    load_vec<NT>(gamma, g, lane);
    load_vec<NT>(beta, b, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t) x[t] = (x[t] * rstd) * g[t] + b[t];
}

// ---- row-major [rows, D] <-> register order.  Lane (j,h) touches 16-byte pieces
//      [8q + 4h, 8q + 4h + 4) of its row's 32-float tile t.
template <int NT>
__device__ __forceinline__ void load_row(const float* base /* row start */, f32x16 (&x)[NT], int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(base + t * 32 + q * 8 + h * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) x[t][q * 4 + c] = a[c];
        }
}

template <int NT>
__device__ __forceinline__ void store_row(float* base, const f32x16 (&x)[NT], int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a;
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] = x[t][q * 4 + c];
            *reinterpret_cast<f32x4*>(base + t * 32 + q * 8 + h * 4) = a;
        }
}

// ---- "tile-native" storage of a [32 rows, D] register block: [NT][4][64 lanes][4] floats, every
//      store instruction writes 1 KiB contiguous.  Used for private per-edge intermediates.
template <int NT>
__device__ __forceinline__ void store_tile(float* tile_base, const f32x16 (&x)[NT], int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a;
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] = x[t][q * 4 + c];
            *reinterpret_cast<f32x4*>(tile_base + ((t * 4 + q) * 64 + lane) * 4) = a;
        }
}

template <int NT>
__device__ __forceinline__ void load_tile(const float* tile_base, f32x16 (&x)[NT], int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(tile_base + ((t * 4 + q) * 64 + lane) * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) x[t][q * 4 + c] = a[c];
        }
}

// streamed-once variant: non-temporal loads keep the private per-edge streams from evicting the
// gathered node rows out of L2
template <int NT>
__device__ __forceinline__ void load_tile_nt(const float* tile_base, f32x16 (&x)[NT], int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(tile_base + ((t * 4 + q) * 64 + lane) * 4));
#pragma unroll
            for (int c = 0; c < 4; ++c) x[t][q * 4 + c] = a[c];
        }
}

// ---- precision-dependent storage of the per-edge tiles (K_e, PE) and of the gathered node rows (A, B):
//      fp32 in the exact and bf16x3 modes; bf16 (RNE) in the bf16 mode, where these values feed bf16 MFMA
//      operands anyway and the message-passing kernels are bandwidth-bound.  Tile image in bf16:
//      [NT][2][64 lanes][8 bf16]; row image: [N, D] bf16 row-major.
template <int P, int NT>
__device__ __forceinline__ void store_tile_p(float* tile_base_f32_units, const f32x16 (&x)[NT], int lane) {
    if constexpr (P != 1) {
        store_tile<NT>(tile_base_f32_units, x, lane);
    } else {
        __bf16* base = reinterpret_cast<__bf16*>(tile_base_f32_units);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const BOp<1> b(x[t]);
            *reinterpret_cast<bf16x8*>(base + ((t * 2 + 0) * 64 + lane) * 8) = b.lo;
            *reinterpret_cast<bf16x8*>(base + ((t * 2 + 1) * 64 + lane) * 8) = b.hi;
        }
    }
}

template <int P, int NT>
__device__ __forceinline__ void load_tile_nt_p(const float* tile_base_f32_units, f32x16 (&x)[NT], int lane) {
    if constexpr (P != 1) {
        load_tile_nt<NT>(tile_base_f32_units, x, lane);
    } else {
        const __bf16* base = reinterpret_cast<const __bf16*>(tile_base_f32_units);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16x8 lo = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(base + ((t * 2 + 0) * 64 + lane) * 8));
            const bf16x8 hi = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(base + ((t * 2 + 1) * 64 + lane) * 8));
            const f32x8 a = __builtin_convertvector(lo, f32x8), b = __builtin_convertvector(hi, f32x8);
#pragma unroll
            for (int r = 0; r < 8; ++r) { x[t][r] = a[r]; x[t][8 + r] = b[r]; }
        }
    }
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// row_base: start of the row in ELEMENTS of the storage type (caller multiplies the row index by D)
template <int P, int NT>
__device__ __forceinline__ void store_row_p(float* array_f32_units, size_t row, const f32x16 (&x)[NT], int h) {
    if constexpr (P != 1) {
        store_row<NT>(array_f32_units + row * (NT * 32), x, h);
    } else {
        // bf16 rows are stored HALF-MAJOR inside every 32-feature tile: element t * 32 + h * 16 + r holds register r of lane half h
        // (feature t * 32 + (r / 4) * 8 + h * 4 + r % 4), so what one lane reads or writes of a tile is 32 contiguous bytes -- two
        // 16-byte accesses, in LDS the conflict-free pattern of the fp32 rows (explorer_kernels.hip read_stage_tile).  The order is
        // private to the kernels: nothing outside reads a bf16 row.
        __bf16* base = reinterpret_cast<__bf16*>(array_f32_units) + row * (NT * 32);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                f32x8 a;
#pragma unroll
                for (int c = 0; c < 8; ++c) a[c] = x[t][qq * 8 + c];
                *reinterpret_cast<bf16x8*>(base + t * 32 + h * 16 + qq * 8) = to_bf16x8(a);
            }
    }
}

template <int P, int NT>
__device__ __forceinline__ void load_row_p(const float* array_f32_units, size_t row, f32x16 (&x)[NT], int h) {
    if constexpr (P != 1) {
        load_row<NT>(array_f32_units + row * (NT * 32), x, h);
    } else {
        const __bf16* base = reinterpret_cast<const __bf16*>(array_f32_units) + row * (NT * 32);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const f32x8 a = __builtin_convertvector(*reinterpret_cast<const bf16x8*>(base + t * 32 + h * 16 + qq * 8), f32x8);
#pragma unroll
                for (int c = 0; c < 8; ++c) x[t][qq * 8 + c] = a[c];
            }
    }
}

// cooperative global -> LDS copy by the whole workgroup (n multiple of 4 floats, 16-B aligned).
// Uses the gfx950 LDS-DMA path (global_load_lds_dwordx4): every wave issues all of its 1-KiB pieces
// back to back with no VGPR round trip, so the L2 latency is paid once per phase instead of once
// per piece.  The LDS destination of one instruction is wave-uniform base + lane*16, which is exactly
// a linear copy.  Completion: the DMA counts on vmcnt; the following __syncthreads() drains it.
__device__ __forceinline__ void stage(float* lds, const float* g, int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int base = wave * 256; base < n; base += nw * 256) {
        const int idx = base + lane * 4;
        if (idx < n)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(g + idx),
                (__attribute__((address_space(3))) void*)(lds + base), 16, 0, 0);
    }
}

}  // namespace gnnmp
