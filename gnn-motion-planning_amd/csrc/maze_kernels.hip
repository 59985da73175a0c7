// maze_kernels.hip -- device-side explore stage for 2-D maze problems (SURVEY.md section 8(f) ranks 1 + 3):
// the reference's greedy best-edge expansion (eval_gnn.py:198-233) together with MazeEnv's grid collision
// checker (environment/maze_env.py:270-326), one wavefront per planning problem, so a batch of problems runs
// the whole explore stage without a host round trip per step.
//
// Decisions are the reference's, exactly:
//   * dense-matrix semantics on the per-edge scores: cell P[a][b] = score of edge (b -> a); diagonal, collided
//     rows / columns and exact zeros are invisible; the next cell is the maximum over explored rows a, first
//     maximum in (position of a in the explored list, column b) order;
//   * a free edge adds b to the tree and kills column b; a blocked edge kills cells (a, b) and (b, a);
//   * collision queries replicate numpy's float32 arithmetic on the float32 node rows bit for bit (no fma
//     contraction: every operation rounded on its own (contraction off)), count one check per in-bounds configuration query, short-circuit left
//     to right, bisect while the end cells are > 1 grid step apart and the L1 distance exceeds RRT_EPS;
//   * goal test in float64 against the float64 goal state, then one more query.
// A launch is one round (one explorer forward per problem): with a fresh tree (the reference's defaults
// batch = t_max = 500 give exactly one round) or with the tree of earlier rounds carried over
// (resample rounds, eval_gnn.py:235-247).  2-D point robot (maze2) and 3-DoF stick robot (maze3).
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.hpp"

// numpy evaluates every float32 operation on its own: no fused multiply-add anywhere in this file (hipcc's default
// is -ffp-contract=fast-honor-pragmas, and HIP's __fmul_rn / __fadd_rn are plain operators that WOULD be fused;
// __fsqrt_rn is the approximate native square root), so the rounded operations are spelled out here.
#pragma clang fp contract(off)

namespace gnnmp {

namespace {

__device__ __forceinline__ float f_add(float x, float y) { return x + y; }
__device__ __forceinline__ float f_sub(float x, float y) { return x - y; }
__device__ __forceinline__ float f_mul(float x, float y) { return x * y; }
__device__ __forceinline__ float f_div(float x, float y) { return x / y; }        // correctly rounded (hipcc default)
__device__ __forceinline__ float f_sqrt(float x) { return __builtin_sqrtf(x); }   // correctly rounded (hipcc default)

constexpr int kMazeLdsCells = 4096;      // maps up to 64 x 64 are staged into LDS as bytes
constexpr int kMazeLdsNodes = 1024;      // problems up to this many nodes keep their per-node explore state in LDS

struct MazeCtx {
    const double* map;          // [w, w] occupancy (1 = obstacle), row-major map[x][y]
    const unsigned char* occ;   // LDS copy (1 = obstacle) or nullptr for maps beyond kMazeLdsCells
    int w;
    long long checks;
    double* stack;              // LDS, 4 x 48 doubles: the bisection's explicit stack (ONE lane of the wave walks a segment at a time;
                                // as private arrays indexed at run time it lived in scratch memory, a global round trip per push / pop)
};

// all lanes of the wave: stage the problem's map into LDS
__device__ __forceinline__ void maze_ctx_init(MazeCtx& m, const double* map, int w, unsigned char* lds, int lane) {
    __shared__ double s_stack[4 * 48];
    m.stack = s_stack;
    m.map = map;
    m.w = w;
    m.checks = 0;
    m.occ = nullptr;
    if (w * w <= kMazeLdsCells) {
        for (int i = lane; i < w * w; i += 64) lds[i] = map[i] == 0.0 ? 0 : 1;
        m.occ = lds;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int maze_cell(float x, int w) {        // ((x + 1.0) * w / 2.0).astype(int), clipped at w-1
    const float t = f_div(f_mul(f_add(x, 1.0f), (float)w), 2.0f);
    int c = (int)t;
    return c > w - 1 ? w - 1 : c;
}

__device__ __forceinline__ bool maze_valid(float x, float y) { return x >= -1.0f && x <= 1.0f && y >= -1.0f && y <= 1.0f; }

__device__ __forceinline__ bool maze_state_fp(MazeCtx& m, float x, float y) {
    if (!maze_valid(x, y)) return false;
    m.checks += 1;
    const int idx = maze_cell(x, m.w) * m.w + maze_cell(y, m.w);
    return m.occ ? m.occ[idx] == 0 : m.map[idx] == 0.0;
}

// iterative form of the recursive bisection (left half first, stop at the first blocked midpoint)
__device__ __forceinline__ bool maze_segment_fp(MazeCtx& m, float ax, float ay, float bx, float by) {
    float* sx0 = reinterpret_cast<float*>(m.stack), *sy0 = sx0 + 48, *sx1 = sx0 + 96, *sy1 = sx0 + 144;
    int sp = 0;
    sx0[0] = ax; sy0[0] = ay; sx1[0] = bx; sy1[0] = by; sp = 1;
    while (sp > 0) {
        --sp;
        const float lx = sx0[sp], ly = sy0[sp], rx = sx1[sp], ry = sy1[sp];
        const int dc = abs(maze_cell(lx, m.w) - maze_cell(rx, m.w)) + abs(maze_cell(ly, m.w) - maze_cell(ry, m.w));
        const float l1 = f_add(fabsf(f_sub(lx, rx)), fabsf(f_sub(ly, ry)));
        if (dc > 1 && l1 > 0.05f) {
            const float mx = f_div(f_add(lx, rx), 2.0f), my = f_div(f_add(ly, ry), 2.0f);
            if (!maze_state_fp(m, mx, my)) return false;
            if (sp + 2 > 48) return false;                       // cannot happen: depth <= ~8 for RRT_EPS = 0.05
            sx0[sp] = mx; sy0[sp] = my; sx1[sp] = rx; sy1[sp] = ry; ++sp;      // right half second ...
            sx0[sp] = lx; sy0[sp] = ly; sx1[sp] = mx; sy1[sp] = my; ++sp;      // ... left half first
        }
    }
    return true;
}

__device__ __forceinline__ bool maze_edge_fp(MazeCtx& m, float ax, float ay, float bx, float by) {
    if (!maze_valid(ax, ay) || !maze_valid(bx, by)) return false;
    if (!maze_state_fp(m, ax, ay)) return false;
    if (!maze_state_fp(m, bx, by)) return false;
    return maze_segment_fp(m, ax, ay, bx, by);
}

// ---- 3-DoF maze (MazeEnv(dim=3)): a stick of length 0.2 at (x, y), orientation coordinate z in [-0.4, 0.4]
// (environment/maze_env.py:254-302,330-347).  The reference's numpy dtype flow is part of the behaviour: node rows are
// float32, the stick ends come out float64 (float32 coordinate / float64 limit), so the 2-D point / segment queries of
// the ends run in float64, while the interpolation along an edge runs in float32.
__device__ __forceinline__ int maze_cell64(double x, int w) {
    int c = (int)((x + 1.0) * (double)w / 2.0);
    return c > w - 1 ? w - 1 : c;
}
__device__ __forceinline__ bool maze_valid64(double x, double y) { return x >= -1.0 && x <= 1.0 && y >= -1.0 && y <= 1.0; }
__device__ __forceinline__ bool maze_point_fp64(MazeCtx& m, double x, double y) {
    if (!maze_valid64(x, y)) return false;
    m.checks += 1;
    const int idx = maze_cell64(x, m.w) * m.w + maze_cell64(y, m.w);
    return m.occ ? m.occ[idx] == 0 : m.map[idx] == 0.0;
}
__device__ bool maze_segment_fp64(MazeCtx& m, double ax, double ay, double bx, double by) {
    double* sx0 = m.stack, *sy0 = sx0 + 48, *sx1 = sx0 + 96, *sy1 = sx0 + 144;
    int sp = 1;
    sx0[0] = ax; sy0[0] = ay; sx1[0] = bx; sy1[0] = by;
    while (sp > 0) {
        --sp;
        const double lx = sx0[sp], ly = sy0[sp], rx = sx1[sp], ry = sy1[sp];
        const int dc = abs(maze_cell64(lx, m.w) - maze_cell64(rx, m.w)) + abs(maze_cell64(ly, m.w) - maze_cell64(ry, m.w));
        const double l1 = fabs(lx - rx) + fabs(ly - ry);
        if (dc > 1 && l1 > 0.05) {
            const double mx = (lx + rx) / 2.0, my = (ly + ry) / 2.0;
            if (!maze_point_fp64(m, mx, my)) return false;
            if (sp + 2 > 48) return false;
            sx0[sp] = mx; sy0[sp] = my; sx1[sp] = rx; sy1[sp] = ry; ++sp;
            sx0[sp] = lx; sy0[sp] = ly; sx1[sp] = mx; sy1[sp] = my; ++sp;
        }
    }
    return true;
}
__device__ __forceinline__ bool maze_valid3(float x, float y, float z) {
    return maze_valid(x, y) && (double)z >= -0.4 && (double)z <= 0.4;
}
// stick ends: theta = z / LIMITS[2] * pi (float64), end = center -+ (STICK_LENGTH / 2) * (cos, sin)
__device__ __forceinline__ void stick_ends(float x, float y, float z, double& ax, double& ay, double& bx, double& by) {
    const double theta = (double)z / 0.4 * 3.141592653589793;
    const double ox = 0.1 * cos(theta), oy = 0.1 * sin(theta);
    ax = (double)x - ox; ay = (double)y - oy;
    bx = (double)x + ox; by = (double)y + oy;
}
__device__ bool stick_state_fp(MazeCtx& m, float x, float y, float z) {        // _stick_in_free_space
    if (!maze_valid3(x, y, z)) return false;
    double ax, ay, bx, by;
    stick_ends(x, y, z, ax, ay, bx, by);
    if (!maze_point_fp64(m, ax, ay) || !maze_point_fp64(m, bx, by)) return false;
    return maze_segment_fp64(m, ax, ay, bx, by);
}
__device__ bool stick_edge_fp(MazeCtx& m, const float* s, const float* t) {      // _edge_fp, state.size == 3
    if (!maze_valid3(s[0], s[1], s[2]) || !maze_valid3(t[0], t[1], t[2])) return false;
    if (!stick_state_fp(m, s[0], s[1], s[2]) || !stick_state_fp(m, t[0], t[1], t[2])) return false;
    const float d0 = f_sub(t[0], s[0]), d1 = f_sub(t[1], s[1]);
    float d2 = f_sub(t[2], s[2]);
    if (fabs((double)d2) > 0.4) d2 = (float)(d2 > 0.f ? (double)d2 - 0.8 : (double)d2 + 0.8);
    // distance(): |diff| in float32, third coordinate wrapped through float64, sqrt of the float32 sum of squares
    const float a0 = fabsf(f_sub(t[0], s[0])), a1 = fabsf(f_sub(t[1], s[1]));
    const float a2r = fabsf(f_sub(t[2], s[2]));
    const double w2 = fabs((double)a2r - 0.8);
    const float a2 = (float)((double)a2r < w2 ? (double)a2r : w2);
    const float dist = f_sqrt(f_add(f_add(f_mul(a0, a0), f_mul(a1, a1)), f_mul(a2, a2)));
    const int K = (int)f_div(dist, 0.015f);
    for (int k = 1; k < K; ++k) {
        const float r = (float)((double)k * 1.0 / (double)K);
        const float cx = f_add(s[0], f_mul(r, d0)), cy = f_add(s[1], f_mul(r, d1)), cz = f_add(s[2], f_mul(r, d2));
        double ax, ay, bx, by;
        stick_ends(cx, cy, cz, ax, ay, bx, by);
        // the 2-D _edge_fp of the two ends
        if (!maze_valid64(ax, ay) || !maze_valid64(bx, by)) return false;
        if (!maze_point_fp64(m, ax, ay) || !maze_point_fp64(m, bx, by)) return false;
        if (!maze_segment_fp64(m, ax, ay, bx, by)) return false;
    }
    return true;
}

}  // namespace

// diagnostics build (-DGNNMP_MAZE_TRACE, tools/diag/maze_trace.py): cycles per phase of the greedy loop, per problem
#ifdef GNNMP_MAZE_TRACE
__device__ long long g_maze_trace[8 * 4096];
#define MZ_T0() long long mz_t = wall_clock64(); long long mz_acc[6] = {0, 0, 0, 0, 0, 0}
#define MZ_LAP(k) { const long long mz_n = wall_clock64(); mz_acc[k] += mz_n - mz_t; mz_t = mz_n; }
#else
#define MZ_T0()
#define MZ_LAP(k)
#endif

// ---- wave-wide maximum of a 64-bit key.  The greedy loop is one dependent chain per problem and every step ends in two
// or three of these: as six rounds of ds_bpermute shuffles of three or four values each they were ~0.5 us of a 3.7 us step;
// four DPP rounds inside the rows of 16 lanes plus four readlanes are ~0.1 us.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long k) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)k, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(k >> 32), CTRL, 0xf, 0xf, true);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
    unsigned long long o;
    o = dpp_u64<0xB1>(k); k = o > k ? o : k;                   // quad_perm [1, 0, 3, 2]
    o = dpp_u64<0x4E>(k); k = o > k ? o : k;                   // quad_perm [2, 3, 0, 1]
    o = dpp_u64<0x141>(k); k = o > k ? o : k;                  // row_half_mirror
    o = dpp_u64<0x140>(k); k = o > k ? o : k;                  // row_mirror: every lane of a row holds the row's maximum
    unsigned long long r = 0;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, 16 * row);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), 16 * row);
        const unsigned long long c = ((unsigned long long)hi << 32) | lo;
        r = c > r ? c : r;
    }
    return r;
}
// order-preserving map of a float onto an unsigned (never 0 for a finite value or -inf: 0 stands for "no cell")
__device__ __forceinline__ unsigned maze_ord(float x) {
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// one workgroup of four waves per problem: all four group the cells by row, then wave 0 walks the greedy loop alone.
// DIM = 2: point robot; DIM = 3: stick robot (maze3).  LDS: the problem's per-node state fits the LDS arrays -- a template
// parameter, not a run-time choice of pointers: pointers that may address LDS or global memory make every access a FLAT
// instruction, which counts against both the vector-memory and the LDS counter, so the loop could not keep a row request
// in flight across an LDS read.
struct MazeShared { int *in_ptr, *pos, *explored, *rb_src, *rb_eid; unsigned* rb_key; float* v; unsigned char* occ; };
template <int DIM, bool LDS>
__device__ __forceinline__ void maze_explore_body(const MazeParams& p, const MazeShared& sh) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int n0 = p.node_ptr[b], N = p.node_ptr[b + 1] - n0;
    const int e0 = p.edge_ptr[b], E = p.edge_ptr[b + 1] - e0;
    const int F = p.n_free[b];
    const long long* src = p.edge_index + e0;
    const long long* dst = p.edge_index + (size_t)p.total_edges + e0;
    const float* sc = p.scores + e0;
    MZ_T0();
    // Per-node state of problems up to kMazeLdsNodes nodes lives in LDS: every step of the greedy loop reads the cached row
    // maxima of the whole frontier, the explored list, row ranges, explored positions and two node rows -- from the
    // workspace in global memory each of them was a dependent round trip of ~1 us inside a loop of ~2 000 steps.
    int* const s_in_ptr = sh.in_ptr; int* const s_pos = sh.pos; int* const s_explored = sh.explored;
    int* const s_rb_src = sh.rb_src; int* const s_rb_eid = sh.rb_eid; unsigned* const s_rb_key = sh.rb_key;
    float* const s_v = sh.v; unsigned char* const occ_lds = sh.occ;
    constexpr bool in_lds = LDS;
    int* in_ptr = in_lds ? s_in_ptr : p.in_ptr + n0 + b;              // [N + 1] per problem
    int* cnt = in_lds ? s_rb_src : p.cnt + n0;                         // (build only: the row caches are set up afterwards)
    // The matrix cells the loop can ever pick -- off the diagonal, non-zero score, free row and column (eval_gnn.py:
    // 189-196) -- grouped by row (= edge target) as (column | dead bit, score) records: a row scan is ONE coalesced read
    // (edge ids + separate alive / source / score arrays made every scan two dependent round trips).
    int2* rec = p.in_rec + e0;
    constexpr int kDead = (int)0x80000000;
    int* pos = in_lds ? s_pos : p.pos + n0;                            // position in the explored list or -1
    int* explored = in_lds ? s_explored : p.explored + n0;            // (copied to p.explored at the end)
    int* prev = p.prev + n0;
    int* ee = p.explored_edges + 2 * ((size_t)2 * e0 + b);      // [2E + 1] (a, b) pairs per problem
    const float* v = p.v + (size_t)n0 * DIM;
    if (in_lds) {
        for (int i = tid; i < N * DIM; i += 256) s_v[i] = v[i];
        v = s_v;
    }

    // ---- visible cells by row (~55 k edges per problem at the published setting: 1.5 ms for one wave, half of the mean
    // problem's time, 0.4 ms for four)
    for (int i = tid; i < N; i += 256) { cnt[i] = 0; pos[i] = -1; }
    __syncthreads();
    auto visible = [&](int e, int& s, int& t) {
        s = (int)src[e]; t = (int)dst[e];
        return s != t && sc[e] != 0.0f && s < F && t < F;                  // diagonal, zeros, collided rows / columns
    };
    for (int e = tid; e < E; e += 256) {
        int s, t;
        if (visible(e, s, t)) atomicAdd(&cnt[t], 1);
    }
    __syncthreads();
    if (tid < 64) {
        int run = 0;                                                       // wave-wide exclusive scan, 64 rows a time
        for (int base = 0; base < N; base += 64) {
            const int i = base + lane;
            const int c = i < N ? cnt[i] : 0;
            int incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            if (i < N) { in_ptr[i] = run + incl - c; cnt[i] = 0; }
            run += __shfl(incl, 63, 64);
        }
        if (lane == 0) in_ptr[N] = run;
    }
    __syncthreads();
    for (int e = tid; e < E; e += 256) {
        int s, t;
        if (visible(e, s, t)) rec[in_ptr[t] + atomicAdd(&cnt[t], 1)] = make_int2(s, __float_as_int(sc[e]));
    }
    __syncthreads();
    if (tid >= 64) return;                                                 // (no workgroup barrier below this line)

    MazeCtx m;
    maze_ctx_init(m, p.maps + (size_t)b * p.w * p.w, p.w, occ_lds, lane);
    const double* goal = p.goal_states + (size_t)DIM * b;
    int n_expl = 1, n_pairs = 1, success = 0, path_len = 0;
    if (p.n_explored0 == nullptr) {                                        // fresh tree (eval_gnn.py:183-186)
        if (lane == 0) { explored[0] = 0; pos[0] = 0; prev[0] = 0; ee[0] = 0; ee[1] = 0; }
    } else {
        // tree carried over from earlier rounds (eval_gnn.py:235-247): explored nodes keep their indices (new free
        // samples are appended behind the old ones).  The masks are rebuilt on the fresh scores as the reference does:
        // columns of explored nodes (through pos[]), and the legacy-index line eval_gnn.py:202, which pairs the FLATTENED
        // pair list's first half with its second half (cell (flat[i], flat[M + i])), not the recorded (a, b) pairs
        n_expl = p.n_explored0[b];
        n_pairs = 0;                                                       // only this round's pairs are written out
        for (int i = lane; i < n_expl; i += 64) {
            const int a = p.explored0[n0 + i];
            explored[i] = a; pos[a] = i; prev[a] = p.prev0[n0 + a];
        }
        const int M = p.n_pairs0[b];
        const int* flat = p.pairs0 + 2 * (size_t)p.pair_ptr0[b];
        for (int i = lane; i < M; i += 64) {
            const int r = flat[i], c = flat[M + i];
            if (r >= N) continue;
            for (int q = in_ptr[r]; q < in_ptr[r + 1]; ++q)
                if ((rec[q].x & ~kDead) == c) rec[q].x = c | kDead;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();

    // Cached best live cell of every explored row i: (ordered value or 0 = none, column, slot in rec), first maximum in
    // column order.  A row's best only changes when that very cell dies -- its column gets explored, or the edge is found
    // blocked -- so a step rescans one to three rows instead of the whole frontier.
    unsigned* rb_key = in_lds ? s_rb_key : reinterpret_cast<unsigned*>(p.rb_val) + n0;
    int* rb_src = in_lds ? s_rb_src : p.rb_src + n0;
    int* rb_eid = in_lds ? s_rb_eid : p.rb_eid + n0;
    // `pre` (when have_pre): the first 64 records of the row, loaded earlier (and patched with this step's kills)
    auto rescan = [&](int i, bool have_pre, int2 pre) {
        const int a = explored[i];
        const int q0 = in_ptr[a], q1 = in_ptr[a + 1];
        unsigned long long best = 0;
        int bq = -1;
        for (int base = q0; base < q1; base += 64) {
            const int q = base + lane;
            int2 r = make_int2(kDead, 0);
            if (have_pre && base == q0) r = pre;
            else if (q < q1) r = rec[q];
            if (q < q1 && r.x >= 0 && pos[r.x] < 0) {                      // live cell, column not explored yet
                const unsigned long long key = ((unsigned long long)maze_ord(__int_as_float(r.y)) << 32) | (0xffffffffu - (unsigned)r.x);
                if (key > best) { best = key; bq = q; }                    // value desc, column asc
            }
        }
        const unsigned long long w = wave_max_u64(best);
        int be = -1, bb = 0x7fffffff;
        if (w) {
            const int wl = __builtin_ctzll(__ballot(best == w));
            be = __builtin_amdgcn_readlane(bq, wl);
            bb = (int)(0xffffffffu - (unsigned)w);
        }
        if (lane == 0) { rb_key[i] = (unsigned)(w >> 32); rb_src[i] = bb; rb_eid[i] = be; }
    };
    auto sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    };
    for (int i = 0; i < n_expl; ++i) rescan(i, false, make_int2(kDead, 0));
    sync();
    MZ_LAP(0);

    while (true) {
        // ---- argmax over the cached row maxima: key (value desc, row position asc); the column is the row's own
        unsigned long long best = 0;
        for (int i = lane; i < n_expl; i += 64) {
            const unsigned k = rb_key[i];
            const unsigned long long key = ((unsigned long long)k << 32) | (0xffffffffu - (unsigned)i);
            if (k && key > best) best = key;
        }
        const unsigned long long w = wave_max_u64(best);
        if (!w) break;                                                     // nothing left on the frontier
        const int bp = (int)(0xffffffffu - (unsigned)w);
        const int a = explored[bp], nb = rb_src[bp], be = rb_eid[bp];
        MZ_LAP(1);
        // the rows this step rescans (row a when the edge is blocked, the new row nb when it is free) and the row that
        // holds the mirrored cell (nb, a) are requested BEFORE the collision check: lane 0 needs ~1 us for it
        const int qa0 = in_ptr[a], qa1 = in_ptr[a + 1], qn0 = in_ptr[nb], qn1 = in_ptr[nb + 1];
        // (unconditional loads from clamped slots: behind a lane mask the compiler waits for them at the end of the branch)
        int2 ra = rec[min(qa0 + lane, qa1 - 1)];                           // row a holds the cell `be`: not empty
        int2 rn = rec[max(min(qn0 + lane, qn1 - 1), 0)];
        int free_edge = 0, goal_hit = 0;
        if (lane == 0) {
            ee[2 * n_pairs] = a; ee[2 * n_pairs + 1] = nb;
            ee[2 * n_pairs + 2] = nb; ee[2 * n_pairs + 3] = a;
            if constexpr (DIM == 2) free_edge = maze_edge_fp(m, v[2 * a], v[2 * a + 1], v[2 * nb], v[2 * nb + 1]) ? 1 : 0;
            else free_edge = stick_edge_fp(m, v + 3 * a, v + 3 * nb) ? 1 : 0;
            if (free_edge) {
                explored[n_expl] = nb; pos[nb] = n_expl; prev[nb] = a;
                if constexpr (DIM == 2) {
                    const double dx = fabs(goal[0] - (double)v[2 * nb]), dy = fabs(goal[1] - (double)v[2 * nb + 1]);
                    const double d = sqrt(dx * dx + dy * dy);              // no contraction: see the pragma above
                    if (d < 0.05) goal_hit = maze_state_fp(m, v[2 * nb], v[2 * nb + 1]) ? 1 : 0;
                } else {                                                   // distance() wraps the orientation coordinate
                    const double dx = fabs(goal[0] - (double)v[3 * nb]), dy = fabs(goal[1] - (double)v[3 * nb + 1]);
                    double dz = fabs(goal[2] - (double)v[3 * nb + 2]);
                    const double wz = fabs(dz - 0.8);
                    dz = dz < wz ? dz : wz;
                    const double d = sqrt((dx * dx + dy * dy) + dz * dz);
                    if (d < 0.05) goal_hit = stick_state_fp(m, v[3 * nb], v[3 * nb + 1], v[3 * nb + 2]) ? 1 : 0;
                }
            } else {
                rec[be].x = nb | kDead;                                    // cell (a, nb)
            }
        }
        n_pairs += 2;
        free_edge = __builtin_amdgcn_readfirstlane(free_edge);
        goal_hit = __builtin_amdgcn_readfirstlane(goal_hit);
        MZ_LAP(2);
        if (qa0 + lane >= qa1) ra.x = kDead;
        if (qn0 + lane >= qn1) rn.x = kDead;
        if (!free_edge) {
            if (qa0 + lane == be) ra.x = nb | kDead;                       // ... in the copy of row a as well
            // cell (nb, a): looked up among nb's cells by the whole wave (one lane walking the ~50 dependent loads of that
            // list was most of a blocked step, and most steps are blocked edges)
            if (qn0 + lane < qn1 && (rn.x & ~kDead) == a) rec[qn0 + lane].x = a | kDead;
            for (int q = qn0 + 64 + lane; q < qn1; q += 64)
                if ((rec[q].x & ~kDead) == a) rec[q].x = a | kDead;
        }
        sync();
        MZ_LAP(3);
        if (free_edge) {
            ++n_expl;
            if (goal_hit) { success = 1; break; }
            // column nb is gone: every row whose cached best sat in it looks again (row a is one of them); the new row is scanned
            for (int base = 0; base < n_expl - 1; base += 64) {
                const int i = base + lane;
                unsigned long long stale = __ballot(i < n_expl - 1 && rb_key[i] != 0 && rb_src[i] == nb);
                while (stale) {
                    const int row = base + __builtin_ctzll(stale);
                    rescan(row, row == bp, ra);
                    stale &= stale - 1;
                }
            }
            rescan(n_expl - 1, true, rn);
        } else {
            rescan(bp, true, ra);                                          // cell (a, nb) died
        }
        sync();
        MZ_LAP(4);
    }
#ifdef GNNMP_MAZE_TRACE
    if (lane == 0 && b < 4096) {
        for (int k = 0; k < 5; ++k) g_maze_trace[8 * b + k] = mz_acc[k];
        g_maze_trace[8 * b + 5] = n_pairs / 2;
        g_maze_trace[8 * b + 6] = n_expl;
        g_maze_trace[8 * b + 7] = E;
    }
#endif
    if (lane == 0) {
        if (success) {                                                     // back-track prev[] to the start node
            int node = explored[n_expl - 1], len = 0;
            int* path = p.path + n0;
            while (true) { path[len++] = node; if (node == 0) break; node = prev[node]; }
            for (int i = 0; i < len / 2; ++i) { const int t = path[i]; path[i] = path[len - 1 - i]; path[len - 1 - i] = t; }
            path_len = len;
        }
        if (p.prev_out) for (int i = 0; i < n_expl; ++i) p.prev_out[n0 + explored[i]] = prev[explored[i]];
        p.success[b] = success;
        p.n_explored[b] = n_expl;
        p.n_pairs[b] = n_pairs;
        p.path_len[b] = path_len;
        p.checks[b] = m.checks;
    }
    if (in_lds)
        for (int i = lane; i < n_expl; i += 64) p.explored[n0 + i] = explored[i];
}

template <int DIM>
__global__ __launch_bounds__(256) void maze_explore_kernel(MazeParams p) {
    __shared__ int s_in_ptr[kMazeLdsNodes + 1], s_pos[kMazeLdsNodes], s_explored[kMazeLdsNodes];
    __shared__ int s_rb_src[kMazeLdsNodes], s_rb_eid[kMazeLdsNodes];
    __shared__ unsigned s_rb_key[kMazeLdsNodes];
    __shared__ float s_v[kMazeLdsNodes * DIM];
    __shared__ unsigned char occ_lds[kMazeLdsCells];
    const MazeShared sh{s_in_ptr, s_pos, s_explored, s_rb_src, s_rb_eid, s_rb_key, s_v, occ_lds};
    const int N = p.node_ptr[blockIdx.x + 1] - p.node_ptr[blockIdx.x];
    if (N <= kMazeLdsNodes) maze_explore_body<DIM, true>(p, sh);
    else maze_explore_body<DIM, false>(p, sh);
}

// ---------------------------------------------------------------------------------------------------------
// Steering of the smoothing stage: proposed_path_smootherv2 (smoother.py:194-216).  Every interior waypoint
// moves at most RRT_EPS per round towards the network's proposal and keeps the move only if both adjacent
// edges stay free (left neighbour already updated, right one not); at most K = ceil(max distance / RRT_EPS)
// rounds, early exit once the accepted waypoints sit on their targets.  The waypoints are numpy float32 arrays
// in the reference, so every operation below is the explicitly rounded float32 one (no fma contraction):
//   norm(x)     = sqrt(x0*x0 + x1*x1)       (np.linalg.norm: sqrt(dot) / sqrt(add.reduce(x*x)))
//   interpolate = a + (b - a) * (RRT_EPS / dist), python-float constants weak-cast to float32 (NEP 50)
// One wavefront per problem; the walk itself is sequential (lane 0), like the reference's.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float maze_norm2(float dx, float dy) {
    return f_sqrt(f_add(f_mul(dx, dx), f_mul(dy, dy)));
}

__global__ __launch_bounds__(64) void maze_steer_kernel(MazeSteerParams p) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int p0 = p.path_ptr[b], P = p.path_ptr[b + 1] - p0;
    const float* oldp = p.old_path + (size_t)p0 * 2;
    const float* newp = p.new_path + (size_t)p0 * 2;
    float* cur = p.out_path + (size_t)p0 * 2;
    float* nxt = p.tmp + (size_t)p0 * 2;
    // K = int(ceil((norm(old - new, axis=-1) / RRT_EPS).max()))
    float mx = 0.0f;
    for (int i = lane; i < P; i += 64) {
        const float n = maze_norm2(f_sub(oldp[2 * i], newp[2 * i]), f_sub(oldp[2 * i + 1], newp[2 * i + 1]));
        mx = fmaxf(mx, f_div(n, 0.05f));
        cur[2 * i] = oldp[2 * i];
        cur[2 * i + 1] = oldp[2 * i + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const int K = (int)ceilf(mx);
    __shared__ unsigned char occ_lds[kMazeLdsCells];
    MazeCtx m;
    maze_ctx_init(m, p.maps + (size_t)b * p.w * p.w, p.w, occ_lds, lane);      // includes the fence + barrier
    if (lane != 0 || P < 3) return;
    for (int r = 0; r < K; ++r) {
        float diff = 0.0f;
        nxt[0] = cur[0]; nxt[1] = cur[1];
        nxt[2 * (P - 1)] = cur[2 * (P - 1)]; nxt[2 * (P - 1) + 1] = cur[2 * (P - 1) + 1];
        for (int i = 1; i < P - 1; ++i) {
            const float ox = cur[2 * i], oy = cur[2 * i + 1], tx = newp[2 * i], ty = newp[2 * i + 1];
            const float dist = maze_norm2(f_sub(ox, tx), f_sub(oy, ty));
            float cx = tx, cy = ty;
            if (!(dist < 0.05f)) {
                const float ratio = f_div(0.05f, dist);
                cx = f_add(ox, f_mul(f_sub(tx, ox), ratio));
                cy = f_add(oy, f_mul(f_sub(ty, oy), ratio));
            }
            // nxt[i-1] is this round's value, the right neighbour still last round's
            const bool ok = maze_edge_fp(m, nxt[2 * (i - 1)], nxt[2 * (i - 1) + 1], cx, cy) &&
                            maze_edge_fp(m, cur[2 * (i + 1)], cur[2 * (i + 1) + 1], cx, cy);
            if (ok) {
                nxt[2 * i] = cx; nxt[2 * i + 1] = cy;
                diff = f_add(diff, maze_norm2(f_sub(cx, tx), f_sub(cy, ty)));
            } else {
                nxt[2 * i] = ox; nxt[2 * i + 1] = oy;
            }
        }
        for (int i = 0; i < 2 * P; ++i) cur[i] = nxt[i];
        if (diff < 1e-5f) break;
    }
    p.checks[b] += m.checks;
}

// ---------------------------------------------------------------------------------------------------------------------
// Rejection sampling on the device.  The reference samples a problem's nodes one by one (eval_gnn.py:180-184 ->
// MazeEnv.sample_n_points, maze_env.py): draw a uniform point of [-1, 1)^2 from the GLOBAL numpy generator, keep it if its grid
// cell is free, until n free points exist; the rejected draws of the same loop become the problem's collided samples (the first n
// of them), and the NEXT problem continues in the same random stream -- so where problem b + 1 starts depends on how many
// draws problem b needed.  The draws themselves stay with the host (numpy's Mersenne twister: cheap, and it keeps the recorded
// known answers); what moves here is everything after them: the classification of every draw (float64 arithmetic of
// ((x + 1.0) * w / 2.0).astype(int), clipped at w - 1, one map lookup), the position of the n-th free draw, the compaction into
// the float32 node rows [start, goal, free ..., rejected ...] the graph builder and the explorer read, the per-problem counts.
// ONE workgroup walks the problems in stream order (the dependency is serial), 1024 draws per step: a block scan of the free
// flags gives every draw its rank; ~1300 draws per problem at the published setting = two steps, ~3 us.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void maze_sample_kernel(MazeSampleParams p) {
    __shared__ unsigned char occ[kMazeLdsCells];
    __shared__ int wsum[16];
    __shared__ int s_tstar, s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long cur0 = *p.cursor;
    long long cur = cur0;
    int node0 = 0;
    const bool in_lds = p.w * p.w <= kMazeLdsCells;
    if (tid == 0) { p.node_ptr[0] = 0; s_fail = 0; }
    for (int b = 0; b < p.B; ++b) {
        const double* map = p.maps + (size_t)b * p.w * p.w;
        __syncthreads();                                        // the previous problem's map is no longer in use
        if (in_lds)
            for (int i = tid; i < p.w * p.w; i += 1024) occ[i] = map[i] == 0.0 ? 0 : 1;
        if (tid == 0) s_tstar = -1;
        __syncthreads();
        int free_before = 0, rej_before = 0, rej_total = 0, used = 0;
        for (long long off = 0;; off += 1024) {
            const long long idx = cur + off + tid;
            const bool in = idx < p.M;
            double x = 0.0, y = 0.0;
            if (in) { x = p.attempts[2 * idx]; y = p.attempts[2 * idx + 1]; }
            bool isfree = false;
            if (in) {
                const int c = maze_cell64(x, p.w) * p.w + maze_cell64(y, p.w);
                isfree = in_lds ? occ[c] == 0 : map[c] == 0.0;
            }
            // inclusive block scan of the free flags: ballot + popcount inside a wave, the sixteen wave totals through LDS
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(isfree);
            const int incl_w = __builtin_popcountll(bal & (~0ull >> (63 - lane)));
            if (lane == 0) wsum[wave] = __builtin_popcountll(bal);
            __syncthreads();
            int before = 0, F = 0;
#pragma unroll
            for (int w2 = 0; w2 < 16; ++w2) { const int c = wsum[w2]; before += w2 < wave ? c : 0; F += c; }
            const int incl = before + incl_w;                   // free draws among this step's draws 0 .. tid
            if (isfree && free_before + incl == p.n) s_tstar = tid;      // the n-th free draw of the problem
            __syncthreads();
            const int tstar = s_tstar;
            const bool done = tstar >= 0;
            const int limit = done ? tstar : 1023;
            if (in && tid <= limit) {
                if (isfree) {
                    const size_t row = (size_t)node0 + 2 + free_before + incl - 1;
                    p.v[2 * row] = (float)x; p.v[2 * row + 1] = (float)y;
                } else {
                    const int rr = rej_before + (tid + 1 - incl) - 1;        // rank among the problem's rejected draws
                    if (rr < p.n) {
                        const size_t row = (size_t)node0 + 2 + p.n + rr;
                        p.v[2 * row] = (float)x; p.v[2 * row + 1] = (float)y;
                    }
                }
            }
            if (done) {
                // every thread derives the same totals: rejected draws up to and including position tstar
                const int free_upto = p.n - free_before;                     // free draws of this step up to tstar
                rej_total = rej_before + (tstar + 1 - free_upto);
                used = (int)off + tstar + 1;
                break;
            }
            if (cur + off + 1024 >= p.M) {                      // the stream ran out (workgroup-uniform)
                if (tid == 0) { *p.ok = 0; *p.cursor = cur0; p.node_ptr[b + 1] = -1; }      // nothing is consumed
                return;
            }
            free_before += F;
            rej_before += 1024 - F;
            __syncthreads();                                    // wsum / s_tstar are rewritten by the next step
        }
        const int Nb = 2 + p.n + (rej_total < p.n ? rej_total : p.n);
        if (tid == 0) {
            const double* is = p.init_states + 2 * (size_t)b;
            const double* gs = p.goal_states + 2 * (size_t)b;
            p.v[2 * (size_t)node0] = (float)is[0]; p.v[2 * (size_t)node0 + 1] = (float)is[1];
            p.v[2 * (size_t)node0 + 2] = (float)gs[0]; p.v[2 * (size_t)node0 + 3] = (float)gs[1];
            p.node_ptr[b + 1] = node0 + Nb;
            p.used[b] = used;
        }
        cur += used;
        node0 += Nb;
    }
    if (tid == 0) { *p.cursor = cur; *p.ok = 1; }
}

hipError_t launch_maze_sample(const MazeSampleParams& p, hipStream_t st) {
    if (p.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(maze_sample_kernel, dim3(1), dim3(1024), 0, st, p);
    return hipGetLastError();
}

hipError_t launch_maze_steer(const MazeSteerParams& p, hipStream_t st) {
    if (p.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(maze_steer_kernel, dim3(p.B), dim3(64), 0, st, p);
    return hipGetLastError();
}

hipError_t launch_maze_explore(const MazeParams& p, hipStream_t st) {
    if (p.B <= 0) return hipSuccess;
    if (p.dim == 3) hipLaunchKernelGGL(maze_explore_kernel<3>, dim3(p.B), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(maze_explore_kernel<2>, dim3(p.B), dim3(256), 0, st, p);
    return hipGetLastError();
}

#ifdef GNNMP_MAZE_TRACE
}  // namespace gnnmp
// copies the counters of the last explore launch (8 per problem: cycles of build / argmax / check / kill / rescan at 100 MHz,
// steps, explored nodes, edges) to `dst` [8 * n]
extern "C" int gnnmp_debug_maze_trace(long long* dst, int n) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(gnnmp::g_maze_trace), sizeof(long long) * 8 * (size_t)(n > 4096 ? 4096 : n)) == hipSuccess ? 0 : -1;
}
namespace gnnmp {
#endif

}  // namespace gnnmp
