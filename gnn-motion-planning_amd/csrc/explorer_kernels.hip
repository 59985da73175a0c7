// explorer_kernels.hip -- HIP kernels (gfx950) of the GNN path-explorer forward pass,
// EncoderProcessDecoder.forward of the reference (model.py:115-150), restructured for MI355X:
//
//   prep_*        caller edge list -> CSR-by-destination in a per-graph padded index space
//   (prep stage)  goal node = argmin_i |v_i - goal|                         (model.py:132)
//   obs_kernel    obstacle codes, their 3+3 FFN stages and the K/V projections of all six
//                 attention blocks, written as ready-made MFMA A operands   (model.py:126-130, obstacle rows)
//   pre_kernel    per node / per edge: encoders + 3 obstacle cross-attention blocks, all in
//                 registers, + the loop-invariant halves of the first message / encoder /
//                 decoder / policy layers                                   (model.py:119-130)
//   mp_fused      one loop iteration per launch: message second layer + max over incoming edges (LDS float
//                 atomics), lin_1, encoder and the next iteration's node-level first-layer terms
//                                                                            (model.py:33,36,38-41,141,143)
//   policy        per-edge 3-layer head, scattered to caller order / dense  (model.py:145-149)
//
// The algebra follows SURVEY.md Appendix E: first layers acting on concatenations are split into
// per-operand matrices so nothing of shape [E, 5d] / [E, 3d] / [E, O+1, d] is ever materialised.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <math.h>
#include "chain.hpp"
#include "layout.hpp"
#include "kernels.hpp"

namespace gnnmp {

// floats of storage per 32-edge x 32-feature tile of the per-edge intermediates (bf16 in the bf16 mode)
#define kETile (P == 1 ? 512 : 1024)

// =====================================================================================================
// prep: padded index spaces + CSR by destination
// =====================================================================================================
__device__ __forceinline__ int round_up(int x, int m) { return (x + m - 1) / m * m; }

__device__ __forceinline__ int find_graph(const int* __restrict__ ptr, int G, int x) {
    int lo = 0, hi = G;            // largest g with ptr[g] <= x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// =====================================================================================================
// prep stage: CSR-by-destination build in a per-graph padded index space.  Graphs are independent and their caller
// columns contiguous, so a graph is built by one 1024-thread workgroup (prep_small_kernel: in-degree histogram + arrival
// rank with LDS atomics, block scan, scatter of the int4 records, pad records, tile -> graph maps, goal node) or, when it
// has more than ~16 k edges, by up to 16 workgroups that split its columns (prep_hist_kernel + prep_scatter_kernel).
// Graphs beyond kPrepCap padded nodes keep their counters in global memory (same code through flat pointers).
// =====================================================================================================
// up to this many 32-node tiles (the launch's PADDED tile count: 39 per 1000-node graph) mp_fused runs one tile per workgroup of
// 8 / 4 waves.  Re-measured in round 6 against the resident eight-wave / four-tiles-per-workgroup forms, which have a floor of
// 0.13 ms (d = 32) / 0.6 ms (d = 64 fp32) / 0.19 ms (d = 64 bf16) per five launches however few tiles there are
// (profiles/r06_small_batch_crossover.txt: 1000-node graphs, crossovers at 16-20 graphs at d = 32 and 24-32 at d = 64; the old
// bounds of 384 / 600 sent 12-16 graphs the slow way: kuka7 fp32, 16 problems, 1.14 -> 0.86 ms per forward)
constexpr int kCoopMaxTiles32 = 680, kCoopMaxTiles64 = 1100;
constexpr int kPrepCap = 8192;
constexpr int kPrepWindowEdges = 65536;   // big-graph prep: from this many edges per graph the CSR records are scattered in windows of target rows
constexpr int kGoalSplitNodes = 4096;     // big-graph prep: from this many nodes the goal arg-min is spread over the graph's part workgroups
constexpr int kPrepEdgesPerPart = 8192;  // prep stage: one workgroup per this many edges of a graph (at least one, at most kPrepMaxParts)

// inclusive prefix sum over the 1024 threads of a workgroup: wave-level shuffles, then the 16 wave totals through LDS
// (three barriers instead of the twenty of a Hillis-Steele scan in LDS); `wsum` holds 16 ints.  Returns the inclusive
// prefix of `v`; `total` = sum over the workgroup.
__device__ __forceinline__ int block_scan_1024(int v, int* wsum, int& total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    __syncthreads();                                   // wsum of the previous call has been read
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int t = wsum[w];
        before += w < wave ? t : 0;
        all += t;
    }
    total = all;
    return x + before;
}

// caller prefix arrays as the prep stage reads them: for ONE graph given by its totals (single_out) the entries are kernel
// arguments -- the arrays in the workspace are written for the later stages, and nothing in this launch waits for them
__device__ __forceinline__ int prep_np(const PrepParams& q, int i) { return q.single_out ? (i > 0 ? q.single_n : 0) : q.node_ptr[i]; }
__device__ __forceinline__ int prep_ep(const PrepParams& q, int i) { return q.single_out ? (i > 0 ? q.single_e : 0) : q.edge_ptr[i]; }

// Workgroup `part` of `parts` builds the CSR rows of the target nodes [lo, hi) of graph g (a slice of its padded node
// range): it walks ALL edge columns of the graph, counts the ones whose target lies below its slice (that count is where
// its slice starts in the slot space -- no communication between the parts) and ranks / scatters the ones inside.
// a caller node id as the prep stage uses it: ids outside [0, N_g) (the reference's tensor indexing would raise) are replaced by
// node 0 -- every later kernel then stays inside the graph's rows -- and reported through gnnmp_explorer_status (gstat)
__device__ __forceinline__ int prep_id(int x, int Ng, int& bad) {
    const bool b = (unsigned)x >= (unsigned)Ng;
    bad |= b ? 1 : 0;
    return b ? 0 : x;
}

__device__ __forceinline__ void prep_graph_body(const PrepParams& q, int g, int part, int parts, int n0, int Np, int e0, int e1,
                                                int* prep_lds) {
    __shared__ int carry;                              // prep_lds: cnt[kPrepCap], rb[kPrepCap], scan[1024]
    __shared__ int below_w[16];
    const int tid = threadIdx.x;
    const int c0 = prep_ep(q, g), Eg = prep_ep(q, g + 1) - c0;            // caller columns of this graph
    const int span = ((Np + parts - 1) / parts + 31) & ~31;
    const int lo = min(part * span, Np), hi = min(lo + span, Np), Nown = hi - lo;
    const bool in_lds = Nown <= kPrepCap;
    int* cnt = in_lds ? prep_lds : q.deg + n0 + lo;
    int* rb = in_lds ? prep_lds + kPrepCap : q.row_beg + n0 + lo;
    int* scan = prep_lds + 2 * kPrepCap;
    const long long* srcs = q.edge_index + c0;
    const long long* dsts = q.edge_index + (size_t)q.E + c0;
    for (int i = tid; i < Nown; i += 1024) cnt[i] = 0;
    __syncthreads();
    // arrival order inside the destination's segment (any order is fine: max-aggregation is order-free);
    // four independent columns per thread and trip so their loads are in flight together
    int below = 0;
    constexpr int U = 16;                               // independent columns per thread and trip: the loop is latency-bound
    // graphs of up to U * 1024 columns built by one workgroup: every thread keeps its columns (source, target, arrival
    // rank) in registers between the ranking and the scatter -- edge_index is read once and the rank never goes to memory
    // (with several slices per graph every workgroup reads all columns and keeps the ones whose target lies in its slice)
    const bool in_regs = Eg <= U * 1024;
    // node ids outside [0, N_g): see prep_id
    const int Ng = prep_np(q, g + 1) - prep_np(q, g);
    int bad_id = 0;
    int sv_r[U], tv_r[U], rk_r[U];
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = tid + u * 1024;
            const bool ok = cc < Eg;
            tv_r[u] = ok ? prep_id((int)dsts[cc], Ng, bad_id) : -1;
            sv_r[u] = ok ? prep_id((int)srcs[cc], Ng, bad_id) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tv_r[u];
            below += (t >= 0 && t < lo) ? 1 : 0;
            const bool mine = t >= lo && t < hi;
            rk_r[u] = mine ? atomicAdd(&cnt[t - lo], 1) : 0;
            if (!mine) tv_r[u] = -1;
        }
    } else {
        for (int c = tid; c < Eg; c += U * 1024) {
            int d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = (c + u * 1024 < Eg) ? prep_id((int)dsts[c + u * 1024], Ng, bad_id) : 0x7fffffff;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (d[u] < lo) ++below;
                else if (d[u] < hi) q.cursor[c0 + c + u * 1024] = atomicAdd(&cnt[d[u] - lo], 1);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) below += __shfl_down(below, off);
    if ((tid & 63) == 0) below_w[tid >> 6] = below;
    __syncthreads();
    if (tid == 0) {
        int sum = e0;
#pragma unroll
        for (int w = 0; w < 16; ++w) sum += below_w[w];
        carry = sum;
    }
    __syncthreads();
    int carry_run = carry;                             // every thread carries the running offset itself
    for (int base = 0; base < Nown; base += 1024) {    // exclusive scan -> absolute first slot of every node
        const int i = base + tid;
        const int d = (i < Nown) ? cnt[i] : 0;
        int chunk_total;
        const int incl = block_scan_1024(d, scan, chunk_total);
        if (i < Nown) {
            const int r = carry_run + incl - d;
            rb[i] = r;
            if (in_lds) { q.row_beg[n0 + lo + i] = r; q.deg[n0 + lo + i] = d; }
        }
        carry_run += chunk_total;
    }
    __syncthreads();                                   // rb is complete
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (tv_r[u] >= 0)                                            // {source, target, caller column}
                q.csr[rb[tv_r[u] - lo] + rk_r[u]] = make_int4(n0 + sv_r[u], n0 + tv_r[u], c0 + tid + u * 1024, 0);
    } else {
        for (int c = tid; c < Eg; c += U * 1024) {
            int sv[U], tv[U], rk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cc = c + u * 1024;
                tv[u] = cc < Eg ? prep_id((int)dsts[cc], Ng, bad_id) : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cc = c + u * 1024;
                const bool mine = tv[u] >= lo && tv[u] < hi;
                sv[u] = mine ? prep_id((int)srcs[cc], Ng, bad_id) : 0;
                rk[u] = mine ? q.cursor[c0 + cc] : 0;
                if (!mine) tv[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (tv[u] >= 0)                                          // {source, target, caller column}
                    q.csr[rb[tv[u] - lo] + rk[u]] = make_int4(n0 + sv[u], n0 + tv[u], c0 + c + u * 1024, 0);
        }
    }
    // pad slots and the per-tile maps of the graph, shared out over its parts
    if (part == 0)
        for (int b = (n0 >> 8) + tid; b < ((n0 + Np) >> 8); b += 1024) q.blk_span[b] = make_int2(n0 >> 7, (n0 + Np) >> 7);
    for (int sl = e0 + Eg + part * 1024 + tid; sl < e1; sl += parts * 1024) q.csr[sl] = make_int4(-1, -1, -1, -1);
    for (int t = (n0 + lo) / 32 + tid; t < (n0 + hi) / 32; t += 1024) q.ntile_graph[t] = g;
    for (int t = e0 / 32 + part * 1024 + tid; t < e1 / 32; t += parts * 1024) {
        q.etile_graph[t] = g;
        q.tile_meta[t] = t * 32 < e0 + Eg ? 4 : 0;     // 4 = tile holds at least one edge (the edge pre kernel skips pure padding)
    }
    const int any_bad = __syncthreads_or(bad_id);
    if (tid == 0) q.gstat[kGstatStride * g + 1 + part] = any_bad ? 2 : 0;
}

// goal node of graph g: argmin_i |v_i - goal|^2, lowest index on ties (model.py:132); any power-of-two workgroup
__device__ __forceinline__ void goal_body(int C, const float* __restrict__ v, const float* __restrict__ goal,
                                          int n0, int n, int g, int n0_pad, int* __restrict__ goal_node) {
    __shared__ float s_d[16];
    __shared__ int s_i[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < n; i += nt) {
        float d = 0.f;
        for (int c = 0; c < C; ++c) {
            const float x = v[(size_t)(n0 + i) * C + c] - goal[(size_t)g * C + c];
            d = fmaf(x, x, d);
        }
        if (d < best) { best = d; bi = i; }
    }
    // (distance, index) minimum, lowest index on ties: inside a wave by shuffles, then the wave results through LDS -- two
    // barriers instead of one per halving step (this runs in the one-workgroup prep stage of a single-graph forward)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_down(best, off);
        const int oi = __shfl_down(bi, off);
        if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
    }
    if ((tid & 63) == 0) { s_d[tid >> 6] = best; s_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid < 64) {
        const int nw = nt >> 6;
        best = tid < nw ? s_d[tid] : INFINITY;
        bi = tid < nw ? s_i[tid] : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_down(best, off);
            const int oi = __shfl_down(bi, off);
            if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
        }
        if (tid == 0) { s_d[0] = best; s_i[0] = bi; }
    }
    __syncthreads();
    if (tid == 0) goal_node[g] = (n > 0) ? n0_pad + s_i[0] : -1;
}

// the same arg-min over the nodes [lo, hi) of graph g only: out[0] = distance bits, out[1] = index within the graph (0x7fffffff if
// the slice is empty); combined over the parts by prep_scatter_kernel
__device__ __forceinline__ void goal_partial(int C, const float* __restrict__ v, const float* __restrict__ goal,
                                             int n0, int lo, int hi, int g, int* __restrict__ out) {
    __shared__ float p_d[16];
    __shared__ int p_i[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int i = lo + tid; i < hi; i += nt) {
        float d = 0.f;
        for (int c = 0; c < C; ++c) {
            const float x = v[(size_t)(n0 + i) * C + c] - goal[(size_t)g * C + c];
            d = fmaf(x, x, d);
        }
        if (d < best) { best = d; bi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_down(best, off);
        const int oi = __shfl_down(bi, off);
        if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
    }
    if ((tid & 63) == 0) { p_d[tid >> 6] = best; p_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < (nt >> 6); ++w)
            if (p_d[w] < best || (p_d[w] == best && p_i[w] < bi)) { best = p_d[w]; bi = p_i[w]; }
        out[0] = __float_as_int(best);
        out[1] = bi;
    }
}

// -----------------------------------------------------------------------------------------------------
// The whole prep stage in ONE launch, `parts` workgroups per graph (1 for graphs up to ~12 k edges, up to 16 for large
// ones): padded prefix arrays (every workgroup reduces the entries before its own graph itself), CSR rows of its slice
// of the graph's target nodes, goal node of the graph (part 0); the last workgroup marks the tiles / slots / nodes
// behind the last graph as unused.  A dependent launch costs ~5 us on this part whatever it does, and the stage used
// to be six to eight of them (two fills of up to 40 MB, prefix arrays, count, scan, fill, tile metadata, goal node):
// 31 -> 19 us for the reference's own call pattern (ONE graph per forward, eval_gnn.py:194).
// -----------------------------------------------------------------------------------------------------
// padded prefix sums over the graphs before graph g (every workgroup reduces them itself -- G loads spread over 1024
// threads -- instead of waiting for a separate scan launch); part 0 publishes the entries of its graph
template <bool FENCE>
__device__ __forceinline__ void prep_prefix(const PrepParams& q, int g, int part, int parts, int& n0, int& n1, int& e0, int& e1) {
    __shared__ long long red[3][16];
    const int tid = threadIdx.x;
    if (q.single_out) {                                // one graph given by its totals: its prefix arrays come first
        if (tid < 6) q.single_out[tid] = (tid & 1) ? (tid == 1 ? q.single_n : (tid == 3 ? q.single_e : q.single_o)) : 0;
        // FENCE: the caller goes on to read the arrays through q.node_ptr / q.edge_ptr; the one-launch form reads the kernel
        // arguments instead (prep_np / prep_ep) and does not pay the ~1.5 us of the device-scope fence
        if constexpr (FENCE) {
            __threadfence();
            __syncthreads();
        }
    }
    long long an = 0, ae = 0, ad = 0;
    for (int t = tid; t < g; t += 1024) {
        const int ng = prep_np(q, t + 1) - prep_np(q, t);
        an += round_up(ng, kPad);
        ae += round_up(prep_ep(q, t + 1) - prep_ep(q, t), kPad);
        ad += (long long)ng * ng;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { an += __shfl_down(an, off); ae += __shfl_down(ae, off); ad += __shfl_down(ad, off); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = an; red[1][tid >> 6] = ae; red[2][tid >> 6] = ad; }
    __syncthreads();
    an = 0; ae = 0; ad = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { an += red[0][w]; ae += red[1][w]; ad += red[2][w]; }
    const int ng_own = prep_np(q, g + 1) - prep_np(q, g);
    n0 = (int)an; e0 = (int)ae;
    n1 = n0 + round_up(ng_own, kPad); e1 = e0 + round_up(prep_ep(q, g + 1) - prep_ep(q, g), kPad);
    if (tid == 0 && part == 0) {
        if (g == 0) { q.node_ptr_pad[0] = 0; q.edge_ptr_pad[0] = 0; q.dense_ptr[0] = 0; }
        q.node_ptr_pad[g + 1] = n1; q.edge_ptr_pad[g + 1] = e1; q.dense_ptr[g + 1] = ad + (long long)ng_own * ng_own;
        // more obstacles than the K/V slabs were sized for (gnnmp_batch.max_obstacles is a caller promise): the attention blocks
        // see only the first obs_cap of them (obs_body clamps) -- reported through gnnmp_explorer_status, never silently
        const int og = q.obs_ptr ? q.obs_ptr[g + 1] - q.obs_ptr[g] : q.single_o;
        q.gstat[kGstatStride * g] = og > q.obs_cap ? 1 : 0;
        for (int pp = parts; pp < kGstatStride - 1; ++pp) q.gstat[kGstatStride * g + 1 + pp] = 0;      // slots of parts this launch does not have
    }
}

// tiles / slots / nodes behind the last graph are unused (the padded totals are upper bounds)
__device__ __forceinline__ void prep_trailing(const PrepParams& q, int n_end, int e_end, int Npad, int Epad) {
    const int tid = threadIdx.x;
    for (int i = n_end + tid; i < Npad; i += 1024) { q.deg[i] = 0; q.row_beg[i] = e_end; }
    for (int t = n_end / 32 + tid; t < Npad / 32; t += 1024) q.ntile_graph[t] = -1;
    for (int t = e_end / 32 + tid; t < Epad / 32; t += 1024) { q.etile_graph[t] = -1; q.tile_meta[t] = -1; }
    for (int sl = e_end + tid; sl < Epad; sl += 1024) q.csr[sl] = make_int4(-1, -1, -1, -1);
}

__global__ __launch_bounds__(1024) void prep_small_kernel(PrepParams q, int Npad, int Epad, int parts) {
    extern __shared__ int prep_lds[];
    const int g = blockIdx.x / parts, part = blockIdx.x - g * parts;
    int n0, n1, e0, e1;
    prep_prefix<false>(q, g, part, parts, n0, n1, e0, e1);
    prep_graph_body(q, g, part, parts, n0, n1 - n0, e0, e1, prep_lds);
    if (part == 0) goal_body(q.C, q.v, q.goal, prep_np(q, g), prep_np(q, g + 1) - prep_np(q, g), g, n0, q.goal_node);
    if (g == q.G - 1 && part == parts - 1) prep_trailing(q, n1, e1, Npad, Epad);
}

// -----------------------------------------------------------------------------------------------------
// Large graphs (more than ~12 k edges on average): one workgroup walking all edge columns of a graph is a long
// latency-bound loop (kuka14, 5000 nodes, k = 16: 80 k columns per graph, 32 graphs = 32 busy CUs).  Two launches with
// the COLUMNS of a graph split over `parts` workgroups instead:
//   prep_hist     part p histograms the targets of its column slice (LDS atomics; the returned count is the column's
//                 rank among the slice's columns with that target) and publishes the histogram H[p][node];
//                 plus prefix arrays, goal node, tile maps, pad records
//   prep_scatter  every part sums the histograms (node totals -> scan -> first slot of every node; columns of earlier
//                 parts come first inside a node's run) and scatters the records of its slice
// -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void prep_hist_kernel(PrepParams q, int Npad, int Epad, int parts, int* __restrict__ H) {
    extern __shared__ int prep_lds[];
    // all parts of a graph run on ONE XCD (workgroup b runs on XCD b % 8): they scatter into the same CSR rows, and lines
    // written piecewise from several XCDs' L2s reach HBM as that many partial writes
    const int idx = blockIdx.x >> 3, g = (idx / parts) * 8 + (blockIdx.x & 7), part = idx % parts, tid = threadIdx.x;
    if (g >= q.G) return;
    int n0, n1, e0, e1;
    prep_prefix<true>(q, g, part, parts, n0, n1, e0, e1);
    const int Np = n1 - n0;
    const int c0 = q.edge_ptr[g], Eg = q.edge_ptr[g + 1] - c0;
    const int cs = (int)((long long)Eg * part / parts), ce = (int)((long long)Eg * (part + 1) / parts);
    int* hslice = H + (size_t)n0 * parts + (size_t)part * Np;
    const bool in_lds = Np <= kPrepCap;
    int* cnt = in_lds ? prep_lds : hslice;
    const long long* dsts = q.edge_index + (size_t)q.E + c0;
    for (int i = tid; i < Np; i += 1024) cnt[i] = 0;
    __syncthreads();
    constexpr int U = 16;
    const int Ng = q.node_ptr[g + 1] - q.node_ptr[g];
    int bad_id = 0;                                    // see prep_id
    for (int c = cs + tid; c < ce; c += U * 1024) {
        int d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = (c + u * 1024 < ce) ? prep_id((int)dsts[c + u * 1024], Ng, bad_id) : -1;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (d[u] >= 0) q.cursor[c0 + c + u * 1024] = atomicAdd(&cnt[d[u]], 1);
    }
    {
        const int any_bad = __syncthreads_or(bad_id);
        if (tid == 0) q.gstat[kGstatStride * g + 1 + part] = any_bad ? 2 : 0;
    }
    if (in_lds)
        for (int i = tid; i < Np; i += 1024) hslice[i] = cnt[i];
    // pad slots and the per-tile maps of the graph, shared out over its parts
    for (int sl = e0 + Eg + part * 1024 + tid; sl < e1; sl += parts * 1024) q.csr[sl] = make_int4(-1, -1, -1, -1);
    for (int t = n0 / 32 + part * 1024 + tid; t < n1 / 32; t += parts * 1024) q.ntile_graph[t] = g;
    if (part == 0)
        for (int b = (n0 >> 8) + tid; b < (n1 >> 8); b += 1024) q.blk_span[b] = make_int2(n0 >> 7, n1 >> 7);
    for (int t = e0 / 32 + part * 1024 + tid; t < e1 / 32; t += parts * 1024) {
        q.etile_graph[t] = g;
        q.tile_meta[t] = t * 32 < e0 + Eg ? 4 : 0;
    }
    if (in_lds && q.node_ptr[g + 1] - q.node_ptr[g] >= kGoalSplitNodes) {
        // the goal arg-min (model.py:132) of a 5000-node graph on ONE of its part workgroups was most of the 31 us this kernel spent
        // outside its counting loop: every part takes a slice of the nodes and leaves (distance, local index) in the half of the
        // histogram workspace that graphs of up to kPrepCap rows do not use; prep_scatter_kernel combines them in part order
        const int Ngl = q.node_ptr[g + 1] - q.node_ptr[g];
        const int lo = (int)((long long)Ngl * part / parts), hi = (int)((long long)Ngl * (part + 1) / parts);
        goal_partial(q.C, q.v, q.goal, q.node_ptr[g], lo, hi, g, H + (size_t)parts * Npad + (size_t)n0 * parts + 2 * part);
    } else if (part == parts - 1) {
        goal_body(q.C, q.v, q.goal, q.node_ptr[g], q.node_ptr[g + 1] - q.node_ptr[g], g, n0, q.goal_node);
    }
    if (g == q.G - 1 && part == 0) prep_trailing(q, n1, e1, Npad, Epad);
}

__global__ __launch_bounds__(1024) void prep_scatter_kernel(PrepParams q, int parts, const int* __restrict__ H, int* __restrict__ Bs) {
    extern __shared__ int prep_lds[];
    const int idx = blockIdx.x >> 3, g = (idx / parts) * 8 + (blockIdx.x & 7), part = idx % parts, tid = threadIdx.x;
    if (g >= q.G) return;
    const int n0 = q.node_ptr_pad[g], Np = q.node_ptr_pad[g + 1] - n0, e0 = q.edge_ptr_pad[g];
    const int c0 = q.edge_ptr[g], Eg = q.edge_ptr[g + 1] - c0;
    const int cs = (int)((long long)Eg * part / parts), ce = (int)((long long)Eg * (part + 1) / parts);
    const int* hg = H + (size_t)n0 * parts;
    int* bs = Np <= kPrepCap ? prep_lds : Bs + (size_t)n0 * parts + (size_t)part * Np;     // first slot of my columns per node
    int* scan = prep_lds + kPrepCap;
    if (Np <= kPrepCap) {
        // rows up to the LDS cap: every thread owns ceil(Np / 1024) <= 8 CONSECUTIVE rows, all of their part counts are requested at
        // once and ONE block scan places them (five rounds of loads + scan at 5000 rows were 43 of this kernel's 94 us at the
        // configs[4] shape, repeated by every part)
        constexpr int RPT = kPrepCap / 1024;
        const int rpt = (Np + 1023) >> 10;                 // consecutive rows per thread: all 1024 threads take part whatever Np is
        int tot[RPT], mine[RPT], local = 0;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int i = tid * rpt + k;
            tot[k] = 0; mine[k] = 0;
            if (k < rpt && i < Np)
                for (int pp = 0; pp < parts; ++pp) {
                    const int hv = hg[(size_t)pp * Np + i];
                    tot[k] += hv;
                    mine[k] += pp < part ? hv : 0;
                }
            local += tot[k];
        }
        int all;
        int run = e0 + block_scan_1024(local, scan, all) - local;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int i = tid * rpt + k;
            if (k < rpt && i < Np) {
                bs[i] = run + mine[k];
                if (part == 0) { q.row_beg[n0 + i] = run; q.deg[n0 + i] = tot[k]; }
            }
            run += tot[k];
        }
        // goal node of the graph from the parts' partial minima (prep_hist_kernel), in part order: lowest distance, lowest index on ties
        const int Ngl = q.node_ptr[g + 1] - q.node_ptr[g];
        if (part == 0 && tid < 64 && Ngl >= kGoalSplitNodes) {          // (one wave: lane = part, at most 16 parts)
            const int* gp = Bs + (size_t)n0 * parts;
            float best = tid < parts ? __int_as_float(gp[2 * tid]) : INFINITY;
            int bi = tid < parts ? gp[2 * tid + 1] : 0x7fffffff;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const float od = __shfl_down(best, off);
                const int oi = __shfl_down(bi, off);
                if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
            }
            if (tid == 0) q.goal_node[g] = n0 + bi;
        }
    } else {
    int carry_run = e0;
    for (int base = 0; base < Np; base += 1024) {
        const int i = base + tid;
        int tot = 0, mine = 0;
        if (i < Np)
            for (int pp = 0; pp < parts; ++pp) {
                const int hv = hg[(size_t)pp * Np + i];
                tot += hv;
                mine += pp < part ? hv : 0;
            }
        int chunk_total;
        const int incl = block_scan_1024(tot, scan, chunk_total);
        if (i < Np) {
            const int r = carry_run + incl - tot;
            bs[i] = r + mine;
            if (part == 0) { q.row_beg[n0 + i] = r; q.deg[n0 + i] = tot; }
        }
        carry_run += chunk_total;
    }
    }
    __threadfence_block();
    __syncthreads();                                   // bs is complete
    const long long* srcs = q.edge_index + c0;
    const long long* dsts = q.edge_index + (size_t)q.E + c0;
    constexpr int U = 8;
    const int Ng = q.node_ptr[g + 1] - q.node_ptr[g];
    int bad_src = 0;
    for (int c = cs + tid; c < ce; c += U * 1024) {
        int sv[U], tv[U], rk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c + u * 1024;
            const bool ok = cc < ce;
            sv[u] = ok ? prep_id((int)srcs[cc], Ng, bad_src) : 0;
            tv[u] = ok ? prep_id((int)dsts[cc], Ng, bad_src) : -1;
            rk[u] = ok ? q.cursor[c0 + cc] : 0;
        }
        // The 16-byte records land all over the graph's CSR region and are combined into full lines by the L2 (as non-temporal stores
        // the stage is 70 % slower).  A 5000-node k = 16 graph's region is 2.1 MB and an XCD's 4 MB L2 serves four such graphs at
        // once, so lines left half-written: graphs of >= 64 k edges are scattered in eight passes over windows of target rows -- the
        // workgroups of a launch run in step closely enough that the live region per XCD shrinks with the window (configs[4]
        // shape: prep 0.123 -> 0.097 ms with eight windows, 0.109 with two; at 30 k edges per graph windows do not pay)
        const int nw = Eg >= kPrepWindowEdges ? 8 : 1;
        for (int w = 0; w < nw; ++w) {
            const int wlo = (int)((long long)Np * w / nw), whi = (int)((long long)Np * (w + 1) / nw);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (tv[u] >= wlo && tv[u] < whi)                             // {source, target, caller column}
                    q.csr[bs[tv[u]] + rk[u]] = make_int4(n0 + sv[u], n0 + tv[u], c0 + c + u * 1024, 0);
        }
    }
    if (__syncthreads_or(bad_src) && tid == 0) q.gstat[kGstatStride * g + 1 + part] = 2;     // (the slot was written by prep_hist)
}

// =====================================================================================================
// shared pieces of the register-resident chains
// =====================================================================================================
// y = W2 . relu(W1s . in + b0) + c0          (Seq(Lin, ReLU, Lin) on raw inputs)
template <int NT, int P, class GetIn>
__device__ __forceinline__ void mlp2_in(const float* as0, int ks, const float* b0, const float* a0, const float* c0,
                                        GetIn getin, f32x16 (&y)[NT], int lane) {
    f32x16 hdn[NT];
    load_vec<NT>(b0, hdn, lane);
    linear_in_p<P, NT>(as0, ks, getin, hdn, lane);
    relu_<NT>(hdn);
    load_vec<NT>(c0, y, lane);
    linear_acc_p<P, NT, NT>(a0, hdn, y, lane);
}

// FeedForward (model.py:192-201): x <- LN(w_2 relu(w_1 x + b1) + b2 + x)
template <int NT, int P>
__device__ __forceinline__ void ffn_(const float* w1, const float* b1, const float* w2, const float* b2,
                                     const float* lng, const float* lnb, f32x16 (&x)[NT], int lane) {
    f32x16 hdn[NT], z[NT];
    load_vec<NT>(b1, hdn, lane);
    linear_acc_p<P, NT, NT>(w1, x, hdn, lane);
    relu_<NT>(hdn);
    load_vec<NT>(b2, z, lane);
    linear_acc_p<P, NT, NT>(w2, hdn, z, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t) x[t] += z[t];
    layer_norm_<NT>(x, lng, lnb, 1e-6f, lane);
}

// =====================================================================================================
// obs_kernel: one workgroup (4 waves) per (graph, side); wave w owns obstacle tiles w, w+4, ...
// Weights are read straight from global memory (this stage is ~0.1 % of the work).
// For side in {node, edge}: code = MLP2(ob); for b in 0..2: K = Wk code, V = Wv code -> KV slab;
// code = FFN_obs_b(code).
// KV slab of (graph, block): Ko A-tiles [ot][ft][1024] then Vo A-tiles [ot][ft][1024] (stride ot_max).
// =====================================================================================================
template <int D, int P>
__device__ __forceinline__ void obs_body(const ObsParams& p, const int vblock) {
    constexpr int NT = D / 32;
    constexpr int TF = Prec<P>::TF;
    // workgroup (g, side): the node-side and edge-side obstacle stacks are independent chains of ~14 dependent layers
    // each; side by side they halve the latency of the stage for a single graph and double the workgroups of a batch
    const int g = vblock >> 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    // max_obstacles is a caller promise; clamp so a too-small value truncates the obstacle set instead of
    // writing past the K/V slab (documented in gnnmp.h)
    const int o0 = p.obs_ptr[g], O = min(p.obs_ptr[g + 1] - o0, p.ot_max * 32);
    const int OT = (O + 31) / 32;
    {
        const int side = vblock & 1;
        const float* W = p.w[side];
        const ObsBlob L = p.blob;
        // d = 32: the side's whole blob (55 KB fp32) goes to LDS in one DMA burst first -- the stage is a chain of ~14 dependent
        // layers, and every layer used to start with an L2 round trip for its operands (single-graph forward: 15 -> ~10 us)
        if constexpr (D == 32 && P != 2) {
            extern __shared__ __attribute__((aligned(16))) float obs_lds[];
            stage(obs_lds, W, L.size);
            __syncthreads();
            W = obs_lds;
        }
        float* kv_side = p.kv[side];
        for (int ot = wave; ot < OT; ot += 4) {
            const int o = ot * 32 + j;
            const bool valid = o < O;
            const float* orow = p.obstacles + (size_t)(o0 + (valid ? o : 0)) * p.S;
            const int S = p.S;
            f32x16 code[NT];
            mlp2_in<NT, P>(W + L.as0, L.ks, W + L.b0, W + L.a0, W + L.c0,
                        [&](int k) { const bool ok = k < S; const float x = orow[ok ? k : 0]; return ok ? x : 0.f; }, code, lane);
            for (int b = 0; b < 3; ++b) {
                const float* Wb = W + L.blk0 + b * L.blk_stride;
                f32x16 K[NT], V[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) { K[t] = splat16(0.f); V[t] = splat16(0.f); }
                linear_acc_p<P, NT, NT>(Wb + L.wk, code, K, lane);
                linear_acc_p<P, NT, NT>(Wb + L.wv, code, V, lane);
                float* slab = kv_side + (size_t)(g * 3 + b) * p.kv_stride;
                if (!valid) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) { K[t] = splat16(0.f); V[t] = splat16(0.f); }
                }
                float* vbase = slab + (size_t)p.ot_max * NT * TF;
                // Vo is the A operand of P.V: A[i = feature][k = obstacle].  Element V[o = 32 ot + j][f = 32 t + phi(r,h)]
                // belongs to lane' = phi(r,h) + 32 h' at register index r' where phi(r', h') = j
                const int hp = (j >> 2) & 1, rp = (j & 3) + 4 * (j >> 3);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float* kt = slab + (size_t)(ot * NT + t) * TF;
                    float* vt = vbase + (size_t)(ot * NT + t) * TF;
                    if constexpr (P == 0) {
                        // Ko: A[i = obstacle][k = feature] -> tile (ot, ft) is exactly the register block
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 a;
#pragma unroll
                            for (int c = 0; c < 4; ++c) a[c] = K[t][q * 4 + c];
                            *reinterpret_cast<f32x4*>(kt + (q * 64 + lane) * 4) = a;
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int lp = phi(r, h) + 32 * hp;
                            vt[((rp >> 2) * 64 + lp) * 4 + (rp & 3)] = V[t][r];
                        }
                    } else if constexpr (P == 1) {
                        const BOp<1> kb(K[t]);                       // registers 0..7 -> MFMA 0, 8..15 -> MFMA 1
                        *reinterpret_cast<bf16x8*>(kt + lane * 4) = kb.lo;
                        *reinterpret_cast<bf16x8*>(kt + (64 + lane) * 4) = kb.hi;
                        __bf16* vtb = reinterpret_cast<__bf16*>(vt);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int lp = phi(r, h) + 32 * hp;
                            vtb[((rp >> 3) * 64 + lp) * 8 + (rp & 7)] = (__bf16)V[t][r];
                        }
                    } else {
                        const BOp<2> kb(K[t]);                       // three exact bf16 pieces of every key
                        __bf16* vtb = reinterpret_cast<__bf16*>(vt);
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) {
                            *reinterpret_cast<bf16x8*>(kt + ((pc * 2) * 64 + lane) * 4) = kb.lo[pc];
                            *reinterpret_cast<bf16x8*>(kt + ((pc * 2 + 1) * 64 + lane) * 4) = kb.hi[pc];
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int lp = phi(r, h) + 32 * hp;
                            float res = V[t][r];
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) {
                                const __bf16 piece = (__bf16)res;
                                vtb[((pc * 2 + (rp >> 3)) * 64 + lp) * 8 + (rp & 7)] = piece;
                                res -= (float)piece;
                            }
                        }
                    }
                }
                if (b < 2)
                    ffn_<NT, P>(Wb + L.fw1, Wb + L.fb1, Wb + L.fw2, Wb + L.fb2, Wb + L.lng, Wb + L.lnb, code, lane);
            }
        }
    }
}


// =====================================================================================================
// node_f64_body: the node side's node_free_code encoder (model.py:122) and the attention sub-block of node block 0
// (model.py:164-181 for node_attentions.0) in DOUBLE precision -- the one stretch of the forward where fp32 rounding is
// amplified: obstacle logits of block 0 are nearly uniform (|x| < 0.3), the attended value is close to a plain average,
// and the LayerNorm behind it divides by the small spread of (average + node_free_code); with fp32 there the scores move
// by ~1e-5 (the reference's own fp32-vs-fp64 distance on the 116-obstacle mazes is made of exactly this,
// tools/diag/parity_sensitivity*.py), everything else contributes <= 3e-6.  The node side is 1/11 of the rows and this is
// one of its three blocks, so double precision here costs a few per cent of the step.
//
// v_mfma_f64_16x16x4_f64 (64 cycles, probed with tools/microbench/mfma_f64_probe.hip): A[i][k] in lane i + 16 k,
// B[k][j] in lane j + 16 k, D[i][j] in lane j + 16 (i % 4), register i / 4.  One wave owns 16 nodes; a node's features live
// in the four lanes j, j+16, j+32, j+48: lane (j, g) holds feature 16 b + 4 r + g in register r of block b -- which is what
// the next layer's B operand needs at k-step r (k slot g <-> input feature 4 r + g), so chains of layers run in registers
// in the natural feature order.  Softmax weights come out as lane (j, g), register r <-> obstacle 16 ob + 4 r + g: the B
// operand of P.V with the same rule.
// Workgroup = 4 waves = 64 padded node rows of ONE graph; it builds the graph's node-side block-0 obstacle operands
// K' = Wqk code, V = Wv code itself (fp32 chain code of obs_body, f64_obs_chunk obstacles at a time into LDS) so that the role has
// no dependency on the obstacle role and both run in the same launch.
// =====================================================================================================
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double gsum4(double x) {
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}
// max over the four lanes of a node on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap: no LDS round trip in the
// per-block critical path): swapping a value with itself leaves {rows 0,0,2,2} and {rows 1,1,3,3} (then {halves 0,0} and {1,1})
__device__ __forceinline__ float gmax4f(float x) {
    const unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned v = __float_as_uint(m);
    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// exp(x) for x <= 0 (softmax weights relative to the running maximum), double precision: x = n ln2 + r, |r| <= 0.347,
// Taylor polynomial of degree 10 (remainder < 3e-13 relative: six digits beyond fp32), scaled by 2^n.  Written with
// full-rate instructions only: n by the 1.5 * 2^52 trick (its integer value is the low word of the sum), the scaling by an
// integer add into the exponent field (the polynomial is in [0.7, 1.42] and n >= -1010 after the clamp, so the result stays
// normal).  A masked obstacle (-inf) comes out as e^-700 = 1e-304: nothing, next to a denominator >= 1.
__device__ __forceinline__ double exp_nonpos(double x) {
    x = fmax(x, -700.0);
    const double t = fma(x, 1.4426950408889634, 6755399441055744.0);
    const double n = t - 6755399441055744.0;
    double r = fma(-n, 6.93147180369123816490e-01, x);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.0 / 3628800.0;
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int ni = __double2loint(t);
    return __hiloint2double(__double2hiint(p) + (ni << 20), __double2loint(p));
}

// y[ob] += sum_ib A[ob][ib] . x[ib]   (A: [NB][NB][4][64] floats)
template <int NB>
__device__ __forceinline__ void lin64(const float* A, const f64x4 (&x)[NB], f64x4 (&y)[NB], int lane) {
#pragma unroll
    for (int ib = 0; ib < NB; ++ib)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
            for (int ob = 0; ob < NB; ++ob)
                y[ob] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)A[((ob * NB + ib) * 4 + st) * 64 + lane], x[ib][st], y[ob], 0, 0, 0);
            // d = 64: without a fence the scheduler hoists all 64 operand loads and conversions of a layer (128 registers)
            if constexpr (NB > 2) __builtin_amdgcn_sched_barrier(0);
        }
}

template <int NB>
__device__ __forceinline__ void vec64(const float* b, f64x4 (&y)[NB], int g) {
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) y[ob][r] = (double)b[16 * ob + 4 * r + g];
}

template <int D, int P>
__device__ __forceinline__ void node_f64_body(const ObsParams& p, const NodeF64Params& q, const int vblock) {
    constexpr int NB = D / 16, NT = D / 32, OC = f64_obs_chunk(D);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const F64Blob L = q.blob;
    float* wl = lds;
    float* kd = lds + ((L.size + 3) & ~3);
    float* vd = kd + OC * D;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j16 = lane & 15, g = lane >> 4;
    // a workgroup owns q.groups consecutive 64-row groups of ONE graph's padded node range (kPad is a multiple of 64 *
    // groups): the obstacle operands are built once per workgroup
    const int row00 = vblock * 64 * q.groups;
    const int gr = q.ntile_graph[row00 >> 5];
    if (gr < 0) return;
    stage(wl, q.w, L.size);
    const int o0 = p.obs_ptr[gr], O = min(p.obs_ptr[gr + 1] - o0, p.ot_max * 32);
    const int OT = (O + 31) / 32;
    const int C = q.C;
    const int nbase_pad = q.node_ptr_pad[gr], nbase = q.node_ptr[gr], ng = q.node_ptr[gr + 1] - nbase;
    const bool single_chunk = O <= OC;
    const double isd = 1.0 / sqrt((double)D);
    for (int grp = 0; grp < q.groups; ++grp) {
    const int row0 = row00 + grp * 64;
    if (row0 - nbase_pad >= ng) {                          // (uniform) nothing but padding rows from here on: finite rows, no work
        for (int i = threadIdx.x; i < (q.groups - grp) * 64 * D; i += 256) q.m0[(size_t)row0 * D + i] = 0.f;
        break;
    }
    const int row = row0 + wave * 16 + j16, local = row - nbase_pad;
    const float* vr = q.v + (size_t)(nbase + (local < ng ? local : 0)) * C;

    f64x4 x[NB], acc[NB];
    float xf[NB][4];
    float mxs = 0.f, l0s = 0.f;
    double den = 0.0;
    for (int oc0 = 0; oc0 == 0 || oc0 < O; oc0 += OC) {
        if (oc0 > 0 || (grp > 0 && !single_chunk)) __syncthreads();    // everybody is done with the previous chunk's operands
        if (grp == 0 || !single_chunk)
        {   // obstacle operands of this chunk: wave w owns the 32-obstacle tiles oc0/32 + w, ... (obs_body's arithmetic for b = 0)
            const ObsBlob Lo = p.blob;
            const int h = lane >> 5, j = lane & 31, S = p.S;
            for (int ot = (oc0 >> 5) + wave; ot < min(OT, (oc0 + OC) >> 5); ot += 4) {
                // the weights are MFMA operands read from global memory: loop-invariant loads that the compiler would hoist out
                // of the three enclosing loops into ~200 registers (spills at d = 64) unless the pointer is laundered here
                const float* W = p.w[0];
                asm volatile("" : "+s"(W));
                const int o = ot * 32 + j;
                const bool valid = o < O;
                const float* orow = p.obstacles + (size_t)(o0 + (valid ? o : 0)) * S;
                f32x16 code[NT], K[NT], V[NT];
                mlp2_in<NT, P>(W + Lo.as0, Lo.ks, W + Lo.b0, W + Lo.a0, W + Lo.c0,
                               [&](int k) { const bool ok = k < S; const float xx = orow[ok ? k : 0]; return ok ? xx : 0.f; }, code, lane);
#pragma unroll
                for (int t = 0; t < NT; ++t) { K[t] = splat16(0.f); V[t] = splat16(0.f); }
                linear_acc_p<P, NT, NT>(W + Lo.blk0 + Lo.wk, code, K, lane);
                linear_acc_p<P, NT, NT>(W + Lo.blk0 + Lo.wv, code, V, lane);
                // scatter into the operand layouts: kd[((ob NB + f/16) 4 + (f%16)/4) 64 + o%16 + 16 (f%4)] = K'[o][f] (lane = obstacle +
                // 16 k-slot), vd[((ob NB + f/16) 4 + o%4) 64 + f%16 + 16 ((o%16)/4)] = V[o][f] (lane = feature + 16 k-slot), ob = o/16.
                // With f = 32 t + phi(r, h) every index is one per-lane base plus a compile-time offset (registers matter at d = 64)
                const int oo = o - oc0;
                float* kb = kd + (oo >> 4) * NB * 256 + (oo & 15) + 64 * h;
                float* vb = vd + (oo >> 4) * NB * 256 + (oo & 3) * 64 + 16 * ((oo & 15) >> 2) + 4 * h;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f0 = 32 * t + phi(r, 0);                       // h adds 4: same 16-block, next k-step / next lane group
                        kb[(f0 >> 4) * 256 + ((f0 & 15) >> 2) * 64 + 16 * (f0 & 3)] = valid ? K[t][r] : 0.f;
                        vb[(f0 >> 4) * 256 + (f0 & 15)] = valid ? V[t][r] : 0.f;
                    }
            }
        }
        if (grp == 0 || !single_chunk) __syncthreads();    // operands (and, the first time, the staged weights) are in LDS
        if (oc0 == 0) {
            // node_free_code = W2 relu(W1 v + b1) + b2
            f64x4 hdn[NB];
            vec64<NB>(wl + L.b1, hdn, g);
            for (int st = 0; st < L.ks; ++st) {
                const int k = 4 * st + g;
                const double xin = k < C ? (double)vr[k] : 0.0;
#pragma unroll
                for (int ob = 0; ob < NB; ++ob)
                    hdn[ob] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)wl[L.w1 + (ob * L.ks + st) * 64 + lane], xin, hdn[ob], 0, 0, 0);
            }
#pragma unroll
            for (int ob = 0; ob < NB; ++ob)
#pragma unroll
                for (int r = 0; r < 4; ++r) hdn[ob][r] = fmax(hdn[ob][r], 0.0);
            vec64<NB>(wl + L.b2, x, g);
            lin64<NB>(wl + L.w2, hdn, x, lane);
            // self term: logit m . (Wqk m), value Wv m with weight exp(0)
#pragma unroll
            for (int ob = 0; ob < NB; ++ob)
#pragma unroll
                for (int r = 0; r < 4; ++r) xf[ob][r] = (float)x[ob][r];
            // Wqk m and Wv m on the fp32 pipe (neither is sensitive: 4e-9 / 2e-7 on the scores when rounded to fp32); the rows of
            // these two operands are packed permuted so that the f32 result layout (feature 4 g + r) lands on the f64 one (4 r + g)
            f32x4 tq[NB], mv[NB];
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) { tq[ob] = f32x4{0.f, 0.f, 0.f, 0.f}; mv[ob] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ib = 0; ib < NB; ++ib)
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int ob = 0; ob < NB; ++ob) {
                        tq[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[L.wqk + ((ob * NB + ib) * 4 + st) * 64 + lane], xf[ib][st], tq[ob], 0, 0, 0);
                        mv[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[L.wv + ((ob * NB + ib) * 4 + st) * 64 + lane], xf[ib][st], mv[ob], 0, 0, 0);
                        if constexpr (NB > 2) __builtin_amdgcn_sched_barrier(0);
                    }
            double l0 = 0.0;
#pragma unroll
            for (int ob = 0; ob < NB; ++ob)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    l0 = fma(x[ob][r], (double)tq[ob][r], l0);
                    acc[ob][r] = (double)mv[ob][r];
                }
            l0s = (float)gsum4(l0);                        // UNSCALED self logit, in float like the obstacle logits
        }
        // Softmax over the chunk in two passes (its logits stay in registers): no rescaling of the accumulators between
        // blocks, so they stay in the matrix pipe's registers, and one exp per weight.  Across chunks (O > 128): online.
        // pass 1: logits on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, half the time of the f64 one): a logit error of
        // 1e-7 moves a softmax weight by 1e-7 relative, no more -- the sensitive quantities are the weights' OWN rounding,
        // the weighted sum and the LayerNorm input, which stay in double.  Result layout of the f32 instruction: lane
        // (j, g), register r <-> obstacle 16 ob + 4 g + r (the f64 one has 4 r + g); vd is packed to match.
        const int nblk = (min(O, oc0 + OC) - oc0 + 15) >> 4;
        f32x4 sl[OC / 16];
        float lmax = -INFINITY;
#pragma unroll
        for (int ob = 0; ob < OC / 16; ++ob) {
            f32x4 sv = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (ob < nblk) {
                sv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ib = 0; ib < NB; ++ib)
#pragma unroll
                    for (int st = 0; st < 4; ++st)
                        sv = __builtin_amdgcn_mfma_f32_16x16x4f32(kd[((ob * NB + ib) * 4 + st) * 64 + lane], xf[ib][st], sv, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sv[r] = (oc0 + 16 * ob + 4 * g + r < O) ? sv[r] : -INFINITY;
                    lmax = fmaxf(lmax, sv[r]);
                }
            }
            sl[ob] = sv;
        }
        const float nmxs = fmaxf(oc0 == 0 ? l0s : mxs, gmax4f(lmax));
        {
            // first chunk: the self term enters with weight exp(self - max); later chunks: rescale what has been summed
            const double alpha = exp_nonpos(((double)(oc0 == 0 ? l0s : mxs) - (double)nmxs) * isd);
            den = (oc0 == 0) ? (g == 0 ? alpha : 0.0) : den * alpha;     // lane-local share of the denominator, summed over g at the end
#pragma unroll
            for (int fb = 0; fb < NB; ++fb) acc[fb] *= alpha;
        }
        mxs = nmxs;
#pragma unroll
        for (int ob = 0; ob < OC / 16; ++ob) {
            if (ob < nblk) {
                f64x4 pr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pr[r] = exp_nonpos(((double)sl[ob][r] - (double)nmxs) * isd);     // 0 for the padding obstacles
                    den += pr[r];
                }
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int fb = 0; fb < NB; ++fb)
                        acc[fb] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)vd[((ob * NB + fb) * 4 + st) * 64 + lane], pr[st], acc[fb], 0, 0, 0);
            }
        }
    }
    // value mix + residual + LayerNorm (biased variance, eps 1e-6 inside the root: model.py:181), rounded to fp32 once
    const double inv = 1.0 / gsum4(den);
    double sum = 0.0;
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x[ob][r] = fma(acc[ob][r], inv, x[ob][r]);
            sum += x[ob][r];
        }
    const double mean = gsum4(sum) * (1.0 / D);
    double sq = 0.0;
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x[ob][r] -= mean;
            sq = fma(x[ob][r], x[ob][r], sq);
        }
    const double rstd = 1.0 / sqrt(gsum4(sq) * (1.0 / D) + 1e-6);
    float* out = q.m0 + (size_t)row * D;
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * ob + 4 * r + g;
            out[f] = (float)(x[ob][r] * rstd * (double)wl[L.lng + f] + (double)wl[L.lnb + f]);
        }
    }   // 64-row group
}

// the obstacle launch: workgroups [0, f64_blocks) run node_f64_body (fp32-class modes only), the rest obs_body
template <int D, int P>
__global__ __launch_bounds__(256) void obs_kernel(ObsParams p, NodeF64Params q, int f64_blocks) {
    // q.f64_first: the double-precision workgroups (long: 256 node rows each) come first and the short obstacle workgroups fill
    // the slots that free up in their last round; otherwise the obstacle workgroups lead (few graphs: both fit at once anyway)
    const int obs_blocks = (int)gridDim.x - f64_blocks;
    if constexpr (P != 1) {
        const int b = (int)blockIdx.x;
        const bool is_f64 = q.f64_first ? b < f64_blocks : b >= obs_blocks;
        if (is_f64) {
            node_f64_body<D, P>(p, q, q.f64_first ? b : b - obs_blocks);
            return;
        }
        obs_body<D, P>(p, q.f64_first ? b - f64_blocks : b);
        return;
    }
    obs_body<D, P>(p, (int)blockIdx.x);
}

// =====================================================================================================
// attention Block on the map rows held in registers (model.py:164-181 + map_feed :212-216).
// wl: LDS copy of AttBlob<D>; kvl: LDS K/V chunk region; kvg: this (graph, block)'s slab in global.
// =====================================================================================================
// MAYSKIP && skip (workgroup-uniform): m already went through the attention sub-block (node_f64_body), only map_feed runs
template <int D, int P, bool MAYSKIP = false>
__device__ __forceinline__ void attention_block(const float* wl, const float* wg, float* kvl, const float* kvg, int O,
                                                int ot_max, int ot_chunk, f32x16 (&m)[D / 32], int lane, bool skip = false) {
    constexpr int NT = D / 32;
    constexpr int TF = Prec<P>::TF;
    using L = AttBlob<D, P>;
    if (MAYSKIP && skip) {
        ffn_<NT, P>(wl + L::w1, wg + L::b1, wl + L::w2, wg + L::b2, wg + L::ln2g, wg + L::ln2b, m, lane);
        return;
    }
    const int h = lane >> 5;
    const int OT = (O + 31) / 32;
    f32x16 acc[NT];
    float mx, psum;
    BOp<P> mop[NT];                                             // one conversion / split of m feeds Wqk, Wv and the logits
    make_ops<P, NT>(m, mop);
    {
        f32x16 Tq[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { Tq[t] = splat16(0.f); acc[t] = splat16(0.f); }
        linear_acc_ops<P, NT, NT>(wl + L::wqk, mop, Tq, lane);  // Wqk m: the self logit is m . (Wqk m)
        linear_acc_ops<P, NT, NT>(wl + L::wv, mop, acc, lane);  // acc starts as 1 * V_self
        float l0 = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) l0 = fmaf(m[t][r], Tq[t][r], l0);
        l0 = xsum(l0);
        // softmax(x / sqrt(d)) evaluated as exp2((x - max) * log2(e) / sqrt(d))
        mx = l0;
        psum = (h == 0) ? 1.0f : 0.0f;     // the self term exp(0) is counted once per row
    }
    const float cs = 1.4426950408889634f / sqrtf((float)D);
    const int chunk_floats = ot_chunk * NT * TF;
    const BOp<P> (&qop)[NT] = mop;                              // obstacle keys are premultiplied with Wqk (obs_kernel)
    for (int c0 = 0; c0 < OT; c0 += ot_chunk) {
        const int c1 = min(OT, c0 + ot_chunk);
        if (c0 > 0) {       // first chunk was staged together with the weights
            __syncthreads();
            stage(kvl, kvg + (size_t)c0 * NT * TF, (c1 - c0) * NT * TF);
            stage(kvl + chunk_floats, kvg + (size_t)(ot_max + c0) * NT * TF, (c1 - c0) * NT * TF);
            __syncthreads();
        }
        for (int ot = c0; ot < c1; ++ot) {
            const float* ko = kvl + (size_t)(ot - c0) * NT * TF;
            const float* vo = kvl + chunk_floats + (size_t)(ot - c0) * NT * TF;
            f32x16 s = splat16(0.f);
#pragma unroll
            for (int t = 0; t < NT; ++t) mfma_tile_p<P>(ko + t * TF, qop[t], s, lane);
            // fp32 MFMA and VALU share the SIMD's issue time (tools/microbench/mfma_chain.hip), so every VALU
            // instruction here costs MFMA throughput: mask only the one partial tile, fold the scale into an
            // fma, and skip the accumulator rescale while the running maximum does not move.
            float nmx, ps;
            if (P == 1 && O - ot * 32 <= 8) {
                // wave-uniform: at most eight obstacles in this tile (the robot-arm environments have five boxes):
                // obstacle phi(r, h) lives in registers 0..3 only, the softmax arithmetic runs on those four.  Compiled
                // into the bf16 kernels only (the bf16 configs are the robot arms; in the fp32 maze kernel the extra
                // branch cost 1.7 % through code layout alone)
                float tmax = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[r] = ((ot * 32 + phi(r, h)) < O) ? s[r] : -INFINITY;
                    tmax = fmaxf(tmax, s[r]);
                }
                tmax = xmax(tmax);
                nmx = fmaxf(mx, tmax);
                const float off = -nmx * cs;
                ps = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], cs, off));
                    ps += s[r];
                }
#pragma unroll
                for (int r = 4; r < 16; ++r) s[r] = 0.f;
            } else {
                if ((ot + 1) * 32 > O) {                 // wave-uniform: only the last tile can hold padding obstacles
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = ((ot * 32 + phi(r, h)) < O) ? s[r] : -INFINITY;
                }
                float tmax = s[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
                tmax = xmax(tmax);
                nmx = fmaxf(mx, tmax);
                const float off = -nmx * cs;
                s = s * cs + off;                           // packed fmas
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
                ps = tree_sum(s);
            }
            const int nq = min(4, (O - ot * 32 + 7) >> 3);
            const BOp<P> pop(s);
            if (__builtin_amdgcn_ballot_w64(nmx != mx) != 0) {      // some row's maximum moved: rescale (alpha = 1 elsewhere)
                const float alpha = __builtin_amdgcn_exp2f((mx - nmx) * cs);
                psum *= alpha;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] *= alpha;
            }
            psum += ps;
#pragma unroll
            for (int t = 0; t < NT; ++t) mfma_tile_q_p<P>(vo + t * TF, pop, acc[t], lane, nq);
            mx = nmx;
        }
    }
    const float den = xsum(psum);                                // reciprocal: hardware estimate + one Newton step
    float inv = __builtin_amdgcn_rcpf(den);
    inv = inv * (2.0f - den * inv);
#pragma unroll
    for (int t = 0; t < NT; ++t) m[t] = acc[t] * inv + m[t];         // value mix + residual
    layer_norm_<NT>(m, wg + L::ln1g, wg + L::ln1b, 1e-6f, lane);        // vectors: global (L1/L2 hits)
    ffn_<NT, P>(wl + L::w1, wg + L::b1, wl + L::w2, wg + L::b2, wg + L::ln2g, wg + L::ln2b, m, lane);
}

// =====================================================================================================
// pre_kernel: encoders + 3 attention blocks + loop-invariant epilogue, one 32-row tile per wave.
// =====================================================================================================
template <int D, int P, bool EDGE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void pre_kernel(PreParams p) {
    constexpr int NT = D / 32;
    constexpr int TF = Prec<P>::TF;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    float* kvl = lds + p.wregion;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    // XCD-aware order (see XcdWalk): XCD b % 8 works on a contiguous eighth of the workgroup tiles, so a
    // graph's K/V slabs and weights are staged from ONE XCD's L2
    // The eighths are cut from the workgroup tiles IN USE (the prep stage's padded total), not from the launch's upper bound:
    // up to 255 spare rows per graph sit behind the last graph and would leave the last XCDs without work
    const int n_wg = min(p.n_wg, p.ptr_pad_total[p.G] / (32 * WAVES));
    const int per = (n_wg + 7) >> 3;
    const int wg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || wg >= n_wg) return;
    const int tile = wg * WAVES + wave;
    const int g = p.tile_graph[wg * WAVES];               // kPad is a multiple of 32*WAVES: uniform per WG
    if (g < 0) return;
    const int row = tile * 32 + j;                        // padded index
    const int nbase_pad = p.node_ptr_pad[g], nbase = p.node_ptr[g];
    const int C = p.C;

    stage(wl, p.enc, p.encb.size);
    __syncthreads();
    const EncBlob E = p.encb;
    f32x16 m[NT], aux[NT];      // m: the "free" code that goes through attention; aux: node_code / edge_code
    if constexpr (EDGE) {
        const int4 rec = p.csr[row];
        const int s = rec.x, t = rec.y;
        // the message-passing kernels read 4 bytes per edge: source id within the graph | (target's row in its 32-node
        // tile) << 27 (written here, coalesced, by the compute-bound kernel that reads the 16-byte records anyway)
        if (h == 0 && s >= 0) p.rec32[row] = (s - nbase_pad) | (((t - nbase_pad) & 31) << 27);
        const float* vs = p.v + (size_t)(nbase + (s >= 0 ? s - nbase_pad : 0)) * C;
        const float* vt = p.v + (size_t)(nbase + (t >= 0 ? t - nbase_pad : 0)) * C;
        auto getin = [&](int k) {                           // [v_src, v_dst], branch-free
            const bool ok = k < 2 * C;
            const float* base = (k < C) ? vs : vt - C;
            const float x = base[ok ? k : C];
            return ok ? x : 0.f;
        };
        mlp2_in<NT, P>(wl + E.as0, E.ks0, wl + E.b0, wl + E.a0, wl + E.c0, getin, aux, lane);   // edge_code
        mlp2_in<NT, P>(wl + E.as1, E.ks1, wl + E.b1, wl + E.a1, wl + E.c1, getin, m, lane);     // edge_free_code
    } else {
        const int local = row - nbase_pad;
        const int ng = p.node_ptr[g + 1] - nbase;
        const float* vr = p.v + (size_t)(nbase + (local < ng ? local : 0)) * C;
        const float* gl = p.goal + (size_t)g * C;
        auto getin_nc = [&](int k) {                        // [v, goal, (v-goal)^2, v-goal]  model.py:119, branch-free
            const bool ok = k < 4 * C;
            const int kk = ok ? k : 0;
            const int part = kk / C, c = kk - part * C;
            const float x = vr[c], gg = gl[c];
            const float dlt = x - gg;
            const float val = part == 0 ? x : (part == 1 ? gg : (part == 2 ? dlt * dlt : dlt));
            return ok ? val : 0.f;
        };
        auto getin_nf = [&](int k) { const bool ok = k < C; const float x = vr[ok ? k : 0]; return ok ? x : 0.f; };
        mlp2_in<NT, P>(wl + E.as0, E.ks0, wl + E.b0, wl + E.a0, wl + E.c0, getin_nc, aux, lane);  // node_code
        if (p.m0) load_row<NT>(p.m0 + (size_t)row * D, m, h);                                     // node_f64_body's rows
        else mlp2_in<NT, P>(wl + E.as1, E.ks1, wl + E.b1, wl + E.a1, wl + E.c1, getin_nf, m, lane);    // node_free_code
    }

    if (p.use_obstacles) {
        const int O = min(p.obs_ptr[g + 1] - p.obs_ptr[g], p.ot_max * 32);
        const int OT = (O + 31) / 32;
        const int chunk_floats = p.ot_chunk * NT * TF;
        for (int b = 0; b < 3; ++b) {
            const float* kvg = p.kv + (size_t)(g * 3 + b) * p.kv_stride;
            __syncthreads();
            const float* attg = p.att + (size_t)b * AttBlob<D, P>::size;
            stage(wl, attg, AttBlob<D, P>::staged);
            const int c1 = min(OT, p.ot_chunk);
            stage(kvl, kvg, c1 * NT * TF);
            stage(kvl + chunk_floats, kvg + (size_t)p.ot_max * NT * TF, c1 * NT * TF);
            __syncthreads();
            attention_block<D, P, !EDGE>(wl, attg, kvl, kvg, O, p.ot_max, p.ot_chunk, m, lane, !EDGE && b == 0 && p.m0 != nullptr);
        }
    }

    if (p.om) store_row<NT>(p.om + (size_t)row * D, m, h);       // training path: frozen node_/edge_free_code
    __syncthreads();
    stage(wl, p.out, p.out_size);
    __syncthreads();
    if constexpr (EDGE) {
        using L = OutEBlob<D, P>;
        f32x16 y[NT];
        BOp<P> mop[NT];
        make_ops<P, NT>(m, mop);
#pragma unroll
        for (int t = 0; t < NT; ++t) y[t] = aux[t];                          // aux = W1e.edge_code + b1 (folded at pack time)
        linear_acc_ops<P, NT, NT>(wl + L::w1d, mop, y, lane);
        store_tile_p<P, NT>(p.o0 + (size_t)tile * NT * kETile, y, lane);     // K_e
        load_vec<NT>(wl + L::bp0, y, lane);
        linear_acc_ops<P, NT, NT>(wl + L::wpc, mop, y, lane);
        store_tile_p<P, NT>(p.o1 + (size_t)tile * NT * kETile, y, lane);     // PE_e
    } else {
        using L = OutNBlob<D, P>;
        const bool isgoal = (row == p.goal_node[g]);
        f32x16 xi[NT], tmp[NT], y[NT];
        load_vec<NT>(wl + L::be, xi, lane);
        linear_acc_p<P, NT, NT>(wl + L::we_nc, aux, xi, lane);
        linear_acc_p<P, NT, NT>(wl + L::we_nf, m, xi, lane);
        load_vec<NT>(wl + L::weg, tmp, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) xi[t] += isgoal ? tmp[t] : splat16(0.f);
        store_row<NT>(p.o0 + (size_t)row * D, xi, h);                         // XI
        load_vec<NT>(wl + L::wehg, tmp, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) xi[t] += isgoal ? tmp[t] : splat16(0.f);
        store_row_p<P == 1 ? 1 : 0, NT>(p.o1, (size_t)row, xi, h);                // X_0 (bf16 rows in the bf16 mode: its only reader is an MFMA operand)
#pragma unroll
        for (int t = 0; t < NT; ++t) y[t] = splat16(0.f);
        linear_acc_p<P, NT, NT>(wl + L::wsrc, xi, y, lane);
        store_row_p<P, NT>(p.o2, (size_t)row, y, h);                           // A_0
#pragma unroll
        for (int t = 0; t < NT; ++t) y[t] = splat16(0.f);
        linear_acc_p<P, NT, NT>(wl + L::wdst, xi, y, lane);
        store_row_p<P, NT>(p.o3, (size_t)row, y, h);                           // B_0
        load_vec<NT>(wl + L::bd, y, lane);
        linear_acc_p<P, NT, NT>(wl + L::wd_nc, aux, y, lane);
        store_row<NT>(p.o4 + (size_t)row * D, y, h);                          // DN
    }
}

// =====================================================================================================
// pre_resident_kernel: same arithmetic as pre_kernel, different residency.  When the three attention
// blocks' weights (3 x 20 KB at d = 32) AND a graph's K/V operands for all three blocks (3 x 32 KB at
// O <= 128) fit the 160 KB of LDS together, one 12-wave workgroup per CU keeps them resident: weights
// are staged once per launch, K/V once per (workgroup, graph) -- instead of once per 128-edge tile and
// phase -- and there is NO barrier inside a tile, so the three waves of a SIMD drift apart and overlap
// each other's MFMA and VALU segments.  Work split: the padded tile space is cut into gridDim.x equal
// contiguous shares (XCD-contiguous); inside a share waves pull 32-row tiles from an LDS counter.
// Encoder / epilogue weights (22 KB) are read as MFMA A operands straight from global memory (L1/L2).
// =====================================================================================================
template <int D, int P, bool EDGE, int WAVES>
__device__ __forceinline__ void pre_resident_body(const PreParams& p, const int vblock, const int vgrid) {
    constexpr int NT = D / 32;
    constexpr bool kRecomputeAux = EDGE && D == 64;
    using AB = AttBlob<D, P>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;                                           // [3][AB::size]: matrices AND the six vectors of a block
    float* kvl = lds + 3 * AB::size;                           // [3][kv_stride]
    int* ctr = reinterpret_cast<int*>(kvl + 3 * (size_t)p.kv_stride);
    float* outl = reinterpret_cast<float*>(ctr) + 16;          // epilogue blob, when the launch reserved room (p.out_in_lds)
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int C = p.C;
    const int total = p.ptr_pad_total[p.G] / 32;               // padded 32-row tiles actually in use
    const int nb = vgrid >> 3;
    const int wg = (vblock & 7) * nb + (vblock >> 3);           // XCD-contiguous order
    const int share = (total + vgrid - 1) / vgrid;
    int t0 = wg * share;
    const int t1 = min(total, t0 + share);
    if (t0 >= t1) return;
    stage(wl, p.att, 3 * AB::size);
    if (p.out_in_lds) stage(outl, p.out, p.out_size);
    const EncBlob E = p.encb;
    while (t0 < t1) {
        const int g = p.tile_graph[t0];
        const int seg_end = min(t1, p.ptr_pad_total[g + 1] / 32);
        __syncthreads();                                       // previous graph's K/V no longer in use
        stage(kvl, p.kv + (size_t)g * 3 * p.kv_stride, 3 * p.kv_stride);
        if (threadIdx.x == 0) *ctr = t0;
        __syncthreads();
        const int nbase_pad = p.node_ptr_pad[g], nbase = p.node_ptr[g];
        const int O = min(p.obs_ptr[g + 1] - p.obs_ptr[g], p.ot_max * 32);
        while (true) {
            int tile = 0;
            if (lane == 0) tile = atomicAdd(ctr, 1);
            tile = __builtin_amdgcn_readfirstlane(tile);
            if (tile >= seg_end) break;
            if (EDGE && p.tile_meta[tile] < 4) continue;       // tile of pure padding
            const int row = tile * 32 + j;
            // encoder / epilogue weights are MFMA operands read from global memory; laundering the pointers per tile
            // keeps the compiler from hoisting those loop-invariant loads out of the tile loop into dozens of registers
            const float* enc_w = p.enc;
            const float* out_w = p.out_in_lds ? outl : p.out;
            if constexpr (D > 32 || !EDGE) asm volatile("" : "+s"(enc_w), "+s"(out_w));    // d = 32 edge kernel: registers to spare, hoisting pays
            f32x16 m[NT], aux[NT];
            if constexpr (EDGE) {
                const int4 rec = p.csr[row];
                const int s = rec.x, t = rec.y;
                if (h == 0 && s >= 0) p.rec32[row] = (s - nbase_pad) | (((t - nbase_pad) & 31) << 27);    // see pre_kernel
                const float* vs = p.v + (size_t)(nbase + (s >= 0 ? s - nbase_pad : 0)) * C;
                const float* vt = p.v + (size_t)(nbase + (t >= 0 ? t - nbase_pad : 0)) * C;
                // [v_src, v_dst]; branch-free: one clamped load + select (a divergent lambda would cost far more
                // VALU/SALU issue time than the handful of MFMAs it feeds)
                auto getin = [&](int k) {
                    const bool ok = k < 2 * C;
                    const float* base = (k < C) ? vs : vt - C;
                    const float x = base[ok ? k : C];
                    return ok ? x : 0.f;
                };
                // d = 64: edge_code (10 MFMAs from the raw coordinates) is recomputed right before its only use in the
                // epilogue instead of living in 32 registers through the three attention blocks
                if constexpr (!kRecomputeAux)
                    mlp2_in<NT, P>(enc_w + E.as0, E.ks0, enc_w + E.b0, enc_w + E.a0, enc_w + E.c0, getin, aux, lane);
                mlp2_in<NT, P>(enc_w + E.as1, E.ks1, enc_w + E.b1, enc_w + E.a1, enc_w + E.c1, getin, m, lane);
            } else {
                const int local = row - nbase_pad;
                const int ng = p.node_ptr[g + 1] - nbase;
                const float* vr = p.v + (size_t)(nbase + (local < ng ? local : 0)) * C;
                const float* gl = p.goal + (size_t)g * C;
                auto getin_nc = [&](int k) {                        // [v, goal, (v-goal)^2, v-goal]  model.py:119, branch-free
                    const bool ok = k < 4 * C;
                    const int kk = ok ? k : 0;
                    const int part = kk / C, c = kk - part * C;
                    const float x = vr[c], gg = gl[c];
                    const float dlt = x - gg;
                    const float val = part == 0 ? x : (part == 1 ? gg : (part == 2 ? dlt * dlt : dlt));
                    return ok ? val : 0.f;
                };
                auto getin_nf = [&](int k) { const bool ok = k < C; const float x = vr[ok ? k : 0]; return ok ? x : 0.f; };
                mlp2_in<NT, P>(enc_w + E.as0, E.ks0, enc_w + E.b0, enc_w + E.a0, enc_w + E.c0, getin_nc, aux, lane);
                if (p.m0) load_row<NT>(p.m0 + (size_t)row * D, m, h);                              // node_f64_body's rows
                else mlp2_in<NT, P>(enc_w + E.as1, E.ks1, enc_w + E.b1, enc_w + E.a1, enc_w + E.c1, getin_nf, m, lane);
            }
            for (int b = 0; b < 3; ++b)
                attention_block<D, P, !EDGE>(wl + b * AB::size, wl + b * AB::size, kvl + (size_t)b * p.kv_stride, nullptr,
                                   O, p.ot_max, p.ot_max, m, lane, !EDGE && b == 0 && p.m0 != nullptr);
            if (p.om) store_row<NT>(p.om + (size_t)row * D, m, h);       // training path: frozen node_/edge_free_code
            if constexpr (EDGE) {
                using L = OutEBlob<D, P>;
                f32x16 y[NT];
                BOp<P> mop[NT];
                make_ops<P, NT>(m, mop);
                // aux = W1e.edge_code + b1 (the product is folded into the encoder's second layer at pack time)
#pragma unroll
                for (int t = 0; t < NT; ++t) y[t] = kRecomputeAux ? splat16(0.f) : aux[t];
                linear_acc_ops<P, NT, NT>(out_w + L::w1d, mop, y, lane);
                if constexpr (kRecomputeAux) {
                    const int4 rec = p.csr[row];
                    const int s = rec.x, t = rec.y;
                    const float* vs = p.v + (size_t)(nbase + (s >= 0 ? s - nbase_pad : 0)) * C;
                    const float* vt = p.v + (size_t)(nbase + (t >= 0 ? t - nbase_pad : 0)) * C;
                    auto getin = [&](int k) {
                        const bool ok = k < 2 * C;
                        const float* base = (k < C) ? vs : vt - C;
                        const float x = base[ok ? k : C];
                        return ok ? x : 0.f;
                    };
                    mlp2_in<NT, P>(enc_w + E.as0, E.ks0, enc_w + E.b0, enc_w + E.a0, enc_w + E.c0, getin, aux, lane);
#pragma unroll
                    for (int t = 0; t < NT; ++t) y[t] += aux[t];
                }
                store_tile_p<P, NT>(p.o0 + (size_t)tile * NT * kETile, y, lane);
                load_vec<NT>(out_w + L::bp0, y, lane);
                linear_acc_ops<P, NT, NT>(out_w + L::wpc, mop, y, lane);
                store_tile_p<P, NT>(p.o1 + (size_t)tile * NT * kETile, y, lane);
            } else {
                using L = OutNBlob<D, P>;
                const bool isgoal = (row == p.goal_node[g]);
                f32x16 xi[NT], tmp[NT], y[NT];
                load_vec<NT>(out_w + L::be, xi, lane);
                linear_acc_p<P, NT, NT>(out_w + L::we_nc, aux, xi, lane);
                linear_acc_p<P, NT, NT>(out_w + L::we_nf, m, xi, lane);
                load_vec<NT>(out_w + L::weg, tmp, lane);
#pragma unroll
                for (int t = 0; t < NT; ++t) xi[t] += isgoal ? tmp[t] : splat16(0.f);
                store_row<NT>(p.o0 + (size_t)row * D, xi, h);
                load_vec<NT>(out_w + L::wehg, tmp, lane);
#pragma unroll
                for (int t = 0; t < NT; ++t) xi[t] += isgoal ? tmp[t] : splat16(0.f);
                store_row_p<P == 1 ? 1 : 0, NT>(p.o1, (size_t)row, xi, h);
#pragma unroll
                for (int t = 0; t < NT; ++t) y[t] = splat16(0.f);
                linear_acc_p<P, NT, NT>(out_w + L::wsrc, xi, y, lane);
                store_row_p<P, NT>(p.o2, (size_t)row, y, h);
#pragma unroll
                for (int t = 0; t < NT; ++t) y[t] = splat16(0.f);
                linear_acc_p<P, NT, NT>(out_w + L::wdst, xi, y, lane);
                store_row_p<P, NT>(p.o3, (size_t)row, y, h);
                load_vec<NT>(out_w + L::bd, y, lane);
                linear_acc_p<P, NT, NT>(out_w + L::wd_nc, aux, y, lane);
                store_row<NT>(p.o4 + (size_t)row * D, y, h);
            }
        }
        t0 = seg_end;
    }
}

template <int D, int P, bool EDGE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void pre_resident_kernel(PreParams p) {
    pre_resident_body<D, P, EDGE, WAVES>(p, (int)blockIdx.x, (int)gridDim.x);
}

// Small batches: the node and the edge pre stages are independent (both need the CSR and the K/V slabs only) and
// neither fills the device, so ONE launch runs them side by side: workgroups [0, node_blocks) take the node role,
// the rest the edge role.  For one 1000-node graph each stage is a ~30 us dependent chain (a wave's 32-row tile is
// ~700 MFMAs in a row) and the two launches used to run back to back.
template <int D, int P, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void pre_resident_both_kernel(PreParams pn, PreParams pe, int node_blocks) {
    if ((int)blockIdx.x < node_blocks) pre_resident_body<D, P, false, WAVES>(pn, (int)blockIdx.x, node_blocks);
    else pre_resident_body<D, P, true, WAVES>(pe, (int)blockIdx.x - node_blocks, (int)gridDim.x - node_blocks);
}

// Grid-stride over tile groups with an XCD-aware order: workgroup b runs on XCD b % 8 (observed
// dispatch rule; used for speed only), so XCD x walks the contiguous eighth [x*per, (x+1)*per) of the
// group space.  Consecutive groups belong to the same graphs, hence each XCD's private L2 only ever
// holds the node rows of "its" graphs instead of every XCD caching every graph.
struct XcdWalk {
    int cur, end, step;
    // order 2 (mp_fused, persistent workgroups): the XCD's 256-row blocks [blk0, blk0 + B) stand for 2 B four-tile groups -- the
    // i-th group from the START of its graph (heavy: the free samples come first) for positions 0 .. B-1, the i-th group from
    // its END (light) for positions B .. 2B-1 in reverse block order.  Workgroup u of U walks the positions on a snake (u,
    // 2U-1-u, 2U+u, 4U-1-u, ...): longest jobs first, every resident workgroup gets a near-equal share of heavy and light groups.
    int snake = 0, k = 0, U = 1, u = 0, blk0 = 0, B = 0, ns = 0;
    const int2* span = nullptr;
    int2 nspan;
    __device__ __forceinline__ int pos(int kk) const { return (kk & 1) ? (kk + 1) * U - 1 - u : kk * U + u; }
    __device__ __forceinline__ int blk_of(int s) const { return blk0 + (s < B ? s : 2 * B - 1 - s); }
    __device__ __forceinline__ void snake_prefetch() {            // the NEXT position's block span, requested a whole group ahead
        ++k;
        ns = pos(k);
        if (ns < 2 * B) nspan = span[blk_of(ns)];
    }
    __device__ __forceinline__ void snake_set(int s, int2 sp) {
        if (s < 2 * B) {
            const int i = blk_of(s) - (sp.x >> 1);
            cur = s < B ? sp.x + i : sp.y - 1 - i;
            end = cur + 1;
        } else {
            end = cur;
        }
    }
    __device__ __forceinline__ XcdWalk(int n_groups) {
        const int nb = gridDim.x >> 3;                    // blocks per XCD (gridDim.x is a multiple of 8)
        const int per = (n_groups + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        cur = xcd * per + (blockIdx.x >> 3);
        end = min(n_groups, (xcd + 1) * per);
        step = nb;
    }
    // the same walk for a VIRTUAL workgroup: index q among the nb (virtual) workgroups of XCD xcd (mp_fused_w8: two four-wave halves
    // of one real workgroup)
    __device__ __forceinline__ XcdWalk(int n_groups, int xcd, int q, int nb) {
        const int per = (n_groups + 7) >> 3;
        cur = xcd * per + q;
        end = min(n_groups, (xcd + 1) * per);
        step = nb;
    }
    __device__ __forceinline__ bool valid() const { return cur < end; }
    __device__ __forceinline__ void next() {
        if (snake) snake_set(ns, nspan);
        else cur += step;
    }
};


// =====================================================================================================
// mp_fused: one message-passing iteration (model.py:139-143 for one loop index) in ONE launch.  A job = one 32-node
// tile (padded node space) handled by one wave:
//   edge phase   the tile's incoming edges are one contiguous CSR range [row_beg[first], row_beg[last] + deg[last]);
//                per 32-edge chunk: hidden = relu(A[src] + B[dst] + K_e), M = W2 hidden + b2 with the MFMA operands
//                SWAPPED (chain.hpp): the accumulator comes out as [lane = feature][16 registers = edges], i.e. the
//                reduction over edges runs over registers and the 32 lanes of a half wave are 32 consecutive
//                features.  Max aggregation = one LDS float atomic (ds_max_f32) per (edge, feature) into the wave's
//                private [32 nodes][D] tile: order-free, exact, conflict-free (lanes hit consecutive banks), no
//                transpose, no segment walk, no partial maxima across tile boundaries, no agg round trip to HBM.
//   node phase   agg tile (0 for nodes without incoming edges: torch_scatter) -> H = Wlx X + Wla agg + bl,
//                Y = R + M1 H, A' = M2 Y, B' = M3 Y  (weights read as MFMA operands from L1/L2).
// A' goes to the OTHER A buffer: other jobs still gather this iteration's A rows.  B' belongs to the tile itself (the
// target's term of ITS incoming edges), so the one-tile-per-wave form does not store it: the next iteration's job
// recomputes it from the X = Y rows it reads anyway (same 16 NT^2 MFMAs, moved from the end of one launch to the start
// of the next; the last iteration stores PT for the policy head there instead).
// =====================================================================================================
// one 32-feature tile `t` of edge slot `slot` (per-edge tiles are stored tile-native, [slot / 32][NT][...][64 lanes][...],
// chain.hpp store_tile_p; a chunk need not start on a tile boundary, so every lane addresses its own slot)
#define GNNMP_KE_LOAD(p) __builtin_nontemporal_load(p)
template <int P, int NT>
__device__ __forceinline__ void load_edge_slot_tile(const float* base_f32_units, int slot, int h, int t, f32x16& x) {
    const size_t tile = (size_t)(slot >> 5);
    const int ln = (slot & 31) + 32 * h;
    if constexpr (P != 1) {
        const float* b = base_f32_units + tile * NT * 1024 + ln * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = GNNMP_KE_LOAD(reinterpret_cast<const f32x4*>(b + (t * 4 + q) * 256));
#pragma unroll
            for (int c = 0; c < 4; ++c) x[q * 4 + c] = a[c];
        }
    } else {
        const __bf16* b = reinterpret_cast<const __bf16*>(base_f32_units) + tile * NT * 1024 + ln * 8;
        const bf16x8 lo = GNNMP_KE_LOAD(reinterpret_cast<const bf16x8*>(b + (t * 2 + 0) * 512));
        const bf16x8 hi = GNNMP_KE_LOAD(reinterpret_cast<const bf16x8*>(b + (t * 2 + 1) * 512));
        const f32x8 a = __builtin_convertvector(lo, f32x8), c = __builtin_convertvector(hi, f32x8);
#pragma unroll
        for (int r = 0; r < 8; ++r) { x[r] = a[r]; x[8 + r] = c[r]; }
    }
}

// the same tile kept as loaded (bf16 tiles stay packed: 8 registers instead of 16 while they wait in the prefetch queue)
template <int P> struct KeRaw { f32x16 v; };
template <> struct KeRaw<1> { bf16x8 lo, hi; };
template <int P, int NT>
__device__ __forceinline__ void load_edge_slot_raw(const float* base_f32_units, int slot, int h, int t, KeRaw<P>& r) {
    if constexpr (P != 1) {
        load_edge_slot_tile<P, NT>(base_f32_units, slot, h, t, r.v);
    } else {
        const size_t tile = (size_t)(slot >> 5);
        const int ln = (slot & 31) + 32 * h;
        const __bf16* b = reinterpret_cast<const __bf16*>(base_f32_units) + tile * NT * 1024 + ln * 8;
        r.lo = GNNMP_KE_LOAD(reinterpret_cast<const bf16x8*>(b + (t * 2 + 0) * 512));
        r.hi = GNNMP_KE_LOAD(reinterpret_cast<const bf16x8*>(b + (t * 2 + 1) * 512));
    }
}
template <int P>
__device__ __forceinline__ void expand_raw(const KeRaw<P>& r, f32x16& x) {
    if constexpr (P != 1) {
        x = r.v;
    } else {
        const f32x8 a = __builtin_convertvector(r.lo, f32x8), c = __builtin_convertvector(r.hi, f32x8);
#pragma unroll
        for (int q = 0; q < 8; ++q) { x[q] = a[q]; x[8 + q] = c[q]; }
    }
}

// one 32-feature tile `t` of row `row` of a [rows, D] array stored in the precision-dependent format of load_row_p
template <int P, int NT>
__device__ __forceinline__ void load_row_tile(const float* array_f32_units, size_t row, int h, int t, f32x16& x) {
    if constexpr (P != 1) {
        const float* base = array_f32_units + row * (NT * 32) + t * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(base + q * 8 + h * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) x[q * 4 + c] = a[c];
        }
    } else {
        // bf16 rows are stored HALF-MAJOR inside every 32-feature tile (chain.hpp store_row_p): the 16 features lane half h owns are
        // 32 contiguous bytes -- two 16-byte loads instead of four 8-byte ones
        const __bf16* base = reinterpret_cast<const __bf16*>(array_f32_units) + row * (NT * 32) + t * 32 + h * 16;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const f32x8 a = __builtin_convertvector(*reinterpret_cast<const bf16x8*>(base + qq * 8), f32x8);
#pragma unroll
            for (int c = 0; c < 8; ++c) x[qq * 8 + c] = a[c];
        }
    }
}

typedef __attribute__((address_space(3))) float lds_float;
typedef int i32x4 __attribute__((ext_vector_type(4)));
// s_waitcnt vmcnt(N) as the BUILTIN, not as inline asm: the compiler's own wait-count bookkeeping reads S_WAITCNT instructions
// that are already there, so after this one it knows that everything but the last N vector-memory operations has completed and
// inserts no (coarser) wait of its own for registers or LDS-DMA destinations filled by the older ones.  An inline-asm wait is
// invisible to it: it then guarded the first use of every prefetched register with s_waitcnt vmcnt(0), which also drained
// the requests issued a moment earlier.  (gfx9 encoding: vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14.)
__device__ __forceinline__ void wait_all() {            // s_waitcnt vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_waitcnt(7 << 4);
    asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}
// byte address of an LDS location as the DS instructions take it
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// ---------------------------------------------------------------------------------------------------------------------
// Row gathers through LDS.  A wave-wide global load in the chain layout (lane = row, 16 bytes per lane per instruction)
// costs ONE L1 access PER LANE: neighbouring lanes hold different rows, nothing coalesces, and the message-passing
// kernels were bound by exactly that (rocprofv3: ~1 TCP access per cycle and CU, 563 accesses per 32-edge chunk at
// d = 32 fp32, 955 at d = 64 bf16; TA busy 52-59 %).  Here 32 rows are fetched with the LDS-DMA path
// (global_load_lds_dwordx4): lane l asks for piece (l % PP) of row (l / PP) of its instruction, so the four lanes of
// a quad read 64 contiguous bytes = one access, 4 x fewer than before, and no VGPR is tied up while the rows are in
// flight.  The DMA writes LDS linearly (wave-uniform base + lane * 16), so the bank-conflict swizzle is applied to the
// SOURCE piece index and again, as the same involution, when the chain layout reads the rows back (16-byte / 8-byte
// LDS reads).
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int P>
struct RowGeom {
    static constexpr int RB = D * (P == 1 ? 2 : 4);         // bytes per row (64, 128 or 256)
    static constexpr int PP = RB / 16;                       // 16-byte pieces per row
    static constexpr int RPI = 64 / PP;                      // rows per DMA instruction
    static constexpr int NI = 32 / RPI;                      // DMA instructions per 32 rows
    static constexpr int SH = RB == 64 ? 2 : (RB == 128 ? 1 : 0);     // rows 2^SH apart share their banks (256-byte period)
    static constexpr int STAGE_FLOATS = 32 * RB / 4;
    __device__ static __forceinline__ int swz(int row) { return (row >> SH) & (PP - 1); }
};

// stage <- 32 rows; `row_id(sr)` = array row of stage row sr as seen by THIS lane (the caller shuffles if needed)
// AUX: cache policy bits of the instruction (2 = nt: rows that are read once per launch should not displace the gathered A rows in L2)
template <int D, int P, int AUX = 0, class RowId>
__device__ __forceinline__ void dma_rows(const float* array_f32_units, RowId row_id, float* stage, int lane) {
    using G = RowGeom<D, P>;
    const char* base = reinterpret_cast<const char*>(array_f32_units);
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
        const int sr = G::RPI * i + lane / G::PP;
        const int pc = (lane % G::PP) ^ G::swz(sr);
        const char* g = base + (size_t)row_id(sr) * G::RB + pc * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(stage + i * 256), 16, 0, AUX);
    }
}

// 32-feature tile `t` of stage row `sr` in the chain layout (lane half h)
template <int D, int P>
__device__ __forceinline__ void read_stage_tile(const float* stage, int sr, int h, int t, f32x16& x) {
    using G = RowGeom<D, P>;
    const char* row = reinterpret_cast<const char*>(stage) + sr * G::RB;
    const int sw = G::swz(sr);
    if constexpr (P != 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(row + (((t * 8 + q * 2 + h) ^ sw) * 16));
#pragma unroll
            for (int c = 0; c < 4; ++c) x[q * 4 + c] = a[c];
        }
    } else {
        // half-major bf16 rows: lane half h owns pieces 4 t + 2 h and 4 t + 2 h + 1 of the row -- two ds_read_b128, the access pattern
        // of the fp32 rows of the same byte pitch (conflict-free under this swizzle), instead of four ds_read_b64 whose 32 lanes of a
        // half wave all sat on the low 8 bytes of their 16-byte slot (32 of the 64 banks: a two-way conflict on every read)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const f32x8 a = __builtin_convertvector(*reinterpret_cast<const bf16x8*>(row + (((t * 4 + h * 2 + qq) ^ sw) * 16)), f32x8);
#pragma unroll
            for (int c = 0; c < 8; ++c) x[qq * 8 + c] = a[c];
        }
    }
}

// the same 32-feature tile kept as read (bf16 rows stay packed: 8 registers instead of 16) -- the message kernel empties the A
// stage into registers at the top of a chunk so that the NEXT chunk's rows can be requested a whole chunk ahead
template <int P> struct StageRaw { f32x4 q[4]; };
template <> struct StageRaw<1> { bf16x8 lo, hi; };          // registers 0-7 / 8-15 of the tile, packed (the two pieces of the lane's half)
template <int D, int P>
__device__ __forceinline__ void read_stage_raw(const float* stage, int sr, int h, int t, StageRaw<P>& r) {
    using G = RowGeom<D, P>;
    const char* row = reinterpret_cast<const char*>(stage) + sr * G::RB;
    const int sw = G::swz(sr);
    if constexpr (P != 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) r.q[q] = *reinterpret_cast<const f32x4*>(row + (((t * 8 + q * 2 + h) ^ sw) * 16));
    } else {
        r.lo = *reinterpret_cast<const bf16x8*>(row + (((t * 4 + h * 2 + 0) ^ sw) * 16));
        r.hi = *reinterpret_cast<const bf16x8*>(row + (((t * 4 + h * 2 + 1) ^ sw) * 16));
    }
}
template <int P>
__device__ __forceinline__ void expand_stage_raw(const StageRaw<P>& r, f32x16& x) {
    if constexpr (P != 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) x[q * 4 + c] = r.q[q][c];
    } else {
        const f32x8 a = __builtin_convertvector(r.lo, f32x8), b = __builtin_convertvector(r.hi, f32x8);
#pragma unroll
        for (int c = 0; c < 8; ++c) { x[c] = a[c]; x[8 + c] = b[c]; }
    }
}

// the inverse of read_stage_tile for a whole row: the lane's NT tiles of stage row sr (its half h) in the row format of
// precision P (fp32 / bf16x3: fp32 rows; bf16: rows rounded to bf16 like store_row_p)
template <int D, int P, int NT>
__device__ __forceinline__ void write_stage_tiles(float* stage, int sr, int h, const f32x16 (&x)[NT]) {
    using G = RowGeom<D, P>;
    char* row = reinterpret_cast<char*>(stage) + sr * G::RB;
    const int sw = G::swz(sr);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if constexpr (P != 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a = {x[t][q * 4 + 0], x[t][q * 4 + 1], x[t][q * 4 + 2], x[t][q * 4 + 3]};
                *reinterpret_cast<f32x4*>(row + (((t * 8 + q * 2 + h) ^ sw) * 16)) = a;
            }
        } else {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {          // half-major bf16 rows: registers 8 qq .. 8 qq + 7 are piece 4 t + 2 h + qq
                f32x8 a;
#pragma unroll
                for (int c = 0; c < 8; ++c) a[c] = x[t][qq * 8 + c];
                *reinterpret_cast<bf16x8*>(row + (((t * 4 + h * 2 + qq) ^ sw) * 16)) = to_bf16x8(a);
            }
        }
    }
}

// 32 staged rows -> 32 CONSECUTIVE rows of a [rows, D] array, the inverse of dma_rows for a contiguous block: lane l stores LDS piece
// (l % PP) of stage row (l / PP) of its instruction, i.e. PP lanes write one whole row (64 .. 256 contiguous bytes) instead of every
// lane writing 8 / 16 bytes of a different row (a store in the chain layout is one L2 write request per lane)
template <int D, int P, bool NT_STORE = false>
__device__ __forceinline__ void store_rows_coalesced(float* array_f32_units, size_t row0, const float* stage, int lane) {
    using G = RowGeom<D, P>;
    char* base = reinterpret_cast<char*>(array_f32_units) + row0 * G::RB;
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
        const int sr = G::RPI * i + lane / G::PP;
        const int lp = lane % G::PP;                              // LDS piece lp of a row holds the row's piece lp ^ swz (dma_rows)
        const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(stage) + sr * G::RB + lp * 16);
        f32x4* dst = reinterpret_cast<f32x4*>(base + (size_t)sr * G::RB + ((lp ^ G::swz(sr)) * 16));
        if constexpr (NT_STORE) __builtin_nontemporal_store(v, dst);
        else *dst = v;
    }
}

// y[ot] += sum_it W[ot][it] . x[it] with the inputs produced one 32-feature tile at a time by `get(it, tile)`: only one
// input tile is live, which is what lets the d = 64 instantiation keep its register count down
// RELU: the input tile goes through max(x, 0) before it is multiplied.  bf16 operands: the conversion comes first and the
// ReLU runs on the PACKED pairs as a signed 16-bit integer maximum with 0 (negative floats are negative integers, -0 is the
// most negative one; rounding to bf16 is monotone and keeps the sign, so this equals rounding the ReLU's output bit for bit)
// -- 8 instead of 16 VALU instructions per tile in a loop that is bound by instruction issue.
typedef short s16x8 __attribute__((ext_vector_type(8)));
template <int P, int NT, bool SWAP, bool RELU = false, class Get>
__device__ __forceinline__ void linear_acc_stream(const float* A, Get get, f32x16 (&y)[NT], int lane) {
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        f32x16 x;
        get(it, x);
        if constexpr (RELU && P != 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = fmaxf(x[r], 0.0f);
        }
        BOp<P> xb(x);
        if constexpr (RELU && P == 1) {
            const s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            xb.lo = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, xb.lo), zero));
            xb.hi = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, xb.hi), zero));
        }
#pragma unroll
        for (int ot = 0; ot < NT; ++ot) mfma_tile_p<P, SWAP>(A + (ot * NT + it) * Prec<P>::TF, xb, y[ot], lane);
    }
}


// the chunk loop's request / wait structure, per operand precision (experiments: -DGNNMP_MP_FLOW_BF16=0 / -DGNNMP_MP_FLOW_F32=1).
// "New flow" = (a) the A stage is emptied into registers at the top of a chunk and the next chunk's rows are requested there, a whole
// chunk ahead; (b) requests inside the loop are unconditional; (c) the prefetched registers' first uses are pinned behind the
// explicit wait; (d) the target offsets and the aggregation atomics are issued by hand.  It is what lets the prefetch run more than
// one request deep (see the chunk body); measured: bf16 d = 64 five launches 0.556 -> 0.545 ms, bf16 d = 32 0.548 -> 0.512, but
// fp32 d = 32 0.704 -> 0.752 (its sixteen 64-cycle MFMAs per chunk cover the latency anyway and the extra registers cost), so
// the exact-fp32 kernels keep the round-3 flow.
#ifndef GNNMP_MP_FLOW_BF16
#define GNNMP_MP_FLOW_BF16 1
#endif
#ifndef GNNMP_MP_FLOW_F32
#define GNNMP_MP_FLOW_F32 0
#endif
#ifndef GNNMP_MP_ROWS_LDS
#define GNNMP_MP_ROWS_LDS 1          // bf16 kernels, node phase: R rows in through LDS-DMA, X' / A' / PT rows out through LDS as whole rows
#endif

// LDS floats of mp_fused besides the staged MpEBlob.  COOP = 1: every wave owns a max-aggregation tile [32][D], 32 row
// offsets and two 32-row gather stages; COOP > 1: the workgroup's COOP waves share the aggregation tile and the B-row
// stage, every wave keeps its own row offsets and A-row stage
// Row pitch (floats) of the [32][D] fp32 aggregation tile.  The tile is written by ds_max_f32 with lane = feature (32 consecutive
// floats of one row per half wave: conflict-free at any pitch) and filled / read back with lane = ROW, 16 bytes per lane: at a pitch
// of D floats (128 / 256 bytes) the 16 lanes of a ds_read_b128 group -- 8 of a ds_write_b128 group -- all sit on the same one or two
// 16-byte slots of the 256-byte bank row, an 8- to 16-way conflict on every fill and read-back (SQ_LDS_BANK_CONFLICT was 0.40-0.53 of
// the LDS-active cycles of mp_fused at every shape, round 5).  One extra slot per row rotates the rows across all sixteen slots.
template <int D> constexpr int kAggPitch = D + 4;
template <int D, int P, int COOP>
__host__ __device__ constexpr int mp_lds_floats() {
    constexpr int stage_f = 32 * D * (P == 1 ? 2 : 4) / 4;
    constexpr int AGG = 32 * kAggPitch<D>;
    // COOP > 1 adds four [32][D] fp32 tiles: X and R rows of the job (requested at its start), the partial H = bl + Wlx X that a
    // wave computes while the others multiply edges, and Y for the wave that computes B'
    return COOP == 1 ? 4 * (AGG + 32 + 2 * stage_f) : (AGG + stage_f + COOP * (32 + stage_f) + (D == 32 ? 4 * 32 * D : 0));
}

// COOP = 1 (large batches): one 32-node tile per WAVE, four independent waves per workgroup.
// COOP > 1 (few tiles -- single graphs, small batches): one tile per WORKGROUP of COOP waves; the tile's 32-edge chunks
// go round-robin to the waves, which all aggregate into the shared LDS tile (float atomics: order-free, exact), and
// wave 0 runs the node phase.  A single 1000-node graph has 32 tiles of ~12 chunks: 25 us per launch with one wave
// per tile, ~9 us with eight.
// X', A', B' / PT rows leave with non-temporal stores: nothing reads them again in this launch, and they would displace the gathered A
// rows in L2 (configs[1] shape: mp 0.667 -> 0.656 ms; neutral at the bf16 shapes)
constexpr bool kNtW4 = true;
template <int D, int P, int COOP>
// d = 64 with fp32 / bf16x3 operands: the LDS tiles leave ONE 4-wave workgroup per CU anyway, so the wave may use the whole
// register file (no spills)
#ifndef GNNMP_MP_WGS32
#define GNNMP_MP_WGS32 2      // the LDS tiles leave two workgroups per CU at d = 32: let the wave use the registers of two
#endif
#ifndef GNNMP_MP_WGS32B
#define GNNMP_MP_WGS32B 2     // d = 32, bf16 operands: three workgroups per CU fit the LDS, but at 168 registers the kernel spills (0.54 vs 0.51 ms at the configs[4] shape)
#endif
__global__ __launch_bounds__(COOP == 1 ? 256 : COOP * 64, COOP == 1 ? ((D > 32 && P != 1) ? 1 : ((P == 2 || D > 32) ? 2 : (P == 1 ? GNNMP_MP_WGS32B : GNNMP_MP_WGS32))) : 1) void mp_fused_kernel(MpFusedParams p) {
    constexpr int NT = D / 32;
    constexpr bool kCoop = COOP > 1;
    // few tiles, d = 32: the node phase is spread over waves (below); at d = 64 the eight-wave workgroup has 256 registers per
    // wave and the extra live tiles spill (measured: kuka7 bf16 single graph 135 -> 156 us), so it keeps the one-wave form
    constexpr bool kSplitNode = kCoop && D == 32;
    constexpr bool kNewFlow = P == 1 ? (GNNMP_MP_FLOW_BF16 != 0) : (GNNMP_MP_FLOW_F32 != 0);
    constexpr bool kAsmTail = kNewFlow, kAEarly = kNewFlow, kUncond = kNewFlow, kLaunder = kNewFlow;
    using LE = MpEBlob<D, P>;
    using LN = MpNBlob<D, P>;
    using G = RowGeom<D, P>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;                                             // MpEBlob
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
#ifdef GNNMP_MP_TRACE
    // diagnostics build: [0] wave start, then per tile: start, end of the edge phase, end of the node phase (100 MHz clock)
    // 32 slots per wave: [0] start, [1 + 5 k .. 5 + 5 k] tile k < 5 (start, edge end, H, Y, node end), [28] HW_ID, [29] / [30] the
    // first two tile ids, [31] wave end
    long long* trc = p.trace ? p.trace + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 32 : nullptr;
    int trc_n = 0, trc_tiles = 0;
#define GNNMP_TRC() do { if (trc && lane == 0 && trc_n < 26) trc[trc_n] = wall_clock64(); ++trc_n; } while (0)
#define GNNMP_TRC_TILE(t) do { if (trc && lane == 0 && trc_tiles < 2) trc[29 + trc_tiles] = (t); ++trc_tiles; } while (0)
#define GNNMP_TRC_END() do { if (trc && lane == 0) trc[31] = wall_clock64(); } while (0)
    if (trc && lane == 0) {
        for (int i = 0; i < 32; ++i) trc[i] = 0;
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trc[28] = (long long)hw | ((long long)xcc << 32);
    }
    GNNMP_TRC();
#else
#define GNNMP_TRC() do {} while (0)
#define GNNMP_TRC_TILE(t) do {} while (0)
#define GNNMP_TRC_END() do {} while (0)
#endif
    float* base = lds + ((LE::size + 3) & ~3);
    constexpr int AP = kAggPitch<D>, AGG = 32 * AP;
    float* agg = kCoop ? base : base + wave * (AGG + 32 + 2 * G::STAGE_FLOATS);                    // [32][D] at pitch AP
    float* btile = kCoop ? base + AGG : agg + AGG + 32 + G::STAGE_FLOATS;                      // B rows of the tile
    float* mine = kCoop ? base + AGG + G::STAGE_FLOATS + wave * (32 + G::STAGE_FLOATS) : agg + AGG;
    int* dl = reinterpret_cast<int*>(mine);                      // [32] agg row offsets (floats) of this chunk's targets
    float* astage = mine + 32;                                   // gathered A rows of the current chunk
    float* xstage = base + AGG + G::STAGE_FLOATS + COOP * (32 + G::STAGE_FLOATS);     // kCoop only (see mp_lds_floats)
    float* rstage = xstage + 32 * D;
    float* hpart = rstage + 32 * D;
    float* ytile = hpart + 32 * D;
    if constexpr (!kCoop) {
        // workgroups of the unused tail of the padded tile space (up to 22 % of the grid) leave before they stage anything
        if (p.tpw > 0 && p.order != 2) {
            const int q8 = blockIdx.x >> 3, per = ((((p.node_ptr_pad[p.G] >> 7) + p.tpw - 1) / p.tpw) + 7) >> 3;
            if (q8 >= per) return;
        }
    }
    // few tiles: a workgroup has ONE tile; its row range (and graph) are requested before the weights are staged, so that
    // round trip runs under the staging instead of behind it (a launch here is a handful of dependent round trips)
    int pre_rb = 0, pre_dg = 0, pre_g = -1;
    if constexpr (kCoop) {
        XcdWalk wk0(min(p.n_tiles, p.node_ptr_pad[p.G] >> 5));
        if (wk0.valid() && wk0.cur < p.n_tiles) {
            pre_g = p.ntile_graph[wk0.cur];
            pre_rb = p.row_beg[wk0.cur * 32 + j];
            pre_dg = p.deg[wk0.cur * 32 + j];
        }
    }
    stage(wl, p.we, LE::size);
    // d = 32: the node phase's weights (MpNBlob, 20 KB) fit next to the tiles -- staged once per workgroup instead of read
    // from L1 / L2 as MFMA operands by every tile (five launches at cfg 2: 0.910 -> 0.872 ms)
#ifndef GNNMP_MP_NODEW_LDS
#define GNNMP_MP_NODEW_LDS 1
#endif
    constexpr bool kNodeWInLds = GNNMP_MP_NODEW_LDS && D == 32 && P != 2;          // bf16x3: 31 KB, would leave one workgroup per CU
    float* wnl = lds + ((LE::size + 3) & ~3) + mp_lds_floats<D, P, COOP>();
    if constexpr (kNodeWInLds) stage(wnl, p.wn, LN::size);
    __syncthreads();
    // b2 of the lane's feature(s): packed vectors are in register order, vec[(t*2 + h')*16 + r'] = b[32 t + phi(r', h')]
    float bias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias[t] = wl[LE::b2 + (t * 2 + ((j >> 2) & 1)) * 16 + (j & 3) + 4 * (j >> 3)];
    // tiles actually in use (the prep stage's padded total): the XCD eighths are cut from those, not from the launch's upper bound
    const int real_tiles = min(p.n_tiles, p.node_ptr_pad[p.G] >> 5);
    XcdWalk wk(kCoop ? real_tiles : (real_tiles + 3) / 4);
    if (!kCoop && p.order == 2) {
        const int blocks = p.node_ptr_pad[p.G] >> 8, perb = (blocks + 7) >> 3;
        wk.snake = 1;
        wk.U = gridDim.x >> 3;
        wk.u = blockIdx.x >> 3;
        wk.blk0 = (blockIdx.x & 7) * perb;
        wk.B = max(0, min(perb, blocks - wk.blk0));
        wk.span = p.blk_span;
        wk.k = 0;
        const int s0 = wk.pos(0);
        int2 sp0 = make_int2(0, 0);
        if (s0 < 2 * wk.B) sp0 = p.blk_span[wk.blk_of(s0)];
        wk.cur = 0;
        wk.snake_set(s0, sp0);
        wk.ns = 2 * wk.B;                       // (until the first prefetch)
    } else
    if (!kCoop && p.tpw > 0) {                 // tpw == 0 (experiment): persistent workgroups, strided walk of the XCD's eighth
        // every workgroup takes p.tpw ADJACENT four-tile groups, one after the other (the launcher picks tpw so that the
        // workgroups fill whole rounds of the resident slots: launch_mp_fused_t)
        // XCD-aware like XcdWalk -- workgroup b runs on XCD b % 8, which works through a contiguous eighth of the group space --
        // but the eighths are cut from the groups IN USE (node_ptr_pad[G], written by the prep stage): the padded tile space
        // has up to 255 spare rows per graph behind the last graph (22 % at 1000 nodes), and eighths of the padded space left
        // XCD 7 and half of XCD 6 without work (round 2's mapping).  Workgroups beyond that take the unused tail in order.
        const int real_wgs = ((p.node_ptr_pad[p.G] >> 7) + p.tpw - 1) / p.tpw;      // exact: the prep stage's padded total / 128 rows per group
        const int per = (real_wgs + 7) >> 3, q8 = blockIdx.x >> 3, x8 = blockIdx.x & 7;
        const int wg = q8 < per ? x8 * per + q8 : 8 * per + (q8 - per) * 8 + x8;
        wk.cur = wg * p.tpw;
        wk.end = min((p.n_tiles + 3) / 4, wk.cur + p.tpw);
        wk.step = 1;
        if (p.order == 1) {
            // MIRRORED PAIRS (tpw == 2).  The reference's graphs list the free samples first, and those have ~1.5 x the incoming
            // edges of the collided ones (kNN over all samples + kNN over the free ones: eval_gnn.py:160-164), so the first half of
            // a graph's tiles is heavy and the second light (first-half groups 2350 edges, second-half 1580 at the configs[2]
            // shape).  With only 2-4 groups per resident workgroup slot and launch the hardware dispatcher cannot level that out:
            // a quarter of the wave slots sat idle (tools/diag/mp_trace.py).  Every graph is a whole number of 256-row blocks
            // (kPad) = pairs of groups; workgroup `wg` stands for block `wg` and takes the group that far from the START of its
            // graph and then the group that far from its END: every workgroup costs about the same, and the launch has half as
            // many workgroups (chosen when that does not add a round of workgroups: launch_mp_fused_t).  Measured: five launches
            // 0.555 -> 0.534 ms at the configs[2] shape; "start groups first" with one group per workgroup measured no gain.
            const int blk_tile = wg * 8;
            const int bg = blk_tile < p.n_tiles ? p.ntile_graph[blk_tile] : -1;
            if (bg >= 0) {
                const int g0 = p.node_ptr_pad[bg] >> 7, g1 = p.node_ptr_pad[bg + 1] >> 7;       // the graph's groups [g0, g1), an even count
                const int i = wg - (g0 >> 1);
                wk.cur = g0 + i;
                wk.step = (g1 - 1 - i) - wk.cur;                  // >= 1
                wk.end = g1 - i;
            } else {
                wk.end = wk.cur;                                   // block behind the last graph
            }
        }
    }
    for (; wk.valid(); wk.next()) {
        // static strided split: at any moment the resident workgroups of an XCD work on ADJACENT tiles, so the K_e stream
        // is one dense front in HBM and neighbouring tiles share gathered A rows (tiles pulled one by one from a per-XCD
        // counter, or one contiguous run of tiles per workgroup, both measured 13-18 % slower)
        const int tile = kCoop ? wk.cur : wk.cur * 4 + wave;
        if constexpr (!kCoop) { if (wk.snake) wk.snake_prefetch(); }
        if (tile >= p.n_tiles) continue;
        const int t0 = tile * 32;
        const int node = t0 + j;
        // one tile per wave: the tile's graph, row range and degrees are requested TOGETHER (row_beg / deg exist for every padded
        // row, so they need not wait for the "is this tile in use" answer; the compiler sank them below the branch): two round
        // trips at a tile start instead of three (configs[4] shape: five launches 0.481 -> 0.478 ms, same box)
        int rb = kCoop ? pre_rb : p.row_beg[node], dg = kCoop ? pre_dg : p.deg[node];
        int tg = kCoop ? pre_g : p.ntile_graph[tile];                          // kCoop: the workgroup's only tile, requested above
        if constexpr (!kCoop) asm volatile("" : "+v"(rb), "+v"(dg), "+v"(tg));
        if (tg < 0) continue;                                                  // workgroup-uniform when kCoop
        const int beg = __builtin_amdgcn_readfirstlane(rb);
        const int end = __builtin_amdgcn_readlane(rb + dg, 31);
        const int n0 = p.node_ptr_pad[tg];
        GNNMP_TRC();
        GNNMP_TRC_TILE(tile);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the previous job's LDS reads are done
        if constexpr (kCoop) __syncthreads();                    // ... by every wave of the workgroup
        // K_e streams from HBM: PF tiles of a chunk are requested KD chunks ahead and wait in registers
        constexpr bool kDeep = P == 1;                           // packed bf16 tiles are cheap to hold (fp32, d = 32, two chunks ahead: measured slower, 0.875 -> 0.906 ms)
        constexpr int PF = kDeep ? NT : 1;                       // prefetched tiles per chunk (the rest is loaded in place)
        constexpr int KD = kDeep ? 2 : 1;                        // chunks ahead
        constexpr int LPT = P == 1 ? 2 : 4;                      // load instructions per tile
        // two K_e register sets used alternately (the chunk loop is unrolled by two): a set is refilled as soon as its chunk
        // has expanded it -- for the chunk two ahead when KD = 2 (each set feeds every second chunk), or the OTHER set is
        // filled for the next chunk when KD = 1.  No queue shifting and no loop-carried copies (they were 16 / 40 register
        // moves per chunk at d = 32 fp32 / d = 64 bf16, in a loop that is bound by instruction issue).
        KeRaw<P> qa[PF], qb[PF];
        // Requests inside the chunk loop are UNCONDITIONAL (a chunk beyond the tile's range re-reads slot `beg`: one cache line):
        // the compiler counts the outstanding vector-memory operations per control-flow path and, where paths differ, waits
        // for all of them (s_waitcnt vmcnt(0)) the first time a loaded register is used -- with the requests behind wave-uniform
        // branches that drained the K_e tiles of the chunk after next in the middle of every chunk.
        auto ke_fetch = [&](int cc, KeRaw<P> (&dst)[PF]) {      // wave-uniform cc
            if (!kUncond && cc >= end) return;
#pragma unroll
            for (int t = 0; t < PF; ++t) {
                const int sl = cc + j < end ? cc + j : beg;
                load_edge_slot_raw<P, NT>(p.Ke, sl, h, t, dst[t]);
            }
        };
        int pre_rec_c = 0, pre_rec_n = 0;
        // d = 32, one tile per wave: the part of H that depends on the tile's X rows, bl + Wlx X, is taken while those rows
        // are in the stage for B' (16 registers through the chunk loop) -- the node phase then neither reads X again (by then
        // the rows had left the L2: a second HBM read of [N, d] per iteration) nor waits for them
        constexpr bool kHpEarly = !kCoop && D == 32 && P != 2;
        f32x16 Hp[kHpEarly ? NT : 1];
        constexpr bool kREarly = kHpEarly && P == 0;             // ... and then the A stage is free for the R rows (fp32 rows)
        constexpr bool kKeEarly = kHpEarly;                      // registers to spare at d = 32: the first chunk's K_e is requested with its A rows
        bool r_requested = false;
        if (!kCoop || wave == 0) {
            // large batches (one tile per wave): the tile's B' rows are NOT read back -- they are W_dst times the tile's own X
            // rows, which the previous iteration wrote anyway: X rows -> stage, 16 NT^2 MFMAs (the ones the node phase no
            // longer spends on B'), result into the stage in the row format the chunks read.  Same chain on the same values:
            // the bits of the stored B'.  Saves a write and a read of [N, d] per iteration (10 % of the launch's HBM bytes).
            constexpr int XP = P == 1 ? 1 : 0;                              // X rows are stored in bf16 in the bf16 mode
            if constexpr (kCoop) dma_rows<D, P>(p.B, [&](int sr) { return t0 + sr; }, btile, lane);
            else dma_rows<D, XP>(p.X, [&](int sr) { return t0 + sr; }, btile, lane);
            const f32x4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};      // agg tile <- -inf
#pragma unroll
            for (int q = 0; q < D / 8; ++q) *reinterpret_cast<f32x4*>(agg + j * AP + h * (D / 2) + q * 4) = ninf;
            if constexpr (!kCoop) {
                // the first two chunks' edge records travel with the X rows, and the first chunk's A rows are requested before
                // the MFMAs below: one dependent round trip at the start of a tile instead of three
                if (beg + j < end) pre_rec_c = p.rec32[beg + j];
                if (beg + 32 + j < end) pre_rec_n = p.rec32[beg + 32 + j];
                wait_vmcnt<0>();
                if (beg < end) {
                    const int mine_row = (beg + j < end) ? n0 + (pre_rec_c & 0x7ffffff) : t0;
                    dma_rows<D, P>(p.A, [&](int sr) { return __builtin_amdgcn_ds_bpermute(sr * 4, mine_row); }, astage, lane);
                    if constexpr (kKeEarly) {                               // ... and its K_e tiles: their HBM latency runs under the MFMAs
                        ke_fetch(beg, qa);
                        if constexpr (KD == 2) ke_fetch(beg + 32, qb);
                    }
                }
                f32x16 z[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) z[tt] = splat16(0.f);
                auto xrows = [&](int it, f32x16& x) { read_stage_tile<D, XP>(btile, j, h, it, x); };
                if constexpr (kHpEarly) {                                   // (kNodeWInLds holds for these instantiations)
                    load_vec<NT>(wnl + LN::bl, Hp, lane);
                    linear_acc_stream<P, NT, false>(wnl + LN::wlx, xrows, Hp, lane);
                }
                if (kNodeWInLds && !p.last) {                               // W_dst is in the staged blob (not in the last iteration's)
                    linear_acc_stream<P, NT, false>(wnl + LN::m3, xrows, z, lane);
                } else {
                    const float* wb = p.wn_std + LN::m3;
                    asm volatile("" : "+s"(wb));                            // (no hoisting of these loads out of the job loop)
                    linear_acc_stream<P, NT, false>(wb, xrows, z, lane);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();                            // both halves of every row have been read
                write_stage_tiles<D, P, NT>(btile, j, h, z);
            }
        }
        // Few tiles (kCoop): the node phase is a chain of five dependent layers on ONE wave behind the edge phase, i.e. pure
        // latency.  What does not depend on the aggregation is taken out of that chain: the X and R rows are requested NOW by
        // the two waves with the fewest chunks, and the last wave computes the partial H = bl + Wlx X right after the barrier,
        // before its own chunks.  The accumulator goes through LDS as it is, so every layer sees the same fmaf chain as the
        // one-wave form: results stay bit-identical to COOP = 1.
        const float* wn = kNodeWInLds ? wnl : p.wn;
        if constexpr (!kNodeWInLds) asm volatile("" : "+s"(wn));
        if constexpr (kSplitNode && P != 1) {
            if (wave == COOP - 1) dma_rows<D, 0>(p.X, [&](int sr) { return t0 + sr; }, xstage, lane);
            if (wave == COOP - 2) dma_rows<D, 0>(p.R, [&](int sr) { return t0 + sr; }, rstage, lane);
        }
        if constexpr (kCoop) {
            wait_all();
            __syncthreads();
            if (kSplitNode && wave == COOP - 1) {
                f32x16 Hp[NT];
                load_vec<NT>(wn + LN::bl, Hp, lane);
                linear_acc_stream<P, NT, false>(wn + LN::wlx, [&](int it, f32x16& x) {
                    if constexpr (P != 1) read_stage_tile<D, 0>(xstage, j, h, it, x);
                    else load_row_tile<1, NT>(p.X, (size_t)node, h, it, x);          // bf16 mode: X rows are stored in bf16
                }, Hp, lane);
                store_tile<NT>(hpart, Hp, lane);
            }
        }
        // software pipeline over 32-edge chunks: while chunk c is multiplied and aggregated, the A rows of this wave's
        // next chunk are in flight to the LDS stage (DMA, no registers), its first K_e tile to registers, and the packed
        // record (source id local to the graph | target's row in this tile << 27) of the one after is requested
        constexpr int STEP = 32 * COOP;
        const int first = beg + (kCoop ? 32 * wave : 0);
        auto src_row = [&](int rec, bool valid) { return valid ? n0 + (rec & 0x7ffffff) : t0; };
        int rec_c = pre_rec_c, rec_n = pre_rec_n;
        if constexpr (kCoop) {
            if (first + j < end) rec_c = p.rec32[first + j];
            if (first + STEP + j < end) rec_n = p.rec32[first + STEP + j];
        }
        if (first < end) {
            if constexpr (kCoop) {                               // (one tile per wave: requested at the start of the tile)
                const int mine_row = src_row(rec_c, first + j < end);
                dma_rows<D, P>(p.A, [&](int sr) { return __builtin_amdgcn_ds_bpermute(sr * 4, mine_row); }, astage, lane);
            }
            if constexpr (!kKeEarly) {
                ke_fetch(first, qa);
                if constexpr (KD == 2) ke_fetch(first + STEP, qb);
            }
        }
        auto chunk = [&](const int c0, KeRaw<P> (&cur)[PF], KeRaw<P> (&fill)[PF]) {
            // this chunk's A rows (and the job's B rows) have landed; with KD = 2 the next chunk's K_e tiles, requested
            // AFTER them, may still be in flight (vector memory returns in order)
            if constexpr (KD == 2) {
                if (kUncond || c0 + STEP < end) wait_vmcnt<LPT * PF>();     // (requests are unconditional)
                else wait_vmcnt<0>();
            } else {
                wait_vmcnt<0>();
            }
            // Nothing the compiler generates may touch a prefetched register or LDS BEFORE that wait: an LDS-DMA in flight counts
            // as a "flat" access for its bookkeeping, and any vector-memory dependency it has to guard while one is pending becomes
            // s_waitcnt vmcnt(0) -- draining the K_e tiles of the chunk after next.  The empty asm statements pin the first use of
            // the next record and of this chunk's K_e registers behind the wait (the scheduler had hoisted the record's address
            // arithmetic above it); after the wait the bookkeeping knows they have arrived and adds nothing.
            if constexpr (kLaunder) {
            asm volatile("" : "+v"(rec_n), "+v"(rec_c));
#pragma unroll
            for (int t = 0; t < PF; ++t) {
                if constexpr (P == 1) asm volatile("" : "+v"(cur[t].lo), "+v"(cur[t].hi));
                else asm volatile("" : "+v"(cur[t].v[0]), "+v"(cur[t].v[4]), "+v"(cur[t].v[8]), "+v"(cur[t].v[12]));
            }
            }
            const int slot = c0 + j;
            const bool valid = slot < end;
            const int rec = rec_c;
            const int dloc = valid ? ((rec >> 27) & 31) : 0;
            const int eslot = valid ? slot : beg;
            if (h == 0) dl[j] = dloc * AP;
            f32x16 M[NT];
#pragma unroll
            for (int ot = 0; ot < NT; ++ot) M[ot] = splat16(bias[ot]);
            // the next chunk's A rows (and the record of the one after) are requested by `request_rows`: with AEARLY right here,
            // after this chunk's rows have been taken out of the stage as they are (packed bf16 rows: 8 registers per tile) -- a
            // whole chunk ahead instead of the second half of one; the gather is an L2 hit, but under load its round trip was
            // longer than the half chunk it had (attribution runs: the chunk loop spent a quarter of its time waiting for it)
            auto request_rows = [&]() {
                    rec_c = rec_n;
                    if ((kUncond && !kREarly) || c0 + STEP < end) {    // wave-uniform
                        const int mine_row = src_row(rec_c, c0 + STEP + j < end);       // beyond the range: the tile's first row, 32 times
                        dma_rows<D, P>(p.A, [&](int sr) { return __builtin_amdgcn_ds_bpermute(sr * 4, mine_row); }, astage, lane);
                    } else if (kREarly) {
                        // last chunk: the A stage is free from here on -- the node phase's R rows travel under this chunk's MFMAs
                        // (the same number of DMA instructions as the A rows they replace)
                        dma_rows<D, 0>(p.R, [&](int sr) { return t0 + sr; }, astage, lane);
                        r_requested = true;
                    }
                    {
                        const int ri = c0 + 2 * STEP + j;
                        if (kUncond) rec_n = p.rec32[ri < end ? ri : end - 1];
                        else if (ri < end) rec_n = p.rec32[ri];
                    }
            };
            // (the target rows B[dst] are taken out of LDS here as well: any LDS read the compiler generates while a DMA is in
            // flight is guarded by a wait for that DMA -- it cannot tell the stages apart)
            StageRaw<P> araw[kAEarly ? NT : 1], braw[kAEarly ? NT : 1];
            if constexpr (kAEarly) {
#pragma unroll
                for (int it = 0; it < NT; ++it) {
                    read_stage_raw<D, P>(astage, j, h, it, araw[it]);
                    read_stage_raw<D, P>(btile, dloc, h, it, braw[it]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                request_rows();
            }
            // hidden = relu(A[src] + B[dst] + K_e), one 32-feature tile at a time, straight into the swapped MFMA
            linear_acc_stream<P, NT, true, true>(wl + LE::w2, [&](int it, f32x16& x) {
                f32x16 a, b;
                if (it < PF) expand_raw<P>(cur[it < PF ? it : 0], x); else load_edge_slot_tile<P, NT>(p.Ke, eslot, h, it, x);
                if constexpr (kAEarly) {
                    expand_stage_raw<P>(araw[it], a);
                    expand_stage_raw<P>(braw[it], b);
                } else {
                    read_stage_tile<D, P>(astage, j, h, it, a);
                    read_stage_tile<D, P>(btile, dloc, h, it, b);
                }
                if (it == NT - 1) {
                  if constexpr (kAEarly) {
                    ke_fetch(c0 + KD * STEP, fill);
                  } else {
                    // the stage has been read: request the next chunk's rows, K_e and the record after that
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    request_rows();
                    ke_fetch(c0 + KD * STEP, fill);
                  }
                }
                x += a + b;                                     // the ReLU is applied by linear_acc_stream
            }, M, lane);
            if (end - c0 < 32) {                                 // wave-uniform: the last, partial chunk -- pad edges aggregate -inf
                const int nv = end - c0;
#pragma unroll
                for (int ot = 0; ot < NT; ++ot)
#pragma unroll
                    for (int r = 0; r < 16; ++r) M[ot][r] = phi(r, h) < nv ? M[ot][r] : -INFINITY;
            }
            // The offsets and the aggregation atomics are issued BY HAND (inline asm).  The next chunk's A rows (LDS-DMA) and the K_e
            // tiles of the chunk after it have just been requested; the compiler cannot tell that the DMA's LDS destination
            // (the A stage) and these LDS locations never alias, and guards any LDS access it generates behind an in-flight
            // DMA with s_waitcnt vmcnt(0) -- which drained EVERY outstanding request (the HBM K_e stream included) at this
            // point of every chunk, i.e. the whole prefetch pipeline ran one request deep (rounds 1-3: 47 % of the wave time in
            // s_waitcnt at d = 64 bf16).  Instructions inside asm statements get no such guard; what they need is stated here:
            // LDS instructions of a wave execute in order, so the offset reads see this chunk's dl[] writes, and the no-return
            // atomics need nothing after them (the node phase waits for lgkmcnt(0) before it reads the tile).
            // register r of this lane is edge phi(r, h) = 8 (r >> 2) + 4 h + (r & 3) of the chunk
            __builtin_amdgcn_wave_barrier();
            i32x4 o4[4];
            if constexpr (!kAsmTail) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) o4[g4] = *reinterpret_cast<const i32x4*>(dl + 8 * g4 + 4 * h);
#pragma unroll
                for (int ot = 0; ot < NT; ++ot)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_ds_fmaxf((lds_float*)(agg + o4[r >> 2][r & 3] + ot * 32 + j), M[ot][r], 0, 0, false);
            } else {
            // The MFMA results are INPUTS of this block although it does not read them: nothing interlocks an MFMA's result
            // register against a DS instruction the compiler cannot see into (its hazard recognizer pads only what it generates),
            // and the scheduler would sink the chunk's last MFMAs below this block, right in front of the atomics that send
            // their accumulators to LDS (seen as run-to-run differences with the 16-pass fp32 MFMAs).  With the dependency the
            // MFMAs are issued first; s_nop 15 + s_nop 7 are the 19 wait states a 16-pass MFMA needs before a DS read of its
            // result, and the LDS round trip of the offsets lies on top.
            // (one element per accumulator is enough: an MFMA writes all sixteen registers; a 512-bit asm operand also breaks the host pass)
            asm volatile("s_nop 15\n\ts_nop 7\n\tds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
                         "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(o4[0]), "=&v"(o4[1]), "=&v"(o4[2]), "=&v"(o4[3])
                         : "v"(lds_addr(dl) + 16u * h), "v"(M[0][0]), "v"(M[NT - 1][15]) : "memory");
            const unsigned agg0 = lds_addr(agg) + 4u * j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned a = agg0 + 4u * (unsigned)o4[r >> 2][r & 3];
#pragma unroll
                for (int ot = 0; ot < NT; ++ot)
                    asm volatile("ds_max_f32 %0, %1 offset:%2" ::"v"(a), "v"(M[ot][r]), "n"(ot * 128) : "memory");
            }
            }
            __builtin_amdgcn_wave_barrier();
        };
        // two chunks per trip in ONE basic block (exit test at the bottom), an odd last chunk behind the loop: with the exit between
        // the two bodies the register allocator copied half of the in-flight K_e registers on the back edge -- and a copy of a
        // register that a load is still filling needs s_waitcnt vmcnt(0)
        if constexpr (!kNewFlow) {
            for (int c0 = first; c0 < end; c0 += 2 * STEP) {      // round-3 form (exact-fp32 kernels)
                if constexpr (KD == 2) chunk(c0, qa, qa); else chunk(c0, qa, qb);
                if (c0 + STEP < end) {                           // wave-uniform
                    if constexpr (KD == 2) chunk(c0 + STEP, qb, qb); else chunk(c0 + STEP, qb, qa);
                }
            }
        } else {
            const int stop = end;
            int c0 = first;
            for (; c0 + STEP < stop; c0 += 2 * STEP) {
                if constexpr (KD == 2) { chunk(c0, qa, qa); chunk(c0 + STEP, qb, qb); }
                else { chunk(c0, qa, qb); chunk(c0 + STEP, qb, qa); }
            }
            if (c0 < stop) {
                if constexpr (KD == 2) chunk(c0, qa, qa); else chunk(c0, qa, qb);
            }
        }
        GNNMP_TRC();
        if constexpr (kCoop && !kSplitNode) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's aggregation atomics have been performed
            __syncthreads();
            if (wave != 0) continue;                             // wave 0 runs the node phase; the others wait at the next tile
        }
        if constexpr (kSplitNode) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's aggregation atomics (and the partial H) have been performed
            __syncthreads();
            // ---- node phase, few tiles: wave 0 finishes H (+= Wla agg), Y = R + M1 H; then waves 0 and 1 compute A' and B' side
            // by side.  Critical path behind the edge phase: three layers instead of five, no global read.
            f32x16 y[NT];
            if (wave == 0) {
                f32x16 H[NT];
                load_tile<NT>(hpart, H, lane);
                linear_acc_stream<P, NT, false>(wn + LN::wla, [&](int it, f32x16& x) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(agg + j * AP + it * 32 + q * 8 + h * 4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) x[q * 4 + c] = dg == 0 ? 0.f : a[c];      // torch_scatter: no incoming edge -> 0
                    }
                }, H, lane);
                if (p.store_h) store_row<NT>(p.Hout + (size_t)node * D, H, h);
                if constexpr (P != 1) {
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) read_stage_tile<D, 0>(rstage, j, h, tt, y[tt]);
                } else {
                    load_row<NT>(p.R + (size_t)node * D, y, h);
                }
                linear_acc_p<P, NT, NT>(wn + LN::m1, H, y, lane);
                store_row_p<P == 1 ? 1 : 0, NT>(p.Xout, (size_t)node, y, h);
                store_tile<NT>(ytile, y, lane);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (wave < 2) {
                if (wave == 1) load_tile<NT>(ytile, y, lane);
                f32x16 z[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) z[tt] = splat16(0.f);
                linear_acc_p<P, NT, NT>(wn + (wave == 0 ? LN::m2 : LN::m3), y, z, lane);
                store_row_p<P, NT>(wave == 0 ? p.Aout : p.Bout, (size_t)node, z, h);
            }
            continue;                                            // the next job's first barrier collects the workgroup
        }
        // ---- node phase.  X and R rows of the tile (contiguous, fp32) come through the two stages as well; the weights
        // (MpNBlob) are read from global memory as MFMA operands (at d = 64 the pointer is laundered per job so that the
        // compiler does not hoist those loop-invariant loads out of the job loop into ~70 registers)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float* rrows = kREarly ? astage : btile;
        // bf16 kernels, one tile per wave: after the last chunk the A and B stages (adjacent: 2 x 32 bf16 rows = 32 fp32 rows) are
        // free -- the tile's fp32 R rows come in through them by DMA and its X' / A' / PT rows go out through them as whole
        // rows.  In the chain layout every lane loads / stores 8-16 bytes of a different row: one L1 / L2 request per lane, 24
        // such loads and 16-24 such stores per tile (the stores + the last layer were 4.8 of the node phase's 12 us at d = 64).
        constexpr bool kRowsLds = GNNMP_MP_ROWS_LDS && P == 1 && !kCoop;      // rows IN through the stages (fp32 kernels: always)
        constexpr bool kStoreLds = GNNMP_MP_ROWS_LDS && !kCoop;               // rows OUT through the stages
        // (d = 64 reads its X rows here, not at the tile start: bringing those through the A stage too measured 0.521 vs 0.517 ms per
        // five launches -- one more exposed round trip and 16 more spilled bytes -- and stays off)
        constexpr bool kXLds = false && kRowsLds && !kHpEarly;
        StageRaw<1> xraw[kXLds ? NT : 1];
        if constexpr (P != 1) {                                  // fp32 rows fill a whole stage each
            if constexpr (!kHpEarly) dma_rows<D, 0>(p.X, [&](int sr) { return t0 + sr; }, astage, lane);
            if (!(kREarly && r_requested)) dma_rows<D, 0>(p.R, [&](int sr) { return t0 + sr; }, rrows, lane);
        } else if constexpr (kRowsLds) {
            if constexpr (kXLds) {                               // X rows (bf16) through the A stage first, then the stages take the R rows
                dma_rows<D, 1>(p.X, [&](int sr) { return t0 + sr; }, astage, lane);
                wait_vmcnt<0>();
#pragma unroll
                for (int it = 0; it < NT; ++it) read_stage_raw<D, 1>(astage, j, h, it, xraw[it]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            dma_rows<D, 0>(p.R, [&](int sr) { return t0 + sr; }, astage, lane);        // 32 fp32 rows = the A stage + the B stage
        }
        f32x16 H[NT];
        if constexpr (kHpEarly) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) H[tt] = Hp[tt];
            if constexpr (P != 1) wait_vmcnt<0>();
        } else {
            load_vec<NT>(wn + LN::bl, H, lane);
            if constexpr (P != 1) wait_vmcnt<0>();
            linear_acc_stream<P, NT, false>(wn + LN::wlx, [&](int it, f32x16& x) {
                if constexpr (P != 1) read_stage_tile<D, 0>(astage, j, h, it, x);
                else if constexpr (kXLds) expand_stage_raw<1>(xraw[it], x);
                else load_row_tile<1, NT>(p.X, (size_t)node, h, it, x);          // bf16 mode: X rows are stored in bf16
            }, H, lane);
        }
        linear_acc_stream<P, NT, false>(wn + LN::wla, [&](int it, f32x16& x) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(agg + j * AP + it * 32 + q * 8 + h * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) x[q * 4 + c] = dg == 0 ? 0.f : a[c];      // torch_scatter: no incoming edge -> 0
            }
        }, H, lane);
        GNNMP_TRC();                                             // (diagnostics build) H done
        if (p.store_h) store_row<NT>(p.Hout + (size_t)node * D, H, h);
        BOp<P> yop[NT];
        {
            f32x16 y[NT];
            if constexpr (P != 1) {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) read_stage_tile<D, 0>(rrows, j, h, tt, y[tt]);
            } else if constexpr (kRowsLds) {
                wait_vmcnt<0>();
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) read_stage_tile<D, 0>(astage, j, h, tt, y[tt]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // R rows are in registers: both stages are free again
            } else {
                load_row<NT>(p.R + (size_t)node * D, y, h);
            }
            linear_acc_p<P, NT, NT>(wn + LN::m1, H, y, lane);
            GNNMP_TRC();                                         // (diagnostics build) Y done
            if constexpr (kStoreLds) {
                constexpr int XP = P == 1 ? 1 : 0;                                  // X' rows: bf16 in the bf16 mode, fp32 otherwise
                float* xs = (P != 1 && !kREarly) ? btile : astage;                  // the stage the R rows were just read from
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                write_stage_tiles<D, XP, NT>(xs, j, h, y);                          // X' rows -> stage -> whole rows out
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                store_rows_coalesced<D, XP, kNtW4>(p.Xout, (size_t)t0, xs, lane);
            } else {
                store_row_p<P == 1 ? 1 : 0, NT>(p.Xout, (size_t)node, y, h);        // X is only ever read as an MFMA operand: bf16 rows lose nothing
            }
            make_ops<P, NT>(y, yop);
        }
        {
            f32x16 z[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) z[tt] = splat16(0.f);
            linear_acc_ops<P, NT, NT>(wn + LN::m2, yop, z, lane);
            if constexpr (kStoreLds) {
                float* as = (P != 1 && !kREarly) ? astage : btile;                  // the OTHER stage (X' may still be on its way out)
                write_stage_tiles<D, P, NT>(as, j, h, z);                           // A' rows -> stage -> whole rows out
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                store_rows_coalesced<D, P, kNtW4>(p.Aout, (size_t)t0, as, lane);
            } else {
                store_row_p<P, NT>(p.Aout, (size_t)node, z, h);
            }
        }
        if (kCoop || p.last) {                                   // one tile per wave: only PT for the policy head -- B' is recomputed by the next iteration
            f32x16 z[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) z[tt] = splat16(0.f);
            linear_acc_ops<P, NT, NT>(wn + LN::m3, yop, z, lane);
            if constexpr (kStoreLds) {
                float* ps = (P != 1 && !kREarly) ? btile : astage;                  // X' stage: its rows have been read out by now
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                write_stage_tiles<D, P, NT>(ps, j, h, z);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                store_rows_coalesced<D, P, kNtW4>(p.Bout, (size_t)t0, ps, lane);
            } else {
                store_row_p<P, NT>(p.Bout, (size_t)node, z, h);
            }
        }
        GNNMP_TRC();
    }
    GNNMP_TRC_END();
}


// =====================================================================================================
// mp_fused_w8: the same message-passing iteration for d = 64 with bf16 operands and large batches, re-cut around the LDS
// budget (round 5).  What bounded mp_fused_kernel<64, 1, 1> was its NODE phase: 16 of the ~60 us of a tile, almost all of
// it waiting for the five 64 x 64 matrices (40 KB) that every tile fetched from L2 as MFMA operands, one dependent round
// trip per layer -- they did not fit the LDS next to eight waves' tiles (8 x 16.1 KB + 48.4 KB = 177 KB).  Here they do:
//   * the tile's B' rows (W_dst X: the target's term of the tile's incoming edges) never go to LDS: they stay in 16 packed
//     registers in the layout the MFMA produced them in (lane = row) and every chunk fetches the row of ITS edge's target
//     with ds_bpermute (the crossbar of the LDS pipe, no LDS storage): same values, 4 KB per wave freed;
//   * ONE workgroup of eight waves per CU (the same eight waves per CU as before), so ONE copy of the weights: message
//     layer 8.4 KB + node phase 40.3 KB + (last iteration only) the standard W_dst 8 KB + 8 x (aggregation tile 8 KB +
//     offsets + A-row stage 4 KB) = 155.7 KB of the 160;
//   * with only the A stage and the (by then consumed) aggregation tile free during the node phase, the tile's rows move
//     through those: X rows in through the A stage (requested under the last chunk's MFMAs), the fp32 R rows in through
//     the aggregation tile once W_la has read it, X' / A' / PT out as whole rows through the A stage and the two halves
//     of the aggregation tile.
// The two four-wave halves of the workgroup behave like two workgroups of mp_fused_kernel (virtual index 2 b + half):
// same walk over the four-tile groups (XCD eighths, resident workgroups on the heavy / light snake), no barrier after
// the weights are staged.  Same arithmetic in the same order on the same values as mp_fused_kernel<64, 1, 1>: the scores
// are bit-identical (tests/test_full_size_bf16_gpu.py compares the two through GNNMP_MP_W8=0).
// =====================================================================================================
// P = 1 (bf16 operands): as described above.  P = 0 (exact fp32, round 5 as well): the same eight-wave cut WITHOUT the node-phase
// matrices in LDS (80 KB in fp32) -- message layer 16.6 KB + 8 x (aggregation tile 8 KB + offsets + A-row stage 8 KB) = 145.7 KB; what
// it buys there is the second wave per SIMD (mp_fused_kernel<64, 0, 1> keeps three 8 KB stages per wave and fits ONE four-wave
// workgroup per CU: one wave per SIMD, nothing to cover its waits with -- the launch ran at 0.25 of the HBM roof / 41 % of the
// matrix pipe).
template <int D, int P>
__host__ __device__ constexpr int mp_w8_lds_floats() {
    return ((MpEBlob<D, P>::size + 3) & ~3) + (P == 1 ? ((MpNBlob<D, P>::size + 3) & ~3) + MpNBlob<D, P>::T : 0) +
           8 * (32 * kAggPitch<D> + 32 + RowGeom<D, P>::STAGE_FLOATS);
}

template <int D, int P>
__global__ __launch_bounds__(512, 1) void mp_fused_w8_kernel(MpFusedParams p) {
    constexpr int NT = D / 32;
    // bf16 (measured at the configs[2] shape, profiles/r05_mp_traffic_attribution.txt): the X', A', PT rows leave with non-temporal
    // stores (they are not read again in this launch and would displace the gathered A rows in L2: mp 0.454 -> 0.440 ms), and the
    // chunks start on the 32-slot tiles K_e is stored in (kAl below: 0.440 -> 0.431 ms, K_e lines fetched 2.14 M -> 1.90 M per launch).
    // fp32: neither pays (the stores are neutral, the extra partial chunk per tile costs 1.5 %); non-temporal X / R row reads cost 2 %
    // in both modes (the second read of the X rows then always misses).
    constexpr int kNtR = 0;
    constexpr bool kNtW = P == 1;
    constexpr bool kAl = P == 1;
    constexpr int XP = P == 1 ? 1 : 0;                           // X rows are stored in bf16 in the bf16 mode
    constexpr bool kWLds = P == 1;                               // node-phase matrices in LDS
    using LE = MpEBlob<D, P>;
    using LN = MpNBlob<D, P>;
    using G = RowGeom<D, P>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;                                             // MpEBlob (message layer)
    float* wnl = lds + ((LE::size + 3) & ~3);                    // MpNBlob (node phase) of THIS iteration (kWLds)
    float* wm3x = wnl + ((LN::size + 3) & ~3);                   // last iteration: the standard W_dst (its blob holds the policy matrix there)
    float* base = kWLds ? wm3x + LN::T : wnl;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int sub = wave >> 2, w4 = wave & 3;
    constexpr int AP = kAggPitch<D>, AGG = 32 * AP;
    float* agg = base + wave * (AGG + 32 + G::STAGE_FLOATS);     // [32][D] fp32 at pitch AP; X rows in / R rows in / A' and PT rows out pass through it
    int* dl = reinterpret_cast<int*>(agg + AGG);                 // [32] agg row offsets (floats) of this chunk's targets
    float* astage = agg + AGG + 32;                              // gathered A rows of the current chunk; X rows in / X' rows out
    // virtual four-wave workgroup (XCD vx, index vq of vU on it)
    const int vx = blockIdx.x & 7, vq = (blockIdx.x >> 3) * 2 + sub, vU = (gridDim.x >> 3) * 2;
#ifdef GNNMP_MP_TRACE
    long long* trc = p.trace ? p.trace + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 32 : nullptr;
    int trc_n = 0, trc_tiles = 0;
    if (trc && lane == 0) {
        for (int i = 0; i < 32; ++i) trc[i] = 0;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trc[28] = (long long)hw | ((long long)xcc << 32);
    }
    GNNMP_TRC();
#endif
    stage(wl, p.we, LE::size);
    if constexpr (kWLds) {
        stage(wnl, p.wn, LN::size);
        if (p.last) stage(wm3x, p.wn_std + LN::m3, LN::T);
    }
    __syncthreads();                                             // the only workgroup barrier
    if (p.tpw > 0 && p.order != 2) {                             // virtual workgroups of the unused tail of the padded tile space
        const int per = ((((p.node_ptr_pad[p.G] >> 7) + p.tpw - 1) / p.tpw) + 7) >> 3;
        if (vq >= per) { GNNMP_TRC_END(); return; }
    }
    float bias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias[t] = wl[LE::b2 + (t * 2 + ((j >> 2) & 1)) * 16 + (j & 3) + 4 * (j >> 3)];
    const int real_tiles = min(p.n_tiles, p.node_ptr_pad[p.G] >> 5);
    XcdWalk wk((real_tiles + 3) / 4, vx, vq, vU);
    if (p.order == 2) {
        const int blocks = p.node_ptr_pad[p.G] >> 8, perb = (blocks + 7) >> 3;
        wk.snake = 1;
        wk.U = vU;
        wk.u = vq;
        wk.blk0 = vx * perb;
        wk.B = max(0, min(perb, blocks - wk.blk0));
        wk.span = p.blk_span;
        wk.k = 0;
        const int s0 = wk.pos(0);
        int2 sp0 = make_int2(0, 0);
        if (s0 < 2 * wk.B) sp0 = p.blk_span[wk.blk_of(s0)];
        wk.cur = 0;
        wk.snake_set(s0, sp0);
        wk.ns = 2 * wk.B;
    } else {
        const int real_wgs = ((p.node_ptr_pad[p.G] >> 7) + p.tpw - 1) / p.tpw;
        const int per = (real_wgs + 7) >> 3;
        const int wg = vq < per ? vx * per + vq : 8 * per + (vq - per) * 8 + vx;
        wk.cur = wg * p.tpw;
        wk.end = min((p.n_tiles + 3) / 4, wk.cur + p.tpw);
        wk.step = 1;
    }
    for (; wk.valid(); wk.next()) {
        const int tile = wk.cur * 4 + w4;
        if (wk.snake) wk.snake_prefetch();
        if (tile >= p.n_tiles) continue;
        // the tile's graph, row range and degrees are requested TOGETHER (row_beg / deg exist for every padded row, so they do not
        // have to wait for the "is this tile in use" answer): one round trip at the tile start instead of two
        const int t0 = tile * 32;
        const int node = t0 + j;
        int rb = p.row_beg[node], dg = p.deg[node];
        int tg = p.ntile_graph[tile];
        asm volatile("" : "+v"(rb), "+v"(dg), "+v"(tg));        // (all three loads in flight before the first is waited for: the compiler sank rb / dg below the branch)
        if (tg < 0) continue;
        const int beg = __builtin_amdgcn_readfirstlane(rb);
        const int end = __builtin_amdgcn_readlane(rb + dg, 31);
        const int n0 = p.node_ptr_pad[tg];
        GNNMP_TRC();
        GNNMP_TRC_TILE(tile);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the previous job's LDS reads (rows on their way out) are done
        // bf16: both K_e tiles of a chunk wait packed in registers, two chunks ahead; fp32: one tile one chunk ahead (16 registers a
        // tile), the second is loaded in place under the first tile's sixteen 64-cycle MFMAs
        constexpr int PF = P == 1 ? NT : 1, KD = P == 1 ? 2 : 1, LPT = P == 1 ? 2 : 4, STEP = 32;
        // matrices of the node phase / of W_dst: LDS (bf16) or global memory with the pointer laundered per tile (fp32: the loads are
        // loop-invariant and would be hoisted out of the tile loop into dozens of registers)
        const float* wn = kWLds ? wnl : p.wn;
        const float* wm3 = kWLds ? (p.last ? wm3x : wnl + LN::m3) : p.wn_std + LN::m3;
        if constexpr (!kWLds) asm volatile("" : "+s"(wn), "+s"(wm3));
        // chunks start on the 32-slot tiles K_e is stored in (kAl): every K_e request is four whole 128-byte lines per half wave; the
        // first chunk's leading lanes (slots of the previous node tile) and the last chunk's trailing lanes are masked
        const int first = kAl ? (beg & ~31) : beg;
        const int end_al = (end + 31) & ~31;
        KeRaw<P> qa[PF], qb[PF];
        auto ke_fetch = [&](int cc, KeRaw<P> (&dst)[PF]) {      // wave-uniform cc; unconditional (see mp_fused_kernel)
#pragma unroll
            for (int t = 0; t < PF; ++t) {
                const int sl = cc + j < (kAl ? end_al : end) ? cc + j : beg;
                load_edge_slot_raw<P, NT>(p.Ke, sl, h, t, dst[t]);
            }
        };
        // ---- tile start: B' = W_dst X of the tile's own rows, kept in registers (packed bf16, lane = row)
        int pre_rec_c = 0, pre_rec_n = 0;
        int l0 = lane;                                           // (laundered per tile: see the node phase)
        asm volatile("" : "+v"(l0));
        dma_rows<D, XP, kNtR>(p.X, [&](int sr) { return t0 + sr; }, agg, l0);           // X rows (4 KB bf16 / 8 KB fp32) through the aggregation tile
        if (first + j >= beg && first + j < end) pre_rec_c = p.rec32[first + j];
        if (first + 32 + j < end) pre_rec_n = p.rec32[first + 32 + j];
        wait_vmcnt<0>();
        // The X rows leave the LDS for registers and the aggregation tile is reset BEFORE anything else is requested: an LDS access
        // the compiler generates while an LDS-DMA is in flight is guarded by s_waitcnt vmcnt(0) (it cannot tell the A stage from
        // this tile), which would put the first chunk's A rows and K_e behind a full round trip at every tile start
        StageRaw<XP> xr[NT];
#pragma unroll
        for (int it = 0; it < NT; ++it) read_stage_raw<D, XP>(agg, j, h, it, xr[it]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();                        // both halves of every X row have been read
        {
            const f32x4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};      // agg tile <- -inf
#pragma unroll
            for (int q = 0; q < D / 8; ++q) *reinterpret_cast<f32x4*>(agg + j * AP + h * (D / 2) + q * 4) = ninf;
        }
        auto src_row = [&](int rec, bool valid) { return valid ? n0 + (rec & 0x7ffffff) : t0; };
        int rec_c = pre_rec_c, rec_n = pre_rec_n;
        if (beg < end) {                                         // the first chunk's A rows and the first two chunks' K_e travel under the MFMAs below
            const int mine_row = src_row(rec_c, first + j >= beg && first + j < end);
            dma_rows<D, P>(p.A, [&](int sr) { return __builtin_amdgcn_ds_bpermute(sr * 4, mine_row); }, astage, lane);
            ke_fetch(first, qa);
            if constexpr (KD == 2) ke_fetch(first + STEP, qb);
        }
        BOp<P> bpk[NT];
        {
            f32x16 z[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) z[tt] = splat16(0.f);
            linear_acc_stream<P, NT, false>(wm3, [&](int it, f32x16& x) { expand_stage_raw<XP>(xr[it], x); }, z, lane);
        
            make_ops<P, NT>(z, bpk);                            // (bf16: rounded exactly like the rows mp_fused_kernel writes into its B stage)
        }
        bool x_requested = false;
        auto chunk = [&](const int c0, KeRaw<P> (&cur)[PF], KeRaw<P> (&fill)[PF]) {
            // this chunk's rows and K_e have landed; with KD = 2 the next chunk's K_e may still be in flight
            if constexpr (KD == 2) wait_vmcnt<LPT * PF>(); else wait_vmcnt<0>();
            asm volatile("" : "+v"(rec_n), "+v"(rec_c));
#pragma unroll
            for (int t = 0; t < PF; ++t) {
                if constexpr (P == 1) asm volatile("" : "+v"(cur[t].lo), "+v"(cur[t].hi));
                else asm volatile("" : "+v"(cur[t].v[0]), "+v"(cur[t].v[4]), "+v"(cur[t].v[8]), "+v"(cur[t].v[12]));
            }
            const int slot = c0 + j;
            const bool valid = slot < end && (!kAl || slot >= beg);
            const int rec = rec_c;
            const int dloc = valid ? ((rec >> 27) & 31) : 0;
            const int eslot = valid ? slot : beg;
            if (h == 0) dl[j] = dloc * AP;
            f32x16 M[NT];
#pragma unroll
            for (int ot = 0; ot < NT; ++ot) M[ot] = splat16(bias[ot]);
            StageRaw<P> araw[NT];
            const int baddr = (dloc + 32 * h) * 4;              // the lane that holds row dloc's features of THIS lane's half
#pragma unroll
            for (int it = 0; it < NT; ++it) read_stage_raw<D, P>(astage, j, h, it, araw[it]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // the next chunk's A rows -- or, behind the last chunk, the tile's X rows for the node phase (the same number of DMA
            // instructions) -- and the record of the chunk after
            rec_c = rec_n;
            if (c0 + STEP < end) {                               // wave-uniform
                const int mine_row = src_row(rec_c, c0 + STEP + j < end);
                dma_rows<D, P>(p.A, [&](int sr) { return __builtin_amdgcn_ds_bpermute(sr * 4, mine_row); }, astage, lane);
            } else {
                int lx = lane;                                   // (laundered: these addresses must not be hoisted out of the chunk loop)
                asm volatile("" : "+v"(lx));
                dma_rows<D, XP, kNtR>(p.X, [&](int sr) { return t0 + sr; }, astage, lx);
                x_requested = true;
            }
            {
                const int ri = c0 + 2 * STEP + j;
                rec_n = p.rec32[ri < end ? ri : end - 1];
            }
            linear_acc_stream<P, NT, true, true>(wl + LE::w2, [&](int it, f32x16& x) {
                // the target row's features of this tile: ds_bpermute moves registers through the LDS crossbar, it reads no LDS
                // memory, so the compiler does not guard it against the rows in flight to the A stage
                f32x16 a, b;
                if constexpr (P == 1) {
                    StageRaw<P> braw;
                    const i32x4 lo = __builtin_bit_cast(i32x4, bpk[it].lo), hi = __builtin_bit_cast(i32x4, bpk[it].hi);
                    i32x4 glo, ghi;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        glo[w] = __builtin_amdgcn_ds_bpermute(baddr, lo[w]);
                        ghi[w] = __builtin_amdgcn_ds_bpermute(baddr, hi[w]);
                    }
                    const bf16x8 blo = __builtin_bit_cast(bf16x8, glo), bhi = __builtin_bit_cast(bf16x8, ghi);
                    braw.lo = blo;
                    braw.hi = bhi;
                    expand_stage_raw<P>(braw, b);
                } else {
                    // (four registers at a time through vector bit casts: the element-wise form -- bit_cast<int>(v[r]) -> ds_bpermute ->
                    // bit_cast<float> in a loop over r -- was compiled to ONE ds_bpermute of v[0] broadcast to all sixteen elements)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 part = {bpk[it].v[q * 4 + 0], bpk[it].v[q * 4 + 1], bpk[it].v[q * 4 + 2], bpk[it].v[q * 4 + 3]};
                        const i32x4 pi = __builtin_bit_cast(i32x4, part);
                        i32x4 g;
#pragma unroll
                        for (int w = 0; w < 4; ++w) g[w] = __builtin_amdgcn_ds_bpermute(baddr, pi[w]);
                        const f32x4 gf = __builtin_bit_cast(f32x4, g);
#pragma unroll
                        for (int w = 0; w < 4; ++w) b[q * 4 + w] = gf[w];
                    }
                }
                if (it < PF) expand_raw<P>(cur[it < PF ? it : 0], x); else load_edge_slot_tile<P, NT>(p.Ke, eslot, h, it, x);
                expand_stage_raw<P>(araw[it], a);
                if (it == NT - 1) ke_fetch(c0 + KD * STEP, KD == 2 ? cur : fill);
                x += a + b;
            }, M, lane);
            if (end - c0 < 32 || (kAl && c0 < beg)) {            // a partial chunk (wave-uniform): pad edges aggregate -inf
                const int nv = end - c0, n0v = kAl ? beg - c0 : 0;
#pragma unroll
                for (int ot = 0; ot < NT; ++ot)
#pragma unroll
                    for (int r = 0; r < 16; ++r) M[ot][r] = (phi(r, h) < nv && phi(r, h) >= n0v) ? M[ot][r] : -INFINITY;
            }
            __builtin_amdgcn_wave_barrier();
            i32x4 o4[4];
            asm volatile("s_nop 15\n\ts_nop 7\n\tds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
                         "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(o4[0]), "=&v"(o4[1]), "=&v"(o4[2]), "=&v"(o4[3])
                         : "v"(lds_addr(dl) + 16u * h), "v"(M[0][0]), "v"(M[NT - 1][15]) : "memory");
            const unsigned agg0 = lds_addr(agg) + 4u * j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned a = agg0 + 4u * (unsigned)o4[r >> 2][r & 3];
#pragma unroll
                for (int ot = 0; ot < NT; ++ot)
                    asm volatile("ds_max_f32 %0, %1 offset:%2" ::"v"(a), "v"(M[ot][r]), "n"(ot * 128) : "memory");
            }
            __builtin_amdgcn_wave_barrier();
        };
        {
            int c0 = beg < end ? first : end;
            for (; c0 + STEP < end; c0 += 2 * STEP) {
                if constexpr (KD == 2) { chunk(c0, qa, qa); chunk(c0 + STEP, qb, qb); }
                else { chunk(c0, qa, qb); chunk(c0 + STEP, qb, qa); }
            }
            if (c0 < end) { if constexpr (KD == 2) chunk(c0, qa, qa); else chunk(c0, qa, qb); }
        }
        GNNMP_TRC();
        // ---- node phase (bf16: every matrix in LDS)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's aggregation atomics have been performed
        // (the per-lane address vectors of the row movers below are invariant across tiles; hoisted out of the tile loop they cost
        // ~40 registers that then spill around the chunk loop: the lane id is laundered per tile so that they are recomputed here)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if (!x_requested) dma_rows<D, XP, kNtR>(p.X, [&](int sr) { return t0 + sr; }, astage, ln);     // a tile without incoming edges
        f32x16 H[NT];
        load_vec<NT>(wn + LN::bl, H, lane);
        wait_vmcnt<0>();
        linear_acc_stream<P, NT, false>(wn + LN::wlx, [&](int it, f32x16& x) { read_stage_tile<D, XP>(astage, j, h, it, x); }, H, lane);
        linear_acc_stream<P, NT, false>(wn + LN::wla, [&](int it, f32x16& x) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(agg + j * AP + it * 32 + q * 8 + h * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) x[q * 4 + c] = dg == 0 ? 0.f : a[c];      // torch_scatter: no incoming edge -> 0
            }
        }, H, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();                        // the aggregation tile has been read by every lane: it takes the R rows
        dma_rows<D, 0, kNtR>(p.R, [&](int sr) { return t0 + sr; }, agg, ln);
        GNNMP_TRC();                                             // (diagnostics build) H done
        if (p.store_h) store_row<NT>(p.Hout + (size_t)node * D, H, h);
        BOp<P> yop[NT];
        {
            f32x16 y[NT];
            wait_vmcnt<0>();
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) read_stage_tile<D, 0>(agg, j, h, tt, y[tt]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            linear_acc_p<P, NT, NT>(wn + LN::m1, H, y, lane);
            GNNMP_TRC();                                         // (diagnostics build) Y done
            __builtin_amdgcn_wave_barrier();
            write_stage_tiles<D, XP, NT>(astage, j, h, y);       // X' rows -> A stage -> whole rows out
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            store_rows_coalesced<D, XP, kNtW>(p.Xout, (size_t)t0, astage, ln);
            make_ops<P, NT>(y, yop);
        }
        {
            f32x16 z[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) z[tt] = splat16(0.f);
            linear_acc_ops<P, NT, NT>(wn + LN::m2, yop, z, lane);
            write_stage_tiles<D, P, NT>(agg, j, h, z);           // A' rows -> (bf16: first half of) the aggregation tile -> whole rows out
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            store_rows_coalesced<D, P, kNtW>(p.Aout, (size_t)t0, agg, ln);
        }
        if (p.last) {                                            // PT for the policy head (B' is recomputed by the next iteration otherwise)
            f32x16 z[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) z[tt] = splat16(0.f);
            linear_acc_ops<P, NT, NT>(wn + LN::m3, yop, z, lane);
            // bf16: second half of the aggregation tile; fp32 (A' fills the whole tile): the A stage, whose X' rows have been read out
            float* ps = P == 1 ? agg + G::STAGE_FLOATS : astage;
            if constexpr (P != 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            write_stage_tiles<D, P, NT>(ps, j, h, z);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            store_rows_coalesced<D, P, kNtW>(p.Bout, (size_t)t0, ps, ln);
        }
    
        GNNMP_TRC();
    }
    GNNMP_TRC_END();
}

// =====================================================================================================
// policy: score_e = w3 . relu(W2 relu(PS[src] - PT[dst] + PE_e) + b2)
// =====================================================================================================
template <int D, int P>
__global__ __launch_bounds__(256) void policy_kernel(PolicyParams p) {
    constexpr int NT = D / 32;
    using L = PolBlob<D, P>;
    using G = RowGeom<D, P>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    // per wave: the PS rows of the tile's 32 sources and the PT rows of its 32 targets, gathered through LDS-DMA in the
    // quad-coalesced layout of the message kernel (a chain-layout gather is one L1 access per lane)
    float* sstage = lds + ((L::size + 3) & ~3) + wave * 2 * G::STAGE_FLOATS;
    float* tstage = sstage + G::STAGE_FLOATS;
    stage(wl, p.w, L::size);
    __syncthreads();
    XcdWalk wk((p.n_tiles + 3) / 4);
    // software pipeline over this wave's tiles: while tile i is multiplied, the rows and the PE tile of tile i+1 are in
    // flight and the records of tile i+2 are requested
    int4 rec_c = make_int4(-1, -1, -1, 0), rec_n = make_int4(-1, -1, -1, 0);
    int g_c = -1, g_n = -1, tile_c = -1, tile_n = -1;
    auto fetch = [&](int grp, int4& rec, int& g, int& tile) {
        tile = grp * 4 + wave;
        g = -1;
        rec = make_int4(-1, -1, -1, 0);
        if (tile < p.n_tiles) {
            g = p.etile_graph[tile];
            rec = p.csr[tile * 32 + j];
        }
    };
    // the prefetched PE tile stays AS LOADED (bf16 tiles packed) until its tile is multiplied: converting it where it is
    // requested made the conversion wait for the load it had just issued (s_waitcnt vmcnt(0) right behind the request: the
    // bf16 policy kernel ran without any prefetch)
    KeRaw<P> hid_n[NT];
    // The 32 slots of a tile are consecutive CSR slots of one graph, i.e. sorted by target: their targets are the CONTIGUOUS node
    // rows t_first .. t_last (3-5 rows at kNN in-degrees; nodes without incoming edges in between widen the range, beyond 32 rows the
    // per-edge gather stays).  Only those rows are fetched -- one LDS-DMA instruction (G::RPI rows)
    // where a gather of the target row per edge took G::NI, a quarter of the bytes this kernel moves through the CU's L1 -- and
    // every lane reads row (its target - t_first) of the stage (same address for the edges of one target: an LDS broadcast).
    int dl_c = 0;                                               // stage row of the lane's target in the tile whose rows are in flight / staged
    auto issue = [&](const int4& rec, int g, int tile, int& dl) {      // rows + PE tile of a fetched tile (g wave-uniform after the wait)
        if (g < 0) return;
        using G = RowGeom<D, P>;
        const int srow = rec.x >= 0 ? rec.x : 0;
        dma_rows<D, P>(p.PS, [&](int sr) { return __builtin_amdgcn_ds_bpermute(sr * 4, srow); }, sstage, lane);
        const int nv = __builtin_popcountll(__builtin_amdgcn_ballot_w64(rec.x >= 0 && h == 0));      // valid slots (pads trail)
        const int t_first = __builtin_amdgcn_readfirstlane(rec.y);
        const int t_last = __builtin_amdgcn_readlane(rec.y, nv > 0 ? nv - 1 : 0);
        const int R = nv > 0 ? t_last - t_first + 1 : 0;          // 1 .. 32 (0: a tile of padding only), wave-uniform
        if (R > 32) {                                               // nodes without incoming edges in between: gather the row of every edge
            const int trow = rec.y >= 0 ? rec.y : 0;
            dl = j;
            dma_rows<D, P>(p.PT, [&](int sr) { return __builtin_amdgcn_ds_bpermute(sr * 4, trow); }, tstage, lane);
        } else {
        dl = rec.x >= 0 ? rec.y - t_first : 0;
        const char* base = reinterpret_cast<const char*>(p.PT);
#pragma unroll
        for (int i = 0; i < G::NI; ++i) {
            if (i * G::RPI >= R) break;
            const int sr = G::RPI * i + lane / G::PP;
            const int pc = (lane % G::PP) ^ G::swz(sr);
            const char* gp = base + (size_t)(t_first + (sr < R ? sr : R - 1)) * G::RB + pc * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)(tstage + i * 256), 16, 0, 0);
        }
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) load_edge_slot_raw<P, NT>(p.PE, tile * 32 + j, h, tt, hid_n[tt]);
    };
    if (wk.valid()) {
        fetch(wk.cur, rec_c, g_c, tile_c);
        wk.next();
        wait_vmcnt<0>();
        g_c = __builtin_amdgcn_readfirstlane(g_c);
        issue(rec_c, g_c, tile_c, dl_c);
        if (wk.valid()) fetch(wk.cur, rec_n, g_n, tile_n);
    }
    while (tile_c >= 0) {
        const int4 rec = rec_c;
        const int g = g_c;
        const int dloc = dl_c;
        f32x16 hid[NT], a[NT], b[NT];
        wait_vmcnt<0>();      // this tile's rows and PE tile (and the next records) have landed
        if (g >= 0) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                expand_raw<P>(hid_n[tt], hid[tt]);
                read_stage_tile<D, P>(sstage, j, h, tt, a[tt]);
                read_stage_tile<D, P>(tstage, dloc, h, tt, b[tt]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the stages have been read: the next tile may overwrite them
        // advance the pipeline
        rec_c = rec_n; tile_c = wk.valid() ? tile_n : -1;
        g_c = tile_c >= 0 ? __builtin_amdgcn_readfirstlane(g_n) : -1;
        if (tile_c >= 0) {
            wk.next();
            issue(rec_c, g_c, tile_c, dl_c);
            if (wk.valid()) fetch(wk.cur, rec_n, g_n, tile_n);
        }
        if (g < 0) continue;
        const int s = rec.x, t = rec.y;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) hid[tt] += a[tt] - b[tt];
        f32x16 y[NT], w3[NT];
        load_vec<NT>(wl + L::b2, y, lane);
        linear_acc_relu_p<P, NT, NT>(wl + L::w2, hid, y, lane);
        relu_<NT>(y);
        load_vec<NT>(wl + L::w3, w3, lane);
        float sc = 0.f;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc = fmaf(w3[tt][r], y[tt][r], sc);
        sc = xsum(sc);
        if (h == 0 && s >= 0) {
            p.scores[rec.z] = sc;
            if (p.dense) {
                const int nb = p.node_ptr_pad[g];
                const long long ng = p.node_ptr[g + 1] - p.node_ptr[g];
                p.dense[p.dense_ptr[g] + (long long)(t - nb) * ng + (s - nb)] = sc;   // P[target, source]
            }
        }
    }
}

__global__ void zero_dense_kernel(float* __restrict__ dense, const long long* __restrict__ n_ptr) {
    const long long n = *n_ptr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dense[i] = 0.f;
}

// tap helper: gather padded-node-space rows back to caller node order
__global__ void unpad_rows_kernel(int G, int total_nodes, int D, const int* __restrict__ node_ptr,
                                  const int* __restrict__ node_ptr_pad, const float* __restrict__ src,
                                  float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)total_nodes * D) return;
    const int n = (int)(i / D), f = (int)(i % D);
    const int g = find_graph(node_ptr, G, n);
    dst[i] = src[(size_t)(node_ptr_pad[g] + n - node_ptr[g]) * D + f];
}

__global__ void goal_tap_kernel(int G, const int* __restrict__ goal_node, const int* __restrict__ node_ptr_pad,
                                float* __restrict__ dst) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) dst[g] = (float)(goal_node[g] - node_ptr_pad[g]);
}

// =====================================================================================================
// host-side launchers (called from api.cpp)
// =====================================================================================================
#define LAUNCH_CHECK()                        \
    do {                                      \
        hipError_t _e = hipGetLastError();    \
        if (_e != hipSuccess) return _e;      \
    } while (0)

template <class K>
static hipError_t set_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

int prep_parts(int G, int E) {
    static const int parts_env = getenv("GNNMP_PREP_PARTS") ? atoi(getenv("GNNMP_PREP_PARTS")) : 0;    // experiments
    const long long g = G > 0 ? G : 1;
    long long parts = (long long)E / (g * kPrepEdgesPerPart);                 // by graph size ...
    if (parts > 256 / g) parts = 256 / g;                                     // ... but no more workgroups than fill the device once
    if (parts > 1) {
        // once the columns of a graph are split at all, finer is better up to eight parts and two workgroups per CU (kuka7, 64 x
        // 30 k edges: 0.071 ms at 3 parts, 0.059 at 8; beyond eight the part histograms every part sums per node take over:
        // kuka14, 32 x 131 k edges, 0.137 ms at 8 parts, 0.154 at 16)
        long long fine = (long long)E / (g * (kPrepEdgesPerPart / 2));
        if (fine > 512 / g) fine = 512 / g;
        if (fine > 8) fine = 8;
        if (fine > parts) parts = fine;
    }
    if (parts_env > 0) parts = parts_env;
    if (parts < 1) parts = 1;
    if (parts > kPrepMaxParts) parts = kPrepMaxParts;
    return (int)parts;
}

// `hist`: 2 * prep_parts * Npad ints of workspace when prep_parts > 1 (unused otherwise)
hipError_t launch_prep(const PrepParams& q, int Npad, int Epad, int* hist, hipStream_t st) {
    const size_t glds = (size_t)(2 * kPrepCap + 1024) * sizeof(int);
    const int parts = prep_parts(q.G, q.E);
    if (parts == 1) {
        const hipError_t attr = set_lds(prep_small_kernel, glds);
        if (attr != hipSuccess) return attr;
        // Few graphs (the reference's call is ONE): a graph's target nodes are cut into slices, one workgroup each.  A
        // workgroup's scatter of the int4 records is bound by its CU's request rate to L2 (one 16-byte store per clock: 7 of the
        // 14 us a single 1000-node graph's CSR build took); every slice's workgroup reads all columns (L2 hits) and ranks /
        // scatters its own.  One slice per 128 target nodes at most, eight at most, and about two workgroups per CU in all
        // (256 graphs of 1000 nodes: 0.068 -> 0.057 ms with two slices, 0.081 with four).
        static const int slices_env = getenv("GNNMP_PREP_SLICES") ? atoi(getenv("GNNMP_PREP_SLICES")) : 0;    // experiments
        int slices = q.G > 0 ? 512 / q.G : 1;
        const int by_nodes = q.G > 0 ? (Npad / q.G) / 128 : 1;
        if (slices > by_nodes) slices = by_nodes;
        if (slices > 8) slices = 8;
        if (slices_env > 0) slices = slices_env;
        if (slices < 1) slices = 1;
        if (slices > kGstatStride - 1) slices = kGstatStride - 1;
        hipLaunchKernelGGL(prep_small_kernel, dim3(q.G * slices), dim3(1024), glds, st, q, Npad, Epad, slices);
        LAUNCH_CHECK();
        return hipSuccess;
    }
    hipError_t attr = set_lds(prep_hist_kernel, glds);
    if (attr != hipSuccess) return attr;
    attr = set_lds(prep_scatter_kernel, glds);
    if (attr != hipSuccess) return attr;
    const unsigned grid = (unsigned)(((q.G + 7) / 8) * 8 * parts);            // (graph, part) -> workgroup: see prep_hist_kernel
    hipLaunchKernelGGL(prep_hist_kernel, dim3(grid), dim3(1024), glds, st, q, Npad, Epad, parts, hist);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(prep_scatter_kernel, dim3(grid), dim3(1024), glds, st, q, parts, hist, hist + (size_t)parts * Npad);
    LAUNCH_CHECK();
    return hipSuccess;
}


template <int D, int P>
static hipError_t launch_obs_t(const ObsParams& p, const NodeF64Params& q, int G, hipStream_t st) {
    const int f64_blocks = P == 1 ? 0 : q.n_wg;
    size_t lds = f64_blocks ? (size_t)(((q.blob.size + 3) & ~3) + 2 * f64_obs_chunk(D) * D) * sizeof(float) : 0;
    const size_t obs_lds = (D == 32 && P != 2) ? (size_t)((p.blob.size + 3) & ~3) * sizeof(float) : 0;     // obstacle role: its blob
    if (obs_lds > lds) lds = obs_lds;
    if (lds) {
        const hipError_t e = set_lds(obs_kernel<D, P>, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((obs_kernel<D, P>), dim3(f64_blocks + 2 * G), dim3(256), lds, st, p, q, f64_blocks);
    LAUNCH_CHECK();
    return hipSuccess;
}
// precision dispatch: P = 0 fp32, 1 bf16, 2 bf16x3
#define GNNMP_DISPATCH_DP(D_, P_, CALL)                                                     \
    do {                                                                                     \
        if ((D_) == 32 && (P_) == 0) { constexpr int DD = 32, PP = 0; return CALL; }         \
        if ((D_) == 32 && (P_) == 1) { constexpr int DD = 32, PP = 1; return CALL; }         \
        if ((D_) == 32 && (P_) == 2) { constexpr int DD = 32, PP = 2; return CALL; }         \
        if ((D_) == 64 && (P_) == 0) { constexpr int DD = 64, PP = 0; return CALL; }         \
        if ((D_) == 64 && (P_) == 1) { constexpr int DD = 64, PP = 1; return CALL; }         \
        if ((D_) == 64 && (P_) == 2) { constexpr int DD = 64, PP = 2; return CALL; }         \
        return hipErrorInvalidValue;                                                         \
    } while (0)

hipError_t launch_obs(int D, int P, const ObsParams& p, const NodeF64Params& q, int G, hipStream_t st) {
    GNNMP_DISPATCH_DP(D, P, (launch_obs_t<DD, PP>(p, q, G, st)));
}

template <int D, int P, bool EDGE, int WAVES>
static hipError_t launch_pre_t(const PreParams& p, int n_wg, size_t lds_bytes, hipStream_t st) {
    hipError_t e = set_lds(pre_kernel<D, P, EDGE, WAVES>, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((pre_kernel<D, P, EDGE, WAVES>), dim3(n_wg), dim3(WAVES * 64), lds_bytes, st, p);
    LAUNCH_CHECK();
    return hipSuccess;
}
template <int D, int P>
static hipError_t launch_pre_dp(bool edge, int waves, const PreParams& p, int n_wg, size_t lds_bytes, hipStream_t st) {
    if (waves == 4) return edge ? launch_pre_t<D, P, true, 4>(p, n_wg, lds_bytes, st) : launch_pre_t<D, P, false, 4>(p, n_wg, lds_bytes, st);
    if (waves == 8) return edge ? launch_pre_t<D, P, true, 8>(p, n_wg, lds_bytes, st) : launch_pre_t<D, P, false, 8>(p, n_wg, lds_bytes, st);
    return hipErrorInvalidValue;
}
hipError_t launch_pre(int D, int P, bool edge, int waves, const PreParams& p_in, int n_tiles32, size_t lds_bytes, hipStream_t st) {
    int n_wg = n_tiles32 / waves;
    PreParams p = p_in;
    p.n_wg = n_wg;
    n_wg = (n_wg + 7) & ~7;
    GNNMP_DISPATCH_DP(D, P, (launch_pre_dp<DD, PP>(edge, waves, p, n_wg, lds_bytes, st)));
}

template <int D, int P, bool EDGE, int WAVES>
static hipError_t launch_pre_resident_t(const PreParams& p, size_t lds_bytes, int n_cu, hipStream_t st) {
    hipError_t e = set_lds(pre_resident_kernel<D, P, EDGE, WAVES>, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((pre_resident_kernel<D, P, EDGE, WAVES>), dim3(n_cu), dim3(WAVES * 64), lds_bytes, st, p);
    LAUNCH_CHECK();
    return hipSuccess;
}
template <int D, int P>
static hipError_t launch_pre_resident_both_t(const PreParams& pn, const PreParams& pe, size_t lds_bytes, int node_blocks, int edge_blocks,
                                             hipStream_t st) {
    hipError_t e = set_lds(pre_resident_both_kernel<D, P, 12>, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((pre_resident_both_kernel<D, P, 12>), dim3(node_blocks + edge_blocks), dim3(12 * 64), lds_bytes, st, pn, pe, node_blocks);
    LAUNCH_CHECK();
    return hipSuccess;
}
// both block counts are multiples of 8 (XCD walk of the body)
hipError_t launch_pre_resident_both(int D, int P, const PreParams& pn, const PreParams& pe, size_t lds_bytes, int node_blocks,
                                    int edge_blocks, hipStream_t st) {
    if (D == 32 && P == 0) return launch_pre_resident_both_t<32, 0>(pn, pe, lds_bytes, node_blocks, edge_blocks, st);
    if (D == 32 && P == 1) return launch_pre_resident_both_t<32, 1>(pn, pe, lds_bytes, node_blocks, edge_blocks, st);
    return hipErrorInvalidValue;
}
// d = 32: 12 waves (3 per SIMD, <= 168 VGPRs); d = 64 (bf16 operands only -- fp32 weights do not fit): 8 waves
hipError_t launch_pre_resident(int D, int P, bool edge, const PreParams& p, size_t lds_bytes, int n_cu, hipStream_t st) {
    if (D == 32 && P == 0) return edge ? launch_pre_resident_t<32, 0, true, 12>(p, lds_bytes, n_cu, st) : launch_pre_resident_t<32, 0, false, 12>(p, lds_bytes, n_cu, st);
    if (D == 32 && P == 1) return edge ? launch_pre_resident_t<32, 1, true, 12>(p, lds_bytes, n_cu, st) : launch_pre_resident_t<32, 1, false, 12>(p, lds_bytes, n_cu, st);
    if (D == 64 && P == 1) return edge ? launch_pre_resident_t<64, 1, true, 12>(p, lds_bytes, n_cu, st) : launch_pre_resident_t<64, 1, false, 8>(p, lds_bytes, n_cu, st);
    return hipErrorInvalidValue;
}

// Grid of the grid-stride kernels: one workgroup per 4 tiles, but never more workgroups than the device keeps
// RESIDENT (occupancy query x CUs): the tile space is split evenly by XcdWalk, so a grid beyond residency only
// adds a second, thinly populated round of workgroups (measured on round 1's message kernel: 0.72 -> 0.66 ms per step at d = 32,
// 0.56 -> 0.52 at d = 64 / bf16).  Multiple of 8 for the XCD walk.
struct Residency { int per_cu, cus; };
static Residency resident_workgroups(const void* kernel, size_t lds_bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, size_t>, Residency> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({kernel, lds_bytes});
    if (it != cache.end()) return it->second;
    Residency r{0, 256};
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) r.cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r.per_cu, kernel, 256, lds_bytes) != hipSuccess || r.per_cu < 1) r.per_cu = 4;
    if (getenv("GNNMP_DEBUG_GRID")) fprintf(stderr, "[gnnmp] resident workgroups per CU: %d (lds %zu)\n", r.per_cu, lds_bytes);
    return cache[{kernel, lds_bytes}] = r;
}

// max_per_cu: measured sweet spot of the kernel -- more resident waves than that only add contention in the memory
// system (round 1's message kernel at d = 32: 0.657 ms per step at 4 workgroups per CU, 0.669 at 5, 0.81 at 6-7 when the compiler is
// forced to fit them; policy: 0.137 at 3, 0.152 at 6); GNNMP_WGS_PER_CU overrides it for experiments
template <class K>
static int grid_for(K kernel, size_t lds_bytes, int n_tiles, int max_per_cu) {
    int groups = ((n_tiles + 3) / 4 + 7) & ~7;
    if (groups < 8) groups = 8;
    const Residency r = resident_workgroups(reinterpret_cast<const void*>(kernel), lds_bytes);
    int per_cu = r.per_cu < max_per_cu ? r.per_cu : max_per_cu;
    static const int forced = getenv("GNNMP_WGS_PER_CU") ? atoi(getenv("GNNMP_WGS_PER_CU")) : 0;
    if (forced > 0) per_cu = forced;
    const int cap = (r.cus * per_cu) & ~7;
    return groups < cap ? groups : (cap < 8 ? 8 : cap);
}

#ifdef GNNMP_MP_TRACE
static long long* g_mp_trace = nullptr;
static size_t g_mp_trace_n = 0;
// diagnostics build: copy the per-wave timestamps of the LAST message-passing launch to the host (tools/diag/mp_trace.py)
extern "C" long long gnnmp_debug_mp_trace(long long* dst, long long cap) {
    if (!g_mp_trace) return 0;
    const size_t n = g_mp_trace_n < (size_t)cap ? g_mp_trace_n : (size_t)cap;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(dst, g_mp_trace, n * sizeof(long long), hipMemcpyDeviceToHost);
    return (long long)n;
}
#endif
template <int D, int P, int COOP>
static hipError_t launch_mp_fused_t(const MpFusedParams& p, hipStream_t st) {
    const size_t lds = (size_t)(((MpEBlob<D, P>::size + 3) & ~3) + mp_lds_floats<D, P, COOP>() + (GNNMP_MP_NODEW_LDS && D == 32 && P != 2 ? ((MpNBlob<D, P>::size + 3) & ~3) : 0)) * sizeof(float);
    hipError_t e = set_lds(mp_fused_kernel<D, P, COOP>, lds);
    if (e != hipSuccess) return e;
    if (COOP == 1) {
        // one workgroup per group of four tiles, NOT capped at what is resident: the hardware dispatcher hands the next
        // group to whichever CU frees up first, which balances the uneven tiles better than a static share per resident
        // workgroup did (five launches: cfg 2 0.952 -> 0.910 ms, kuka7 bf16 0.72 -> 0.70, kuka14 bf16 0.633 -> 0.592;
        // a grid of 1.5x the resident workgroups is the worst case: 1.087 ms at cfg 2)
        // p.n_tiles is the padded upper bound (up to 255 spare rows per graph); the kernel cuts its XCD eighths from the tiles
        // actually in use.  A workgroup may take several ADJACENT groups (GNNMP_MP_TPW, experiments only): with the eighths cut
        // right, one group per workgroup is the fastest everywhere -- five launches at t = 1 / 2 / 4: cfg 2 0.740 / 0.779 / 0.780 ms,
        // kuka7 bf16 0.582 / 0.589 / 0.779, kuka14 bf16 0.562 / 0.597 / 0.865
        const int groups_cap = (p.n_tiles + 3) / 4;
        static const int forced_tpw = getenv("GNNMP_MP_TPW") ? atoi(getenv("GNNMP_MP_TPW")) : 0;      // experiments
        int tpw = 1;
        if (forced_tpw > 0) tpw = forced_tpw;
        MpFusedParams q = p;
        q.tpw = tpw;
#ifdef GNNMP_MP_TRACE
        {
            static long long* tbuf = nullptr;
            static size_t tcap = 0;
            const size_t need = (size_t)(groups_cap + 16) * 4 * 32;
            if (tcap < need) { if (tbuf) (void)hipFree(tbuf); (void)hipMalloc(&tbuf, need * sizeof(long long)); tcap = need; }
            q.trace = tbuf;
            g_mp_trace = tbuf; g_mp_trace_n = need;
        }
#endif
        // dispatch order (see the kernel): more groups than resident workgroup slots -> the resident workgroups stay and share
        // the groups out on a snake (order 2); otherwise one workgroup per group.  Five launches, plain / mirrored pairs / snake:
        // configs[2] bf16 0.524 / 0.500 / 0.496 ms, configs[4] bf16 0.509 / - / 0.483, configs[1] fp32 0.701 / - / 0.669.
        // GNNMP_MP_ORDER = 0 / 1 / 2 forces one (experiments, tests/test_full_size_gpu.py)
        static const int order_env = getenv("GNNMP_MP_ORDER") ? atoi(getenv("GNNMP_MP_ORDER")) : -1;
        q.order = 0;
        if (forced_tpw == 0) {
            const Residency r = resident_workgroups(reinterpret_cast<const void*>(mp_fused_kernel<D, P, COOP>), lds);
            const long long slots = (long long)r.cus * (r.per_cu > 0 ? r.per_cu : 1);
            q.order = groups_cap > slots && slots >= 8 ? 2 : 0;
            if (order_env >= 0) q.order = order_env;
            if (q.order == 2 && slots < 8) q.order = 0;
            if (q.order == 1) q.tpw = tpw = 2;
            if (q.order == 2) {                // persistent: exactly the resident workgroups (a multiple of 8)
                hipLaunchKernelGGL((mp_fused_kernel<D, P, COOP>), dim3((unsigned)((slots / 8) * 8)), dim3(256), lds, st, q);
                LAUNCH_CHECK();
                return hipSuccess;
            }
        }
        int grid = ((groups_cap + tpw - 1) / tpw + 7) & ~7;
        if (forced_tpw < 0) {                  // experiment: persistent workgroups, -forced_tpw per CU
            q.tpw = 0;
            grid = 256 * (-forced_tpw);
        }
        hipLaunchKernelGGL((mp_fused_kernel<D, P, COOP>), dim3(grid < 8 ? 8 : grid), dim3(256), lds, st, q);
    } else {
        const int grid = ((p.n_tiles + 7) & ~7) < 8 ? 8 : ((p.n_tiles + 7) & ~7);          // one workgroup per tile
        hipLaunchKernelGGL((mp_fused_kernel<D, P, COOP>), dim3(grid), dim3(COOP * 64), lds, st, p);
    }
    LAUNCH_CHECK();
    return hipSuccess;
}
// d = 64, bf16 operands, large batches: eight waves and every matrix in LDS (mp_fused_w8_kernel)
template <int D, int P>
static hipError_t launch_mp_fused_w8(const MpFusedParams& p, hipStream_t st) {
    const size_t lds = (size_t)mp_w8_lds_floats<D, P>() * sizeof(float);
    hipError_t e = set_lds(mp_fused_w8_kernel<D, P>, lds);
    if (e != hipSuccess) return e;
    const int groups_cap = (p.n_tiles + 3) / 4;
    MpFusedParams q = p;
    q.tpw = 1;
#ifdef GNNMP_MP_TRACE
    {
        static long long* tbuf = nullptr;
        static size_t tcap = 0;
        const size_t need = (size_t)(groups_cap + 600) * 4 * 32;
        if (tcap < need) { if (tbuf) (void)hipFree(tbuf); (void)hipMalloc(&tbuf, need * sizeof(long long)); tcap = need; }
        q.trace = tbuf;
        g_mp_trace = tbuf; g_mp_trace_n = need;
    }
#endif
    // one workgroup per CU stays resident (two virtual four-wave workgroups each); more groups than virtual slots -> the resident
    // workgroups share the groups out on the snake (order 2), else one virtual workgroup per group
    Residency r{1, 256};
    {
        int dev = 0;
        hipDeviceProp_t prop;
        static int cus = 0;
        if (!cus && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus) r.cus = cus;
    }
    const int slots = (r.cus / 8) * 8;
    static const int order_env = getenv("GNNMP_MP_ORDER") ? atoi(getenv("GNNMP_MP_ORDER")) : -1;
    q.order = groups_cap > 2 * slots && slots >= 8 ? 2 : 0;
    if (order_env == 0 || order_env == 2) q.order = order_env;
    if (q.order == 2 && slots < 8) q.order = 0;
    unsigned grid = q.order == 2 ? (unsigned)slots : (unsigned)((((groups_cap + 1) / 2) + 7) & ~7);
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL((mp_fused_w8_kernel<D, P>), dim3(grid), dim3(512), lds, st, q);
    LAUNCH_CHECK();
    return hipSuccess;
}

// few tiles (single graphs, small batches): eight waves share a tile
template <int D, int P>
static hipError_t launch_mp_fused_dp(const MpFusedParams& p, hipStream_t st) {
    static const int forced = getenv("GNNMP_MP_COOP") ? atoi(getenv("GNNMP_MP_COOP")) : -1;
    const bool coop = forced >= 0 ? forced != 0 : p.n_tiles <= (D > 32 ? kCoopMaxTiles64 : kCoopMaxTiles32);
    // d = 64 with fp32 / bf16x3 operands: four waves per tile instead of eight -- at eight waves a wave has 256 registers and the
    // kernel spilled 116 of them (single 2000-node kuka7 graph: 42 us per launch)
    static const int w8 = getenv("GNNMP_MP_W8") ? atoi(getenv("GNNMP_MP_W8")) : 1;      // 0: the four-wave form (mp_fused_kernel<64, P, 1>), for A/B runs and the bit-identity tests
    if constexpr (D == 64 && P == 0) {
        if (coop) return launch_mp_fused_t<D, P, 4>(p, st);
        return w8 ? launch_mp_fused_w8<D, P>(p, st) : launch_mp_fused_t<D, P, 1>(p, st);
    }
    else if constexpr (D > 32 && P != 1) return coop ? launch_mp_fused_t<D, P, 4>(p, st) : launch_mp_fused_t<D, P, 1>(p, st);
    else if constexpr (D == 64 && P == 1) {
        if (coop) return launch_mp_fused_t<D, P, 8>(p, st);
        return w8 ? launch_mp_fused_w8<D, P>(p, st) : launch_mp_fused_t<D, P, 1>(p, st);
    }
    else return coop ? launch_mp_fused_t<D, P, 8>(p, st) : launch_mp_fused_t<D, P, 1>(p, st);
}
hipError_t launch_mp_fused(int D, int P, const MpFusedParams& p_in, hipStream_t st) {
    const MpFusedParams& p = p_in;
    GNNMP_DISPATCH_DP(D, P, (launch_mp_fused_dp<DD, PP>(p, st)));
}

template <int D, int P>
static hipError_t launch_policy_t(const PolicyParams& p, hipStream_t st) {
    const size_t lds = (size_t)(((PolBlob<D, P>::size + 3) & ~3) + 4 * 2 * RowGeom<D, P>::STAGE_FLOATS) * sizeof(float);
    hipError_t e = set_lds(policy_kernel<D, P>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((policy_kernel<D, P>), dim3(grid_for(policy_kernel<D, P>, lds, p.n_tiles, 3)), dim3(256), lds, st, p);
    LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_policy(int D, int P, const PolicyParams& p, hipStream_t st) {
    GNNMP_DISPATCH_DP(D, P, (launch_policy_t<DD, PP>(p, st)));
}

hipError_t launch_unpad_rows(int G, int total_nodes, int D, const int* node_ptr, const int* node_ptr_pad,
                             const float* src, float* dst, hipStream_t st) {
    const size_t n = (size_t)total_nodes * D;
    hipLaunchKernelGGL(unpad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, G, total_nodes, D,
                       node_ptr, node_ptr_pad, src, dst);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_zero_dense(float* dense, const long long* n_ptr, hipStream_t st) {
    hipLaunchKernelGGL(zero_dense_kernel, dim3(2048), dim3(256), 0, st, dense, n_ptr);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_goal_tap(int G, const int* goal_node, const int* node_ptr_pad, float* dst, hipStream_t st) {
    hipLaunchKernelGGL(goal_tap_kernel, dim3((G + 255) / 256), dim3(256), 0, st, G, goal_node, node_ptr_pad, dst);
    LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace gnnmp
