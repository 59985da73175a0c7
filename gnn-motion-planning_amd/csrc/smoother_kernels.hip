// smoother_kernels.hip -- HIP kernels (gfx950) of the GNN path-smoother forward pass,
// ModelSmoother.forward of the reference (model_smoother.py:104-142), batched over independent
// smoothing problems.  Per `loop` iteration:
//
//   sm_graph   sm_knn + sm_edges in ONE launch, a workgroup per problem, samples staged in LDS (the usual form; the two
//              kernels below remain for problems whose buffers exceed a workgroup's LDS share)
//   sm_knn     10 nearest samples (free + collided) of every path node     (model_smoother.py:125)
//   sm_edges   caller edges + kNN edges -> sorted by (target, source), duplicates dropped
//                                                                            (model_smoother.py:126-128)
//   sm_msg     per edge: x_j, x_i = node_code(...), message = lin_0([x_j - x_i, x_j, x_i])
//                                                                            (model_smoother.py:135-136, 36-39)
//   sm_node    per path node: S = ordered sum of incoming messages, h = x + lin_1(S),
//              new waypoint = smooth_node(h); end points kept              (model_smoother.py:32-34,139-140)
//
// Only rows < P of h are ever consumed (model_smoother.py:139), so only edges whose target is a path
// node are evaluated, and x is recomputed per edge endpoint instead of per node (13 P endpoints vs
// P + F + Co nodes).  The sum over incoming messages runs in coalesced edge order (increasing source
// id per target), i.e. the order a sequential scatter-add over the coalesced edge list uses.
// Weights (d = 128: 0.46 MB) are read as MFMA A operands straight from global memory / L2.
#include <hip/hip_runtime.h>
#include <math.h>
#include <mutex>
#include <stdlib.h>
#include "chain.hpp"
#include "layout.hpp"
#include "kernels.hpp"

namespace gnnmp {

__device__ __forceinline__ int sm_round32(int x) { return (x + 31) & ~31; }
// start of problem b in the padded path-node / edge index spaces (no scan needed)
// prefix arrays; ONE problem may be given by its totals alone (all four pointers NULL, gnnmp.h)
__device__ __forceinline__ int sm_pp(const SmParams& p, int i) { return p.path_ptr ? p.path_ptr[i] : (i > 0 ? p.total_path : 0); }
__device__ __forceinline__ int sm_fp(const SmParams& p, int i) { return p.path_ptr ? p.free_ptr[i] : (i > 0 ? p.one_free : 0); }
__device__ __forceinline__ int sm_cp(const SmParams& p, int i) { return p.path_ptr ? p.coll_ptr[i] : (i > 0 ? p.one_coll : 0); }
__device__ __forceinline__ int sm_ep(const SmParams& p, int i) { return p.path_ptr ? p.edge_ptr[i] : (i > 0 ? p.total_edges : 0); }
__device__ __forceinline__ int sm_poff(const SmParams& p, int b) { return sm_round32(sm_pp(p, b)) + 32 * b; }
__device__ __forceinline__ int sm_eoff(const SmParams& p, int b) {
    return sm_round32(sm_ep(p, b) + kSmK * sm_pp(p, b)) + 32 * b;
}

// path_cur = path / scale     (model_smoother.py:118)
__global__ void sm_init_kernel(int n, float scale, const float* __restrict__ path, float* __restrict__ cur) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cur[i] = path[i] / scale;
}

// out = path_cur * scale      (model_smoother.py:142)
__global__ void sm_final_kernel(int n, float scale, const float* __restrict__ cur, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cur[i] * scale;
}

// ---------------------------------------------------------------------------------------------------
// kNN: one wave per path node; lanes stride over the problem's samples keeping a per-lane running
// best; k rounds of (lane-local argmin, wave argmin, retire the winner).
// ---------------------------------------------------------------------------------------------------
// compact lists of the tiles in use: this problem's runs of edge / path tiles appended at positions reserved with one atomic each (any
// order: a tile's work does not depend on where it is listed).  Problem 0 zeroes the counters of the other parity for the next iteration.
__device__ __forceinline__ void sm_list_tiles(const SmParams& p, int b, int et0, int et1, int pt0, int pt1, int tid, int nthreads) {
    if (!p.tile_cnt) return;
    __shared__ int s_lbase[2];
    if (tid == 0) {
        s_lbase[0] = atomicAdd(&p.tile_cnt[p.parity * 2 + 0], et1 - et0);
        s_lbase[1] = atomicAdd(&p.tile_cnt[p.parity * 2 + 1], pt1 - pt0);
        if (b == 0) { p.tile_cnt[(p.parity ^ 1) * 2 + 0] = 0; p.tile_cnt[(p.parity ^ 1) * 2 + 1] = 0; }
    }
    __syncthreads();
    for (int i = tid; i < et1 - et0; i += nthreads) p.elist[s_lbase[0] + i] = et0 + i;
    for (int i = tid; i < pt1 - pt0; i += nthreads) p.plist[s_lbase[1] + i] = pt0 + i;
}

__global__ __launch_bounds__(256) void sm_knn_kernel(SmParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int node = blockIdx.x * 4 + wave;               // global path row
    if (node >= p.total_path) return;
    int lo = 0, hi = p.B;                                  // problem of this path row
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sm_pp(p, mid) <= node) lo = mid; else hi = mid; }
    const int b = lo;
    const int C = p.C;
    const int f0 = sm_fp(p, b), F = sm_fp(p, b + 1) - f0;
    const int c0 = sm_cp(p, b), Co = sm_cp(p, b + 1) - c0;
    const int ns = F + Co;
    // first iteration: the scaled working copy of the path (model_smoother.py:118) is written here, not by a launch of
    // its own; this wave reads its row from the caller's path (same arithmetic, path / scale)
    const bool first = p.init_from_path != 0;
    const float* q = first ? p.path + (size_t)node * C : p.cur + (size_t)node * C;
    const float qdiv = first ? p.scale : 1.0f;
    if (first && lane < C) p.cur[(size_t)node * C + lane] = q[lane] / p.scale;
    constexpr int kMaxPerLane = 32;                        // up to 2048 samples per problem
    float dist[kMaxPerLane];
#pragma unroll
    for (int t = 0; t < kMaxPerLane; ++t) {
        const int s = t * 64 + lane;
        float d = INFINITY;
        if (s < ns) {
            const float* x = (s < F) ? p.free_pts + (size_t)(f0 + s) * C : p.collided + (size_t)(c0 + s - F) * C;
            d = 0.f;
            for (int c = 0; c < C; ++c) {
                const float df = x[c] / p.scale - (first ? q[c] / qdiv : q[c]);
                d = fmaf(df, df, d);
            }
        }
        dist[t] = d;
    }
    const int k = ns < kSmK ? ns : kSmK;
    for (int r = 0; r < kSmK; ++r) {
        int res = -1;
        if (r < k) {
            float bd = INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int t = 0; t < kMaxPerLane; ++t) {
                const int s = t * 64 + lane;
                if (dist[t] < bd) { bd = dist[t]; bi = s; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float od = __shfl_xor(bd, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
            }
            res = bi;
#pragma unroll
            for (int t = 0; t < kMaxPerLane; ++t)
                if (t * 64 + lane == bi) dist[t] = INFINITY;
        }
        if (lane == 0) p.knn[(size_t)node * kSmK + r] = res;
    }
}

// ---------------------------------------------------------------------------------------------------
// edge list: one workgroup per problem.  Candidates = caller edges with a path-node target + kNN
// edges (sample -> path node).  Sorted by key = target * M + source, duplicates dropped.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sm_edges_kernel(SmParams p) {
    extern __shared__ int sm_lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int p0 = sm_pp(p, b), P = sm_pp(p, b + 1) - p0;
    const int F = sm_fp(p, b + 1) - sm_fp(p, b), Co = sm_cp(p, b + 1) - sm_cp(p, b);
    const int M = P + F + Co;
    const int e0 = sm_ep(p, b), ne = sm_ep(p, b + 1) - e0;
    // the caps are the caller's promises (gnnmp_smooth_batch.max_*): a problem beyond them gets NO edges (the sort buffers are
    // sized from them) and is reported through gnnmp_smoother_status
    const bool fits = P <= p.path_cap && F + Co <= p.samp_cap && ne + kSmK * P <= p.cand_cap;
    const int ncand = fits ? ne + kSmK * P : 0;
    int* key = sm_lds;                    // [cap]
    int* sorted = sm_lds + p.cand_cap;    // [cap]
    __shared__ int s_scan[256];
    __shared__ int s_carry;
    for (int c = tid; c < ncand; c += 256) {
        int src, dst;
        if (c < ne) {
            src = (int)p.edge_index[e0 + c];
            dst = (int)p.edge_index[(size_t)p.total_edges + e0 + c];
        } else {
            const int pn = (c - ne) / kSmK, r = (c - ne) % kSmK;
            const int s = p.knn[(size_t)(p0 + pn) * kSmK + r];
            src = (s >= 0) ? P + s : -1;
            dst = pn;
        }
        const bool ok = src >= 0 && src < M && dst >= 0 && dst < P;   // only rows < P of h are consumed
        key[c] = ok ? dst * M + src : 0x7fffffff;
    }
    __syncthreads();
    for (int c = tid; c < ncand; c += 256) {          // stable rank sort
        const int kc = key[c];
        int rank = 0;
        for (int x = 0; x < ncand; ++x) {
            const int kx = key[x];
            rank += (kx < kc) || (kx == kc && x < c);
        }
        sorted[rank] = kc;
    }
    __syncthreads();
    const int eoff = sm_eoff(p, b);
    const int poff = sm_poff(p, b);
    if (tid == 0) s_carry = 0;
    for (int i = tid; i < sm_round32(P); i += 256) { p.seg_beg[poff + i] = 0; p.seg_cnt[poff + i] = 0; }
    __syncthreads();
    for (int base = 0; base < ncand; base += 256) {   // unique + compaction
        const int c = base + tid;
        const int kc = (c < ncand) ? sorted[c] : 0x7fffffff;
        const int keep = (c < ncand) && kc != 0x7fffffff && (c == 0 || sorted[c - 1] != kc);
        s_scan[tid] = keep;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            int a = 0;
            if (tid >= off) a = s_scan[tid - off];
            __syncthreads();
            s_scan[tid] += a;
            __syncthreads();
        }
        if (keep) {
            const int pos = s_carry + s_scan[tid] - 1;
            const int dst = kc / M, src = kc - dst * M;
            p.e_src[eoff + pos] = src;
            p.e_dst[eoff + pos] = dst;
            atomicAdd(&p.seg_cnt[poff + dst], 1);
            // first edge of a target's run
            if (c == 0 || sorted[c - 1] / M != dst || sorted[c - 1] == 0x7fffffff) p.seg_beg[poff + dst] = eoff + pos;
        }
        __syncthreads();
        if (tid == 255) s_carry += s_scan[255];
        __syncthreads();
    }
    const int n = s_carry;
    if (tid == 0) { p.e_count[b] = n; p.stat[b] = fits ? 0 : 1; }
    // tile -> problem maps of this problem's whole capacity range (-1 = unused tile; no separate fill launch): the range
    // ends where the next problem's begins, the last problem's at the end of the tile space
    const int et_used = (eoff + sm_round32(n)) / 32, pt_used = (poff + sm_round32(P)) / 32;
    const int et_end = b + 1 < p.B ? sm_eoff(p, b + 1) / 32 : p.n_etiles;
    const int pt_end = b + 1 < p.B ? sm_poff(p, b + 1) / 32 : p.n_ptiles;
    for (int t = eoff / 32 + tid; t < et_end; t += 256) p.etile_prob[t] = t < et_used ? b : -1;
    for (int t = poff / 32 + tid; t < pt_end; t += 256) { p.ptile_prob[t] = t < pt_used ? b : -1; if (p.tgt_flag) p.tgt_flag[t] = 0; }
    sm_list_tiles(p, b, eoff / 32, et_used, poff / 32, pt_used, tid, 256);
}

// ---------------------------------------------------------------------------------------------------
// kNN + edge list in ONE launch, one 1024-thread workgroup per problem (the form used whenever a problem's samples fit
// the LDS; sm_knn_kernel + sm_edges_kernel above remain for larger ones).  The two-launch form spent 44 + 26 us of a
// 115 us single call (the reference's pattern: five calls with loop = 1 per planning problem, smoother.py:243) on this:
// every path node's wave re-read and re-divided all samples from global memory with one dependent round trip per
// coordinate, and the edge list ran its compaction scan through 32 barriers.  Here the scaled samples are staged in LDS
// once per problem (row stride made odd: lane = sample reads are conflict-free), a wave handles a path node with all
// its sample slots in flight per coordinate, the neighbour ids never leave LDS, the rank sort uses four lanes per
// candidate (inside the candidate's target bucket) and the compaction is one wave-shuffle scan per 1024 candidates: 24 us
// for the 20-waypoint, 1000-sample call (stage 4, kNN 14 -- twenty path nodes' distance and selection loops share the four
// SIMDs of one CU --, sort 4, compaction 2).
// Same arithmetic as the two kernels: x / scale and path / scale as IEEE divisions, squared distance accumulated over the
// coordinates in order with fmaf, (distance, sample index) minimum with infinite / NaN distances never selected, keys
// target * M + source sorted ascending, duplicates dropped -- identical neighbour sets and edge lists.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sm_scan_1024(int v, int* wsum, int& total) {     // inclusive prefix over the workgroup
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

// (distance, index) minimum over the 64 lanes, lowest index on ties, result in every lane: first the minimum distance, then
// the lowest index among the lanes that hold it.  Inside a row of 16 lanes the partner values come through the VALU's DPP
// path (quad permutes, then the half-row and row mirrors: any pairing that crosses the halves works for a reduction),
// across rows through v_permlane16_swap / v_permlane32_swap -- no LDS round trips (ds_bpermute shuffles made the ten
// selection rounds the longest part of a path node's kNN).
template <int CTRL> __device__ __forceinline__ float sm_dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ int sm_dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ float sm_wave_min(float x) {
    x = fminf(x, sm_dpp_f<0xB1>(x));           // quad_perm [1,0,3,2]
    x = fminf(x, sm_dpp_f<0x4E>(x));           // quad_perm [2,3,0,1]
    x = fminf(x, sm_dpp_f<0x141>(x));          // row_half_mirror
    x = fminf(x, sm_dpp_f<0x140>(x));          // row_mirror
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fminf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fminf(__uint_as_float(c[0]), __uint_as_float(c[1]));
}
__device__ __forceinline__ int sm_wave_min(int x) {
    x = min(x, sm_dpp_i<0xB1>(x));
    x = min(x, sm_dpp_i<0x4E>(x));
    x = min(x, sm_dpp_i<0x141>(x));
    x = min(x, sm_dpp_i<0x140>(x));
    const auto a = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
    x = min((int)a[0], (int)a[1]);
    const auto c = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    return min((int)c[0], (int)c[1]);
}

template <int kSlots>          // 64-sample slots per lane: 16 (up to 1024 samples per problem) or 32 (up to 2048)
__global__ __launch_bounds__(1024) void sm_graph_kernel(SmParams p) {
    extern __shared__ int sm_lds[];
    __shared__ int s_wsum[16];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int C = p.C, cs = C | 1;
    const int p0 = sm_pp(p, b), P = sm_pp(p, b + 1) - p0;
    const int f0 = sm_fp(p, b), F = sm_fp(p, b + 1) - f0;
    const int c0 = sm_cp(p, b), Co = sm_cp(p, b + 1) - c0;
    const int ns = F + Co, M = P + ns;
    const int e0 = sm_ep(p, b), ne = sm_ep(p, b + 1) - e0;
    const int ncand = ne + kSmK * P;
    float* xs = reinterpret_cast<float*>(sm_lds);                    // [samp_cap][cs] scaled samples
    float* qs = xs + (size_t)p.samp_cap * cs;                        // [path_cap][cs] scaled path rows
    int* knn_l = reinterpret_cast<int*>(qs + (size_t)p.path_cap * cs);   // [path_cap][kSmK]
    int* key = knn_l + p.path_cap * kSmK;                            // [cand_cap]
    int* sorted = key + p.cand_cap;                                  // [cand_cap]
    int* cnt_l = sorted + p.cand_cap;                                // [path_cap]
    int* beg_l = cnt_l + p.path_cap;                                 // [path_cap]
    const int eoff = sm_eoff(p, b), poff = sm_poff(p, b);
    // the caps are the caller's upper bounds (gnnmp_smooth_batch.max_*): a problem beyond them gets no edges instead of
    // running over the LDS carve-up
    const bool fits = P <= p.path_cap && ns <= p.samp_cap && ncand <= p.cand_cap;
    const bool first = p.init_from_path != 0;
    // the first 1024 candidates' caller edges are requested now and consumed after the kNN
    int pre_src = 0, pre_dst = 0;
    if (fits && tid < ne) {
        pre_src = (int)p.edge_index[e0 + tid];
        pre_dst = (int)p.edge_index[(size_t)p.total_edges + e0 + tid];
    }
    if (fits) {
        // the free rows and the collided rows of a problem are two contiguous runs: element i of the staged block is element i
        // of the first run or element i - F C of the second; sixteen independent loads per thread and trip
        constexpr int U = 16;
        const int nfe = F * C, nel = ns * C;
        const float* fr = p.free_pts + (size_t)f0 * C;
        const float* co = p.collided + (size_t)c0 * C - nfe;
        for (int base = tid; base < nel; base += U * 1024) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * 1024;
                v[u] = i < nel ? (i < nfe ? fr[i] : co[i]) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * 1024;
                if (i < nel) {
                    const int sidx = i / C, c = i - sidx * C;
                    xs[sidx * cs + c] = v[u] / p.scale;
                }
            }
        }
    }
    // first iteration: the scaled working copy of the path (model_smoother.py:118) is written here
    for (int i = tid; i < P * C; i += 1024) {
        const int n = i / C, c = i - n * C;
        const size_t at = (size_t)(p0 + n) * C + c;
        const float q = first ? p.path[at] / p.scale : p.cur[at];
        if (first) p.cur[at] = q;
        if (fits) qs[n * cs + c] = q;
    }
    for (int i = tid; i < P && i < p.path_cap; i += 1024) { cnt_l[i] = 0; beg_l[i] = 0; }
    __syncthreads();
    if (fits) {
        // ---- kNN: wave w takes path nodes w, w + 16, ...; lane l holds samples l, l + 64, ...
        const int k = ns < kSmK ? ns : kSmK;
        for (int n = wave; n < P; n += 16) {
            float dist[kSlots];
#pragma unroll
            for (int t = 0; t < kSlots; ++t) dist[t] = 0.f;
            for (int c = 0; c < C; ++c) {                   // coordinates in order, all slots of the lane in flight
                const float q = qs[n * cs + c];
                // branch-free over the slots (a guard per slot puts every LDS read into a basic block of its own and the reads
                // are then waited for one by one: 34 us instead of 16 for the kNN of a 20-waypoint problem); slots beyond the
                // problem's samples read row 0 and are set to infinity below
#pragma unroll
                for (int t = 0; t < kSlots; ++t) {
                    const int s = t * 64 + lane;
                    const float df = xs[(s < ns ? s : 0) * cs + c] - q;
                    dist[t] = fmaf(df, df, dist[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < kSlots; ++t)
                if (!(t * 64 + lane < ns)) dist[t] = INFINITY;
            for (int r = 0; r < kSmK; ++r) {
                int res = -1;
                if (r < k) {
                    float bd = INFINITY;
                    int bi = 0x7fffffff;
#pragma unroll
                    for (int t = 0; t < kSlots; ++t)
                        if (dist[t] < bd) { bd = dist[t]; bi = t * 64 + lane; }
                    // lowest index among the samples at the minimum distance; no finite distance left: every lane still holds
                    // (inf, 0x7fffffff) and that comes out (dropped by the candidate check below)
                    const float gm = sm_wave_min(bd);
                    bi = sm_wave_min(bd == gm ? bi : 0x7fffffff);
                    res = bi;
#pragma unroll
                    for (int t = 0; t < kSlots; ++t)
                        if (t * 64 + lane == bi) dist[t] = INFINITY;
                }
                if (lane == 0) knn_l[n * kSmK + r] = res;
            }
        }
    }
    __syncthreads();
    const int nc = fits ? ncand : 0;
    // ---- candidates: caller edges with a path-node target, then the kNN edges (sample -> path node)
    for (int c = tid; c < nc; c += 1024) {
        int src, dst;
        if (c < ne) {
            src = c < 1024 ? pre_src : (int)p.edge_index[e0 + c];
            dst = c < 1024 ? pre_dst : (int)p.edge_index[(size_t)p.total_edges + e0 + c];
        } else {
            const int pn = (c - ne) / kSmK, r = (c - ne) - pn * kSmK;
            const int s = knn_l[pn * kSmK + r];
            src = (s >= 0 && s < ns) ? P + s : -1;
            dst = pn;
        }
        const bool ok = src >= 0 && src < M && dst >= 0 && dst < P;   // only rows < P of h are consumed
        key[c] = ok ? dst * M + src : 0x7fffffff;
    }
    __syncthreads();
    // ---- stable rank sort.  Keys are target * M + source: a candidate's rank is the number of valid candidates with a smaller
    // target (histogram over the targets + prefix) plus its rank among the candidates of ITS target -- and those are the
    // caller edges (scanned, four lanes per candidate) and the ten kNN candidates of that target, not all ne + 10 P.
    // Invalid candidates write nothing: the sorted array is pre-filled with the invalid key.
    for (int c = tid; c < nc; c += 1024) sorted[c] = 0x7fffffff;
    for (int c = tid; c < nc; c += 1024) {
        const int kc = key[c];
        if (kc != 0x7fffffff) atomicAdd(&cnt_l[kc / M], 1);
    }
    __syncthreads();
    if (tid < 64) {                                       // exclusive prefix over the targets (beg_l), one wave
        int run = 0;
        const int Pl = fits ? P : 0;
        for (int base = 0; base < Pl; base += 64) {
            const int i = base + tid;
            const int v = i < Pl ? cnt_l[i] : 0;
            int x = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(x, off);
                if (tid >= off) x += y;
            }
            if (i < Pl) beg_l[i] = run + x - v;
            run += __shfl(x, 63);
        }
    }
    __syncthreads();
    for (int base = 0; base < nc; base += 256) {
        const int c = base + (tid >> 2), part = tid & 3;
        int rank = 0;
        int kc = 0x7fffffff;
        if (c < nc) {
            kc = key[c];
            if (kc != 0x7fffffff) {
                const int dst = kc / M;
                const int lo = dst * M, hi = lo + M;        // keys of this target
                const int x0 = (int)((long long)ne * part / 4), x1 = (int)((long long)ne * (part + 1) / 4);
#pragma unroll 4
                for (int x = x0; x < x1; ++x) {
                    const int kx = key[x];
                    rank += (kx >= lo && kx < hi) && ((kx < kc) || (kx == kc && x < c));
                }
                for (int r = part; r < kSmK; r += 4) {      // kNN candidates of this target
                    const int x = ne + dst * kSmK + r;
                    const int kx = key[x];
                    rank += (kx != 0x7fffffff) && ((kx < kc) || (kx == kc && x < c));
                }
            }
        }
        rank += __shfl_xor(rank, 1, 64);
        rank += __shfl_xor(rank, 2, 64);
        if (kc != 0x7fffffff && part == 0) sorted[beg_l[kc / M] + rank] = kc;
    }
    __syncthreads();
    for (int i = tid; i < P && i < p.path_cap; i += 1024) { cnt_l[i] = 0; beg_l[i] = 0; }    // reused as the segment counters below
    __syncthreads();
    // ---- unique + compaction
    int carry = 0;
    for (int base = 0; base < nc; base += 1024) {
        const int c = base + tid;
        const int kc = (c < nc) ? sorted[c] : 0x7fffffff;
        const int keep = (c < nc) && kc != 0x7fffffff && (c == 0 || sorted[c - 1] != kc);
        int chunk_total;
        const int incl = sm_scan_1024(keep, s_wsum, chunk_total);
        if (keep) {
            const int pos = carry + incl - 1;
            const int dst = kc / M, src = kc - dst * M;
            p.e_src[eoff + pos] = src;
            p.e_dst[eoff + pos] = dst;
            atomicAdd(&cnt_l[dst], 1);
            if (c == 0 || sorted[c - 1] / M != dst) beg_l[dst] = eoff + pos;     // first edge of a target's run
        }
        carry += chunk_total;
    }
    __syncthreads();
    for (int i = tid; i < sm_round32(P); i += 1024) {
        const bool in = i < P && i < p.path_cap;
        p.seg_beg[poff + i] = in ? beg_l[i] : 0;
        p.seg_cnt[poff + i] = in ? cnt_l[i] : 0;
    }
    const int n = carry;
    if (tid == 0) { p.e_count[b] = n; p.stat[b] = fits ? 0 : 1; }
    // tile -> problem maps of this problem's whole capacity range (-1 = unused tile; no separate fill launch): the range
    // ends where the next problem's begins, the last problem's at the end of the tile space
    const int et_used = (eoff + sm_round32(n)) / 32, pt_used = (poff + sm_round32(P)) / 32;
    const int et_end = b + 1 < p.B ? sm_eoff(p, b + 1) / 32 : p.n_etiles;
    const int pt_end = b + 1 < p.B ? sm_poff(p, b + 1) / 32 : p.n_ptiles;
    for (int t = eoff / 32 + tid; t < et_end; t += 1024) p.etile_prob[t] = t < et_used ? b : -1;
    for (int t = poff / 32 + tid; t < pt_end; t += 1024) { p.ptile_prob[t] = t < pt_used ? b : -1; if (p.tgt_flag) p.tgt_flag[t] = 0; }
    sm_list_tiles(p, b, eoff / 32, et_used, poff / 32, pt_used, tid, 1024);
}

// node features [coords / scale (path rows are already scaled), one-hot(kind)]   model_smoother.py:130-135
struct SmNodeIn {
    const float* row;
    float inv_is_path;     // 1 -> row already divided by scale
    float scale;
    int C, kind;
    __device__ __forceinline__ float operator()(int k) const {
        if (k < C) return inv_is_path != 0.f ? row[k] : row[k] / scale;
        return (k - C == kind) ? 1.0f : 0.f;
    }
};

__device__ __forceinline__ SmNodeIn sm_node_in(const SmParams& p, int b, int n) {
    const int p0 = sm_pp(p, b), P = sm_pp(p, b + 1) - p0;
    const int f0 = sm_fp(p, b), F = sm_fp(p, b + 1) - f0;
    const int c0 = sm_cp(p, b);
    SmNodeIn in;
    in.C = p.C; in.scale = p.scale;
    if (n < P) { in.row = p.cur + (size_t)(p0 + n) * p.C; in.kind = 0; in.inv_is_path = 1.f; }
    else if (n < P + F) { in.row = p.free_pts + (size_t)(f0 + n - P) * p.C; in.kind = 1; in.inv_is_path = 0.f; }
    else { in.row = p.collided + (size_t)(c0 + n - P - F) * p.C; in.kind = 2; in.inv_is_path = 0.f; }
    return in;
}

// x = node_code.3( relu( BN_eval( node_code.0(in) ) ) ) with BN folded into node_code.0 at pack time
template <int NT, int P>
__device__ __forceinline__ void sm_node_code(const SmParams& p, const SmNodeIn& in, f32x16 (&x)[NT], int lane) {
    const float* W = p.w;
    f32x16 hdn[NT];
    load_vec<NT>(W + p.L.b0, hdn, lane);
    linear_in_p<P, NT>(W + p.L.as0, p.L.ks, in, hdn, lane);
    relu_<NT>(hdn);
    load_vec<NT>(W + p.L.b3, x, lane);
    linear_acc_p<P, NT, NT>(W + p.L.w3, hdn, x, lane);
}

// ---------------------------------------------------------------------------------------------------
// messages: one wave per 32-edge tile
// ---------------------------------------------------------------------------------------------------
template <int D, int P>
__global__ __launch_bounds__(256) void sm_msg_kernel(SmParams p) {
    constexpr int NT = D / 32;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= p.n_etiles) return;
    const int b = p.etile_prob[tile];
    if (b < 0) return;
    const int eoff = sm_eoff(p, b);
    const int e = tile * 32 + j;
    const bool valid = (e - eoff) < p.e_count[b];
    const int src = valid ? p.e_src[e] : 0, dst = valid ? p.e_dst[e] : 0;
    const float* W = p.w;
    f32x16 z[NT];
    load_vec<NT>(W + p.L.b00, z, lane);
    {
        f32x16 x[NT];
        sm_node_code<NT, P>(p, sm_node_in(p, b, dst), x, lane);          // x_i (target)
        linear_acc_p<P, NT, NT>(W + p.L.wdst, x, z, lane);                  // (W_c - W_a) x_i
    }
    {
        f32x16 x[NT];
        sm_node_code<NT, P>(p, sm_node_in(p, b, src), x, lane);          // x_j (source)
        linear_acc_p<P, NT, NT>(W + p.L.wsrc, x, z, lane);                  // (W_a + W_b) x_j
    }
    relu_<NT>(z);
    f32x16 m[NT];
    load_vec<NT>(W + p.L.b02, m, lane);
    linear_acc_p<P, NT, NT>(W + p.L.w02, z, m, lane);
    store_row<NT>(p.msg + (size_t)e * D, m, h);
}

// ---------------------------------------------------------------------------------------------------
// path-node update: one wave per 32 path nodes
// ---------------------------------------------------------------------------------------------------
template <int D, int P>
__global__ __launch_bounds__(256) void sm_node_kernel(SmParams p) {
    constexpr int NT = D / 32;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= p.n_ptiles) return;
    const int b = p.ptile_prob[tile];
    if (b < 0) return;
    const int poff = sm_poff(p, b);
    const int p0 = sm_pp(p, b), PN = sm_pp(p, b + 1) - p0;
    const int n = tile * 32 + j - poff;                  // local path index
    const bool valid = n < PN;
    const float* W = p.w;
    f32x16 S[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) S[t] = splat16(0.f);
    if (valid) {
        const int beg = p.seg_beg[poff + n], cnt = p.seg_cnt[poff + n];
        for (int i = 0; i < cnt; ++i) {                  // coalesced edge order: increasing source id
            f32x16 m[NT];
            load_row<NT>(p.msg + (size_t)(beg + i) * D, m, h);
#pragma unroll
            for (int t = 0; t < NT; ++t) S[t] += m[t];
        }
    }
    f32x16 y[NT];
    {
        f32x16 hdn[NT];
        load_vec<NT>(W + p.L.b10, hdn, lane);
        linear_acc_p<P, NT, NT>(W + p.L.w10, S, hdn, lane);
        relu_<NT>(hdn);
        load_vec<NT>(W + p.L.b12, y, lane);
        linear_acc_p<P, NT, NT>(W + p.L.w12, hdn, y, lane);
    }
    {
        f32x16 x[NT];
        sm_node_code<NT, P>(p, sm_node_in(p, b, valid ? n : 0), x, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) y[t] += x[t];        // h = x + lin_1(S)      model_smoother.py:34
    }
    f32x16 o[1];
    load_vec<1>(W + p.L.bs, o, lane);
    linear_acc_p<P, 1, NT>(W + p.L.ws, y, o, lane);            // smooth_node (out features padded to 32)
    const bool interior = valid && n >= 1 && n <= PN - 2;  // path[1:-1] = ...       model_smoother.py:139
    if (interior || (valid && p.out != nullptr)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = phi(r, h);
            if (f < p.C) {
                const size_t at = (size_t)(p0 + n) * p.C + f;
                const float val = interior ? o[0][r] : p.cur[at];
                if (interior) p.cur_next[at] = val;
                if (p.out) p.out[at] = val * p.scale;      // last iteration: out = path_cur * scale (model_smoother.py:142)
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// A tile's layers are a chain of ~1300 exact-fp32 MFMAs for one wave (34 us at 64 cycles each; the reference's call is ONE
// problem, ~9 edge tiles and one path tile).  Here a tile belongs to a WORKGROUP of NT waves and
// wave w computes output tile w (32 features) of every layer, i.e. a quarter of the MFMAs at d = 128; the waves
// exchange their tiles through LDS between layers (all waves share the row <-> lane mapping, so the exchange is a
// plain register dump and reload).  Every output element is accumulated in the same order as in the one-wave kernels:
// the results are bit-identical.
// ---------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void sm_exchange_get(const float* buf, f32x16 (&all)[NT], int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) all[t][r] = buf[(t * 16 + r) * 64 + lane];
}
__device__ __forceinline__ void sm_exchange_put(float* buf, int wave, const f32x16& mine, int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[(wave * 16 + r) * 64 + lane] = mine[r];
    __syncthreads();
}
template <int NT>
__device__ __forceinline__ void sm_exchange(float* buf, int wave, const f32x16& mine, f32x16 (&all)[NT], int lane) {
    sm_exchange_put(buf, wave, mine, lane);
    sm_exchange_get<NT>(buf, all, lane);
}

// tile `wave` of x = node_code(in): layer 0 (a few MFMAs from the raw inputs) is computed whole by every wave
template <int NT, int P>
__device__ __forceinline__ void sm_node_code_tile(const SmParams& p, const SmNodeIn& in, int wave, f32x16& x, int lane) {
    const float* W = p.w;
    f32x16 hdn[NT];
    load_vec<NT>(W + p.L.b0, hdn, lane);
    linear_in_p<P, NT>(W + p.L.as0, p.L.ks, in, hdn, lane);
    relu_<NT>(hdn);
    f32x16 y[1];
    load_vec<1>(W + p.L.b3 + wave * 32, y, lane);
    linear_acc_p<P, 1, NT>(W + p.L.w3 + (size_t)wave * NT * Prec<P>::TF, hdn, y, lane);
    x = y[0];
}

// Target role.  The first-layer term of a message that belongs to the TARGET, b00 + (W_c - W_a) x_i, is the same for every
// incoming edge of a path node (~13 of them), and it is 40 % of an edge tile's MFMAs (node_code of the target + one d x d
// layer).  The first n_ptiles workgroups of the launch compute it once per path node (one 32-node tile each, same chain,
// same accumulation order) and publish the rows; an edge tile computes its source half first, then takes its rows from
// there -- the accumulator it continues with is bit for bit the one it would have computed itself.  Target workgroups have
// the lowest block indices, i.e. they are resident before any edge workgroup; should a flag nevertheless not show up within
// the polling budget, the edge tile computes the target half itself (same bits), so the wait can never hang a launch.
// Registers: without a bound the compiler spreads the chain's operand prefetch over 94 VGPRs + 80 AGPRs (two workgroups per
// CU); asked for four workgroups per CU it fits 124 without spilling.
#ifdef GNNMP_SM_TRACE
// diagnostics build: eight 100 MHz stamps per workgroup of the LAST split message launch (tools/diag/sm_trace.py)
__device__ long long g_sm_trace[8 * 8192];
#define SM_TRC(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_sm_trace[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define SM_TRC(k) do {} while (0)
#endif
template <int D, int P>
__global__ __launch_bounds__(D * 2, D == 128 ? 4 : 1) void sm_msg_split_kernel(SmParams p) {
    constexpr int NT = D / 32;
    __shared__ float xbuf[2][NT * 16 * 64];
    __shared__ int s_ready;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const float* W = p.w;
    constexpr size_t TF = Prec<P>::TF;
    const int n_troles = p.tgt_flag ? p.n_ptiles : 0;
    const bool trole = (int)blockIdx.x < n_troles;       // workgroup-uniform
    int b, src = 0, dst = 0, e = 0, trow = 0;
    if (trole) {
        const int tile = blockIdx.x;
        b = p.ptile_prob[tile];
        if (b < 0) return;
        const int PN = sm_pp(p, b + 1) - sm_pp(p, b);
        const int n = tile * 32 + j - sm_poff(p, b);     // local path index
        dst = n < PN ? n : 0;
        trow = tile * 32 + j;
    } else {
        const int tile = blockIdx.x - n_troles;
        b = p.etile_prob[tile];
        if (b < 0) return;
        e = tile * 32 + j;
        const bool valid = (e - sm_eoff(p, b)) < p.e_count[b];
        src = valid ? p.e_src[e] : 0;
        dst = valid ? p.e_dst[e] : 0;
        trow = sm_poff(p, b) + dst;                      // the target's row in the padded path space
    }
    bool target_half = trole;                            // does this workgroup compute b00 + (W_c - W_a) x_i itself?
#ifdef GNNMP_SM_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        for (int i = 0; i < 8; ++i) g_sm_trace[blockIdx.x * 8 + i] = 0;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_sm_trace[blockIdx.x * 8 + 6] = (long long)hw | ((long long)xcc << 32);
        g_sm_trace[blockIdx.x * 8 + 7] = trole ? 1 : 0;
    }
#endif
    SM_TRC(0);
    if (!trole) {
        // source half first: it does not depend on the target rows.  The tile stays in xbuf[1] until it is needed
        f32x16 mine;
        sm_node_code_tile<NT, P>(p, sm_node_in(p, b, src), wave, mine, lane);      // x_j (source)
        sm_exchange_put(xbuf[1], wave, mine, lane);
        SM_TRC(1);
        if (n_troles > 0) {
            if (wave == 0) {                             // the flags of the path tiles this edge tile's targets live in
                bool ready = false;
                for (int spin = 0; spin < 256 && !ready; ++spin) {
                    // relaxed: an acquire here would invalidate the CU's L1 (every wave's weight operands) on every poll; the
                    // one acquire fence below, after the flags are up, is what orders the row loads
                    const int f = __hip_atomic_load(&p.tgt_flag[trow >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ready = __all(f != 0);
                    if (!ready) __builtin_amdgcn_s_sleep(8);
                }
                if (lane == 0) s_ready = ready ? 1 : 0;
            }
            __syncthreads();
            target_half = s_ready == 0;                  // the rare path: a flag did not show up in time
            SM_TRC(2);
        } else {
            target_half = true;
        }
    }
    f32x16 z[1];
    if (target_half) {                                   // workgroup-uniform; ONE copy of this code for both roles
        f32x16 mine, x[NT];
        load_vec<1>(W + p.L.b00 + wave * 32, z, lane);
        sm_node_code_tile<NT, P>(p, sm_node_in(p, b, dst), wave, mine, lane);      // x_i (target)
        sm_exchange<NT>(xbuf[0], wave, mine, x, lane);
        linear_acc_p<P, 1, NT>(W + p.L.wdst + (size_t)wave * NT * TF, x, z, lane);   // (W_c - W_a) x_i
        if (trole) {
            SM_TRC(3);
            store_row<1>(p.tgt + (size_t)trow * D + wave * 32, z, h);
            __threadfence();                             // the rows are visible device-wide before the flag is
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(&p.tgt_flag[blockIdx.x], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            SM_TRC(5);
            return;
        }
        __syncthreads();                                 // every wave has read xbuf[0] before the z exchange below reuses it
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        load_row<1>(p.tgt + (size_t)trow * D + wave * 32, z, h);
    }
    SM_TRC(3);
    {
        f32x16 xs[NT];
        sm_exchange_get<NT>(xbuf[1], xs, lane);
        linear_acc_p<P, 1, NT>(W + p.L.wsrc + (size_t)wave * NT * TF, xs, z, lane);  // (W_a + W_b) x_j
    }
    relu_<1>(z);
    SM_TRC(4);
    f32x16 zall[NT], m[1];
    sm_exchange<NT>(xbuf[0], wave, z[0], zall, lane);
    load_vec<1>(W + p.L.b02 + wave * 32, m, lane);
    linear_acc_p<P, 1, NT>(W + p.L.w02 + (size_t)wave * NT * TF, zall, m, lane);
    store_row<1>(p.msg + (size_t)e * D + wave * 32, m, h);
    SM_TRC(5);
}

// ---------------------------------------------------------------------------------------------------
// Large batches, exact-fp32 operands: the message kernel around weights that pass through LDS.
//
// The split kernel above reads every MFMA's A operand (256 bytes) straight from L2: 0.19 MB of matrices per edge tile, issued
// a register budget's worth ahead, i.e. a fresh L2 round trip per ~16 MFMAs in four waves that meet at a barrier after every
// layer -- the matrix pipe is 35 % busy and the waves wait 55 % of their life (profiles/r05_pmc_smoother_14d.txt).  Here a
// workgroup is EIGHT waves (two per SIMD, one workgroup per CU), every wave owns one 32-row tile and computes ALL d outputs of
// every layer (the tile-per-wave chain of sm_msg_kernel: same accumulation order, bit-identical), and the workgroup walks the
// layers together: a layer's d x d matrix comes through LDS one input-tile column at a time (NT tiles = 16 KB in fp32), the
// next column in flight (LDS-DMA, every wave issues its share) while the eight waves multiply the current one from LDS
// (ds_read_b128, conflict-free: the packed tile layout is lane-contiguous).  Two 16 KB slots; one barrier per column.
//
// Roles: the first `wgs_t` workgroups are TARGET workgroups -- two path tiles each, four waves per tile in the split form above (wave w
// computes output tile w of node_code and of (W_c - W_a) x_i, operands from L2, tiles exchanged through LDS): a quarter of the chain
// per wave, so the rows b00 + (W_c - W_a) x_i are published (p.tgt + flags) ~25 k cycles into the launch, before any edge workgroup has
// finished its source half (55 k).  A first version ran the targets as streamed rounds of their own: a 99 k-cycle chain that every
// edge round then waited 40 k cycles for (tools/diag/sm_stream_trace.py).  The other workgroups are EDGE workgroups and take edge
// rounds r = blockIdx.x - wgs_t, + (gridDim.x - wgs_t), ... of eight 32-edge tiles, streaming the matrices as described.  An edge
// round waits for its targets' flags after the source half (bounded), and if one does not show up the whole workgroup computes the
// target half itself -- same bits, so the wait can never hang a launch (another stream may keep the target workgroups off the device).
// ---------------------------------------------------------------------------------------------------
template <int D, int P>
__global__ __launch_bounds__(512, 1) void sm_msg_stream_kernel(SmParams p, int wgs_t, int rounds_e) {
    constexpr int NT = D / 32;
    constexpr int TF = Prec<P>::TF;
    constexpr int CH = NT * TF;                           // floats of one column (NT tiles) of a packed matrix
    constexpr int PER_TILE = TF / 256, PIECES = NT * PER_TILE;      // 1 KB LDS-DMA pieces per tile / per column
    extern __shared__ __attribute__((aligned(16))) float lds[];     // two column slots
    __shared__ int s_notready;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const float* W = p.w;
#ifdef GNNMP_SM_TRACE
    // diagnostics build: wave 0's shader-clock cycles per phase of a column (vmcnt wait / barrier / DMA issue + reads + MFMAs), summed
    long long trc_acc[4] = {0, 0, 0, 0}, trc_t[4] = {0, 0, 0, 0}, trc_m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SM_STRC_M(k) do { trc_m[k] = clock64(); } while (0)
    const long long trc_begin = clock64();
#define SM_STRC_T(k) do { trc_t[k] = clock64(); if ((k) == 3) { trc_acc[0] += trc_t[1] - trc_t[0]; trc_acc[1] += trc_t[2] - trc_t[1]; trc_acc[2] += trc_t[3] - trc_t[2]; ++trc_acc[3]; } } while (0)
#else
#define SM_STRC_T(k) do {} while (0)
#define SM_STRC_M(k) do {} while (0)
#endif
    int step = 0;                                         // columns consumed so far (workgroup-uniform): slot = step & 1
    auto issue = [&](const float* Wm, int it, int sl) {   // column `it` of packed matrix Wm -> slot sl, this wave's share
        for (int q = wave; q < PIECES; q += 8) {
            const int ot = q / PER_TILE, piece = q % PER_TILE;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wm + (size_t)(ot * NT + it) * TF + piece * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(lds + sl * CH + ot * TF + piece * 256), 16, 0, 0);
        }
    };
    // y[ot] += sum_it Wm[ot][it] x[it], it outer / ot inner like linear_acc_ops.  primed: column 0 of Wm is already on its way
    // to slot step & 1 (issued by the previous layer's last column).  Wnext: the matrix whose column 0 is requested behind
    // this layer's last column (nullptr: nothing).
    // compute = false (wave-uniform): a wave without a tile streams its share and meets the barriers but leaves the matrix pipe alone
    auto layer = [&](const float* Wm, bool primed, const float* Wnext, const BOp<P> (&xb)[NT], f32x16 (&y)[NT], bool compute) {
        if (!primed) issue(Wm, 0, step & 1);
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            SM_STRC_T(0);
            __builtin_amdgcn_s_waitcnt(7 << 4);           // vmcnt(0) lgkmcnt(0): this wave's pieces of the column have landed
            asm volatile("" ::: "memory");
            SM_STRC_T(1);
            __builtin_amdgcn_s_barrier();                 // everyone's have; everyone is done reading the other slot
            asm volatile("" ::: "memory");
            SM_STRC_T(2);
            if (it + 1 < NT) issue(Wm, it + 1, (step + 1) & 1);
            else if (Wnext) issue(Wnext, 0, (step + 1) & 1);
            if (compute) {
                // The operand reads are issued BY HAND: the next column's LDS-DMA has just been requested, the compiler cannot tell
                // that its destination (the other slot) and these reads never alias, and guards every LDS read it generates behind
                // an in-flight DMA with s_waitcnt vmcnt(0) -- each column then waited out the L2 round trip of the NEXT one before
                // its first MFMA (measured: 19 k instead of 8 k cycles per column, the matrix pipe 35 % busy).  Instructions inside
                // asm statements get no such guard; the waits they need are stated here.  Tile ot + 1's four 16-byte reads are in
                // flight while tile ot's sixteen MFMAs issue.
                static_assert(P == 0, "hand-issued operand reads: exact-fp32 tiles (4 x 16 bytes per lane)");
                const unsigned abase = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds + (unsigned)((step & 1) * CH * 4) + 16u * (unsigned)lane;
                f32x4 w[2][4];
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                             : "=&v"(w[0][0]), "=&v"(w[0][1]), "=&v"(w[0][2]), "=&v"(w[0][3]) : "v"(abase) : "memory");
#pragma unroll
                for (int ot = 0; ot < NT; ++ot) {
                    if (ot + 1 < NT) {
                        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                                     "s_waitcnt lgkmcnt(4)"
                                     : "=&v"(w[(ot + 1) & 1][0]), "=&v"(w[(ot + 1) & 1][1]), "=&v"(w[(ot + 1) & 1][2]), "=&v"(w[(ot + 1) & 1][3]),
                                       "+v"(w[ot & 1][0]), "+v"(w[ot & 1][1]), "+v"(w[ot & 1][2]), "+v"(w[ot & 1][3])
                                     : "v"(abase + (unsigned)((ot + 1) * TF * 4)) : "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[ot & 1][0]), "+v"(w[ot & 1][1]), "+v"(w[ot & 1][2]), "+v"(w[ot & 1][3]) :: "memory");
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            y[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ot & 1][q][c], xb[it].v[q * 4 + c], y[ot], 0, 0, 0);
                }
            }
            ++step;
            SM_STRC_T(3);
        }
    };
    // x = node_code(in): layer 0 from the raw inputs (a few MFMAs, weights from L2), layer 3 streamed
    auto node_code = [&](const SmNodeIn& in, bool primed, const float* Wnext, f32x16 (&x)[NT], bool compute) {
        f32x16 hdn[NT];
        load_vec<NT>(W + p.L.b0, hdn, lane);
        if (compute) linear_in_p<P, NT>(W + p.L.as0, p.L.ks, in, hdn, lane);
        relu_<NT>(hdn);
        BOp<P> hb[NT];
        make_ops<P, NT>(hdn, hb);
        load_vec<NT>(W + p.L.b3, x, lane);
        layer(W + p.L.w3, primed, Wnext, hb, x, compute);
    };
    if ((int)blockIdx.x < wgs_t) {
        // ---------------- target workgroup: path tiles 2 blockIdx.x + (wave >> 2), output tile wave & 3 (sm_msg_split_kernel's target role)
        const int grp = wave >> 2, w4 = wave & 3;
        const int slot_t = blockIdx.x * 2 + grp;
        const bool act = slot_t < p.tile_cnt[p.parity * 2 + 1];
        const int tile = act ? p.plist[slot_t] : 0;
        if (__syncthreads_or(act ? 1 : 0) == 0) return;    // both tiles unused (the tile space is padded per problem)
        const int b = act ? p.ptile_prob[tile] : 0;
        const int PN = sm_pp(p, b + 1) - sm_pp(p, b);
        const int n = tile * 32 + j - sm_poff(p, b);
        const int dst = (act && n >= 0 && n < PN) ? n : 0;
        float* xbuf = lds + grp * (NT * 16 * 64);          // the column slots are not in use here: 16 KB per group for the exchange
        constexpr size_t TFs = TF;
        f32x16 mine, x[NT], z[1];
        load_vec<1>(W + p.L.b00 + w4 * 32, z, lane);
        sm_node_code_tile<NT, P>(p, sm_node_in(p, b, dst), w4, mine, lane);         // x_i (target), this wave's 32 features
        sm_exchange<NT>(xbuf, w4, mine, x, lane);
        linear_acc_p<P, 1, NT>(W + p.L.wdst + (size_t)w4 * NT * TFs, x, z, lane);   // (W_c - W_a) x_i
        if (act) store_row<1>(p.tgt + (size_t)(tile * 32 + j) * D + w4 * 32, z, h);
        __threadfence();                                   // the rows are visible device-wide before the flag is
        __syncthreads();
        if (act && w4 == 0 && lane == 0) __hip_atomic_store(&p.tgt_flag[tile], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int wgs_e = (int)gridDim.x - wgs_t;
    const int n_e = p.tile_cnt[p.parity * 2 + 0];        // edge tiles in use (listed by the graph stage); rounds_e is the launch's upper bound
    rounds_e = (n_e + 7) >> 3;
    bool primed = false;
    for (int r = (int)blockIdx.x - wgs_t; r < rounds_e; r += wgs_e) {
        const bool last = r + wgs_e >= rounds_e;
        // -------------------- edge round r: edge tiles 8 r + wave
        SM_STRC_M(0);
        const bool active = r * 8 + wave < n_e;
        const int tile = active ? p.elist[r * 8 + wave] : 0;
        const int b = active ? p.etile_prob[tile] : 0;
        const int e = tile * 32 + j;
        const bool valid = active && (e - sm_eoff(p, b)) < p.e_count[b];
        const int src = valid ? p.e_src[e] : 0, dst = valid ? p.e_dst[e] : 0;
        const int trow = sm_poff(p, b) + dst;             // the target's row in the padded path space
        if (threadIdx.x == 0) s_notready = 0;             // (ordered before the reads below by the barriers of the source half)
        f32x16 xs[NT], z[NT];
#ifdef GNNMP_SM_TRACE
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(src), "v"(dst) : "memory");
#endif
        SM_STRC_M(1);
        node_code(sm_node_in(p, b, src), primed, W + p.L.wsrc, xs, active);  // x_j (source); wsrc's first column follows
        SM_STRC_M(2);
        {
            bool ready = !active;
            for (int spin = 0; spin < 4096 && !ready; ++spin) {
                // relaxed: an acquire here would invalidate the CU's L1 on every poll; the one acquire fence below orders the row loads
                const int f = __hip_atomic_load(&p.tgt_flag[trow >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ready = __all(f != 0);
                if (!ready) __builtin_amdgcn_s_sleep(8);
            }
            if (!ready && lane == 0) s_notready = 1;
        }
        __builtin_amdgcn_s_waitcnt(7 << 4);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool fallback = s_notready != 0;            // workgroup-uniform; the rare path
        bool wsrc_primed = true;
        SM_STRC_M(3);
        if (fallback) {
            // the target half here: node_code(dst) and (W_c - W_a) x_i through the same stream (wsrc's prefetched column is dropped:
            // the column counter moves on, its slot is simply overwritten)
            __builtin_amdgcn_s_waitcnt(7 << 4);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            ++step;                                        // the slot wsrc's column went to counts as consumed
            f32x16 xd[NT];
            node_code(sm_node_in(p, b, dst), false, W + p.L.wdst, xd, active);
            BOp<P> xb[NT];
            make_ops<P, NT>(xd, xb);
            load_vec<NT>(W + p.L.b00, z, lane);
            layer(W + p.L.wdst, true, W + p.L.wsrc, xb, z, active);
        } else {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (active) load_row<NT>(p.tgt + (size_t)trow * D, z, h);
            else {
#pragma unroll
                for (int t = 0; t < NT; ++t) z[t] = splat16(0.f);
            }
        }
        {
            BOp<P> xb[NT];
            make_ops<P, NT>(xs, xb);
            SM_STRC_M(4);
            layer(W + p.L.wsrc, wsrc_primed, W + p.L.w02, xb, z, active);    // += (W_a + W_b) x_j
        }
        SM_STRC_M(5);
        relu_<NT>(z);
        f32x16 m[NT];
        {
            BOp<P> zb[NT];
            make_ops<P, NT>(z, zb);
            load_vec<NT>(W + p.L.b02, m, lane);
            layer(W + p.L.w02, true, last ? nullptr : W + p.L.w3, zb, m, active);
        }
        primed = !last;
        if (active) store_row<NT>(p.msg + (size_t)e * D, m, h);
        SM_STRC_M(6);
    }
#ifdef GNNMP_SM_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        long long* t = g_sm_trace + blockIdx.x * 16;      // two 8-slot records per workgroup
        t[0] = trc_begin; t[1] = trc_acc[0]; t[2] = trc_acc[1]; t[3] = trc_acc[2]; t[4] = trc_acc[3]; t[5] = clock64(); t[6] = 0; t[7] = 2;
        for (int i = 0; i < 7; ++i) t[8 + i] = trc_m[i];
        t[15] = 3;
    }
#endif
}

template <int D, int P>
__global__ __launch_bounds__(D * 2) void sm_node_split_kernel(SmParams p) {
    constexpr int NT = D / 32;
    __shared__ float xbuf[2][NT * 16 * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int tile = blockIdx.x;
    const int b = p.ptile_prob[tile];
    if (b < 0) return;
    const int poff = sm_poff(p, b);
    const int p0 = sm_pp(p, b), PN = sm_pp(p, b + 1) - p0;
    const int n = tile * 32 + j - poff;                  // local path index
    const bool valid = n < PN;
    const float* W = p.w;
    constexpr size_t TF = Prec<P>::TF;
    f32x16 S[NT];
    {
        f32x16 mine = splat16(0.f);
        if (valid) {
            const int beg = p.seg_beg[poff + n], cnt = p.seg_cnt[poff + n];
            for (int i = 0; i < cnt; ++i) {              // coalesced edge order: increasing source id
                f32x16 m[1];
                load_row<1>(p.msg + (size_t)(beg + i) * D + wave * 32, m, h);
                mine += m[0];
            }
        }
        sm_exchange<NT>(xbuf[0], wave, mine, S, lane);
    }
    f32x16 y[1];
    {
        f32x16 hw[1], hdn[NT];
        load_vec<1>(W + p.L.b10 + wave * 32, hw, lane);
        linear_acc_p<P, 1, NT>(W + p.L.w10 + (size_t)wave * NT * TF, S, hw, lane);
        relu_<1>(hw);
        sm_exchange<NT>(xbuf[1], wave, hw[0], hdn, lane);
        load_vec<1>(W + p.L.b12 + wave * 32, y, lane);
        linear_acc_p<P, 1, NT>(W + p.L.w12 + (size_t)wave * NT * TF, hdn, y, lane);
    }
    {
        f32x16 x;
        sm_node_code_tile<NT, P>(p, sm_node_in(p, b, valid ? n : 0), wave, x, lane);
        y[0] += x;                                        // h = x + lin_1(S)      model_smoother.py:34
    }
    f32x16 yall[NT];
    sm_exchange<NT>(xbuf[0], wave, y[0], yall, lane);
    if (wave != 0) return;                                // smooth_node: one output tile, accumulated in tile order
    f32x16 o[1];
    load_vec<1>(W + p.L.bs, o, lane);
    linear_acc_p<P, 1, NT>(W + p.L.ws, yall, o, lane);
    const bool interior = valid && n >= 1 && n <= PN - 2;  // path[1:-1] = ...       model_smoother.py:139
    if (interior || (valid && p.out != nullptr)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = phi(r, h);
            if (f < p.C) {
                const size_t at = (size_t)(p0 + n) * p.C + f;
                const float val = interior ? o[0][r] : p.cur[at];
                if (interior) p.cur_next[at] = val;
                if (p.out) p.out[at] = val * p.scale;      // last iteration: out = path_cur * scale (model_smoother.py:142)
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
#define LAUNCH_CHECK()                        \
    do {                                      \
        hipError_t _e = hipGetLastError();    \
        if (_e != hipSuccess) return _e;      \
    } while (0)

hipError_t launch_sm_init(int n, float scale, const float* path, float* cur, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sm_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, scale, path, cur);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_sm_final(int n, float scale, const float* cur, float* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sm_final_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, scale, cur, out);
    LAUNCH_CHECK();
    return hipSuccess;
}

// exact-fp32 operands, d = 128: the streamed-weights message kernel from this many 32-edge tiles of CAPACITY on (~1000 problems of 20
// waypoints).  Measured, C = 14, 20 waypoints, 500 + 500 samples, whole forward, split / streamed: 256 problems 0.247 / 0.248 ms, 512
// 0.426 / 0.431, 1024 0.779 / 0.745, 2048 1.527 / 1.367, 4096 3.026 / 2.638 (tools/diag/sm_stream_sizes.py).  Below ~1000 problems the
// eight-tile rounds quantise badly on 256 CUs (256 problems = 2304 tiles in use = 288 rounds: 32 workgroups run a second round while
// 224 idle) and a round's fixed latencies (tile look-ups, layer 0 from L2, target rows: ~45 k of its ~150 k cycles,
// tools/diag/sm_stream_trace.py) have nothing to hide behind; the split kernel's four one-tile workgroups per CU hide them behind each other.
constexpr int kSmStreamMinTiles = 9000;
// (Also built and measured: the SPLIT form with streamed weights -- two tiles per eight-wave workgroup, four waves per tile, columns
// through LDS, two workgroups per CU.  Bit-identical and slower than the split kernel at every size: 64 problems 0.135 vs 0.119 ms,
// 256 0.276 vs 0.245, 2048 1.725 vs 1.527 -- sixteen MFMAs per wave between barriers do not cover a column's LDS-DMA, and the form
// spills at 128 registers.  Not kept.  A third form -- LAYER-synchronous: four split-form tiles per sixteen-wave workgroup, a layer's
// whole 64 KB matrix copied into LDS once for all four, no barrier inside a layer, five workgroup-wide barriers per tile set -- was
// also bit-identical and also slower: 16 problems 0.118 vs 0.080 ms, 256 0.286 vs 0.238, 2048 1.723 vs 1.510.  Every form that makes
// the waves of a CU walk the layers in lockstep exposes a tile's fixed latencies (look-ups, layer 0, flags, exchanges) to all of them
// at once; four independent one-tile workgroups per CU with operands from L2 hide them behind each other, and that is worth more than
// the operand latency it costs.)
constexpr int kSmSplitMaxTilesBf16 = 2048;   // bf16 operands: the split kernels up to this many 32-edge tiles

// graph stage of one iteration: the one-launch form when a problem's staged samples, path rows, neighbour ids, sort
// buffers and segment counters fit the LDS of a workgroup, else kNN and edge list as two launches
static size_t sm_graph_lds_bytes(const SmParams& p) {
    const size_t cs = (size_t)(p.C | 1);
    return sizeof(float) * cs * ((size_t)p.samp_cap + p.path_cap) + sizeof(int) * ((size_t)p.path_cap * (kSmK + 2) + 2 * (size_t)p.cand_cap);
}
static hipError_t launch_sm_graph(const SmParams& p, hipStream_t st) {
    static const int fused_env = getenv("GNNMP_SM_FUSED_GRAPH") ? atoi(getenv("GNNMP_SM_FUSED_GRAPH")) : 1;
    const size_t lds = sm_graph_lds_bytes(p);
    if (fused_env != 0 && lds <= 150 * 1024) {
        // the raised dynamic-LDS limit is a property of (kernel, CURRENT device): set on every launch like the explorer's
        // set_lds (a process that drives a second GPU would otherwise launch there with the default 64 KB limit and fail as
        // soon as a problem needs more).  If the attribute cannot be set the two-launch form below runs instead.
        const void* fn = p.samp_cap <= 1024 ? reinterpret_cast<const void*>(sm_graph_kernel<16>) : reinterpret_cast<const void*>(sm_graph_kernel<32>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess) {
            if (p.samp_cap <= 1024) hipLaunchKernelGGL(sm_graph_kernel<16>, dim3(p.B), dim3(1024), lds, st, p);
            else hipLaunchKernelGGL(sm_graph_kernel<32>, dim3(p.B), dim3(1024), lds, st, p);
            LAUNCH_CHECK();
            return hipSuccess;
        }
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(sm_knn_kernel, dim3((p.total_path + 3) / 4), dim3(256), 0, st, p);
    LAUNCH_CHECK();
    const size_t lds2 = (size_t)2 * p.cand_cap * sizeof(int);
    hipLaunchKernelGGL(sm_edges_kernel, dim3(p.B), dim3(256), lds2, st, p);
    LAUNCH_CHECK();
    return hipSuccess;
}

template <int D, int P>
static hipError_t launch_sm_iter_t(const SmParams& p, hipStream_t st) {
    {
        hipError_t e = launch_sm_graph(p, st);
        if (e != hipSuccess) return e;
    }
    // one tile per workgroup, its layers split over D / 32 waves (bit-identical, a quarter of the chain depth): measured
    // faster at EVERY batch size with exact-fp32 MFMAs (1 problem 143 -> 88 us, 256 problems 313 -> 238 us, 2048 problems
    // 1.67 -> 1.43 ms) and up to ~250 problems with bf16 operands (beyond that the tile-per-wave kernels win by 5-10 %)
    static const int split_env = getenv("GNNMP_SM_SPLIT") ? atoi(getenv("GNNMP_SM_SPLIT")) : -1;
    const bool split = D >= 64 && (split_env >= 0 ? split_env != 0 : (P == 0 || p.n_etiles <= kSmSplitMaxTilesBf16));
    // large fp32 batches at d = 128: the streamed-weights message kernel (eight waves, a tile per wave, weights through LDS)
    static const int stream_env = getenv("GNNMP_SM_STREAM") ? atoi(getenv("GNNMP_SM_STREAM")) : -1;
    if constexpr (D == 128 && P == 0) {
        if (p.tgt_flag && p.tile_cnt && (stream_env >= 0 ? stream_env != 0 : p.n_etiles >= kSmStreamMinTiles)) {
            const int wgs_t = (p.n_ptiles + 1) / 2, rounds_e = (p.n_etiles + 7) / 8;
            static int cus = 0;
            if (!cus) {
                int dev = 0;
                hipDeviceProp_t prop;
                cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
            }
            const size_t lds = (size_t)2 * (D / 32) * Prec<P>::TF * sizeof(float);
            static std::once_flag once;
            std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sm_msg_stream_kernel<D, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
            // target workgroups first (lowest block ids: dispatched first, gone after ~10 us), then one resident edge workgroup per CU
            const int grid = wgs_t + (rounds_e < cus ? rounds_e : cus);
            hipLaunchKernelGGL((sm_msg_stream_kernel<D, P>), dim3(grid), dim3(512), lds, st, p, wgs_t, rounds_e);
            LAUNCH_CHECK();
            hipLaunchKernelGGL((sm_node_split_kernel<D, P>), dim3(p.n_ptiles), dim3(D * 2), 0, st, p);
            LAUNCH_CHECK();
            return hipSuccess;
        }
    }
    if (split) {
        hipLaunchKernelGGL((sm_msg_split_kernel<D, P>), dim3(p.n_etiles + (p.tgt_flag ? p.n_ptiles : 0)), dim3(D * 2), 0, st, p);
        LAUNCH_CHECK();
        hipLaunchKernelGGL((sm_node_split_kernel<D, P>), dim3(p.n_ptiles), dim3(D * 2), 0, st, p);
        LAUNCH_CHECK();
        return hipSuccess;
    }
    hipLaunchKernelGGL((sm_msg_kernel<D, P>), dim3((p.n_etiles + 3) / 4), dim3(256), 0, st, p);
    LAUNCH_CHECK();
    hipLaunchKernelGGL((sm_node_kernel<D, P>), dim3((p.n_ptiles + 3) / 4), dim3(256), 0, st, p);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_sm_knn_edges(const SmParams& p, hipStream_t st) { return launch_sm_graph(p, st); }

// Does a forward of this shape use the compact tile lists (the streamed-weights message kernel)?  Only then does the host arm the
// counters and hand the graph stage the list pointers: the bookkeeping (one memset per forward, two atomics and a barrier per
// problem) cost the small and the bf16 batches 3-6 % when it ran unconditionally.
bool sm_wants_tile_lists(int D, int P, int n_etiles) {
    static const int stream_env = getenv("GNNMP_SM_STREAM") ? atoi(getenv("GNNMP_SM_STREAM")) : -1;
    return D == 128 && P == 0 && (stream_env >= 0 ? stream_env != 0 : n_etiles >= kSmStreamMinTiles);
}

hipError_t launch_sm_iter(int D, int P, const SmParams& p, hipStream_t st) {
    switch (D) {
        case 32: return P ? launch_sm_iter_t<32, 1>(p, st) : launch_sm_iter_t<32, 0>(p, st);
        case 64: return P ? launch_sm_iter_t<64, 1>(p, st) : launch_sm_iter_t<64, 0>(p, st);
        case 128: return P ? launch_sm_iter_t<128, 1>(p, st) : launch_sm_iter_t<128, 0>(p, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace gnnmp

#ifdef GNNMP_SM_TRACE
extern "C" int gnnmp_debug_sm_trace(long long* dst, int n_wgs) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(gnnmp::g_sm_trace), sizeof(long long) * 8 * (size_t)(n_wgs > 8192 ? 8192 : n_wgs)) == hipSuccess ? 0 : -1;
}
#endif
