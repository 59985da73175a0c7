// kernels.hpp -- parameter blocks and launcher prototypes shared by api.cpp and the .hip files.
#pragma once
#include <hip/hip_runtime.h>
#include "layout.hpp"

namespace gnnmp {

struct PrepParams {
    int G, E, C;
    const long long* edge_index;                 // [2, E] graph-local ids
    const int *node_ptr, *edge_ptr;              // caller prefix arrays [G+1]
    const float *v, *goal;
    int *node_ptr_pad, *edge_ptr_pad;            // [G+1]
    long long* dense_ptr;                        // [G+1]
    int *deg, *cursor, *row_beg;                 // [Npad]
    int *ntile_graph, *etile_graph;              // per 32-row tile
    int4* csr;                                   // [Epad] {source, target, caller column, 0}; -1 = pad slot
    int* goal_node;                              // [G] padded node id
    int* tile_meta;                              // per 32-edge tile, see prep_graph_body
    int2* blk_span;                              // per 256-row block of the padded node space: its graph's four-tile groups [first, end)
    int* single_out;                             // non-null: ONE graph given by its totals; node_ptr / edge_ptr / obs_ptr point here
    int single_n, single_e, single_o;            //           ([0,N | 0,E | 0,O], written by the prep stage before anything reads them)
    int n_etiles;
    // device-side status of the forward (gnnmp_explorer_status): 17 ints per graph, every slot written unconditionally by exactly
    // one thread per forward (no zero-fill, no atomics): [17 g] = GNNMP_STATUS_OBSTACLES when the graph has more obstacles than
    // obs_cap (the K/V slabs are sized for max_obstacles; the attention then sees only the first obs_cap), [17 g + 1 + part] =
    // GNNMP_STATUS_NODE_ID when part `part` of the prep stage met a node id outside [0, N_g)
    const int* obs_ptr;                          // caller prefix array or nullptr (single graph: single_o)
    int obs_cap;                                 // 32 * ot_max, or INT_MAX when the forward ignores obstacles
    int* gstat;
};
constexpr int kGstatStride = 17;                 // 1 + the largest `parts` of the prep stage

struct ObsParams {
    const float* obstacles;
    const int* obs_ptr;
    int S;
    const float* w[2];       // packed ObsBlob for node side / edge side
    ObsBlob blob;
    float* kv[2];            // K/V slabs, node side / edge side
    int kv_stride;           // floats per (graph, block) slab = 2 * ot_max * NT * 1024
    int ot_max;
};

// node side, block 0 in fp64: extra workgroups of the obstacle launch (node_f64_body)
struct NodeF64Params {
    const float* v;
    int C;
    const int *node_ptr, *node_ptr_pad, *ntile_graph;      // ntile_graph: graph of every 32-row tile of the padded node space
    const float* w;          // F64Blob
    F64Blob blob;
    float* m0;               // out [Npad, d]: node_free_code after the attention sub-block of block 0 (input of its map_feed)
    int n_wg;                // workgroups of this role (64 * groups padded node rows each); 0 = role not used
    int f64_first;           // dispatch order inside the obstacle launch: 1 = these workgroups before the obstacle ones
    int groups;              // 64-row groups per workgroup: 1 (few graphs: shortest chains) or 4 (obstacle operands built once per 256 rows)
};

struct PreParams {
    const float *v, *goal;
    int C;
    const int *node_ptr, *node_ptr_pad;
    const int* tile_graph;
    const int4* csr;
    int* rec32;              // out (EDGE): packed per-edge record of the message-passing kernels
    const int* obs_ptr;
    const int* goal_node;
    const float* enc;
    EncBlob encb;
    const float* att;        // 3 consecutive AttBlob
    const float* out;
    int out_size;
    const float* kv;
    int kv_stride, ot_max, ot_chunk;
    int wregion;             // floats reserved for the weight region of LDS
    int n_wg;                // workgroup tiles in the padded index space (set by launch_pre)
    // resident variant only
    const int* ptr_pad_total;   // node_ptr_pad or edge_ptr_pad [G+1] (the index space this launch tiles)
    const int* tile_meta;       // per 32-edge tile (EDGE only)
    int G;
    int use_obstacles;
    int out_in_lds;          // resident variant: the epilogue blob is staged into LDS too (it fits)
    float *o0, *o1, *o2, *o3, *o4;
    const float* m0;         // NODE only, optional: rows produced by node_f64_body; the kernel then skips node_free_code's encoder
                             // and the attention sub-block of block 0
    float* om;               // optional: the attention output itself (node_free_code / edge_free_code after the 3 blocks),
                             // row-major [rows of this launch's padded index space, d]: frozen input of the training path
};

struct MpFusedParams {
    const int* rec32;
    const int *row_beg, *deg, *ntile_graph, *node_ptr_pad;
    const float *A, *B, *Ke, *X, *R;
    const float *we, *wn;        // MpEBlob (staged into LDS), MpNBlob (read from global)
    const float* wn_std;         // the MpNBlob of the non-last iterations (its m3 = W_dst: B' of a tile is recomputed from its X rows)
    int last;                    // last iteration: wn holds the decoder / policy matrices and the node phase writes PT where B' went
    float *Hout, *Xout, *Aout, *Bout;
    int n_tiles;                 // 32-node tiles of the padded node space
    int store_h;
    int tpw;                     // adjacent four-tile groups per workgroup (set by launch_mp_fused)
    int G;                       // graphs: node_ptr_pad[G] / 32 = tiles actually in use (n_tiles is an upper bound)
    int order;                   // order of the four-tile groups: 0 plain, 1 mirrored pairs (tpw == 2), 2 persistent workgroups on a snake (mp_fused_kernel)
    const int2* blk_span;        // order 2: PrepParams::blk_span
    long long* trace;            // diagnostics builds only (-DGNNMP_MP_TRACE): per-wave timestamps, else nullptr
};

struct PolicyParams {
    const int4* csr;
    const int *etile_graph, *node_ptr, *node_ptr_pad;
    const long long* dense_ptr;
    const float *PS, *PT, *PE, *w;
    float *scores, *dense;
    int n_tiles;
};

struct SmParams {
    int B, C, total_path, total_edges;
    float scale;
    const float *path, *free_pts, *collided;
    const long long* edge_index;
    const int *path_ptr, *free_ptr, *coll_ptr, *edge_ptr;
    const float* w;
    SmLayout L;
    float *cur, *cur_next;            // scaled working path [total_path, C]
    int* knn;                         // [total_path, kSmK] sample index or -1
    int *e_src, *e_dst, *e_count;     // sorted unique edges per problem (padded space), count per problem
    int *seg_beg, *seg_cnt;           // per padded path node: its run of incoming edges
    int *etile_prob, *ptile_prob;     // tile -> problem (-1 unused)
    float* msg;                       // [edge capacity, d]
    float* tgt;                       // [path capacity, d] per padded path node: b00 + (W_c - W_a) x_i, written by the target role of the
    int* tgt_flag;                    //   split message kernel; [path tiles] 1 = that tile's rows are published (reset by the graph stage); may be null
    int *tile_cnt, *elist, *plist;    // [2][2] counters {edge tiles, path tiles in use} per iteration parity + the tiles' indices (any order): appended by
    int parity;                       //   the graph stage of iteration `parity`, which also zeroes the other pair; may be null (training path)
    int cand_cap, n_etiles, n_ptiles;
    int samp_cap, path_cap;           // caller's upper bounds max_b (F_b + Co_b), max_b P_b (sm_graph_kernel's LDS carve-up)
    int one_free, one_coll;           // path_ptr == nullptr: ONE problem, its sample counts (waypoints / edges: total_path / total_edges)
    int init_from_path;               // first iteration: the knn kernel also writes cur = path / scale
    float* out;                       // last iteration: the node kernel also writes out = cur * scale (else nullptr)
    int* stat;                        // [B] device-side status (gnnmp_smoother_status): 1 = the problem exceeds the caller's max_path /
                                      //   max_samples / max_edges promises and got NO edges; written unconditionally by the graph stage
};

hipError_t launch_sm_init(int n, float scale, const float* path, float* cur, hipStream_t st);
hipError_t launch_sm_final(int n, float scale, const float* cur, float* out, hipStream_t st);
hipError_t launch_sm_iter(int D, int P, const SmParams& p, hipStream_t st);

struct GbParams {
    int G, C, total_nodes, kmax;
    const float* v;
    const int *node_ptr, *n_free, *k1;
    int *nb_all, *nb_free;            // [total_nodes, kmax] neighbour ids (graph-local) or -1
    int *cnt, *cur, *off, *ucnt, *uoff, *gtotal;
    int *large_cnt, *large;           // nodes of graphs beyond 1024 nodes, listed by the first kNN launch for the second (large aliases uoff)
    int* bucket;                      // [4 * kmax * total_nodes] targets grouped by source
    int* edge_ptr;                    // out [G+1]
    long long* edge_index;            // out [2, out_cap]
    long long out_cap;
};
hipError_t launch_graph_build(const GbParams& p, hipStream_t st);

struct MazeParams {
    int B, total_edges, w;
    const float* v;                       // [sumN, 2]
    const int *node_ptr, *edge_ptr, *n_free;
    const long long* edge_index;          // [2, sumE] graph-local
    const float* scores;                  // [sumE]
    const double *maps, *goal_states;     // [B, w, w], [B, 2]
    int *in_ptr, *cnt, *pos, *prev;
    int2* in_rec;                         // [sumE] visible cells grouped by row: (column | dead bit 31, score bits)
    float* rb_val;                        // cached best live cell per explored row (value, column, slot in in_rec)
    int *rb_src, *rb_eid;
    int *success, *n_explored, *explored, *n_pairs, *explored_edges, *path_len, *path;
    long long* checks;
    int dim;                              // 2 (point robot, v [.,2]) or 3 (stick robot, v [.,3])
    // resume (all nullptr = fresh trees): tree and pair list of the earlier rounds
    const int *n_explored0, *explored0, *prev0, *n_pairs0, *pairs0, *pair_ptr0;
    int* prev_out;                        // optional [sumN]: parent of every explored node
};
hipError_t launch_maze_explore(const MazeParams& p, hipStream_t st);

// collision-checked steering of the smoothing stage (smoother.py:194-216) for 2-D mazes
struct MazeSteerParams {
    int B, w;
    const double* maps;                   // [B, w, w]
    const int* path_ptr;                  // [B + 1]
    const float *old_path, *new_path;     // [sumP, 2]
    float *out_path, *tmp;                // [sumP, 2]
    long long* checks;                    // [B], incremented
};
hipError_t launch_maze_steer(const MazeSteerParams& p, hipStream_t st);

// rejection sampling of the explore stage for 2-D mazes on the device (eval_gnn.py:180-184 through MazeEnv.sample_n_points)
struct MazeSampleParams {
    int B, w, n;                          // problems, map width, free samples wanted per problem
    const double* attempts;               // [M, 2] the raw uniform(-1, 1) draws of the host generator, stream order
    long long M;
    const double *maps, *init_states, *goal_states;      // [B, w, w], [B, 2], [B, 2]
    float* v;                             // out: node rows [init, goal, free x n, rejected x min(rejected, n)] per problem, compact
    int* node_ptr;                        // out [B + 1]
    int* used;                            // out [B]: attempts consumed (= collision checks of the sampling)
    long long* cursor;                    // in / out: index of the next unconsumed attempt
    int* ok;                              // out: 0 = the stream ran out before the last problem had its n free samples
};
hipError_t launch_maze_sample(const MazeSampleParams& p, hipStream_t st);

// ---- training path (train_kernels.hip)
struct TrainGeom {
    int G, C, Npad, Epad;
    const float *v, *goal;
    const int *node_ptr, *node_ptr_pad, *ntile_graph, *goal_node, *row_beg, *deg;
    const int4* csr;
    int *out_beg, *out_cnt, *out_cur, *out_slot;      // edges grouped by source (t_out_csr): [Npad] x 3, [Epad] CSR slots
};
hipError_t t_out_csr(const TrainGeom& q, hipStream_t st);
hipError_t t_sort_csr(int Npad, int4* csr, const int* row_beg, const int* deg, hipStream_t st);
size_t t_linear_dw_scratch_floats(int R, int K, int O);
hipError_t t_linear(int R, int K, int O, const float* X, const float* W, const float* b, float* Y, bool relu, hipStream_t st);
hipError_t t_linear_dx(int R, int K, int O, const float* dY, const float* W, float* dX, bool accumulate, hipStream_t st);
hipError_t t_linear_dw(int R, int K, int O, const float* dY, const float* X, float* dW, float* db, float* scratch, hipStream_t st);
hipError_t t_relu_bwd(size_t n, const float* y, float* dy, hipStream_t st);
hipError_t t_fill(size_t n, float* x, float val, hipStream_t st);
hipError_t t_node_in(const TrainGeom& q, float* out, hipStream_t st);
hipError_t t_edge_in(const TrainGeom& q, float* out, hipStream_t st);
hipError_t t_h0(const TrainGeom& q, int D, const float* goal_encoder, float* H0, hipStream_t st);
hipError_t t_h0_bwd(const TrainGeom& q, int D, const float* dH0, float* d_goal_encoder, hipStream_t st);
hipError_t t_concat(int R, int D, int parts, const float* a0, const float* a1, const float* a2, const float* a3, float* out,
                    hipStream_t st);
hipError_t t_split(int R, int D, int parts, int part, const float* d_in, float* dst, bool accumulate, hipStream_t st);
hipError_t t_msg_in(const TrainGeom& q, int D, const float* X, const float* EF, const float* EC, float* out, hipStream_t st);
hipError_t t_msg_in_bwd(const TrainGeom& q, int D, const float* dZ, float* dX, float* dEC, hipStream_t st);
hipError_t t_pol_in(const TrainGeom& q, int D, const float* Dn, const float* EF, float* out, hipStream_t st);
hipError_t t_pol_in_bwd(const TrainGeom& q, int D, const float* dP, float* dDn, hipStream_t st);
hipError_t t_segment_max(const TrainGeom& q, int D, const float* M, float* A, int* arg, hipStream_t st);
hipError_t t_segment_max_bwd(int Npad, int D, const float* dA, const int* arg, float* dM, hipStream_t st);
hipError_t t_scores_out(const TrainGeom& q, const float* slot_scores, float* out, hipStream_t st);
hipError_t t_scores_in(const TrainGeom& q, const float* d_out, float* d_slot, hipStream_t st);

hipError_t t_sm_nodes_in(int P, int F, int Co, int C, float scale, const float* cur, const float* free_pts, const float* coll, float* out,
                         hipStream_t st);
hipError_t t_bn_fwd(int N, int D, const float* x, const float* gamma, const float* beta, float* y, float* stats, bool relu, hipStream_t st);
hipError_t t_bn_bwd(int N, int D, const float* x, const float* dy, const float* gamma, const float* stats, float* dx, float* dgamma,
                    float* dbeta, hipStream_t st);
hipError_t t_sm_msg_in(const int* n_edges, int D, const int* e_src, const int* e_dst, const float* X, float* out, int cap, hipStream_t st);
hipError_t t_sm_msg_in_bwd(const int* n_edges, int D, const int* e_src, const int* e_dst, const float* dZ, float* dX, int n_rows, hipStream_t st);
hipError_t t_sm_scatter_add(const int* n_edges, int D, const int* e_dst, const float* M, float* S, int n_rows, hipStream_t st);
hipError_t t_sm_scatter_add_bwd(const int* n_edges, int D, const int* e_dst, const float* dS, float* dM, int cap, hipStream_t st);
hipError_t t_add_rows(size_t n, const float* a, const float* b, float* out, hipStream_t st);
hipError_t t_sm_path_update(int P, int C, const float* prev, const float* proposal, float* next, hipStream_t st);
hipError_t t_sm_path_update_bwd(int P, int C, const float* d_next, float* d_proposal, float* d_prev, hipStream_t st);
hipError_t t_sm_coords_bwd(int P, int C, const float* dXin, float* d_prev, hipStream_t st);
hipError_t t_scale(int n, float s, const float* x, float* y, hipStream_t st);
hipError_t launch_sm_knn_edges(const SmParams& p, hipStream_t st);      // kNN + coalesced edge list only (training path)

int prep_parts(int G, int E);
hipError_t launch_prep(const PrepParams& q, int Npad, int Epad, int* hist, hipStream_t st);
hipError_t launch_obs(int D, int P, const ObsParams& p, const NodeF64Params& q, int G, hipStream_t st);
hipError_t launch_pre(int D, int P, bool edge, int waves, const PreParams& p, int n_tiles32, size_t lds_bytes, hipStream_t st);
hipError_t launch_pre_resident_both(int D, int P, const PreParams& pn, const PreParams& pe, size_t lds_bytes, int node_blocks,
                                    int edge_blocks, hipStream_t st);
hipError_t launch_pre_resident(int D, int P, bool edge, const PreParams& p, size_t lds_bytes, int n_cu, hipStream_t st);
hipError_t launch_mp_fused(int D, int P, const MpFusedParams& p, hipStream_t st);
hipError_t launch_policy(int D, int P, const PolicyParams& p, hipStream_t st);
hipError_t launch_unpad_rows(int G, int total_nodes, int D, const int* node_ptr, const int* node_ptr_pad,
                             const float* src, float* dst, hipStream_t st);
hipError_t launch_zero_dense(float* dense, const long long* n_ptr, hipStream_t st);
hipError_t launch_goal_tap(int G, const int* goal_node, const int* node_ptr_pad, float* dst, hipStream_t st);

}  // namespace gnnmp
