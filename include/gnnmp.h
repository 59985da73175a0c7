/*
 * gnnmp.h -- C ABI of libgnnmp.so: MI355X (gfx950) implementation of the GNN path-explorer and
 * path-smoother forward passes of rainorangelemon/gnn-motion-planning.
 *
 * The reference has no FFI layer; its boundary for this path is the torch.nn.Module protocol
 * of two classes.  Each entry point below names the reference interface it replaces:
 *
 *   gnnmp_explorer_manifest / _create / _destroy
 *       EncoderProcessDecoder.__init__ + load_state_dict      model.py:49-105, eval_gnn.py:99-101
 *   gnnmp_explorer_workspace_bytes / gnnmp_explorer_forward
 *       EncoderProcessDecoder.forward                         model.py:115-150 (call eval_gnn.py:194)
 *   gnnmp_smoother_manifest / _create / _destroy
 *       ModelSmoother.__init__ + load_state_dict              model_smoother.py:51-94, eval_gnn.py:102-104
 *   gnnmp_smoother_workspace_bytes / gnnmp_smoother_forward
 *       ModelSmoother.forward                                 model_smoother.py:104-142 (call smoother.py:243)
 *   gnnmp_explorer_forward_ex / gnnmp_smoother_forward_ex, gnnmp_*_status*          (ABI 4 / ABI 3)
 *       no counterpart: what the reference reports by raising (a node id outside the graph: tensor indexing, model.py:120) or does
 *       not have at all (it attends over ALL obstacles it is given, model.py:125-130), reported without synchronising the forward
 *   gnnmp_graph_workspace_bytes / gnnmp_graph_build
 *       create_data's edge construction                       eval_gnn.py:159-164
 *   gnnmp_maze_sample
 *       explore()'s rejection sampling (classification + compaction)   eval_gnn.py:180-184, environment/maze_env.py
 *   gnnmp_maze_steer
 *       proposed_path_smootherv2 (steering of the smoothing stage)  smoother.py:194-216
 *   gnnmp_maze_explore_workspace_bytes / gnnmp_maze_explore / gnnmp_maze_explore_ex
 *       explore()'s greedy loop + MazeEnv._edge_fp            eval_gnn.py:198-233, environment/maze_env.py:270-326
 *
 * Conventions
 *   - plain C types only; every pointer in a batch / forward call is a DEVICE pointer unless the
 *     name ends in _host; tensors are contiguous row-major fp32 / int64 / int32.
 *   - the library owns only the opaque handle (device copy of the packed weights).  Inputs,
 *     outputs and workspace are caller-allocated; nothing is retained past the call.
 *   - every kernel is enqueued on the hipStream_t passed in (as void*); no hidden device
 *     synchronisation (gnnmp_*_status and gnnmp_explorer_profile_read are the documented exceptions), no allocation inside forward -> forward is hipGraph-capturable.
 *   - functions return 0 (GNNMP_OK) or a negative gnnmp_status; no C++ exception crosses the ABI.
 *   - a handle is immutable after create: concurrent forwards from several host threads are legal
 *     with distinct workspaces / streams.
 */
#ifndef GNNMP_H
#define GNNMP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GNNMP_OK = 0,
    GNNMP_ERR_NULL = -1,          /* required pointer is NULL                               */
    GNNMP_ERR_DIMS = -2,          /* unsupported / inconsistent dimensions                   */
    GNNMP_ERR_WEIGHTS = -3,       /* weight blob size does not match the manifest            */
    GNNMP_ERR_WORKSPACE = -4,     /* workspace too small or misaligned                       */
    GNNMP_ERR_HIP = -5,           /* a HIP runtime call failed (see gnnmp_last_hip_error)    */
    GNNMP_ERR_ARG = -6,           /* bad scalar argument (loop < 1, negative counts, ...)    */
    /* the two below are found ON THE DEVICE during a forward and returned by gnnmp_*_status (never by forward itself, which
     * does not synchronise): the scores / waypoints of that forward are WRONG */
    GNNMP_ERR_CAPS = -7,          /* a caller promise sizing the kernels was exceeded: a graph has more obstacles than
                                     max_obstacles; a smoothing problem exceeds max_path / max_samples / max_edges */
    GNNMP_ERR_INDEX = -8          /* edge_index holds a node id outside [0, N_g)             */
} gnnmp_status;

const char* gnnmp_status_string(int status);
/* hipGetErrorString of the last failing HIP call on this host thread ("" if none). */
const char* gnnmp_last_hip_error(void);
/* ABI version of this build (bumped on any incompatible change). */
int gnnmp_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Explorer   (EncoderProcessDecoder, model.py:48-150)
 * ---------------------------------------------------------------------------------------- */
typedef struct gnnmp_explorer gnnmp_explorer;   /* opaque */
enum { GNNMP_F32 = 0, GNNMP_BF16 = 1, GNNMP_BF16X3 = 2 };

typedef struct {
    int32_t config_size;   /* C  (model.py:49 config_size)                                    */
    int32_t embed_size;    /* d  (embed_size): 32 or 64 (every shipped checkpoint)             */
    int32_t obs_size;      /* S  (obs_size): obstacles are viewed as [-1, S] (model.py:126)   */
    int32_t mlp_dtype;     /* GNNMP_F32 (exact fp32 MFMA, the reference's precision) or GNNMP_BF16:
                              MFMA operands rounded to bf16, fp32 accumulate, fp32 everywhere else
                              (BASELINE configs[2], [4]); or GNNMP_BF16X3: every fp32 operand split exactly
                              into three bf16 pieces, six piece products per MAC on the bf16 matrix pipe,
                              fp32 accumulate -> fp32-class results (explorer only).  Inputs and outputs
                              stay fp32 in every mode */
} gnnmp_explorer_dims;

/* Manifest of the state_dict tensors forward() actually uses (142 of the 200 keys), in the
 * order the weight blob of gnnmp_explorer_create must concatenate them (row-major fp32, exactly
 * as stored in the reference's .pt files).  Returns the number of entries when index < 0;
 * otherwise writes the reference parameter name (NUL-terminated, <= name_cap) and its element
 * count and returns 0. */
int gnnmp_explorer_manifest(const gnnmp_explorer_dims* dims, int index,
                            char* name, size_t name_cap, int64_t* numel);

/* weights_host: HOST pointer to the concatenated manifest tensors; n_floats must equal the
 * manifest total.  device: HIP device ordinal the handle lives on (the packed weights are allocated
 * there; forward must be given streams, inputs and workspace of the SAME device).  The calling
 * thread's current HIP device is saved and restored: creating a handle never changes it. */
int gnnmp_explorer_create(gnnmp_explorer** out, const gnnmp_explorer_dims* dims,
                          const float* weights_host, size_t n_floats, int device);
int gnnmp_explorer_destroy(gnnmp_explorer* h);

/* A block-diagonal batch of independent planning graphs (the reference processes one graph per
 * call; G = 1 reproduces that).  Graph g owns node rows [node_ptr[g], node_ptr[g+1]) of v,
 * columns [edge_ptr[g], edge_ptr[g+1]) of edge_index and rows [obs_ptr[g], obs_ptr[g+1]) of
 * obstacles.  edge_index holds GRAPH-LOCAL node ids (what each problem's create_data produced),
 * row 0 = message source j, row 1 = message target i (PyG flow source_to_target); any order,
 * duplicates allowed (each column is scored independently).  Node ids must lie in [0, N_g): the reference's
 * tensor indexing would raise; here the prep stage replaces an out-of-range id by node 0 (every kernel stays inside the
 * graph's rows) and raises GNNMP_ERR_INDEX in the device-side status (gnnmp_explorer_status).
 * One graph (G = 1, the reference's own call, model.py:115) may leave node_ptr, edge_ptr and obs_ptr all NULL: the
 * three totals describe it (inference entry points only; the training entry points take explicit prefix arrays). */
typedef struct {
    int32_t n_graphs;            /* G >= 1                                                    */
    int32_t total_nodes;         /* sum_g N_g                                                 */
    int32_t total_edges;         /* sum_g E_g                                                 */
    int32_t total_obstacles;     /* sum_g O_g (may be 0)                                      */
    int32_t max_obstacles;       /* >= max_g O_g (upper bound is fine; sizes the K/V slabs).  A PROMISE: obs_ptr lives on
                                    the device and forward never reads it back.  A graph with more obstacles than
                                    this (rounded up to 32) is attended over its first ones only -- the reference
                                    attends over ALL obstacles (model.py:125-130) -- and the forward's device-side
                                    status becomes GNNMP_ERR_CAPS (gnnmp_explorer_status)     */
    const float* v;              /* [total_nodes, C]                                          */
    const float* goal;           /* [G, C]                                                    */
    const float* obstacles;      /* [total_obstacles, S]                                      */
    const int64_t* edge_index;   /* [2, total_edges]                                          */
    const int32_t* node_ptr;     /* [G+1]   (or all three NULL when G == 1)                   */
    const int32_t* edge_ptr;     /* [G+1]                                                     */
    const int32_t* obs_ptr;      /* [G+1]                                                     */
} gnnmp_batch;

int gnnmp_explorer_workspace_bytes(const gnnmp_explorer* h, const gnnmp_batch* shape /* counts only */,
                                   size_t* bytes);

/* edge_scores [total_edges]: score of every edge_index column, in the caller's column order
 *   (= policy_output[target, source] of model.py:149).
 * dense_or_null: if non-NULL, sum_g N_g^2 floats; graph g's block starts at sum_{g'<g} N_g'^2 and
 *   is the reference's zero-filled policy_output[N_g, N_g] with P[target, source] = score.
 * loop >= 1 (model.py:139); use_obstacles mirrors the attribute read at model.py:125.
 * workspace must be 256-byte aligned. */
int gnnmp_explorer_forward(const gnnmp_explorer* h, const gnnmp_batch* batch, int loop, int use_obstacles,
                           float* edge_scores, float* dense_or_null,
                           void* workspace, size_t workspace_bytes, void* hip_stream);

/* gnnmp_explorer_forward with the forward's status words (see gnnmp_explorer_status below) written to `status_out` instead of
 * the workspace region: status_out points at gnnmp_explorer_status_words(batch) ints the DEVICE can write -- normally pinned
 * (hipHostMalloc'ed) host memory: the few threads that own a status word store it straight across the bus, so there is no copy
 * behind the forward and nothing to synchronise with except an event the caller records after this call; decode with
 * gnnmp_explorer_status_decode once that event has completed.  status_out == NULL is gnnmp_explorer_forward.  (What the Python
 * wrapper does on every forward, the reference's one-graph call included: a ring of pinned slots, looked at on a later call.) */
int gnnmp_explorer_forward_ex(const gnnmp_explorer* h, const gnnmp_batch* batch, int loop, int use_obstacles,
                              float* edge_scores, float* dense_or_null,
                              void* workspace, size_t workspace_bytes, void* hip_stream, int32_t* status_out_or_null);
int gnnmp_explorer_status_words(const gnnmp_batch* shape, size_t* n_words);

/* Device-side status of the LAST forward that ran on `workspace` (same batch shape): GNNMP_OK, or GNNMP_ERR_CAPS (a graph with
 * more obstacles than batch->max_obstacles: its scores are NOT the reference's), or GNNMP_ERR_INDEX (a node id outside
 * [0, N_g)); *first_graph_or_null = the first offending graph (-1 if none).  forward() itself never synchronises, so the
 * conditions it can only see on the device are collected in a small status region of the workspace (17 ints per graph, each
 * written unconditionally by one thread per forward: no fill launch, no atomics) and read HERE: this call copies the region
 * to the host on hip_stream and WAITS for the stream (the only synchronising entry point besides profile_read).  Callers that
 * must not block use gnnmp_explorer_status_region + their own asynchronous copy + gnnmp_explorer_status_decode (what the
 * Python wrapper does: the copy rides behind the forward and is looked at on a later call).  The reference has no counterpart:
 * it attends over all obstacles it is given and its indexing raises. */
int gnnmp_explorer_status(const gnnmp_explorer* h, const gnnmp_batch* shape, const void* workspace, size_t workspace_bytes,
                          void* hip_stream, int32_t* first_graph_or_null);
int gnnmp_explorer_status_region(const gnnmp_explorer* h, const gnnmp_batch* shape, size_t* offset, size_t* bytes);
/* The non-blocking copy itself: n_words ints from a status region (device) into HOST memory that the device can write (pinned /
 * hipHostMalloc'ed), as a one-block kernel on hip_stream -- not a hipMemcpyAsync, which would order the next forward behind a DMA
 * engine that may be busy with the caller's own result copies.  Record an event behind it, decode when the event has completed. */
int gnnmp_status_copy(const int32_t* src_device, int32_t* dst_host_mapped, int32_t n_words, void* hip_stream);
int gnnmp_explorer_status_decode(const int32_t* words_host, int n_graphs, int32_t* first_graph_or_null);

/* Optional per-stage timing.  While enabled, every forward on this handle records a HIP event pair
 * around each stage ON THE STREAM THE STAGE IS LAUNCHED ON; gnnmp_explorer_profile_read waits for
 * the recorded events, returns the summed milliseconds and the number of launches per stage since
 * the last read, and recycles the events.  Profiling mutates the handle: use it from one host
 * thread.  (There is no reference counterpart; eval_gnn.py:193-196 only wall-clocks the call.) */
enum {
    GNNMP_STAGE_PREP = 0,      /* memsets + CSR-by-destination build + goal node                */
    GNNMP_STAGE_OBS = 1,       /* obstacle codes and K/V operands                               */
    GNNMP_STAGE_NODE_PRE = 2,  /* node encoders + 3 attention blocks + loop invariants          */
    GNNMP_STAGE_EDGE_PRE = 3,  /* edge encoders + 3 attention blocks + loop invariants (dominant) */
    GNNMP_STAGE_MP = 4,        /* message passing: message MLP + max aggregation + node update, ONE fused launch
                                  per loop iteration                                            */
    GNNMP_STAGE_POLICY = 5,    /* per-edge policy head                                          */
    GNNMP_N_STAGES = 6
};
int gnnmp_explorer_profile(gnnmp_explorer* h, int enable);
int gnnmp_explorer_profile_read(gnnmp_explorer* h, double* ms_sum /* [GNNMP_N_STAGES] */,
                                int64_t* launches /* [GNNMP_N_STAGES] */);

/* Test hook: after a forward, copy an intermediate out of `workspace` into `dst` (device, fp32,
 * row-major, caller node/edge order).  which: 0 = loop-invariant part of the encoder output [total_nodes, d]
 * (model.py:141 without the h_i term), 1 = final h_i [total_nodes, d] (model.py:142), 2 = decode [total_nodes, d]
 * (model.py:143), 3 = goal node per graph as float [G].  Returns GNNMP_ERR_ARG if unknown; tap 2 is kept in bf16 in the
 * GNNMP_BF16 mode (its only reader is an MFMA operand) and returns GNNMP_ERR_DIMS there. */
int gnnmp_explorer_debug_tap(const gnnmp_explorer* h, const gnnmp_batch* batch, int which, float* dst,
                             void* workspace, size_t workspace_bytes, void* hip_stream);

/* ------------------------------------------------------------------------------------------
 * Training path of the explorer   (train_explorer.py:156-186: loss.backward() through model.py:115-150)
 * ------------------------------------------------------------------------------------------
 * The reference detaches node_free_code / edge_free_code before every use (model.py:141,142,146), so its policy loss
 * trains node_code, edge_code, goal_encoder, encoder, process.lin_0 / lin_1, decoder and policy and leaves the
 * obstacle-attention stack untouched.  train_forward computes the same scores as forward (the frozen attention outputs
 * come from the inference kernels, the trainable part is re-evaluated in the reference's formulation with activations
 * kept in `workspace`); train_backward turns d loss / d edge_scores into d loss / d parameters.
 * grad: DEVICE buffer of gnnmp_explorer_grad_floats(h) floats in MANIFEST order (the layout of the weight blob given to
 * gnnmp_explorer_create); entries of frozen tensors are zero.  The handle must be GNNMP_F32.  The same workspace
 * (>= gnnmp_explorer_train_workspace_bytes, 256-byte aligned) must be passed to forward and backward, untouched in
 * between; both enqueue on hip_stream and never synchronise.  Weight-gradient sums use float atomics: bit patterns may
 * differ from run to run in the last digits. */
int64_t gnnmp_explorer_grad_floats(const gnnmp_explorer* h);
int gnnmp_explorer_train_workspace_bytes(const gnnmp_explorer* h, const gnnmp_batch* shape, int loop, size_t* bytes);
int gnnmp_explorer_train_forward(const gnnmp_explorer* h, const gnnmp_batch* batch, int loop, int use_obstacles,
                                 float* edge_scores, void* workspace, size_t workspace_bytes, void* hip_stream);
int gnnmp_explorer_train_backward(const gnnmp_explorer* h, const gnnmp_batch* batch, int loop, const float* d_edge_scores,
                                  float* grad, void* workspace, size_t workspace_bytes, void* hip_stream);

/* ------------------------------------------------------------------------------------------
 * Smoother   (ModelSmoother, model_smoother.py:46-142)
 * ---------------------------------------------------------------------------------------- */
typedef struct gnnmp_smoother gnnmp_smoother;   /* opaque */

typedef struct {
    int32_t config_size;   /* C                                                               */
    int32_t embed_size;    /* d (128 in every shipped checkpoint; 32/64/128 supported)         */
    float scale;           /* ModelSmoother(scale=...) (model_smoother.py:51, str2name.py:40)   */
    int32_t mlp_dtype;     /* GNNMP_F32 or GNNMP_BF16 (MFMA operands only, as for the explorer)   */
} gnnmp_smoother_dims;

int gnnmp_smoother_manifest(const gnnmp_smoother_dims* dims, int index,
                            char* name, size_t name_cap, int64_t* numel);
/* device semantics as for gnnmp_explorer_create (current device saved and restored) */
int gnnmp_smoother_create(gnnmp_smoother** out, const gnnmp_smoother_dims* dims,
                          const float* weights_host, size_t n_floats, int device);
int gnnmp_smoother_destroy(gnnmp_smoother* h);

/* A batch of independent smoothing problems.  Problem b: path rows [path_ptr[b], path_ptr[b+1]),
 * free rows [free_ptr[b], ...), collided rows [coll_ptr[b], ...), edge columns
 * [edge_ptr[b], ...) with node ids local to that problem's [path; free; collided] stacking
 * (model_smoother.py:121).  The caller's path is never written. */
typedef struct {
    int32_t n_problems;
    int32_t total_path, total_free, total_collided, total_edges;
    /* The three max_* fields are caller PROMISES that size the kernels' LDS carve-up; the prefix arrays live on the device
     * and forward() never reads them back (no hidden synchronisation), so they are not checked against the arrays on the
     * host.  A problem that exceeds any of them gets NO kNN / chain edges (its interior waypoints then follow smooth_node of
     * the plain node codes: a wrong path) and the forward's device-side status becomes GNNMP_ERR_CAPS
     * (gnnmp_smoother_status).  Compute them from the same host-side counts the prefix arrays are built from (the Python
     * wrapper does). */
    int32_t max_path;            /* >= max_b P_b                                              */
    int32_t max_samples;         /* >= max_b (F_b + Co_b)                                     */
    int32_t max_edges;           /* >= max_b E_b                                              */
    const float* path;           /* [total_path, C]                                           */
    const float* free_pts;       /* [total_free, C]                                           */
    const float* collided;       /* [total_collided, C]                                       */
    const int64_t* edge_index;   /* [2, total_edges]                                          */
    const int32_t* path_ptr;     /* [B+1]                                                     */
    const int32_t* free_ptr;     /* [B+1]                                                     */
    const int32_t* coll_ptr;     /* [B+1]                                                     */
    const int32_t* edge_ptr;     /* [B+1]                                                     */
} gnnmp_smooth_batch;

int gnnmp_smoother_workspace_bytes(const gnnmp_smoother* h, const gnnmp_smooth_batch* shape, size_t* bytes);

/* out_path [total_path, C]: the new waypoints (end points copied through, model_smoother.py:139).
 * ONE problem (n_problems == 1, the reference's own call) may leave path_ptr, free_ptr, coll_ptr and edge_ptr all NULL:
 * the totals describe it (inference entry point only).
 * Per-problem limits of the kernels (GNNMP_ERR_DIMS beyond them, nothing is silently truncated): at most 2048 samples
 * (free + collided; the reference's planner passes at most 500 + 500, smoother.py:57-58) and at most 7500 candidate
 * edges (caller edges + 10 kNN edges per waypoint). */
int gnnmp_smoother_forward(const gnnmp_smoother* h, const gnnmp_smooth_batch* batch, int loop,
                           float* out_path, void* workspace, size_t workspace_bytes, void* hip_stream);

/* As gnnmp_explorer_forward_ex: the status words of this forward (one int per problem) go to `status_out` (device-writable,
 * normally pinned host memory) instead of the workspace; NULL = gnnmp_smoother_forward. */
int gnnmp_smoother_forward_ex(const gnnmp_smoother* h, const gnnmp_smooth_batch* batch, int loop,
                              float* out_path, void* workspace, size_t workspace_bytes, void* hip_stream, int32_t* status_out_or_null);

/* Device-side status of the LAST forward on `workspace`, as for the explorer: GNNMP_OK or GNNMP_ERR_CAPS (a problem exceeded
 * max_path / max_samples / max_edges and was smoothed WITHOUT its edges); one int per problem.  gnnmp_smoother_status
 * synchronises hip_stream; _region / _decode are the non-blocking pieces. */
int gnnmp_smoother_status(const gnnmp_smoother* h, const gnnmp_smooth_batch* shape, const void* workspace, size_t workspace_bytes,
                          void* hip_stream, int32_t* first_problem_or_null);
int gnnmp_smoother_status_region(const gnnmp_smoother* h, const gnnmp_smooth_batch* shape, size_t* offset, size_t* bytes);
int gnnmp_smoother_status_decode(const int32_t* words_host, int n_problems, int32_t* first_problem_or_null);

/* ------------------------------------------------------------------------------------------
 * Training path of the smoother   (train_smoother.py:33-61 through model_smoother.py:104-142 under model.train())
 * ------------------------------------------------------------------------------------------
 * One problem per call (batch->n_problems == 1), fp32 handle.  The training forward differs from the inference one where
 * the reference's does: BatchNorm (node_code.1) normalises with the statistics of THIS call's node rows (path + free +
 * collided) in every loop iteration, and activations are kept.  bn_stats_or_null [loop][2][d] receives, per iteration, the
 * batch mean and the UNBIASED batch variance (what a caller needs to update running_mean / running_var the way
 * torch.nn.BatchNorm1d does).  train_backward: d loss / d out_path [P, C] -> d loss / d parameters in manifest order
 * (gnnmp_smoother_grad_floats floats; running_mean / running_var entries stay zero).  Same workspace for both calls. */
int64_t gnnmp_smoother_grad_floats(const gnnmp_smoother* h);
int gnnmp_smoother_train_workspace_bytes(const gnnmp_smoother* h, const gnnmp_smooth_batch* shape, int loop, size_t* bytes);
int gnnmp_smoother_train_forward(const gnnmp_smoother* h, const gnnmp_smooth_batch* batch, int loop, float* out_path,
                                 float* bn_stats_or_null, void* workspace, size_t workspace_bytes, void* hip_stream);
int gnnmp_smoother_train_backward(const gnnmp_smoother* h, const gnnmp_smooth_batch* batch, int loop, const float* d_out_path,
                                  float* grad, void* workspace, size_t workspace_bytes, void* hip_stream);

/* ------------------------------------------------------------------------------------------
 * Graph construction on the device   (create_data, eval_gnn.py:150-165; knn_graph :160,162; coalesce :164)
 * ---------------------------------------------------------------------------------------- */
/* Graph g owns node rows [node_ptr[g], node_ptr[g+1]) of v; its first n_free[g] rows are the
 * collision-free samples; k1[g] = ceil(k * ln(n_free) / ln(100)) (eval_gnn.py:159) is computed by the
 * caller.  The result is the reference's edge set: kNN_k1(all) + reversed + kNN_k1(free only) + reversed,
 * self loops included, coalesced (sorted by (source, target), duplicates dropped), graph-local ids. */
typedef struct {
    int32_t n_graphs;
    int32_t total_nodes;
    int32_t k1_max;              /* >= max_g k1[g]                                            */
    int32_t config_size;         /* C                                                         */
    const float* v;              /* [total_nodes, C]                                          */
    const int32_t* node_ptr;     /* [G+1]                                                     */
    const int32_t* n_free;       /* [G]                                                       */
    const int32_t* k1;           /* [G]                                                       */
} gnnmp_graph_batch;

int gnnmp_graph_workspace_bytes(const gnnmp_graph_batch* shape, size_t* bytes);
/* edge_index_out: [2, out_cap] int64 with row stride out_cap (row 0 = source, row 1 = target); graph g's
 * columns are [edge_ptr_out[g], edge_ptr_out[g+1]).  4 * k1_max * total_nodes columns always suffice (the
 * no-duplicate worst case); with a smaller out_cap the columns beyond it are not written while edge_ptr_out
 * still reports the true totals, so a caller may try an estimate and repeat with edge_ptr_out[G] columns.
 * edge_ptr_out: [G+1] int32 (device). */
int gnnmp_graph_build(const gnnmp_graph_batch* batch, int64_t* edge_index_out, int64_t out_cap,
                      int32_t* edge_ptr_out, void* workspace, size_t workspace_bytes, void* hip_stream);

/* ------------------------------------------------------------------------------------------
 * Explore stage on the device for 2-D maze problems
 *   (greedy best-edge expansion eval_gnn.py:198-233 + MazeEnv collision checker maze_env.py:270-326)
 * ---------------------------------------------------------------------------------------- */
/* Problem b: node rows [node_ptr[b], node_ptr[b+1]) of v (float32 [.,2]; the first n_free[b] rows are the free
 * samples, row 0 = start, row 1 = goal sample), edge columns [edge_ptr[b], ...) of edge_index (graph-local ids)
 * with their explorer scores, its width x width occupancy map (float64, 1 = obstacle, map[x][y]) and float64
 * goal state.  One explorer forward per problem, fresh search tree (the reference's default batch = t_max). */
typedef struct {
    int32_t n_problems, total_nodes, total_edges, width;
    const float* v;
    const int32_t *node_ptr, *edge_ptr, *n_free;
    const int64_t* edge_index;
    const float* scores;
    const double* maps;          /* [B, width, width]                                         */
    const double* goal_states;   /* [B, 2]                                                    */
} gnnmp_maze_batch;

int gnnmp_maze_explore_workspace_bytes(const gnnmp_maze_batch* shape, size_t* bytes);
/* Outputs (device): success [B]; n_explored [B] and explored [total_nodes] (explored node ids in order, problem b
 * at node_ptr[b]); n_pairs [B] and explored_edges [2 * (2 * total_edges + B)] ((a, b) pairs in order, starting
 * with the reference's initial [0, 0]; problem b at int offset 2 * (2 * edge_ptr[b] + b)); path_len [B] and
 * path [total_nodes] (node ids start -> goal); checks [B] = collision-check count of the explore stage. */
int gnnmp_maze_explore(const gnnmp_maze_batch* batch, int32_t* success, int32_t* n_explored, int32_t* explored,
                       int32_t* n_pairs, int32_t* explored_edges, int32_t* path_len, int32_t* path, int64_t* checks,
                       void* workspace, size_t workspace_bytes, void* hip_stream);

/* The general form: `dim` = 2 (point robot, as above) or 3 (the stick robot of MazeEnv(dim=3): v [.,3], goal_states
 * [B,3], collision model maze_env.py:254-302,330-347), and optionally the search trees of EARLIER rounds, which is what
 * the reference's resample loop carries across explorer forwards (eval_gnn.py:235-247; node ids of free samples are
 * stable because new samples are appended behind the old ones).  resume == NULL or resume->n_explored == NULL: fresh
 * trees.  Otherwise, per problem b: n_explored[b] >= 1 explored node ids at explored[node_ptr[b] ..] (order kept),
 * prev[node_ptr[b] + node] = tree parent of every explored node, and the (a, b) pairs recorded so far
 * (n_pairs[b] of them, starting with the initial [0, 0]) at pairs[2 * pair_ptr[b] ..]; they feed the reference's
 * legacy-index mask (eval_gnn.py:202) on the new scores.  Outputs as for gnnmp_maze_explore, except that in resume mode
 * explored_edges / n_pairs hold only the pairs added by THIS round (from index 0 of the problem's slot), and
 * prev_out_or_null [total_nodes] receives the parents of all explored nodes (for the next round). */
typedef struct {
    const int32_t* n_explored;   /* [B] or NULL                                               */
    const int32_t* explored;     /* [total_nodes]                                             */
    const int32_t* prev;         /* [total_nodes]                                             */
    const int32_t* n_pairs;      /* [B]                                                       */
    const int32_t* pairs;        /* 2 ints per pair                                           */
    const int32_t* pair_ptr;     /* [B+1] (in pairs)                                          */
} gnnmp_maze_resume;
int gnnmp_maze_explore_ex(const gnnmp_maze_batch* batch, int32_t dim, const gnnmp_maze_resume* resume_or_null,
                          int32_t* success, int32_t* n_explored, int32_t* explored, int32_t* n_pairs,
                          int32_t* explored_edges, int32_t* path_len, int32_t* path, int64_t* checks,
                          int32_t* prev_out_or_null, void* workspace, size_t workspace_bytes, void* hip_stream);

/* Collision-checked steering of the smoothing stage for 2-D mazes: proposed_path_smootherv2 (smoother.py:194-216)
 * with MazeEnv's checker (maze_env.py:270-326), batched.  Problem b owns waypoints [path_ptr[b], path_ptr[b+1])
 * of old_path / new_path (float32 [.,2]: the current path and the smoother network's proposal, smoother.py:243-245).
 * out_path receives the steered path (may NOT alias old_path / new_path); tmp: [total_path, 2] float32 scratch;
 * checks [B] (int64) is INCREMENTED by the collision checks spent.  All pointers are device pointers. */
int gnnmp_maze_steer(int32_t n_problems, int32_t total_path, int32_t width, const double* maps, const int32_t* path_ptr,
                     const float* old_path, const float* new_path, float* out_path, float* tmp, int64_t* checks,
                     void* hip_stream);

/* Rejection sampling of the explore stage for 2-D mazes (eval_gnn.py:180-184 through MazeEnv.sample_n_points / uniform_sample,
 * environment/maze_env.py): the raw draws stay with the host's numpy generator (`attempts` is the stream of uniform(-1, 1) pairs in
 * draw order, float64, DEVICE memory); the device classifies every draw against the problem's occupancy map (float64 arithmetic of
 * maze_env's cell lookup), finds the n_free-th free draw of every problem IN STREAM ORDER (problem b + 1 starts behind the last
 * draw problem b consumed, like the reference's one global generator), and writes the float32 node rows the graph builder and the
 * explorer read: per problem [init_state, goal_state, the n_free free draws, the first min(rejected, n_free) rejected draws]
 * (eval_gnn.py:182: collided[:len(free)]), compact, with node_ptr_out [B + 1].  v_out must hold n_problems * (2 + 2 * n_free) rows.
 * *cursor (device, in / out): index of the first unconsumed draw; used_out [B]: draws consumed per problem (the sampling's collision
 * checks, maze_env.py counts one per draw); *ok_out = 1, or 0 when the stream ran out before the last problem was complete (nothing
 * is consumed then: *cursor is unchanged -- append draws and call again).  One launch on hip_stream, no synchronisation. */
typedef struct {
    int32_t n_problems, width, n_free;
    int64_t n_attempts;
    const double* attempts;      /* [n_attempts, 2]                                           */
    const double* maps;          /* [B, width, width]                                         */
    const double* init_states;   /* [B, 2]                                                    */
    const double* goal_states;   /* [B, 2]                                                    */
} gnnmp_maze_sample_batch;
int gnnmp_maze_sample(const gnnmp_maze_sample_batch* batch, int64_t* cursor, float* v_out, int32_t* node_ptr_out,
                      int32_t* used_out, int32_t* ok_out, void* hip_stream);

/* ------------------------------------------------------------------------------------------
 * Host-only helpers exported for the CPU test-suite (no device needed)
 * ---------------------------------------------------------------------------------------- */
/* Pack a row-major weight matrix W[out_f, in_f] (leading dimension ld, column offset col0, n_in
 * columns used) into the MFMA A-operand tile format consumed by the kernels:
 * dst[((ot*NTI + it)*1024) + ((r>>2)*64 + lane)*4 + (r&3)] =
 *     W[32*ot + (lane&31)][col0 + 32*it + phi(r, lane>>5)],  phi(r,h) = (r&3) + 8*(r>>2) + 4*h.
 * out_f and n_in must be multiples of 32.  Returns the number of floats written. */
int64_t gnnmp_pack_a_tiles(const float* w, int out_f, int ld, int col0, int n_in, float* dst);
/* Pack the first-layer ("raw input") form: dst[(ot*ksteps + st)*64 + lane] =
 *     W[32*ot + (lane&31)][col0 + 2*st + (lane>>5)]  (0 beyond n_in); ksteps = ceil(n_in/2). */
int64_t gnnmp_pack_a_small(const float* w, int out_f, int ld, int col0, int n_in, float* dst);
/* Per-feature vector in register order: dst[(t*2 + h)*16 + r] = b[32*t + phi(r,h)]. */
int64_t gnnmp_pack_vec(const float* b, int n, float* dst);
/* A operands of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 for the double-precision stretch of the node side
 * (node_free_code encoder and the attention sub-block of node_attentions.0, reference model.py:122,164-181):
 * dst[(ob*ceil(n_in/4) + ks)*64 + lane] = W[16*ob + i'][col0 + 4*ks + (lane>>4)] (0 beyond n_in) with i = lane&15 and
 * i' = i, or i' = 4*(i%4) + i/4 when row_perm (matrices multiplied by the f32 instruction, whose result register r of lane
 * group g is row 4g + r where the f64 instruction's is 4r + g).  Returns the floats written; out_f a multiple of 16. */
int64_t gnnmp_pack_f64_ops(const float* w, int out_f, int ld, int col0, int n_in, int row_perm, float* dst);

#ifdef __cplusplus
}
#endif
#endif /* GNNMP_H */
