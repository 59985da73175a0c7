// api.cpp -- host side of the C ABI (include/gnnmp.h): weight manifest and packing, workspace
// carving and the kernel sequence of the explorer forward pass.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gnnmp.h"
#include "kernels.hpp"
#include "layout.hpp"

using namespace gnnmp;

namespace {

thread_local std::string g_hip_error;

int hip_fail(hipError_t e) {
    g_hip_error = hipGetErrorString(e);
    return GNNMP_ERR_HIP;
}

#define HIP_TRY(expr)                              \
    do {                                           \
        hipError_t _e = (expr);                    \
        if (_e != hipSuccess) return hip_fail(_e); \
    } while (0)

struct Entry {
    std::string name;
    int rows, cols;     // cols == 0 -> vector of `rows`
    int64_t numel() const { return (int64_t)rows * (cols ? cols : 1); }
};

// ---------------------------------------------------------------------------------------------
// explorer manifest: the 142 state_dict tensors forward() reads
// ---------------------------------------------------------------------------------------------
std::vector<Entry> explorer_manifest(const gnnmp_explorer_dims& d) {
    const int C = d.config_size, D = d.embed_size, S = d.obs_size;
    std::vector<Entry> m;
    auto lin = [&](const std::string& n, int out, int in, bool bias = true) {
        m.push_back({n + ".weight", out, in});
        if (bias) m.push_back({n + ".bias", out, 0});
    };
    auto mlp2 = [&](const std::string& n, int in) { lin(n + ".0", D, in); lin(n + ".2", D, D); };
    m.push_back({"goal_encoder", D, 0});
    mlp2("node_code", 4 * C);
    mlp2("edge_code", 2 * C);
    mlp2("obs_node_code", S);
    mlp2("obs_edge_code", S);
    mlp2("node_free_code", C);
    mlp2("edge_free_code", 2 * C);
    for (const char* side : {"node_attentions", "edge_attentions"})
        for (int b = 0; b < 3; ++b) {
            const std::string p = std::string(side) + "." + std::to_string(b);
            lin(p + ".attention.key", D, D, false);
            lin(p + ".attention.query", D, D, false);
            lin(p + ".attention.value", D, D, false);
            m.push_back({p + ".attention.layer_norm.weight", D, 0});
            m.push_back({p + ".attention.layer_norm.bias", D, 0});
            for (const char* ff : {"map_feed", "obs_feed"}) {
                lin(p + "." + ff + ".w_1", D, D);
                lin(p + "." + ff + ".w_2", D, D);
                m.push_back({p + "." + ff + ".layer_norm.weight", D, 0});
                m.push_back({p + "." + ff + ".layer_norm.bias", D, 0});
            }
        }
    lin("encoder", D, 4 * D);
    lin("process.lin_0.0", D, 5 * D);
    lin("process.lin_0.2", D, D);
    lin("process.lin_1", D, 2 * D);
    lin("decoder", D, 2 * D);
    lin("policy.0", D, 3 * D);
    lin("policy.2", D, D);
    lin("policy.4", 1, D, false);
    return m;
}

struct Blob {          // host view of the caller's concatenated weights
    std::vector<Entry> man;
    std::vector<int64_t> off;
    const float* base;
    const float* get(const std::string& n) const {
        for (size_t i = 0; i < man.size(); ++i)
            if (man[i].name == n) return base + off[i];
        return nullptr;
    }
};

bool dims_ok(const gnnmp_explorer_dims& d) {
    return d.config_size >= 1 && d.config_size <= 64 && (d.embed_size == 32 || d.embed_size == 64) &&
           d.obs_size >= 1 && d.obs_size <= 64 &&
           (d.mlp_dtype == GNNMP_F32 || d.mlp_dtype == GNNMP_BF16 || d.mlp_dtype == GNNMP_BF16X3);
}

// round-to-nearest-even fp32 -> bf16 (what v_cvt_pk_bf16_f32 does on the device)
uint16_t to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);     // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

void pack_a_tiles_bf16(const float* w, int out_f, int ld, int col0, int n_in, float* dst_f) {
    uint16_t* dst = reinterpret_cast<uint16_t*>(dst_f);
    const int nto = out_f / 32, nti = n_in / 32;
    for (int ot = 0; ot < nto; ++ot)
        for (int it = 0; it < nti; ++it)
            for (int m = 0; m < 2; ++m)
                for (int lane = 0; lane < 64; ++lane)
                    for (int t = 0; t < 8; ++t)
                        dst[((size_t)((ot * nti + it) * 2 + m) * 64 + lane) * 8 + t] =
                            to_bf16(w[(size_t)(32 * ot + (lane & 31)) * ld + col0 + 32 * it + phi(8 * m + t, lane >> 5)]);
}

// bf16x3: every fp32 weight as three bf16 pieces w = w0 + w1 + w2 (exact for normal numbers)
void split3(float w, uint16_t (&pc)[3]) {
    float r = w;
    for (int i = 0; i < 3; ++i) {
        pc[i] = to_bf16(r);
        uint32_t u = (uint32_t)pc[i] << 16;
        float f;
        std::memcpy(&f, &u, 4);
        r -= f;
    }
}

void pack_a_tiles_bf16x3(const float* w, int out_f, int ld, int col0, int n_in, float* dst_f) {
    uint16_t* dst = reinterpret_cast<uint16_t*>(dst_f);
    const int nto = out_f / 32, nti = n_in / 32;
    for (int ot = 0; ot < nto; ++ot)
        for (int it = 0; it < nti; ++it)
            for (int m = 0; m < 2; ++m)
                for (int lane = 0; lane < 64; ++lane)
                    for (int t = 0; t < 8; ++t) {
                        uint16_t pc[3];
                        split3(w[(size_t)(32 * ot + (lane & 31)) * ld + col0 + 32 * it + phi(8 * m + t, lane >> 5)], pc);
                        for (int q = 0; q < 3; ++q)
                            dst[(((size_t)(ot * nti + it) * 3 + q) * 2 + m) * 512 + lane * 8 + t] = pc[q];
                    }
}

void pack_a_small_bf16x3(const float* w, int out_f, int ld, int col0, int n_in, float* dst_f) {
    uint16_t* dst = reinterpret_cast<uint16_t*>(dst_f);
    const int nto = out_f / 32, ks = (n_in + 15) / 16;
    for (int ot = 0; ot < nto; ++ot)
        for (int st = 0; st < ks; ++st)
            for (int lane = 0; lane < 64; ++lane)
                for (int t = 0; t < 8; ++t) {
                    const int k = 16 * st + 8 * (lane >> 5) + t;
                    uint16_t pc[3];
                    split3(k < n_in ? w[(size_t)(32 * ot + (lane & 31)) * ld + col0 + k] : 0.f, pc);
                    for (int q = 0; q < 3; ++q) dst[(((size_t)(ot * ks + st) * 3 + q) * 64 + lane) * 8 + t] = pc[q];
                }
}

void pack_a_small_bf16(const float* w, int out_f, int ld, int col0, int n_in, float* dst_f) {
    uint16_t* dst = reinterpret_cast<uint16_t*>(dst_f);
    const int nto = out_f / 32, ks = (n_in + 15) / 16;
    for (int ot = 0; ot < nto; ++ot)
        for (int st = 0; st < ks; ++st)
            for (int lane = 0; lane < 64; ++lane)
                for (int t = 0; t < 8; ++t) {
                    const int k = 16 * st + 8 * (lane >> 5) + t;
                    dst[((size_t)(ot * ks + st) * 64 + lane) * 8 + t] =
                        to_bf16(k < n_in ? w[(size_t)(32 * ot + (lane & 31)) * ld + col0 + k] : 0.f);
                }
}

std::vector<float> matvec(const float* w, int D, int ld, int col0, int n, const float* x) {
    std::vector<float> r(D);
    for (int i = 0; i < D; ++i) {
        float s = 0.f;
        for (int k = 0; k < n; ++k) s = std::fmaf(w[(size_t)i * ld + col0 + k], x[k], s);
        r[i] = s;
    }
    return r;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// exported host-only packing helpers
// ---------------------------------------------------------------------------------------------
extern "C" int64_t gnnmp_pack_a_tiles(const float* w, int out_f, int ld, int col0, int n_in, float* dst) {
    const int nto = out_f / 32, nti = n_in / 32;
    for (int ot = 0; ot < nto; ++ot)
        for (int it = 0; it < nti; ++it)
            for (int r = 0; r < 16; ++r)
                for (int lane = 0; lane < 64; ++lane)
                    dst[(size_t)(ot * nti + it) * 1024 + ((r >> 2) * 64 + lane) * 4 + (r & 3)] =
                        w[(size_t)(32 * ot + (lane & 31)) * ld + col0 + 32 * it + phi(r, lane >> 5)];
    return (int64_t)nto * nti * 1024;
}

extern "C" int64_t gnnmp_pack_a_small(const float* w, int out_f, int ld, int col0, int n_in, float* dst) {
    const int nto = out_f / 32, ks = (n_in + 1) / 2;
    for (int ot = 0; ot < nto; ++ot)
        for (int st = 0; st < ks; ++st)
            for (int lane = 0; lane < 64; ++lane) {
                const int k = 2 * st + (lane >> 5);
                dst[(size_t)(ot * ks + st) * 64 + lane] = (k < n_in) ? w[(size_t)(32 * ot + (lane & 31)) * ld + col0 + k] : 0.f;
            }
    return (int64_t)nto * ks * 64;
}

extern "C" int64_t gnnmp_pack_f64_ops(const float* w, int out_f, int ld, int col0, int n_in, int row_perm, float* dst) {
    if (!w || !dst || out_f < 16 || out_f % 16 || n_in < 1) return GNNMP_ERR_ARG;
    const int nks = (n_in + 3) / 4;
    for (int ob = 0; ob < out_f / 16; ++ob)
        for (int ks = 0; ks < nks; ++ks)
            for (int l = 0; l < 64; ++l) {
                const int k = 4 * ks + (l >> 4), i = l & 15;
                const int row = 16 * ob + (row_perm ? 4 * (i & 3) + (i >> 2) : i);
                dst[((size_t)ob * nks + ks) * 64 + l] = k < n_in ? w[(size_t)row * ld + col0 + k] : 0.f;
            }
    return (int64_t)(out_f / 16) * nks * 64;
}

extern "C" int64_t gnnmp_pack_vec(const float* b, int n, float* dst) {
    const int nt = n / 32;
    for (int t = 0; t < nt; ++t)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) dst[(t * 2 + h) * 16 + r] = b[32 * t + phi(r, h)];
    return (int64_t)nt * 32;
}

// ---------------------------------------------------------------------------------------------
// status
// ---------------------------------------------------------------------------------------------
extern "C" const char* gnnmp_status_string(int s) {
    switch (s) {
        case GNNMP_OK: return "ok";
        case GNNMP_ERR_NULL: return "null pointer argument";
        case GNNMP_ERR_DIMS: return "unsupported or inconsistent dimensions";
        case GNNMP_ERR_WEIGHTS: return "weight blob size does not match the manifest";
        case GNNMP_ERR_WORKSPACE: return "workspace too small or misaligned";
        case GNNMP_ERR_HIP: return "HIP runtime error";
        case GNNMP_ERR_ARG: return "bad scalar argument";
        case GNNMP_ERR_CAPS: return "a caller promise was exceeded on the device (max_obstacles / max_path / max_samples / max_edges): results are wrong";
        case GNNMP_ERR_INDEX: return "edge_index holds node ids outside [0, N_g): results are wrong";
    }
    return "unknown status";
}
extern "C" const char* gnnmp_last_hip_error(void) { return g_hip_error.c_str(); }
// 2: stage list of gnnmp_explorer_profile_read (fused message passing); 3: device-side status words (gnnmp_*_status*)
extern "C" int gnnmp_abi_version(void) { return 4; }

// ---------------------------------------------------------------------------------------------
// explorer handle
// ---------------------------------------------------------------------------------------------
// optional per-stage timing with HIP events recorded on the forward's own stream
struct StageProf {
    bool on = false;
    struct Rec { int stage; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};

struct gnnmp_explorer {
    gnnmp_explorer_dims dims;
    int device;
    float* w_dev;
    ExplorerOffsets off;
    EncBlob enc_e, enc_n;
    ObsBlob obs;
    StageProf* prof;      // mutable side state (profiling is single-threaded by contract)
    int n_cu;             // compute units of the device (persistent-kernel grid)
    int resident;         // use pre_resident_kernel when it fits (GNNMP_RESIDENT=0 disables)
    int resident_both;    // small batches: node and edge pre stages in one launch (GNNMP_PRE_BOTH=0 disables)
    int node_f64;         // node side block 0 in double precision (fp32-class modes; GNNMP_NODE_F64=0 disables, for attribution runs)
    float* w_raw_dev;     // the caller's weight blob as given (manifest order, torch row-major): the training path's view
    std::vector<Entry> man;
    std::vector<int64_t> man_off;
    int64_t n_raw;
};

namespace {
struct StageScope {
    StageProf* p; hipStream_t st; hipEvent_t b = nullptr;
    StageScope(StageProf* p_, int stage, hipStream_t s) : p(p_ && p_->on ? p_ : nullptr), st(s) {
        if (!p) return;
        hipEvent_t a = p->get();
        b = p->get();
        (void)hipEventRecord(a, st);
        p->recs.push_back({stage, a, b});
    }
    ~StageScope() { if (p) (void)hipEventRecord(b, st); }
};
}  // namespace

extern "C" int gnnmp_explorer_manifest(const gnnmp_explorer_dims* dims, int index, char* name, size_t name_cap,
                                       int64_t* numel) {
    if (!dims) return GNNMP_ERR_NULL;
    if (!dims_ok(*dims)) return GNNMP_ERR_DIMS;
    const auto m = explorer_manifest(*dims);
    if (index < 0) return (int)m.size();
    if (index >= (int)m.size()) return GNNMP_ERR_ARG;
    if (name && name_cap) {
        std::strncpy(name, m[index].name.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (numel) *numel = m[index].numel();
    return GNNMP_OK;
}

namespace {

template <int D, int P>
void pack_explorer(const Blob& B, const gnnmp_explorer_dims& dm, gnnmp_explorer* h, std::vector<float>& out) {
    const int C = dm.config_size, S = dm.obs_size;
    h->enc_e = EncBlob::make(P, D, 2 * C, 2 * C);              // both edge encoders read [v_src, v_dst]
    h->enc_n = EncBlob::make(P, D, 4 * C, C);                  // node_code reads 4C numbers, node_free_code C
    h->obs = ObsBlob::make(P, D, S);
    ExplorerOffsets& o = h->off;
    int cur = 0;
    auto take = [&](int n) { const int r = cur; cur += (n + 3) & ~3; return r; };
    o.enc_e = take(h->enc_e.size);
    o.enc_n = take(h->enc_n.size);
    o.att_e = take(3 * AttBlob<D, P>::size);
    o.att_n = take(3 * AttBlob<D, P>::size);
    o.out_e = take(OutEBlob<D, P>::size);
    o.out_n = take(OutNBlob<D, P>::size);
    o.mpn = take(MpNBlob<D, P>::size);
    o.mpn_last = take(MpNBlob<D, P>::size);
    o.mpe = take(MpEBlob<D, P>::size);
    o.pol = take(PolBlob<D, P>::size);
    o.obs_e = take(h->obs.size);
    o.obs_n = take(h->obs.size);
    o.f64 = take(F64Blob::make(D, C).size);
    o.total = cur;
    out.assign(cur, 0.f);
    float* PK = out.data();
    auto W = [&](const std::string& n) { return B.get(n); };
    auto tiles = [&](const float* w, int ld, int col0, float* dst) {
        if (P == 2) pack_a_tiles_bf16x3(w, D, ld, col0, D, dst);
        else if (P == 1) pack_a_tiles_bf16(w, D, ld, col0, D, dst);
        else gnnmp_pack_a_tiles(w, D, ld, col0, D, dst);
    };
    auto small = [&](const float* w, int ld, int n_in, float* dst) {
        if (P == 2) pack_a_small_bf16x3(w, D, ld, 0, n_in, dst);
        else if (P == 1) pack_a_small_bf16(w, D, ld, 0, n_in, dst);
        else gnnmp_pack_a_small(w, D, ld, 0, n_in, dst);
    };
    auto vec = [&](const float* b, float* dst) { gnnmp_pack_vec(b, D, dst); };

    // --- encoders on raw inputs
    auto enc = [&](const EncBlob& e, float* dst, const std::string& n0, int k0, const std::string& n1, int k1) {
        small(W(n0 + ".0.weight"), k0, k0, dst + e.as0);
        vec(W(n0 + ".0.bias"), dst + e.b0);
        tiles(W(n0 + ".2.weight"), D, 0, dst + e.a0);
        vec(W(n0 + ".2.bias"), dst + e.c0);
        small(W(n1 + ".0.weight"), k1, k1, dst + e.as1);
        vec(W(n1 + ".0.bias"), dst + e.b1);
        tiles(W(n1 + ".2.weight"), D, 0, dst + e.a1);
        vec(W(n1 + ".2.bias"), dst + e.c1);
    };
    enc(h->enc_e, PK + o.enc_e, "edge_code", 2 * C, "edge_free_code", 2 * C);
    {
        // edge_code is consumed by ONE matrix only: K_e = W1d.EF + W1e.EC + b1 with EC = W2.relu(..) + c (model.py:120, :38).
        // The edge kernels therefore produce W1e.EC + b1 directly: second encoder layer F = W1e.W2, bias W1e.c + b1
        // (products in double), one d x d product per edge less.  OutEBlob::w1e / b1 stay in the layout, unused.
        const float* w1m = W("process.lin_0.0.weight");             // [d, 5d]: columns 4d.. multiply edge_code
        const float* w2 = W("edge_code.2.weight");
        const float* c2 = W("edge_code.2.bias");
        const float* b1m = W("process.lin_0.0.bias");
        std::vector<float> F((size_t)D * D), cF(D);
        for (int i = 0; i < D; ++i) {
            for (int j = 0; j < D; ++j) {
                double acc = 0.0;
                for (int k = 0; k < D; ++k) acc += (double)w1m[(size_t)i * 5 * D + 4 * D + k] * (double)w2[(size_t)k * D + j];
                F[(size_t)i * D + j] = (float)acc;
            }
            double acc = (double)b1m[i];
            for (int k = 0; k < D; ++k) acc += (double)w1m[(size_t)i * 5 * D + 4 * D + k] * (double)c2[k];
            cF[i] = (float)acc;
        }
        tiles(F.data(), D, 0, PK + o.enc_e + h->enc_e.a0);
        vec(cF.data(), PK + o.enc_e + h->enc_e.c0);
    }
    enc(h->enc_n, PK + o.enc_n, "node_code", 4 * C, "node_free_code", C);

    // --- attention blocks (map side) and obstacle side
    using A = AttBlob<D, P>;
    for (int side = 0; side < 2; ++side) {
        const std::string sname = side == 0 ? "edge_attentions" : "node_attentions";
        float* att = PK + (side == 0 ? o.att_e : o.att_n);
        float* ob = PK + (side == 0 ? o.obs_e : o.obs_n);
        const std::string oc = side == 0 ? "obs_edge_code" : "obs_node_code";
        const ObsBlob& L = h->obs;
        small(W(oc + ".0.weight"), S, S, ob + L.as0);
        vec(W(oc + ".0.bias"), ob + L.b0);
        tiles(W(oc + ".2.weight"), D, 0, ob + L.a0);
        vec(W(oc + ".2.bias"), ob + L.c0);
        for (int b = 0; b < 3; ++b) {
            const std::string p = sname + "." + std::to_string(b);
            float* a = att + (size_t)b * A::size;
            // Wqk = Wq^T Wk (see AttBlob): [d, d], Wqk[i][j] = sum_c Wq[c][i] Wk[c][j]
            std::vector<float> wqk((size_t)D * D);
            {
                const float *wq = W(p + ".attention.query.weight"), *wk = W(p + ".attention.key.weight");
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j) {
                        double acc = 0.0;
                        for (int c2 = 0; c2 < D; ++c2) acc += (double)wq[(size_t)c2 * D + i] * (double)wk[(size_t)c2 * D + j];
                        wqk[(size_t)i * D + j] = (float)acc;
                    }
            }
            tiles(wqk.data(), D, 0, a + A::wqk);
            tiles(W(p + ".attention.value.weight"), D, 0, a + A::wv);
            if (side == 1 && b == 0) {
                // node side, block 0 in fp64 (node_f64_body): node_free_code's encoder + this block's Wqk, Wv, LayerNorm
                const F64Blob F = F64Blob::make(D, C);
                float* f = PK + o.f64;
                gnnmp_pack_f64_ops(W("node_free_code.0.weight"), D, C, 0, C, 0, f + F.w1);
                std::memcpy(f + F.b1, W("node_free_code.0.bias"), sizeof(float) * D);
                gnnmp_pack_f64_ops(W("node_free_code.2.weight"), D, D, 0, D, 0, f + F.w2);
                std::memcpy(f + F.b2, W("node_free_code.2.bias"), sizeof(float) * D);
                gnnmp_pack_f64_ops(wqk.data(), D, D, 0, D, 1, f + F.wqk);               // these two run on the f32 instruction
                gnnmp_pack_f64_ops(W(p + ".attention.value.weight"), D, D, 0, D, 1, f + F.wv);
                std::memcpy(f + F.lng, W(p + ".attention.layer_norm.weight"), sizeof(float) * D);
                std::memcpy(f + F.lnb, W(p + ".attention.layer_norm.bias"), sizeof(float) * D);
            }
            vec(W(p + ".attention.layer_norm.weight"), a + A::ln1g);
            vec(W(p + ".attention.layer_norm.bias"), a + A::ln1b);
            tiles(W(p + ".map_feed.w_1.weight"), D, 0, a + A::w1);
            vec(W(p + ".map_feed.w_1.bias"), a + A::b1);
            tiles(W(p + ".map_feed.w_2.weight"), D, 0, a + A::w2);
            vec(W(p + ".map_feed.w_2.bias"), a + A::b2);
            vec(W(p + ".map_feed.layer_norm.weight"), a + A::ln2g);
            vec(W(p + ".map_feed.layer_norm.bias"), a + A::ln2b);
            float* q = ob + L.blk0 + (size_t)b * L.blk_stride;
            tiles(wqk.data(), D, 0, q + L.wk);                       // obstacle keys are premultiplied: K' = Wqk code
            tiles(W(p + ".attention.value.weight"), D, 0, q + L.wv);
            tiles(W(p + ".obs_feed.w_1.weight"), D, 0, q + L.fw1);
            vec(W(p + ".obs_feed.w_1.bias"), q + L.fb1);
            tiles(W(p + ".obs_feed.w_2.weight"), D, 0, q + L.fw2);
            vec(W(p + ".obs_feed.w_2.bias"), q + L.fb2);
            vec(W(p + ".obs_feed.layer_norm.weight"), q + L.lng);
            vec(W(p + ".obs_feed.layer_norm.bias"), q + L.lnb);
        }
    }

    // --- message first layer W1 = [xj-xi | xj | xi | EF | EC]  (model.py:38-39)
    const float* w1 = W("process.lin_0.0.weight");
    std::vector<float> wsrc((size_t)D * D), wdst((size_t)D * D);
    for (int i = 0; i < D; ++i)
        for (int k = 0; k < D; ++k) {
            const float a = w1[(size_t)i * 5 * D + k], b = w1[(size_t)i * 5 * D + D + k], c = w1[(size_t)i * 5 * D + 2 * D + k];
            wsrc[(size_t)i * D + k] = a + b;      // multiplies x_j (source)
            wdst[(size_t)i * D + k] = c - a;      // multiplies x_i (target)
        }
    // --- policy first layer P0 = [D_s | D_s - D_t | EF]  (model.py:145)
    const float* p0 = W("policy.0.weight");
    std::vector<float> wps((size_t)D * D), wpt((size_t)D * D);
    for (int i = 0; i < D; ++i)
        for (int k = 0; k < D; ++k) {
            wps[(size_t)i * D + k] = p0[(size_t)i * 3 * D + k] + p0[(size_t)i * 3 * D + D + k];
            wpt[(size_t)i * D + k] = p0[(size_t)i * 3 * D + D + k];
        }
    {
        using L = OutEBlob<D, P>;
        float* q = PK + o.out_e;
        tiles(w1, 5 * D, 3 * D, q + L::w1d);
        tiles(w1, 5 * D, 4 * D, q + L::w1e);
        vec(W("process.lin_0.0.bias"), q + L::b1);
        tiles(p0, 3 * D, 2 * D, q + L::wpc);
        vec(W("policy.0.bias"), q + L::bp0);
    }
    const float* we = W("encoder.weight");      // [NC | NF | H0 | H]  (model.py:141)
    const float* wd = W("decoder.weight");      // [NC | H]            (model.py:143)
    const float* wl1 = W("process.lin_1.weight");   // [X | agg]       (model.py:36)
    {
        using L = OutNBlob<D, P>;
        float* q = PK + o.out_n;
        tiles(we, 4 * D, 0, q + L::we_nc);
        tiles(we, 4 * D, D, q + L::we_nf);
        vec(W("encoder.bias"), q + L::be);
        const float* ge = W("goal_encoder");
        vec(matvec(we, D, 4 * D, 2 * D, D, ge).data(), q + L::weg);
        vec(matvec(we, D, 4 * D, 3 * D, D, ge).data(), q + L::wehg);
        tiles(wsrc.data(), D, 0, q + L::wsrc);
        tiles(wdst.data(), D, 0, q + L::wdst);
        tiles(wd, 2 * D, 0, q + L::wd_nc);
        vec(W("decoder.bias"), q + L::bd);
    }
    for (int last = 0; last < 2; ++last) {
        using L = MpNBlob<D, P>;
        float* q = PK + (last ? o.mpn_last : o.mpn);
        tiles(wl1, 2 * D, 0, q + L::wlx);
        tiles(wl1, 2 * D, D, q + L::wla);
        vec(W("process.lin_1.bias"), q + L::bl);
        if (!last) {
            tiles(we, 4 * D, 3 * D, q + L::m1);
            tiles(wsrc.data(), D, 0, q + L::m2);
            tiles(wdst.data(), D, 0, q + L::m3);
        } else {
            tiles(wd, 2 * D, D, q + L::m1);
            tiles(wps.data(), D, 0, q + L::m2);
            tiles(wpt.data(), D, 0, q + L::m3);
        }
    }
    {
        using L = MpEBlob<D, P>;
        float* q = PK + o.mpe;
        tiles(W("process.lin_0.2.weight"), D, 0, q + L::w2);
        vec(W("process.lin_0.2.bias"), q + L::b2);
    }
    {
        using L = PolBlob<D, P>;
        float* q = PK + o.pol;
        tiles(W("policy.2.weight"), D, 0, q + L::w2);
        vec(W("policy.2.bias"), q + L::b2);
        vec(W("policy.4.weight"), q + L::w3);
    }
}

}  // namespace

extern "C" int gnnmp_explorer_create(gnnmp_explorer** out, const gnnmp_explorer_dims* dims, const float* weights_host,
                                     size_t n_floats, int device) {
    if (!out || !dims || !weights_host) return GNNMP_ERR_NULL;
    if (!dims_ok(*dims)) return GNNMP_ERR_DIMS;
    Blob B;
    B.man = explorer_manifest(*dims);
    B.base = weights_host;
    int64_t tot = 0;
    for (auto& e : B.man) { B.off.push_back(tot); tot += e.numel(); }
    if ((int64_t)n_floats != tot) return GNNMP_ERR_WEIGHTS;
    gnnmp_explorer* h = new gnnmp_explorer();
    h->dims = *dims;
    h->device = device;
    h->w_dev = nullptr;
    h->w_raw_dev = nullptr;
    h->man = B.man; h->man_off = B.off; h->n_raw = tot;
    h->prof = new StageProf();
    std::vector<float> packed;
    const int P = dims->mlp_dtype;
    if (dims->embed_size == 32) {
        if (P == 2) pack_explorer<32, 2>(B, *dims, h, packed);
        else if (P == 1) pack_explorer<32, 1>(B, *dims, h, packed);
        else pack_explorer<32, 0>(B, *dims, h, packed);
    } else {
        if (P == 2) pack_explorer<64, 2>(B, *dims, h, packed);
        else if (P == 1) pack_explorer<64, 1>(B, *dims, h, packed);
        else pack_explorer<64, 0>(B, *dims, h, packed);
    }
    // the handle lives on `device`; the caller's current device is restored before returning
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) {
        int ncu = 0;
        e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
        h->n_cu = (ncu > 0 ? ncu : 256) & ~7;
        if (h->n_cu < 8) h->n_cu = 8;
        const char* env = std::getenv("GNNMP_RESIDENT");
        h->resident = !(env && env[0] == '0');
        const char* env2 = std::getenv("GNNMP_PRE_BOTH");
        h->resident_both = !(env2 && env2[0] == '0');
        const char* env3 = std::getenv("GNNMP_NODE_F64");
        h->node_f64 = !(env3 && env3[0] == '0');
    }
    if (e == hipSuccess) e = hipMalloc((void**)&h->w_dev, packed.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->w_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&h->w_raw_dev, (size_t)tot * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->w_raw_dev, weights_host, (size_t)tot * sizeof(float), hipMemcpyHostToDevice);
    if (prev_dev >= 0 && prev_dev != device) (void)hipSetDevice(prev_dev);
    if (e != hipSuccess) {
        if (h->w_dev) (void)hipFree(h->w_dev);
        if (h->w_raw_dev) (void)hipFree(h->w_raw_dev);
        delete h->prof;
        delete h;
        return hip_fail(e);
    }
    *out = h;
    return GNNMP_OK;
}

extern "C" int gnnmp_explorer_destroy(gnnmp_explorer* h) {
    if (!h) return GNNMP_ERR_NULL;
    if (h->w_dev) (void)hipFree(h->w_dev);
    if (h->w_raw_dev) (void)hipFree(h->w_raw_dev);
    if (h->prof) {
        for (auto& r : h->prof->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        for (auto e : h->prof->pool) (void)hipEventDestroy(e);
        delete h->prof;
    }
    delete h;
    return GNNMP_OK;
}

extern "C" int gnnmp_explorer_profile(gnnmp_explorer* h, int enable) {
    if (!h) return GNNMP_ERR_NULL;
    h->prof->on = enable != 0;
    // create the events up front so that no hipEventCreate lands inside a timed region
    if (h->prof->on)
        while (h->prof->pool.size() < 2048) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            h->prof->pool.push_back(e);
        }
    return GNNMP_OK;
}

extern "C" int gnnmp_explorer_profile_read(gnnmp_explorer* h, double* ms_sum, int64_t* launches) {
    if (!h || !ms_sum || !launches) return GNNMP_ERR_NULL;
    for (int i = 0; i < GNNMP_N_STAGES; ++i) { ms_sum[i] = 0.0; launches[i] = 0; }
    for (auto& r : h->prof->recs) {
        HIP_TRY(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        ms_sum[r.stage] += ms;
        launches[r.stage] += 1;
        h->prof->pool.push_back(r.a);
        h->prof->pool.push_back(r.b);
    }
    h->prof->recs.clear();
    return GNNMP_OK;
}

// ---------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------
namespace {

struct Carve {
    // sizes
    int G, Npad, Epad, ot_max, kv_stride, D;
    // offsets in bytes
    size_t node_ptr_pad, edge_ptr_pad, goal_node, dense_ptr, in_ptrs, prep_hist, gstat;
    size_t zero_beg, deg, cursor, zero_end;
    size_t ff_beg, ntile_graph, etile_graph, csr, ff_end, tile_meta, rec32, blk_span;
    size_t row_beg;
    size_t XI, X, A, A2, B, DN, H, Ke, PE, kv_e, kv_n, M0;
    size_t total;
};

int round_up_i(int x, int m) { return (x + m - 1) / m * m; }

bool carve(const gnnmp_explorer* h, const gnnmp_batch* b, Carve& c) {
    if (b->n_graphs < 1 || b->total_nodes < 0 || b->total_edges < 0 || b->total_obstacles < 0 || b->max_obstacles < 0)
        return false;
    const int D = h->dims.embed_size, NT = D / 32;
    c.D = D;
    c.G = b->n_graphs;
    const long long np = (long long)b->total_nodes + (long long)c.G * (kPad - 1);
    const long long ep = (long long)b->total_edges + (long long)c.G * (kPad - 1);
    if (np > 0x3fffffff || ep > 0x3fffffff) return false;
    c.Npad = round_up_i((int)np, kPad);
    c.Epad = round_up_i((int)ep, kPad);
    c.ot_max = (b->max_obstacles + 31) / 32;
    if (c.ot_max < 1) c.ot_max = 1;
    c.kv_stride = 2 * c.ot_max * NT * tile_unit(h->dims.mlp_dtype);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    c.node_ptr_pad = take(sizeof(int) * (c.G + 1));
    c.edge_ptr_pad = take(sizeof(int) * (c.G + 1));
    c.goal_node = take(sizeof(int) * c.G);
    c.dense_ptr = take(sizeof(long long) * (c.G + 1));
    c.in_ptrs = take(sizeof(int) * 6);          // prefix arrays of a single graph given by its totals (NULL prefix pointers)
    c.gstat = take(sizeof(int) * kGstatStride * (size_t)c.G);      // device-side status of the forward (gnnmp_explorer_status)
    c.zero_beg = o;
    c.deg = take(sizeof(int) * c.Npad);
    c.zero_end = o;
    c.cursor = take(sizeof(int) * (size_t)(b->total_edges > 0 ? b->total_edges : 1));   // per-edge rank in its segment
    c.ff_beg = o;
    c.ntile_graph = take(sizeof(int) * (c.Npad / 32));
    c.etile_graph = take(sizeof(int) * (c.Epad / 32));
    c.tile_meta = take(sizeof(int) * (c.Epad / 32));            // -1 = tile beyond the last graph
    c.csr = take(sizeof(int) * 4 * (size_t)c.Epad);
    c.ff_end = o;
    c.rec32 = take(sizeof(int) * (size_t)c.Epad);
    c.blk_span = take(sizeof(int) * 2 * (size_t)(c.Npad / kPad));
    c.row_beg = take(sizeof(int) * c.Npad);
    {   // per-part target histograms + first slots of the prep stage's two-launch form (large graphs only)
        const int parts = prep_parts(c.G, b->total_edges);
        c.prep_hist = take(parts > 1 ? sizeof(int) * 2 * (size_t)parts * c.Npad : 0);
    }
    const size_t nrow = sizeof(float) * (size_t)c.Npad * D, erow = sizeof(float) * (size_t)c.Epad * D;
    c.XI = take(nrow); c.X = take(nrow); c.A = take(nrow); c.A2 = take(nrow); c.B = take(nrow); c.DN = take(nrow);
    c.H = take(nrow);
    c.M0 = take(h->dims.mlp_dtype == GNNMP_BF16 ? 0 : nrow);      // node_f64_body -> node pre kernel (fp32-class modes)
    c.Ke = take(erow); c.PE = take(erow);
    c.kv_e = take(sizeof(float) * (size_t)c.G * 3 * c.kv_stride);
    c.kv_n = take(sizeof(float) * (size_t)c.G * 3 * c.kv_stride);
    c.total = o;
    return true;
}

template <class T>
T* at(void* ws, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(ws) + off); }

// LDS plan of pre_kernel: weight region + K/V chunk region
struct PrePlan { int waves, ot_chunk, wregion; size_t lds_bytes; };

// blob sizes by (d, precision): T = (d/32)^2 * tile_unit(P), V = d
int out_e_size(int D, int P) { return 3 * tile_floats(D, P) + 2 * vec_floats(D); }
int out_n_size(int D, int P) { return 5 * tile_floats(D, P) + 4 * vec_floats(D); }
int att_staged(int D, int P) { return 4 * tile_floats(D, P); }

PrePlan plan_pre(int D, int P, int ot_max, int enc_size, int out_size, bool use_obs) {
    const int NT = D / 32;
    const int att = att_staged(D, P);
    int wr = enc_size > out_size ? enc_size : out_size;
    if (use_obs && att > wr) wr = att;
    wr = (wr + 3) & ~3;
    PrePlan p;
    p.wregion = wr;
    p.waves = (D == 32) ? 4 : 8;
    const size_t per_tile = (size_t)2 * NT * tile_unit(P) * sizeof(float);
    p.ot_chunk = 1;
    if (use_obs) {
        // prefer the highest workgroups-per-CU residency that still keeps a graph's whole K/V resident
        const size_t limit = 163840, wbytes = (size_t)wr * sizeof(float);
        for (int target = (D == 32 ? 3 : 1); target >= 1; --target) {
            const size_t budget = (limit / target) & ~(size_t)1023;
            if (budget < wbytes + per_tile) continue;
            int ch = (int)((budget - wbytes) / per_tile);
            if (ch > ot_max) ch = ot_max;
            p.ot_chunk = ch;
            if (ch >= ot_max) break;
        }
    }
    p.lds_bytes = (size_t)wr * sizeof(float) + (use_obs ? (size_t)p.ot_chunk * per_tile : 0);
    return p;
}

}  // namespace

extern "C" int gnnmp_explorer_workspace_bytes(const gnnmp_explorer* h, const gnnmp_batch* shape, size_t* bytes) {
    if (!h || !shape || !bytes) return GNNMP_ERR_NULL;
    Carve c;
    if (!carve(h, shape, c)) return GNNMP_ERR_ARG;
    *bytes = c.total;
    return GNNMP_OK;
}

namespace {
int forward_impl(const gnnmp_explorer* h, const gnnmp_batch* b, int loop, int use_obstacles, float* edge_scores, float* dense,
                 void* ws, size_t ws_bytes, void* hip_stream, float* om_nodes, float* om_edges, bool pre_only, int32_t* status_out = nullptr);
}
extern "C" int gnnmp_explorer_forward(const gnnmp_explorer* h, const gnnmp_batch* b, int loop, int use_obstacles,
                                      float* edge_scores, float* dense, void* ws, size_t ws_bytes, void* hip_stream) {
    return forward_impl(h, b, loop, use_obstacles, edge_scores, dense, ws, ws_bytes, hip_stream, nullptr, nullptr, false);
}
extern "C" int gnnmp_explorer_forward_ex(const gnnmp_explorer* h, const gnnmp_batch* b, int loop, int use_obstacles,
                                         float* edge_scores, float* dense, void* ws, size_t ws_bytes, void* hip_stream,
                                         int32_t* status_out) {
    return forward_impl(h, b, loop, use_obstacles, edge_scores, dense, ws, ws_bytes, hip_stream, nullptr, nullptr, false, status_out);
}
extern "C" int gnnmp_explorer_status_words(const gnnmp_batch* shape, size_t* n_words) {
    if (!shape || !n_words) return GNNMP_ERR_NULL;
    if (shape->n_graphs < 1) return GNNMP_ERR_ARG;
    *n_words = (size_t)kGstatStride * (size_t)shape->n_graphs;
    return GNNMP_OK;
}

namespace gnnmp { hipError_t launch_status_copy(const int* src, int* dst_host_mapped, int n, hipStream_t st); }
extern "C" int gnnmp_status_copy(const int32_t* src_device, int32_t* dst_host_mapped, int32_t n_words, void* hip_stream) {
    if (!src_device || !dst_host_mapped) return GNNMP_ERR_NULL;
    if (n_words < 0) return GNNMP_ERR_ARG;
    HIP_TRY(gnnmp::launch_status_copy(src_device, dst_host_mapped, n_words, static_cast<hipStream_t>(hip_stream)));
    return GNNMP_OK;
}

extern "C" int gnnmp_explorer_status_region(const gnnmp_explorer* h, const gnnmp_batch* shape, size_t* offset, size_t* bytes) {
    if (!h || !shape || !offset || !bytes) return GNNMP_ERR_NULL;
    Carve c;
    if (!carve(h, shape, c)) return GNNMP_ERR_ARG;
    *offset = c.gstat;
    *bytes = sizeof(int) * kGstatStride * (size_t)c.G;
    return GNNMP_OK;
}

extern "C" int gnnmp_explorer_status_decode(const int32_t* words_host, int n_graphs, int32_t* first_graph) {
    if (!words_host || n_graphs < 0) return GNNMP_ERR_NULL;
    int caps = -1, ids = -1;
    for (int g = 0; g < n_graphs; ++g) {
        if (caps < 0 && (words_host[kGstatStride * g] & 1)) caps = g;
        for (int k = 1; k < kGstatStride && ids < 0; ++k)
            if (words_host[kGstatStride * g + k] & 2) ids = g;
    }
    if (ids >= 0) { if (first_graph) *first_graph = ids; return GNNMP_ERR_INDEX; }
    if (caps >= 0) { if (first_graph) *first_graph = caps; return GNNMP_ERR_CAPS; }
    if (first_graph) *first_graph = -1;
    return GNNMP_OK;
}

extern "C" int gnnmp_explorer_status(const gnnmp_explorer* h, const gnnmp_batch* shape, const void* ws, size_t ws_bytes,
                                     void* hip_stream, int32_t* first_graph) {
    if (!h || !shape || !ws) return GNNMP_ERR_NULL;
    Carve c;
    if (!carve(h, shape, c)) return GNNMP_ERR_ARG;
    if (ws_bytes < c.total) return GNNMP_ERR_WORKSPACE;
    std::vector<int32_t> host((size_t)kGstatStride * c.G);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    HIP_TRY(hipMemcpyAsync(host.data(), static_cast<const char*>(ws) + c.gstat, host.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return gnnmp_explorer_status_decode(host.data(), c.G, first_graph);
}

namespace {
// om_nodes / om_edges: optional [Npad, d] / [Epad, d] outputs of the attention stacks (training path); pre_only: stop
// after the pre kernels (CSR, goal node, NF / EF are what the training path needs)
int forward_impl(const gnnmp_explorer* h, const gnnmp_batch* b, int loop, int use_obstacles, float* edge_scores, float* dense,
                 void* ws, size_t ws_bytes, void* hip_stream, float* om_nodes, float* om_edges, bool pre_only, int32_t* status_out) {
    if (!h || !b || !ws) return GNNMP_ERR_NULL;
    if (!b->v || !b->goal) return GNNMP_ERR_NULL;
    // ONE graph may be given by its totals alone (all three prefix pointers NULL): the reference's call (model.py:115)
    // has no prefix arrays, and building them on the device costs the caller three host-to-device copies per call
    const bool implicit = !b->node_ptr && !b->edge_ptr && !b->obs_ptr;
    if (implicit ? b->n_graphs != 1 : (!b->node_ptr || !b->edge_ptr || !b->obs_ptr)) return GNNMP_ERR_NULL;
    if (b->total_edges > 0 && (!b->edge_index || (!edge_scores && !pre_only))) return GNNMP_ERR_NULL;
    if (use_obstacles && b->total_obstacles > 0 && !b->obstacles) return GNNMP_ERR_NULL;
    if (loop < 1) return GNNMP_ERR_ARG;                    // model.py:139-145: decode unbound for loop = 0
    Carve c;
    if (!carve(h, b, c)) return GNNMP_ERR_ARG;
    if (ws_bytes < c.total || (reinterpret_cast<uintptr_t>(ws) & 255)) return GNNMP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int D = h->dims.embed_size, C = h->dims.config_size, P = h->dims.mlp_dtype;
    const float* W = h->w_dev;

    StageProf* prof = h->prof;
    PrepParams q;
    const int* node_ptr = implicit ? at<int>(ws, c.in_ptrs) : b->node_ptr;
    const int* edge_ptr = implicit ? at<int>(ws, c.in_ptrs) + 2 : b->edge_ptr;
    const int* obs_ptr = implicit ? at<int>(ws, c.in_ptrs) + 4 : b->obs_ptr;
    {
    StageScope sc(prof, GNNMP_STAGE_PREP, st);
    q.G = c.G; q.E = b->total_edges; q.C = C;
    q.edge_index = reinterpret_cast<const long long*>(b->edge_index);
    q.single_out = implicit ? at<int>(ws, c.in_ptrs) : nullptr;
    q.single_n = b->total_nodes; q.single_e = b->total_edges; q.single_o = b->total_obstacles;
    q.node_ptr = node_ptr; q.edge_ptr = edge_ptr; q.v = b->v; q.goal = b->goal;
    q.node_ptr_pad = at<int>(ws, c.node_ptr_pad); q.edge_ptr_pad = at<int>(ws, c.edge_ptr_pad);
    q.dense_ptr = at<long long>(ws, c.dense_ptr);
    q.deg = at<int>(ws, c.deg); q.cursor = at<int>(ws, c.cursor); q.row_beg = at<int>(ws, c.row_beg);
    q.ntile_graph = at<int>(ws, c.ntile_graph); q.etile_graph = at<int>(ws, c.etile_graph);
    q.csr = at<int4>(ws, c.csr);
    q.goal_node = at<int>(ws, c.goal_node);
    q.tile_meta = at<int>(ws, c.tile_meta); q.n_etiles = c.Epad / 32;
    q.blk_span = at<int2>(ws, c.blk_span);
    q.obs_ptr = implicit ? nullptr : b->obs_ptr;
    q.obs_cap = use_obstacles ? 32 * c.ot_max : 0x7fffffff;
    // the status words of this forward: the workspace region (read back by gnnmp_explorer_status), or -- gnnmp_explorer_forward_ex --
    // memory of the caller's that the device can write (pinned host memory: no copy behind the forward, the words simply arrive)
    q.gstat = status_out ? reinterpret_cast<int*>(status_out) : at<int>(ws, c.gstat);
    HIP_TRY(launch_prep(q, c.Npad, c.Epad, at<int>(ws, c.prep_hist), st));
    // zero-fill of policy_output (model.py:148); sum_g N_g^2 is read from dense_ptr[G] on the device
    if (dense) HIP_TRY(launch_zero_dense(dense, q.dense_ptr + c.G, st));
    // training path: CSR slots in a timing-independent order (the gradients sum over slots)
    if (pre_only) HIP_TRY(t_sort_csr(c.Npad, q.csr, q.row_beg, q.deg, st));
    }

    const bool use_obs = use_obstacles != 0;
    bool use_m0 = false;
    if (use_obs) {
        ObsParams op;
        op.obstacles = b->obstacles; op.obs_ptr = obs_ptr; op.S = h->dims.obs_size;
        op.w[0] = W + h->off.obs_n; op.w[1] = W + h->off.obs_e;
        op.blob = h->obs;
        op.kv[0] = at<float>(ws, c.kv_n); op.kv[1] = at<float>(ws, c.kv_e);
        op.kv_stride = c.kv_stride; op.ot_max = c.ot_max;
        NodeF64Params nq;
        nq.v = b->v; nq.C = C; nq.node_ptr = node_ptr; nq.node_ptr_pad = q.node_ptr_pad; nq.ntile_graph = q.ntile_graph;
        nq.w = W + h->off.f64; nq.blob = F64Blob::make(D, C);
        nq.m0 = at<float>(ws, c.M0);
        nq.groups = c.Npad / 64 > 2 * h->n_cu ? 4 : 1;                          // kPad = 256 rows = 4 groups
        {
            static const char* ord = std::getenv("GNNMP_F64_FIRST");
            // measured: cfg 2 (d = 32, 1024 double-precision workgroups = two rounds) 0.145 -> 0.139 ms with them first;
            // kuka7 fp32 (d = 64, 512 = one round) 0.153 -> 0.162 ms, so d = 64 keeps the obstacle workgroups in front
            nq.f64_first = ord ? (ord[0] != '0') : (nq.groups > 1 && D == 32);
        }
        nq.n_wg = (P != GNNMP_BF16 && h->node_f64) ? c.Npad / (64 * nq.groups) : 0;
        use_m0 = nq.n_wg > 0;
        StageScope sc(prof, GNNMP_STAGE_OBS, st);
        HIP_TRY(launch_obs(D, P, op, nq, c.G, st));
    }

    PreParams pp[2];
    PrePlan plan[2];
    size_t res_lds[2] = {0, 0};                            // LDS bytes of the resident variant, 0 = not applicable
    const size_t res_bytes = ((size_t)3 * (att_staged(D, P) + 6 * vec_floats(D)) + (size_t)3 * c.kv_stride) * sizeof(float) + 64;
    const bool resident_ok = ((D == 32 && P == 0) || P == 1) && use_obs && res_bytes <= 163840 && h->resident;
    for (int edge = 0; edge < 2; ++edge) {
        PreParams& p = pp[edge];
        p.v = b->v; p.goal = b->goal; p.C = C;
        p.node_ptr = node_ptr; p.node_ptr_pad = q.node_ptr_pad;
        p.tile_graph = edge ? q.etile_graph : q.ntile_graph;
        p.csr = q.csr;
        p.rec32 = at<int>(ws, c.rec32);
        p.obs_ptr = obs_ptr; p.goal_node = q.goal_node;
        p.enc = W + (edge ? h->off.enc_e : h->off.enc_n);
        p.encb = edge ? h->enc_e : h->enc_n;
        p.att = W + (edge ? h->off.att_e : h->off.att_n);
        p.out = W + (edge ? h->off.out_e : h->off.out_n);
        p.out_size = edge ? out_e_size(D, P) : out_n_size(D, P);
        p.kv = at<float>(ws, edge ? c.kv_e : c.kv_n);
        p.kv_stride = c.kv_stride; p.ot_max = c.ot_max;
        plan[edge] = plan_pre(D, P, c.ot_max, p.encb.size, p.out_size, use_obs);
        p.ot_chunk = plan[edge].ot_chunk; p.wregion = plan[edge].wregion; p.use_obstacles = use_obs ? 1 : 0;
        p.om = edge ? om_edges : om_nodes;
        p.m0 = (!edge && use_m0) ? at<float>(ws, c.M0) : nullptr;
        if (edge) { p.o0 = at<float>(ws, c.Ke); p.o1 = at<float>(ws, c.PE); p.o2 = p.o3 = p.o4 = nullptr; }
        else {
            p.o0 = at<float>(ws, c.XI); p.o1 = at<float>(ws, c.X); p.o2 = at<float>(ws, c.A);
            p.o3 = at<float>(ws, c.B); p.o4 = at<float>(ws, c.DN);
        }
        // resident variant: all three blocks' weights + the graph's K/V for all three blocks in LDS
        p.ptr_pad_total = edge ? q.edge_ptr_pad : q.node_ptr_pad;
        p.tile_meta = q.tile_meta;
        p.G = c.G;
        p.out_in_lds = 0;
        if (resident_ok) {
            res_lds[edge] = res_bytes;
            if (res_bytes + (size_t)p.out_size * sizeof(float) <= 163840) {      // the epilogue weights fit as well
                p.out_in_lds = 1;
                res_lds[edge] += (size_t)p.out_size * sizeof(float);
            }
        }
    }
    // small batches: both stages in one launch, side by side (see pre_resident_both_kernel); `share` 32-row tiles per
    // 12-wave workgroup, as few as keep all workgroups of both roles resident at once
    bool both_done = false;
    if (resident_ok && D == 32 && b->total_edges > 0 && h->resident_both) {
        const int nt = c.Npad / 32, et = c.Epad / 32;
        for (int share = 1; share <= 12 && !both_done; ++share) {
            const int nb = ((nt + share - 1) / share + 7) & ~7, eb = ((et + share - 1) / share + 7) & ~7;
            if (nb + eb > h->n_cu) continue;
            StageScope sc(prof, GNNMP_STAGE_EDGE_PRE, st);
            HIP_TRY(launch_pre_resident_both(D, P, pp[0], pp[1], res_lds[0] > res_lds[1] ? res_lds[0] : res_lds[1], nb, eb, st));
            both_done = true;
        }
    }
    for (int edge = 0; edge < 2 && !both_done; ++edge) {
        if (edge && b->total_edges == 0) continue;
        StageScope sc(prof, edge ? GNNMP_STAGE_EDGE_PRE : GNNMP_STAGE_NODE_PRE, st);
        if (res_lds[edge]) HIP_TRY(launch_pre_resident(D, P, edge != 0, pp[edge], res_lds[edge], h->n_cu, st));
        else HIP_TRY(launch_pre(D, P, edge != 0, plan[edge].waves, pp[edge], (edge ? c.Epad : c.Npad) / 32, plan[edge].lds_bytes, st));
    }

    if (pre_only) return GNNMP_OK;

    // message passing: one fused launch per iteration (edge phase + node phase per 32-node tile); the gathered A rows
    // ping-pong between two buffers because other tiles still read this iteration's A while a tile writes the next one
    float* Abuf[2] = {at<float>(ws, c.A), at<float>(ws, c.A2)};
    int cur = 0;
    for (int it = 0; it < loop; ++it) {
        const bool last = (it == loop - 1);
        MpFusedParams f;
        f.rec32 = at<int>(ws, c.rec32); f.row_beg = q.row_beg; f.deg = q.deg; f.ntile_graph = q.ntile_graph; f.node_ptr_pad = q.node_ptr_pad;
        f.A = Abuf[cur]; f.B = at<float>(ws, c.B); f.Ke = at<float>(ws, c.Ke);
        f.X = at<float>(ws, c.X); f.R = at<float>(ws, last ? c.DN : c.XI);
        f.we = W + h->off.mpe; f.wn = W + (last ? h->off.mpn_last : h->off.mpn);
        f.wn_std = W + h->off.mpn; f.last = last ? 1 : 0;
        f.Hout = at<float>(ws, c.H); f.Xout = at<float>(ws, c.X); f.Aout = Abuf[cur ^ 1]; f.Bout = at<float>(ws, c.B);
        f.n_tiles = c.Npad / 32;
        f.tpw = 1;
        f.order = 0;
        f.blk_span = q.blk_span;
        f.trace = nullptr;
        f.G = c.G;
        f.store_h = last ? 1 : 0;
        StageScope sc(prof, GNNMP_STAGE_MP, st);
        HIP_TRY(launch_mp_fused(D, P, f, st));
        cur ^= 1;
    }

    if (b->total_edges > 0) {
        PolicyParams p;
        p.csr = q.csr; p.etile_graph = q.etile_graph;
        p.node_ptr = node_ptr; p.node_ptr_pad = q.node_ptr_pad; p.dense_ptr = q.dense_ptr;
        p.PS = Abuf[cur]; p.PT = at<float>(ws, c.B); p.PE = at<float>(ws, c.PE);
        p.w = W + h->off.pol;
        p.scores = edge_scores; p.dense = dense;
        p.n_tiles = c.Epad / 32;
        StageScope sc(prof, GNNMP_STAGE_POLICY, st);
        HIP_TRY(launch_policy(D, P, p, st));
    }
    return GNNMP_OK;
}
}  // namespace

extern "C" int gnnmp_explorer_debug_tap(const gnnmp_explorer* h, const gnnmp_batch* b, int which, float* dst, void* ws,
                                        size_t ws_bytes, void* hip_stream) {
    if (!h || !b || !dst || !ws) return GNNMP_ERR_NULL;
    Carve c;
    if (!carve(h, b, c)) return GNNMP_ERR_ARG;
    if (ws_bytes < c.total) return GNNMP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int D = h->dims.embed_size;
    size_t src;
    switch (which) {
        case 0: src = c.XI; break;
        case 1: src = c.H; break;
        case 2:
            if (h->dims.mlp_dtype == GNNMP_BF16) return GNNMP_ERR_DIMS;      // X rows are stored in bf16 in that mode
            src = c.X; break;
        case 3:
            HIP_TRY(launch_goal_tap(c.G, at<int>(ws, c.goal_node), at<int>(ws, c.node_ptr_pad), dst, st));
            return GNNMP_OK;
        default: return GNNMP_ERR_ARG;
    }
    HIP_TRY(launch_unpad_rows(c.G, b->total_nodes, D, b->node_ptr ? b->node_ptr : at<int>(ws, c.in_ptrs), at<int>(ws, c.node_ptr_pad),
                              at<float>(ws, src), dst, st));
    return GNNMP_OK;
}

// =============================================================================================
// smoother
// =============================================================================================
namespace {

std::vector<Entry> smoother_manifest(const gnnmp_smoother_dims& d) {
    const int C = d.config_size, D = d.embed_size;
    std::vector<Entry> m;
    auto lin = [&](const std::string& n, int out, int in) {
        m.push_back({n + ".weight", out, in});
        m.push_back({n + ".bias", out, 0});
    };
    lin("node_code.0", D, C + 3);
    m.push_back({"node_code.1.weight", D, 0});
    m.push_back({"node_code.1.bias", D, 0});
    m.push_back({"node_code.1.running_mean", D, 0});
    m.push_back({"node_code.1.running_var", D, 0});
    lin("node_code.3", D, D);
    lin("process.lin_0.0", D, 3 * D);
    lin("process.lin_0.2", D, D);
    lin("process.lin_1.0", D, D);
    lin("process.lin_1.2", D, D);
    lin("smooth_node", C, D);
    return m;
}

bool sm_dims_ok(const gnnmp_smoother_dims& d) {
    return d.config_size >= 1 && d.config_size <= 29 && (d.embed_size == 32 || d.embed_size == 64 || d.embed_size == 128) &&
           d.scale > 0.f && (d.mlp_dtype == GNNMP_F32 || d.mlp_dtype == GNNMP_BF16);
}

}  // namespace

struct gnnmp_smoother {
    gnnmp_smoother_dims dims;
    int device;
    float* w_dev;
    SmLayout L;
    float* w_raw_dev;     // the caller's weight blob as given (training path)
    std::vector<Entry> man;
    std::vector<int64_t> man_off;
    int64_t n_raw;
};

extern "C" int gnnmp_smoother_manifest(const gnnmp_smoother_dims* dims, int index, char* name, size_t name_cap,
                                       int64_t* numel) {
    if (!dims) return GNNMP_ERR_NULL;
    if (!sm_dims_ok(*dims)) return GNNMP_ERR_DIMS;
    const auto m = smoother_manifest(*dims);
    if (index < 0) return (int)m.size();
    if (index >= (int)m.size()) return GNNMP_ERR_ARG;
    if (name && name_cap) {
        std::strncpy(name, m[index].name.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (numel) *numel = m[index].numel();
    return GNNMP_OK;
}

extern "C" int gnnmp_smoother_create(gnnmp_smoother** out, const gnnmp_smoother_dims* dims, const float* weights_host,
                                     size_t n_floats, int device) {
    if (!out || !dims || !weights_host) return GNNMP_ERR_NULL;
    if (!sm_dims_ok(*dims)) return GNNMP_ERR_DIMS;
    Blob B;
    B.man = smoother_manifest(*dims);
    B.base = weights_host;
    int64_t tot = 0;
    for (auto& e : B.man) { B.off.push_back(tot); tot += e.numel(); }
    if ((int64_t)n_floats != tot) return GNNMP_ERR_WEIGHTS;
    const int C = dims->config_size, D = dims->embed_size;
    gnnmp_smoother* h = new gnnmp_smoother();
    h->dims = *dims;
    h->device = device;
    h->w_dev = nullptr;
    h->w_raw_dev = nullptr;
    h->man = B.man; h->man_off = B.off; h->n_raw = tot;
    const int PR = dims->mlp_dtype;
    h->L = SmLayout::make(PR, D, C);
    const SmLayout& L = h->L;
    std::vector<float> P(L.total, 0.f);
    auto ptiles = [&](const float* w, int out_f, int ld, float* dst) {
        if (PR) pack_a_tiles_bf16(w, out_f, ld, 0, D, dst); else gnnmp_pack_a_tiles(w, out_f, ld, 0, D, dst);
    };
    auto W = [&](const std::string& n) { return B.get(n); };
    // fold eval-mode BatchNorm (eps 1e-5) into node_code.0:  y = (W x + b - mean) * g + beta,  g = gamma / sqrt(var + eps)
    {
        const float *w0 = W("node_code.0.weight"), *b0 = W("node_code.0.bias"), *ga = W("node_code.1.weight"),
                    *be = W("node_code.1.bias"), *mu = W("node_code.1.running_mean"), *va = W("node_code.1.running_var");
        const int K = C + 3;
        std::vector<float> wf((size_t)D * K), bf(D);
        for (int i = 0; i < D; ++i) {
            const float g = ga[i] / std::sqrt(va[i] + 1e-5f);
            for (int k = 0; k < K; ++k) wf[(size_t)i * K + k] = w0[(size_t)i * K + k] * g;
            bf[i] = (b0[i] - mu[i]) * g + be[i];
        }
        if (PR) pack_a_small_bf16(wf.data(), D, K, 0, K, P.data() + L.as0); else gnnmp_pack_a_small(wf.data(), D, K, 0, K, P.data() + L.as0);
        gnnmp_pack_vec(bf.data(), D, P.data() + L.b0);
    }
    ptiles(W("node_code.3.weight"), D, D, P.data() + L.w3);
    gnnmp_pack_vec(W("node_code.3.bias"), D, P.data() + L.b3);
    {
        const float* w1 = W("process.lin_0.0.weight");      // [x_j - x_i | x_j | x_i]  model_smoother.py:37
        std::vector<float> ws((size_t)D * D), wd((size_t)D * D);
        for (int i = 0; i < D; ++i)
            for (int k = 0; k < D; ++k) {
                const float a = w1[(size_t)i * 3 * D + k], b = w1[(size_t)i * 3 * D + D + k], c = w1[(size_t)i * 3 * D + 2 * D + k];
                ws[(size_t)i * D + k] = a + b;
                wd[(size_t)i * D + k] = c - a;
            }
        ptiles(ws.data(), D, D, P.data() + L.wsrc);
        ptiles(wd.data(), D, D, P.data() + L.wdst);
        gnnmp_pack_vec(W("process.lin_0.0.bias"), D, P.data() + L.b00);
    }
    ptiles(W("process.lin_0.2.weight"), D, D, P.data() + L.w02);
    gnnmp_pack_vec(W("process.lin_0.2.bias"), D, P.data() + L.b02);
    ptiles(W("process.lin_1.0.weight"), D, D, P.data() + L.w10);
    gnnmp_pack_vec(W("process.lin_1.0.bias"), D, P.data() + L.b10);
    ptiles(W("process.lin_1.2.weight"), D, D, P.data() + L.w12);
    gnnmp_pack_vec(W("process.lin_1.2.bias"), D, P.data() + L.b12);
    {
        std::vector<float> ws((size_t)32 * D, 0.f), bs(32, 0.f);
        const float *w = W("smooth_node.weight"), *b = W("smooth_node.bias");
        for (int i = 0; i < C; ++i) {
            for (int k = 0; k < D; ++k) ws[(size_t)i * D + k] = w[(size_t)i * D + k];
            bs[i] = b[i];
        }
        ptiles(ws.data(), 32, D, P.data() + L.ws);
        gnnmp_pack_vec(bs.data(), 32, P.data() + L.bs);
    }
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void**)&h->w_dev, P.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->w_dev, P.data(), P.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&h->w_raw_dev, (size_t)tot * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->w_raw_dev, weights_host, (size_t)tot * sizeof(float), hipMemcpyHostToDevice);
    if (prev_dev >= 0 && prev_dev != device) (void)hipSetDevice(prev_dev);
    if (e != hipSuccess) {
        if (h->w_dev) (void)hipFree(h->w_dev);
        if (h->w_raw_dev) (void)hipFree(h->w_raw_dev);
        delete h;
        return hip_fail(e);
    }
    *out = h;
    return GNNMP_OK;
}

extern "C" int gnnmp_smoother_destroy(gnnmp_smoother* h) {
    if (!h) return GNNMP_ERR_NULL;
    if (h->w_dev) (void)hipFree(h->w_dev);
    if (h->w_raw_dev) (void)hipFree(h->w_raw_dev);
    delete h;
    return GNNMP_OK;
}

namespace {
struct SmCarve {
    int ecap, pcap;
    size_t cur, knn, e_src, e_dst, e_count, stat, seg_beg, seg_cnt, ff_beg, etile, ptile, ff_end, msg, tgt, tflag, tcnt, elist, plist, total;
};

bool sm_carve(const gnnmp_smoother* h, const gnnmp_smooth_batch* b, SmCarve& c) {
    if (b->n_problems < 1 || b->total_path < 0 || b->total_edges < 0 || b->total_free < 0 || b->total_collided < 0)
        return false;
    const int D = h->dims.embed_size, C = h->dims.config_size;
    const long long ecap = (long long)b->total_edges + (long long)kSmK * b->total_path + 64LL * b->n_problems + 32;
    const long long pcap = (long long)b->total_path + 64LL * b->n_problems + 32;
    if (ecap > 0x3fffffff || pcap > 0x3fffffff) return false;
    c.ecap = (int)((ecap + 31) / 32 * 32);
    c.pcap = (int)((pcap + 31) / 32 * 32);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    c.cur = take(sizeof(float) * (size_t)b->total_path * C);
    c.knn = take(sizeof(int) * (size_t)b->total_path * kSmK);
    c.e_src = take(sizeof(int) * c.ecap);
    c.e_dst = take(sizeof(int) * c.ecap);
    c.e_count = take(sizeof(int) * b->n_problems);
    c.stat = take(sizeof(int) * b->n_problems);                 // device-side status per problem (gnnmp_smoother_status)
    c.seg_beg = take(sizeof(int) * c.pcap);
    c.seg_cnt = take(sizeof(int) * c.pcap);
    c.ff_beg = o;
    c.etile = take(sizeof(int) * (c.ecap / 32));
    c.ptile = take(sizeof(int) * (c.pcap / 32));
    c.ff_end = o;
    c.msg = take(sizeof(float) * (size_t)c.ecap * D);
    c.tgt = take(sizeof(float) * (size_t)c.pcap * D);
    c.tflag = take(sizeof(int) * (c.pcap / 32));
    // compact lists of the tiles in use (the tile space is padded per problem): two pairs of counters (iteration parity) + the lists,
    // appended by the graph stage, read by the streamed-weights message kernel
    c.tcnt = take(sizeof(int) * 4);
    c.elist = take(sizeof(int) * (c.ecap / 32));
    c.plist = take(sizeof(int) * (c.pcap / 32));
    c.total = o;
    return true;
}
}  // namespace

extern "C" int gnnmp_smoother_workspace_bytes(const gnnmp_smoother* h, const gnnmp_smooth_batch* shape, size_t* bytes) {
    if (!h || !shape || !bytes) return GNNMP_ERR_NULL;
    SmCarve c;
    if (!sm_carve(h, shape, c)) return GNNMP_ERR_ARG;
    *bytes = c.total;
    return GNNMP_OK;
}

namespace gnnmp { bool sm_wants_tile_lists(int D, int P, int n_etiles); }      // smoother_kernels.hip
extern "C" int gnnmp_smoother_forward_ex(const gnnmp_smoother* h, const gnnmp_smooth_batch* b, int loop, float* out_path,
                                         void* ws, size_t ws_bytes, void* hip_stream, int32_t* status_out);
extern "C" int gnnmp_smoother_forward(const gnnmp_smoother* h, const gnnmp_smooth_batch* b, int loop, float* out_path,
                                      void* ws, size_t ws_bytes, void* hip_stream) {
    return gnnmp_smoother_forward_ex(h, b, loop, out_path, ws, ws_bytes, hip_stream, nullptr);
}
extern "C" int gnnmp_smoother_forward_ex(const gnnmp_smoother* h, const gnnmp_smooth_batch* b, int loop, float* out_path,
                                         void* ws, size_t ws_bytes, void* hip_stream, int32_t* status_out) {
    if (!h || !b || !ws || !out_path) return GNNMP_ERR_NULL;
    if (!b->path) return GNNMP_ERR_NULL;
    // ONE problem may be given by its totals alone (the four prefix pointers NULL): the reference's call has no prefix
    // arrays (model_smoother.py:104) and building them costs the caller a host-to-device copy per call
    const bool implicit = !b->path_ptr && !b->free_ptr && !b->coll_ptr && !b->edge_ptr;
    if (implicit ? b->n_problems != 1 : (!b->path_ptr || !b->free_ptr || !b->coll_ptr || !b->edge_ptr)) return GNNMP_ERR_NULL;
    if ((b->total_free > 0 && !b->free_pts) || (b->total_collided > 0 && !b->collided) ||
        (b->total_edges > 0 && !b->edge_index))
        return GNNMP_ERR_NULL;
    if (loop < 0) return GNNMP_ERR_ARG;
    if (b->max_samples > 2048) return GNNMP_ERR_DIMS;          // kNN keeps <= 32 samples per lane
    SmCarve c;
    if (!sm_carve(h, b, c)) return GNNMP_ERR_ARG;
    if (ws_bytes < c.total || (reinterpret_cast<uintptr_t>(ws) & 255)) return GNNMP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int C = h->dims.config_size, D = h->dims.embed_size;
    SmParams p;
    p.B = b->n_problems; p.C = C; p.total_path = b->total_path; p.total_edges = b->total_edges;
    p.scale = h->dims.scale;
    p.path = b->path; p.free_pts = b->free_pts; p.collided = b->collided;
    p.edge_index = reinterpret_cast<const long long*>(b->edge_index);
    p.path_ptr = b->path_ptr; p.free_ptr = b->free_ptr; p.coll_ptr = b->coll_ptr; p.edge_ptr = b->edge_ptr;
    p.w = h->w_dev; p.L = h->L;
    p.cur = at<float>(ws, c.cur); p.cur_next = p.cur;
    p.knn = at<int>(ws, c.knn);
    p.e_src = at<int>(ws, c.e_src); p.e_dst = at<int>(ws, c.e_dst); p.e_count = at<int>(ws, c.e_count);
    p.stat = status_out ? reinterpret_cast<int*>(status_out) : at<int>(ws, c.stat);      // one word per problem (gnnmp_smoother_forward_ex)
    p.seg_beg = at<int>(ws, c.seg_beg); p.seg_cnt = at<int>(ws, c.seg_cnt);
    p.etile_prob = at<int>(ws, c.etile); p.ptile_prob = at<int>(ws, c.ptile);
    p.msg = at<float>(ws, c.msg);
    p.tgt = at<float>(ws, c.tgt); p.tgt_flag = at<int>(ws, c.tflag);
    p.tile_cnt = at<int>(ws, c.tcnt); p.elist = at<int>(ws, c.elist); p.plist = at<int>(ws, c.plist); p.parity = 0;
    { static const int no_tgt = getenv("GNNMP_SM_NO_TARGET_ROLE") ? atoi(getenv("GNNMP_SM_NO_TARGET_ROLE")) : 0; if (no_tgt) p.tgt_flag = nullptr; }      // experiments
    p.cand_cap = b->max_edges + kSmK * b->max_path;
    if (p.cand_cap < 1) p.cand_cap = 1;
    p.samp_cap = b->max_samples > 0 ? b->max_samples : 0; p.path_cap = b->max_path > 0 ? b->max_path : 0;
    if ((size_t)2 * p.cand_cap * sizeof(int) > 60000) return GNNMP_ERR_DIMS;
    p.n_etiles = c.ecap / 32; p.n_ptiles = c.pcap / 32;
    p.one_free = b->total_free; p.one_coll = b->total_collided;
    if (loop == 0) {                                         // nothing moves: out = (path / scale) * scale
        HIP_TRY(hipMemsetAsync(p.stat, 0, sizeof(int) * b->n_problems, st));
        HIP_TRY(launch_sm_init(b->total_path * C, p.scale, b->path, p.cur, st));
        HIP_TRY(launch_sm_final(b->total_path * C, p.scale, p.cur, out_path, st));
        return GNNMP_OK;
    }
    // three launches per iteration (graph stage = kNN + edge list, messages, path update; four when a problem's buffers exceed
    // the LDS share of the one-launch graph stage): the scaled working copy is written by the
    // first kNN launch, the tile maps by the edge-list kernel, the result by the last path-update launch
    if (gnnmp::sm_wants_tile_lists(D, h->dims.mlp_dtype, p.n_etiles))
        HIP_TRY(hipMemsetAsync(p.tile_cnt, 0, sizeof(int) * 4, st));  // the tile-list counters (then re-armed by the graph stage itself)
    else
        p.tile_cnt = nullptr;                                          // no lists: the graph stage skips the bookkeeping
    for (int it = 0; it < loop; ++it) {
        p.init_from_path = it == 0 ? 1 : 0;
        p.out = it == loop - 1 ? out_path : nullptr;
        p.parity = it & 1;
        HIP_TRY(launch_sm_iter(D, h->dims.mlp_dtype, p, st));
    }
    return GNNMP_OK;
}


extern "C" int gnnmp_smoother_status_region(const gnnmp_smoother* h, const gnnmp_smooth_batch* shape, size_t* offset, size_t* bytes) {
    if (!h || !shape || !offset || !bytes) return GNNMP_ERR_NULL;
    SmCarve c;
    if (!sm_carve(h, shape, c)) return GNNMP_ERR_ARG;
    *offset = c.stat;
    *bytes = sizeof(int) * (size_t)shape->n_problems;
    return GNNMP_OK;
}

extern "C" int gnnmp_smoother_status_decode(const int32_t* words_host, int n_problems, int32_t* first_problem) {
    if (!words_host || n_problems < 0) return GNNMP_ERR_NULL;
    for (int b = 0; b < n_problems; ++b)
        if (words_host[b] != 0) { if (first_problem) *first_problem = b; return GNNMP_ERR_CAPS; }
    if (first_problem) *first_problem = -1;
    return GNNMP_OK;
}

extern "C" int gnnmp_smoother_status(const gnnmp_smoother* h, const gnnmp_smooth_batch* shape, const void* ws, size_t ws_bytes,
                                     void* hip_stream, int32_t* first_problem) {
    if (!h || !shape || !ws) return GNNMP_ERR_NULL;
    SmCarve c;
    if (!sm_carve(h, shape, c)) return GNNMP_ERR_ARG;
    if (ws_bytes < c.total) return GNNMP_ERR_WORKSPACE;
    std::vector<int32_t> host((size_t)shape->n_problems);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    HIP_TRY(hipMemcpyAsync(host.data(), static_cast<const char*>(ws) + c.stat, host.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return gnnmp_smoother_status_decode(host.data(), shape->n_problems, first_problem);
}


// =============================================================================================
// device-side graph construction (create_data counterpart, eval_gnn.py:159-164)
// =============================================================================================
namespace {
struct GbCarve { size_t nb_all, nb_free, zero_beg, cnt, cur, large_cnt, zero_end, off, ucnt, uoff, gtotal, bucket, total; };
bool gb_carve(const gnnmp_graph_batch* b, GbCarve& c) {
    if (b->n_graphs < 1 || b->total_nodes < 0 || b->k1_max < 1 || b->config_size < 1) return false;
    if ((long long)4 * b->k1_max * b->total_nodes > 0x3fffffffLL) return false;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    const size_t n = (size_t)(b->total_nodes > 0 ? b->total_nodes : 1);
    c.nb_all = take(sizeof(int) * n * b->k1_max);
    c.nb_free = take(sizeof(int) * n * b->k1_max);
    c.zero_beg = o;
    c.cnt = take(sizeof(int) * n);
    c.cur = take(sizeof(int) * n);
    c.large_cnt = take(sizeof(int));
    c.zero_end = o;
    c.off = take(sizeof(int) * n);
    c.ucnt = take(sizeof(int) * n);
    c.uoff = take(sizeof(int) * n);
    c.gtotal = take(sizeof(int) * b->n_graphs);
    c.bucket = take(sizeof(int) * n * 4 * b->k1_max);
    c.total = o;
    return true;
}
}  // namespace

extern "C" int gnnmp_graph_workspace_bytes(const gnnmp_graph_batch* b, size_t* bytes) {
    if (!b || !bytes) return GNNMP_ERR_NULL;
    GbCarve c;
    if (!gb_carve(b, c)) return GNNMP_ERR_ARG;
    *bytes = c.total;
    return GNNMP_OK;
}

extern "C" int gnnmp_graph_build(const gnnmp_graph_batch* b, int64_t* edge_index_out, int64_t out_cap,
                                 int32_t* edge_ptr_out, void* ws, size_t ws_bytes, void* hip_stream) {
    if (!b || !edge_index_out || !edge_ptr_out || !ws) return GNNMP_ERR_NULL;
    if (!b->v || !b->node_ptr || !b->n_free || !b->k1) return GNNMP_ERR_NULL;
    GbCarve c;
    if (!gb_carve(b, c)) return GNNMP_ERR_ARG;
    if (out_cap < 1) return GNNMP_ERR_ARG;           // columns beyond out_cap are dropped; edge_ptr_out holds the true counts
    if (ws_bytes < c.total || (reinterpret_cast<uintptr_t>(ws) & 255)) return GNNMP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    HIP_TRY(hipMemsetAsync(at<char>(ws, c.zero_beg), 0, c.zero_end - c.zero_beg, st));
    GbParams p;
    p.G = b->n_graphs; p.C = b->config_size; p.total_nodes = b->total_nodes; p.kmax = b->k1_max;
    p.v = b->v; p.node_ptr = b->node_ptr; p.n_free = b->n_free; p.k1 = b->k1;
    p.nb_all = at<int>(ws, c.nb_all); p.nb_free = at<int>(ws, c.nb_free);
    p.cnt = at<int>(ws, c.cnt); p.cur = at<int>(ws, c.cur); p.off = at<int>(ws, c.off);
    p.ucnt = at<int>(ws, c.ucnt); p.uoff = at<int>(ws, c.uoff); p.gtotal = at<int>(ws, c.gtotal);
    p.bucket = at<int>(ws, c.bucket);
    p.large_cnt = at<int>(ws, c.large_cnt); p.large = p.uoff;      // (uoff is first written after the kNN launches)
    p.edge_ptr = edge_ptr_out;
    p.edge_index = reinterpret_cast<long long*>(edge_index_out);
    p.out_cap = out_cap;
    if (b->total_nodes > 0) HIP_TRY(launch_graph_build(p, st));
    return GNNMP_OK;
}


// =============================================================================================
// device-side explore stage for 2-D mazes (eval_gnn.py:198-233 + environment/maze_env.py:270-326)
// =============================================================================================
namespace {
struct MzCarve { size_t in_ptr, cnt, in_rec, pos, prev, rb_val, rb_src, rb_eid, total; };
bool mz_carve(const gnnmp_maze_batch* b, MzCarve& c) {
    if (b->n_problems < 1 || b->total_nodes < 0 || b->total_edges < 0 || b->width < 1) return false;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    const size_t n = (size_t)b->total_nodes + b->n_problems + 1, e = (size_t)(b->total_edges > 0 ? b->total_edges : 1);
    c.in_ptr = take(sizeof(int) * n);
    c.cnt = take(sizeof(int) * n);
    c.in_rec = take(sizeof(int) * 2 * e);
    c.pos = take(sizeof(int) * n);
    c.prev = take(sizeof(int) * n);
    c.rb_val = take(sizeof(float) * n);
    c.rb_src = take(sizeof(int) * n);
    c.rb_eid = take(sizeof(int) * n);
    c.total = o;
    return true;
}
}  // namespace

extern "C" int gnnmp_maze_explore_workspace_bytes(const gnnmp_maze_batch* b, size_t* bytes) {
    if (!b || !bytes) return GNNMP_ERR_NULL;
    MzCarve c;
    if (!mz_carve(b, c)) return GNNMP_ERR_ARG;
    *bytes = c.total;
    return GNNMP_OK;
}

extern "C" int gnnmp_maze_explore_ex(const gnnmp_maze_batch* b, int32_t dim, const gnnmp_maze_resume* resume,
                                     int32_t* success, int32_t* n_explored, int32_t* explored, int32_t* n_pairs,
                                     int32_t* explored_edges, int32_t* path_len, int32_t* path, int64_t* checks,
                                     int32_t* prev_out, void* ws, size_t ws_bytes, void* hip_stream) {
    if (!b || !success || !n_explored || !explored || !n_pairs || !explored_edges || !path_len || !path || !checks || !ws)
        return GNNMP_ERR_NULL;
    if (!b->v || !b->node_ptr || !b->edge_ptr || !b->n_free || !b->maps || !b->goal_states) return GNNMP_ERR_NULL;
    if (b->total_edges > 0 && (!b->edge_index || !b->scores)) return GNNMP_ERR_NULL;
    if (dim != 2 && dim != 3) return GNNMP_ERR_DIMS;
    const bool res = resume && resume->n_explored;
    if (res && (!resume->explored || !resume->prev || !resume->n_pairs || !resume->pairs || !resume->pair_ptr))
        return GNNMP_ERR_NULL;
    MzCarve c;
    if (!mz_carve(b, c)) return GNNMP_ERR_ARG;
    if (ws_bytes < c.total || (reinterpret_cast<uintptr_t>(ws) & 255)) return GNNMP_ERR_WORKSPACE;
    MazeParams p;
    p.B = b->n_problems; p.total_edges = b->total_edges; p.w = b->width;
    p.v = b->v; p.node_ptr = b->node_ptr; p.edge_ptr = b->edge_ptr; p.n_free = b->n_free;
    p.edge_index = reinterpret_cast<const long long*>(b->edge_index); p.scores = b->scores;
    p.maps = b->maps; p.goal_states = b->goal_states;
    p.in_ptr = at<int>(ws, c.in_ptr); p.cnt = at<int>(ws, c.cnt); p.in_rec = at<int2>(ws, c.in_rec);
    p.pos = at<int>(ws, c.pos); p.prev = at<int>(ws, c.prev);
    p.rb_val = at<float>(ws, c.rb_val); p.rb_src = at<int>(ws, c.rb_src); p.rb_eid = at<int>(ws, c.rb_eid);
    p.success = success; p.n_explored = n_explored; p.explored = explored; p.n_pairs = n_pairs;
    p.explored_edges = explored_edges; p.path_len = path_len; p.path = path;
    p.checks = reinterpret_cast<long long*>(checks);
    p.dim = dim;
    p.n_explored0 = res ? resume->n_explored : nullptr; p.explored0 = res ? resume->explored : nullptr;
    p.prev0 = res ? resume->prev : nullptr; p.n_pairs0 = res ? resume->n_pairs : nullptr;
    p.pairs0 = res ? resume->pairs : nullptr; p.pair_ptr0 = res ? resume->pair_ptr : nullptr;
    p.prev_out = prev_out;
    HIP_TRY(launch_maze_explore(p, static_cast<hipStream_t>(hip_stream)));
    return GNNMP_OK;
}

extern "C" int gnnmp_maze_explore(const gnnmp_maze_batch* b, int32_t* success, int32_t* n_explored, int32_t* explored,
                                  int32_t* n_pairs, int32_t* explored_edges, int32_t* path_len, int32_t* path,
                                  int64_t* checks, void* ws, size_t ws_bytes, void* hip_stream) {
    return gnnmp_maze_explore_ex(b, 2, nullptr, success, n_explored, explored, n_pairs, explored_edges, path_len, path, checks,
                                 nullptr, ws, ws_bytes, hip_stream);
}

extern "C" int gnnmp_maze_steer(int32_t n_problems, int32_t total_path, int32_t width, const double* maps,
                                const int32_t* path_ptr, const float* old_path, const float* new_path, float* out_path,
                                float* tmp, int64_t* checks, void* hip_stream) {
    if (!maps || !path_ptr || !checks) return GNNMP_ERR_NULL;
    if (n_problems < 1 || total_path < 0 || width < 1) return GNNMP_ERR_ARG;
    if (total_path > 0 && (!old_path || !new_path || !out_path || !tmp)) return GNNMP_ERR_NULL;
    if (out_path == old_path || out_path == new_path) return GNNMP_ERR_ARG;
    MazeSteerParams p;
    p.B = n_problems; p.w = width; p.maps = maps; p.path_ptr = path_ptr;
    p.old_path = old_path; p.new_path = new_path; p.out_path = out_path; p.tmp = tmp;
    p.checks = reinterpret_cast<long long*>(checks);
    HIP_TRY(launch_maze_steer(p, static_cast<hipStream_t>(hip_stream)));
    return GNNMP_OK;
}

extern "C" int gnnmp_maze_sample(const gnnmp_maze_sample_batch* b, int64_t* cursor, float* v_out, int32_t* node_ptr_out,
                                 int32_t* used_out, int32_t* ok_out, void* hip_stream) {
    if (!b || !cursor || !v_out || !node_ptr_out || !used_out || !ok_out) return GNNMP_ERR_NULL;
    if (!b->attempts || !b->maps || !b->init_states || !b->goal_states) return GNNMP_ERR_NULL;
    if (b->n_problems < 1 || b->width < 1 || b->n_free < 1 || b->n_attempts < 0) return GNNMP_ERR_ARG;
    MazeSampleParams p;
    p.B = b->n_problems; p.w = b->width; p.n = b->n_free;
    p.attempts = b->attempts; p.M = b->n_attempts;
    p.maps = b->maps; p.init_states = b->init_states; p.goal_states = b->goal_states;
    p.v = v_out; p.node_ptr = node_ptr_out; p.used = used_out;
    p.cursor = reinterpret_cast<long long*>(cursor); p.ok = ok_out;
    HIP_TRY(launch_maze_sample(p, static_cast<hipStream_t>(hip_stream)));
    return GNNMP_OK;
}

// =============================================================================================
// training path of the explorer (train_explorer.py:156-186)
// =============================================================================================
namespace {

struct TrainCarve {
    size_t inf_bytes;                  // the inference workspace comes first (CSR, goal node, padded pointers stay valid)
    // offsets in floats from the start of the train region
    size_t NF, EF, NCin, NCh, NC, ECin, ECh, EC, H0, it0, it_stride, Xin, X, Zh, A, arg, H, DinCat, Dn, Pin, P1, P2, sc;
    size_t T5a, T5b, Te1, Te2, T3, dX, dH, dA, dNC, dH0, dXin, dEC, cat2, dcat2, dDn;
    size_t obeg, ocnt, ocur, oslot;    // edges grouped by source (ints), built by the forward, read by the backward
    size_t dwp;                        // per-block partial sums of the weight gradients (two-stage, deterministic)
    size_t total_floats;
};

bool train_carve(const gnnmp_explorer* h, const gnnmp_batch* b, int loop, const Carve& c, TrainCarve& t) {
    const size_t Np = c.Npad, Ep = c.Epad, d = h->dims.embed_size, C = h->dims.config_size;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
    t.inf_bytes = (c.total + 255) & ~(size_t)255;
    t.NF = take(Np * d); t.EF = take(Ep * d);
    t.NCin = take(Np * 4 * C); t.NCh = take(Np * d); t.NC = take(Np * d);
    t.ECin = take(Ep * 2 * C); t.ECh = take(Ep * d); t.EC = take(Ep * d);
    t.H0 = take(Np * d);
    t.it0 = o;
    t.Xin = take(Np * 4 * d) - t.it0; t.X = take(Np * d) - t.it0; t.Zh = take(Ep * d) - t.it0; t.A = take(Np * d) - t.it0;
    t.arg = take(Np * d) - t.it0; t.H = take(Np * d) - t.it0;
    t.it_stride = o - t.it0;
    o = t.it0 + t.it_stride * (size_t)loop;
    t.DinCat = take(Np * 2 * d); t.Dn = take(Np * d);
    t.Pin = take(Ep * 3 * d); t.P1 = take(Ep * d); t.P2 = take(Ep * d); t.sc = take(Ep);
    t.T5a = take(Ep * 5 * d); t.T5b = take(Ep * 5 * d); t.Te1 = take(Ep * d); t.Te2 = take(Ep * d); t.T3 = take(Ep * 3 * d);
    t.dX = take(Np * d); t.dH = take(Np * d); t.dA = take(Np * d); t.dNC = take(Np * d); t.dH0 = take(Np * d);
    t.dXin = take(Np * 4 * d); t.dEC = take(Ep * d); t.cat2 = take(Np * 2 * d); t.dcat2 = take(Np * 2 * d); t.dDn = take(Np * d);
    t.obeg = take(Np); t.ocnt = take(Np); t.ocur = take(Np); t.oslot = take(Ep);
    {
        size_t m = t_linear_dw_scratch_floats((int)Ep, (int)(5 * d), (int)d);          // the widest layer over edge rows ...
        const size_t n4 = t_linear_dw_scratch_floats((int)Np, (int)(4 * d), (int)d);  // ... and over node rows
        t.dwp = take(m > n4 ? m : n4);
    }
    t.total_floats = o;
    (void)b;
    return true;
}

struct WRef { const float* w; const float* b; float* gw; float* gb; int out, in; };

// weight / bias of a Linear named `name` inside the raw blob, and the same slots inside the gradient blob
WRef wref(const gnnmp_explorer* h, float* grad, const std::string& name, bool bias = true) {
    WRef r{nullptr, nullptr, nullptr, nullptr, 0, 0};
    for (size_t i = 0; i < h->man.size(); ++i) {
        if (h->man[i].name == name + ".weight") {
            r.w = h->w_raw_dev + h->man_off[i]; r.gw = grad ? grad + h->man_off[i] : nullptr;
            r.out = h->man[i].rows; r.in = h->man[i].cols;
        }
        if (bias && h->man[i].name == name + ".bias") { r.b = h->w_raw_dev + h->man_off[i]; r.gb = grad ? grad + h->man_off[i] : nullptr; }
    }
    return r;
}

TrainGeom train_geom(const gnnmp_explorer* h, const gnnmp_batch* b, const Carve& c, const TrainCarve& t, void* ws) {
    TrainGeom q;
    q.G = c.G; q.C = h->dims.config_size; q.Npad = c.Npad; q.Epad = c.Epad;
    q.v = b->v; q.goal = b->goal; q.node_ptr = b->node_ptr;
    q.node_ptr_pad = at<int>(ws, c.node_ptr_pad); q.ntile_graph = at<int>(ws, c.ntile_graph);
    q.goal_node = at<int>(ws, c.goal_node); q.row_beg = at<int>(ws, c.row_beg); q.deg = at<int>(ws, c.deg);
    q.csr = at<int4>(ws, c.csr);
    int* TI = reinterpret_cast<int*>(static_cast<char*>(ws) + t.inf_bytes);
    q.out_beg = TI + t.obeg; q.out_cnt = TI + t.ocnt; q.out_cur = TI + t.ocur; q.out_slot = TI + t.oslot;
    return q;
}

}  // namespace

extern "C" int64_t gnnmp_explorer_grad_floats(const gnnmp_explorer* h) { return h ? h->n_raw : GNNMP_ERR_NULL; }

extern "C" int gnnmp_explorer_train_workspace_bytes(const gnnmp_explorer* h, const gnnmp_batch* shape, int loop, size_t* bytes) {
    if (!h || !shape || !bytes) return GNNMP_ERR_NULL;
    if (loop < 1) return GNNMP_ERR_ARG;
    if (h->dims.mlp_dtype != GNNMP_F32) return GNNMP_ERR_DIMS;          // training runs in fp32
    Carve c;
    if (!carve(h, shape, c)) return GNNMP_ERR_ARG;
    TrainCarve t;
    train_carve(h, shape, loop, c, t);
    *bytes = t.inf_bytes + t.total_floats * sizeof(float);
    return GNNMP_OK;
}

extern "C" int gnnmp_explorer_train_forward(const gnnmp_explorer* h, const gnnmp_batch* b, int loop, int use_obstacles,
                                            float* edge_scores, void* ws, size_t ws_bytes, void* hip_stream) {
    if (!h || !b || !ws || (b->total_edges > 0 && !edge_scores)) return GNNMP_ERR_NULL;
    if (!b->node_ptr || !b->edge_ptr || !b->obs_ptr) return GNNMP_ERR_NULL;     // the training path takes explicit prefix arrays
    if (loop < 1) return GNNMP_ERR_ARG;
    if (h->dims.mlp_dtype != GNNMP_F32) return GNNMP_ERR_DIMS;
    Carve c;
    if (!carve(h, b, c)) return GNNMP_ERR_ARG;
    TrainCarve t;
    train_carve(h, b, loop, c, t);
    if (ws_bytes < t.inf_bytes + t.total_floats * sizeof(float) || (reinterpret_cast<uintptr_t>(ws) & 255)) return GNNMP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    float* T = reinterpret_cast<float*>(static_cast<char*>(ws) + t.inf_bytes);
    const int d = h->dims.embed_size, C = h->dims.config_size, Np = c.Npad, Ep = c.Epad;
    // rows of padding tiles no kernel writes must hold finite numbers: the weight gradients sum 0 * activation over them
    HIP_TRY(hipMemsetAsync(T, 0, t.total_floats * sizeof(float), st));
    // frozen inputs (model.py:141,142,146 detach them): node_free_code / edge_free_code after the attention stacks
    const int rc = forward_impl(h, b, loop, use_obstacles, nullptr, nullptr, ws, t.inf_bytes, hip_stream, T + t.NF, T + t.EF, true);
    if (rc != GNNMP_OK) return rc;
    const TrainGeom q = train_geom(h, b, c, t, ws);
    const WRef nc0 = wref(h, nullptr, "node_code.0"), nc2 = wref(h, nullptr, "node_code.2"), ec0 = wref(h, nullptr, "edge_code.0"),
               ec2 = wref(h, nullptr, "edge_code.2"), enc = wref(h, nullptr, "encoder"), l00 = wref(h, nullptr, "process.lin_0.0"),
               l02 = wref(h, nullptr, "process.lin_0.2"), l1 = wref(h, nullptr, "process.lin_1"), dec = wref(h, nullptr, "decoder"),
               p0 = wref(h, nullptr, "policy.0"), p2 = wref(h, nullptr, "policy.2"), p4 = wref(h, nullptr, "policy.4", false);
    const float* ge = nullptr;
    for (size_t i = 0; i < h->man.size(); ++i) if (h->man[i].name == "goal_encoder") ge = h->w_raw_dev + h->man_off[i];
    HIP_TRY(t_out_csr(q, st));                                   // edges by source: the backward's gather adjoints walk it in order
    HIP_TRY(t_node_in(q, T + t.NCin, st));
    HIP_TRY(t_linear(Np, 4 * C, d, T + t.NCin, nc0.w, nc0.b, T + t.NCh, true, st));
    HIP_TRY(t_linear(Np, d, d, T + t.NCh, nc2.w, nc2.b, T + t.NC, false, st));
    HIP_TRY(t_edge_in(q, T + t.ECin, st));
    HIP_TRY(t_linear(Ep, 2 * C, d, T + t.ECin, ec0.w, ec0.b, T + t.ECh, true, st));
    HIP_TRY(t_linear(Ep, d, d, T + t.ECh, ec2.w, ec2.b, T + t.EC, false, st));
    HIP_TRY(t_h0(q, d, ge, T + t.H0, st));
    const float* Hprev = T + t.H0;
    for (int it = 0; it < loop; ++it) {
        float* I = T + t.it0 + t.it_stride * (size_t)it;
        HIP_TRY(t_concat(Np, d, 4, T + t.NC, T + t.NF, T + t.H0, Hprev, I + t.Xin, st));                 // model.py:141
        HIP_TRY(t_linear(Np, 4 * d, d, I + t.Xin, enc.w, enc.b, I + t.X, false, st));
        HIP_TRY(t_msg_in(q, d, I + t.X, T + t.EF, T + t.EC, T + t.T5a, st));                             // model.py:38-39
        HIP_TRY(t_linear(Ep, 5 * d, d, T + t.T5a, l00.w, l00.b, I + t.Zh, true, st));
        HIP_TRY(t_linear(Ep, d, d, I + t.Zh, l02.w, l02.b, T + t.Te1, false, st));
        HIP_TRY(t_segment_max(q, d, T + t.Te1, I + t.A, reinterpret_cast<int*>(I + t.arg), st));         // model.py:33
        HIP_TRY(t_concat(Np, d, 2, I + t.X, I + t.A, nullptr, nullptr, T + t.cat2, st));
        HIP_TRY(t_linear(Np, 2 * d, d, T + t.cat2, l1.w, l1.b, I + t.H, false, st));                     // model.py:36
        Hprev = I + t.H;
    }
    HIP_TRY(t_concat(Np, d, 2, T + t.NC, Hprev, nullptr, nullptr, T + t.DinCat, st));                    // model.py:143
    HIP_TRY(t_linear(Np, 2 * d, d, T + t.DinCat, dec.w, dec.b, T + t.Dn, false, st));
    HIP_TRY(t_pol_in(q, d, T + t.Dn, T + t.EF, T + t.Pin, st));                                          // model.py:145
    HIP_TRY(t_linear(Ep, 3 * d, d, T + t.Pin, p0.w, p0.b, T + t.P1, true, st));
    HIP_TRY(t_linear(Ep, d, d, T + t.P1, p2.w, p2.b, T + t.P2, true, st));
    HIP_TRY(t_linear(Ep, d, 1, T + t.P2, p4.w, nullptr, T + t.sc, false, st));
    if (b->total_edges > 0) HIP_TRY(t_scores_out(q, T + t.sc, edge_scores, st));
    return GNNMP_OK;
}

extern "C" int gnnmp_explorer_train_backward(const gnnmp_explorer* h, const gnnmp_batch* b, int loop, const float* d_edge_scores,
                                             float* grad, void* ws, size_t ws_bytes, void* hip_stream) {
    if (!h || !b || !ws || !grad || (b->total_edges > 0 && !d_edge_scores)) return GNNMP_ERR_NULL;
    if (!b->node_ptr || !b->edge_ptr || !b->obs_ptr) return GNNMP_ERR_NULL;
    if (loop < 1) return GNNMP_ERR_ARG;
    if (h->dims.mlp_dtype != GNNMP_F32) return GNNMP_ERR_DIMS;           // same argument checks as train_forward
    Carve c;
    if (!carve(h, b, c)) return GNNMP_ERR_ARG;
    TrainCarve t;
    train_carve(h, b, loop, c, t);
    if (ws_bytes < t.inf_bytes + t.total_floats * sizeof(float) || (reinterpret_cast<uintptr_t>(ws) & 255)) return GNNMP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    float* T = reinterpret_cast<float*>(static_cast<char*>(ws) + t.inf_bytes);
    const int d = h->dims.embed_size, C = h->dims.config_size, Np = c.Npad, Ep = c.Epad;
    const TrainGeom q = train_geom(h, b, c, t, ws);
    HIP_TRY(hipMemsetAsync(grad, 0, (size_t)h->n_raw * sizeof(float), st));
    const WRef nc0 = wref(h, grad, "node_code.0"), nc2 = wref(h, grad, "node_code.2"), ec0 = wref(h, grad, "edge_code.0"),
               ec2 = wref(h, grad, "edge_code.2"), enc = wref(h, grad, "encoder"), l00 = wref(h, grad, "process.lin_0.0"),
               l02 = wref(h, grad, "process.lin_0.2"), l1 = wref(h, grad, "process.lin_1"), dec = wref(h, grad, "decoder"),
               p0 = wref(h, grad, "policy.0"), p2 = wref(h, grad, "policy.2"), p4 = wref(h, grad, "policy.4", false);
    float* g_ge = nullptr;
    for (size_t i = 0; i < h->man.size(); ++i) if (h->man[i].name == "goal_encoder") g_ge = grad + h->man_off[i];
    // ---- policy head (model.py:145-146)
    HIP_TRY(t_scores_in(q, d_edge_scores, T + t.Te1, st));                          // [Ep, 1]
    HIP_TRY(t_linear_dw(Ep, d, 1, T + t.Te1, T + t.P2, p4.gw, nullptr, T + t.dwp, st));
    HIP_TRY(t_linear_dx(Ep, d, 1, T + t.Te1, p4.w, T + t.Te2, false, st));          // dP2
    HIP_TRY(t_relu_bwd((size_t)Ep * d, T + t.P2, T + t.Te2, st));
    HIP_TRY(t_linear_dw(Ep, d, d, T + t.Te2, T + t.P1, p2.gw, p2.gb, T + t.dwp, st));
    HIP_TRY(t_linear_dx(Ep, d, d, T + t.Te2, p2.w, T + t.Te1, false, st));          // dP1
    HIP_TRY(t_relu_bwd((size_t)Ep * d, T + t.P1, T + t.Te1, st));
    HIP_TRY(t_linear_dw(Ep, 3 * d, d, T + t.Te1, T + t.Pin, p0.gw, p0.gb, T + t.dwp, st));
    HIP_TRY(t_linear_dx(Ep, 3 * d, d, T + t.Te1, p0.w, T + t.T3, false, st));       // dPin
    HIP_TRY(t_fill((size_t)Np * d, T + t.dDn, 0.f, st));
    HIP_TRY(t_pol_in_bwd(q, d, T + t.T3, T + t.dDn, st));
    // ---- decoder (model.py:143)
    HIP_TRY(t_linear_dw(Np, 2 * d, d, T + t.dDn, T + t.DinCat, dec.gw, dec.gb, T + t.dwp, st));
    HIP_TRY(t_linear_dx(Np, 2 * d, d, T + t.dDn, dec.w, T + t.dcat2, false, st));
    HIP_TRY(t_split(Np, d, 2, 0, T + t.dcat2, T + t.dNC, false, st));
    HIP_TRY(t_split(Np, d, 2, 1, T + t.dcat2, T + t.dH, false, st));
    HIP_TRY(t_fill((size_t)Np * d, T + t.dH0, 0.f, st));
    HIP_TRY(t_fill((size_t)Ep * d, T + t.dEC, 0.f, st));
    // ---- the loop, backwards (model.py:139-142)
    for (int it = loop - 1; it >= 0; --it) {
        float* I = T + t.it0 + t.it_stride * (size_t)it;
        HIP_TRY(t_concat(Np, d, 2, I + t.X, I + t.A, nullptr, nullptr, T + t.cat2, st));
        HIP_TRY(t_linear_dw(Np, 2 * d, d, T + t.dH, T + t.cat2, l1.gw, l1.gb, T + t.dwp, st));
        HIP_TRY(t_linear_dx(Np, 2 * d, d, T + t.dH, l1.w, T + t.dcat2, false, st));
        HIP_TRY(t_split(Np, d, 2, 0, T + t.dcat2, T + t.dX, false, st));
        HIP_TRY(t_split(Np, d, 2, 1, T + t.dcat2, T + t.dA, false, st));
        HIP_TRY(t_fill((size_t)Ep * d, T + t.Te1, 0.f, st));                        // dM
        HIP_TRY(t_segment_max_bwd(Np, d, T + t.dA, reinterpret_cast<const int*>(I + t.arg), T + t.Te1, st));
        HIP_TRY(t_linear_dw(Ep, d, d, T + t.Te1, I + t.Zh, l02.gw, l02.gb, T + t.dwp, st));
        HIP_TRY(t_linear_dx(Ep, d, d, T + t.Te1, l02.w, T + t.Te2, false, st));      // dZh
        HIP_TRY(t_relu_bwd((size_t)Ep * d, I + t.Zh, T + t.Te2, st));
        HIP_TRY(t_msg_in(q, d, I + t.X, T + t.EF, T + t.EC, T + t.T5a, st));         // Zin recomputed
        HIP_TRY(t_linear_dw(Ep, 5 * d, d, T + t.Te2, T + t.T5a, l00.gw, l00.gb, T + t.dwp, st));
        HIP_TRY(t_linear_dx(Ep, 5 * d, d, T + t.Te2, l00.w, T + t.T5b, false, st));   // dZin
        HIP_TRY(t_msg_in_bwd(q, d, T + t.T5b, T + t.dX, T + t.dEC, st));
        HIP_TRY(t_linear_dw(Np, 4 * d, d, T + t.dX, I + t.Xin, enc.gw, enc.gb, T + t.dwp, st));
        HIP_TRY(t_linear_dx(Np, 4 * d, d, T + t.dX, enc.w, T + t.dXin, false, st));
        HIP_TRY(t_split(Np, d, 4, 0, T + t.dXin, T + t.dNC, true, st));              // node_code
        HIP_TRY(t_split(Np, d, 4, 2, T + t.dXin, T + t.dH0, true, st));              // h_0   (part 1 = node_free_code: detached)
        if (it > 0) HIP_TRY(t_split(Np, d, 4, 3, T + t.dXin, T + t.dH, false, st));  // h_{i-1}
        else HIP_TRY(t_split(Np, d, 4, 3, T + t.dXin, T + t.dH0, true, st));         // h_i of the first iteration IS h_0
    }
    HIP_TRY(t_h0_bwd(q, d, T + t.dH0, g_ge, st));
    // ---- edge_code, node_code encoders (model.py:119-120)
    HIP_TRY(t_linear_dw(Ep, d, d, T + t.dEC, T + t.ECh, ec2.gw, ec2.gb, T + t.dwp, st));
    HIP_TRY(t_linear_dx(Ep, d, d, T + t.dEC, ec2.w, T + t.Te1, false, st));
    HIP_TRY(t_relu_bwd((size_t)Ep * d, T + t.ECh, T + t.Te1, st));
    HIP_TRY(t_linear_dw(Ep, 2 * C, d, T + t.Te1, T + t.ECin, ec0.gw, ec0.gb, T + t.dwp, st));
    HIP_TRY(t_linear_dw(Np, d, d, T + t.dNC, T + t.NCh, nc2.gw, nc2.gb, T + t.dwp, st));
    HIP_TRY(t_linear_dx(Np, d, d, T + t.dNC, nc2.w, T + t.dX, false, st));
    HIP_TRY(t_relu_bwd((size_t)Np * d, T + t.NCh, T + t.dX, st));
    HIP_TRY(t_linear_dw(Np, 4 * C, d, T + t.dX, T + t.NCin, nc0.gw, nc0.gb, T + t.dwp, st));
    return GNNMP_OK;
}

// =============================================================================================
// training path of the smoother (train_smoother.py:33-61; model_smoother.py:104-142 under model.train())
// =============================================================================================
namespace {

struct SmTrainCarve {
    size_t inf_bytes;
    int Nn, K0, ecap;
    // floats from the start of the train region
    size_t states, it0, it_stride, Xin, X0, stats, X1, X, esrc, edst, ne, Zh, S, L1h, Hh;
    size_t Zin, M, dZh, dZin, dX, dX1, dX0, dXin, dS, dL1h, dHh, prop, dprop, dpa, dpb, tmpP, dwp;
    size_t total_floats;
};

bool sm_train_carve(const gnnmp_smoother* h, const gnnmp_smooth_batch* b, int loop, const SmCarve& c, SmTrainCarve& t) {
    const size_t d = h->dims.embed_size, C = h->dims.config_size, P = b->total_path;
    t.Nn = b->total_path + b->total_free + b->total_collided;
    t.K0 = (int)C + 3;
    t.ecap = c.ecap;
    const size_t Nn = t.Nn, Ec = t.ecap, K0 = t.K0;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
    t.inf_bytes = (c.total + 255) & ~(size_t)255;
    t.states = take((size_t)(loop + 1) * P * C);
    t.it0 = o;
    t.Xin = take(Nn * K0) - t.it0; t.X0 = take(Nn * d) - t.it0; t.stats = take(3 * d) - t.it0; t.X1 = take(Nn * d) - t.it0;
    t.X = take(Nn * d) - t.it0; t.esrc = take(Ec) - t.it0; t.edst = take(Ec) - t.it0; t.ne = take(64) - t.it0;
    t.Zh = take(Ec * d) - t.it0; t.S = take(P * d) - t.it0; t.L1h = take(P * d) - t.it0; t.Hh = take(P * d) - t.it0;
    t.it_stride = o - t.it0;
    o = t.it0 + t.it_stride * (size_t)(loop > 0 ? loop : 1);
    t.Zin = take(Ec * 3 * d); t.M = take(Ec * d); t.dZh = take(Ec * d); t.dZin = take(Ec * 3 * d);
    t.dX = take(Nn * d); t.dX1 = take(Nn * d); t.dX0 = take(Nn * d); t.dXin = take(Nn * K0);
    t.dS = take(P * d); t.dL1h = take(P * d); t.dHh = take(P * d); t.prop = take(P * C); t.dprop = take(P * C);
    t.dpa = take(P * C); t.dpb = take(P * C); t.tmpP = take(P * d);
    {
        size_t m = t_linear_dw_scratch_floats((int)Ec, (int)(3 * d), (int)d);
        const size_t n1 = t_linear_dw_scratch_floats((int)Nn, (int)d, (int)d), n0 = t_linear_dw_scratch_floats((int)Nn, (int)K0, (int)d);
        m = m > n1 ? m : n1;
        t.dwp = take(m > n0 ? m : n0);
    }
    t.total_floats = o;
    return true;
}

struct SmW { const float* w; const float* b; float* gw; float* gb; };
SmW sm_wref(const gnnmp_smoother* h, float* grad, const std::string& name) {
    SmW r{nullptr, nullptr, nullptr, nullptr};
    for (size_t i = 0; i < h->man.size(); ++i) {
        if (h->man[i].name == name + ".weight") { r.w = h->w_raw_dev + h->man_off[i]; r.gw = grad ? grad + h->man_off[i] : nullptr; }
        if (h->man[i].name == name + ".bias") { r.b = h->w_raw_dev + h->man_off[i]; r.gb = grad ? grad + h->man_off[i] : nullptr; }
    }
    return r;
}

bool sm_train_ok(const gnnmp_smoother* h, const gnnmp_smooth_batch* b) {
    return b->path_ptr && b->free_ptr && b->coll_ptr && b->edge_ptr && b->n_problems == 1 && b->total_path >= 1 && h->dims.mlp_dtype == GNNMP_F32 && b->max_samples <= 2048;
}

void sm_fill_params(const gnnmp_smoother* h, const gnnmp_smooth_batch* b, const SmCarve& c, void* ws, SmParams& p) {
    p.B = 1; p.C = h->dims.config_size; p.total_path = b->total_path; p.total_edges = b->total_edges;
    p.scale = h->dims.scale;
    p.path = b->path; p.free_pts = b->free_pts; p.collided = b->collided;
    p.edge_index = reinterpret_cast<const long long*>(b->edge_index);
    p.path_ptr = b->path_ptr; p.free_ptr = b->free_ptr; p.coll_ptr = b->coll_ptr; p.edge_ptr = b->edge_ptr;
    p.w = h->w_dev; p.L = h->L;
    p.knn = at<int>(ws, c.knn);
    p.e_src = at<int>(ws, c.e_src); p.e_dst = at<int>(ws, c.e_dst); p.e_count = at<int>(ws, c.e_count);
    p.stat = at<int>(ws, c.stat);
    p.seg_beg = at<int>(ws, c.seg_beg); p.seg_cnt = at<int>(ws, c.seg_cnt);
    p.etile_prob = at<int>(ws, c.etile); p.ptile_prob = at<int>(ws, c.ptile);
    p.msg = at<float>(ws, c.msg);
    p.tgt = at<float>(ws, c.tgt); p.tgt_flag = at<int>(ws, c.tflag);
    p.tile_cnt = nullptr; p.elist = nullptr; p.plist = nullptr; p.parity = 0;      // training path: no tile lists (its kernels walk the padded tile space)
    { static const int no_tgt = getenv("GNNMP_SM_NO_TARGET_ROLE") ? atoi(getenv("GNNMP_SM_NO_TARGET_ROLE")) : 0; if (no_tgt) p.tgt_flag = nullptr; }      // experiments
    p.cand_cap = b->max_edges + kSmK * b->max_path;
    if (p.cand_cap < 1) p.cand_cap = 1;
    p.samp_cap = b->max_samples > 0 ? b->max_samples : 0; p.path_cap = b->max_path > 0 ? b->max_path : 0;
    p.n_etiles = c.ecap / 32; p.n_ptiles = c.pcap / 32;
    p.one_free = b->total_free; p.one_coll = b->total_collided; p.init_from_path = 0; p.out = nullptr;
}

}  // namespace

extern "C" int64_t gnnmp_smoother_grad_floats(const gnnmp_smoother* h) { return h ? h->n_raw : GNNMP_ERR_NULL; }

extern "C" int gnnmp_smoother_train_workspace_bytes(const gnnmp_smoother* h, const gnnmp_smooth_batch* shape, int loop, size_t* bytes) {
    if (!h || !shape || !bytes) return GNNMP_ERR_NULL;
    if (loop < 0) return GNNMP_ERR_ARG;
    if (!sm_train_ok(h, shape)) return GNNMP_ERR_DIMS;
    SmCarve c;
    if (!sm_carve(h, shape, c)) return GNNMP_ERR_ARG;
    SmTrainCarve t;
    sm_train_carve(h, shape, loop, c, t);
    *bytes = t.inf_bytes + t.total_floats * sizeof(float);
    return GNNMP_OK;
}

extern "C" int gnnmp_smoother_train_forward(const gnnmp_smoother* h, const gnnmp_smooth_batch* b, int loop, float* out_path,
                                            float* bn_stats, void* ws, size_t ws_bytes, void* hip_stream) {
    if (!h || !b || !ws || !out_path) return GNNMP_ERR_NULL;
    if (loop < 0) return GNNMP_ERR_ARG;
    if (!sm_train_ok(h, b)) return GNNMP_ERR_DIMS;
    SmCarve c;
    if (!sm_carve(h, b, c)) return GNNMP_ERR_ARG;
    SmTrainCarve t;
    sm_train_carve(h, b, loop, c, t);
    if (ws_bytes < t.inf_bytes + t.total_floats * sizeof(float) || (reinterpret_cast<uintptr_t>(ws) & 255)) return GNNMP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    float* T = reinterpret_cast<float*>(static_cast<char*>(ws) + t.inf_bytes);
    const int d = h->dims.embed_size, C = h->dims.config_size, P = b->total_path, F = b->total_free, Co = b->total_collided;
    const int Nn = t.Nn, K0 = t.K0, Ec = t.ecap;
    SmParams p;
    sm_fill_params(h, b, c, ws, p);
    if ((size_t)2 * p.cand_cap * sizeof(int) > 60000) return GNNMP_ERR_DIMS;
    const SmW nc0 = sm_wref(h, nullptr, "node_code.0"), bn = sm_wref(h, nullptr, "node_code.1"), nc3 = sm_wref(h, nullptr, "node_code.3"),
              l00 = sm_wref(h, nullptr, "process.lin_0.0"), l02 = sm_wref(h, nullptr, "process.lin_0.2"),
              l10 = sm_wref(h, nullptr, "process.lin_1.0"), l12 = sm_wref(h, nullptr, "process.lin_1.2"), sn = sm_wref(h, nullptr, "smooth_node");
    float* states = T + t.states;
    HIP_TRY(hipMemsetAsync(T, 0, t.total_floats * sizeof(float), st));                         // edge slots past the count stay finite
    HIP_TRY(launch_sm_init(P * C, p.scale, b->path, states, st));                              // model_smoother.py:118
    for (int it = 0; it < loop; ++it) {
        float* I = T + t.it0 + t.it_stride * (size_t)it;
        float* cur = states + (size_t)it * P * C;
        p.cur = cur; p.cur_next = cur;
        HIP_TRY(hipMemsetAsync(at<char>(ws, c.ff_beg), 0xFF, c.ff_end - c.ff_beg, st));
        HIP_TRY(launch_sm_knn_edges(p, st));                                                    // :125-128
        HIP_TRY(hipMemcpyAsync(I + t.esrc, p.e_src, sizeof(int) * Ec, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(I + t.edst, p.e_dst, sizeof(int) * Ec, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(I + t.ne, p.e_count, sizeof(int), hipMemcpyDeviceToDevice, st));
        const int* ne = reinterpret_cast<const int*>(I + t.ne);
        const int* es = reinterpret_cast<const int*>(I + t.esrc);
        const int* ed = reinterpret_cast<const int*>(I + t.edst);
        HIP_TRY(t_sm_nodes_in(P, F, Co, C, p.scale, cur, b->free_pts, b->collided, I + t.Xin, st));   // :130-135
        HIP_TRY(t_linear(Nn, K0, d, I + t.Xin, nc0.w, nc0.b, I + t.X0, false, st));
        HIP_TRY(t_bn_fwd(Nn, d, I + t.X0, bn.w, bn.b, I + t.X1, I + t.stats, true, st));       // BatchNorm, batch statistics + ReLU
        HIP_TRY(t_linear(Nn, d, d, I + t.X1, nc3.w, nc3.b, I + t.X, false, st));
        HIP_TRY(t_sm_msg_in(ne, d, es, ed, I + t.X, T + t.Zin, Ec, st));                       // :36-37
        HIP_TRY(t_linear(Ec, 3 * d, d, T + t.Zin, l00.w, l00.b, I + t.Zh, true, st));
        HIP_TRY(t_linear(Ec, d, d, I + t.Zh, l02.w, l02.b, T + t.M, false, st));
        HIP_TRY(t_fill((size_t)P * d, I + t.S, 0.f, st));
        HIP_TRY(t_sm_scatter_add(ne, d, ed, T + t.M, I + t.S, P, st));                        // aggr = 'add' (:32)
        HIP_TRY(t_linear(P, d, d, I + t.S, l10.w, l10.b, I + t.L1h, true, st));
        HIP_TRY(t_linear(P, d, d, I + t.L1h, l12.w, l12.b, T + t.tmpP, false, st));
        HIP_TRY(t_add_rows((size_t)P * d, I + t.X, T + t.tmpP, I + t.Hh, st));                 // x + lin_1(out) (:34), rows < P
        HIP_TRY(t_linear(P, d, C, I + t.Hh, sn.w, sn.b, T + t.prop, false, st));
        HIP_TRY(t_sm_path_update(P, C, cur, T + t.prop, cur + (size_t)P * C, st));             // :139
        if (bn_stats) {
            HIP_TRY(hipMemcpyAsync(bn_stats + (size_t)it * 2 * d, I + t.stats, sizeof(float) * d, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(bn_stats + (size_t)it * 2 * d + d, I + t.stats + 2 * d, sizeof(float) * d, hipMemcpyDeviceToDevice, st));
        }
    }
    HIP_TRY(t_scale(P * C, p.scale, states + (size_t)loop * P * C, out_path, st));                  // :142
    return GNNMP_OK;
}

extern "C" int gnnmp_smoother_train_backward(const gnnmp_smoother* h, const gnnmp_smooth_batch* b, int loop, const float* d_out_path,
                                             float* grad, void* ws, size_t ws_bytes, void* hip_stream) {
    if (!h || !b || !ws || !grad || !d_out_path) return GNNMP_ERR_NULL;
    if (loop < 0) return GNNMP_ERR_ARG;
    if (!sm_train_ok(h, b)) return GNNMP_ERR_DIMS;
    SmCarve c;
    if (!sm_carve(h, b, c)) return GNNMP_ERR_ARG;
    SmTrainCarve t;
    sm_train_carve(h, b, loop, c, t);
    if (ws_bytes < t.inf_bytes + t.total_floats * sizeof(float) || (reinterpret_cast<uintptr_t>(ws) & 255)) return GNNMP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    float* T = reinterpret_cast<float*>(static_cast<char*>(ws) + t.inf_bytes);
    const int d = h->dims.embed_size, C = h->dims.config_size, P = b->total_path;
    const int Nn = t.Nn, K0 = t.K0, Ec = t.ecap;
    HIP_TRY(hipMemsetAsync(grad, 0, (size_t)h->n_raw * sizeof(float), st));
    const SmW nc0 = sm_wref(h, grad, "node_code.0"), bn = sm_wref(h, grad, "node_code.1"), nc3 = sm_wref(h, grad, "node_code.3"),
              l00 = sm_wref(h, grad, "process.lin_0.0"), l02 = sm_wref(h, grad, "process.lin_0.2"),
              l10 = sm_wref(h, grad, "process.lin_1.0"), l12 = sm_wref(h, grad, "process.lin_1.2"), sn = sm_wref(h, grad, "smooth_node");
    float* dcur = T + t.dpa;
    float* dprev = T + t.dpb;
    HIP_TRY(t_scale(P * C, h->dims.scale, d_out_path, dcur, st));
    for (int it = loop - 1; it >= 0; --it) {
        float* I = T + t.it0 + t.it_stride * (size_t)it;
        const int* ne = reinterpret_cast<const int*>(I + t.ne);
        const int* es = reinterpret_cast<const int*>(I + t.esrc);
        const int* ed = reinterpret_cast<const int*>(I + t.edst);
        HIP_TRY(t_sm_path_update_bwd(P, C, dcur, T + t.dprop, dprev, st));
        HIP_TRY(t_linear_dw(P, d, C, T + t.dprop, I + t.Hh, sn.gw, sn.gb, T + t.dwp, st));
        HIP_TRY(t_linear_dx(P, d, C, T + t.dprop, sn.w, T + t.dHh, false, st));
        HIP_TRY(t_fill((size_t)Nn * d, T + t.dX, 0.f, st));
        HIP_TRY(hipMemcpyAsync(T + t.dX, T + t.dHh, sizeof(float) * P * d, hipMemcpyDeviceToDevice, st));   // x + ...: identity branch
        HIP_TRY(t_linear_dw(P, d, d, T + t.dHh, I + t.L1h, l12.gw, l12.gb, T + t.dwp, st));
        HIP_TRY(t_linear_dx(P, d, d, T + t.dHh, l12.w, T + t.dL1h, false, st));
        HIP_TRY(t_relu_bwd((size_t)P * d, I + t.L1h, T + t.dL1h, st));
        HIP_TRY(t_linear_dw(P, d, d, T + t.dL1h, I + t.S, l10.gw, l10.gb, T + t.dwp, st));
        HIP_TRY(t_linear_dx(P, d, d, T + t.dL1h, l10.w, T + t.dS, false, st));
        HIP_TRY(t_sm_scatter_add_bwd(ne, d, ed, T + t.dS, T + t.M, Ec, st));                    // dM
        HIP_TRY(t_linear_dw(Ec, d, d, T + t.M, I + t.Zh, l02.gw, l02.gb, T + t.dwp, st));
        HIP_TRY(t_linear_dx(Ec, d, d, T + t.M, l02.w, T + t.dZh, false, st));
        HIP_TRY(t_relu_bwd((size_t)Ec * d, I + t.Zh, T + t.dZh, st));
        HIP_TRY(t_sm_msg_in(ne, d, es, ed, I + t.X, T + t.Zin, Ec, st));                        // Zin recomputed
        HIP_TRY(t_linear_dw(Ec, 3 * d, d, T + t.dZh, T + t.Zin, l00.gw, l00.gb, T + t.dwp, st));
        HIP_TRY(t_linear_dx(Ec, 3 * d, d, T + t.dZh, l00.w, T + t.dZin, false, st));
        HIP_TRY(t_sm_msg_in_bwd(ne, d, es, ed, T + t.dZin, T + t.dX, Nn, st));
        HIP_TRY(t_linear_dw(Nn, d, d, T + t.dX, I + t.X1, nc3.gw, nc3.gb, T + t.dwp, st));
        HIP_TRY(t_linear_dx(Nn, d, d, T + t.dX, nc3.w, T + t.dX1, false, st));
        HIP_TRY(t_relu_bwd((size_t)Nn * d, I + t.X1, T + t.dX1, st));
        HIP_TRY(t_bn_bwd(Nn, d, I + t.X0, T + t.dX1, bn.w, I + t.stats, T + t.dX0, bn.gw, bn.gb, st));
        HIP_TRY(t_linear_dw(Nn, K0, d, T + t.dX0, I + t.Xin, nc0.gw, nc0.gb, T + t.dwp, st));
        HIP_TRY(t_linear_dx(Nn, K0, d, T + t.dX0, nc0.w, T + t.dXin, false, st));
        HIP_TRY(t_sm_coords_bwd(P, C, T + t.dXin, dprev, st));                                  // nodes[:P] = path (:140)
        float* sw = dcur; dcur = dprev; dprev = sw;
    }
    return GNNMP_OK;
}
