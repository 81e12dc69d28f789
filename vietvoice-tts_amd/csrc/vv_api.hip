// C ABI + native stage drivers: context, named weight binding, workspace arena, the preprocess /
// transformer-steps / decode launch sequences, and HIP-event profiling per kernel class.
// No torch types, no exceptions across the ABI, no device allocation inside the step loop.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "vv_kernels.h"

#define VV_VERSION_STR "vvtts-hip 0.1 (gfx950)"

namespace {
std::string g_create_error;

struct Bound { const void* p; uint64_t bytes; };

struct ProfRec { hipEvent_t a, b; int cls, sub; };

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int pad_to(int v, int a) { return (v + a - 1) / a * a; }
}  // namespace

struct vv_ctx {
    int device = 0;
    vv_model_cfg cfg{};
    int dt = VV_DTYPE_BF16;             // acoustic operand dtype
    std::string err;
    std::map<std::string, Bound> w;
    bool finalized = false;
    float post_bias = 0.f;
    // time grid tables
    int n_steps = 0;
    std::vector<float> dt_host;
    float* modtab = nullptr;            // [depth][n_steps][6D]
    float* fintab = nullptr;            // [n_steps][2D]
    // workspace arena
    char* ws = nullptr;                 // context-owned arena: may MOVE when a later call needs more bytes (ensure_ws)
    size_t ws_cap = 0;
    char* ar = nullptr;                 // active arena of the current call: ws, or a caller-owned block (vv_decode_into)
    size_t ar_cap = 0, ar_off = 0;
    uint64_t ws_generation = 0;         // bumped whenever ws is reallocated
    int* d_mult = nullptr;              // decode length multipliers
    int rope_rows = 0;                  // 1: the QKV rope epilogue reads row-gathered tables (vv_rope_rows).  Off by default: 11 % faster in a
                                        // back-to-back GEMM loop (tables stay cached), 0.9 % SLOWER inside the step, where the 52 MB of row
                                        // tables come from HBM each time while the 0.8 MB position tables stay in L2 (profiles/r02/gemm_notes.md)
    int split_k_tail = 0;               // bf16 gate-store GEMMs: split the K range of the last row panels when the tile count leaves a partial
                                        // last round of the persistent kernel (vv_gemm_tail_plan); the LayerNorm sums the fp32 parts and rounds
                                        // once.  0 off (default), 1 out-projection and FF2, 2 FF2 only.  OFF since round 4: a tail row's fp32
                                        // summation order differs from a plain row's, the next bf16 rounding turns that into bf16-level noise
                                        // (22 LSB of PCM between two batchings of one text, tests/test_longform_gpu.py), and with the parts in
                                        // fp32 the tail no longer pays either (GEMM -3.6 ms, norms +7.1 ms per headline step,
                                        // profiles/r04/tail_fp32_notes.md).  Off, every row's arithmetic is independent of its batch neighbours.
    int chip_share = 1;                 // 2 while a call runs two lanes (vv_gemm_args.chip_share of its GEMM launches)
    int ring_tiles_max = 256;           // the tile-count bound of that rule (the CU count)
    int ring_tiles = 1;                 // 1 (default since round 5; same bits): N <= 1024 bf16 GEMMs whose 64 x 128 tiles are fewer than the CUs take the 64 x 64 three-stage-ring tiling (vv_gemm tile 6464)
    int pp_min_tiles = -1;              // bf16 GEMMs of the path: -1 = the launcher's own choice between the persistent 256 x 256 kernel and the 128 x 128 one
                                        // (vv_gemm.hip launch(): about one full round of 256-tiles, or the wide QKV shape); n >= 0 = the persistent
                                        // kernel whenever M >= 4096, N % 256 == 0 and the shape has >= n 256-tiles (0 = the rule of rounds 1-3; lets a
                                        // small model exercise the persistent kernel in the whole pipeline, tests/test_e2e_gpu.py).  Same bits either way.
    int lanes = 0;                      // transformer steps as two half batches on two streams: 0 auto (bf16, >= VV_LANE_MIN_ROWS packed rows), 1 never,
                                        // 2 whenever B >= 2
    hipStream_t side_stream[1] = {};    // lane 1's stream (created on first use), forked from / joined to the caller's stream inside the call
    hipEvent_t ev_fork = nullptr, ev_join[1] = {};
    float rope_theta = 0.f;             // > 0 (vv_set_rope_theta): the caller's rope tables are the standard ones of this base; the bf16 QKV epilogue then
                                        // computes cos / sin from the position (no table load), q leaves the GEMM without the softmax scale and the
                                        // attention kernel applies it (q_scale).  Overrides rope_q_attn.
    int rope_q_attn = 1;                // bf16: 1 = the QKV GEMM ropes the k columns only and the attention kernel ropes Q while loading it
    int voc_x3 = -1;                    // vocoder conv products: 0 = v_mfma_f32_32x32x2_f32, 1 = exact 3-way bf16 split on the bf16 matrix pipe
                                        // (vv_vocoder_x3.hip: six piece products, fp32 accumulate, fp32 fidelity); -1 = by acoustic dtype (bf16
                                        // context: 1, fp32 context: 0 -- the numerics configuration stays on the f32 instruction)
    int voc_x3_rows = 0;                // x3 workgroup rows for stages with > 64 rows: 0 = default (64, 4 waves), 128 = 8-wave workgroups
    char* x3_buf = nullptr;             // split weight slabs of every vocoder conv (built by vv_finalize_weights)
    std::map<std::string, const void*> x3_w;
    int fuse_mrf = 2;                   // K12 fused MRF pairs (C <= 64 stages): 0 never, 1 always, 2 auto = for decodes of <= 8 items
                                        // (fewer launches win when the stage is launch-bound; at B = 32 the halo recompute costs 1.3 %)
    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    int64_t p_launch[VV_PROF_NCLASS]{};
    double p_flops[VV_PROF_NCLASS]{}, p_bytes[VV_PROF_NCLASS]{}, p_ms[VV_PROF_NCLASS]{};

    int fail(int code, const char* fmt, ...) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
    const void* W(const std::string& n) const {
        auto it = w.find(n);
        return it == w.end() ? nullptr : it->second.p;
    }
    const float* Wf(const std::string& n) const { return (const float*)W(n); }
    int esz() const { return dt == VV_DTYPE_BF16 ? 2 : 4; }
    int bk() const { return dt == VV_DTYPE_BF16 ? 64 : 32; }
};

namespace {

#define HIPCHK(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) return (ctx)->fail(-5, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

#define KCHK(ctx, call)                                              \
    do {                                                             \
        const char* m__ = "";                                        \
        int r__ = (call);                                            \
        if (r__ != 0) return (ctx)->fail(r__, "%s", m__);            \
    } while (0)

struct Prof {
    vv_ctx* c; int idx = -1; hipStream_t st;
    Prof(vv_ctx* c_, int cls, double flops, double bytes, hipStream_t st_, int sub = -1) : c(c_), st(st_) {
        if (!c->prof) return;
        ProfRec r; r.cls = cls; r.sub = sub;
        if (sub >= 0) { c->p_launch[sub]++; c->p_flops[sub] += flops; c->p_bytes[sub] += bytes; }
        auto get = [&]() { hipEvent_t e; if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); } else hipEventCreate(&e); return e; };
        r.a = get(); r.b = get();
        hipEventRecord(r.a, st);
        c->recs.push_back(r); idx = (int)c->recs.size() - 1;
        c->p_launch[cls]++; c->p_flops[cls] += flops; c->p_bytes[cls] += bytes;
    }
    ~Prof() { if (idx >= 0) hipEventRecord(c->recs[idx].b, st); }
};

int ensure_ws(vv_ctx* c, size_t bytes) {
    if (bytes > c->ws_cap) {
        if (c->ws) { hipDeviceSynchronize(); hipFree(c->ws); c->ws = nullptr; c->ws_cap = 0; }
        const size_t want = align_up(bytes + bytes / 16, 1 << 20);
        hipError_t e = hipMalloc((void**)&c->ws, want);
        if (e != hipSuccess) return c->fail(-12, "workspace hipMalloc(%zu MiB): %s", want >> 20, hipGetErrorString(e));
        c->ws_cap = want;
        ++c->ws_generation;
    }
    c->ar = c->ws; c->ar_cap = c->ws_cap; c->ar_off = 0;
    return 0;
}
// A caller-owned block as the arena of one call (memory that can never move: what a captured hipGraph must point into).
int use_ws(vv_ctx* c, void* block, size_t have, size_t need) {
    if ((uintptr_t)block % 256) return c->fail(-22, "workspace block must be 256-byte aligned");
    if (have < need) return c->fail(-22, "workspace block too small: %zu < %zu bytes", have, need);
    c->ar = (char*)block; c->ar_cap = have; c->ar_off = 0;
    return 0;
}
template <typename T> T* carve(vv_ctx* c, size_t n) {
    c->ar_off = align_up(c->ar_off, 256);
    T* p = (T*)(c->ar + c->ar_off);
    c->ar_off += n * sizeof(T);
    return p;
}
struct Need { size_t b = 0; void add(size_t bytes) { b = align_up(b, 256) + bytes; } };

// ---- GEMM helper over bound weights -------------------------------------------------------
int gemm(vv_ctx* c, int dtype, int out_dtype, int mode, int act, const void* A, int lda, const char* wname, int ldw, const char* bname,
         void* C, int ldc, int M, int N, int K, hipStream_t st, const float* gate = nullptr, int n_store = 0,
         const float* const* rope = nullptr, int seq_n = 0, int rope_dim = 0, double alg_flops = -1, const int* rope_pos = nullptr,
         int rope_by_row = 0, void* c_tail = nullptr, int tail_row0 = 0, int tail_parts = 0, int rope_skip_q = 0, float rope_theta = 0.f) {
    // (c->chip_share: 2 while vv_transformer_steps runs two lanes -- set around its launch loop)
    // rope: [cos_q, sin_q, cos_k, sin_k, compact_q, compact_k]
    vv_gemm_args g{};
    g.dtype = dtype; g.out_dtype = out_dtype; g.mode = mode; g.act = act;
    g.A = A; g.lda = lda; g.W = c->W(wname); g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bname ? c->Wf(bname) : nullptr; g.gate = gate; g.n_store = n_store; g.seq_n = seq_n; g.rope_dim = rope_dim;
    if (rope) { g.cos_q = rope[0]; g.sin_q = rope[1]; g.cos_k = rope[2]; g.sin_k = rope[3]; g.rope_cs_q = rope[4]; g.rope_cs_k = rope[5]; }
    g.rope_pos = rope_pos; g.rope_by_row = rope_by_row;
    g.C_tail = c_tail; g.tail_row0 = tail_row0; g.tail_parts = tail_parts; g.rope_skip_q = rope_skip_q; g.rope_theta = rope_theta;
    g.chip_share = c->chip_share;
    if (!g.W) return c->fail(-2, "weight '%s' is not bound", wname);
    if (c->pp_min_tiles >= 0 && dtype == VV_DTYPE_BF16 && N % 256 == 0)
        g.tile = (M >= 4096 && (long long)((M + 255) / 256) * (N / 256) >= c->pp_min_tiles) ? 256 : 128;
    if (c->ring_tiles && g.tile == 0 && dtype == VV_DTYPE_BF16 && N <= 1024 && N % 64 == 0 && (long long)((M + 63) / 64) * (N / 128) <= c->ring_tiles_max)
        g.tile = 6464;
    const int esz = dtype == VV_DTYPE_BF16 ? 2 : 4, osz = out_dtype == VV_DTYPE_BF16 ? 2 : 4;
    const double fl = alg_flops >= 0 ? alg_flops : 2.0 * M * (double)N * K;
    Prof p(c, VV_PROF_GEMM, fl, (double)M * K * esz + (double)N * K * esz + (double)M * N * osz * (mode == VV_EPI_GATE_RES ? 2 : 1), st);
    const char* m = "";
    int r = vvk_gemm(&g, st, &m);
    if (r) return c->fail(r, "%s (weight %s, M=%d N=%d K=%d)", m, wname, M, N, K);
    return 0;
}

std::string blk(int i, const char* s) { return "blocks." + std::to_string(i) + s; }

}  // namespace

// =====================================================================================  ABI
extern "C" {

const char* vv_version(void) { return VV_VERSION_STR; }

const char* vv_last_error(const vv_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int vv_create(vv_ctx** out, int device, const vv_model_cfg* cfg, int acoustic_dtype) {
    if (!out || !cfg) { g_create_error = "vv_create: null argument"; return -22; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_create_error = "vv_create: no HIP device visible (the HIP path has no CPU fallback)"; return -19; }
    if (device < 0 || device >= n) { g_create_error = "vv_create: device index out of range"; return -22; }
    if (acoustic_dtype != VV_DTYPE_F32 && acoustic_dtype != VV_DTYPE_BF16) { g_create_error = "vv_create: dtype must be f32 or bf16"; return -22; }
    if (cfg->head_dim != 64 || cfg->heads * 64 != cfg->dim) { g_create_error = "vv_create: head_dim must be 64"; return -22; }
    if (cfg->dim % 128 || cfg->text_dim % 128 || cfg->dim / cfg->pos_conv_groups != 64 || cfg->n_mel % 4 || cfg->dim > 1024 ||
        cfg->text_dim > 1024) {
        g_create_error = "vv_create: dim/text_dim must be multiples of 128 (<=1024), 64 channels per pos-conv group"; return -22;
    }
    if (cfg->voc_n_up < 1 || cfg->voc_n_up > VV_MAX_UP || cfg->voc_n_res > VV_MAX_RES || cfg->voc_n_dil > VV_MAX_RES) {
        g_create_error = "vv_create: vocoder topology out of range"; return -22;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return -5; }
    vv_ctx* c = new vv_ctx();
    c->device = device; c->cfg = *cfg; c->dt = acoustic_dtype;
    *out = c;
    return 0;
}

void vv_destroy(vv_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    if (c->ws) hipFree(c->ws);
    if (c->modtab) hipFree(c->modtab);
    if (c->fintab) hipFree(c->fintab);
    if (c->d_mult) hipFree(c->d_mult);
    if (c->x3_buf) hipFree(c->x3_buf);
    for (auto q : c->side_stream) if (q) hipStreamDestroy(q);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    for (auto e : c->ev_join) if (e) hipEventDestroy(e);
    for (auto& r : c->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (auto e : c->pool) hipEventDestroy(e);
    delete c;
}

int vv_bind_weight(vv_ctx* c, const char* name, const void* p, uint64_t bytes) {
    if (!c || !name || !p) return c ? c->fail(-22, "vv_bind_weight: null argument") : -22;
    if ((uintptr_t)p % 16) return c->fail(-22, "vv_bind_weight(%s): pointer must be 16-byte aligned", name);
    c->w[name] = Bound{p, bytes};
    c->finalized = false;
    return 0;
}

int vv_finalize_weights(vv_ctx* c) {
    if (!c) return -22;
    const vv_model_cfg& g = c->cfg;
    const int D = g.dim, Dt = g.text_dim, FF = g.dim * g.ff_mult, es = c->esz();
    const int KP = pad_to(2 * g.n_mel + Dt, 64), MP = pad_to(g.n_mel, 128);
    std::string missing;
    auto need = [&](const std::string& n, uint64_t bytes) {
        auto it = c->w.find(n);
        if (it == c->w.end()) { missing += n + " "; return; }
        if (it->second.bytes < bytes) missing += n + "(short:" + std::to_string(it->second.bytes) + "<" + std::to_string(bytes) + ") ";
    };
    need("const.window", 4ull * g.n_fft); need("const.tw_cos", 4ull * g.n_fft); need("const.tw_sin", 4ull * g.n_fft);
    need("const.mel_fb", 4ull * (g.n_fft / 2 + 1) * g.n_mel);
    need("const.text_pos", 4ull * g.max_pos * Dt);
    need("text.embed.weight", 4ull * g.vocab_rows * Dt);
    for (int i = 0; i < g.text_layers; ++i) {
        const std::string p = "text.blocks." + std::to_string(i);
        need(p + ".dwconv.weight", 4ull * Dt * g.text_conv_k); need(p + ".dwconv.bias", 4ull * Dt);
        need(p + ".norm.weight", 4ull * Dt); need(p + ".norm.bias", 4ull * Dt);
        need(p + ".pwconv1.weight", (uint64_t)es * Dt * g.text_ff_mult * Dt); need(p + ".pwconv1.bias", 4ull * Dt * g.text_ff_mult);
        need(p + ".grn.gamma", 4ull * Dt * g.text_ff_mult); need(p + ".grn.beta", 4ull * Dt * g.text_ff_mult);
        need(p + ".pwconv2.weight", (uint64_t)es * Dt * g.text_ff_mult * Dt); need(p + ".pwconv2.bias", 4ull * Dt);
    }
    need("input.proj.weight", (uint64_t)es * D * KP); need("input.proj.bias", 4ull * D);
    for (int j = 1; j <= 2; ++j) {
        need("input.pos_conv" + std::to_string(j) + ".weight", (uint64_t)es * D * 64 * g.pos_conv_k);
        need("input.pos_conv" + std::to_string(j) + ".bias", 4ull * D);
    }
    need("time.mlp1.weight", 4ull * D * g.time_freq_dim); need("time.mlp1.bias", 4ull * D);
    need("time.mlp2.weight", 4ull * D * D); need("time.mlp2.bias", 4ull * D);
    for (int i = 0; i < g.depth; ++i) {
        need(blk(i, ".adaln.weight"), 4ull * 6 * D * D); need(blk(i, ".adaln.bias"), 4ull * 6 * D);
        need(blk(i, ".attn.qkv.weight"), (uint64_t)es * 3 * D * D); need(blk(i, ".attn.qkv.bias"), 4ull * 3 * D);
        need(blk(i, ".attn.out.weight"), (uint64_t)es * D * D); need(blk(i, ".attn.out.bias"), 4ull * D);
        need(blk(i, ".ff1.weight"), (uint64_t)es * FF * D); need(blk(i, ".ff1.bias"), 4ull * FF);
        need(blk(i, ".ff2.weight"), (uint64_t)es * D * FF); need(blk(i, ".ff2.bias"), 4ull * D);
    }
    need("final.adaln.weight", 4ull * 2 * D * D); need("final.adaln.bias", 4ull * 2 * D);
    need("final.proj.weight", (uint64_t)es * MP * D); need("final.proj.bias", 4ull * MP);
    int ch = g.voc_pre_ch;
    struct ConvW { std::string name; int cin_pad, kw, rows_pad; };
    std::vector<ConvW> convs;                              // every vocoder conv slab [cin_pad][kw][rows_pad]: split for the x3 kernels below
    convs.push_back({"voc.pre.weight", pad_to(g.n_mel, 8), g.voc_pre_k, pad_to(ch, 64)});
    need("voc.pre.weight", 4ull * pad_to(g.n_mel, 8) * g.voc_pre_k * pad_to(ch, 64)); need("voc.pre.bias", 4ull * ch);
    for (int s = 0; s < g.voc_n_up; ++s) {
        const int cin = ch, cout = ch / 2, u = g.voc_up_rates[s];
        const std::string p = "voc.up." + std::to_string(s);
        need(p + ".weight", 4ull * pad_to(cin, 8) * 2 * pad_to(cout * u, 64)); need(p + ".bias", 4ull * cout);
        convs.push_back({p + ".weight", pad_to(cin, 8), 2, pad_to(cout * u, 64)});
        for (int a = 0; a < g.voc_n_res; ++a)
            for (int b = 0; b < g.voc_n_dil; ++b)
                for (int k = 1; k <= 2; ++k) {
                    const std::string q = "voc.res." + std::to_string(s) + "." + std::to_string(a) + "." + std::to_string(b) + ".conv" + std::to_string(k);
                    need(q + ".weight", 4ull * pad_to(cout, 8) * g.voc_res_kernels[a] * pad_to(cout, 64)); need(q + ".bias", 4ull * cout);
                    convs.push_back({q + ".weight", pad_to(cout, 8), g.voc_res_kernels[a], pad_to(cout, 64)});
                }
        ch = cout;
    }
    need("voc.post.weight", 4ull * ch * g.voc_post_k); need("voc.post.bias", 4);
    if (!missing.empty()) return c->fail(-2, "missing or short weights: %s", missing.c_str());
    hipSetDevice(c->device);
    HIPCHK(c, hipMemcpy(&c->post_bias, c->W("voc.post.bias"), 4, hipMemcpyDeviceToHost));
    {   // x3 slabs: w = h + m + l in bf16 pieces, [chunk][kw][piece][row][16] (vv_vocoder_x3.hip); 1.5x the fp32 bytes
        size_t total = 0;
        for (const ConvW& w : convs) total = align_up(total, 256) + vvk_conv_split_bytes(w.cin_pad, w.kw, w.rows_pad);
        if (c->x3_buf) { HIPCHK(c, hipFree(c->x3_buf)); c->x3_buf = nullptr; }
        c->x3_w.clear();
        HIPCHK(c, hipMalloc((void**)&c->x3_buf, total));
        size_t off = 0;
        for (const ConvW& w : convs) {
            off = align_up(off, 256);
            const char* m__ = "";
            if (int r = vvk_conv_split_weights(c->Wf(w.name), w.cin_pad, w.kw, w.rows_pad, c->x3_buf + off, nullptr, &m__))
                return c->fail(r, "%s (%s)", m__, w.name.c_str());
            c->x3_w[w.name] = c->x3_buf + off;
            off += vvk_conv_split_bytes(w.cin_pad, w.kw, w.rows_pad);
        }
        HIPCHK(c, hipStreamSynchronize(nullptr));
    }
    // decode length multipliers: level 0 = frames, level s+1 = after upsample s
    std::vector<int> mult(g.voc_n_up + 1, 1);
    for (int s = 0; s < g.voc_n_up; ++s) mult[s + 1] = mult[s] * g.voc_up_rates[s];
    if (!c->d_mult) HIPCHK(c, hipMalloc((void**)&c->d_mult, sizeof(int) * (VV_MAX_UP + 1)));
    HIPCHK(c, hipMemcpy(c->d_mult, mult.data(), sizeof(int) * mult.size(), hipMemcpyHostToDevice));
    c->finalized = true;
    return 0;
}

int vv_set_time_grid(vv_ctx* c, const float* sinus_host, const float* dt_host, int n_steps, void* stream) {
    if (!c) return -22;
    if (!c->finalized) return c->fail(-1, "vv_set_time_grid: weights not finalized");
    if (n_steps < 1 || n_steps > 512 || !sinus_host || !dt_host) return c->fail(-22, "vv_set_time_grid: bad arguments");
    hipSetDevice(c->device);
    hipStream_t st = (hipStream_t)stream;
    const int D = c->cfg.dim, S = n_steps, TF = c->cfg.time_freq_dim, L = c->cfg.depth;
    if (TF % 32) return c->fail(-22, "time_freq_dim must be a multiple of 32");
    if (c->modtab) { hipDeviceSynchronize(); hipFree(c->modtab); hipFree(c->fintab); c->modtab = c->fintab = nullptr; }
    HIPCHK(c, hipMalloc((void**)&c->modtab, sizeof(float) * (size_t)L * S * 6 * D));
    HIPCHK(c, hipMalloc((void**)&c->fintab, sizeof(float) * (size_t)S * 2 * D));
    Need nd; nd.add(4ull * S * TF); nd.add(4ull * S * D); nd.add(4ull * S * D);
    if (int r = ensure_ws(c, nd.b)) return r;
    float* sin_d = carve<float>(c, (size_t)S * TF);
    float* t1 = carve<float>(c, (size_t)S * D);
    float* t2 = carve<float>(c, (size_t)S * D);
    HIPCHK(c, hipMemcpyAsync(sin_d, sinus_host, 4ull * S * TF, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));     // host buffer may be pageable; keep its lifetime simple
    if (int r = gemm(c, VV_DTYPE_F32, VV_DTYPE_F32, VV_EPI_STORE, VV_ACT_SILU_, sin_d, TF, "time.mlp1.weight", TF, "time.mlp1.bias", t1, D, S, D, TF, st)) return r;
    if (int r = gemm(c, VV_DTYPE_F32, VV_DTYPE_F32, VV_EPI_STORE, VV_ACT_SILU_, t1, D, "time.mlp2.weight", D, "time.mlp2.bias", t2, D, S, D, D, st)) return r;
    // t2 = SiLU(t_emb): every AdaLN consumes the embedding through SiLU only
    for (int l = 0; l < L; ++l) {
        const std::string wn = blk(l, ".adaln.weight"), bn = blk(l, ".adaln.bias");
        if (int r = gemm(c, VV_DTYPE_F32, VV_DTYPE_F32, VV_EPI_STORE, VV_ACT_NONE_, t2, D, wn.c_str(), D, bn.c_str(),
                         c->modtab + (size_t)l * S * 6 * D, 6 * D, S, 6 * D, D, st)) return r;
    }
    if (int r = gemm(c, VV_DTYPE_F32, VV_DTYPE_F32, VV_EPI_STORE, VV_ACT_NONE_, t2, D, "final.adaln.weight", D, "final.adaln.bias", c->fintab, 2 * D, S, 2 * D, D, st)) return r;
    HIPCHK(c, hipStreamSynchronize(st));
    c->n_steps = S;
    c->dt_host.assign(dt_host, dt_host + S);
    return 0;
}

// --------------------------------------------------------------------------------- preprocess
static int preprocess_impl(vv_ctx* c, int B, int N, const int16_t* audio, int ld_audio, int max_audio_len, const int32_t* audio_len,
                           const int32_t* audio_len_host, const int32_t* text_ids, int ld_text, const int32_t* text_len, const int32_t* seq_len,
                           float* cat, float* cat_drop, int32_t* ref_len, void* stream) {
    if (!c) return -22;
    if (!c->finalized) return c->fail(-1, "vv_preprocess: weights not finalized");
    if (B < 1 || N < 1 || !audio || !audio_len || !text_ids || !text_len || !seq_len || !cat || !cat_drop || !ref_len)
        return c->fail(-22, "vv_preprocess: bad arguments");
    const vv_model_cfg& g = c->cfg;
    if (N > g.max_pos) return c->fail(-22, "vv_preprocess: N=%d exceeds the position tables (%d)", N, g.max_pos);
    if (max_audio_len < g.n_fft || max_audio_len > ld_audio) return c->fail(-22, "vv_preprocess: reference clips must hold >= n_fft samples and fit ld_audio");
    // A centred STFT reflects n_fft / 2 samples at both ends of a clip: defined only for clips of more than n_fft / 2 samples (torch.stft
    // refuses shorter ones).  The host lengths let the call refuse such an item before anything is launched; without them the mel kernel
    // clamps the doubly reflected index (finite and deterministic, but no reference defines it).
    if (audio_len_host)
        for (int b = 0; b < B; ++b)
            if (audio_len_host[b] < g.n_fft / 2 + 1 || audio_len_host[b] > max_audio_len)
                return c->fail(-22, "vv_preprocess: reference clip %d has %d samples; a clip needs between n_fft/2 + 1 = %d and max_audio_len = %d",
                               b, audio_len_host[b], g.n_fft / 2 + 1, max_audio_len);
    hipSetDevice(c->device);
    hipStream_t st = (hipStream_t)stream;
    const int Dt = g.text_dim, C2 = Dt * g.text_ff_mult, M = g.n_mel, es = c->esz();
    const int F_max = max_audio_len / g.hop_length + 1;
    const size_t R2 = (size_t)2 * B * N;
    Need nd;
    nd.add(4ull * B * F_max * M); nd.add(4ull * R2 * Dt); nd.add(4ull * R2 * Dt); nd.add((size_t)es * R2 * Dt);
    nd.add((size_t)es * R2 * C2); nd.add(4ull * 2 * B * C2);
    if (int r = ensure_ws(c, nd.b)) return r;
    float* mel = carve<float>(c, (size_t)B * F_max * M);
    float* tx = carve<float>(c, R2 * Dt);
    float* ty = carve<float>(c, R2 * Dt);
    char* th = carve<char>(c, (size_t)es * R2 * Dt);
    char* tm = carve<char>(c, (size_t)es * R2 * C2);
    float* sumsq = carve<float>(c, (size_t)2 * B * C2);

    KCHK(c, vvk_ref_len(audio_len, ref_len, B, g.hop_length, st, &m__));
    {
        Prof p(c, VV_PROF_MEL, 4.0 * B * F_max * (double)g.n_fft * (g.n_fft / 2 + 1), 2.0 * B * max_audio_len + 4.0 * B * F_max * M, st);
        KCHK(c, vvk_mel(audio, ld_audio, audio_len, c->Wf("const.window"), c->Wf("const.tw_cos"), c->Wf("const.tw_sin"),
                        c->Wf("const.mel_fb"), mel, B, F_max, g.n_fft, g.hop_length, M, st, &m__));
    }
    {
        Prof p(c, VV_PROF_TEXT, 0, 8.0 * R2 * Dt, st);
        KCHK(c, vvk_text_embed(text_ids, ld_text, text_len, c->Wf("text.embed.weight"), c->Wf("const.text_pos"), tx, B, N, Dt, g.vocab_rows, st, &m__));
    }
    for (int i = 0; i < g.text_layers; ++i) {
        const std::string p = "text.blocks." + std::to_string(i);
        {
            Prof pr(c, VV_PROF_TEXT, 2.0 * R2 * Dt * g.text_conv_k, 8.0 * R2 * Dt, st);
            KCHK(c, vvk_dwconv(tx, ty, c->Wf(p + ".dwconv.weight"), c->Wf(p + ".dwconv.bias"), seq_len, B, 2 * B, N, Dt, g.text_conv_k, st, &m__));
        }
        {
            vv_ln_args a{}; a.out_dtype = c->dt; a.x = ty; a.ldx = Dt; a.y = th; a.ldy = Dt; a.R = (int)R2; a.D = Dt;
            a.w = c->Wf(p + ".norm.weight"); a.b = c->Wf(p + ".norm.bias"); a.add_one = 0; a.eps = 1e-6f;
            Prof pr(c, VV_PROF_NORM, 0, (4.0 + es) * R2 * Dt, st);
            KCHK(c, vvk_ln_mod(&a, st, &m__));
        }
        if (int r = gemm(c, c->dt, c->dt, VV_EPI_STORE, VV_ACT_GELU_ERF_, th, Dt, (p + ".pwconv1.weight").c_str(), Dt, (p + ".pwconv1.bias").c_str(), tm, C2, (int)R2, C2, Dt, st)) return r;
        {
            Prof pr(c, VV_PROF_TEXT, 0, 3.0 * es * R2 * C2, st);
            KCHK(c, vvk_grn(c->dt, tm, sumsq, c->Wf(p + ".grn.gamma"), c->Wf(p + ".grn.beta"), seq_len, B, 2 * B, N, C2, st, &m__));
        }
        if (int r = gemm(c, c->dt, VV_DTYPE_F32, VV_EPI_GATE_RES, VV_ACT_NONE_, tm, C2, (p + ".pwconv2.weight").c_str(), C2, (p + ".pwconv2.bias").c_str(), tx, Dt, (int)R2, Dt, C2, st)) return r;
    }
    {
        Prof p(c, VV_PROF_ELEMWISE, 0, 12.0 * B * N * (M + Dt), st);
        KCHK(c, vvk_build_cat(mel, F_max, ref_len, tx, cat, cat_drop, B, N, M, Dt, st, &m__));
    }
    return 0;
}

int vv_preprocess(vv_ctx* c, int B, int N, const int16_t* audio, int ld_audio, int max_audio_len, const int32_t* audio_len,
                  const int32_t* text_ids, int ld_text, const int32_t* text_len, const int32_t* seq_len, float* cat,
                  float* cat_drop, int32_t* ref_len, void* stream) {
    return preprocess_impl(c, B, N, audio, ld_audio, max_audio_len, audio_len, nullptr, text_ids, ld_text, text_len, seq_len, cat, cat_drop, ref_len, stream);
}

// Same, with the clip lengths ALSO given on the host (the same values as the device array): an item shorter than n_fft / 2 + 1 samples
// is refused with -22 before anything is launched.
int vv_preprocess_h(vv_ctx* c, int B, int N, const int16_t* audio, int ld_audio, int max_audio_len, const int32_t* audio_len,
                    const int32_t* audio_len_host, const int32_t* text_ids, int ld_text, const int32_t* text_len, const int32_t* seq_len,
                    float* cat, float* cat_drop, int32_t* ref_len, void* stream) {
    if (c && !audio_len_host) return c->fail(-22, "vv_preprocess_h: host lengths missing");
    return preprocess_impl(c, B, N, audio, ld_audio, max_audio_len, audio_len, audio_len_host, text_ids, ld_text, text_len, seq_len, cat, cat_drop, ref_len, stream);
}

// --------------------------------------------------------------------------- transformer steps
// ws_only != nullptr: only compute the workspace bytes the call would carve (nothing is launched; the data pointers may be null).
// ext_ws != nullptr: carve from that caller-owned block instead of the context arena (what a captured hipGraph must point into).
//
// LANES (round 4): a batch of independent items may run as TWO half batches ("lanes") on two HIP streams at once -- lane 0 on the
// caller's stream, lane 1 on a context-owned side stream forked from it and joined back before the call returns.  Every kernel of
// the path is row- or sequence-local with one arithmetic whatever the launch size (tests/test_mixed256_gpu.py), so the lanes produce
// exactly the bits of the whole batch; what changes is the schedule: the partial last round of one lane's persistent GEMM (1,600 tiles
// on 256 CUs = 6.25 rounds at the headline shape), the tails of its other kernels and the launch gaps of small batches are filled by
// the other lane's kernels instead of idling: -1.3 % of the step at B = 32, -3 % at 24, -5.5 % at 16 and 8, -9 % at 4
// (profiles/r04/lanes_notes.md; three and four lanes, unequal cuts and CU-masked streams measured slower).  A lane is a complete
// sub-problem: its own packed rows, row tables and buffers; the cut is the item boundary closest to half of the rows.
namespace {
constexpr size_t VV_LANE_MIN_ROWS = 1024;     // "lanes" auto: at ~600 packed rows the step is bound by the launch rate and a second stream of launches
                                              // costs ~1 % (profiles/r04/lanes_notes.md); from 1,200 rows on two lanes win at every size measured
struct Lane {
    int B = 0, b0 = 0;
    int n_seq = 0;                     // sequences the block kernels of this lane (or branch view) run on: 2 B, or B for one CFG branch
    size_t Rc = 0, R = 0, n_tab = 0, tail_rows = 0;
    double sum_sq = 0;                 // sum of len^2 over those sequences (attention flops)
    bool uniform = true, pending = false;
    int tail_row0 = 0, tp_o = 0, tp_f = 0;
    const int32_t* seq_len = nullptr;
    float* x = nullptr;
    const float *cat = nullptr, *cat_drop = nullptr;
    hipStream_t st = nullptr;
    char *xcat = nullptr, *h = nullptr, *h2 = nullptr, *h3 = nullptr, *qkv = nullptr, *att = nullptr, *ffm = nullptr;
    float *xres = nullptr, *pred = nullptr, *csq = nullptr, *csk = nullptr, *csq_rows = nullptr, *csk_rows = nullptr, *h2_tail = nullptr, *h3_tail = nullptr;
    int *kv_len = nullptr, *tab = nullptr;
    const int *row_start = nullptr, *row_src = nullptr, *row_pos = nullptr, *qkv_pos = nullptr;
};
}  // namespace

static int transformer_steps_impl(vv_ctx* c, int B, int N, const int32_t* seq_len, const int32_t* seq_len_host, float* x, const float* cat,
                                  const float* cat_drop, const float* rope_cos_q, const float* rope_sin_q, const float* rope_cos_k,
                                  const float* rope_sin_k, int step0, int n_steps, void* stream, void* ext_ws = nullptr,
                                  uint64_t ext_bytes = 0, uint64_t* ws_only = nullptr) {
    if (!c) return -22;
    if (!c->finalized || !c->modtab) return c->fail(-1, "vv_transformer_steps: weights/time grid not ready");
    if (B < 1 || N < 1 || (!ws_only && (!seq_len || !x || !cat || !cat_drop || !rope_cos_q || !rope_sin_q || !rope_cos_k || !rope_sin_k)))
        return c->fail(-22, "vv_transformer_steps: bad arguments");
    if (step0 < 0 || n_steps < 0 || step0 + n_steps > c->n_steps) return c->fail(-22, "vv_transformer_steps: steps [%d,%d) outside the time grid (%d)", step0, step0 + n_steps, c->n_steps);
    const vv_model_cfg& g = c->cfg;
    hipSetDevice(c->device);
    hipStream_t st = (hipStream_t)stream;
    const int D = g.dim, FF = D * g.ff_mult, M = g.n_mel, CD = M + g.text_dim, es = c->esz();
    const int KP = pad_to(M + CD, 64), MP = pad_to(M, 128);
    if ((size_t)2 * B * N > (size_t)1 << 30) return c->fail(-22, "batch too large");
    // Ragged rows are PACKED: sequence (branch, b) owns rows [row_start, +len_b) of every activation buffer, the conditional
    // branch first, so GEMMs / norms / convs touch sum(len) rows instead of B x N_max.  The host needs the row count for the
    // launch shapes: the caller hands the lengths over on the host as well (vv_transformer_steps_h: no synchronisation at all,
    // the call can be captured into a hipGraph), or they are read back once (4*B bytes, one stream synchronisation).  The row
    // tables themselves are built on the device.  x, cat and cat_drop keep their padded [B][N] layout (row_src maps).
    std::vector<int> hlen(B);
    if (seq_len_host) {
        std::copy(seq_len_host, seq_len_host + B, hlen.begin());
    } else {
        HIPCHK(c, hipMemcpyAsync(hlen.data(), seq_len, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    size_t Rc_all = 0;
    for (int b = 0; b < B; ++b) {
        if (hlen[b] < 1 || hlen[b] > N) return c->fail(-22, "vv_transformer_steps: seq_len[%d] = %d outside [1, %d]", b, hlen[b], N);
        Rc_all += hlen[b];
    }
    if (2 * Rc_all * 3 * (size_t)D * es >= ((size_t)1 << 31))
        return c->fail(-22, "vv_transformer_steps: %zu packed rows make a %zu-byte qkv buffer; kernels address it with 32-bit byte offsets "
                            "(< 2 GiB) -- synthesise fewer units per call", 2 * Rc_all, 2 * Rc_all * 3 * (size_t)D * es);
    // lane plan: option "lanes" 1 = one lane, 2 = two whenever B >= 2, 0 (default) = two when both halves still fill the persistent
    // GEMM for several rounds (>= VV_LANE_MIN_ROWS packed rows in all); the cut is the item boundary closest to half of the rows
    int n_lanes = 1, cuts[3] = {0, B, B};
    if (B >= 2 && (c->lanes == 2 || (c->lanes == 0 && c->dt == VV_DTYPE_BF16 && 2 * Rc_all >= (size_t)VV_LANE_MIN_ROWS))) {
        size_t acc = 0, best = (size_t)-1;
        for (int b = 1; b < B; ++b) {
            acc += hlen[b - 1];
            const size_t d = acc * 2 > Rc_all ? acc * 2 - Rc_all : Rc_all - acc * 2;
            if (d < best) { best = d; cuts[1] = b; }
        }
        n_lanes = 2; cuts[2] = B;
    }
    // One problem only (a single item, or a batch under the row threshold with "lanes" 2): its two CFG BRANCHES are the lanes -- the
    // conditional rows [0, Rc) and the unconditional rows [Rc, 2 Rc) of the same packed buffers are independent from the conditioning
    // pack at the top of a step to the CFG combine at its end, so the side stream is forked and joined once per step around them.
    const bool branch_lanes = n_lanes == 1 && !(c->split_k_tail && c->dt == VV_DTYPE_BF16) &&
                              (c->lanes == 2 || (c->lanes == 0 && c->dt == VV_DTYPE_BF16 && 2 * Rc_all >= (size_t)VV_LANE_MIN_ROWS));
    Lane lanes[2];
    const int S = c->n_steps;
    Need nd;
    for (int li = 0; li < n_lanes; ++li) {
        Lane& L = lanes[li];
        L.b0 = cuts[li]; L.B = cuts[li + 1] - cuts[li];
        for (int b = L.b0; b < L.b0 + L.B; ++b) { L.Rc += hlen[b]; L.sum_sq += 2.0 * hlen[b] * hlen[b]; L.uniform = L.uniform && hlen[b] == N; }
        L.R = 2 * L.Rc; L.n_seq = 2 * L.B;
        // split-K tails of the two N = D gate-store GEMMs (out-proj K = D, FF2 K = FF): same row0 (it depends on M and N only)
        if (c->split_k_tail && c->dt == VV_DTYPE_BF16) {
            int r0o = 0, r0f = 0;
            if (c->split_k_tail == 1) vvk_gemm_tail_plan((int)L.R, D, D, D, D, D, &r0o, &L.tp_o);          // (M, N, K, lda, ldw, ldc) of the launches below
            vvk_gemm_tail_plan((int)L.R, D, FF, FF, FF, D, &r0f, &L.tp_f);
            if (L.tp_o && L.tp_f && r0o != r0f) L.tp_o = L.tp_f = 0;   // cannot happen (row0 is a function of M, N and the CU count); refuse rather than mix
            L.tail_row0 = L.tp_o ? r0o : r0f;
        }
        L.tail_rows = (L.tp_o || L.tp_f) ? L.R - L.tail_row0 : 0;
        L.n_tab = 2 * (size_t)L.B + L.Rc + L.R;          // row_start[2B] | row_src[Rc] | row_pos[R]
        const size_t R = L.R;
        nd.add(es * R * KP); nd.add(es * R * D); nd.add(es * R * D); nd.add(es * R * D); nd.add(4 * R * D); nd.add(es * R * 3 * D); nd.add(es * R * D);
        nd.add(es * R * FF); nd.add(4 * R * MP); nd.add(4 * 2 * L.B); nd.add(4ull * N * 64); nd.add(4ull * N * 64); nd.add(4ull * L.n_tab);
        nd.add(4ull * R * 64); nd.add(4ull * R * 64);
        nd.add(4 * L.tail_rows * D * (L.tp_o > 1 ? L.tp_o : 0)); nd.add(4 * L.tail_rows * D * (L.tp_f > 1 ? L.tp_f : 0));
    }
    if (ws_only) { *ws_only = (uint64_t)align_up(nd.b, 256); return 0; }
    if (ext_ws) { if (int r = use_ws(c, ext_ws, (size_t)ext_bytes, nd.b)) return r; }
    else if (int r = ensure_ws(c, nd.b)) return r;
    for (int li = 1; li < (branch_lanes ? 2 : n_lanes); ++li)
        if (!c->side_stream[li - 1]) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->side_stream[li - 1], hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[li - 1], hipEventDisableTiming));
        }
    if ((n_lanes > 1 || branch_lanes) && !c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int li = 0; li < n_lanes; ++li) {
        Lane& L = lanes[li];
        const size_t R = L.R;
        L.seq_len = seq_len + L.b0; L.x = x + (size_t)L.b0 * N * M; L.cat = cat + (size_t)L.b0 * N * CD; L.cat_drop = cat_drop + (size_t)L.b0 * N * CD;
        L.st = li == 0 ? st : c->side_stream[li - 1];
        L.xcat = carve<char>(c, es * R * KP);
        L.h = carve<char>(c, es * R * D);
        L.h2 = carve<char>(c, es * R * D);
        L.h3 = carve<char>(c, es * R * D);
        L.xres = carve<float>(c, R * D);
        L.qkv = carve<char>(c, es * R * 3 * D);
        L.att = carve<char>(c, es * R * D);
        L.ffm = carve<char>(c, es * R * FF);
        L.pred = carve<float>(c, R * MP);
        L.kv_len = carve<int>(c, 2 * L.B);
        L.csq = carve<float>(c, (size_t)N * 64);
        L.csk = carve<float>(c, (size_t)N * 64);
        L.tab = carve<int>(c, L.n_tab);
        L.csq_rows = carve<float>(c, R * 64);          // compact rope tables gathered per packed row, once per call
        L.csk_rows = carve<float>(c, R * 64);
        L.h2_tail = carve<float>(c, L.tail_rows * D * (L.tp_o > 1 ? L.tp_o : 0));     // fp32 [parts][tail_rows][D] K parts of the tail rows' deltas
        L.h3_tail = carve<float>(c, L.tail_rows * D * (L.tp_f > 1 ? L.tp_f : 0));
        L.row_start = L.tab; L.row_src = L.tab + 2 * L.B; L.row_pos = L.row_src + L.Rc;
        L.qkv_pos = L.uniform ? nullptr : L.row_pos;   // every sequence N rows: position = packed row mod N, no table lookup
    }
    // bf16: the query side of the rope moves from the QKV GEMM's epilogue into the attention kernel's Q load (option "rope_q_attn",
    // default on; profiles/r04/attention_notes.md); the fp32 (numerics) path keeps all of it in the GEMM
    const float rope_theta = c->dt == VV_DTYPE_BF16 ? c->rope_theta : 0.f;        // computed rope: both q and k in the GEMM epilogue, scale in attention
    const int q_rope_attn = (c->rope_q_attn && c->dt == VV_DTYPE_BF16 && !(rope_theta > 0.f)) ? 1 : 0;

    auto setup = [&](Lane& L) -> int {
        hipStream_t st = L.st;
        KCHK(c, vvk_row_tables(L.seq_len, L.B, N, (int)L.Rc, L.tab, L.tab + 2 * L.B, L.tab + 2 * L.B + L.Rc, st, &m__));
        KCHK(c, vvk_rope_compact(rope_cos_q, rope_sin_q, L.csq, N, st, &m__));
        KCHK(c, vvk_rope_compact(rope_cos_k, rope_sin_k, L.csk, N, st, &m__));
        if (c->rope_rows) {
            KCHK(c, vvk_rope_rows(L.csq, L.row_pos, L.csq_rows, (int)L.R, st, &m__));
            KCHK(c, vvk_rope_rows(L.csk, L.row_pos, L.csk_rows, (int)L.R, st, &m__));
        }
        KCHK(c, vvk_dup_len(L.seq_len, L.kv_len, L.B, st, &m__));
        Prof p(c, VV_PROF_ELEMWISE, 0, 4.0 * L.R * (M + CD) / 2 + (double)es * L.R * KP, st);
        KCHK(c, vvk_pack_cat(c->dt, L.x, L.cat, L.cat_drop, L.xcat, KP, (int)L.Rc, M, CD, 0, L.row_src, st, &m__));
        return 0;
    };
    // input embedding of step s: proj, then conv position embedding (two grouped convs + Mish) + residual
    auto step_pack = [&](Lane& L, int s) -> int {
        hipStream_t st = L.st;
        const size_t Rc = L.Rc;
        if (s != step0) {
            Prof p(c, VV_PROF_ELEMWISE, 0, 4.0 * Rc * M + 2.0 * es * Rc * M, st);
            KCHK(c, vvk_pack_cat(c->dt, L.x, L.cat, L.cat_drop, L.xcat, KP, (int)Rc, M, CD, 1, L.row_src, st, &m__));
        }
        return 0;
    };
    auto step_head = [&](Lane& L, int s) -> int {
        (void)s;
        hipStream_t st = L.st;
        const size_t R = L.R;
        if (int r = gemm(c, c->dt, c->dt, VV_EPI_STORE, VV_ACT_NONE_, L.xcat, KP, "input.proj.weight", KP, "input.proj.bias", L.h, D, (int)R, D, KP, st,
                         nullptr, 0, nullptr, 0, 0, 2.0 * R * D * (M + CD))) return r;
        for (int j = 1; j <= 2; ++j) {
            vv_posconv_args a{};
            a.dtype = c->dt; a.out_dtype = (j == 1) ? c->dt : VV_DTYPE_F32;
            a.in = (j == 1) ? L.h : L.h2; a.ld_in = D;
            const std::string wn = "input.pos_conv" + std::to_string(j) + ".weight", bn = "input.pos_conv" + std::to_string(j) + ".bias";
            a.W = c->W(wn); a.bias = c->Wf(bn);
            a.out = (j == 1) ? (void*)L.h2 : (void*)L.xres; a.ld_out = D;
            a.resid = (j == 2) ? L.h : nullptr; a.ld_resid = D;
            a.n_seq = L.n_seq; a.seq_n = N; a.groups = g.pos_conv_groups; a.KW = g.pos_conv_k; a.B = L.n_seq; a.seq_len = L.kv_len; a.row_start = L.row_start;
            Prof p(c, VV_PROF_POSCONV, 2.0 * R * D * 64 * g.pos_conv_k, (double)es * R * D * 2 + (j == 2 ? 4.0 * R * D : 0), st);
            KCHK(c, vvk_posconv(&a, st, &m__));
        }
        L.pending = false;
        return 0;
    };
    // Residual stream protocol: a branch GEMM writes delta = gate * (out + bias) (operand dtype) and a LayerNorm kernel adds
    // it while it streams x anyway (no read-modify-write in a GEMM epilogue).  x is REWRITTEN ONCE PER BLOCK: the norm
    // before the MLP normalises x + d_attn without storing it (keep_x), the next block's first norm stores
    // (x + d_attn) + d_mlp -- the same additions in the same order, a third fewer bytes written by the norm kernels.
    auto block = [&](Lane& L, int s, int l) -> int {
        hipStream_t st = L.st;
        const size_t R = L.R;
        const bool pending = L.pending;
        const float* rope[6] = {rope_cos_q, rope_sin_q, rope_cos_k, rope_sin_k, c->rope_rows ? L.csq_rows : L.csq, c->rope_rows ? L.csk_rows : L.csk};
        const float* mod = c->modtab + ((size_t)l * S + s) * 6 * D;
        const std::string qkvw = blk(l, ".attn.qkv.weight"), qkvb = blk(l, ".attn.qkv.bias"), ow = blk(l, ".attn.out.weight"),
                          ob = blk(l, ".attn.out.bias"), f1w = blk(l, ".ff1.weight"), f1b = blk(l, ".ff1.bias"),
                          f2w = blk(l, ".ff2.weight"), f2b = blk(l, ".ff2.bias");
        vv_ln_args a{}; a.out_dtype = c->dt; a.x = L.xres; a.ldx = D; a.y = L.h; a.ldy = D; a.R = (int)R; a.D = D; a.add_one = 1; a.eps = 1e-6f;
        a.delta_dtype = c->dt; a.ld_delta = D;
        a.tail_row0 = L.tail_row0; a.delta_tail = L.h2_tail; a.delta2_tail = L.h3_tail;
        a.delta_tail_parts = pending ? L.tp_o : 0; a.delta2_tail_parts = pending ? L.tp_f : 0;
        a.w = mod + D; a.b = mod;                       // scale_msa, shift_msa
        a.delta = pending ? L.h2 : nullptr; a.delta2 = pending ? L.h3 : nullptr; a.keep_x = 0;
        { Prof p(c, VV_PROF_NORM, 0, (4.0 + es) * R * D + (pending ? (4.0 + 2.0 * es) * R * D : 0), st); KCHK(c, vvk_ln_mod(&a, st, &m__)); }
        if (int r = gemm(c, c->dt, c->dt, VV_EPI_QKV_ROPE, VV_ACT_NONE_, L.h, D, qkvw.c_str(), D, qkvb.c_str(), L.qkv, 3 * D, (int)R, 3 * D, D, st, nullptr, 0, rope, N, D, -1, L.qkv_pos, c->rope_rows,
                         nullptr, 0, 0, q_rope_attn, rope_theta)) return r;
        {
            vv_attn_args t{}; t.dtype = c->dt; t.qkv = L.qkv; t.ld_qkv = 3 * D; t.out = L.att; t.ld_out = D; t.n_seq = L.n_seq; t.seq_n = N;
            t.heads = g.heads; t.dim = D; t.kv_len = L.kv_len; t.row_start = L.row_start; t.total_rows = (int)R;
            t.rope_cs_q = q_rope_attn ? L.csq : nullptr;
            t.q_scale = rope_theta > 0.f ? 1.0f / sqrtf((float)g.head_dim) : 0.f;
            Prof p(c, VV_PROF_ATTN, 4.0 * g.heads * L.sum_sq * 64, (double)es * R * 4 * D, st);
            KCHK(c, vvk_attention(&t, st, &m__));
        }
        if (int r = gemm(c, c->dt, c->dt, VV_EPI_GATE_STORE, VV_ACT_NONE_, L.att, D, ow.c_str(), D, ob.c_str(), L.h2, D, (int)R, D, D, st, mod + 2 * D,
                         0, nullptr, 0, 0, -1, nullptr, 0, L.tp_o ? L.h2_tail : nullptr, L.tp_o ? L.tail_row0 : 0, L.tp_o)) return r;
        a.w = mod + 4 * D; a.b = mod + 3 * D;           // scale_mlp, shift_mlp
        a.delta = L.h2; a.delta2 = nullptr; a.keep_x = 1; a.delta_tail_parts = L.tp_o; a.delta2_tail_parts = 0;
        { Prof p(c, VV_PROF_NORM, 0, (4.0 + 2.0 * es) * R * D, st); KCHK(c, vvk_ln_mod(&a, st, &m__)); }
        if (int r = gemm(c, c->dt, c->dt, VV_EPI_STORE, VV_ACT_GELU_TANH_, L.h, D, f1w.c_str(), D, f1b.c_str(), L.ffm, FF, (int)R, FF, D, st)) return r;
        if (int r = gemm(c, c->dt, c->dt, VV_EPI_GATE_STORE, VV_ACT_NONE_, L.ffm, FF, f2w.c_str(), FF, f2b.c_str(), L.h3, D, (int)R, D, FF, st, mod + 5 * D,
                         0, nullptr, 0, 0, -1, nullptr, 0, L.tp_f ? L.h3_tail : nullptr, L.tp_f ? L.tail_row0 : 0, L.tp_f)) return r;
        L.pending = true;
        return 0;
    };
    auto step_tail = [&](Lane& L, int s) -> int {
        hipStream_t st = L.st;
        const size_t R = L.R;
        const bool pending = L.pending;
        {
            const float* fm = c->fintab + (size_t)s * 2 * D;
            vv_ln_args a{}; a.out_dtype = c->dt; a.x = L.xres; a.ldx = D; a.y = L.h; a.ldy = D; a.R = (int)R; a.D = D; a.add_one = 1; a.eps = 1e-6f;
            a.w = fm; a.b = fm + D;                          // scale, shift
            a.delta = pending ? L.h2 : nullptr; a.delta2 = pending ? L.h3 : nullptr; a.keep_x = 1;     // x is re-initialised by the next step
            a.delta_dtype = c->dt; a.ld_delta = D;
            a.tail_row0 = L.tail_row0; a.delta_tail = L.h2_tail; a.delta2_tail = L.h3_tail;
            a.delta_tail_parts = pending ? L.tp_o : 0; a.delta2_tail_parts = pending ? L.tp_f : 0;
            Prof p(c, VV_PROF_NORM, 0, es * R * D + (pending ? (4.0 + 2.0 * es) * R * D : 4.0 * R * D), st);
            KCHK(c, vvk_ln_mod(&a, st, &m__));
        }
        if (int r = gemm(c, c->dt, VV_DTYPE_F32, VV_EPI_STORE, VV_ACT_NONE_, L.h, D, "final.proj.weight", D, "final.proj.bias", L.pred, MP, (int)R, MP, D, st,
                         nullptr, M, nullptr, 0, 0, 2.0 * R * D * M)) return r;
        return 0;
    };
    auto step_euler = [&](Lane& L, int s) -> int {
        hipStream_t st = L.st;
        Prof p(c, VV_PROF_ELEMWISE, 0, 16.0 * L.Rc * M, st);
        KCHK(c, vvk_cfg_euler(L.x, L.pred, MP, (int)L.Rc, M, g.cfg_strength, c->dt_host[s], L.row_src, st, &m__));
        return 0;
    };

    // Launch order: the lanes alternate block by block, so both streams always hold work and neither lane's enqueue waits for the
    // other's queue to drain; the device orders each stream by itself.  Item lanes: lane 1 forks from the caller's stream once (what
    // the caller enqueued before this call is visible to it) and joins back at the end (what the caller enqueues next sees both
    // lanes).  Branch lanes: the same fork / join once per STEP, between the conditioning pack and the CFG combine.
    Lane views[2];                                        // what the block-level kernels run on
    int n_views = n_lanes;
    if (branch_lanes) {
        const Lane& L = lanes[0];
        n_views = 2;
        for (int v = 0; v < 2; ++v) {
            Lane& V = views[v];
            V = L;
            const size_t r0 = v ? L.Rc : 0;             // the unconditional branch's rows follow the conditional branch's
            V.st = v ? c->side_stream[0] : st;
            V.R = L.Rc; V.n_seq = L.B; V.sum_sq = L.sum_sq / 2;
            V.xcat += r0 * KP * es; V.h += r0 * D * es; V.h2 += r0 * D * es; V.h3 += r0 * D * es; V.att += r0 * D * es;
            V.qkv += r0 * 3 * D * es; V.ffm += r0 * FF * es; V.xres += r0 * D; V.pred += r0 * MP;
            V.csq_rows += r0 * 64; V.csk_rows += r0 * 64;
            // row_start / kv_len: the conditional branch's entries serve both views (same lengths, rows relative to the view's base);
            // row_pos: the first Rc entries (a row's position does not depend on its branch)
        }
    } else {
        for (int li = 0; li < n_lanes; ++li) views[li] = lanes[li];
    }
    auto fork = [&]() -> int {
        HIPCHK(c, hipEventRecord(c->ev_fork, st));
        HIPCHK(c, hipStreamWaitEvent(c->side_stream[0], c->ev_fork, 0));
        return 0;
    };
    auto join = [&]() {                                   // always reached, also after a failed launch: the side stream must not outlive the call
        hipEventRecord(c->ev_join[0], c->side_stream[0]);
        hipStreamWaitEvent(st, c->ev_join[0], 0);
    };
    int rc = 0;
    c->chip_share = n_views;
    if (n_lanes > 1) rc = fork();
    for (int li = 0; li < n_lanes && !rc; ++li) rc = setup(lanes[li]);
    for (int s = step0; s < step0 + n_steps && !rc; ++s) {
        for (int li = 0; li < n_lanes && !rc; ++li) rc = step_pack(lanes[li], s);
        if (branch_lanes && !rc) rc = fork();
        const bool forked = branch_lanes && !rc;
        for (int v = 0; v < n_views && !rc; ++v) rc = step_head(views[v], s);
        for (int l = 0; l < g.depth && !rc; ++l)
            for (int v = 0; v < n_views && !rc; ++v) rc = block(views[v], s, l);
        for (int v = 0; v < n_views && !rc; ++v) rc = step_tail(views[v], s);
        if (forked) join();
        for (int li = 0; li < n_lanes && !rc; ++li) rc = step_euler(lanes[li], s);
    }
    if (n_lanes > 1) join();
    c->chip_share = 1;
    return rc;
}

int vv_transformer_steps(vv_ctx* c, int B, int N, const int32_t* seq_len, float* x, const float* cat, const float* cat_drop,
                         const float* rope_cos_q, const float* rope_sin_q, const float* rope_cos_k, const float* rope_sin_k,
                         int step0, int n_steps, void* stream) {
    return transformer_steps_impl(c, B, N, seq_len, nullptr, x, cat, cat_drop, rope_cos_q, rope_sin_q, rope_cos_k, rope_sin_k, step0, n_steps, stream);
}

// Same, with the per-item lengths ALSO given on the host (they must equal the device array): no read-back, no stream
// synchronisation anywhere in the call -- with a workspace that is already large enough it can be captured into a hipGraph.
int vv_transformer_steps_h(vv_ctx* c, int B, int N, const int32_t* seq_len, const int32_t* seq_len_host, float* x, const float* cat,
                           const float* cat_drop, const float* rope_cos_q, const float* rope_sin_q, const float* rope_cos_k,
                           const float* rope_sin_k, int step0, int n_steps, void* stream) {
    if (c && !seq_len_host) return c->fail(-22, "vv_transformer_steps_h: host lengths missing");
    return transformer_steps_impl(c, B, N, seq_len, seq_len_host, x, cat, cat_drop, rope_cos_q, rope_sin_q, rope_cos_k, rope_sin_k, step0, n_steps, stream);
}

// The same call with every intermediate carved from a CALLER-OWNED block (>= vv_transformer_ws_bytes for the same B, N and host
// lengths, 256-byte aligned): no allocation and no synchronisation anywhere in the call, and nothing it points at can move -- the
// form to capture into a hipGraph (all Euler steps of an utterance + vv_decode_into = one graph launch).
int vv_transformer_ws_bytes(vv_ctx* c, int B, int N, const int32_t* seq_len_host, uint64_t* bytes) {
    if (c && (!seq_len_host || !bytes)) return c->fail(-22, "vv_transformer_ws_bytes: bad arguments");
    return transformer_steps_impl(c, B, N, nullptr, seq_len_host, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, bytes);
}

int vv_transformer_steps_into(vv_ctx* c, int B, int N, const int32_t* seq_len, const int32_t* seq_len_host, float* x, const float* cat,
                              const float* cat_drop, const float* rope_cos_q, const float* rope_sin_q, const float* rope_cos_k,
                              const float* rope_sin_k, int step0, int n_steps, void* ws, uint64_t ws_bytes, void* stream) {
    if (c && (!seq_len_host || !ws)) return c->fail(-22, "vv_transformer_steps_into: host lengths and a workspace block are required");
    return transformer_steps_impl(c, B, N, seq_len, seq_len_host, x, cat, cat_drop, rope_cos_q, rope_sin_q, rope_cos_k, rope_sin_k, step0, n_steps, stream, ws, ws_bytes);
}

// --------------------------------------------------------------------------------------- decode
static size_t decode_need(const vv_ctx* c, int B, int t_gen_max) {
    const vv_model_cfg& g = c->cfg;
    const int nu = g.voc_n_up;
    size_t big = (size_t)g.voc_pre_ch * t_gen_max, T = t_gen_max;
    int ch = g.voc_pre_ch;
    for (int s = 0; s < nu; ++s) { T *= g.voc_up_rates[s]; ch /= 2; big = std::max(big, (size_t)ch * T); }
    Need nd; nd.add(4ull * B * g.n_mel * t_gen_max);
    for (int i = 0; i < 5; ++i) nd.add(4ull * B * big);
    nd.add(4ull * (nu + 1) * B);
    return nd.b;
}

int vv_decode_ws_bytes(vv_ctx* c, int B, int t_gen_max, uint64_t* bytes) {
    if (!c || !bytes || B < 1 || t_gen_max < 1) return c ? c->fail(-22, "vv_decode_ws_bytes: bad arguments") : -22;
    *bytes = (uint64_t)align_up(decode_need(c, B, t_gen_max), 256);
    return 0;
}

uint64_t vv_ws_generation(const vv_ctx* c) { return c ? c->ws_generation : 0; }

static int decode_impl(vv_ctx* c, int B, int N, const float* x, const int32_t* ref_len, const int32_t* seq_len, int t_gen_max, int16_t* pcm,
                       int ld_pcm, int32_t* pcm_len, float* wave_f32, void* ext_ws, uint64_t ext_bytes, void* stream) {
    if (!c) return -22;
    if (!c->finalized) return c->fail(-1, "vv_decode: weights not finalized");
    if (B < 1 || N < 1 || !x || !ref_len || !seq_len || !pcm || !pcm_len || t_gen_max < 1 || t_gen_max > N)
        return c->fail(-22, "vv_decode: bad arguments");
    const vv_model_cfg& g = c->cfg;
    if (ld_pcm < t_gen_max * g.hop_length) return c->fail(-22, "vv_decode: ld_pcm too small");
    hipSetDevice(c->device);
    hipStream_t st = (hipStream_t)stream;
    const int M = g.n_mel, nu = g.voc_n_up;
    // buffer plan
    std::vector<int> Ts(nu + 1), Cs(nu + 1);
    Ts[0] = t_gen_max; Cs[0] = g.voc_pre_ch;
    size_t big = 0;
    for (int s = 0; s < nu; ++s) { Ts[s + 1] = Ts[s] * g.voc_up_rates[s]; Cs[s + 1] = Cs[s] / 2; big = std::max(big, (size_t)Cs[s + 1] * Ts[s + 1]); }
    big = std::max(big, (size_t)Cs[0] * Ts[0]);
    const size_t need = decode_need(c, B, t_gen_max);
    if (ext_ws) { if (int r = use_ws(c, ext_ws, (size_t)ext_bytes, need)) return r; }
    else if (int r = ensure_ws(c, need)) return r;
    float* v0 = carve<float>(c, (size_t)B * M * Ts[0]);
    float* buf[5];
    for (int i = 0; i < 5; ++i) buf[i] = carve<float>(c, (size_t)B * big);
    int* lens = carve<int>(c, (size_t)(nu + 1) * B);

    KCHK(c, vvk_decode_len(seq_len, ref_len, lens, B, nu + 1, c->d_mult, st, &m__));
    HIPCHK(c, hipMemcpyAsync(pcm_len, lens + (size_t)nu * B, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
    {
        Prof p(c, VV_PROF_ELEMWISE, 0, 8.0 * B * M * Ts[0], st);
        KCHK(c, vvk_mel_slice(x, B, N, M, ref_len, seq_len, v0, Ts[0], st, &m__));
    }
    const bool x3 = c->voc_x3 == 1 || (c->voc_x3 < 0 && c->dt == VV_DTYPE_BF16);
    auto conv = [&](const float* in, const std::string& name, float* out, const float* resid, int Cin, int Cout, int T_in, int T_out, int KW,
                    int dil, int up, float pre_slope, float scale, int accumulate, const int* len_in, int stage_cls) -> int {
        vv_conv_args a{};
        a.in = in; a.W = c->Wf(name + ".weight"); a.bias = c->Wf(name + ".bias"); a.out = out; a.resid = resid;
        if (x3) {
            auto it = c->x3_w.find(name + ".weight");
            if (it == c->x3_w.end()) return c->fail(-2, "split weights of %s missing", name.c_str());
            a.W_x3 = it->second; a.wg_rows = c->voc_x3_rows;
        }
        a.B = B; a.Cin = Cin; a.Cout = Cout; a.T_in = T_in; a.T_out = T_out; a.KW = KW; a.dil = dil; a.transposed = up > 0; a.up = up;
        a.rows_total = up > 0 ? Cout * up : Cout; a.rows_pad = pad_to(a.rows_total, 64); a.accumulate = accumulate;
        a.pre_slope = pre_slope; a.out_scale = scale; a.len_in = len_in;
        const double taps = up > 0 ? 2.0 : (double)KW;
        Prof p(c, VV_PROF_VOC_CONV, 2.0 * B * (double)T_out * Cout * Cin * taps,
               4.0 * B * ((double)Cin * T_in + (double)Cout * T_out * (1 + (resid ? 1 : 0) + (accumulate ? 1 : 0))), st, stage_cls);
        const char* m = "";
        int r = vvk_conv(&a, st, &m);
        if (r) return c->fail(r, "%s (%s)", m, name.c_str());
        return 0;
    };
    if (int r = conv(v0, "voc.pre", buf[0], nullptr, M, Cs[0], Ts[0], Ts[0], g.voc_pre_k, 1, 0, 1.0f, 1.0f, 0, lens, VV_PROF_VOC_PRE)) return r;
    float* cur = buf[0];
    float* up_out = buf[1];
    for (int s = 0; s < nu; ++s) {
        const int C = Cs[s + 1], T = Ts[s + 1];
        const int* len_s = lens + (size_t)(s + 1) * B;
        if (int r = conv(cur, "voc.up." + std::to_string(s), up_out, nullptr, Cs[s], C, Ts[s], T, 2, 1, g.voc_up_rates[s], g.voc_lrelu, 1.0f, 0, lens + (size_t)s * B,
                         s < 4 ? VV_PROF_VOC_UP0 + s : -1)) return r;
        // MRF: acc = (1/n_res) * sum_a resblock_a(up_out)
        float* acc = cur;                                   // previous stage input is dead now
        float* t1 = (up_out == buf[1]) ? buf[2] : buf[1];
        float* ya = buf[3];
        float* yb = buf[4];
        const float inv = 1.0f / (float)g.voc_n_res;
        const int mrf_cls = s < 4 ? VV_PROF_VOC_MRF0 + s : -1;
        for (int a = 0; a < g.voc_n_res; ++a) {
            const float* y = up_out;
            for (int b = 0; b < g.voc_n_dil; ++b) {
                const std::string q = "voc.res." + std::to_string(s) + "." + std::to_string(a) + "." + std::to_string(b);
                const bool last = b == g.voc_n_dil - 1;
                float* dst = last ? acc : ((y == ya) ? yb : ya);
                const int kw = g.voc_res_kernels[a], dil = g.voc_res_dilations[b];
                // (the fused pair exists on the f32 instruction only: with x3 products the two x3 launches are taken unless fusion is forced)
                if ((c->fuse_mrf == 1 || (c->fuse_mrf == 2 && B <= 8 && !x3)) && (C == 32 || C == 64) && (kw == 3 || kw == 7 || kw == 11)) {
                    // K12 fused through LDS: the intermediate of the pair never reaches HBM (bit-identical to the two launches below)
                    vv_mrf_args m{};
                    m.y = y; m.W1 = c->Wf(q + ".conv1.weight"); m.b1 = c->Wf(q + ".conv1.bias"); m.W2 = c->Wf(q + ".conv2.weight"); m.b2 = c->Wf(q + ".conv2.bias");
                    m.out = dst; m.B = B; m.C = C; m.T = T; m.KW = kw; m.dil = dil; m.rows_pad = 64; m.accumulate = last && a > 0;
                    m.slope = g.voc_lrelu; m.out_scale = last ? inv : 1.0f; m.len_in = len_s;
                    Prof p(c, VV_PROF_VOC_CONV, 2.0 * 2.0 * B * (double)T * C * C * kw, 4.0 * B * (double)C * T * (2 + (m.accumulate ? 1 : 0)), st, mrf_cls);
                    const char* em = "";
                    if (int r = vvk_mrf_pair(&m, st, &em)) return c->fail(r, "%s (%s)", em, q.c_str());
                } else {
                    if (int r = conv(y, q + ".conv1", t1, nullptr, C, C, T, T, kw, dil, 0, g.voc_lrelu, 1.0f, 0, len_s, mrf_cls)) return r;
                    if (int r = conv(t1, q + ".conv2", dst, y, C, C, T, T, kw, 1, 0, g.voc_lrelu, last ? inv : 1.0f, last && a > 0, len_s, mrf_cls)) return r;
                }
                y = dst;
            }
        }
        // rotate: acc becomes the next stage input; up_out buffer is free
        float* old_up = up_out;
        cur = acc;
        up_out = old_up;
        // make sure next up_out differs from cur (it does: acc was the previous cur)
    }
    {
        const int C = Cs[nu], T = Ts[nu];
        Prof p(c, VV_PROF_VOC_POST, 2.0 * B * (double)T * C * g.voc_post_k, 4.0 * B * (double)C * T + 2.0 * B * T, st);
        KCHK(c, vvk_conv_post(cur, c->Wf("voc.post.weight"), c->post_bias, pcm, ld_pcm, wave_f32, B, C, T, g.voc_post_k, 0.01f,
                              lens + (size_t)nu * B, st, &m__));
    }
    return 0;
}

int vv_set_rope_theta(vv_ctx* c, float theta) {
    if (!c) return -22;
    if (theta != 0.f && !(theta > 1.f)) return c->fail(-22, "vv_set_rope_theta: the base must be > 1 (0 = read the tables)");
    c->rope_theta = theta;
    return 0;
}

int vv_set_option(vv_ctx* c, const char* name, int value) {
    if (!c || !name) return -22;
    if (!strcmp(name, "rope_rows")) { c->rope_rows = value != 0; return 0; }
    if (!strcmp(name, "rope_q_attn")) { c->rope_q_attn = value != 0; return 0; }
    if (!strcmp(name, "voc_x3_rows")) {
        if (value != 0 && value != 128) return c->fail(-22, "vv_set_option: voc_x3_rows takes 0 (64-row workgroups) or 128");
        c->voc_x3_rows = value; return 0;
    }
    if (!strcmp(name, "voc_x3")) {
        if (value < -1 || value > 1) return c->fail(-22, "vv_set_option: voc_x3 takes -1 (by acoustic dtype), 0 (f32 MFMA) or 1 (3-way bf16 split)");
        c->voc_x3 = value; return 0;
    }
    if (!strcmp(name, "split_k_tail")) {
        if (value < 0 || value > 2) return c->fail(-22, "vv_set_option: split_k_tail takes 0 (off), 1 (out-projection and FF2) or 2 (FF2 only)");
        c->split_k_tail = value; return 0;
    }
    if (!strcmp(name, "ring_tiles")) { c->ring_tiles = value != 0; if (value > 1) c->ring_tiles_max = value; return 0; }
    if (!strcmp(name, "pp_min_tiles")) {
        if (value < -1) return c->fail(-22, "vv_set_option: pp_min_tiles takes -1 (the launcher's rule) or a 256-tile count >= 0");
        c->pp_min_tiles = value; return 0;
    }
    if (!strcmp(name, "lanes")) {
        if (value < 0 || value > 2) return c->fail(-22, "vv_set_option: lanes takes 0 (auto), 1 (one lane) or 2 (two lanes whenever the batch has two items)");
        c->lanes = value; return 0;
    }
    if (!strcmp(name, "fuse_mrf")) {
        if (value < 0 || value > 2) return c->fail(-22, "vv_set_option: fuse_mrf takes 0 (off), 1 (on) or 2 (auto)");
        c->fuse_mrf = value; return 0;
    }
    return c->fail(-22, "vv_set_option: unknown option '%s'", name);
}

int vv_decode(vv_ctx* c, int B, int N, const float* x, const int32_t* ref_len, const int32_t* seq_len, int t_gen_max, int16_t* pcm,
              int ld_pcm, int32_t* pcm_len, float* wave_f32, void* stream) {
    return decode_impl(c, B, N, x, ref_len, seq_len, t_gen_max, pcm, ld_pcm, pcm_len, wave_f32, nullptr, 0, stream);
}

// Same stage with every intermediate in a CALLER-OWNED block of vv_decode_ws_bytes() bytes: nothing the launches point at can
// be moved by a later call that grows the context arena, so the launch sequence may be captured into a hipGraph and replayed
// for the lifetime of that block (reference has no counterpart; BASELINE.json configs[4] "hipGraph-captured vocoder step").
int vv_decode_into(vv_ctx* c, int B, int N, const float* x, const int32_t* ref_len, const int32_t* seq_len, int t_gen_max, int16_t* pcm,
                   int ld_pcm, int32_t* pcm_len, float* wave_f32, void* ws, uint64_t ws_bytes, void* stream) {
    if (!c) return -22;
    if (!ws) return c->fail(-22, "vv_decode_into: null workspace block");
    return decode_impl(c, B, N, x, ref_len, seq_len, t_gen_max, pcm, ld_pcm, pcm_len, wave_f32, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------ profiling
int vv_prof_enable(vv_ctx* c, int on) {
    if (!c) return -22;
    c->prof = on != 0;
    return 0;
}
int vv_prof_collect(vv_ctx* c, int64_t* launches, double* ms, double* flops, double* bytes) {
    if (!c) return -22;
    hipSetDevice(c->device);
    HIPCHK(c, hipDeviceSynchronize());
    for (auto& r : c->recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { c->p_ms[r.cls] += t; if (r.sub >= 0) c->p_ms[r.sub] += t; }
        c->pool.push_back(r.a); c->pool.push_back(r.b);
    }
    c->recs.clear();
    for (int i = 0; i < VV_PROF_NCLASS; ++i) {
        if (launches) launches[i] = c->p_launch[i];
        if (ms) ms[i] = c->p_ms[i];
        if (flops) flops[i] = c->p_flops[i];
        if (bytes) bytes[i] = c->p_bytes[i];
        c->p_launch[i] = 0; c->p_ms[i] = 0; c->p_flops[i] = 0; c->p_bytes[i] = 0;
    }
    return 0;
}

// ----------------------------------------------------------------------- single-kernel entries
#define SINGLE(ctx, call)                                     \
    do {                                                      \
        if (!(ctx)) return -22;                               \
        hipSetDevice((ctx)->device);                          \
        const char* m__ = "";                                 \
        int r__ = (call);                                     \
        if (r__) return (ctx)->fail(r__, "%s", m__);          \
        return 0;                                             \
    } while (0)

int vv_gemm(vv_ctx* c, const vv_gemm_args* a, void* st) { SINGLE(c, vvk_gemm(a, (hipStream_t)st, &m__)); }
int vv_gemm_tail_plan(vv_ctx* c, int32_t M, int32_t N, int32_t K, int32_t* row0, int32_t* parts) {
    // ctx may be NULL: the plan for the process's current device (256 CUs when there is none), so that planner / launcher
    // consistency can be checked without a GPU.  Operands are taken as contiguous (lda = ldw = K, ldc = N).
    if (!row0 || !parts || M < 1 || N < 1 || K < 1) return c ? c->fail(-22, "vv_gemm_tail_plan: bad arguments") : -22;
    if (c) HIPCHK(c, hipSetDevice(c->device));
    vvk_gemm_tail_plan(M, N, K, K, K, N, row0, parts);
    return 0;
}
int vv_attention(vv_ctx* c, const vv_attn_args* a, void* st) { SINGLE(c, vvk_attention(a, (hipStream_t)st, &m__)); }
int vv_layernorm(vv_ctx* c, const vv_ln_args* a, void* st) { SINGLE(c, vvk_ln_mod(a, (hipStream_t)st, &m__)); }
int vv_posconv(vv_ctx* c, const vv_posconv_args* a, void* st) { SINGLE(c, vvk_posconv(a, (hipStream_t)st, &m__)); }
int vv_conv1d(vv_ctx* c, const vv_conv_args* a, void* st) { SINGLE(c, vvk_conv(a, (hipStream_t)st, &m__)); }
uint64_t vv_conv_split_bytes(int32_t Cin_pad, int32_t KW, int32_t rows_pad) {
    return (Cin_pad < 1 || KW < 1 || rows_pad < 1) ? 0 : (uint64_t)vvk_conv_split_bytes(Cin_pad, KW, rows_pad);
}
int vv_conv_split_weights(vv_ctx* c, const float* W, int32_t Cin_pad, int32_t KW, int32_t rows_pad, void* out, void* st) {
    SINGLE(c, vvk_conv_split_weights(W, Cin_pad, KW, rows_pad, out, (hipStream_t)st, &m__));
}
int vv_mrf_resblock(vv_ctx* c, const vv_mrf_args* a, void* st) { SINGLE(c, vvk_mrf_pair(a, (hipStream_t)st, &m__)); }
int vv_conv_post(vv_ctx* c, const float* in, const float* w, float bias, int16_t* pcm, int ld_pcm, float* wave_f32, int B, int C, int T,
                 int KW, float pre_slope, const int32_t* len_in, void* st) {
    SINGLE(c, vvk_conv_post(in, w, bias, pcm, ld_pcm, wave_f32, B, C, T, KW, pre_slope, len_in, (hipStream_t)st, &m__));
}
int vv_mel(vv_ctx* c, const int16_t* audio, int ld_audio, const int32_t* audio_len, float* mel, int B, int F_max, void* st) {
    if (!c) return -22;
    if (!c->W("const.window") || !c->W("const.tw_cos") || !c->W("const.tw_sin") || !c->W("const.mel_fb"))
        return c->fail(-2, "vv_mel: constant tables not bound");
    SINGLE(c, vvk_mel(audio, ld_audio, audio_len, c->Wf("const.window"), c->Wf("const.tw_cos"), c->Wf("const.tw_sin"), c->Wf("const.mel_fb"),
                      mel, B, F_max, c->cfg.n_fft, c->cfg.hop_length, c->cfg.n_mel, (hipStream_t)st, &m__));
}
int vv_groupnorm(vv_ctx* c, const float* x, float* y, const float* gamma, const float* beta, int B, int C, int T, int G, float eps, int act, void* st) {
    SINGLE(c, vvk_groupnorm(x, y, gamma, beta, B, C, T, G, eps, act, (hipStream_t)st, &m__));
}
int vv_rope_rows(vv_ctx* c, const float* compact, const int32_t* pos, float* out, int rows, void* st) {
    SINGLE(c, vvk_rope_rows(compact, pos, out, rows, (hipStream_t)st, &m__));
}
int vv_rope_compact(vv_ctx* c, const float* cos_t, const float* sin_t, float* out, int n, void* st) {
    SINGLE(c, vvk_rope_compact(cos_t, sin_t, out, n, (hipStream_t)st, &m__));
}
int vv_resample_poly(vv_ctx* c, const float* x, int n_in, const double* taps, int n_taps, int up, int down, int skip, float* y, int n_out, void* st) {
    SINGLE(c, vvk_resample_poly(x, n_in, taps, n_taps, up, down, skip, y, n_out, (hipStream_t)st, &m__));
}
int vv_ingest_pcm(vv_ctx* c, const void* pcm, const int64_t* desc, int n_clips, int64_t max_out, float* out, void* st) {
    SINGLE(c, vvk_ingest_pcm(pcm, (const long long*)desc, n_clips, (long long)max_out, out, (hipStream_t)st, &m__));
}
size_t vv_normalize_scratch_bytes(int n_clips, int64_t total_len) { return vvk_normalize_scratch_bytes(n_clips, (long long)total_len); }
int vv_normalize_clips(vv_ctx* c, const float* x, const int64_t* offsets, int n_clips, int64_t max_len, void* scratch, int16_t* out, void* st) {
    SINGLE(c, vvk_normalize_clips(x, (const long long*)offsets, n_clips, (long long)max_len, scratch, out, (hipStream_t)st, &m__));
}
int vv_cfg_euler(vv_ctx* c, float* x, const float* pred, int ldp, int BN, int n_mel, float cfg, float dt, void* st) {
    SINGLE(c, vvk_cfg_euler(x, pred, ldp, BN, n_mel, cfg, dt, nullptr, (hipStream_t)st, &m__));
}

}  // extern "C"
