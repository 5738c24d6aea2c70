// engine.hip -- model context + launch orchestration (host side of libvisualcla_hip.so).
//
// The three macro entry points issue every kernel of their stage from C++ on the caller's stream, so a
// forward / decode step costs ONE Python->C transition instead of ~200:
//   vcla_vision_forward     ViT-L/14 -> post-LN(all tokens) -> Resampler -> projection
//   vcla_llama_prefill      LLaMA decoder over the spliced prompt embeds, fills the KV cache
//   vcla_llama_decode_step  one greedy step against the cache (position / context length live in device memory
//                           so the identical launch sequence can be replayed from a hipGraph)
// Memory: nothing is allocated here.  Weights are caller-owned device tensors registered by name; activations
// live in a caller-provided workspace carved by a bump allocator; the KV cache is caller-owned.
#include "vcla_common.h"
#include "decode_engine.h"
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <unordered_map>
#include <vector>

// ------------------------------------------------------------------ errors / version
static thread_local char g_err[1024] = "";

void vcla_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int vcla_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char* vcla_last_error(void) { return g_err; }
extern "C" int vcla_version(void) { return VCLA_ABI_VERSION; }

extern "C" int vcla_device_check(void) {
    int dev = 0;
    VCLA_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    VCLA_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    if (std::string(p.gcnArchName).rfind("gfx950", 0) != 0)
        return vcla_fail(VCLA_ERR_UNSUPPORTED_ARCH, "device %d is %s; this library is built for gfx950 only", dev, p.gcnArchName);
    return VCLA_OK;
}

// ------------------------------------------------------------------ tiny helper kernels
template <typename T>
__global__ __launch_bounds__(256) void bcast_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t n_per) {
    // dst[b, :] = src[:] for b = blockIdx.y
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_per) dst[(int64_t)blockIdx.y * n_per + i] = src[i];
}

__global__ void advance_pos_kernel(int32_t* pos) { *pos += 1; }

// End of a decode-loop step, ONE launch instead of three (record_ids + advance_pos + the next step's embed_splice): workgroup b
//   ids_out[(*pos_dev - step_base) * B + b] = cur[b];   x[b, :] = embed[cur[b], :]  (the next step's decoder input);
// the last workgroup to finish advances *pos_dev (every workgroup has read it by then) and re-arms the ticket.
template <typename T>
__global__ __launch_bounds__(256) void post_select_kernel(const int64_t* __restrict__ cur, int64_t* __restrict__ ids_out, int32_t* __restrict__ pos_dev,
                                                          int step_base, int B, const bf16_t* __restrict__ table, T* __restrict__ x, int D, int V,
                                                          int* __restrict__ ticket) {
    const int b = blockIdx.x;
    int64_t id = cur[b];
    if (threadIdx.x == 0) ids_out[(int64_t)(*pos_dev - step_base) * B + b] = id;
    if (id < 0 || id >= V) id = 0;          // as embed_splice_kernel: stay in bounds
    const bf16_t* src = table + id * D;
    T* o = x + (int64_t)b * D;
    if constexpr (sizeof(T) == 2) {
        for (int c = threadIdx.x * 8; c < D; c += 256 * 8) *reinterpret_cast<uint4*>(o + c) = *reinterpret_cast<const uint4*>(src + c);   // D % 8 == 0 (checked by the caller)
    } else {
        for (int c = threadIdx.x; c < D; c += 256) Act<T>::st(o + c, bf2f(src[c]));
    }
    __syncthreads();                        // thread 0's read of *pos_dev is done
    if (threadIdx.x == 0 && atomicAdd(ticket, 1) == B - 1) {
        *pos_dev += 1;
        *ticket = 0;
    }
}

// ------------------------------------------------------------------ context
struct TensorRef {
    const void* ptr;
    size_t nbytes;
};

struct VitLayer {
    const float *ln1g, *ln1b, *bqkv, *bo, *ln2g, *ln2b, *b1, *b2;
    const void *wqkv, *wo, *w1, *w2;
};
struct ResLayer {
    const float *bq, *bkv, *bo, *ln1g, *ln1b, *b1, *b2, *ln2g, *ln2b;
    const void *wq, *wkv, *wo, *w1, *w2;
};
// optional alternative copies of one weight matrix for the M <= 128 (decode) kernels
struct WVar {
    const void* frag = nullptr;   // bf16 fragment-major twin                    ("<name>.f")
    const void* q8 = nullptr;     // fp8 e4m3fn row-major                        ("<name>.q8")
    const void* q8f = nullptr;    // fp8 fragment-pair-major                     ("<name>.q8f")
    const float* s8 = nullptr;    // per-row fp32 scale of the fp8 copies         ("<name>.s8")
};
struct LlamaLayer {
    const float *ln1g, *ln2g;
    const void *wqkv, *wo, *wgu, *wd;
    WVar vqkv, vo, vgu, vd;
};

struct vcla_ctx {
    vcla_model_cfg c;
    std::unordered_map<std::string, TensorRef> tensors;
    bool finalized = false;
    // resolved
    const void* vit_patch_w = nullptr;
    const float *vit_cls = nullptr, *vit_pos = nullptr, *vit_pre_g = nullptr, *vit_pre_b = nullptr, *vit_post_g = nullptr,
                *vit_post_b = nullptr;
    std::vector<VitLayer> vit;
    const void* res_query = nullptr;
    std::vector<ResLayer> res;
    const void* res_wkv_all = nullptr;     // optional: K/V projections of all resampler layers stacked [L * 2D, D]
    const float* res_bkv_all = nullptr;
    const void* proj_w = nullptr;
    const float* proj_b = nullptr;
    const void* embed = nullptr;
    std::vector<LlamaLayer> llama;
    const float* norm_g = nullptr;
    const void* lm_head = nullptr;
    WVar vlm;
    const float *rope_cos = nullptr, *rope_sin = nullptr;
    // the persistent B = 1 decode step (decode_engine.hip): its weight stream + norm gains, when registered and the geometry fits
    const void* eng_w = nullptr;
    const float* eng_g = nullptr;
    vcla_engine_geom eng_geom = {};
    int k_pad = 0;  // padded im2col width
    // per-call state of the macro entry points (one thread drives a context at a time): the workspace carve of the
    // running call and what the last streaming GEMM left behind.  Kept here, not in thread-local globals, so that nothing
    // leaks between contexts or into the public vcla_gemm entry point.
    struct {
        void* splitk_ws = nullptr;     // fp32 split-K scratch of the running entry point's workspace
        bool decode_step = false;      // inside decode_step_impl: the fp8 weight copies (if loaded) may be used
        void* q8_ws = nullptr;         // fp8 activation staging of the running prefill (t_fp8_mfma), else NULL
        float* q8s_ws = nullptr;
        int ssq_parts = 0;             // layout of the deferred-RMSNorm row statistics the last producer wrote (see gemm_ds)
    } run;
    // cached graphs of the vision stack and of the prefill (fixed launch sequences per shape: ~250 / ~290 launches that are host-bound
    // at B = 1).  Keyed on every pointer and shape the captured launches bake in; see run_macro.
    struct MacroGraph {
        static constexpr int kSlots = 2;        // two shapes in rotation (e.g. forward(all_logits) and generate()'s prefill, or two batch sizes) both stay cached
        hipGraphExec_t exec[kSlots] = {};
        const void* key[kSlots][8] = {};        // of the captured graphs
        unsigned long long used[kSlots] = {};   // last-use tick: the older slot is the one replaced
        unsigned long long tick = 0;
        const void* seen[8] = {};               // of the previous eager call
        int has_seen = 0;
    } vision_graph, prefill_graph;
    // cached decode graph
    hipGraphExec_t graph_exec = nullptr;        // one decode step
    hipGraphExec_t graph_exec_multi = nullptr;  // VCLA_GRAPH_STEPS decode steps (same key), built on the first loop long enough to use it
    struct {
        const void *ids, *kv, *mask, *ws, *out;
        int B, pos0, ctx_max, step_base;
        const void* pos_dev;
        int has_samp, n_hist0, engine;
        vcla_sample_args samp;
    } graph_key = {};
};

static inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }
static inline size_t esz(const vcla_ctx* c) { return vcla_dtype_size(c->c.act_dtype); }

extern "C" int vcla_ctx_create(const vcla_model_cfg* cfg, vcla_ctx** out) {
    VCLA_REQUIRE(cfg && out, VCLA_ERR_BAD_ARG, "ctx_create: null pointer");
    const vcla_model_cfg& c = *cfg;
    VCLA_REQUIRE(c.act_dtype == VCLA_F32 || c.act_dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "ctx_create: bad act_dtype %d", c.act_dtype);
    auto head_ok = [](int hidden, int heads) {
        if (heads <= 0 || hidden % heads) return false;
        const int d = hidden / heads;
        return d == 32 || d == 64 || d == 128;
    };
    VCLA_REQUIRE(head_ok(c.v_hidden, c.v_heads) && head_ok(c.r_hidden, c.r_heads) && head_ok(c.t_hidden, c.t_heads),
                 VCLA_ERR_BAD_SHAPE, "ctx_create: head dims must be 32, 64 or 128 (vision %d/%d, resampler %d/%d, text %d/%d)",
                 c.v_hidden, c.v_heads, c.r_hidden, c.r_heads, c.t_hidden, c.t_heads);
    VCLA_REQUIRE(c.v_hidden % 64 == 0 && c.v_inter % 64 == 0 && c.r_hidden % 64 == 0 && c.r_inter % 64 == 0 &&
                     c.t_hidden % 64 == 0 && c.t_inter % 64 == 0, VCLA_ERR_BAD_SHAPE,
                 "ctx_create: hidden / intermediate sizes must be multiples of 64");
    VCLA_REQUIRE(c.t_inter % 16 == 0, VCLA_ERR_BAD_SHAPE, "ctx_create: t_inter must be a multiple of 16 (SwiGLU packing)");
    VCLA_REQUIRE(!c.t_kv_fp8 || (c.act_dtype == VCLA_BF16 && c.t_heads > 0 && (c.t_hidden / c.t_heads == 64 || c.t_hidden / c.t_heads == 128)), VCLA_ERR_BAD_ARG,
                 "ctx_create: t_kv_fp8 (e4m3 K/V cache) needs bf16 activations and a text head dim of 64 or 128");
    VCLA_REQUIRE(c.r_hidden == c.v_hidden, VCLA_ERR_BAD_SHAPE,
                 "ctx_create: resampler hidden (%d) must equal vision hidden (%d): latents are concatenated with image tokens",
                 c.r_hidden, c.v_hidden);
    // t_layers == 0 && t_vocab == 0: a vision-only context (ViT + resampler + projection into t_hidden), the half the
    // text-generation-webui pipeline consumes (next row N4); the llama entry points refuse it.
    const bool vision_only = c.t_layers == 0 && c.t_vocab == 0;
    VCLA_REQUIRE(c.v_patch > 0 && c.v_image % c.v_patch == 0 && c.v_layers > 0 && c.r_layers > 0 && c.r_queries > 0 && c.v_channels > 0 &&
                     (vision_only || (c.t_layers > 0 && c.t_vocab > 0 && c.t_max_pos > 0)),
                 VCLA_ERR_BAD_SHAPE, "ctx_create: bad geometry");
    vcla_ctx* x = new (std::nothrow) vcla_ctx();
    VCLA_REQUIRE(x, VCLA_ERR_BAD_ARG, "ctx_create: out of host memory");
    x->c = c;
    x->k_pad = pad_to(c.v_channels * c.v_patch * c.v_patch, 64);
    *out = x;
    return VCLA_OK;
}

// The cached graphs bake in weight pointers: registering a tensor (LoRA swap, adding fp8 copies) makes all of them stale.
static void drop_graphs(vcla_ctx* ctx) {
    // An exec may still be executing (its last hipGraphLaunch is asynchronous) and HIP does not promise deferred destruction on every ROCm
    // version: drain the device first.  Only when there is something to destroy -- registering the tensors of a fresh context costs nothing.
    bool any = ctx->graph_exec || ctx->graph_exec_multi;
    for (const vcla_ctx::MacroGraph* g : {&ctx->vision_graph, &ctx->prefill_graph})
        for (int i = 0; i < vcla_ctx::MacroGraph::kSlots; ++i) any = any || g->exec[i];
    if (any && hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
    if (ctx->graph_exec) { (void)hipGraphExecDestroy(ctx->graph_exec); ctx->graph_exec = nullptr; }
    if (ctx->graph_exec_multi) { (void)hipGraphExecDestroy(ctx->graph_exec_multi); ctx->graph_exec_multi = nullptr; }
    for (vcla_ctx::MacroGraph* g : {&ctx->vision_graph, &ctx->prefill_graph}) {
        for (int i = 0; i < vcla_ctx::MacroGraph::kSlots; ++i)
            if (g->exec[i]) { (void)hipGraphExecDestroy(g->exec[i]); g->exec[i] = nullptr; }
        g->has_seen = 0;
    }
}

extern "C" void vcla_ctx_destroy(vcla_ctx* ctx) {
    if (!ctx) return;
    drop_graphs(ctx);
    delete ctx;
}

extern "C" int vcla_ctx_set_tensor(vcla_ctx* ctx, const char* name, const void* ptr, size_t nbytes) {
    VCLA_REQUIRE(ctx && name && ptr, VCLA_ERR_BAD_ARG, "ctx_set_tensor: null pointer");
    VCLA_REQUIRE(vcla_aligned(ptr, 16), VCLA_ERR_BAD_ARG, "ctx_set_tensor: %s is not 16-byte aligned", name);
    ctx->tensors[name] = TensorRef{ptr, nbytes};
    ctx->finalized = false;
    drop_graphs(ctx);
    return VCLA_OK;
}

static int get_tensor(vcla_ctx* ctx, const std::string& name, size_t want_bytes, const void** out) {
    auto it = ctx->tensors.find(name);
    if (it == ctx->tensors.end()) return vcla_fail(VCLA_ERR_MISSING_TENSOR, "tensor '%s' was not registered", name.c_str());
    if (it->second.nbytes != want_bytes)
        return vcla_fail(VCLA_ERR_BAD_SHAPE, "tensor '%s' has %zu bytes, expected %zu", name.c_str(), it->second.nbytes, want_bytes);
    *out = it->second.ptr;
    return VCLA_OK;
}

// optional tensor: nullptr when absent, size-checked when present
static int get_tensor_opt(vcla_ctx* ctx, const std::string& name, size_t want_bytes, const void** out) {
    *out = nullptr;
    if (ctx->tensors.find(name) == ctx->tensors.end()) return VCLA_OK;
    return get_tensor(ctx, name, want_bytes, out);
}
static int get_variants(vcla_ctx* ctx, const std::string& name, int rows, int cols, WVar* v) {
    const size_t np = (size_t)pad_to(rows, 128);
    const void* s8 = nullptr;
    int rc = get_tensor_opt(ctx, name + ".f", np * cols * 2, &v->frag);
    if (!rc) rc = get_tensor_opt(ctx, name + ".q8", np * cols, &v->q8);
    if (!rc) rc = get_tensor_opt(ctx, name + ".q8f", np * cols, &v->q8f);
    if (!rc) rc = get_tensor_opt(ctx, name + ".s8", np * 4, &s8);
    v->s8 = (const float*)s8;
    if (!rc && (v->q8 || v->q8f) && !(v->q8 && v->q8f && v->s8))
        return vcla_fail(VCLA_ERR_MISSING_TENSOR, "fp8 copies of '%s' need all of .q8, .q8f and .s8", name.c_str());
    return rc;
}
#define GET_WF(dst, name, rows, cols)                             \
    do {                                                          \
        int _rc = get_variants(ctx, name, rows, cols, &dst);      \
        if (_rc) return _rc;                                      \
    } while (0)

#define GET_W(dst, name, rows, cols)                                                                       \
    do {                                                                                                   \
        const void* _p;                                                                                    \
        int _rc = get_tensor(ctx, name, (size_t)pad_to(rows, 128) * (size_t)(cols) * 2, &_p);               \
        if (_rc) return _rc;                                                                               \
        dst = _p;                                                                                          \
    } while (0)
#define GET_F(dst, name, n)                                                         \
    do {                                                                            \
        const void* _p;                                                             \
        int _rc = get_tensor(ctx, name, (size_t)(n) * 4, &_p);                      \
        if (_rc) return _rc;                                                        \
        dst = (const float*)_p;                                                     \
    } while (0)

extern "C" int vcla_ctx_finalize(vcla_ctx* ctx) {
    VCLA_REQUIRE(ctx, VCLA_ERR_BAD_ARG, "ctx_finalize: null ctx");
    const vcla_model_cfg& c = ctx->c;
    const int np = (c.v_image / c.v_patch) * (c.v_image / c.v_patch), N = np + 1;
    GET_W(ctx->vit_patch_w, "vit.patch_w", c.v_hidden, ctx->k_pad);
    GET_F(ctx->vit_cls, "vit.cls", c.v_hidden);
    GET_F(ctx->vit_pos, "vit.pos", (size_t)N * c.v_hidden);
    GET_F(ctx->vit_pre_g, "vit.pre_ln.g", c.v_hidden);
    GET_F(ctx->vit_pre_b, "vit.pre_ln.b", c.v_hidden);
    GET_F(ctx->vit_post_g, "vit.post_ln.g", c.v_hidden);
    GET_F(ctx->vit_post_b, "vit.post_ln.b", c.v_hidden);
    ctx->vit.resize(c.v_layers);
    for (int i = 0; i < c.v_layers; ++i) {
        const std::string p = "vit.l" + std::to_string(i) + ".";
        VitLayer& L = ctx->vit[i];
        GET_F(L.ln1g, p + "ln1.g", c.v_hidden); GET_F(L.ln1b, p + "ln1.b", c.v_hidden);
        GET_W(L.wqkv, p + "wqkv", 3 * c.v_hidden, c.v_hidden); GET_F(L.bqkv, p + "bqkv", 3 * c.v_hidden);
        GET_W(L.wo, p + "wo", c.v_hidden, c.v_hidden); GET_F(L.bo, p + "bo", c.v_hidden);
        GET_F(L.ln2g, p + "ln2.g", c.v_hidden); GET_F(L.ln2b, p + "ln2.b", c.v_hidden);
        GET_W(L.w1, p + "w1", c.v_inter, c.v_hidden); GET_F(L.b1, p + "b1", c.v_inter);
        GET_W(L.w2, p + "w2", c.v_hidden, c.v_inter); GET_F(L.b2, p + "b2", c.v_hidden);
    }
    {
        const void* q;
        int rc = get_tensor(ctx, "res.query", (size_t)c.r_queries * c.r_hidden * esz(ctx), &q);
        if (rc) return rc;
        ctx->res_query = q;
    }
    ctx->res.resize(c.r_layers);
    for (int i = 0; i < c.r_layers; ++i) {
        const std::string p = "res.l" + std::to_string(i) + ".";
        ResLayer& L = ctx->res[i];
        GET_W(L.wq, p + "wq", c.r_hidden, c.r_hidden); GET_F(L.bq, p + "bq", c.r_hidden);
        GET_W(L.wkv, p + "wkv", 2 * c.r_hidden, c.r_hidden); GET_F(L.bkv, p + "bkv", 2 * c.r_hidden);
        GET_W(L.wo, p + "wo", c.r_hidden, c.r_hidden); GET_F(L.bo, p + "bo", c.r_hidden);
        GET_F(L.ln1g, p + "ln1.g", c.r_hidden); GET_F(L.ln1b, p + "ln1.b", c.r_hidden);
        GET_W(L.w1, p + "w1", c.r_inter, c.r_hidden); GET_F(L.b1, p + "b1", c.r_inter);
        GET_W(L.w2, p + "w2", c.r_hidden, c.r_inter); GET_F(L.b2, p + "b2", c.r_hidden);
        GET_F(L.ln2g, p + "ln2.g", c.r_hidden); GET_F(L.ln2b, p + "ln2.b", c.r_hidden);
    }
    {   // optional stacked K/V weights (no row padding between the layers: 2D % 128 == 0)
        const void *wa = nullptr, *ba = nullptr;
        int rc = get_tensor_opt(ctx, "res.wkv_all", (size_t)c.r_layers * 2 * c.r_hidden * c.r_hidden * 2, &wa);
        if (!rc) rc = get_tensor_opt(ctx, "res.bkv_all", (size_t)c.r_layers * 2 * c.r_hidden * 4, &ba);
        if (rc) return rc;
        const bool ok = wa && ba && (2 * c.r_hidden) % 128 == 0;
        ctx->res_wkv_all = ok ? wa : nullptr;
        ctx->res_bkv_all = ok ? (const float*)ba : nullptr;
    }
    GET_W(ctx->proj_w, "proj.w", c.t_hidden, c.r_hidden);
    GET_F(ctx->proj_b, "proj.b", c.t_hidden);
    if (c.t_layers == 0) {   // vision-only context
        ctx->finalized = true;
        return VCLA_OK;
    }
    {
        const void* e;
        int rc = get_tensor(ctx, "llama.embed", (size_t)c.t_vocab * c.t_hidden * 2, &e);
        if (rc) return rc;
        ctx->embed = e;
    }
    ctx->llama.resize(c.t_layers);
    for (int i = 0; i < c.t_layers; ++i) {
        const std::string p = "llama.l" + std::to_string(i) + ".";
        LlamaLayer& L = ctx->llama[i];
        GET_F(L.ln1g, p + "ln1.g", c.t_hidden); GET_F(L.ln2g, p + "ln2.g", c.t_hidden);
        GET_W(L.wqkv, p + "wqkv", 3 * c.t_hidden, c.t_hidden);
        GET_W(L.wo, p + "wo", c.t_hidden, c.t_hidden);
        GET_W(L.wgu, p + "wgu", 2 * c.t_inter, c.t_hidden);
        GET_W(L.wd, p + "wd", c.t_hidden, c.t_inter);
        GET_WF(L.vqkv, p + "wqkv", 3 * c.t_hidden, c.t_hidden);
        GET_WF(L.vo, p + "wo", c.t_hidden, c.t_hidden);
        GET_WF(L.vgu, p + "wgu", 2 * c.t_inter, c.t_hidden);
        GET_WF(L.vd, p + "wd", c.t_hidden, c.t_inter);
    }
    GET_F(ctx->norm_g, "llama.norm.g", c.t_hidden);
    GET_W(ctx->lm_head, "llama.lm_head", c.t_vocab, c.t_hidden);
    GET_WF(ctx->vlm, "llama.lm_head", c.t_vocab, c.t_hidden);
    const int d = c.t_hidden / c.t_heads;
    GET_F(ctx->rope_cos, "llama.rope_cos", (size_t)c.t_max_pos * (d / 2));
    GET_F(ctx->rope_sin, "llama.rope_sin", (size_t)c.t_max_pos * (d / 2));
    ctx->eng_w = nullptr; ctx->eng_g = nullptr;
    if (c.act_dtype == VCLA_BF16 && !c.t_kv_fp8 && vcla_engine_geometry(c.t_hidden, c.t_heads, c.t_inter, c.t_vocab, c.t_layers, &ctx->eng_geom)) {
        const void *ew = nullptr, *eg = nullptr;
        int rc = get_tensor_opt(ctx, "llama.engine.w", (size_t)EG_NCU * ctx->eng_geom.slots_total * EG_SLOT, &ew);
        if (!rc) rc = get_tensor_opt(ctx, "llama.engine.g", (size_t)(2 * c.t_layers + 1) * EG_D * 4, &eg);
        if (rc) return rc;
        if (!rc && !ew) {      // A/B only: "llama.engine.w.cu" = the first form of the round, [CU][slot][16 KiB] (every CU's run contiguous)
            rc = get_tensor_opt(ctx, "llama.engine.w.cu", (size_t)EG_NCU * ctx->eng_geom.slots_total * EG_SLOT, &ew);
            if (rc) return rc;
            if (ew) { ctx->eng_geom.cu_stride = (size_t)ctx->eng_geom.slots_total * EG_SLOT; ctx->eng_geom.slot_stride = EG_SLOT; }
        }
        if (ew && eg) { ctx->eng_w = ew; ctx->eng_g = (const float*)eg; }
    }
    ctx->finalized = true;
    return VCLA_OK;
}

// ------------------------------------------------------------------ workspace
struct Bump {
    char* base;
    size_t off, cap;
    void* take(size_t bytes) {
        const size_t o = (off + 255) & ~(size_t)255;
        off = o + bytes;
        return base ? base + o : nullptr;
    }
};

#define SPLITK_WS_BYTES ((size_t)64 << 20)   // fp32 scratch for the split-K panel GEMM (M <= 128)
struct VisionWs {
    void *patches, *patch_emb, *x, *h, *qkv, *mlp;          // ViT
    void *lat, *q, *kv, *ao, *t, *h2, *ffn;                 // resampler
    void* splitk;
};
static size_t carve_vision(const vcla_ctx* ctx, int B, char* base, VisionWs* w) {
    const vcla_model_cfg& c = ctx->c;
    const size_t e = esz(ctx);
    const size_t np = (size_t)(c.v_image / c.v_patch) * (c.v_image / c.v_patch), N = np + 1, Q = c.r_queries;
    Bump b{base, 0, 0};
    VisionWs t;
    t.patches = b.take(B * np * ctx->k_pad * e);
    t.patch_emb = b.take(B * np * c.v_hidden * e);
    t.x = b.take(B * N * c.v_hidden * e);
    t.h = b.take(B * N * c.v_hidden * e);
    t.qkv = b.take(B * N * 3 * c.v_hidden * e);
    t.mlp = b.take(B * N * c.v_inter * e);
    t.lat = b.take(B * Q * c.r_hidden * e);
    t.q = b.take(B * Q * c.r_hidden * e);
    t.kv = b.take(B * (Q + N) * 2 * c.r_hidden * e * (ctx->res_wkv_all ? c.r_layers : 1));   // [B, Q+N, L, 2D] when the image-row K/V are hoisted
    t.ao = b.take(B * Q * c.r_hidden * e);
    t.t = b.take(B * Q * c.r_hidden * e);
    t.h2 = b.take(B * Q * c.r_hidden * e);
    t.ffn = b.take(B * Q * c.r_inter * e);
    t.splitk = b.take(SPLITK_WS_BYTES);
    if (w) *w = t;
    return b.off + 256;
}
extern "C" size_t vcla_vision_workspace_bytes(const vcla_ctx* ctx, int B) {
    if (!ctx || B <= 0) return 0;
    return carve_vision(ctx, B, nullptr, nullptr);
}

struct LlamaWs {
    void *x, *h, *qkv, *ao, *act, *hl;
    float* logits;
    int64_t* ids;
    void* splitk;
    float* ssq;     // [64][t_hidden / 16] per-row partial sums of squares (deferred RMSNorm of the streaming decode GEMMs)
    void* q8;       // [M][max(t_hidden, t_inter)] fp8 copy of the activation operand (fp8 MFMA prefill, t_fp8_mfma)
    float* q8s;     // [M] its per-row scales
    int* ticket;    // arrival counter of post_select_kernel (zeroed by the decode loop before its first step)
    void* eng;      // mailboxes + state words of the persistent decode step (EG_WS_BYTES; zeroed before the first step of a loop)
};
static size_t carve_llama(const vcla_ctx* ctx, int B, int T, char* base, LlamaWs* w) {
    const vcla_model_cfg& c = ctx->c;
    const size_t e = esz(ctx);
    const size_t M = ((size_t)B * T + 15) / 16 * 16;   // rows padded to whole 16-row MFMA tiles (fragment-major decode buffers)
    Bump b{base, 0, 0};
    LlamaWs t;
    t.x = b.take(M * c.t_hidden * e);
    t.h = b.take(M * c.t_hidden * e);
    t.qkv = b.take(M * 3 * c.t_hidden * e);
    t.ao = b.take(M * c.t_hidden * e);
    t.act = b.take(M * c.t_inter * e);
    t.hl = b.take((size_t)B * c.t_hidden * e);
    t.logits = (float*)b.take((size_t)B * c.t_vocab * 4);
    t.ids = (int64_t*)b.take((size_t)B * 8);
    t.splitk = b.take(SPLITK_WS_BYTES);
    t.ssq = (float*)b.take((size_t)64 * ((c.t_hidden + 15) / 16) * 4);
    t.ticket = (int*)b.take(256);
    t.eng = b.take(EG_WS_BYTES);
    t.q8 = nullptr; t.q8s = nullptr;
    if (c.t_fp8_mfma && (size_t)B * T > 128) {
        t.q8 = b.take(M * (size_t)(c.t_hidden > c.t_inter ? c.t_hidden : c.t_inter));
        t.q8s = (float*)b.take(M * 4);
    }
    if (w) *w = t;
    return b.off + 256;
}
extern "C" size_t vcla_llama_workspace_bytes(const vcla_ctx* ctx, int B, int T) {
    if (!ctx || B <= 0 || T <= 0) return 0;
    return carve_llama(ctx, B, T, nullptr, nullptr);
}
extern "C" size_t vcla_kv_cache_bytes(const vcla_ctx* ctx, int B, int ctx_max) {
    if (!ctx || B <= 0 || ctx_max <= 0) return 0;
    return (size_t)ctx->c.t_layers * 2 * B * ctx->c.t_hidden * (size_t)ctx_max * (ctx->c.t_kv_fp8 ? 1 : esz(ctx));   // t_kv_fp8: e4m3 bytes
}

// ------------------------------------------------------------------ small wrappers
static int gemm(vcla_ctx* ctx, hipStream_t s, const void* A, int64_t lda, const void* W, const float* bias,
                const void* residual, int64_t ldr, void* C, int64_t ldc, int M, int N, int K, int epi, int out_f32 = 0,
                int grp_rows = 0, int grp_stride = 0, int row_off = 0, const float* norm_gamma = nullptr, float norm_eps = 0.f,
                const WVar* wv = nullptr, const float* post_gamma = nullptr, float post_eps = 0.f, void* post_out = nullptr,
                int64_t post_ld = 0) {
    vcla_gemm_args a{};
    a.post_norm_gamma = post_gamma; a.post_norm_eps = post_eps; a.post_norm_out = post_out; a.post_norm_ld = post_ld;
    a.A = A; a.lda = lda; a.W = W; a.bias = bias; a.residual = residual; a.ldr = ldr; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.epilogue = epi; a.out_f32 = out_f32;
    a.c_group_rows = grp_rows; a.c_group_stride = grp_stride; a.c_row_offset = row_off;
    a.force_kernel = 0;
    a.norm_gamma = norm_gamma; a.norm_eps = norm_eps;
    a.splitk_ws = ctx->run.splitk_ws; a.splitk_ws_bytes = ctx->run.splitk_ws ? SPLITK_WS_BYTES : 0;
    if (wv && wv->q8 && wv->s8 && ctx->run.q8_ws && M > 128 && K % 128 == 0 && lda == K && ctx->c.act_dtype == VCLA_BF16 && !norm_gamma && grp_rows == 0) {
        // BASELINE configs[4]: prefill on the fp8 MFMA pipe -- quantise the activation rows (one pass over M x K), then fp8 x fp8
        int rc = vcla_quant_fp8_rows(A, lda, ctx->run.q8_ws, ctx->run.q8s_ws, M, K, s);
        if (rc) return rc;
        a.A = nullptr; a.A_q8 = ctx->run.q8_ws; a.a_scale = ctx->run.q8s_ws; a.W_q8 = wv->q8; a.w_scale = wv->s8; a.force_kernel = 10;
        return vcla_gemm(&a, ctx->c.act_dtype, s);
    }
    static const int ring_env = getenv("VCLA_RING") ? atoi(getenv("VCLA_RING")) : 1;      // VCLA_RING=0: the round-4 dispatch everywhere (gemm.hip), also for the fp8 rows below
    if (ring_env && wv && M > 128 && M <= 256 && ctx->run.decode_step && wv->q8 && wv->s8 && ctx->c.act_dtype == VCLA_BF16 && !norm_gamma && grp_rows == 0 &&
        (epi == VCLA_EPI_NONE || (epi == VCLA_EPI_SWIGLU && !out_f32))) {
        // BASELINE configs[4], decode batches of 129 - 256 sequences (its N = 1 leg, B = 256): the ring kernel stages the fp8 rows as they are and
        // widens them in registers -- the SAME W8A16 function of the dequantised weights the M <= 128 decode kernels compute, whatever the batch
        // size (round 4 read the bf16 matrices here); lm_head included (fp32 logits).
        a.W_q8 = wv->q8; a.w_scale = wv->s8; a.force_kernel = 11;
        return vcla_gemm(&a, ctx->c.act_dtype, s);
    }
    if (wv && M > 128 && M <= 256 && wv->frag && ctx->c.act_dtype == VCLA_BF16 && !out_f32) {
        // 129 - 256 rows (ring kernel): the fragment-major twin makes the weight half of every stage contiguous 1-KiB DMA pieces
        a.W_frag = wv->frag;
        return vcla_gemm(&a, ctx->c.act_dtype, s);
    }
    if (wv && M <= 128 && ctx->c.act_dtype == VCLA_BF16) {   // decode-side weight copies (prefill tiles read the bf16 row-major W)
        // fp8 copies (when loaded) serve the DECODE steps only (ctx->run.decode_step; a short prefill keeps the bf16 values): half the HBM
        // bytes.  M = 1 GEMV: 1.3x end to end; panel kernel: 112 vs 132 us per layer at
        // M = 64, 88 vs 111 us at M = 32 (tools/bench_kernels.py panel, after the fetch-past-the-slice fix)
        if (ctx->run.decode_step && wv->q8 && wv->q8f && wv->s8) { a.W_q8 = wv->q8; a.W_q8_frag = wv->q8f; a.w_scale = wv->s8; }
        else a.W_frag = wv->frag;
    }
    return vcla_gemm(&a, ctx->c.act_dtype, s);
}

// streaming decode GEMM (gemm_stream.hip): fragment-major activations in, optional fragment-major copy out
static int gemm_ds(vcla_ctx* ctx, hipStream_t s, const void* A_frag, const void* W, const WVar& wv, const void* residual, int64_t ldr,
                   void* C, int64_t ldc, void* C_frag, int M, int N, int K, int epi, int out_f32 = 0, const float* a_ssq = nullptr,
                   int a_parts = 0, const float* c_gamma = nullptr, float* c_ssq = nullptr, int splitk = 0, bool raw = false) {
    vcla_gemm_args a{};
    if (splitk > 1 && ctx->run.splitk_ws && (size_t)splitk * M * N * 4 <= SPLITK_WS_BYTES && epi == VCLA_EPI_NONE && !out_f32) {
        a.ds_splitk = splitk; a.splitk_ws = ctx->run.splitk_ws; a.splitk_ws_bytes = SPLITK_WS_BYTES;
        a.ds_raw_partials = raw ? 1 : 0;
    }
    // partial sums of squares per row this call leaves in c_ssq: the split-K reduce launch writes one per 256 columns (N % 256 ==
    // 0), the in-kernel epilogue one per 16-column tile; the consumer (the next gemm_ds with a_ssq) must be told which
    if (c_ssq) ctx->run.ssq_parts = (a.ds_splitk > 1 && (N & 255) == 0) ? N / 256 : N / 16;
    a.a_row_ssq = a_ssq; a.a_row_ssq_parts = a_parts; a.a_norm_eps = ctx->c.t_eps; a.c_frag_gamma = c_gamma; a.c_row_ssq = c_ssq;
    a.A_frag = A_frag; a.W = W; a.residual = residual; a.ldr = ldr; a.C = C; a.ldc = ldc; a.C_frag = C_frag;
    a.M = M; a.N = N; a.K = K; a.epilogue = epi; a.out_f32 = out_f32; a.force_kernel = 9;
    if (ctx->run.decode_step && wv.q8f && wv.s8) { a.W_q8_frag = wv.q8f; a.w_scale = wv.s8; }
    else a.W_frag = wv.frag;
    return vcla_gemm(&a, VCLA_BF16, s);
}

#define RUN(expr)            \
    do {                     \
        int _rc = (expr);    \
        if (_rc) return _rc; \
    } while (0)

static int tap_copy(void* tap, size_t index, const void* src, size_t bytes, hipStream_t s) {
    if (!tap) return VCLA_OK;
    VCLA_CHECK_HIP(hipMemcpyAsync((char*)tap + index * bytes, src, bytes, hipMemcpyDeviceToDevice, s));
    return VCLA_OK;
}

// ------------------------------------------------------------------ macro graphs (vision stack, prefill)
// run(stream) issues the launch sequence.  Replay when `key` matches the captured graph; capture when it matches the previous eager
// call; else run eagerly and remember the key.  Capture is illegal on the legacy default stream (s == 0): eager there.
template <typename F>
static int run_macro(vcla_ctx::MacroGraph& g, const void* const (&key)[8], hipStream_t s, F&& run) {
    static const int genv = getenv("VCLA_MACRO_GRAPH") ? atoi(getenv("VCLA_MACRO_GRAPH")) : 1;
    if (!genv || s == nullptr) return run(s);
    // The caller is capturing on this stream itself (torch.cuda.graph, or its own hipGraph around this entry point): the launches
    // belong in ITS graph -- replaying ours or opening a nested capture would both be wrong.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return run(s); }
    if (cap != hipStreamCaptureStatusNone) return run(s);
    static const int dbg = getenv("VCLA_MACRO_GRAPH_DEBUG") ? atoi(getenv("VCLA_MACRO_GRAPH_DEBUG")) : 0;
    constexpr int NS = vcla_ctx::MacroGraph::kSlots;
    for (int i = 0; i < NS; ++i) {
        if (g.exec[i] && memcmp(g.key[i], key, sizeof(g.key[i])) == 0) {
            VCLA_CHECK_HIP(hipGraphLaunch(g.exec[i], s));
            g.used[i] = ++g.tick;
            if (dbg) fprintf(stderr, "[vcla] macro graph %p: replay (slot %d)\n", (void*)&g, i);
            return VCLA_OK;
        }
    }
    if (dbg) fprintf(stderr, "[vcla] macro graph %p: eager call (no captured graph has this key)\n", (void*)&g);
    // not a captured key: this call runs eagerly (which also leaves every per-device function attribute set); the graph is
    // captured behind it -- recorded, not executed -- when the key is the very first one or repeats the previous eager call's, so a
    // caller whose buffers move on every call never pays for captures it cannot reuse
    const bool repeat = g.has_seen && memcmp(g.seen, key, sizeof(g.seen)) == 0;
    const bool first = !g.has_seen;
    memcpy(g.seen, key, sizeof(g.seen));
    g.has_seen = 1;
    const int rc_eager = run(s);
    if (rc_eager || !(repeat || first)) return rc_eager;
    // From here on the call has SUCCEEDED (the eager launches are issued): a capture that cannot be taken or instantiated only means
    // "no cached graph" -- the next call runs eagerly again.
    auto no_graph = [&](const char* what, hipError_t e) {
        if (dbg) fprintf(stderr, "[vcla] macro graph %p: %s failed (%s), staying eager\n", (void*)&g, what, hipGetErrorString(e));
        (void)hipGetLastError();
        return VCLA_OK;
    };
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return no_graph("hipStreamBeginCapture", e);
    const int rc = run(s);
    e = hipStreamEndCapture(s, &graph);
    if (rc || e != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        return no_graph("capture", e);
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return no_graph("hipGraphInstantiate", e);
    int victim = 0;
    for (int i = 0; i < NS; ++i) {
        if (!g.exec[i]) { victim = i; break; }
        if (g.used[i] < g.used[victim]) victim = i;
    }
    if (g.exec[victim]) (void)hipGraphExecDestroy(g.exec[victim]);
    g.exec[victim] = exec;
    g.used[victim] = ++g.tick;
    memcpy(g.key[victim], key, sizeof(g.key[victim]));
    return VCLA_OK;
}

// ------------------------------------------------------------------ vision: ViT + post-LN + resampler + projection
static int vision_forward_impl(vcla_ctx* ctx, const void* pixel_values, void* image_embeds, int B, void* ws, void* vit_tap, void* res_tap,
                               hipStream_t s);

extern "C" int vcla_vision_forward(vcla_ctx* ctx, const void* pixel_values, void* image_embeds, int B, void* ws,
                                   size_t ws_bytes, void* vit_tap, void* res_tap, void* stream) {
    VCLA_REQUIRE(ctx && ctx->finalized, VCLA_ERR_BAD_ARG, "vision_forward: context not finalized");
    VCLA_REQUIRE(pixel_values && image_embeds && ws, VCLA_ERR_BAD_ARG, "vision_forward: null pointer");
    VCLA_REQUIRE(B > 0, VCLA_ERR_BAD_SHAPE, "vision_forward: B=%d", B);
    VCLA_REQUIRE(ws_bytes >= vcla_vision_workspace_bytes(ctx, B), VCLA_ERR_WORKSPACE, "vision_forward: workspace %zu < %zu bytes",
                 ws_bytes, vcla_vision_workspace_bytes(ctx, B));
    hipStream_t s = (hipStream_t)stream;
    if (vit_tap || res_tap) return vision_forward_impl(ctx, pixel_values, image_embeds, B, ws, vit_tap, res_tap, s);   // parity runs: eager
    const void* const key[8] = {pixel_values, image_embeds, ws, (const void*)(intptr_t)B, (const void*)(intptr_t)ctx->c.v_image, nullptr, nullptr, nullptr};
    return run_macro(ctx->vision_graph, key, s, [&](hipStream_t st) { return vision_forward_impl(ctx, pixel_values, image_embeds, B, ws, nullptr, nullptr, st); });
}

static int vision_forward_impl(vcla_ctx* ctx, const void* pixel_values, void* image_embeds, int B, void* ws, void* vit_tap, void* res_tap,
                               hipStream_t s) {
    const vcla_model_cfg& c = ctx->c;
    const int dt = c.act_dtype;
    const size_t e = esz(ctx);
    const int g = c.v_image / c.v_patch, np = g * g, N = np + 1, D = c.v_hidden, H = c.v_heads, d = D / H;
    VisionWs w;
    carve_vision(ctx, B, (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), &w);
    ctx->run.splitk_ws = w.splitk;
    const int M = B * N;

    // patch embedding: im2col -> GEMM (no bias) -> class/position embedding + pre-LN
    RUN(vcla_im2col(pixel_values, w.patches, B, c.v_channels, c.v_image, c.v_image, c.v_patch, ctx->k_pad, dt, s));
    RUN(gemm(ctx, s, w.patches, ctx->k_pad, ctx->vit_patch_w, nullptr, nullptr, 0, w.patch_emb, D, B * np, D, ctx->k_pad, VCLA_EPI_NONE));
    RUN(vcla_vit_assemble(w.patch_emb, ctx->vit_cls, ctx->vit_pos, ctx->vit_pre_g, ctx->vit_pre_b, w.x, B, np, D, c.v_eps, dt, s));
    RUN(tap_copy(vit_tap, c.v_layers + 1, w.x, (size_t)M * D * e, s));   // embeddings + pre-LN (hf CLIPVisionTransformer: pre_layrnorm output)

    for (int l = 0; l < c.v_layers; ++l) {
        const VitLayer& L = ctx->vit[l];
        RUN(vcla_layernorm(w.x, D, L.ln1g, L.ln1b, w.h, D, M, D, c.v_eps, dt, s));
        RUN(gemm(ctx, s, w.h, D, L.wqkv, L.bqkv, nullptr, 0, w.qkv, 3 * D, M, 3 * D, D, VCLA_EPI_NONE));
        vcla_attn_args a{};
        a.q = w.qkv; a.k = (char*)w.qkv + (size_t)D * e; a.v = (char*)w.qkv + (size_t)2 * D * e; a.o = w.h;
        a.q_bs = a.k_bs = a.v_bs = (int64_t)N * 3 * D; a.q_hs = a.k_hs = a.v_hs = d; a.q_rs = a.k_rs = a.v_rs = 3 * D;
        a.o_bs = (int64_t)N * D; a.o_hs = d; a.o_rs = D;
        a.B = B; a.H = H; a.Tq = N; a.Tk = N; a.D = d; a.scale = 1.0f / sqrtf((float)d); a.causal = 0;
        RUN(vcla_attention(&a, dt, s));
        RUN(gemm(ctx, s, w.h, D, L.wo, L.bo, w.x, D, w.x, D, M, D, D, VCLA_EPI_NONE));
        RUN(vcla_layernorm(w.x, D, L.ln2g, L.ln2b, w.h, D, M, D, c.v_eps, dt, s));
        RUN(gemm(ctx, s, w.h, D, L.w1, L.b1, nullptr, 0, w.mlp, c.v_inter, M, c.v_inter, D, VCLA_EPI_QUICK_GELU));
        RUN(gemm(ctx, s, w.mlp, c.v_inter, L.w2, L.b2, w.x, D, w.x, D, M, D, c.v_inter, VCLA_EPI_NONE));
        RUN(tap_copy(vit_tap, l, w.x, (size_t)M * D * e, s));
    }
    // post_layernorm over ALL tokens (models/visualcla/modeling_visualcla.py:284)
    RUN(vcla_layernorm(w.x, D, ctx->vit_post_g, ctx->vit_post_b, w.h, D, M, D, c.v_eps, dt, s));
    RUN(tap_copy(vit_tap, c.v_layers, w.h, (size_t)M * D * e, s));

    // ---- resampler
    const int Q = c.r_queries, Dr = c.r_hidden, Hr = c.r_heads, dr = Dr / Hr, KV = Q + N;
    {
        const int64_t n_per = (int64_t)Q * Dr;
        dim3 grid((unsigned)((n_per + 255) / 256), B);
        if (dt == VCLA_F32) bcast_rows_kernel<float><<<grid, 256, 0, s>>>((const float*)ctx->res_query, (float*)w.lat, n_per);
        else bcast_rows_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)ctx->res_query, (bf16_t*)w.lat, n_per);
        VCLA_CHECK_LAUNCH("bcast_rows_kernel");
    }
    const int Mq = B * Q;
    // K/V source of layer l = cat([latents_l, image tokens]) (modeling_visual_resampler.py:315): the image-token rows are the same
    // in every layer, so their K/V for ALL layers come from one GEMM [B*N, D] x [D, L*2D] up front (stacked weights,
    // res.wkv_all); layer l then only projects its 64 latent rows into its column block of the same [B, Q+N, L, 2D] buffer.
    const bool hoist = ctx->res_wkv_all != nullptr;
    const int kv_ld = hoist ? c.r_layers * 2 * Dr : 2 * Dr;          // row stride of the K/V buffer
    if (hoist)
        RUN(gemm(ctx, s, w.h, Dr, ctx->res_wkv_all, ctx->res_bkv_all, nullptr, 0, w.kv, kv_ld, M, c.r_layers * 2 * Dr, Dr, VCLA_EPI_NONE, 0, N, KV, Q));
    for (int l = 0; l < c.r_layers; ++l) {
        const ResLayer& L = ctx->res[l];
        char* kvl = (char*)w.kv + (hoist ? (size_t)l * 2 * Dr * e : 0);
        RUN(gemm(ctx, s, w.lat, Dr, L.wq, L.bq, nullptr, 0, w.q, Dr, Mq, Dr, Dr, VCLA_EPI_NONE));
        RUN(gemm(ctx, s, w.lat, Dr, L.wkv, L.bkv, nullptr, 0, kvl, kv_ld, Mq, 2 * Dr, Dr, VCLA_EPI_NONE, 0, Q, KV, 0));
        if (!hoist) RUN(gemm(ctx, s, w.h, Dr, L.wkv, L.bkv, nullptr, 0, kvl, kv_ld, M, 2 * Dr, Dr, VCLA_EPI_NONE, 0, N, KV, Q));
        vcla_attn_args a{};
        a.q = w.q; a.k = kvl; a.v = kvl + (size_t)Dr * e; a.o = w.ao;
        a.q_bs = (int64_t)Q * Dr; a.q_hs = dr; a.q_rs = Dr;
        a.k_bs = a.v_bs = (int64_t)KV * kv_ld; a.k_hs = a.v_hs = dr; a.k_rs = a.v_rs = kv_ld;
        a.o_bs = (int64_t)Q * Dr; a.o_hs = dr; a.o_rs = Dr;
        a.B = B; a.H = Hr; a.Tq = Q; a.Tk = KV; a.D = dr; a.scale = 1.0f / sqrtf((float)dr); a.causal = 0;
        RUN(vcla_attention(&a, dt, s));
        RUN(gemm(ctx, s, w.ao, Dr, L.wo, L.bo, w.lat, Dr, w.t, Dr, Mq, Dr, Dr, VCLA_EPI_NONE));
        RUN(vcla_layernorm(w.t, Dr, L.ln1g, L.ln1b, w.h2, Dr, Mq, Dr, c.r_eps, dt, s));
        RUN(gemm(ctx, s, w.h2, Dr, L.w1, L.b1, nullptr, 0, w.ffn, c.r_inter, Mq, c.r_inter, Dr, VCLA_EPI_GELU_ERF));
        RUN(gemm(ctx, s, w.ffn, c.r_inter, L.w2, L.b2, w.h2, Dr, w.t, Dr, Mq, Dr, c.r_inter, VCLA_EPI_NONE));
        RUN(vcla_layernorm(w.t, Dr, L.ln2g, L.ln2b, w.lat, Dr, Mq, Dr, c.r_eps, dt, s));
        RUN(tap_copy(res_tap, l, w.lat, (size_t)Mq * Dr * e, s));
    }
    // ---- projection into the text embedding space
    RUN(gemm(ctx, s, w.lat, Dr, ctx->proj_w, ctx->proj_b, nullptr, 0, image_embeds, c.t_hidden, Mq, c.t_hidden, Dr, VCLA_EPI_NONE));
    return VCLA_OK;
}

// ------------------------------------------------------------------ LLaMA decoder
// One decoder layer over M = B*T rows held in ws.x (updated in place).
static int llama_layer(vcla_ctx* ctx, hipStream_t s, const LlamaLayer& L, const LlamaWs& w, int l, int B, int T, int pos0,
                       const int32_t* pos_dev, void* kv_cache, int ctx_max, const int32_t* key_mask, bool h_ready = false,
                       const float* next_gamma = nullptr) {
    const vcla_model_cfg& c = ctx->c;
    const int dt = c.act_dtype;
    const size_t e = esz(ctx);
    const int D = c.t_hidden, H = c.t_heads, d = D / H, M = B * T;
    const bool kv8 = c.t_kv_fp8 != 0;                   // e4m3 cache rows (bf16 activations; checked at ctx creation)
    const int dkv = dt | (kv8 ? VCLA_KV_FP8 : 0);
    const size_t per = (size_t)B * H * ctx_max * d * (kv8 ? 1 : e);  // bytes of one K (or V) slab of one layer
    char* kc = (char*)kv_cache + (size_t)(2 * l) * per;
    char* vc = kc + per;
    // 2 <= M <= 64 decode rows with fragment-major weight copies: the streaming GEMMs (gemm_stream.hip).  Every operand is
    // fragment-major: RMSNorm writes its output that way (vcla_rmsnorm_pack), the attention kernel stores its output that way,
    // the SwiGLU epilogue stores the activations that way; the residual stream w.x stays row-major.  7 launches per layer,
    // no split-K partials.
    static const int ds_env = getenv("VCLA_DSTREAM") ? atoi(getenv("VCLA_DSTREAM")) : 1;
    const bool has_frag = (L.vqkv.frag && L.vo.frag && L.vgu.frag && L.vd.frag) || (L.vqkv.q8f && L.vo.q8f && L.vgu.q8f && L.vd.q8f);
    // Decode steps only (ctx->run.decode_step): the branch leaves w.h fragment-major and un-normalised, which decode_step_impl's
    // final norm + lm_head expect and vcla_llama_prefill's all_logits tail (row-major w.h) does not -- a single-token
    // forward(input_ids[B, 1], past_key_values=cache) goes through the panel kernels below.
    if (ds_env && ctx->run.decode_step && T == 1 && dt == VCLA_BF16 && M >= 2 && M <= 64 && has_frag && D % 32 == 0 && c.t_inter % 32 == 0) {
        // RMSNorm is deferred across the GEMMs (VCLA_DS_DEFER=0: a vcla_rmsnorm_pack launch per norm instead): o_proj / down_proj
        // store gamma * x fragment-major next to the residual stream plus per-row partial sums of squares, and the consuming
        // GEMM scales its accumulators by rstd(x): W . (gamma * x) * rstd = W . RMSNorm(x).  5 launches per layer.
        static const int defer_env = getenv("VCLA_DS_DEFER") ? atoi(getenv("VCLA_DS_DEFER")) : 1;
        const bool defer = defer_env != 0 && D % 16 == 0;
        // o_proj / down_proj (N = 4096 outputs, 16 columns per workgroup) are bound by re-reading the activations (a CU ingests only
        // ~45 GB/s from L2), not by the weights: K slices per tile group cut that traffic; a parallel reduce launch finishes the
        // tiles.  Measured at M = 64 (tools/bench_kernels.py dstream): down_proj 33.2 -> 25.7 us with 4 slices, o_proj 15.7 -> 15.0
        // with 2 (4: 15.8); at M = 32 only down_proj gains (24.4 -> 21.6 us).
        static const int sk_env = getenv("VCLA_DS_SPLITK") ? atoi(getenv("VCLA_DS_SPLITK")) : 4;
        const int sk_o = (M > 32 && D >= 2048 && sk_env > 1) ? 2 : 0, sk_d = (M > 16 && c.t_inter >= 4096) ? sk_env : 0;
        const float scale_ = 1.0f / sqrtf((float)d);
        if (!(defer && h_ready)) RUN(vcla_rmsnorm_pack(w.x, D, L.ln1g, w.h, M, D, c.t_eps, s));
        // qkv: like o_proj / down_proj it is bound by every CU reading the whole activation panel -- but its consumer reads just ONE row per
        // workgroup, so the K slices need no reduce launch: the projection leaves two raw fp32 slices (6 tiles per pair of workgroups, one
        // pass over half of K each) and the attention kernel sums them while it loads q / k / v, applying the deferred-RMSNorm rstd and the
        // fp8 weight scale itself.  M > 32 (below that the panel is small), B * H >= 1024 (the 2-wave attention form), 16 partial sums per row.
        static const int qs_env = getenv("VCLA_DS_QKV_SPLIT") ? atoi(getenv("VCLA_DS_QKV_SPLIT")) : 1;
        const bool a_def = defer && h_ready;
        const bool qkv_split = qs_env && M > 32 && (int64_t)B * H >= 1024 && d >= 64 && ctx->run.splitk_ws && (size_t)2 * M * 3 * D * 4 <= SPLITK_WS_BYTES &&
                               (!a_def || ctx->run.ssq_parts == 16);
        if (qkv_split) {
            RUN(gemm_ds(ctx, s, w.h, L.wqkv, L.vqkv, nullptr, 0, w.qkv, 3 * D, nullptr, M, 3 * D, D, VCLA_EPI_NONE, 0,
                        nullptr, 0, nullptr, nullptr, /*splitk=*/2, /*raw=*/true));
            const bool q8 = ctx->run.decode_step && L.vqkv.q8f && L.vqkv.s8;
            RUN(vcla_attn_decode_fused_parts((const float*)ctx->run.splitk_ws, (int64_t)M * 3 * D, a_def ? w.ssq : nullptr, q8 ? (const float*)L.vqkv.s8 : nullptr,
                                             c.t_eps, kc, vc, ctx->rope_cos, ctx->rope_sin, w.ao, B, H, d, ctx_max, pos0, pos_dev, key_mask, ctx_max,
                                             scale_, dkv, /*out_frag=*/1, s));
        } else {
            RUN(gemm_ds(ctx, s, w.h, L.wqkv, L.vqkv, nullptr, 0, w.qkv, 3 * D, nullptr, M, 3 * D, D, VCLA_EPI_NONE, 0,
                        a_def ? w.ssq : nullptr, a_def ? ctx->run.ssq_parts : 0));
            RUN(vcla_attn_decode_fused(w.qkv, kc, vc, ctx->rope_cos, ctx->rope_sin, w.ao, B, H, d, ctx_max, pos0, pos_dev, key_mask, ctx_max,
                                       scale_, dkv, /*out_frag=*/1, s));
        }
        if (defer) {
            RUN(gemm_ds(ctx, s, w.ao, L.wo, L.vo, w.x, D, w.x, D, w.h, M, D, D, VCLA_EPI_NONE, 0, nullptr, 0, L.ln2g, w.ssq, sk_o));
            RUN(gemm_ds(ctx, s, w.h, L.wgu, L.vgu, nullptr, 0, nullptr, 0, w.act, M, 2 * c.t_inter, D, VCLA_EPI_SWIGLU, 0, w.ssq, ctx->run.ssq_parts));
            RUN(gemm_ds(ctx, s, w.act, L.wd, L.vd, w.x, D, w.x, D, next_gamma ? w.h : nullptr, M, D, c.t_inter, VCLA_EPI_NONE, 0, nullptr, 0,
                        next_gamma, next_gamma ? w.ssq : nullptr, sk_d));
        } else {
            RUN(gemm_ds(ctx, s, w.ao, L.wo, L.vo, w.x, D, w.x, D, nullptr, M, D, D, VCLA_EPI_NONE, 0, nullptr, 0, nullptr, nullptr, sk_o));
            RUN(vcla_rmsnorm_pack(w.x, D, L.ln2g, w.h, M, D, c.t_eps, s));
            RUN(gemm_ds(ctx, s, w.h, L.wgu, L.vgu, nullptr, 0, nullptr, 0, w.act, M, 2 * c.t_inter, D, VCLA_EPI_SWIGLU));
            RUN(gemm_ds(ctx, s, w.act, L.wd, L.vd, w.x, D, w.x, D, nullptr, M, D, c.t_inter, VCLA_EPI_NONE, 0, nullptr, 0, nullptr, nullptr, sk_d));
        }
        return VCLA_OK;
    }
    // M <= 8 rows (decode): the GEMV kernel applies RMSNorm in its prologue -- no norm launch, no normalised copy.
    const bool fused = (dt == VCLA_F32) ? (M <= 8) : (M == 1);  // must mirror vcla_gemm's kernel choice
    // Otherwise the norms ride on the producing GEMM (post_norm_*: fused into the split-K reduction for M <= 128, a plain
    // vcla_rmsnorm launch behind the big tiles): o_proj emits post_attention_layernorm(x) and down_proj emits the NEXT
    // layer's input_layernorm(x) -- or the final norm -- into w.h.  `h_ready`: w.h already holds this layer's ln1(x).
    if (fused) {
        RUN(gemm(ctx, s, w.x, D, L.wqkv, nullptr, nullptr, 0, w.qkv, 3 * D, M, 3 * D, D, VCLA_EPI_NONE, 0, 0, 0, 0, L.ln1g, c.t_eps, &L.vqkv));
    } else {
        if (!h_ready) RUN(vcla_rmsnorm(w.x, D, L.ln1g, w.h, D, M, D, c.t_eps, dt, s));
        RUN(gemm(ctx, s, w.h, D, L.wqkv, nullptr, nullptr, 0, w.qkv, 3 * D, M, 3 * D, D, VCLA_EPI_NONE, 0, 0, 0, 0, nullptr, 0.f, &L.vqkv));
    }
    const float scale = 1.0f / sqrtf((float)d);
    if (T == 1) {
        // decode: RoPE + KV append + attention over the cache in one launch
        RUN(vcla_attn_decode_fused(w.qkv, kc, vc, ctx->rope_cos, ctx->rope_sin, w.ao, B, H, d, ctx_max, pos0, pos_dev, key_mask,
                                   ctx_max, scale, dkv, 0, s));
    } else {
        VCLA_REQUIRE(!kv8 || (pos0 == 0 && !pos_dev), VCLA_ERR_BAD_ARG,
                     "llama: with the fp8 K/V cache a multi-token forward must start at position 0 (it attends over its own bf16 rows; "
                     "the 1-byte cache serves single-token steps)");
        RUN(vcla_rope_kv_append(w.qkv, kc, vc, ctx->rope_cos, ctx->rope_sin, B, T, H, d, ctx_max, pos0, pos_dev, dkv, s));
        vcla_attn_args a{};
        a.q = w.qkv; a.k = kc; a.v = vc; a.o = w.ao;
        a.q_bs = (int64_t)T * 3 * D; a.q_hs = d; a.q_rs = 3 * D;
        a.k_bs = a.v_bs = (int64_t)H * ctx_max * d; a.k_hs = a.v_hs = (int64_t)ctx_max * d; a.k_rs = a.v_rs = d;
        if (kv8) {   // keys / values of this very forward, exact bf16, straight from the qkv rows (k was rotated in place)
            a.k = (const char*)w.qkv + (size_t)D * e; a.v = (const char*)w.qkv + (size_t)2 * D * e;
            a.k_bs = a.v_bs = a.q_bs; a.k_hs = a.v_hs = d; a.k_rs = a.v_rs = 3 * D;
        }
        a.o_bs = (int64_t)T * D; a.o_hs = d; a.o_rs = D;
        a.B = B; a.H = H; a.Tq = T; a.D = d; a.scale = scale; a.causal = 1;
        a.key_mask = key_mask; a.key_mask_ld = ctx_max;
        a.Tk = pos0 + T;
        RUN(vcla_attention(&a, dt, s));
    }
    if (fused) {
        RUN(gemm(ctx, s, w.ao, D, L.wo, nullptr, w.x, D, w.x, D, M, D, D, VCLA_EPI_NONE, 0, 0, 0, 0, nullptr, 0.f, &L.vo));
        RUN(gemm(ctx, s, w.x, D, L.wgu, nullptr, nullptr, 0, w.act, c.t_inter, M, 2 * c.t_inter, D, VCLA_EPI_SWIGLU, 0, 0, 0, 0, L.ln2g, c.t_eps, &L.vgu));
        RUN(gemm(ctx, s, w.act, c.t_inter, L.wd, nullptr, w.x, D, w.x, D, M, D, c.t_inter, VCLA_EPI_NONE, 0, 0, 0, 0, nullptr, 0.f, &L.vd));
    } else {
        RUN(gemm(ctx, s, w.ao, D, L.wo, nullptr, w.x, D, w.x, D, M, D, D, VCLA_EPI_NONE, 0, 0, 0, 0, nullptr, 0.f, &L.vo, L.ln2g, c.t_eps, w.h, D));
        RUN(gemm(ctx, s, w.h, D, L.wgu, nullptr, nullptr, 0, w.act, c.t_inter, M, 2 * c.t_inter, D, VCLA_EPI_SWIGLU, 0, 0, 0, 0, nullptr, 0.f, &L.vgu));
        RUN(gemm(ctx, s, w.act, c.t_inter, L.wd, nullptr, w.x, D, w.x, D, M, D, c.t_inter, VCLA_EPI_NONE, 0, 0, 0, 0, nullptr, 0.f, &L.vd, next_gamma,
                 c.t_eps, next_gamma ? w.h : nullptr, D));
    }
    return VCLA_OK;
}

static int llama_prefill_impl(vcla_ctx* ctx, const void* inputs_embeds, int B, int T, int pos0, void* kv_cache, int ctx_max,
                              const int32_t* key_mask, float* logits, int all_logits, void* ws, void* layer_tap, hipStream_t s);

extern "C" int vcla_llama_prefill(vcla_ctx* ctx, const void* inputs_embeds, int B, int T, int pos0, void* kv_cache,
                                  int ctx_max, const int32_t* key_mask, float* logits, int all_logits, void* ws,
                                  size_t ws_bytes, void* layer_tap, void* stream) {
    VCLA_REQUIRE(ctx && ctx->finalized && ctx->c.t_layers > 0, VCLA_ERR_BAD_ARG, "llama_prefill: context not finalized, or vision-only");
    VCLA_REQUIRE(inputs_embeds && kv_cache && ws, VCLA_ERR_BAD_ARG, "llama_prefill: null pointer");
    const vcla_model_cfg& c = ctx->c;
    VCLA_REQUIRE(B > 0 && T > 0 && pos0 >= 0 && pos0 + T <= ctx_max && ctx_max <= c.t_max_pos, VCLA_ERR_BAD_SHAPE,
                 "llama_prefill: B=%d T=%d pos0=%d ctx_max=%d (max_pos %d)", B, T, pos0, ctx_max, c.t_max_pos);
    VCLA_REQUIRE(ws_bytes >= vcla_llama_workspace_bytes(ctx, B, T), VCLA_ERR_WORKSPACE, "llama_prefill: workspace %zu < %zu bytes",
                 ws_bytes, vcla_llama_workspace_bytes(ctx, B, T));
    VCLA_REQUIRE(!(c.t_kv_fp8 && T > 1 && pos0 > 0), VCLA_ERR_BAD_ARG,
                 "llama_prefill: with the fp8 K/V cache a multi-token forward must start at position 0 (it attends over its own bf16 "
                 "rows; the 1-byte cache serves single-token steps) -- nothing was written");
    hipStream_t s = (hipStream_t)stream;
    if (layer_tap) return llama_prefill_impl(ctx, inputs_embeds, B, T, pos0, kv_cache, ctx_max, key_mask, logits, all_logits, ws, layer_tap, s);
    const void* const key[8] = {inputs_embeds, kv_cache, key_mask, logits, ws, (const void*)(((intptr_t)B << 32) | (uint32_t)T),
                                (const void*)(((intptr_t)pos0 << 32) | (uint32_t)ctx_max), (const void*)(intptr_t)(all_logits + 2 * c.t_fp8_mfma)};
    return run_macro(ctx->prefill_graph, key, s, [&](hipStream_t st) {
        return llama_prefill_impl(ctx, inputs_embeds, B, T, pos0, kv_cache, ctx_max, key_mask, logits, all_logits, ws, nullptr, st);
    });
}

static int llama_prefill_impl(vcla_ctx* ctx, const void* inputs_embeds, int B, int T, int pos0, void* kv_cache, int ctx_max,
                              const int32_t* key_mask, float* logits, int all_logits, void* ws, void* layer_tap, hipStream_t s) {
    const vcla_model_cfg& c = ctx->c;
    const int dt = c.act_dtype;
    const size_t e = esz(ctx);
    const int D = c.t_hidden, M = B * T;
    LlamaWs w;
    carve_llama(ctx, B, T, (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), &w);
    ctx->run.splitk_ws = w.splitk;
    struct Q8Scope { vcla_ctx* c; Q8Scope(vcla_ctx* c_, void* q, float* s_) : c(c_) { c->run.q8_ws = q; c->run.q8s_ws = s_; } ~Q8Scope() { c->run.q8_ws = nullptr; c->run.q8s_ws = nullptr; } } q8_scope(ctx, w.q8, w.q8s);
    VCLA_CHECK_HIP(hipMemcpyAsync(w.x, inputs_embeds, (size_t)M * D * e, hipMemcpyDeviceToDevice, s));
    for (int l = 0; l < c.t_layers; ++l) {
        // every layer's down_proj also emits the next norm into w.h (the final norm only when all rows need it)
        const float* next_gamma = l + 1 < c.t_layers ? ctx->llama[l + 1].ln1g : (all_logits ? ctx->norm_g : nullptr);
        RUN(llama_layer(ctx, s, ctx->llama[l], w, l, B, T, pos0, nullptr, kv_cache, ctx_max, key_mask, l > 0, next_gamma));
        RUN(tap_copy(layer_tap, l, w.x, (size_t)M * D * e, s));
    }
    const bool norm_fused = (dt == VCLA_F32) ? (M <= 8) : (M == 1);   // llama_layer's GEMV mode ignores next_gamma
    if (all_logits) {
        if (norm_fused) RUN(vcla_rmsnorm(w.x, D, ctx->norm_g, w.h, D, M, D, c.t_eps, dt, s));
        RUN(tap_copy(layer_tap, c.t_layers, w.h, (size_t)M * D * e, s));
        if (logits)
            RUN(gemm(ctx, s, w.h, D, ctx->lm_head, nullptr, nullptr, 0, logits, c.t_vocab, M, c.t_vocab, D, VCLA_EPI_NONE, 1));
    } else if (logits) {
        // last position of every sequence only: rows (b, T-1) -> hl [B, D]
        RUN(vcla_rmsnorm((char*)w.x + (size_t)(T - 1) * D * e, (int64_t)T * D, ctx->norm_g, w.hl, D, B, D, c.t_eps, dt, s));
        RUN(gemm(ctx, s, w.hl, D, ctx->lm_head, nullptr, nullptr, 0, logits, c.t_vocab, B, c.t_vocab, D, VCLA_EPI_NONE, 1));
    }
    return VCLA_OK;
}

// The persistent decode step serves B = 1 in the bf16 mode when its weight stream is registered, no fp8 decode copies are loaded (those
// steps read the 1-byte weights) and the device is the 256-CU part the row ownership is laid out for.  VCLA_ENGINE=0: the launch path.
static bool engine_step_ok(const vcla_ctx* ctx, int B) {
    const char* eng_s = getenv("VCLA_ENGINE");        // read per call: A/B runs and the tests flip it inside one process
    const int eng_env = eng_s ? atoi(eng_s) : 1;
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
        return n;
    }();
    return eng_env && B == 1 && ctx->eng_w && ctx->eng_g && ctx->c.act_dtype == VCLA_BF16 && !ctx->c.t_kv_fp8 && n_cu == EG_NCU &&
           !(ctx->llama[0].vqkv.q8 && ctx->llama[0].vqkv.s8);
}

static int decode_step_impl(vcla_ctx* ctx, hipStream_t s, const int64_t* ids_in, int B, int pos0, int32_t* pos_dev,
                            int advance_pos, void* kv_cache, int ctx_max, const int32_t* key_mask, float* logits,
                            int64_t* ids_out, const LlamaWs& w, const vcla_sample_args* samp = nullptr, int n_hist0 = 0,
                            bool skip_embed = false, int64_t* loop_ids_out = nullptr, int loop_step_base = 0, bool* tail_folded = nullptr) {
    const vcla_model_cfg& c = ctx->c;
    const int dt = c.act_dtype;
    const int D = c.t_hidden;
    struct Scope { vcla_ctx* c; explicit Scope(vcla_ctx* c_) : c(c_) { c->run.decode_step = true; } ~Scope() { c->run.decode_step = false; } } decode_scope(ctx);
    if (!skip_embed) RUN(vcla_embed_splice(ids_in, ctx->embed, nullptr, nullptr, w.x, B, 1, 0, D, c.t_vocab, dt, s));   // else: w.x was filled by post_select_kernel
    float* lg = logits ? logits : w.logits;
    if (engine_step_ok(ctx, B)) {
        // B = 1, bf16, LLaMA-7B geometry: the whole step (32 layers + lm_head) is ONE persistent launch (decode_engine.hip)
        vcla_engine_args ea{};
        ea.g = ctx->eng_geom; ea.stream = (const unsigned char*)ctx->eng_w; ea.gamma = ctx->eng_g; ea.x_in = (const bf16_t*)w.x;
        ea.kv = (bf16_t*)kv_cache; ea.ctx_max = ctx_max; ea.pos0 = pos0; ea.pos_dev = pos_dev; ea.key_mask = key_mask;
        ea.rope_cos = ctx->rope_cos; ea.rope_sin = ctx->rope_sin; ea.scale = 1.0f / sqrtf((float)(D / c.t_heads)); ea.eps = c.t_eps;
        ea.logits = lg; ea.mbox = (unsigned long long*)w.eng;
        // the decode LOOP without a sampler: argmax, token record, next embedding and position advance happen inside the launch (VCLA_ENGINE_TAIL=0: the
        // two separate launches, as the loop has done since round 3)
        const char* tail_s = getenv("VCLA_ENGINE_TAIL");
        bool fold = loop_ids_out && !samp && ids_out == w.ids && pos_dev && !logits && (tail_s ? atoi(tail_s) != 0 : true);
        if (fold) {
            ea.tail_ids_out = loop_ids_out; ea.tail_cur = w.ids; ea.tail_embed = (const bf16_t*)ctx->embed; ea.tail_x = (bf16_t*)w.x;
            ea.tail_pos = pos_dev; ea.tail_step_base = loop_step_base;
        }
        if (tail_folded) *tail_folded = fold;
        { const char* e_ = getenv("VCLA_ENGINE_SPLIT"); ea.split_min = e_ ? atoi(e_) : 512; }      // VCLA_ENGINE_SPLIT=0: one CU per head at every context
        { const char* e_ = getenv("VCLA_ENGINE_FAULT"); ea.fault = e_ ? atoi(e_) : 0; }            // test hook, see decode_engine.h
        if (const char* tl = getenv("VCLA_ENGINE_TL")) ea.timeline = (unsigned long long*)strtoull(tl, nullptr, 16);   // debug: tools/engine_probe.py --timeline
        RUN(vcla_engine_launch(&ea, s));
        if (fold) return VCLA_OK;
    } else {
    for (int l = 0; l < c.t_layers; ++l)   // batched mode: the norms ride on the producing GEMMs, the last one is the final norm
        RUN(llama_layer(ctx, s, ctx->llama[l], w, l, B, 1, pos0, pos_dev, kv_cache, ctx_max, key_mask, l > 0,
                        l + 1 < c.t_layers ? ctx->llama[l + 1].ln1g : ctx->norm_g));
    static const int ds_env = getenv("VCLA_DSTREAM") ? atoi(getenv("VCLA_DSTREAM")) : 1;
    const LlamaLayer& L0 = ctx->llama[0];
    const bool ds_layers = ds_env && dt == VCLA_BF16 && B >= 2 && B <= 64 && D % 32 == 0 && c.t_inter % 32 == 0 &&
                           ((L0.vqkv.frag && L0.vo.frag && L0.vgu.frag && L0.vd.frag) || (L0.vqkv.q8f && L0.vo.q8f && L0.vgu.q8f && L0.vd.q8f));
    if (ds_layers && (ctx->vlm.frag || ctx->vlm.q8f)) {
        // streaming layers leave the residual stream in w.x: final norm -> fragment-major, lm_head streamed the same way
        static const int defer_env = getenv("VCLA_DS_DEFER") ? atoi(getenv("VCLA_DS_DEFER")) : 1;
        const bool defer = defer_env != 0 && D % 16 == 0;     // the last down_proj left gamma_final * x and its row statistics in w.h / w.ssq
        if (!defer) RUN(vcla_rmsnorm_pack(w.x, D, ctx->norm_g, w.h, B, D, c.t_eps, s));
        RUN(gemm_ds(ctx, s, w.h, ctx->lm_head, ctx->vlm, nullptr, 0, lg, c.t_vocab, nullptr, B, c.t_vocab, D, VCLA_EPI_NONE, 1,
                    defer ? w.ssq : nullptr, ctx->run.ssq_parts));
    } else if (ds_layers) {
        RUN(vcla_rmsnorm(w.x, D, ctx->norm_g, w.h, D, B, D, c.t_eps, dt, s));
        RUN(gemm(ctx, s, w.h, D, ctx->lm_head, nullptr, nullptr, 0, lg, c.t_vocab, B, c.t_vocab, D, VCLA_EPI_NONE, 1, 0, 0, 0, nullptr, 0.f, &ctx->vlm));
    } else if ((dt == VCLA_F32) ? (B <= 8) : (B == 1)) {
        RUN(gemm(ctx, s, w.x, D, ctx->lm_head, nullptr, nullptr, 0, lg, c.t_vocab, B, c.t_vocab, D, VCLA_EPI_NONE, 1, 0, 0, 0, ctx->norm_g, c.t_eps, &ctx->vlm));
    } else {
        RUN(gemm(ctx, s, w.h, D, ctx->lm_head, nullptr, nullptr, 0, lg, c.t_vocab, B, c.t_vocab, D, VCLA_EPI_NONE, 1, 0, 0, 0, nullptr, 0.f, &ctx->vlm));
    }
    }
    if (ids_out && samp) RUN(vcla_sample_launch(lg, c.t_vocab, B, c.t_vocab, n_hist0, pos_dev, samp, ids_out, s));
    else if (ids_out) RUN(vcla_argmax(lg, c.t_vocab, ids_out, B, c.t_vocab, s));
    if (advance_pos && pos_dev) {
        advance_pos_kernel<<<1, 1, 0, s>>>(pos_dev);
        VCLA_CHECK_LAUNCH("advance_pos_kernel");
    }
    return VCLA_OK;
}

static int check_decode_args(vcla_ctx* ctx, const int64_t* ids_in, int B, int pos0, const int32_t* pos_dev, void* kv_cache,
                             int ctx_max, void* ws, size_t ws_bytes) {
    VCLA_REQUIRE(ctx && ctx->finalized && ctx->c.t_layers > 0, VCLA_ERR_BAD_ARG, "llama_decode: context not finalized, or vision-only");
    VCLA_REQUIRE(ids_in && kv_cache && ws, VCLA_ERR_BAD_ARG, "llama_decode: null pointer");
    VCLA_REQUIRE(B > 0 && pos0 >= 0 && ctx_max <= ctx->c.t_max_pos && (pos_dev || pos0 < ctx_max), VCLA_ERR_BAD_SHAPE,
                 "llama_decode: B=%d pos0=%d ctx_max=%d (max_pos %d)", B, pos0, ctx_max, ctx->c.t_max_pos);
    VCLA_REQUIRE(ws_bytes >= vcla_llama_workspace_bytes(ctx, B, 1), VCLA_ERR_WORKSPACE, "llama_decode: workspace %zu < %zu bytes",
                 ws_bytes, vcla_llama_workspace_bytes(ctx, B, 1));
    return VCLA_OK;
}

extern "C" int vcla_llama_decode_step(vcla_ctx* ctx, const int64_t* ids_in, int B, int pos0, int32_t* pos_dev,
                                      int advance_pos, void* kv_cache, int ctx_max, const int32_t* key_mask,
                                      float* logits, int64_t* ids_out, void* ws, size_t ws_bytes, void* stream) {
    RUN(check_decode_args(ctx, ids_in, B, pos0, pos_dev, kv_cache, ctx_max, ws, ws_bytes));
    LlamaWs w;
    carve_llama(ctx, B, 1, (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), &w);
    ctx->run.splitk_ws = w.splitk;
    if (engine_step_ok(ctx, B)) VCLA_CHECK_HIP(hipMemsetAsync(w.eng, 0, EG_WS_BYTES, (hipStream_t)stream));
    return decode_step_impl(ctx, (hipStream_t)stream, ids_in, B, pos0, pos_dev, advance_pos, kv_cache, ctx_max, key_mask,
                            logits, ids_out, w);
}

extern "C" int vcla_llama_decode_loop(vcla_ctx* ctx, const int64_t* ids_in, int B, int pos0, int32_t* pos_dev, int n_steps,
                                      void* kv_cache, int ctx_max, const int32_t* key_mask, int64_t* ids_out, void* ws,
                                      size_t ws_bytes, int use_graph, void* stream) {
    return vcla_llama_decode_loop_sampled(ctx, ids_in, B, pos0, pos_dev, n_steps, kv_cache, ctx_max, key_mask, ids_out, ws, ws_bytes,
                                          use_graph, nullptr, 0, stream);
}

extern "C" int vcla_llama_decode_loop_sampled(vcla_ctx* ctx, const int64_t* ids_in, int B, int pos0, int32_t* pos_dev, int n_steps,
                                              void* kv_cache, int ctx_max, const int32_t* key_mask, int64_t* ids_out, void* ws,
                                              size_t ws_bytes, int use_graph, const vcla_sample_args* sampling, int n_hist0,
                                              void* stream) {
    RUN(check_decode_args(ctx, ids_in, B, pos0, pos_dev, kv_cache, ctx_max, ws, ws_bytes));
    VCLA_REQUIRE(pos_dev && ids_out && n_steps >= 0, VCLA_ERR_BAD_ARG, "llama_decode_loop: needs pos_dev, ids_out, n_steps >= 0");
    VCLA_REQUIRE(!sampling || (sampling->history && n_hist0 >= 0), VCLA_ERR_BAD_ARG, "llama_decode_loop: sampling needs the history buffer");
    if (n_steps == 0) return VCLA_OK;
    hipStream_t s = (hipStream_t)stream;
    LlamaWs w;
    carve_llama(ctx, B, 1, (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), &w);
    ctx->run.splitk_ws = w.splitk;
    // w.ids holds the current token of every sequence; each step consumes it and overwrites it with the argmax.
    VCLA_CHECK_HIP(hipMemcpyAsync(w.ids, ids_in, (size_t)B * 8, hipMemcpyDeviceToDevice, s));
    // step_base: value of *pos_dev at the first step is unknown to the host -> the caller passes pos0 as the absolute
    // position of the first decoded token and keeps *pos_dev == 0 at entry (documented in INTEGRATION.md).
    const int step_base = 0;
    const vcla_model_cfg& c = ctx->c;
    // The first step's decoder input is embedded here; every later one by the post_select launch that ends the step before it
    // (record the ids, embed them, advance the position: one launch instead of three per step).
    RUN(vcla_embed_splice(w.ids, ctx->embed, nullptr, nullptr, w.x, B, 1, 0, c.t_hidden, c.t_vocab, c.act_dtype, s));
    VCLA_CHECK_HIP(hipMemsetAsync(w.ticket, 0, 4, s));
    // mailboxes + launch sequence of the persistent step: zero ONCE per loop (the launches tag their granules with the sequence number they
    // read from the state words and advance it themselves, so the replayed graph needs no memset node)
    if (engine_step_ok(ctx, B)) VCLA_CHECK_HIP(hipMemsetAsync(w.eng, 0, EG_WS_BYTES, s));
    auto one_step = [&](hipStream_t st) -> int {
        bool folded = false;
        RUN(decode_step_impl(ctx, st, w.ids, B, pos0, pos_dev, 0, kv_cache, ctx_max, key_mask, nullptr, w.ids, w, sampling, n_hist0, /*skip_embed=*/true,
                             ids_out, step_base, &folded));
        if (folded) return VCLA_OK;            // (the persistent B = 1 step did the token bookkeeping itself)
        if (c.act_dtype == VCLA_BF16 && c.t_hidden % 8 == 0)
            post_select_kernel<bf16_t><<<B, 256, 0, st>>>(w.ids, ids_out, pos_dev, step_base, B, (const bf16_t*)ctx->embed, (bf16_t*)w.x, c.t_hidden, c.t_vocab, w.ticket);
        else if (c.act_dtype == VCLA_BF16)
            return vcla_fail(VCLA_ERR_BAD_SHAPE, "llama_decode_loop: hidden size %d is not a multiple of 8", c.t_hidden);
        else
            post_select_kernel<float><<<B, 256, 0, st>>>(w.ids, ids_out, pos_dev, step_base, B, (const bf16_t*)ctx->embed, (float*)w.x, c.t_hidden, c.t_vocab, w.ticket);
        VCLA_CHECK_LAUNCH("post_select_kernel");
        return VCLA_OK;
    };
    hipStreamCaptureStatus outer = hipStreamCaptureStatusNone;   // the caller captures this stream itself: the steps belong in its graph
    if (use_graph && s != nullptr && hipStreamIsCapturing(s, &outer) != hipSuccess) { (void)hipGetLastError(); outer = hipStreamCaptureStatusNone; }
    if (!use_graph || s == nullptr || outer != hipStreamCaptureStatusNone) {  // stream capture is illegal on the legacy default stream
        for (int i = 0; i < n_steps; ++i) RUN(one_step(s));
        return VCLA_OK;
    }
    // hipGraph path: capture one step once per (buffers, shapes) key, replay n_steps times.
    auto& k = ctx->graph_key;
    const bool same = ctx->graph_exec && k.ids == (const void*)w.ids && k.kv == kv_cache && k.mask == (const void*)key_mask &&
                      k.ws == ws && k.out == (const void*)ids_out && k.B == B && k.pos0 == pos0 && k.ctx_max == ctx_max &&
                      k.pos_dev == (const void*)pos_dev && k.step_base == step_base && k.has_samp == (sampling != nullptr) && k.engine == (int)engine_step_ok(ctx, B) &&
                      (!sampling || (k.n_hist0 == n_hist0 && memcmp(&k.samp, sampling, sizeof(*sampling)) == 0));
    // several steps per graph launch (VCLA_GRAPH_STEPS=4): measured EQUAL to one step per launch on MI355X (B = 1: 359.7 vs 359.3
    // tok/s; the ~9 us between graph launches seen in round 1 are gone with the shorter step tail) -> off by default
    static const int gsteps_env = getenv("VCLA_GRAPH_STEPS") ? atoi(getenv("VCLA_GRAPH_STEPS")) : 1;
    const int G = gsteps_env > 1 ? gsteps_env : 1;
    auto capture = [&](int count, hipGraphExec_t* out) -> int {
        hipGraph_t graph = nullptr;
        VCLA_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int rc = VCLA_OK;
        for (int i = 0; i < count && !rc; ++i) rc = one_step(s);
        hipError_t ce = hipStreamEndCapture(s, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (ce != hipSuccess) return vcla_fail(VCLA_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
        hipError_t ie = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) { *out = nullptr; return vcla_fail(VCLA_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie)); }
        return VCLA_OK;
    };
    if (!same) {
        if (ctx->graph_exec) { (void)hipGraphExecDestroy(ctx->graph_exec); ctx->graph_exec = nullptr; }
        if (ctx->graph_exec_multi) { (void)hipGraphExecDestroy(ctx->graph_exec_multi); ctx->graph_exec_multi = nullptr; }
        RUN(capture(1, &ctx->graph_exec));
        k.ids = w.ids; k.kv = kv_cache; k.mask = key_mask; k.ws = ws; k.out = ids_out; k.B = B; k.pos0 = pos0;
        k.ctx_max = ctx_max; k.pos_dev = pos_dev; k.step_base = step_base;
        k.has_samp = sampling != nullptr; k.n_hist0 = n_hist0; k.engine = (int)engine_step_ok(ctx, B);
        if (sampling) memcpy(&k.samp, sampling, sizeof(*sampling));
    }
    int left = n_steps;
    if (G > 1 && left >= G) {
        if (!ctx->graph_exec_multi) RUN(capture(G, &ctx->graph_exec_multi));
        for (; left >= G; left -= G) VCLA_CHECK_HIP(hipGraphLaunch(ctx->graph_exec_multi, s));
    }
    for (; left > 0; --left) VCLA_CHECK_HIP(hipGraphLaunch(ctx->graph_exec, s));
    return VCLA_OK;
}

// The persistent decode step bounds every wait; a wait that ran out leaves (site code | CU << 16) in the workspace.  Synchronises the stream.
extern "C" int vcla_llama_decode_status(vcla_ctx* ctx, int B, const void* ws, size_t ws_bytes, void* stream) {
    VCLA_REQUIRE(ctx && ctx->finalized && ctx->c.t_layers > 0 && ws && B > 0, VCLA_ERR_BAD_ARG, "llama_decode_status: bad arguments");
    VCLA_REQUIRE(ws_bytes >= vcla_llama_workspace_bytes(ctx, B, 1), VCLA_ERR_WORKSPACE, "llama_decode_status: workspace too small");
    VCLA_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (!engine_step_ok(ctx, B)) return VCLA_OK;
    LlamaWs w;
    carve_llama(ctx, B, 1, (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), &w);
    unsigned st[2] = {0, 0};
    VCLA_CHECK_HIP(hipMemcpy(st, (const char*)w.eng + EG_WS_STATE_OFF, sizeof st, hipMemcpyDeviceToHost));
    if (st[1]) return vcla_fail(VCLA_ERR_HIP, "decode engine: a wait timed out (site 0x%x on CU %u after %u launches); its output is invalid.  The persistent launch needs all %d CUs to itself "
                                                   "(one workgroup each, co-resident): on a shared or partitioned GPU set VCLA_ENGINE=0 to decode on the per-operator launches",
                               st[1] & 0xffffu, st[1] >> 16, st[0], EG_NCU);
    return VCLA_OK;
}
