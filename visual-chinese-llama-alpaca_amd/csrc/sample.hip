// sample.hip -- next-row N2: the step after the logits, on the device.  One launch per decode step replaces the chain of
// torch ops the reference's default generation config (models/visualcla/modeling_utils.py:36-47: do_sample, top_p .9,
// top_k 40, temperature .5, repetition_penalty 1.1, no_repeat_ngram_size 15) runs through HF's logits processors:
//   RepetitionPenaltyLogitsProcessor -> NoRepeatNGramLogitsProcessor -> MinNewTokensLengthLogitsProcessor ->
//   TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> softmax -> one draw
// (hf:generation/logits_process.py, order of hf:generation/utils.py _get_logits_processor).  The draw is the inverse CDF of
// the kept set (descending probability, ties by index) at a caller-supplied uniform, so a run is a pure function of
// (logits, history, uniforms).  One 1024-thread workgroup per sequence; the 200 KB fp32 logits row stays in L2 across the
// passes.  HBM/L2-bound integer + compare work; no GEMM shape anywhere.
#include "vcla_common.h"

#define SAMPLE_THREADS 1024
#define SAMPLE_CAP 512       // kept-set capacity: top_k <= 256 plus ties at the k-th value
#define SAMPLE_MAX_HIST 4096 // generated tokens visible to the repetition / n-gram processors (max_position_embeddings = 2048)
#define SAMPLE_PER_THREAD 52 // logits per thread held in registers (no spills at 128 VGPRs): V <= 53248
#define SAMPLE_BIN_COPIES 8  // histogram replicas: spreads same-bin LDS atomics of neighbouring lanes

// order-preserving float -> unsigned key (ascending)
__device__ __forceinline__ unsigned fkey(float x) {
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct SampleParams {
    float repetition_penalty;
    int no_repeat_ngram;
    int min_new_tokens;
    int n_eos;
    int eos[VCLA_SAMPLE_MAX_EOS];
    float temperature;
    int top_k;
    float top_p_complement;   // (float)(1 - top_p): the fp32 threshold HF's `cumulative_probs <= 1 - top_p` compares against
    int use_top_p;
    int min_keep;
    const float* uniforms;
    const int64_t* history;
    int64_t* kept_ids;
    float* kept_probs;
    int32_t* n_kept;
};

static_assert(SAMPLE_THREADS * SAMPLE_PER_THREAD == VCLA_SAMPLE_MAX_VOCAB, "register slice must cover the largest vocabulary");

__global__ __launch_bounds__(SAMPLE_THREADS) void sample_kernel(float* __restrict__ logits, int64_t ld, int V, int B, int n_hist,
                                                                const int32_t* __restrict__ n_hist_dev, SampleParams a,
                                                                int64_t* __restrict__ out) {
    __shared__ int s_hist[SAMPLE_MAX_HIST];
    __shared__ unsigned s_bins[256 * SAMPLE_BIN_COPIES];
    __shared__ float s_cv[SAMPLE_CAP], s_sv[SAMPLE_CAP];
    __shared__ int s_ci[SAMPLE_CAP], s_si[SAMPLE_CAP];
    __shared__ unsigned s_prefix;
    __shared__ int s_kk, s_ncand;
    const int tid = threadIdx.x, b = blockIdx.x;
    float* x = logits + (int64_t)b * ld;
    int h = n_hist + (n_hist_dev ? *n_hist_dev : 0);
    h = h < 0 ? 0 : (h > SAMPLE_MAX_HIST ? SAMPLE_MAX_HIST : h);
    for (int i = tid; i < h; i += SAMPLE_THREADS) s_hist[i] = (int)a.history[(int64_t)i * B + b];
    __syncthreads();

    // ---- RepetitionPenaltyLogitsProcessor: gather / rescale / scatter, i.e. once per DISTINCT generated token
    if (a.repetition_penalty != 1.f) {
        for (int i = tid; i < h; i += SAMPLE_THREADS) {
            const int tok = s_hist[i];
            bool first = tok >= 0 && tok < V;
            for (int j = 0; j < i && first; ++j) first = s_hist[j] != tok;
            if (first) {
                const float v = x[tok];
                x[tok] = v < 0.f ? v * a.repetition_penalty : v / a.repetition_penalty;
            }
        }
        __syncthreads();
    }
    // ---- NoRepeatNGramLogitsProcessor: ban every token that followed an earlier copy of the last n-1 tokens
    const int n = a.no_repeat_ngram;
    if (n > 0 && h + 1 >= n) {
        const int p0 = h + 1 - n;   // the current (n-1)-token prefix is s_hist[p0 .. h)
        for (int i = tid; i + n <= h; i += SAMPLE_THREADS) {
            bool match = true;
            for (int t = 0; t < n - 1 && match; ++t) match = s_hist[i + t] == s_hist[p0 + t];
            const int tok = s_hist[i + n - 1];
            if (match && tok >= 0 && tok < V) x[tok] = -INFINITY;
        }
    }
    // ---- MinNewTokensLengthLogitsProcessor
    if (tid < a.n_eos && h < a.min_new_tokens && a.eos[tid] >= 0 && a.eos[tid] < V) x[a.eos[tid]] = -INFINITY;
    __syncthreads();

    // ---- TemperatureLogitsWarper on read: this thread's slice of the row goes to registers once (64 loads in flight),
    //      the radix passes below never touch memory again
    const float T = a.temperature;
    float y[SAMPLE_PER_THREAD];
#pragma unroll
    for (int j = 0; j < SAMPLE_PER_THREAD; ++j) {
        const int i = tid + j * SAMPLE_THREADS;
        y[j] = i < V ? x[i] / T : -INFINITY;
    }
    // ---- TopKLogitsWarper: exact k-th largest key by 4 x 8-bit radix select
    if (tid == 0) { s_prefix = 0u; s_kk = a.top_k < V ? a.top_k : V; s_ncand = 0; }
    unsigned mask = 0u;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256 * SAMPLE_BIN_COPIES; i += SAMPLE_THREADS) s_bins[i] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix;
#pragma unroll
        for (int j = 0; j < SAMPLE_PER_THREAD; ++j) {
            const unsigned key = fkey(y[j]);
            if (tid + j * SAMPLE_THREADS < V && (key & mask) == prefix)
                atomicAdd(&s_bins[((key >> shift) & 255u) * SAMPLE_BIN_COPIES + (tid & (SAMPLE_BIN_COPIES - 1))], 1u);
        }
        __syncthreads();
        if (tid < 256) {
            unsigned c = 0;
#pragma unroll
            for (int r = 0; r < SAMPLE_BIN_COPIES; ++r) c += s_bins[tid * SAMPLE_BIN_COPIES + r];
            s_bins[tid * SAMPLE_BIN_COPIES] = c;
        }
        __syncthreads();
        if (tid == 0) {
            int kk = s_kk, d = 255;
            for (; d > 0; --d) {
                const int c = (int)s_bins[d * SAMPLE_BIN_COPIES];
                if (c >= kk) break;
                kk -= c;
            }
            s_kk = kk;
            s_prefix = prefix | ((unsigned)d << shift);
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    // ---- kept candidates: everything >= the k-th value (HF removes `scores < kth`, so ties stay), -inf never
    const unsigned thr = s_prefix;
#pragma unroll
    for (int j = 0; j < SAMPLE_PER_THREAD; ++j) {
        if (fkey(y[j]) >= thr && y[j] > -INFINITY) {   // slots past V hold -inf
            const int slot = atomicAdd(&s_ncand, 1);
            if (slot < SAMPLE_CAP) { s_cv[slot] = y[j]; s_ci[slot] = tid + j * SAMPLE_THREADS; }
        }
    }
    __syncthreads();
    const int nc = s_ncand < SAMPLE_CAP ? s_ncand : SAMPLE_CAP;
    // rank sort, descending value; ties in DESCENDING index = the reverse of the stable ascending sort HF's top-p cut walks
    // (torch.sort on the scores), so a cut that falls inside a group of equal scores drops the lower token ids first
    for (int t = tid; t < nc; t += SAMPLE_THREADS) {
        const float v = s_cv[t];
        const int idx = s_ci[t];
        int rank = 0;
        for (int j = 0; j < nc; ++j) rank += (s_cv[j] > v) || (s_cv[j] == v && s_ci[j] > idx);
        s_sv[rank] = v;
        s_si[rank] = idx;
    }
    __syncthreads();
    if (nc == 0) {   // every logit is -inf / NaN: nothing to draw from
        if (tid == 0) { out[b] = 0; if (a.n_kept) a.n_kept[b] = 0; }
        return;
    }
    const float vmax = s_sv[0];
    for (int t = tid; t < nc; t += SAMPLE_THREADS) s_cv[t] = expf(s_sv[t] - vmax);
    __syncthreads();
    if (tid == 0) {
        // ---- TopPLogitsWarper: ascending cumulative probability <= 1 - top_p is cut, at least min_keep survive
        int keep = nc;
        if (a.use_top_p) {
            float Z = 0.f;
            for (int r = nc - 1; r >= 0; --r) Z += s_cv[r];
            float c = 0.f;
            keep = 0;
            for (int r = nc - 1; r >= 0; --r) {
                c += s_cv[r] / Z;
                if (c > a.top_p_complement) { keep = r + 1; break; }
            }
            if (keep < a.min_keep) keep = a.min_keep < nc ? a.min_keep : nc;
        }
        s_kk = keep;
    }
    __syncthreads();
    const int keep = s_kk;
    // ---- the draw walks the survivors in descending probability with ties in ASCENDING token id (so top_k = 1 is
    //      torch.argmax: first maximum): re-rank inside the kept prefix
    for (int t = tid; t < keep; t += SAMPLE_THREADS) {
        const float v = s_sv[t];
        const int idx = s_si[t];
        int rank = 0;
        for (int j = 0; j < keep; ++j) rank += (s_sv[j] > v) || (s_sv[j] == v && s_si[j] < idx);
        s_ci[rank] = idx;
    }
    __syncthreads();
    // s_cv (exp of the value) is constant inside a tie group, so it is already in the final order
    if (tid == 0) {
        float Zk = 0.f;
        for (int r = 0; r < keep; ++r) Zk += s_cv[r];
        const float u = a.uniforms ? a.uniforms[(int64_t)h * B + b] : 0.f;
        const float target = u * Zk;
        float c = 0.f;
        int pick = keep - 1;
        for (int r = 0; r < keep; ++r) {
            c += s_cv[r];
            if (c > target) { pick = r; break; }
        }
        out[b] = s_ci[pick];
        if (a.n_kept) a.n_kept[b] = keep;
        s_sv[0] = Zk;
    }
    if (a.kept_ids || a.kept_probs) {
        __syncthreads();
        const float Zk = s_sv[0];
        for (int t = tid; t < keep; t += SAMPLE_THREADS) {
            if (a.kept_ids) a.kept_ids[(int64_t)b * SAMPLE_CAP + t] = s_ci[t];
            if (a.kept_probs) a.kept_probs[(int64_t)b * SAMPLE_CAP + t] = s_cv[t] / Zk;
        }
    }
}

int vcla_sample_launch(float* logits, int64_t ld, int B, int V, int n_hist, const int32_t* n_hist_dev, const vcla_sample_args* a,
                       int64_t* out, hipStream_t s) {
    VCLA_REQUIRE(logits && a && out, VCLA_ERR_BAD_ARG, "sample: null pointer");
    VCLA_REQUIRE(B >= 0 && V > 0 && V <= VCLA_SAMPLE_MAX_VOCAB && ld >= V && n_hist >= 0, VCLA_ERR_BAD_SHAPE,
                 "sample: B=%d V=%d (max %d) ld=%lld n_hist=%d", B, V, VCLA_SAMPLE_MAX_VOCAB, (long long)ld, n_hist);
    VCLA_REQUIRE(a->top_k >= 1 && a->top_k <= VCLA_SAMPLE_MAX_TOP_K, VCLA_ERR_BAD_ARG, "sample: top_k=%d outside [1, %d]", a->top_k, VCLA_SAMPLE_MAX_TOP_K);
    VCLA_REQUIRE(a->temperature > 0.f && a->repetition_penalty > 0.f && a->top_p > 0.0 && a->top_p <= 1.0 && a->no_repeat_ngram_size >= 0 &&
                 a->min_tokens_to_keep >= 1 && a->n_eos >= 0 && a->n_eos <= VCLA_SAMPLE_MAX_EOS, VCLA_ERR_BAD_ARG,
                 "sample: temperature=%g repetition_penalty=%g top_p=%g no_repeat_ngram_size=%d min_tokens_to_keep=%d n_eos=%d", a->temperature,
                 a->repetition_penalty, a->top_p, a->no_repeat_ngram_size, a->min_tokens_to_keep, a->n_eos);
    VCLA_REQUIRE(a->history || (n_hist == 0 && !n_hist_dev), VCLA_ERR_BAD_ARG, "sample: history is NULL but tokens were generated");
    if (B == 0) return VCLA_OK;
    SampleParams p;
    p.repetition_penalty = a->repetition_penalty;
    p.no_repeat_ngram = a->no_repeat_ngram_size;
    p.min_new_tokens = a->min_new_tokens;
    p.n_eos = a->n_eos;
    for (int i = 0; i < VCLA_SAMPLE_MAX_EOS; ++i) p.eos[i] = i < a->n_eos ? a->eos_ids[i] : -1;
    p.temperature = a->temperature;
    p.top_k = a->top_k;
    p.top_p_complement = (float)(1.0 - a->top_p);
    p.use_top_p = a->top_p < 1.0;
    p.min_keep = a->min_tokens_to_keep;
    p.uniforms = a->uniforms;
    p.history = a->history;
    p.kept_ids = a->kept_ids;
    p.kept_probs = a->kept_probs;
    p.n_kept = a->n_kept;
    sample_kernel<<<B, SAMPLE_THREADS, 0, s>>>(logits, ld, V, B, n_hist, n_hist_dev, p, out);
    VCLA_CHECK_LAUNCH("sample_kernel");
    return VCLA_OK;
}

extern "C" int vcla_sample(float* logits, int64_t ld, int B, int V, int n_hist, const int32_t* n_hist_dev, const vcla_sample_args* args,
                           int64_t* ids_out, void* stream) {
    return vcla_sample_launch(logits, ld, B, V, n_hist, n_hist_dev, args, ids_out, (hipStream_t)stream);
}
