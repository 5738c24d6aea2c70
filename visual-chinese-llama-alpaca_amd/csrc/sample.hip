// sample.hip -- next-row N2: the step after the logits, on the device.  One launch per decode step replaces the chain of
// torch ops the reference's default generation config (models/visualcla/modeling_utils.py:36-47: do_sample, top_p .9,
// top_k 40, temperature .5, repetition_penalty 1.1, no_repeat_ngram_size 15) runs through HF's logits processors:
//   RepetitionPenaltyLogitsProcessor -> NoRepeatNGramLogitsProcessor -> MinNewTokensLengthLogitsProcessor ->
//   TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> softmax -> one draw
// (hf:generation/logits_process.py, order of hf:generation/utils.py _get_logits_processor).  The draw is the inverse CDF of
// the kept set (descending probability, ties by token id) at a caller-supplied uniform, so a run is a pure function of
// (logits, history, uniforms).  One 1024-thread workgroup per sequence; after the penalties the 200 KB fp32 logits row is
// read ONCE into registers (52 order-preserving keys per thread) and every later pass is register compares.  Integer +
// compare work, bound by one CU's issue rate and a handful of barriers (42 us per step); no GEMM shape anywhere.
#include "vcla_common.h"

#define SAMPLE_THREADS 1024
#define SAMPLE_CAP 512       // kept-set capacity: top_k <= 256 plus ties at the k-th value
#define SAMPLE_MAX_HIST 4096 // generated tokens visible to the repetition / n-gram processors (max_position_embeddings = 2048)
#define SAMPLE_PER_THREAD 52 // logits per thread held in registers (no spills at 128 VGPRs): V <= 53248

// order-preserving float -> unsigned key (ascending)
__device__ __forceinline__ unsigned fkey(float x) {
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float funkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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
    __shared__ unsigned s_seen[VCLA_SAMPLE_MAX_VOCAB / 32];   // one bit per token id: "already penalised"
    __shared__ int s_cnt[3];
    __shared__ float s_cv[SAMPLE_CAP], s_sv[SAMPLE_CAP];
    __shared__ int s_ci[SAMPLE_CAP], s_si[SAMPLE_CAP];
    __shared__ int s_kk, s_ncand;
    const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63;
    float* x = logits + (int64_t)b * ld;
    int h = n_hist + (n_hist_dev ? *n_hist_dev : 0);
    h = h < 0 ? 0 : (h > SAMPLE_MAX_HIST ? SAMPLE_MAX_HIST : h);
    for (int i = tid; i < h; i += SAMPLE_THREADS) s_hist[i] = (int)a.history[(int64_t)i * B + b];
    const bool penalise = a.repetition_penalty != 1.f && h > 0;
    if (penalise)
        for (int i = tid; i < VCLA_SAMPLE_MAX_VOCAB / 32; i += SAMPLE_THREADS) s_seen[i] = 0u;
    if (tid < 3) s_cnt[tid] = 0;
    if (tid == 0) s_ncand = 0;
    __syncthreads();

    // ---- RepetitionPenaltyLogitsProcessor: gather / rescale / scatter, i.e. once per DISTINCT generated token -- the
    //      thread whose atomicOr flips the token's bit is the one that applies it
    if (penalise) {
        for (int i = tid; i < h; i += SAMPLE_THREADS) {
            const int tok = s_hist[i];
            if (tok < 0 || tok >= V) continue;
            const unsigned bit = 1u << (tok & 31);
            if (!(atomicOr(&s_seen[tok >> 5], bit) & bit)) {
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

    // ---- this thread's slice of the row goes to registers once (all loads in flight) as order-preserving keys of the
    //      RAW scores; nothing below touches memory again.  Selection runs on the raw scores because dividing by the
    //      temperature is monotone; the division itself is applied to the few candidates only.
    unsigned key[SAMPLE_PER_THREAD];
#pragma unroll
    for (int j = 0; j < SAMPLE_PER_THREAD; ++j) {
        const int i = tid + j * SAMPLE_THREADS;
        key[j] = i < V ? fkey(x[i]) : 0u;   // 0 sorts below every real key (fkey(-inf) = 0x007fffff)
    }
    // ---- TopKLogitsWarper: bisection on the key for the largest t with count(key >= t) >= k.  A pass is 52 register
    //      compares + a wave reduction + one LDS atomic per wave; it stops as soon as the survivors are few enough to sort
    //      (k <= count <= 2k, at least 64), typically after ~12 passes: 8 fix the exponent, a few the mantissa.
    const int k = a.top_k < V ? a.top_k : V;
    const int few = 2 * k > 64 ? 2 * k : 64;   // <= SAMPLE_CAP since k <= 256
    unsigned lo = 1u, hi = 0xffffffffu;        // count(key >= 1) = V >= k
    for (int pass = 0; lo < hi; ++pass) {
        const unsigned mid = lo + ((hi - lo) >> 1) + ((hi - lo) & 1u);
        int c = 0;
#pragma unroll
        for (int j = 0; j < SAMPLE_PER_THREAD; ++j) c += key[j] >= mid;
        c = wave_sum_i(c);
        if (lane == 0) atomicAdd(&s_cnt[pass % 3], c);
        if (tid == 0) s_cnt[(pass + 1) % 3] = 0;   // last read two passes ago
        __syncthreads();
        const int total = s_cnt[pass % 3];
        if (total >= k) {
            lo = mid;
            if (total <= few) break;
        } else {
            hi = mid - 1u;
        }
    }
    // ---- candidates: every key >= lo - 4 ulp (a superset of the top k: x1 < x2 can still give x1 / T == x2 / T, and such
    //      a tie with the k-th divided score must survive like any other); -inf never.  Wave-aggregated compaction.
    const unsigned lo_c = lo > 0x00800004u ? lo - 4u : 0x00800000u;
#pragma unroll
    for (int j = 0; j < SAMPLE_PER_THREAD; ++j) {
        const bool take = key[j] >= lo_c && key[j] > 0x007fffffu;
        const unsigned long long m = __ballot(take);
        if (m) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_ncand, __popcll(m));
            base = __shfl(base, 0, 64);
            const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
            if (take && slot < SAMPLE_CAP) { s_cv[slot] = funkey(key[j]) / a.temperature; s_ci[slot] = tid + j * SAMPLE_THREADS; }
        }
    }
    __syncthreads();
    if (s_ncand > SAMPLE_CAP) {
        // Rare: more than SAMPLE_CAP candidates (a run of equal scores at the k-th value; HF would keep all of them).  The
        // arrival-order compaction above would then drop an arbitrary subset, so redo it deterministically: every key
        // strictly above the threshold first (fewer than k of those, by construction of the bisection), then the keys tied
        // around the threshold in ASCENDING token id until the capacity is reached.
        __shared__ int s_wcnt[SAMPLE_THREADS / 64];
        __syncthreads();
        if (tid == 0) s_ncand = 0;
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < SAMPLE_PER_THREAD; ++j) {
            const bool take = key[j] > lo;
            const unsigned long long m = __ballot(take);
            if (m) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_ncand, __popcll(m));
                base = __shfl(base, 0, 64);
                const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                if (take) { s_cv[slot] = funkey(key[j]) / a.temperature; s_ci[slot] = tid + j * SAMPLE_THREADS; }
            }
        }
        __syncthreads();
        int filled = s_ncand;                      // < k <= SAMPLE_CAP / 2, identical in every thread
#pragma unroll 1
        for (int j = 0; j < SAMPLE_PER_THREAD && filled < SAMPLE_CAP; ++j) {   // token id = tid + j * 1024: ascending in (j, tid)
            const bool take = key[j] <= lo && key[j] >= lo_c && key[j] > 0x007fffffu;
            const unsigned long long m = __ballot(take);
            if (lane == 0) s_wcnt[tid >> 6] = __popcll(m);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < SAMPLE_THREADS / 64; ++w) { const int c = s_wcnt[w]; before += w < (tid >> 6) ? c : 0; total += c; }
            const int slot = filled + before + __popcll(m & ((1ull << lane) - 1ull));
            if (take && slot < SAMPLE_CAP) { s_cv[slot] = funkey(key[j]) / a.temperature; s_ci[slot] = tid + j * SAMPLE_THREADS; }
            filled += total;
            __syncthreads();
        }
        if (tid == 0) s_ncand = filled < SAMPLE_CAP ? filled : SAMPLE_CAP;
        __syncthreads();
    }
    const int nc0 = s_ncand < SAMPLE_CAP ? s_ncand : SAMPLE_CAP;
    // rank sort, descending value; ties in DESCENDING index = the reverse of the stable ascending sort HF's top-p cut walks
    // (torch.sort on the scores), so a cut that falls inside a group of equal scores drops the lower token ids first
    for (int t = tid; t < nc0; t += SAMPLE_THREADS) {
        const float v = s_cv[t];
        const int idx = s_ci[t];
        int rank = 0;
        for (int j = 0; j < nc0; ++j) rank += (s_cv[j] > v) || (s_cv[j] == v && s_ci[j] > idx);
        s_sv[rank] = v;
        s_si[rank] = idx;
    }
    __syncthreads();
    // HF removes `scores < k-th largest`: survivors are the sorted prefix down to the last score equal to the k-th
    if (nc0 > k) {
        const float kth = s_sv[k - 1];
        for (int t = tid + k - 1; t < nc0; t += SAMPLE_THREADS)
            if (s_sv[t] == kth && (t + 1 == nc0 || s_sv[t + 1] < kth)) s_ncand = t + 1;
    }
    __syncthreads();
    const int nc = nc0 > k ? s_ncand : nc0;
    if (nc == 0) {   // every logit is -inf / NaN: nothing to draw from
        if (tid == 0) { out[b] = 0; if (a.n_kept) a.n_kept[b] = 0; }
        return;
    }
    const float vmax = s_sv[0];
    for (int t = tid; t < SAMPLE_CAP; t += SAMPLE_THREADS) s_cv[t] = t < nc ? expf(s_sv[t] - vmax) : 0.f;
    __syncthreads();
    // ---- TopPLogitsWarper in wave 0: lane L owns sorted ranks [8L, 8L + 8); sums by shuffle scans
    constexpr int E = SAMPLE_CAP / 64;
    if (tid < 64) {
        float e[E], part = 0.f;
#pragma unroll
        for (int q = E - 1; q >= 0; --q) { e[q] = s_cv[lane * E + q]; part += e[q]; }
        int keep = nc;
        if (a.use_top_p) {
            const float Z = wave_sum(part);
            // ascending cumulative probability = suffix sums over the descending order; the cut keeps every rank whose
            // suffix sum exceeds 1 - top_p (suffix sums fall with rank, so that is a prefix)
            float suf = part / Z;   // -> sum of the lanes above
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float up = __shfl_down(suf, o, 64);
                if (lane + o < 64) suf += up;
            }
            float c = suf - part / Z;
            int cnt = 0;
#pragma unroll
            for (int q = E - 1; q >= 0; --q) {
                c += e[q] / Z;
                cnt += (lane * E + q < nc) && c > a.top_p_complement;
            }
            keep = wave_sum_i(cnt);
            if (keep < a.min_keep) keep = a.min_keep < nc ? a.min_keep : nc;
        }
        if (lane == 0) s_kk = keep;
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
    // s_cv (exp of the score) is constant inside a tie group, so it already is in the final order
    if (tid < 64) {
        float e[E], part = 0.f;
#pragma unroll
        for (int q = 0; q < E; ++q) { e[q] = lane * E + q < keep ? s_cv[lane * E + q] : 0.f; part += e[q]; }
        const float Zk = wave_sum(part);
        const float u = a.uniforms ? a.uniforms[(int64_t)h * B + b] : 0.f;
        const float target = u * Zk;
        float pre = part;   // -> inclusive prefix over lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float dn = __shfl_up(pre, o, 64);
            if (lane >= o) pre += dn;
        }
        float c = pre - part;
        int below = 0;      // ranks whose inclusive cumulative sum is still <= target: the pick is the first one above
#pragma unroll
        for (int q = 0; q < E; ++q) {
            c += e[q];
            below += (lane * E + q < keep) && !(c > target);
        }
        int pick = wave_sum_i(below);
        pick = pick < keep ? pick : keep - 1;
        if (lane == 0) {
            out[b] = s_ci[pick];
            if (a.n_kept) a.n_kept[b] = keep;
        }
        if (a.kept_ids || a.kept_probs) {
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const int t = lane * E + q;
                if (t < keep) {
                    if (a.kept_ids) a.kept_ids[(int64_t)b * SAMPLE_CAP + t] = s_ci[t];
                    if (a.kept_probs) a.kept_probs[(int64_t)b * SAMPLE_CAP + t] = e[q] / Zk;
                }
            }
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
