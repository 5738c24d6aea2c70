/*
 * visualcla_hip.h -- C ABI of libvisualcla_hip.so: the MI355X (gfx950) hot path of
 * VisualCLA (CLIP-ViT-L/14 -> Resampler -> projection -> LLaMA decoder -> LM head).
 *
 * Boundary contract:
 *   - plain pointers + sizes, no torch types.  All device buffers (weights, activations,
 *     KV cache, workspace) are allocated and owned by the caller (PyTorch-ROCm); the
 *     library never allocates/frees device memory and never synchronises: every entry
 *     point enqueues work on the given hipStream_t (passed as void*) and returns.
 *   - every function returns an int status (VCLA_OK = 0); vcla_last_error() gives the
 *     text of the last failure on the calling thread.  Never abort()s.
 *   - callable from any host thread (chat_in_stream runs generate() in a worker thread,
 *     models/visualcla/modeling_utils.py:215-225); one vcla_ctx is used by one thread at
 *     a time.
 *
 * "act dtype": VCLA_BF16 = bf16 activations / fp32 accumulate (the product path);
 * VCLA_F32 = fp32 activations (the algorithmic-parity test mode, checked at 1e-3
 * against the fp32 CPU oracle).  GEMM weights are bf16 in both modes; small vectors
 * (biases, norm gains, class/position embeddings, RoPE tables) are fp32.
 *
 * What each entry point replaces in the reference (paths relative to /root/reference;
 * hf: = the pinned third-party dependency transformers, setup.py:13):
 *   vcla_layernorm          torch.nn.LayerNorm call sites hf:clip/modeling_clip.py:368,377,642;
 *                           models/visualcla/modeling_visualcla.py:284,350;
 *                           models/visualcla/modeling_visual_resampler.py:276,356
 *   vcla_rmsnorm            LlamaRMSNorm.forward hf:llama/modeling_llama.py:62-67
 *   vcla_rmsnorm_pack       the same, output in the MFMA-fragment-major layout of the batch-decode GEMMs
 *   vcla_gemm               every nn.Linear on the path (hf:clip/modeling_clip.py:280-282,332,345-347;
 *                           modeling_visual_resampler.py:174,187-188,274,341,354;
 *                           modeling_visualcla.py:288,354; hf:llama/modeling_llama.py:175,250-252,280,480)
 *                           with the activation / residual fused (quick_gelu, erf-gelu, SiLU*up)
 *   vcla_im2col             the patch-embedding Conv2d hf:clip/modeling_clip.py:209 (as GEMM operand)
 *   vcla_vit_assemble       class/position embedding + pre_layrnorm hf:clip/modeling_clip.py:211-218,642
 *   vcla_attention          eager_attention_forward hf:clip/modeling_clip.py:259-277,
 *                           hf:llama/modeling_llama.py:191-214; modeling_visual_resampler.py:213-253
 *   vcla_image_preprocess_batch   the same call sites, N same-sized images per launch pair
 *   vcla_image_preprocess   (next row N1) CLIPImageProcessor.__call__ as invoked at models/visualcla/modeling_utils.py:150-152
 *   vcla_check_request      the request checks in front of them: image-slot shape modeling_visualcla.py:296-302 / :362-367, ids / labels in range, mask shape
 *   vcla_embed_splice       embed_tokens + image splice modeling_visualcla.py:280,292-305 / :346,358-370
 *   vcla_rope_kv_append     apply_rotary_pos_emb + cache update hf:llama/modeling_llama.py:130-160,255-259
 *   vcla_attn_decode_fused  the decode-step instance of LlamaAttention.forward hf:llama/modeling_llama.py:217-281
 *                           (RoPE + DynamicCache.update + eager attention) in one launch
 *   vcla_argmax             greedy token selection hf:generation/utils.py (argmax over fp32 logits)
 *   vcla_causal_lm_loss     the labels branch of forward, modeling_visualcla.py:321-328 (hf:loss/loss_utils.py ForCausalLMLoss)
 *   vcla_sample             (next row N2) the logits processors / warpers / draw of HF sample() under
 *                           DEFAULT_GENERATION_CONFIG models/visualcla/modeling_utils.py:36-47
 *   vcla_vision_forward     modeling_visualcla.py:283-288 / :349-354 (and tgwebui embed_images,
 *                           scripts/inference/text_generation_webui/visualcla/visualcla.py:116-129)
 *   vcla_llama_prefill      LlamaForCausalLM.forward over the spliced embeds, modeling_visualcla.py:321-328
 *   vcla_llama_decode_step  one iteration of the HF generate loop, modeling_visualcla.py:382-391
 */
#ifndef VISUALCLA_HIP_H
#define VISUALCLA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCLA_ABI_VERSION 5

/* status codes */
enum {
    VCLA_OK = 0,
    VCLA_ERR_BAD_SHAPE = 1,
    VCLA_ERR_BAD_DTYPE = 2,
    VCLA_ERR_UNSUPPORTED_ARCH = 3,
    VCLA_ERR_HIP = 4,
    VCLA_ERR_BAD_ARG = 5,
    VCLA_ERR_WORKSPACE = 6,
    VCLA_ERR_MISSING_TENSOR = 7
};

/* activation dtypes */
enum { VCLA_F32 = 0, VCLA_BF16 = 1 };
/* OR-ed into the `dtype` argument of vcla_rope_kv_append / vcla_attn_decode_fused (with VCLA_BF16 only): the K / V cache holds
   OCP fp8 (e4m3fn) values, unit scale -- one byte per element, [B, H, ctx_max, d] bytes per slab.  The lossy companion of the
   fp8 weight path (BASELINE configs[4]): at B = 64 the bf16 cache is as many bytes per decode step as the fp8 weights. */
#define VCLA_KV_FP8 0x100

/* GEMM epilogues (applied to acc + bias) */
enum {
    VCLA_EPI_NONE = 0,
    VCLA_EPI_QUICK_GELU = 1, /* x * sigmoid(1.702 x)            (CLIP MLP)        */
    VCLA_EPI_GELU_ERF = 2,   /* 0.5 x (1 + erf(x / sqrt 2))     (resampler FFN)   */
    VCLA_EPI_SWIGLU = 3      /* silu(gate) * up; W rows interleaved in blocks of 16
                                (16 gate rows, 16 up rows, ...); output width N/2  */
};

int vcla_version(void);
const char* vcla_last_error(void);
/* 0 if the current device is a gfx950 and kernels can launch; else an error code. */
int vcla_device_check(void);

/* ---------------------------------------------------------------- primitives */

/* y[r,:] = LN(x[r,:]) * gamma + beta        rows x cols, row strides ldx/ldy (elements) */
int vcla_layernorm(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                   int rows, int cols, float eps, int dtype, void* stream);

/* y[r,:] = gamma * round_to_dtype(x[r,:] * rsqrt(mean(x^2) + eps)) */
int vcla_rmsnorm(const void* x, int64_t ldx, const float* gamma, void* y, int64_t ldy, int rows, int cols,
                 float eps, int dtype, void* stream);

typedef struct vcla_gemm_args {
    const void* A;        /* [M, K] activations, row stride lda (elements)                     */
    int64_t lda;
    const void* W;        /* [N_pad, K] bf16, K contiguous; N_pad = N rounded up to 128 rows    */
    const float* bias;    /* [N] or NULL                                                        */
    const void* residual; /* [M, N_out] act dtype, row stride ldr, added last; or NULL          */
    int64_t ldr;
    void* C;              /* [M, N_out], row stride ldc                                         */
    int64_t ldc;
    int M, N, K;          /* K % 64 == 0; N = rows of W actually used (SWIGLU: N_out = N/2)     */
    int epilogue;         /* VCLA_EPI_*                                                         */
    int out_f32;          /* 1: C is fp32 even when dtype == VCLA_BF16 (logits)                 */
    /* output row remap: row m is stored at row (m / c_group_rows) * c_group_stride +
       (m % c_group_rows) + c_row_offset of C (c_group_rows == 0: identity)                    */
    int c_group_rows, c_group_stride, c_row_offset;
    int force_kernel;     /* 0 auto; 1 MFMA 128x128 tile; 2 GEMV (M <= 8); 3 fp32 tile; 4 MFMA 256x256 direct-to-LDS;
                             5 = 4 without the hand-placed ds_read/MFMA interleave; 6 generic GEMV (no LDS x staging);
                             7 skinny MFMA (2 <= M <= 128, W streamed once, intra-workgroup split-K);
                             8 panel MFMA (M <= 128, activations shared through LDS, split-K over workgroups);
                             9 streaming MFMA (M <= 64, needs A_frag + W_frag / W_q8_frag: every CU streams an equal share of W over
                               the FULL K, no split-K partials, no LDS in the main loop);
                             10 fp8 MFMA 256x256x128 direct-to-LDS (needs A_q8 + a_scale + W_q8 + w_scale);
                             11 ring MFMA (the default for 129 <= M <= 256: full-K tiles fed by an LDS-DMA ring, row-major bf16 A, bf16 W or
                               fp8 W_q8 + w_scale, epilogue NONE / SWIGLU; tile chosen per shape; with bf16 W, bf16 output and W_frag set the
                               weight pieces come from the fragment-major twin: bit-identical, contiguous DMA), 12 / 13 / 14 = the same
                               with the 256x96 / 128x96 / 64x64 tile forced */
    /* optional fused RMSNorm prologue (GEMV kernel, M <= 8 only): A holds the UN-normalised rows and the
       kernel computes gamma * x * rsqrt(mean(x^2) + eps) on the fly (LlamaRMSNorm + Linear in one launch) */
    const float* norm_gamma; /* [K] or NULL */
    float norm_eps;
    /* optional fp32 scratch for the split-K panel kernel (kernel 8): >= S * M * N_pad * 4 bytes lets it use S K-slices */
    void* splitk_ws;
    size_t splitk_ws_bytes;
    /* optional fragment-major twin of W, [N_pad/16][K/32][64 lanes][8] bf16 (visualcla/weights.py:to_fragment_major):
       lets the M <= 128 panel kernel stream each 16-row tile as one contiguous region and the 129 - 256-row ring kernel DMA
       its weight pieces as contiguous KiBs */
    const void* W_frag;
    /* optional OCP fp8 (e4m3fn) weight copies for the HBM-bound decode kernels, per-row fp32 scale w_scale [N_pad]
       (W ~= q * w_scale[row]); W_q8: [N_pad, K] row-major (M = 1 GEMV); W_q8_frag: [N_pad/16][K/64][64 lanes][16]
       (two MFMA k-steps per 16-byte lane load; panel kernel).  Dequantised to bf16 in registers; fp32 accumulate. */
    const void* W_q8;
    const void* W_q8_frag;
    const float* w_scale;
    /* optional fused "next RMSNorm" (LLaMA batch decode: o_proj / down_proj feed input_layernorm / post_attention_layernorm,
       hf:llama/modeling_llama.py:300-320): after C (residual included) is stored, also store
       post_norm_out[m, :] = post_norm_gamma * round(C[m, :] * rstd(C[m, :])), exactly what vcla_rmsnorm would compute
       from C.  Fused into the split-K reduction of the panel kernel; every other kernel runs vcla_rmsnorm afterwards.
       Needs epilogue NONE, out_f32 0, no row regrouping, N <= 8192. */
    const float* post_norm_gamma;
    float post_norm_eps;
    void* post_norm_out;
    int64_t post_norm_ld;
    /* optional fragment-major activations for the streaming decode GEMM (kernel 9): A_frag [K/32][ceil(M/16)][64 lanes][8]
       act dtype bf16, element (m, k) at ((k/32 * MT + m/16) * 64 + ((k%32)/8)*16 + m%16) * 8 + k%8 -- each MFMA operand
       fragment is one contiguous 1 KiB block (vcla_rmsnorm_pack, vcla_attn_decode_fused(out_frag) and C_frag produce it).
       When set (with M <= 64 and W_frag or W_q8_frag) `A` may be NULL. */
    const void* A_frag;
    /* optional second output in the same fragment-major layout, [N_out/32][ceil(M/16)][64][8] bf16 (N_out % 32 == 0): feeds the
       next streaming GEMM (SwiGLU activations -> down_proj).  When set, `C` may be NULL (no row-major copy is stored). */
    void* C_frag;
    /* Deferred RMSNorm across two streaming GEMMs (LLaMA batch decode: o_proj -> post_attention_layernorm -> gate/up, down_proj ->
       next input_layernorm -> qkv).  PRODUCER side: c_frag_gamma [N] makes C_frag hold bf16(gamma[n] * C[m, n]) and c_row_ssq
       [M][ceil(N/16)] receives, per row and 16-column tile, the sum of squares of the stored (rounded) C values (with ds_splitk > 1 and
       N % 256 == 0 the reduce launch writes one value per 256 columns instead: [M][N/256], packed at the front of the buffer).  CONSUMER side:
       a_row_ssq (the producer's c_row_ssq, a_row_ssq_parts partial sums per row) turns the accumulator of row m into
       acc * rsqrt(sum(a_row_ssq[m, :]) / K + a_norm_eps) before bias / activation: together
       W . (gamma * x) * rstd(x) = W . RMSNorm(x), with no norm launch in between.  Kernel 9 only. */
    const float* c_frag_gamma;
    float* c_row_ssq;
    const float* a_row_ssq;
    int a_row_ssq_parts;
    float a_norm_eps;
    /* fp8 x fp8 on the fp8 MFMA pipe (kernel 10, BASELINE configs[4]; the MI355X analogue of the reference's load_in_8bit,
       models/visualcla/modeling_visualcla.py:155): A_q8 [M, K] OCP e4m3fn activations with one fp32 scale per row (a_scale [M],
       A ~= A_q8 * a_scale[row]; vcla_quant_fp8_rows produces both) against W_q8 / w_scale.  The products run on
       v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales), fp32 accumulate; both row scales are applied in the epilogue.
       K % 128 == 0, M > 0.  `A` is not read. */
    const void* A_q8;
    const float* a_scale;
    /* streaming kernel (9) only: split K into ds_splitk slices (0 / 1 = none).  Workgroups then own ds_splitk times as many
       weight tiles over 1/ds_splitk of K -- each re-reads only that share of the activations (what bounds the short-N / long-K
       GEMMs o_proj and down_proj at M = 64) -- store fp32 partial tiles in splitk_ws (>= ds_splitk * M * N * 4 bytes) and a
       second, fully parallel launch sums the slices in order and applies the epilogue (bias, residual, C, C_frag, c_row_ssq).
       Epilogue NONE only. */
    int ds_splitk;
    /* with ds_splitk > 1: 1 = stop after the first launch -- splitk_ws then holds the RAW fp32 slices [ds_splitk][M][N] (no rstd of
       a_row_ssq, no fp8 weight scale, no bias applied; C / C_frag / c_row_ssq untouched) for a consumer that sums them itself
       (vcla_attn_decode_fused_parts: the qkv projection of a batch decode step).  a_row_ssq may be set (it is not applied). */
    int ds_raw_partials;
    /* SLAB-MAJOR operands for the LDS-DMA tile kernels (ABI v4).  Measured on MI355X (profiles/r05_l2_intake.txt): a wave instruction
       that moves 1 KiB CONTIGUOUS global bytes into LDS sustains ~50 B/clk/CU, one that gathers 8 rows x 128 B (or 4 x 256, 2 x 512) of a
       row-major matrix only ~19 B/clk/CU -- and the K slab (64 elements = 128 B per row) of a tile is exactly such a gather.  In the
       slab-major layout the matrix is stored K-slab by K-slab, [K/64][rows][64]: element (r, k) at ((k/64) * rows + r) * 64 + k % 64, so the
       slab of ANY run of 8 rows is one contiguous 1 KiB.  A_slab [K/64][a_slab_rows][64] bf16 (a_slab_rows >= M) replaces A when set;
       W_slab [K/64][N_pad][64] bf16 replaces W; W_q8_slab [K/64][N_pad][64] e4m3 bytes (16 rows = 1 KiB) replaces W_q8 (w_scale as before).
       Ring kernel (11 - 14) and 256x256 kernel (4); visualcla/weights.py:to_slab_major.  (ABI v4 also declared a slab-major OUTPUT, C_slab /
       c_slab_rows, that no kernel ever wrote; ABI v5 removed the two fields -- they were the last members, so the struct only got shorter.) */
    const void* A_slab;
    int64_t a_slab_rows;
    const void* W_slab;
    const void* W_q8_slab;
} vcla_gemm_args;

/* C = epilogue(A . W^T + bias) (+ residual) */
int vcla_gemm(const vcla_gemm_args* args, int dtype, void* stream);

/* LlamaRMSNorm (hf:llama/modeling_llama.py:62-67) of `rows` <= 64 bf16 rows, written in the fragment-major layout the
   streaming decode GEMM reads (vcla_gemm_args.A_frag): y_frag [cols/32][ceil(rows/16)][64][8] bf16; values bit-identical to
   vcla_rmsnorm.  gamma == NULL: plain re-layout (no normalisation).  cols % 32 == 0. */
int vcla_rmsnorm_pack(const void* x, int64_t ldx, const float* gamma, void* y_frag, int rows, int cols, float eps,
                      void* stream);

/* Per-row dynamic fp8 quantisation of bf16 activations for the fp8 MFMA GEMM (vcla_gemm_args.A_q8 / a_scale):
   scale[r] = max(|x[r, :]|) / 448 (>= 1e-20), q[r, c] = e4m3fn(x[r, c] / scale[r]) round-to-nearest-even.  cols % 16 == 0. */
int vcla_quant_fp8_rows(const void* x, int64_t ldx, void* q, float* scale, int rows, int cols, void* stream);

/* pixel_values [B, C, H, W] -> patches [B * (H/P) * (W/P), k_pad]; column order (c, ky, kx), zero padded */
int vcla_im2col(const void* pixels, void* patches, int B, int C, int H, int W, int P, int k_pad, int dtype,
                void* stream);

/* x[b, 0] = cls + pos[0]; x[b, 1+p] = patch[b*np + p] + pos[1+p]; then LayerNorm(gamma, beta)  -> y [B*(np+1), D] */
int vcla_vit_assemble(const void* patch_embeds, const float* cls, const float* pos, const float* gamma,
                      const float* beta, void* y, int B, int np, int D, float eps, int dtype, void* stream);

typedef struct vcla_attn_args {
    const void *q, *k, *v;
    void* o;
    /* element (b, h, i, d) of X sits at X[b * x_bs + h * x_hs + i * x_rs + d]  (strides in elements) */
    int64_t q_bs, q_hs, q_rs;
    int64_t k_bs, k_hs, k_rs;
    int64_t v_bs, v_hs, v_rs;
    int64_t o_bs, o_hs, o_rs;
    int B, H, Tq, Tk, D; /* D in {32, 64, 128} */
    float scale;
    int causal;             /* 1: query i attends keys j <= i + (Tk - Tq)                            */
    const int32_t* key_mask; /* [B, key_mask_ld] 1 = attend, 0 = masked; or NULL                      */
    int64_t key_mask_ld;
    const int32_t* tk_dev;  /* optional device scalar: effective Tk = *tk_dev + tk_dev_add (Tq = 1)  */
    int tk_dev_add;
    int force_kernel;       /* 0 auto; 1 generic (fp32 math) kernel; 2 MFMA flash kernel (tile by tile);
                               3 whole-sequence ViT kernel (bidirectional, unmasked, d = 64, 65 / 257 tokens) */
} vcla_attn_args;

/* o = softmax(scale * q k^T + mask) v */
int vcla_attention(const vcla_attn_args* args, int dtype, void* stream);

/* Next-row N1: CLIP preprocessing of ONE uint8 HWC RGB image on the device (the reference does it on the host with
   PIL/numpy, models/visualcla/modeling_utils.py:150-152): Pillow-exact fixed-point bicubic resize (coefficient tables
   {first tap, tap count, 22-bit weights [n, kmax]} for the S cropped output columns / rows come from the host), centre
   crop to S x S, float32(float64(v) * rescale), then (x - mean) / std in float32 (host float arrays), CHW store in the activation dtype.
   tmp: H * S * 3 bytes of scratch.  mean3 / std3 are HOST pointers. */
int vcla_image_preprocess(const uint8_t* img, int H, int W, uint8_t* tmp, int S, const int32_t* h_lo, const int32_t* h_cnt,
                          const int32_t* h_k, int h_kmax, const int32_t* v_lo, const int32_t* v_cnt, const int32_t* v_k,
                          int v_kmax, double rescale, const float* mean3, const float* std3, void* out, int dtype,
                          void* stream);

/* The same for N images of ONE size in one launch pair (imgs [N, H, W, 3] uint8, tmp N * H * S * 3 bytes, out [N, 3, S, S]): the
   batched form a server feeds from one pinned staging buffer, so preprocessing costs two launches and one host-to-device copy
   per BATCH instead of per image. */
int vcla_image_preprocess_batch(const uint8_t* imgs, int N, int H, int W, uint8_t* tmp, int S, const int32_t* h_lo,
                                const int32_t* h_cnt, const int32_t* h_k, int h_kmax, const int32_t* v_lo, const int32_t* v_cnt,
                                const int32_t* v_k, int v_kmax, double rescale, const float* mean3, const float* std3, void* out,
                                int dtype, void* stream);

/* Every data-dependent validation of a request (ids [B, T] int64) in ONE launch; the caller copies the five flags back and raises the reference's
   errors (modeling_visualcla.py:296-302 forward, :362-367 generate).  flags (int32[5], zeroed here): [0] an id outside [0, vocab); [1] q_tokens > 0 and
   a row whose first start_id (<img>) is not followed by q_tokens fillers and end_id (</img>) inside the row -- rows without start_id (need_tok: or without
   tok_id, as `forward` asks) carry no image and pass; [2] mask has a zero; [3] mask has a visible position after a masked one that follows a visible
   one (a hole rather than padding at either end; mask_prefix_visible: as if visible columns preceded the mask, the image_at_head placement); [4] a label
   outside [0, vocab) other than -100.  img_pos[b] (int32, may be NULL; written when q_tokens > 0) = position of the row's first start_id, -1 for rows
   without an image.  mask [B, Tm] / labels [B, Tl] int64 or NULL. */
int vcla_check_request(const int64_t* ids, int B, int T, int vocab, int q_tokens, int64_t start_id, int64_t end_id, int64_t tok_id, int need_tok,
                       const int64_t* mask, int Tm, int mask_prefix_visible, const int64_t* labels, int Tl, int32_t* img_pos, int32_t* flags,
                       void* stream);

/* out[b, t] = table[ids[b, t]], except rows img_pos[b]+1 .. img_pos[b]+Q which take image_embeds[b, :]
   (img_pos[b] < 0: no image in that sample).  table is bf16 [V, D]. */
int vcla_embed_splice(const int64_t* ids, const void* table, const void* image_embeds, const int32_t* img_pos,
                      void* out, int B, int T, int Q, int D, int V, int dtype, void* stream);

/* RoPE on q (in place) and k of a fused qkv buffer [B*T, 3*H*d], append k / v to the cache.
   position of row t = pos0 + (pos_dev ? *pos_dev : 0) + t.  cos/sin tables fp32 [max_pos, d/2].
   k_cache / v_cache: [B, H, ctx_max, d].  dtype | VCLA_KV_FP8: the cache receives e4m3 bytes and the rotated k is ALSO written back
   into the qkv buffer (bf16), so that a prefill can attend over the exact bf16 rows while the cache keeps the 1-byte copies. */
int vcla_rope_kv_append(void* qkv, void* k_cache, void* v_cache, const float* cos_tab, const float* sin_tab,
                        int B, int T, int H, int d, int ctx_max, int pos0, const int32_t* pos_dev, int dtype,
                        void* stream);

/* Fused decode-step attention for every (sequence, head): RoPE on q and the new k of the fused qkv row [B, 3*H*d],
   append k / v to the cache at position pos0 + (pos_dev ? *pos_dev : 0), softmax(scale q K^T) V over keys 0..pos.
   out [B, H*d]; out_frag != 0 (bf16 only, B <= 64): out is written in the fragment-major layout of vcla_gemm_args.A_frag
   ([H*d/32][ceil(B/16)][64][8]) for the streaming o_proj GEMM.  key_mask [B, key_mask_ld] optional.
   dtype | VCLA_KV_FP8 (d = 128 or 64): the cache rows are e4m3 bytes -- 16 elements per 16-byte lane load, converted in
   registers; the new token's k / v enter the cache, and this step's own softmax, as their e4m3 roundings. */
int vcla_attn_decode_fused(const void* qkv, void* k_cache, void* v_cache, const float* cos_tab, const float* sin_tab,
                           void* out, int B, int H, int d, int ctx_max, int pos0, const int32_t* pos_dev,
                           const int32_t* key_mask, int64_t key_mask_ld, float scale, int dtype, int out_frag, void* stream);

/* The same step when the qkv projection ran as TWO K slices that were not reduced (vcla_gemm_args.ds_raw_partials): qkv_parts = slice 0,
   fp32 [B][3*H*d], slice 1 slice_stride elements behind it.  The kernel sums the slices and applies what the GEMM epilogue would have:
   row_ssq (optional, [B][16] partial sums of squares of the un-normalised row: rstd = rsqrt(sum / (H*d) + norm_eps), the deferred RMSNorm)
   and w_scale (optional, [3*H*d] fp8 weight scales), rounds to bf16 and proceeds as vcla_attn_decode_fused.  bf16 (| VCLA_KV_FP8), head dim
   64 / 128, B * H >= 1024 (the 2-wave batch form).  Saves the reduce launch AND half of every CU's activation reads in the qkv GEMM. */
int vcla_attn_decode_fused_parts(const float* qkv_parts, int64_t slice_stride, const float* row_ssq, const float* w_scale, float norm_eps,
                                 void* k_cache, void* v_cache, const float* cos_tab, const float* sin_tab, void* out, int B, int H, int d,
                                 int ctx_max, int pos0, const int32_t* pos_dev, const int32_t* key_mask, int64_t key_mask_ld, float scale,
                                 int dtype, int out_frag, void* stream);

/* ids_out[b] = argmax_j logits[b, j] (first maximum); logits fp32 [B, ld] */
int vcla_argmax(const float* logits, int64_t ld, int64_t* ids_out, int B, int V, void* stream);

/* forward(labels=...) -> .loss: the shifted causal-LM cross-entropy LlamaForCausalLM.forward computes when the reference passes
   `labels` through (models/visualcla/modeling_visualcla.py:321-328 -> hf:loss/loss_utils.py ForCausalLMLoss): logits fp32 [B * T, ld_row]
   (row b * T + t), labels int64 [B, T]; position t is scored against labels[b, t + 1], targets equal to ignore_index (-100) are
   skipped, loss_out[0] = mean over the scored positions (nan when there is none).  row_loss_ws: B * T floats of scratch. */
int vcla_causal_lm_loss(const float* logits, int64_t ld_row, const int64_t* labels, int B, int T, int V, int64_t ignore_index,
                        float* row_loss_ws, float* loss_out, void* stream);

/* Next-row N2: HF's logits processors + warpers + one draw, on the device, for every sequence of the batch.
   Replaces, per decode step, what `model.generate` runs under the reference's DEFAULT_GENERATION_CONFIG
   (models/visualcla/modeling_utils.py:36-47) via hf:generation/logits_process.py, in HF's order:
   repetition penalty -> no-repeat-ngram -> min-new-tokens -> temperature -> top-k -> top-p -> softmax -> draw.
   The draw is the inverse CDF of the kept set (descending probability, ties by token id) at uniforms[h, b], h = number of
   tokens generated so far, so the result is a pure function of (logits, history, uniforms).  top_k = 1 is greedy
   decoding over the processed scores.  `logits` [B, ld] fp32 is MODIFIED IN PLACE (penalties / bans). */
#define VCLA_SAMPLE_MAX_TOP_K 256
#define VCLA_SAMPLE_MAX_EOS 4
#define VCLA_SAMPLE_MAX_VOCAB 53248 /* the logits row is held in the registers of one 1024-thread workgroup */
#define VCLA_SAMPLE_KEPT_LD 512
typedef struct vcla_sample_args {
    float repetition_penalty;   /* 1 = off; applied once per distinct generated token                         */
    int no_repeat_ngram_size;   /* 0 = off                                                                    */
    int min_new_tokens;         /* eos ids are banned while fewer tokens than this were generated; 0 = off    */
    int n_eos;
    int eos_ids[VCLA_SAMPLE_MAX_EOS];
    float temperature;          /* 1 = off                                                                    */
    int top_k;                  /* 1 .. VCLA_SAMPLE_MAX_TOP_K (ties at the k-th value are kept, like HF)       */
    double top_p;               /* 1 = off; tokens whose ascending cumulative probability <= 1 - top_p are cut */
    int min_tokens_to_keep;     /* >= 1                                                                       */
    const float* uniforms;      /* device [*, B] in [0, 1): row h is consumed at history length h; NULL = 0   */
    const int64_t* history;     /* device [*, B] step-major: the tokens generated so far (prompt excluded,
                                   as in HF generate driven by inputs_embeds, modeling_visualcla.py:382-391)   */
    /* optional taps of the kept set, descending probability (device, may be NULL) */
    int64_t* kept_ids;          /* [B, VCLA_SAMPLE_KEPT_LD] */
    float* kept_probs;          /* [B, VCLA_SAMPLE_KEPT_LD] */
    int32_t* n_kept;            /* [B] */
} vcla_sample_args;

/* history length h = n_hist + (n_hist_dev ? *n_hist_dev : 0), at most 4096 */
int vcla_sample(float* logits, int64_t ld, int B, int V, int n_hist, const int32_t* n_hist_dev,
                const vcla_sample_args* args, int64_t* ids_out, void* stream);

/* ---------------------------------------------------------------- model context */

typedef struct vcla_model_cfg {
    int act_dtype; /* VCLA_F32 | VCLA_BF16 */
    /* vision tower */
    int v_hidden, v_layers, v_heads, v_inter, v_patch, v_image, v_channels;
    float v_eps;
    /* resampler */
    int r_hidden, r_layers, r_heads, r_inter, r_queries;
    float r_eps;
    /* text decoder */
    int t_hidden, t_layers, t_heads, t_inter, t_vocab, t_max_pos;
    float t_eps;
    float t_rope_theta;
    /* 1: decoder GEMMs with more than 128 rows (prefill) whose weight has fp8 copies registered ("<name>.q8" + ".s8") run on the
       fp8 MFMA pipe, activations quantised per row on the fly (BASELINE configs[4]); 0: bf16 MFMA */
    int t_fp8_mfma;
    /* 1 (bf16 activations only): the LLaMA K / V cache holds e4m3 bytes (VCLA_KV_FP8): vcla_kv_cache_bytes halves, prefill (pos0 = 0
       only) attends over the bf16 rows of its own qkv buffer, decode steps read the 1-byte cache */
    int t_kv_fp8;
} vcla_model_cfg;

typedef struct vcla_ctx vcla_ctx;

int vcla_ctx_create(const vcla_model_cfg* cfg, vcla_ctx** out);
void vcla_ctx_destroy(vcla_ctx* ctx);
/* Register a caller-owned device tensor under a name (see csrc/engine.cpp for the list:
   "vit.patch_w", "vit.l3.wqkv", "llama.l0.wgu", ...).  The caller keeps the memory alive. */
int vcla_ctx_set_tensor(vcla_ctx* ctx, const char* name, const void* ptr, size_t nbytes);
/* Checks that every tensor the config needs is registered with the right size. */
int vcla_ctx_finalize(vcla_ctx* ctx);

size_t vcla_vision_workspace_bytes(const vcla_ctx* ctx, int B);
size_t vcla_llama_workspace_bytes(const vcla_ctx* ctx, int B, int T);
size_t vcla_kv_cache_bytes(const vcla_ctx* ctx, int B, int ctx_max);

/* pixel_values [B, C, H, W] (act dtype) -> image_embeds [B, Q, t_hidden].
   Optional taps (act dtype, may be NULL): vit_tap [v_layers + 2][B * N * v_hidden]
   (entry v_layers = post-LN output, entry v_layers + 1 = the embedding stage: patch + class + position
   embeddings after pre_layrnorm; ABI v3 -- v2 had v_layers + 1 entries), res_tap [r_layers][B * Q * r_hidden]. */
int vcla_vision_forward(vcla_ctx* ctx, const void* pixel_values, void* image_embeds, int B, void* ws,
                        size_t ws_bytes, void* vit_tap, void* res_tap, void* stream);

/* Decoder over inputs_embeds [B, T, t_hidden] at positions pos0 .. pos0+T-1; fills the KV cache
   ([L][2][B][H][ctx_max][d]).  logits (fp32): all_logits != 0 -> [B, T, V] else last position only [B, V].
   key_mask: optional [B, ctx_max] int32.  layer_tap: optional [L + 1][B*T*hidden] (final-norm output last). */
int vcla_llama_prefill(vcla_ctx* ctx, const void* inputs_embeds, int B, int T, int pos0, void* kv_cache,
                       int ctx_max, const int32_t* key_mask, float* logits, int all_logits, void* ws,
                       size_t ws_bytes, void* layer_tap, void* stream);

/* One greedy decode step for B sequences: embeds ids_in[b] (device int64), runs the decoder at position
   pos0 + *pos_dev against the cache, writes fp32 logits [B, V] (optional) and ids_out[b] = argmax
   (optional).  If advance_pos != 0 the device counter *pos_dev is incremented afterwards, so steps can be
   enqueued back to back (or replayed from a hipGraph) without host round trips. */
int vcla_llama_decode_step(vcla_ctx* ctx, const int64_t* ids_in, int B, int pos0, int32_t* pos_dev,
                           int advance_pos, void* kv_cache, int ctx_max, const int32_t* key_mask, float* logits,
                           int64_t* ids_out, void* ws, size_t ws_bytes, void* stream);

/* Greedy decode of n_steps tokens, fully enqueued on the stream: ids_out [n_steps, B] (device int64);
   the first input token is ids_in.  use_graph != 0 captures one step in a hipGraph and replays it. */
int vcla_llama_decode_loop(vcla_ctx* ctx, const int64_t* ids_in, int B, int pos0, int32_t* pos_dev, int n_steps,
                           void* kv_cache, int ctx_max, const int32_t* key_mask, int64_t* ids_out, void* ws,
                           size_t ws_bytes, int use_graph, void* stream);

/* The same loop with vcla_sample in place of the argmax (sampling == NULL: greedy).  sampling->history must be the
   step-major token buffer whose row n_hist0 + *pos_dev the step is about to produce -- i.e. ids_out - n_hist0 * B when
   the caller keeps the n_hist0 earlier tokens (the one drawn from the prefill logits) in front of ids_out; the step reads
   rows [0, n_hist0 + *pos_dev) of it and uniforms row n_hist0 + *pos_dev. */
int vcla_llama_decode_loop_sampled(vcla_ctx* ctx, const int64_t* ids_in, int B, int pos0, int32_t* pos_dev, int n_steps,
                                   void* kv_cache, int ctx_max, const int32_t* key_mask, int64_t* ids_out, void* ws,
                                   size_t ws_bytes, int use_graph, const vcla_sample_args* sampling, int n_hist0,
                                   void* stream);

/* ABI v5.  At B = 1 (bf16, LLaMA-7B geometry, "llama.engine.w" / "llama.engine.g" registered) a decode step is ONE persistent launch
   (csrc/decode_engine.hip) that replaces the per-token loop body of hf generation/utils.py reached from
   models/visualcla/modeling_visualcla.py:382-391.  Its workgroups exchange operator outputs inside the launch and bound every wait; this call
   synchronises `stream` and returns VCLA_ERR_HIP (vcla_last_error() names the wait site and the CU) when a wait of any step since the
   workspace was last handed to vcla_llama_decode_step / _loop ran out -- the tokens of that loop are then invalid.  VCLA_OK otherwise, and
   always when the steps ran as separate launches (VCLA_ENGINE=0, other batch sizes, fp8 weights). */
int vcla_llama_decode_status(vcla_ctx* ctx, int B, const void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VISUALCLA_HIP_H */
