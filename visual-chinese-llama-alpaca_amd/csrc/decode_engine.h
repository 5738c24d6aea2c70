// decode_engine.h -- internal interface of the persistent B = 1 decode step (decode_engine.hip), used by engine.hip only.
#pragma once
#include "vcla_common.h"

// geometry the engine is compiled for (LLaMA-7B: hidden 4096 = 32 heads x 128) and the chip it assumes (256 CUs, one workgroup each)
#define EG_NCU 256
#define EG_D 4096
#define EG_H 32
#define EG_HD 128
#define EG_SLOT 16384          // one ring slot = one LDS-DMA fill = 16 pieces of 1 KiB
#define EG_S_QKV 24            // slots per CU and layer: 48 rows of wqkv (16 q + 16 k + 16 v rows of ONE head), 2 rows per slot
#define EG_S_O 8               // 16 rows of wo, K-major: slot j = the 16 rows x k in [512 j, 512 j + 512)
#define EG_NRING 9             // LDS ring slots
// mailboxes (8-byte {tag, two bf16} granules), per layer parity: X | QKV | AO | X1 | ACT | PART
#define EG_MB_X 0
#define EG_MB_QKV 2048
#define EG_MB_AO (2048 + 6144)
#define EG_MB_X1 (2048 + 6144 + 2048)
#define EG_MB_ACT (2048 + 6144 + 2048 + 2048)
#define EG_MB_MAX_ACT 5632     // intermediate <= 11264 (44 granule slots per CU)
#define EG_MB_PART (EG_MB_ACT + EG_MB_MAX_ACT)      // split attention: [32 heads][8 CUs of the group][136]: unnormalised o[128], m, l as fp32 granules
#define EG_PART_GRAN 136
#define EG_MB_PER_PARITY (EG_MB_PART + EG_H * 8 * EG_PART_GRAN)
#define EG_WS_STATE_OFF ((size_t)2 * EG_MB_PER_PARITY * 8)           // state words behind the mailboxes: [0] launch sequence, [1] failure code
#define EG_WS_BYTES (EG_WS_STATE_OFF + 256)
#define EG_TL_STRIDE 2048      // debug stamps per CU: 16 per layer (leader), then loader totals at [2040..]

struct vcla_engine_geom {
    int n_layers, inter, vocab;
    int upc;            // SwiGLU units (gate row + up row = one slot) per CU = inter / 256
    int s_dn;           // down_proj slots per CU = ceil(2 * ceil(upc / 2) * 256 / 512): K in granule order, 512 k per slot
    int s_lm;           // lm_head slots per CU (2 rows each)
    int slots_layer, slots_total;
    size_t cu_stride, slot_stride;      // bytes between the runs of consecutive CUs / between consecutive slots of one CU
};
// false when the model cannot run on the engine (geometry); fills g otherwise
bool vcla_engine_geometry(int hidden, int heads, int inter, int vocab, int n_layers, vcla_engine_geom* g);

struct vcla_engine_args {
    vcla_engine_geom g;
    const unsigned char* stream;   // "llama.engine.w": [slots_total][256 CUs][16 KiB] (g.slot_stride / g.cu_stride), every CU's weights in consumption order
    const float* gamma;            // "llama.engine.g": [2 L + 1][4096] fp32: ln1 / ln2 of every layer, then the final norm
    const bf16_t* x_in;            // [4096] embedding of the current token
    bf16_t* kv;                    // K/V cache [L][2][1][H][ctx_max][d] bf16
    int ctx_max, pos0;
    const int32_t* pos_dev;
    const int32_t* key_mask;       // [ctx_max] or NULL
    const float *rope_cos, *rope_sin;
    float scale, eps;
    float* logits;                 // [vocab] fp32
    unsigned long long* mbox;      // EG_WS_BYTES of workspace, zeroed by the caller before the FIRST step of a sequence of launches
    // greedy tail folded into the launch (decode loops without a sampler; NULL tail_ids_out: off): CU 0 reduces the per-CU maxima of the logits, then does what
    // vcla_argmax + post_select_kernel did in two more launches -- cur[0] = ids_out[*pos - step_base] = argmax (lowest index among equal maxima), x = embed[argmax], ++*pos
    int64_t* tail_ids_out;
    int64_t* tail_cur;
    const bf16_t* tail_embed;
    bf16_t* tail_x;
    int32_t* tail_pos;
    int tail_step_base;
    int fault;                     // test hook (VCLA_ENGINE_FAULT=1): the consumers of CU 7 leave at once -- every wait on their outputs must run out, not hang
    int split_min;                 // contexts of at least this many cached keys split a head's attention over the 8 CUs of its group (0: never)
    unsigned long long* timeline;  // debug (tools/engine_probe.py --timeline): [256 CUs][EG_TL_STRIDE] wall-clock stamps (100 MHz), or NULL
};

int vcla_engine_launch(const vcla_engine_args* a, hipStream_t s);
