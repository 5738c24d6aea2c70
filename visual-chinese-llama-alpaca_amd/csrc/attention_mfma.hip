// attention_mfma.hip -- MFMA flash attention (bf16).  Placeholder until the tuned kernel lands: reports
// "unsupported" so vcla_attention() uses the generic kernel.
#include "vcla_common.h"
bool vcla_attention_mfma_supported(const vcla_attn_args*) { return false; }
int vcla_attention_mfma(const vcla_attn_args*, void*) { return vcla_fail(VCLA_ERR_BAD_ARG, "attention: MFMA kernel not built"); }
