/* Compiled by tests/test_host_cpu.py with `gcc -std=c99 -Wall -Wextra -Werror`: the public header must be plain C, and a
   C program must be able to link libvisualcla_hip.so and call the entry points that need no GPU. */
#include <stdio.h>
#include <string.h>
#include "visualcla_hip.h"

int main(void) {
    vcla_gemm_args g;
    vcla_attn_args at;
    vcla_sample_args sa;
    vcla_model_cfg cfg;
    vcla_ctx* ctx = NULL;
    int rc;
    memset(&g, 0, sizeof g); memset(&at, 0, sizeof at); memset(&sa, 0, sizeof sa); memset(&cfg, 0, sizeof cfg);
    printf("abi %d sizes %zu %zu %zu %zu\n", vcla_version(), sizeof g, sizeof at, sizeof sa, sizeof cfg);
    /* argument validation happens before any device work: these must fail with a message, not crash */
    rc = vcla_gemm(&g, VCLA_BF16, NULL);
    if (rc == VCLA_OK || strlen(vcla_last_error()) == 0) return 2;
    rc = vcla_ctx_create(&cfg, &ctx);
    if (rc == VCLA_OK) return 3;
    rc = vcla_sample(NULL, 0, 0, 0, 0, NULL, &sa, NULL, NULL);
    if (rc == VCLA_OK) return 4;
    printf("last error: %s\n", vcla_last_error());
    return 0;
}
