// gemm_tiles.h -- tile order and LDS swizzle shared by the MFMA tile kernels (gemm.hip: 128 x 128; gemm_mfma256.hip: 256 x 256, bf16 and fp8)
#pragma once
#include "vcla_common.h"

// XCD-aware tile order shared by both MFMA kernels: block b runs on XCD b % 8, so give each XCD a contiguous run of
// tiles (bijective for any block count), then sweep N inside groups of GRP m-tiles so A panels stay L2-resident.
__device__ __forceinline__ void tile_assign(int bid, int tiles_m, int tiles_n, int GRP, int& tm, int& tn) {
    const int nblk = tiles_m * tiles_n;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_grp = GRP * tiles_n;
    const int g = bid / per_grp;
    const int gm0 = g * GRP;
    const int gsz = (tiles_m - gm0) < GRP ? (tiles_m - gm0) : GRP;
    tm = gm0 + (bid % per_grp) % gsz;
    tn = (bid % per_grp) / gsz;
}

#define GM_BM 128
#define GM_BN 128
#define GM_BK 64

// byte offset of 16-byte chunk `ch` (0..7) of row `row` inside a [128][64] bf16 tile, XOR-swizzled
__device__ __forceinline__ int lds_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }
