// Tile order shared by the persistent GEMM kernels (gemm.hip, gemm_bf16x3.hip).
#pragma once
#include "mss_common.h"

#ifndef GEMM_GROUP_M_DEFAULT
#define GEMM_GROUP_M_DEFAULT 8
#endif
// Tile index -> (row tile, column tile). group_m == 0: column tile fastest (the 16 column tiles of one row block of a 2048 -> 4096
// product are consecutive). group_m > 0 (r04): groups of `group_m` row tiles, row tile fastest inside a group, so that the ~64
// workgroups an XCD runs together form a SQUARE-ish block of tiles (8 x 8 instead of 4 x 16): every K-slice of A is then shared by
// 8 and every K-slice of B by 8 of them in that XCD's L2, instead of 16 / 4. A bijection of the same tile set: results unchanged.
__device__ __forceinline__ void mss_tile_mn(int v, int mtiles, int ntiles, int group_m, int& mt, int& nt) {
  if (group_m <= 0 || ntiles == 1) { mt = v / ntiles; nt = v - mt * ntiles; return; }
  const int gsz = group_m * ntiles;
  const int g = v / gsz, r = v - g * gsz;
  const int first = g * group_m;
  const int rows = mtiles - first < group_m ? mtiles - first : group_m;
  nt = r / rows;
  mt = first + (r - nt * rows);
}

