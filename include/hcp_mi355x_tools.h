/* hcp_mi355x_tools.h — tuning / ablation hooks of libhcp_mi355x_tools.so (the same kernels built with -DHCP_TOOLS), used by the scripts under tools/ and
 * by the tests that force kernel variants.  They set process-global knobs: NOT part of the product ABI (include/hcp_mi355x.h), and the
 * product library libhcp_mi355x.so does not export them (tests/test_abi.py checks its symbol table). */
#ifndef HCP_MI355X_TOOLS_H
#define HCP_MI355X_TOOLS_H
#include "hcp_mi355x.h"
#ifdef __cplusplus
extern "C" {
#endif
int hcp_debug_gemm_table_stats(long* hits, long* misses); /* tools only: dispatch-table lookups since the last call; resets */
int hcp_debug_set_gemm_config(int cfg); /* tools/tune_gemm.py only: force tile id + 16*nsplit, -1 = heuristic */
int hcp_debug_set_gemm_ablation(int flags); /* tools only (wrong results when != 0): 1 no DMA, 2 no MFMA, 4 no LDS reads */
int hcp_debug_set_gn_target(int workgroups);   /* tools only: workgroups a two-launch GroupNorm aims for (default 512); -1 / -2: one-launch slab path off / on */
int hcp_debug_set_gemm_epilogue(int mode); /* tools only: -1 size rule, 0 lane-layout epilogue, 1 tile epilogue (16-byte row pieces through LDS) where possible */
int hcp_debug_set_conv_patch(int on); /* tools only: 0 = no LDS-resident-patch convolution kernel (csrc/conv_patch.hip), 1 = default rule (data gradients, split-K), 2 = every eligible conv */
int hcp_debug_set_gemm_loaders(int mode); /* tools only: -1 table, 0 never, 1 / 3 / 4 loader-wave variant with a 2 / 3 / 4 tile LDS ring */
int hcp_debug_set_gemm_glds(int on);    /* tools only: 1 = default (v2 main loop where eligible), 0/2 = first LDS-DMA loop everywhere */
int hcp_debug_set_attention_config(int cfg); /* tools only: bit0/1/2 = 32 rows per wave in fwd / dQ / dK,dV; -1 = heuristic */
int hcp_debug_set_wgrad_tile(int wx);
#ifdef __cplusplus
}
#endif
#endif
