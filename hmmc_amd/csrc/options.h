// Process-wide A/B switches of the library, set through hmmc_set_option (include/hmmc_hip.h).  The library itself reads no
// environment variable: hmmc_amd/_lib.py translates the HMMC_* variables INTEGRATION.md lists into hmmc_set_option calls
// when it loads the library.
#pragma once
#include <atomic>

enum HmmcOption {
  HMMC_OPT_NO_WGRAD_GROUP = 0,   // one launch per weight gradient instead of the grouped launch (as before round 3)
  HMMC_OPT_NO_F32_WAVEK = 1,     // fp32 GEMM: never the wave-split-K kernel
  HMMC_OPT_NO_F32_DMA = 2,       // fp32 GEMM / eval scorer: never the LDS-DMA kernel
  HMMC_OPT_NO_LEAD_ATTN = 3,     // last block of a lead_only tower: attention for all queries (as before round 4)
  HMMC_OPT_COUNT = 4
};
extern std::atomic<int> g_hmmc_options[HMMC_OPT_COUNT];
static inline bool hmmc_option(HmmcOption o) { return g_hmmc_options[o].load(std::memory_order_relaxed) != 0; }

// Live timing of GEMM launches for bench.py (hmmc_gemm_profile_start / _stop, gemm_f16.hip): a launch site brackets its kernels
// with hmmc_prof_begin / hmmc_prof_end; slot 0-2 = the fp16 GEMM by operand layout, 3 = the exact-fp32 GEMM.  begin returns a
// token (< 0: timing is off, end is a no-op).
long hmmc_prof_begin(double flops, double bytes, int slot, hipStream_t stream);
void hmmc_prof_end(long token, hipStream_t stream);
