// Build-switch guard of the HIP sources (included first by every kernel header).
//
// The kernels carry two kinds of compile-time switches for tools/build_variant.sh A/B arms:
//   * schedule / placement variants whose results are those of the default build (FA_DMA_SPREAD, FA_SGB_VARIANT, FA_QB, ...);
//   * TIMING-ONLY ablations (FA_ABL_*, FA8_ABL_*, FA_BWD_ABL_*) that drop work and return WRONG results.
// A stray -D of the second kind must not be able to produce a library that looks shippable: such a switch compiles only next
// to -DFA_TIMING_ONLY, and fa_build_is_default() (include/fa_mi355.h; checked by tests/test_host_logic.py and by
// __graft_entry__.build()) reports whether ANY non-default switch was set for this library.
#pragma once

#if defined(FA_ABL_NOVALU) || defined(FA_ABL_NOEXP) || defined(FA_ABL_NOSUM) || defined(FA_ABL_NOLDS) || defined(FA_ABL_NODMA) || \
    defined(FA_ABL_NOBARRIER) || defined(FA_ABL_NOVMWAIT) || defined(FA_ABL_MFMA16) || defined(FA8_ABL_NODMA) || \
    defined(FA8_ABL_NOBARRIER) || defined(FA_BWD_ABL_NOSC) || defined(FA_BWD_ABL_NOEW) || defined(FA_BWD_ABL_NOBAR) || \
    defined(FA_BWD_ABL_NOSCORE) || defined(FA_BWD_ABL_NOGRAD) || defined(FA_BWD_ABL_NOGR) || defined(FA_BWD_ABL_NODMA)
#define FA_BUILD_WRONG_RESULTS 1
#if !defined(FA_TIMING_ONLY)
#error "timing-only ablation switch (FA_ABL_* / FA8_ABL_* / FA_BWD_ABL_*) without -DFA_TIMING_ONLY: this build would return wrong results"
#endif
#endif

// any switch that changes the generated code or the host heuristics away from the shipped default
#if defined(FA_BUILD_WRONG_RESULTS) || defined(FA_TIMING_ONLY) || defined(FA_STAMP) || defined(FA_BWD_STAMP) || defined(FA_MFMA32) || \
    defined(FA_QB) || defined(FA_ASM_SMFMA) || defined(FA_ASM_ADD) || defined(FA_PRIO) || defined(FA_NO_SGB) || defined(FA8_NO_SGB) || \
    defined(FA_NO_ROWS128) || defined(FA_SGB_VARIANT) || defined(FA_EARLY_EPILOGUE) || defined(FA_CONT_RING) || \
    defined(FA_CONT_EARLY_Q) || defined(FA_HALF_PRIO) || defined(FA8_HALF_PRIO) || defined(FA_BWD_HALF_PRIO) || defined(FA_DMA_SPREAD) || \
    defined(FA8_GAP_FREE) || defined(FA_FP8_K32) || defined(FA_FP8_PV_BF16) || defined(FA_FP8_CONVERT_ALL) || defined(FA_BWD_ONLY) || \
    defined(FA_BWD_DKDV_SINGLE) || defined(FA_BWD_EXPERIMENTS) || defined(FA_BWD_DS_DISABLE) || defined(FA_BWD_DS_ALWAYS) || \
    defined(FA_BWD_DS_STORE_AUX) || defined(FA_BWD_DS_LOAD_NT) || defined(FA_BWD_DMA_ALL) || defined(FA_FWD_EXPERIMENTS) || \
    defined(FA8_SAMPLED_CHECK) || defined(FA_ROWS128_MFMA32) || defined(FA8_EARLY_EPILOGUE) || defined(FA8_CONT_RING) || \
    defined(FA8_ODD_UNMASKED) || defined(FA_WIDE_QT) || defined(FA_BWD_WIDE_QT)
#define FA_BUILD_NON_DEFAULT 1
#else
#define FA_BUILD_NON_DEFAULT 0
#endif
