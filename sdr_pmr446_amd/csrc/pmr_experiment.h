/* pmr_experiment.h -- ONE gate for every compile-time experiment hook of the library.
 *
 * The sources carry hooks that exist to MEASURE, not to ship: timing-only builds that give wrong results on purpose (a launch left
 * out, a phase boundary the kernel stops at, stores sent into a 4 KB window: profiles/r05_ab_log.txt, r06_ab_log.txt) and tuning
 * knobs of the tile shapes.  None of them may reach the library the headline is measured with, so:
 *   - every hook macro is listed below, and defining any of them without -DPMR_EXPERIMENT is a compile error;
 *   - `python3 sdr_pmr446_amd/build.py --variant NAME "-DFLAGS"` (-> build_ab/NAME/, never the in-tree library) adds
 *     -DPMR_EXPERIMENT by itself; the product build (build.build()) takes no extra flags at all;
 *   - a library compiled with it says so: pmr_chain_info(NULL, PMR_INFO_EXPERIMENT_BUILD, 0) == 1 (include/pmr_chain.h), and
 *     bench.py refuses to print a headline from such a library (or from any library selected by PMR_LIBRARY).
 * Included first by pmr_kernels.h, i.e. by every unit that has a hook, before any hook's default value is defined. */
#ifndef PMR_EXPERIMENT_H
#define PMR_EXPERIMENT_H

#ifdef PMR_EXPERIMENT
#define PMR_EXPERIMENT_BUILD 1
#else
#define PMR_EXPERIMENT_BUILD 0
#if defined(EXP_SKIP_FIR) || defined(EXP_SKIP_CHAN) || defined(EXP_SKIP_L2) || defined(EXP_SKIP_CT) || defined(EXP_NO_TAIL) ||          \
    defined(EXP_NO_CARRY5) || defined(EXP_ARG_CHEAP) || defined(EXP_BE_WIN) || defined(EXP_L2_NOFIX) || defined(EXP_L2_STOP1) ||        \
    defined(EXP_CT_NO_AGG) || defined(EXP_CT_NO_SCAN) || defined(EXP_CT_NO_GOERTZEL) || defined(EXP_CT_NO_FINAL) ||                     \
    defined(EXP_CG_EXTRA_LDS) || defined(EXP_L2_INLINE) || defined(EXP_L2_INLINE_ATOMIC) || defined(EXP_L1_EXTRA_HALO) || defined(EXP_FE_PRIO_EQUAL) || defined(EXP_FE_PRIO_HIGH) ||                                             \
    defined(PMR_CARRY_NOOP) || defined(CW_STOP) || defined(CW_NT) ||                                                                    \
    defined(FE_STOP) || defined(FE_STAMP) || defined(FE_OUT_AND) || defined(FE_OUT_SKIP) || defined(FE_OUT_NT) || defined(FE_OUT_SC) || \
    defined(FE_NO_PAIRS) || defined(FE_LAST_LDS) || defined(FE_S1_LDS) || defined(FE_L1_LDS23) || defined(FE_NO_TIGHT) || defined(FE_DMA_AUX) || defined(FE_EXTRA_LDS) || defined(FE_LDS_PAD_256) || defined(FE_LDS_PAD_16) ||          \
    defined(PW_F) || defined(PW_FPW1024) || defined(PW_MINB) || defined(PW_RB) || defined(PF_G) || defined(PF_RB) || defined(PF_MINB)
#error "experiment hook defined without -DPMR_EXPERIMENT: build with `python3 sdr_pmr446_amd/build.py --variant NAME \"-DFLAGS\"` (pmr_experiment.h)"
#endif
#endif

#endif
