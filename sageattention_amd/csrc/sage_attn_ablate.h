// Timing-only ablations and mechanism probes of attn_i8_kernel, selected with -D on a VARIANT build
// (sageattention_amd/_build.py build_variant, tools/ab_bench.py, tools/build_probe_libs.py).  The product library is
// built with none of them: every switch below is then a compile-time false and its branch is discarded.
// A build with a SAGE_ABL_* switch computes WRONG results by design (it measures what a part of the loop costs).
#pragma once

#ifndef SAGE_MINWAVES  // second argument of the kernel's __launch_bounds__ (occupancy experiments)
#define SAGE_MINWAVES 2
#endif

namespace sage {
namespace abl {

#define SAGE_ABL_FLAG(name, macro) constexpr bool name = macro
#ifdef SAGE_ABL_SAMETILE   // every K/V tile copy re-reads tile 0 (vector-L1 hits): the L2-latency share of staging
SAGE_ABL_FLAG(kSameTile, true);
#else
SAGE_ABL_FLAG(kSameTile, false);
#endif
#ifdef SAGE_ABL_NOQK       // no K fragment reads, no S MFMAs
SAGE_ABL_FLAG(kNoQK, true);
#else
SAGE_ABL_FLAG(kNoQK, false);
#endif
#ifdef SAGE_ABL_NOEXP      // fma instead of fma + exp2
SAGE_ABL_FLAG(kNoExp, true);
#else
SAGE_ABL_FLAG(kNoExp, false);
#endif
#ifdef SAGE_ABL_NOPV       // no P.V MFMAs (generic body)
SAGE_ABL_FLAG(kNoPV, true);
#else
SAGE_ABL_FLAG(kNoPV, false);
#endif
#ifdef SAGE_ABL_NOSTAGE    // no tile copies inside the fast loop
SAGE_ABL_FLAG(kNoStage, true);
#else
SAGE_ABL_FLAG(kNoStage, false);
#endif
#ifdef SAGE_ABL_NOLDSK     // fast loop: no K fragment reads
SAGE_ABL_FLAG(kNoLdsK, true);
#else
SAGE_ABL_FLAG(kNoLdsK, false);
#endif
#ifdef SAGE_ABL_NOLDSV     // fast loop: no V^T fragment reads
SAGE_ABL_FLAG(kNoLdsV, true);
#else
SAGE_ABL_FLAG(kNoLdsV, false);
#endif
#if defined(SAGE_ABL_HALFLDS)     // fp16 stream: every second K / V^T fragment read and every second tile copy dropped -- what a wave
constexpr bool kHalfReads = true, kHalfCopies = true;   // with TWO 32-row sub-tiles (fragments shared) would save per row, at today's occupancy
#elif defined(SAGE_ABL_HALFREADS) // ... only the fragment reads
constexpr bool kHalfReads = true, kHalfCopies = false;
#elif defined(SAGE_ABL_HALFCOPIES) // ... only the tile copies
constexpr bool kHalfReads = false, kHalfCopies = true;
#else
constexpr bool kHalfReads = false, kHalfCopies = false;
#endif
#ifdef SAGE_ABL_CONSTODD   // control of HALFLDS: every fragment is still READ and every tile copied, but every second MFMA is fed a
SAGE_ABL_FLAG(kConstOdd, true);  // constant register instead (the share of HALFLDS's gain that is operand-toggle power, not LDS)
#else
SAGE_ABL_FLAG(kConstOdd, false);
#endif
#ifdef SAGE_ABL_NOROWSUM_F8 // FP8 stream: 8 instead of 32 row-sum adds per tile (what the adds cost: C4 +4 %, D=64 8K +7 %)
SAGE_ABL_FLAG(kNoRowSumF8, true);
#else
SAGE_ABL_FLAG(kNoRowSumF8, false);
#endif
#ifdef SAGE_ABL_NOBAR      // fast loop: no per-tile workgroup barrier
SAGE_ABL_FLAG(kNoBar, true);
#else
SAGE_ABL_FLAG(kNoBar, false);
#endif

// Cross-check builds (results must NOT change by one bit)
#ifdef SAGE_ABL_ALLGENERIC // every tile through the generic (masked) body
SAGE_ABL_FLAG(kAllGeneric, true);
#else
SAGE_ABL_FLAG(kAllGeneric, false);
#endif
#ifdef SAGE_NO_ODD_FAST    // an odd remaining fast tile goes through the generic body
SAGE_ABL_FLAG(kNoOddFast, true);
#else
SAGE_ABL_FLAG(kNoOddFast, false);
#endif
#if defined(SAGE_SCHED_COMPILER)       // both fast loops scheduled by hipcc instead of the hand-placed streams
constexpr int kHandPlacedF16 = 0, kHandPlacedF8 = 0;
#elif defined(SAGE_SCHED_COMPILER_FP8) // only the FP8 loop left to hipcc
constexpr int kHandPlacedF16 = 1, kHandPlacedF8 = 0;
#else
constexpr int kHandPlacedF16 = 1, kHandPlacedF8 = 2;
#endif
#ifdef SAGE_EXP_PRIO       // static s_setprio for the second-dispatched half of the workgroup
constexpr int kPrio = SAGE_EXP_PRIO;
#else
constexpr int kPrio = -1;
#endif

#ifdef SAGE_EXP_LIGHT_FIRST  // causal: lightest q-blocks of a head first (measured: -1..2 %, MORE L2 misses; see the kernel)
SAGE_ABL_FLAG(kLightFirst, true);
#else
SAGE_ABL_FLAG(kLightFirst, false);
#endif

// Mechanism probes of the round-2 race fix (profiles/r02_race_evidence.md, tools/race_probe.sh)
#ifdef SAGE_EXP_DELAY_WAVE           // wave 1 sleeps ~8 us between the prologue barrier and its K(0) fragment reads
SAGE_ABL_FLAG(kDelayWave, true);
#else
SAGE_ABL_FLAG(kDelayWave, false);
#endif
#ifdef SAGE_EXP_NO_PROLOGUE_BARRIER  // round-1 structure: no barrier between the prologue S(0) and the first K(2) copy
SAGE_ABL_FLAG(kNoPrologueBarrier, true);
#else
SAGE_ABL_FLAG(kNoPrologueBarrier, false);
#endif
#ifdef SAGE_EXP_CTEMP                // round-1 form of the first S MFMA: C operand = a re-materialised temporary
SAGE_ABL_FLAG(kCTemp, true);
#else
SAGE_ABL_FLAG(kCTemp, false);
#endif
// Row sums of the fp16 P on v_mfma_f32_16x16x32_f16 (see the kernel): the product does this at head_dim 64 only
#ifdef SAGE_EXP_MFMA_ROWSUM_D128     // ... also at head_dim 128 (measured: C3 +0.2 %, C3-causal +0.8 %; rounded-P sums)
SAGE_ABL_FLAG(kMfmaRowSum128, true);
#else
SAGE_ABL_FLAG(kMfmaRowSum128, false);
#endif
#ifdef SAGE_EXP_VALU_ROWSUM_D64      // ... nowhere: the round-1/2 form, 32 v_add_f32 of the unrounded p per tile
SAGE_ABL_FLAG(kValuRowSum64, true);
#else
SAGE_ABL_FLAG(kValuRowSum64, false);
#endif
#undef SAGE_ABL_FLAG

}  // namespace abl
}  // namespace sage
