// clm_lab.h -- the ONE place where timing-only builds touch the kernels.
//
// The product build (python -m chimeralm_amd.build) never defines CLM_LAB: every switch below is then a compile-time `false`
// and the branches it guards are dead code that does not reach the object file.  tools/build_variant.sh NAME -DCLM_LAB -DCLM_EXP_x
// builds a library in which ONE kind of work is removed (results WRONG by construction, instruction stream otherwise the same);
// tools/abn.sh alternates it with the product build on one box.  What those builds measured is in profiles/r03_timing_only.txt
// and HISTORY.md section 4.9.
#pragma once

namespace clm {
namespace lab {
#if defined(CLM_LAB) && defined(CLM_EXP_A0)
constexpr bool A0 = true;          // every A fragment of a weight set is its first one: one LDS read per set instead of 16
#else
constexpr bool A0 = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_NOLO)
constexpr bool NOLO = true;        // no lo half in the compensated products: what the fp8 MFMAs + the byte gathers cost
#else
constexpr bool NOLO = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_NOLN)
constexpr bool NOLN = true;        // no LayerNorm statistics (one barrier kept: it orders earlier LDS reads before the tile writes)
#else
constexpr bool NOLN = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_NOW)
constexpr bool NOW = true;         // only the first weight set of a column block is ever loaded: no weight stream at all
#else
constexpr bool NOW = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_W0)
constexpr bool W0 = true;          // every weight set is the wave's first one: weights from the L1
#else
constexpr bool W0 = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_NOGELU)
constexpr bool NOGELU = true;      // no transcendental in the MLP's GELU
#else
constexpr bool NOGELU = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_NOZSTORE)
constexpr bool NOZSTORE = true;    // z is never written (one lane keeps the data alive)
#else
constexpr bool NOZSTORE = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_NOFIR)
constexpr bool NOFIR = true;       // no short filter, no lane exchange in the gated in_proj epilogue
#else
constexpr bool NOFIR = false;
#endif
// round 4: what the activations' lo bytes cost, piece by piece
#if defined(CLM_LAB) && defined(CLM_EXP_NOLO2)
constexpr bool NOLO2 = true;       // no activations' lo term in the compensated products (no lo-tile reads, no weight-byte gather, no MFMA)
#else
constexpr bool NOLO2 = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_NOLOPACK)
constexpr bool NOLOPACK = true;    // producers write zero lo bytes instead of packing them (LayerNorm tiles, z rows)
#else
constexpr bool NOLOPACK = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_NOZLO)
constexpr bool NOZLO = true;       // the gated in_proj stage stages and stores no lo planes (halfs only)
#else
constexpr bool NOZLO = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_NOYLO)
constexpr bool NOYLO = true;       // the tail kernel neither loads nor stages y's lo bytes
#else
constexpr bool NOYLO = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_YLO_NOMFMA)
constexpr bool YLO_NOMFMA = true;  // y's lo plane loaded and staged, but out_proj without its lo term
#else
constexpr bool YLO_NOMFMA = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_YLO_NOLOAD)
constexpr bool YLO_NOLOAD = true;  // y's lo plane not loaded: zeros staged instead (transposes, LDS writes and the lo term as in the product)
#else
constexpr bool YLO_NOLOAD = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_YLO_NOPERM)
constexpr bool YLO_NOPERM = true;  // y's lo plane loaded and written to the tile without the 4 x 4 byte transposes
#else
constexpr bool YLO_NOPERM = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_YLO_MAP64)
constexpr bool YLO_MAP64 = true;   // y's lo tile: a lane = a 4-channel column, a wave = 16 tokens (64 lines per load instruction; round 4's first form)
#else
constexpr bool YLO_MAP64 = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_NORESID)
constexpr bool NORESID = true;     // the tail kernel loads no residual rows (zeros): what the loads at the tile boundary cost
#else
constexpr bool NORESID = false;
#endif
// round 5: placement experiments of the next tile's prefetches in the gated in_proj stage (results stay correct)
#if defined(CLM_LAB) && defined(CLM_EXP_YLATE)
constexpr bool YLATE = true;       // y pieces + y's lo bytes requested behind the x0 block's last weight set instead of in its hooks
#else
constexpr bool YLATE = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_LN1PASS)
constexpr bool LN1PASS = true;     // LayerNorm statistics as sum and sum of squares in ONE exchange (var = E[x^2] - mean^2: not the reference's two-pass form)
#else
constexpr bool LN1PASS = false;
#endif
#if defined(CLM_LAB) && defined(CLM_EXP_AHEAD)
constexpr int AHEAD = CLM_EXP_AHEAD;   // A fragments requested this many (k-step, row tile) items before their MFMA (product: 4)
#else
constexpr int AHEAD = 4;
#endif
}  // namespace lab
}  // namespace clm
