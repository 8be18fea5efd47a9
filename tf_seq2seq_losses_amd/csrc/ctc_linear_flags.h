// Constants of the linear-domain number format (float32 mantissas, one integer exponent per lane) and of the flags D1..D7 that say
// when an utterance leaves it -- shared by the loss + gradient kernel (ctc_fused6.hip) and the Hessian-vector kernel
// (ctc_hvp_fused.hip), which until r04 each carried a hand-kept copy (the copies had drifted -- the HVP's mass check still
// used 1e-4 after fused6 had tightened it to 3e-5).  See ctc_fused6.hip for what each one guards.
#pragma once

namespace ctc {
namespace linear {

constexpr int DEAD = -(1 << 24);  // exponent of a lane whose mantissas are all zero
constexpr int GAP = 16;           // a lane adopts / is lifted to its upstream neighbour's exponent minus GAP, per adoption level
constexpr int GAP_WIDE = 64;      // ... when ONE level suffices (a lane of 4 or 8 label positions is never crossed within a period)
constexpr int DOWN_MAX = 96;      // D3: a renormalisation pushes a lane's own live values down by more than 2^-96
constexpr int DECAY_MAX = 96;     // D4: a lane's maximum decays by more than 2^-96 within one renormalisation period
constexpr int KK_MAX = 90;        // posterior scale 2^KK_MAX at most in ONE factor (fused6 applies the excess to its operand first)
constexpr int KK_MAX2 = 200;      // D5 (fused6): beyond this even the pre-scaled operand would leave float32
constexpr float EMIS_MIN = 7.52316384526264e-37f;  // D2: 2^-120 of the row maximum
constexpr float EMIS_SOFT = 1.52587890625e-05f;    // D7: 2^-16 of the row maximum (loss-only calls)
constexpr float MASS_TOL = 3e-5f; // D6: tolerated deviation of a frame's posterior mass from 1 (1e-4 left no margin: r03 soak)

}  // namespace linear
}  // namespace ctc
