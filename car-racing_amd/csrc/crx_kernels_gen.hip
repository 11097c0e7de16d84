// libcrx: the GENERAL instantiations of crx_solve_kernel as their own translation unit [r5] -- run-time horizon (every N other than
// 10 / 12 / 20), run-time exponent (CBF degree 2 / 4 / 8), and the generic 4..6-obstacle ones.  Same source as the tuned
// instantiations (crx_kernels.hip), built CONSERVATIVELY: these are the instantiations in which A/B builds of rounds 4 and 5 computed
// wrong numbers (DESIGN.md section 8 has the two causes, both in the compiler's hands: LDS accesses scheduled across a phase boundary,
// and a VGPR spill store placed in front of a loop exit's EXEC restore).  What this unit gives up:
//   * no inline-assembly DPP: the sweeps broadcast with v_readlane, every hazard is the compiler's to see;
//   * nothing lane-derived is carried across the interior-point loop: the lane index is made opaque once per iteration (half the
//     register pressure of the hoisted lane maps);
//   * the default machine scheduler (Makefile), not iterative-ilp.
// Registers: static LDS, the whole unified register file (AGPR copies where 256 VGPRs do not suffice).  The other choice -- dynamic LDS +
// a floor of two waves per SIMD = 256 registers and scratch spills, -DCRX_STATIC_LDS=0 -DCRX_GEN_WAVES=2 -- was the shipped one for a few
// hours of round 5 and is where the spill-placement bug bit (tools/exec_prologue_check.py finds the pattern in the assembly; it is a
// test: every translation unit of the library must be free of it).
// Slower than the tuned path by design (correct first); every BASELINE config runs on the tuned instantiations.
#define CRX_TU_GENERAL 1
#define CRX_TU_OBSTACLES 1      /* not the main unit: no selection kernel, no diagnostics */
#ifndef CRX_ROWDPP
#define CRX_ROWDPP 0
#endif
#ifndef CRX_OPAQUE_LANE
#define CRX_OPAQUE_LANE 2
#endif
#ifndef CRX_SWEEP_LOCAL_LANE
#define CRX_SWEEP_LOCAL_LANE 0
#endif
#include "crx_kernels.hip"
