// libcrx: the GENERAL instantiations of crx_solve_kernel as their own translation unit [r5] -- run-time horizon (every N other than
// 10 / 12 / 20), run-time exponent (CBF degree 2 / 4 / 8), and the generic 4..6-obstacle ones.  Same source as the tuned
// instantiations (crx_kernels.hip), built CONSERVATIVELY: these are the instantiations in which two A/B builds of round 4 computed
// wrong numbers under 256 VGPRs + 80..250 AGPRs, so they give up what those builds had in common:
//   * 256 registers, no AGPRs: dynamic LDS + a floor of two waves per SIMD (with the static layout the compiler sees that the LDS admits
//     one wave per SIMD and hands the kernel the whole unified register file); what does not fit is spilled to scratch memory;
//   * no inline-assembly DPP: the sweeps broadcast with v_readlane, every hazard is the compiler's to see;
//   * nothing lane-derived is carried across the interior-point loop: the lane index is made opaque once per iteration.
// Slower than the tuned path by design (correct first); every BASELINE config runs on the tuned instantiations.
#define CRX_TU_GENERAL 1
#define CRX_TU_OBSTACLES 1      /* not the main unit: no selection kernel, no diagnostics */
#ifndef CRX_STATIC_LDS
#define CRX_STATIC_LDS 0
#endif
#ifndef CRX_GEN_WAVES
#define CRX_GEN_WAVES 2
#endif
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
