// libcrx: the TWO-WAVE instantiations crx_solve_kernel<1, 12, 6, {12, 10}, SPEC = 1> as their own translation unit [r6] -- same source as the tuned
// obstacle instantiations (crx_kernels.hip), same build flags (Makefile: iterative-ilp), one more wave per problem: while wave 0 factorises the
// reduced Hessian with the current entry of the inertia-correction schedule, wave 1 factorises it with the next one in a second set of work arrays
// (crx_kernels.hip, crx_solve_kernel: SPEC).  Used for launches that leave SIMDs idle (crx_api.hip launch_solve: batch <= 2 x CUs); the iterates
// are the one-wave kernel's bit for bit (tests/test_gpu_parity.py::test_speculating_wave_is_bit_identical).
#define CRX_TU_OBSTACLES 1      /* not the main unit: no selection kernel, no diagnostics */
#define CRX_TU_SPEC 1
#include "crx_kernels.hip"
