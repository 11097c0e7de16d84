// libcrx: the obstacle instantiations crx_solve_kernel<1..3, *> (MPC-CBF NLP, tracking NLP with CBF rows) as their own
// translation unit -- the source is crx_kernels.hip, compiled here with the machine scheduler's iterative-ilp strategy
// (Makefile; measured in tools/gpu_round3_l.sh, see the note in crx_kernels.hip section (6)).
#define CRX_TU_OBSTACLES 1
#include "crx_kernels.hip"
