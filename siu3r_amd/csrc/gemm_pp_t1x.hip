// The 256 x 256 bf16x3 instantiations of the ping-pong GEMM (gemm_pp_kernel.h): a translation unit per tile shape and operand mode, so
// that the six compile in parallel (one unit with all thirty variants took four minutes).
#include "gemm_pp_kernel.h"

#ifndef SIU3R_PP_MINI
void siu3r_gemm_pp_go_t1x(const siu3r_gemm_params& p, int mode, bool lnf, dim3 grid, hipStream_t s) {
  using namespace siu3r_gemm_pp;
  const dim3 block(512);
#define SIU3R_PP_GO(MODE_, RELU_, LNF_) hipLaunchKernelGGL((gemm_pp_kernel<true, 2, 4, MODE_, RELU_, LNF_>), grid, block, 0, s, p)
  if (mode == 0 && p.a_x3) {  // pre-split A operand (hi | lo planes from the producer's epilogue)
    if (lnf) hipLaunchKernelGGL((gemm_pp_kernel<true, 2, 4, 0, false, true, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_pp_kernel<true, 2, 4, 0, false, false, true>), grid, block, 0, s, p);
  } else if (mode == 0 && lnf) SIU3R_PP_GO(0, false, true);
  else if (mode == 0) SIU3R_PP_GO(0, false, false);
  else if (mode == 1 && p.a_x3) hipLaunchKernelGGL((gemm_pp_kernel<true, 2, 4, 1, false, false, true>), grid, block, 0, s, p);
  else if (mode == 1 && p.relu_in) SIU3R_PP_GO(1, true, false);
  else if (mode == 1) SIU3R_PP_GO(1, false, false);
  else SIU3R_PP_GO(2, false, false);
#undef SIU3R_PP_GO
}
#endif
