// Launchers of the low-rank box-QP kernels (qp_lr.h).
#include "qp_lr.h"

int gq_qp_lr_launch_iter(const GqQpArgs& a, hipStream_t st) {
  const dim3 grid(a.B), block(GQ_WAVE);
  const bool two = a.nz > GQ_WAVE;
  if (a.m <= 6) {
    if (two) hipLaunchKernelGGL((gq_qp_lr_iter_kernel<6, 2>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gq_qp_lr_iter_kernel<6, 1>), grid, block, 0, st, a);
  } else {
    if (two) hipLaunchKernelGGL((gq_qp_lr_iter_kernel<8, 2>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gq_qp_lr_iter_kernel<8, 1>), grid, block, 0, st, a);
  }
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_qp_lr_launch_bwd(const GqQpBwdArgs& a, hipStream_t st) {
  const dim3 grid(a.B), block(GQ_WAVE);
  const bool two = a.nz > GQ_WAVE;
  if (a.m <= 6) {
    if (two) hipLaunchKernelGGL((gq_qp_lr_bwd_kernel<6, 2>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gq_qp_lr_bwd_kernel<6, 1>), grid, block, 0, st, a);
  } else {
    if (two) hipLaunchKernelGGL((gq_qp_lr_bwd_kernel<8, 2>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gq_qp_lr_bwd_kernel<8, 1>), grid, block, 0, st, a);
  }
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}
