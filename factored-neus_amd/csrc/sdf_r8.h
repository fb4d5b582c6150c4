// launchers of the resident-weight 8-wave SDF kernels (sdf_r8_kernels.hip), called by the C-ABI entry points in sdf_kernels.hip
#pragma once
#include "fneus_kernels.h"

namespace fneus {

// reverse sweep of K2 (normal, a_l planes) on the sigma' blocks of a forward launch; 64-sample groups g_begin .. g_end - 1
int sdf_grad_rev_r8(const unsigned char* blob, const PointSrc& src, long n_pts, const SdfStash& st, float* normal_out, int prec, int train,
                    int gp, long g_begin, long g_end, hipStream_t stream);

// K3 (both backward chains; writes the planes of SdfBwdBufs); gp = 3: hi + lo planes
int sdf_bwd_r8(const unsigned char* blob, const PointSrc& src, long n_pts, const SdfStash& st, const SdfBwdBufs& bb, const float* d_sdf,
               const float* d_feat, const float* d_normal, int prec, int gp, hipStream_t stream);

}  // namespace fneus
