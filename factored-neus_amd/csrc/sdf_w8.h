// launchers of the 8-wave-workgroup SDF kernels (sdf_w8_kernels.hip), called by the C-ABI entry points in sdf_kernels.hip
#pragma once
#include "fneus_kernels.h"

namespace fneus {

// K1 in the two-pass pipelined form (sdf_p2_kernels.hip): 128 samples per 4-wave workgroup
int sdf_fwd_p2(const unsigned char* blob, const PointSrc& src, long n_pts, float* sdf_out, int prec, int tn, hipStream_t stream);
int sdf_fwd_p2_rays(const unsigned char* blob, const PointSrc& src, long n_pts, const unsigned char* ray_mask, float fill, int32_t* work,
                    float* sdf_out, int prec, hipStream_t stream);

// forward chain of K2 with the stash (sdf_p2_train_kernels.hip); mode 0: inference, 1 / 3: training with bf16 / hi + lo planes;
// the launch covers the 128-sample units unit_begin .. unit_end - 1 of the n_pts samples
int sdf_fwd_stash_p2(const unsigned char* blob, const PointSrc& src, long n_pts, const SdfStash& st, float* sdf_out, float* feat_out,
                     int prec, int mode, long unit_begin, long unit_end, hipStream_t stream);

// K1 for latency-bound launches: one tile per 8-wave workgroup, the whole layer's weight fragments primed in registers
int sdf_fwd_w8p(const unsigned char* blob, const PointSrc& src, long n_pts, float* sdf_out, int prec, hipStream_t stream);

}  // namespace fneus
