// launcher of the stage-3 visibility kernel in the two-pass pipelined form (lvis_p2_kernels.hip), called by fneus_lvis_visibility
#pragma once
#include <hip/hip_runtime.h>

namespace fneus {

int lvis_visibility_p2(const unsigned char* blob, const float* points, const float* normals, const float* dirs, const float* weights,
                       const unsigned char* point_mask, int n_pts, int n_lobes, float* vis, int prec, hipStream_t stream);

}  // namespace fneus
