// Packed-weight blob layouts for the 16-sample-tile engine (v_mfma_f32_16x16x32_bf16): output tiles of 16 features,
// k-steps of 32 features.  Same networks as fneus_layout.h.
#pragma once
#include "fneus_layout.h"

namespace fneus {

// {ksf, ntf, ksr, ntr}: forward k-steps(32) / output tiles(16); reverse k-steps over outputs / tiles over inputs
constexpr LayerGeom kSdfGeom16[kSdfLayers] = {
    {2, 16, 8, 3},    // 0: PE(39->64)            -> 256 ; reverse rows: 39 -> 3 tiles
    {8, 16, 8, 16},   // 1
    {8, 16, 8, 16},   // 2
    {8, 14, 7, 16},   // 3: 256 -> 217 (224 = 14 tiles)
    {9, 16, 8, 17},   // 4: [h(224: 7 k-steps) ; PE(64: 2 k-steps)] -> 256 ; reverse rows: 14 tiles h + 3 tiles PE
    {8, 16, 8, 16},   // 5
    {8, 16, 8, 16},   // 6
    {8, 16, 8, 16},   // 7
    {8, 17, 9, 16},   // 8: 256 -> 257 (tiles 0..15 = feature rows 1..256, tile 16 row 0 = sdf row); reverse k: 17 tiles -> 9 k-steps
};
constexpr LayerGeom kColGeom16[kColLayers] = {
    {10, 16, 8, 19},  // 0: [feat 256 (8 k-steps) ; side 33->64 (2 k-steps)] -> 256 ; reverse rows: 16 + 3 tiles
    {8, 16, 8, 16},
    {8, 16, 8, 16},
    {8, 16, 8, 16},
    {8, 1, 1, 16},    // 4: 256 -> 3 (one tile); reverse k: 1 tile -> 1 k-step (second half zero)
};

constexpr NetLayout<kSdfLayers> kSdfLayout16 = make_layout<kSdfLayers>(kSdfGeom16, 16 * 16 * 4);
constexpr NetLayout<kColLayers> kColLayout16 = make_layout<kColLayers>(kColGeom16, 0);

}  // namespace fneus
