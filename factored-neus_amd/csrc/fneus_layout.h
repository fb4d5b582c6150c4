// Packed-weight blob layouts (compile-time) for the MLPs on the stage-1 hot path.
//
// SDF network   : reference models/fields.py:9-91   dims [39,256,256,256,256(217 out),256,256,256,256,257], skip at 4
// colour network: reference models/fields.py:114-175 dims [289,256,256,256,256,3]
// background NeRF++ (womask): reference models/fields.py:178-259  84 -> 8 x 256 (input re-concatenated before layer 5)
//                 -> alpha 1 | feature 256 -> [feature | PE4(view) 27] -> 128 -> rgb 3
//
// A "fragment" is one MFMA A operand: 64 lanes x 8 bf16 = 1 KiB.  A layer's forward pack holds KS*NT fragments
// ordered [ks][t] (hi plane, then lo plane); the reverse pack (A = W^T) likewise.  Bias packs are fp32 in
// accumulator layout [t][h][16].
#pragma once
#include <stdint.h>

namespace fneus {

constexpr int kFragBytes = 1024;

struct LayerGeom {
    int ksf, ntf;   // forward : k-steps (of 16) and output tiles (of 32)
    int ksr, ntr;   // reverse : k-steps over this layer's outputs, tiles over its inputs
};

constexpr int kSdfLayers = 9;
constexpr LayerGeom kSdfGeom[kSdfLayers] = {
    {3, 8, 16, 2},   // 0: PE(39->48)      -> 256
    {16, 8, 16, 8},  // 1
    {16, 8, 16, 8},  // 2
    {16, 7, 14, 8},  // 3: 256 -> 217 (224)
    {17, 8, 16, 9},  // 4: [h(217->224) ; PE(39->48)] -> 256 ; reverse rows: 7 tiles h + 2 tiles PE
    {16, 8, 16, 8},  // 5
    {16, 8, 16, 8},  // 6
    {16, 8, 16, 8},  // 7
    {16, 9, 18, 8},  // 8: 256 -> 257 (tiles 0..7 = feature rows 1..256, tile 8 row 0 = sdf row)
};

constexpr int kColLayers = 5;
constexpr LayerGeom kColGeom[kColLayers] = {
    {19, 8, 16, 10},  // 0: [feat 256 ; side 33->48] -> 256 ; reverse rows: 8 tiles feat + 2 tiles side
    {16, 8, 16, 8},
    {16, 8, 16, 8},
    {16, 8, 16, 8},
    {16, 1, 2, 8},    // 4: 256 -> 3 (one tile)
};

// One entry per PACK (not per nn.Linear): pts_linears.5 takes [PE 84 | h 256] and is packed as two operand groups
// (entries 5 and 6) that accumulate into the same tiles; feature_linear and alpha_linear share entry 9 (257 rows).
constexpr int kNerfLayers = 12;
constexpr LayerGeom kNerfGeom[kNerfLayers] = {
    {6, 8, 0, 0},     // 0: pts_linears.0   PE10(4-D point) 84 -> 96 k-slots
    {16, 8, 16, 8},   // 1: pts_linears.1
    {16, 8, 16, 8},   // 2
    {16, 8, 16, 8},   // 3
    {16, 8, 16, 8},   // 4: pts_linears.4
    {16, 8, 16, 8},   // 5: pts_linears.5, columns of h (reference columns 84..339)
    {6, 8, 0, 0},     // 6: pts_linears.5, columns of the re-concatenated PE (reference columns 0..83); no reverse: the
                      //    encoding has no trainable ancestor
    {16, 8, 16, 8},   // 7: pts_linears.6
    {16, 8, 16, 8},   // 8: pts_linears.7
    {16, 9, 18, 8},   // 9: feature_linear (tiles 0..7) + alpha_linear (tile 8, row 0)
    {18, 4, 8, 8},    // 10: views_linears.0  [feature 256 ; PE4(view) 27 -> 32] -> 128; reverse rows: the 256 feature inputs
    {8, 1, 2, 4},     // 11: rgb_linear 128 -> 3 (one tile)
};

// Lvis, the stage-2 distilled light-visibility network (reference models/fields.py:338-369), inference only (stage 3 evaluates
// it 4096 times per surface point, inverRender.py:163-180): [PE10(point) 63 -> 64 slots | PE4(direction) 27 -> 32 slots] -> 4 x 256
// ReLU -> 1.  No reverse packs.
constexpr int kLvisLayers = 5;
constexpr LayerGeom kLvisGeom[kLvisLayers] = {
    {6, 8, 0, 0},
    {16, 8, 0, 0},
    {16, 8, 0, 0},
    {16, 8, 0, 0},
    {16, 1, 0, 0},    // 256 -> 1 (one tile, row 0)
};

struct LayerOff {
    uint32_t fwd_hi, fwd_lo, rev_hi, rev_lo, bias;
};

template <int NL>
struct NetLayout {
    LayerOff L[NL];
    uint32_t extra;   // SDF, Lvis: row 0 of the last layer in accumulator layout (8 tiles x 2 x 16 fp32); colour: its 3 rows
    uint32_t total;
};

template <int NL>
constexpr NetLayout<NL> make_layout(const LayerGeom (&g)[NL], int extra_bytes) {
    NetLayout<NL> r{};
    uint32_t off = 0;
    for (int l = 0; l < NL; ++l) {
        r.L[l].fwd_hi = off; off += g[l].ksf * g[l].ntf * kFragBytes;
        r.L[l].fwd_lo = off; off += g[l].ksf * g[l].ntf * kFragBytes;
        r.L[l].rev_hi = off; off += g[l].ksr * g[l].ntr * kFragBytes;
        r.L[l].rev_lo = off; off += g[l].ksr * g[l].ntr * kFragBytes;
        r.L[l].bias = off;   off += g[l].ntf * 2 * 16 * 4;
    }
    r.extra = off; off += extra_bytes;
    r.total = off;
    return r;
}

constexpr NetLayout<kSdfLayers> kSdfLayout = make_layout<kSdfLayers>(kSdfGeom, 8 * 2 * 16 * 4);
constexpr NetLayout<kColLayers> kColLayout = make_layout<kColLayers>(kColGeom, 3 * 8 * 2 * 16 * 4);      // extra: the 3 rows of the last layer
constexpr NetLayout<kNerfLayers> kNerfLayout = make_layout<kNerfLayers>(kNerfGeom, 0);
constexpr NetLayout<kLvisLayers> kLvisLayout = make_layout<kLvisLayers>(kLvisGeom, 8 * 2 * 16 * 4);     // extra: row 0 of the last layer

// flat fp32 parameter layouts (natural order): for each layer W[out][in] row-major, then b[out]
constexpr int kSdfIn[kSdfLayers]  = {39, 256, 256, 256, 256, 256, 256, 256, 256};
constexpr int kSdfOut[kSdfLayers] = {256, 256, 256, 217, 256, 256, 256, 256, 257};
constexpr int kColIn[kColLayers]  = {289, 256, 256, 256, 256};
constexpr int kColOut[kColLayers] = {256, 256, 256, 256, 3};

}  // namespace fneus
