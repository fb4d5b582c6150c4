// Library-level C ABI: version, last error, layout query.
#include <string.h>
#include "fneus_kernels.h"
#include "fneus_layout.h"

namespace fneus {
static char g_err[256] = "";
void set_last_error(const char* msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
template <int NL>
static int write_layout(const NetLayout<NL>& ly, const LayerGeom (&g)[NL], int32_t* out, int cap) {
    const int need = 3 + 9 * NL;
    if (cap < need) return -2;
    out[0] = NL;
    out[1] = (int32_t)ly.total;
    out[2] = (int32_t)ly.extra;
    for (int l = 0; l < NL; ++l) {
        int32_t* o = out + 3 + 9 * l;
        o[0] = ly.L[l].fwd_hi; o[1] = ly.L[l].fwd_lo; o[2] = ly.L[l].rev_hi; o[3] = ly.L[l].rev_lo; o[4] = ly.L[l].bias;
        o[5] = g[l].ksf; o[6] = g[l].ntf; o[7] = g[l].ksr; o[8] = g[l].ntr;
    }
    return need;
}
}  // namespace fneus

extern "C" int fneus_version(void) { return 100; }
extern "C" const char* fneus_last_error(void) { return fneus::g_err; }
extern "C" int fneus_layout(int which, int32_t* out, int cap) {
    if (which == 0) return fneus::write_layout(fneus::kSdfLayout, fneus::kSdfGeom, out, cap);
    if (which == 1) return fneus::write_layout(fneus::kColLayout, fneus::kColGeom, out, cap);
    if (which == 2) return fneus::write_layout(fneus::kNerfLayout, fneus::kNerfGeom, out, cap);
    if (which == 3) return fneus::write_layout(fneus::kLvisLayout, fneus::kLvisGeom, out, cap);
    return -2;
}
