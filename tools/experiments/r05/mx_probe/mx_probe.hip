// Probe of the gfx950 block-scaled MFMA path with fp6 (e2m3) operands, on exact data (round 5, DESIGN.md section 4.4 plan (ii)):
//   1. bit layout and element order of v_cvt_scalef32_2xpk16_fp6_f32 / v_cvt_scalef32_pk32_fp6_f16, meaning of their scale operand;
//   2. operand lane map of v_mfma_scale_f32_32x32x64_f8f6f4 with cbsz = blgp = 2 (fp6): lane l, element j <-> (row / column, k);
//   3. which lane's scale byte applies to which (row, k-block), and that the result is sum a b 2^(sa - 127) 2^(sb - 127).
// Build: hipcc --offload-arch=gfx950 -O2 mx_probe.hip -o mx_probe ; run on the GPU box, prints PASS / FAIL lines.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(6))) unsigned int u32x6;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(32))) float f32x32;
typedef __attribute__((ext_vector_type(32))) _Float16 f16x32;

static float fp6_val(int c) {          // e2m3 code (6 bits) -> value
    const int s = c >> 5, e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m / 8.0f : (1.0f + m / 8.0f) * (float)(1 << (e - 1));
    return s ? -v : v;
}

__global__ void cvt_kernel(const float* in, uint32_t* bits32, uint32_t* bits16, float* back, float scale) {
    const int l = threadIdx.x;
    f32x16 a, b;
    f16x32 hh;
    for (int i = 0; i < 16; ++i) { a[i] = in[l * 32 + i]; b[i] = in[l * 32 + 16 + i]; }
    for (int i = 0; i < 32; ++i) hh[i] = (_Float16)in[l * 32 + i];
    u32x6 q = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    u32x6 q2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hh, scale);
    f32x32 bk = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(q, 1.0f);
    for (int i = 0; i < 6; ++i) { bits32[l * 6 + i] = q[i]; bits16[l * 6 + i] = q2[i]; }
    for (int i = 0; i < 32; ++i) back[l * 32 + i] = bk[i];
}

template <int OPA, int OPB>
__global__ void mfma_kernel(const uint32_t* A, const uint32_t* B, const int* sa, const int* sb, float* out) {
    const int l = threadIdx.x;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    for (int i = 0; i < 6; ++i) { a[i] = (int)A[l * 6 + i]; b[i] = (int)B[l * 6 + i]; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, OPA, sa[l], OPB, sb[l]);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
}

static void pack6(uint32_t* dst, const int* codes) {      // 32 codes -> 192 bits, element j at bits [6j, 6j + 6)
    for (int i = 0; i < 6; ++i) dst[i] = 0;
    for (int j = 0; j < 32; ++j) {
        const int bit = 6 * j;
        const uint64_t v = (uint64_t)(codes[j] & 63) << (bit & 31);
        dst[bit >> 5] |= (uint32_t)v;
        if ((bit & 31) > 26) dst[(bit >> 5) + 1] |= (uint32_t)(v >> 32);
    }
}

int main() {
    // ---- 1. the conversions
    std::vector<float> in(64 * 32);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) in[l * 32 + j] = fp6_val((l * 5 + j * 11 + 3) & 63);
    float *d_in, *d_back; uint32_t *d_b32, *d_b16;
    hipMalloc(&d_in, in.size() * 4); hipMalloc(&d_back, in.size() * 4); hipMalloc(&d_b32, 64 * 6 * 4); hipMalloc(&d_b16, 64 * 6 * 4);
    hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    for (float scale : {1.0f, 2.0f, 0.5f}) {
        cvt_kernel<<<1, 64>>>(d_in, d_b32, d_b16, d_back, scale);
        std::vector<uint32_t> b32(64 * 6), b16(64 * 6); std::vector<float> back(64 * 32);
        hipMemcpy(b32.data(), d_b32, b32.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(b16.data(), d_b16, b16.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(back.data(), d_back, back.size() * 4, hipMemcpyDeviceToHost);
        int bad_bits = 0, bad_16 = 0, bad_back = 0, div = 0, mul = 0;
        for (int l = 0; l < 64; ++l) {
            int codes[32]; uint32_t want[6];
            for (int j = 0; j < 32; ++j) codes[j] = (l * 5 + j * 11 + 3) & 63;
            pack6(want, codes);
            for (int i = 0; i < 6; ++i) { bad_bits += want[i] != b32[l * 6 + i]; bad_16 += b16[l * 6 + i] != b32[l * 6 + i]; }
            for (int j = 0; j < 32; ++j) {
                const float v = in[l * 32 + j], r = back[l * 32 + j];
                bad_back += r != v;
                if (fabsf(v) >= 0.5f && fabsf(v) <= 3.0f) { div += r == v / scale; mul += r == v * scale; }
            }
        }
        if (scale != 0.5f) {
            for (int l = 0; l < 2; ++l) {
                printf("  lane %d in   :", l); for (int j = 0; j < 32; ++j) printf(" %g", in[l * 32 + j]); printf("\n");
                printf("  lane %d back :", l); for (int j = 0; j < 32; ++j) printf(" %g", back[l * 32 + j]); printf("\n");
                printf("  lane %d f32->fp6 codes (element j at bit 6j) as values:", l);
                for (int j = 0; j < 32; ++j) { const int bit = 6 * j; uint64_t w = b32[l * 6 + (bit >> 5)]; if ((bit >> 5) < 5) w |= (uint64_t)b32[l * 6 + (bit >> 5) + 1] << 32; printf(" %g", fp6_val((int)((w >> (bit & 31)) & 63))); } printf("\n");
                printf("  lane %d f16->fp6 codes as values:", l);
                for (int j = 0; j < 32; ++j) { const int bit = 6 * j; uint64_t w = b16[l * 6 + (bit >> 5)]; if ((bit >> 5) < 5) w |= (uint64_t)b16[l * 6 + (bit >> 5) + 1] << 32; printf(" %g", fp6_val((int)((w >> (bit & 31)) & 63))); } printf("\n");
            }
        }
        printf("cvt scale %.1f: f32 bits vs element-j-at-bit-6j packing: %d words differ; f16 vs f32 conversion: %d words differ; "
               "decode(scale 1) == input: %d differ; decode == input / scale: %d, == input * scale: %d (of mid-range values)\n",
               scale, bad_bits, bad_16, bad_back, div, mul);
    }
    // ---- 2. + 3. the MFMA: A[r][k], B[k][c] exact small values; hypotheses on the lane map
    std::vector<int> ca(32 * 64), cb(64 * 32);
    for (int r = 0; r < 32; ++r) for (int k = 0; k < 64; ++k) ca[r * 64 + k] = (r * 7 + k * 13 + 1) & 63;
    for (int k = 0; k < 64; ++k) for (int c = 0; c < 32; ++c) cb[k * 32 + c] = (k * 3 + c * 17 + 5) & 63;
    std::vector<uint32_t> A(64 * 6), B(64 * 6);
    for (int l = 0; l < 64; ++l) {
        int codes[32];
        for (int j = 0; j < 32; ++j) codes[j] = ca[(l & 31) * 64 + 32 * (l >> 5) + j];
        pack6(&A[l * 6], codes);
        for (int j = 0; j < 32; ++j) codes[j] = cb[(32 * (l >> 5) + j) * 32 + (l & 31)];
        pack6(&B[l * 6], codes);
    }
    uint32_t *dA, *dB; int *dsa, *dsb; float* dout;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dout, 64 * 16 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 4; ++variant) {
        // scales: variant 0: all 127 (1.0); 1: A's scale depends on the lane; 2: B's; 3: both, in byte 1 of the register (op_sel 1)
        std::vector<int> sa(64), sb(64);
        for (int l = 0; l < 64; ++l) {
            const int ea = (variant == 1 || variant == 3) ? 127 + ((l * 3) % 5) - 2 : 127;
            const int eb = (variant == 2 || variant == 3) ? 127 + ((l * 7) % 3) - 1 : 127;
            sa[l] = variant == 3 ? (ea << 8) | 0x11 : ea;
            sb[l] = variant == 3 ? (eb << 8) | 0x22 : eb;
        }
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        if (variant == 3) mfma_kernel<1, 1><<<1, 64>>>(dA, dB, dsa, dsb, dout);
        else mfma_kernel<0, 0><<<1, 64>>>(dA, dB, dsa, dsb, dout);
        std::vector<float> out(64 * 16);
        hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0; double worst = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
                double s = 0;
                for (int k = 0; k < 64; ++k) {
                    // hypothesis: the scale of A's (row, k-block kb) sits in lane row + 32 kb; B's (k-block, col) in lane col + 32 kb
                    const int kb = k >> 5;
                    const int ea = (variant == 3 ? sa[row + 32 * kb] >> 8 : sa[row + 32 * kb]) & 255;
                    const int eb = (variant == 3 ? sb[col + 32 * kb] >> 8 : sb[col + 32 * kb]) & 255;
                    s += (double)fp6_val(ca[row * 64 + k]) * fp6_val(cb[k * 32 + col]) * ldexp(1.0, ea - 127) * ldexp(1.0, eb - 127);
                }
                const double e = fabs(s - out[l * 16 + i]);
                if (e > 1e-3) ++bad;
                if (e > worst) worst = e;
            }
        printf("mfma variant %d: %d of 1024 outputs differ from the hypothesis (worst %.3g)  %s\n", variant, bad, worst, bad ? "FAIL" : "PASS");
    }
    return 0;
}
