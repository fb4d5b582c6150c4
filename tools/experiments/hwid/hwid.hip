// Which workgroups share a CU (and a SIMD), and what tells the two partners apart?  Launch shape of the 64-sample chain
// kernels: 256 threads, 76 KB of dynamic LDS (two workgroups per CU), 2048 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void __launch_bounds__(256, 2) probe(unsigned* out, int spin) {
    extern __shared__ unsigned char lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
    volatile unsigned char* p = lds;
    unsigned acc = 0;
    for (int i = 0; i < spin; ++i) { p[threadIdx.x] = (unsigned char)i; acc += p[(threadIdx.x + 1) & 255]; __builtin_amdgcn_s_sleep(8); }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        out[(blockIdx.x * 4 + w) * 3 + 0] = hw;
        out[(blockIdx.x * 4 + w) * 3 + 1] = xcc;
        out[(blockIdx.x * 4 + w) * 3 + 2] = (unsigned)(wall_clock64() & 0xFFFFFFFFu) + (acc & 0);
    }
}
int main() {
    const int nb = 2048;
    unsigned* d; hipMalloc(&d, nb * 4 * 3 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 76 * 1024, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 12);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] tg_id[19:16] vm_id[23:20] queue_id[26:24] state_id[29:27] me_id[31:30]
    for (int b = 0; b < 24; ++b) {
        printf("block %4d:", b);
        for (int w = 0; w < 4; ++w) {
            unsigned hw = h[(b * 4 + w) * 3], x = h[(b * 4 + w) * 3 + 1] & 0xF;
            printf("  [xcc %u se %u sh %u cu %2u simd %u slot %u]", x, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
        }
        printf("\n");
    }
    // first-round residents (blocks < 512): group by (xcc, se, sh, cu)
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < 512; ++b) {
        unsigned hw = h[(b * 4) * 3], x = h[(b * 4) * 3 + 1] & 0xF;
        cu[(x << 16) | (hw & 0xFF00)].push_back(b);
    }
    int shown = 0; std::map<int, int> hist, dslot;
    for (auto& kv : cu) {
        hist[(int)kv.second.size()]++;
        if (kv.second.size() == 2) {
            int a = kv.second[0], b2 = kv.second[1];
            dslot[((h[(a * 4) * 3] & 15) << 4) | (h[(b2 * 4) * 3] & 15)]++;
            if (shown++ < 12) printf("CU %06x: blocks %d %d (diff %d) slots %u %u\n", kv.first, a, b2, b2 - a, h[(a * 4) * 3] & 15, h[(b2 * 4) * 3] & 15);
        }
    }
    for (auto& kv : hist) printf("%d CUs hold %d of the first 512 blocks\n", kv.second, kv.first);
    for (auto& kv : dslot) printf("slot pair (%d, %d): %d CUs\n", kv.first >> 4, kv.first & 15, kv.second);
    return 0;
}
