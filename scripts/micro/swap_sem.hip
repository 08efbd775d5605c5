// Prints what v_permlane16_swap / v_permlane32_swap / DPP row_ror:8 do to lane ids (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned x = threadIdx.x, y = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
    o[256 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, true);
}
int main() {
    unsigned* o; hipMallocManaged(&o, 320 * 4);
    k<<<1, 64>>>(o); hipDeviceSynchronize();
    const char* nm[5] = {"swap16 vdst", "swap16 src ", "swap32 vdst", "swap32 src ", "row_ror:8  "};
    for (int j = 0; j < 5; ++j) { printf("%s:", nm[j]); for (int i = 0; i < 64; ++i) printf(" %u", o[j * 64 + i]); printf("\n"); }
    return 0;
}
