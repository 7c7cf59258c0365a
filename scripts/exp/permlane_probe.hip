// what v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950) do to two registers: prints lane -> (a', b') for a = lane, b = 100 + lane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
    out[l] = r[0]; out[64 + l] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
    out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int part = 0; part < 4; ++part) {
        printf("%s %s:", part < 2 ? "permlane16_swap" : "permlane32_swap", part % 2 ? "b'" : "a'");
        for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[part * 64 + l]);
        printf("\n");
    }
    return 0;
}
