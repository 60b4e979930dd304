// Which XCD does a workgroup run on?  s_getreg_b32 HW_REG_XCC_ID (id 20, bits 3:0) per block,
// against the round-robin rule (blocks b and b + 8 share an XCD) the flow kernel uses for speed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
}
int main() {
    const int n = 512;
    int *d; hipMalloc(&d, n * 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d);
        std::vector<int> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        int same = 0; for (int b = 8; b < n; ++b) same += h[b] == h[b - 8];
        printf("rep %d: first 16 ids:", rep); for (int b = 0; b < 16; ++b) printf(" %d", h[b]);
        printf("  | b and b+8 equal for %d of %d\n", same, n - 8);
    }
    return 0;
}
