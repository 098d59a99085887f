// What happens to a launch whose x dimension holds 2^32 work-items or more?  
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_grid.hip -o tools/ubench_grid.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_count(unsigned long long* n) { if (threadIdx.x == 0) atomicAdd(n, 1ull); }
int main() {
    unsigned long long* d; unsigned long long h;
    hipMalloc(&d, 8);
    const unsigned grids[] = {1u << 20, (1u << 24) - 1, 1u << 24, (1u << 24) + 5, 1u << 25};
    for (unsigned g : grids) {
        hipMemset(d, 0, 8);
        hipLaunchKernelGGL(k_count, dim3(g), dim3(256), 0, 0, d);
        hipError_t e1 = hipGetLastError();
        hipError_t e2 = hipDeviceSynchronize();
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("grid.x = %10u x 256 threads = %11llu work-items: launch %s, sync %s, workgroups that ran: %llu\n", g, (unsigned long long)g * 256,
               hipGetErrorName(e1), hipGetErrorName(e2), h);
    }
    // the same number of workgroups folded into two dimensions
    hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k_count, dim3(1u << 21, 16), dim3(256), 0, 0, d);
    hipDeviceSynchronize();
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("grid = (2^21, 16) x 256 threads: workgroups that ran: %llu\n", h);
    return 0;
}
