// How many 256-thread blocks with a given dynamic LDS size fit one CU (LDS allocation granularity probe).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256, 5) k(float *o) { extern __shared__ float s[]; s[threadIdx.x] = 1.f; __syncthreads(); o[threadIdx.x] = s[255 - threadIdx.x]; }
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerMultiprocessor %zu sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", (size_t)p.sharedMemPerMultiprocessor, (size_t)p.sharedMemPerBlock, (size_t)p.maxSharedMemoryPerMultiProcessor);
    for (int b : {32768, 32704, 32640, 32512, 32256, 32000, 31744, 31232, 30720, 30336, 27304, 27136}) {
        int n = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, (size_t)b);
        printf("dyn LDS %6d B -> %d blocks per CU (err %d)\n", b, n, (int)e);
    }
    return 0;
}
