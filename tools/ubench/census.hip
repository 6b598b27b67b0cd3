// Where do the workgroups of a full-chip launch land?  Prints, for a grid of 8 blocks per CU (256 threads, 20 KiB
// of LDS each, as the traversal kernel), the XCC, shader engine and CU of every block.  Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
__global__ __launch_bounds__(256) void k_census(uint32_t* out, int spin) {
    __shared__ uint32_t pad[5120];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    uint32_t xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
    // keep the block resident for a while so that the grid really fills the chip
    long long t0 = clock64();
    while (clock64() - t0 < spin) { pad[threadIdx.x] += 1; }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc + (pad[3] & 0); }
}
int main() {
    int blocks = 256 * 8;
    uint32_t* d; hipMalloc(&d, blocks * 8);
    hipLaunchKernelGGL(k_census, dim3(blocks), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(blocks * 2);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < blocks; ++b) {
        uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
        // HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
        printf("%d xcc %u se %u sh %u cu %u simd %u\n", b, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3);
    }
    return 0;
}
