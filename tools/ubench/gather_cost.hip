// Microbenchmark: cost of scattered 16-byte loads per wave instruction as a function of the number of
// active lanes and of the table size (L1 / L2 resident).  Development aid, not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

__global__ void k_gather(const float4* __restrict__ table, uint32_t mask_words, uint32_t active, int iters, float* out) {
    uint32_t lane = threadIdx.x & 63;
    uint32_t idx = (blockIdx.x * 977u + threadIdx.x * 7919u) & mask_words;
    float acc = 0.f;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            float4 v = table[idx];
            acc += v.x + v.w;
            idx = (idx * 1664525u + 1013904223u + __float_as_uint(v.y)) & mask_words;   // dependent address
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// independent loads (4 in flight per lane)
__global__ void k_gather4(const float4* __restrict__ table, uint32_t mask_words, uint32_t active, int iters, float* out) {
    uint32_t lane = threadIdx.x & 63;
    uint32_t idx = (blockIdx.x * 977u + threadIdx.x * 7919u) & mask_words;
    float acc = 0.f;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            uint32_t i0 = idx, i1 = (idx * 3u + 17u) & mask_words, i2 = (idx * 5u + 101u) & mask_words, i3 = (idx * 7u + 1001u) & mask_words;
            float4 a = table[i0], b = table[i1], c = table[i2], d = table[i3];
            acc += a.x + b.x + c.x + d.x;
            idx = (idx * 1664525u + 1013904223u) & mask_words;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// all active lanes of a wave read the same 16 bytes (a wave of coherent rays on one node)
__global__ void k_same4(const float4* __restrict__ table, uint32_t mask_words, uint32_t active, int iters, float* out) {
    uint32_t lane = threadIdx.x & 63;
    uint32_t idx = (blockIdx.x * 977u + (threadIdx.x >> 6) * 7919u) & mask_words;      // wave-uniform
    float acc = 0.f;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            uint32_t i0 = idx, i1 = (idx * 3u + 17u) & mask_words, i2 = (idx * 5u + 101u) & mask_words, i3 = (idx * 7u + 1001u) & mask_words;
            float4 a = table[i0], b = table[i1], c = table[i2], d = table[i3];
            acc += a.x + b.x + c.x + d.x;
            idx = (idx * 1664525u + 1013904223u) & mask_words;
        }
    }
    if (acc == 12345.678f) out[0] = acc + lane;
}

int main() {
    float* out; hipMalloc(&out, 4);
    for (size_t bytes : {size_t(16) << 10, size_t(1) << 20, size_t(64) << 20}) {
        size_t n = bytes / 16;
        std::vector<float> h(n * 4, 0.0f);
        float4* t; hipMalloc(&t, bytes); hipMemcpy(t, h.data(), bytes, hipMemcpyHostToDevice);
        for (int variant = 0; variant < 3; ++variant)
        for (uint32_t active : {64u, 32u, 16u, 8u, 4u}) {
            int iters = 2000, blocks = 256 * 8;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (variant == 0) hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, t, (uint32_t)(n - 1), active, iters, out);
                else if (variant == 1) hipLaunchKernelGGL(k_gather4, dim3(blocks), dim3(256), 0, 0, t, (uint32_t)(n - 1), active, iters, out);
                else hipLaunchKernelGGL(k_same4, dim3(blocks), dim3(256), 0, 0, t, (uint32_t)(n - 1), active, iters, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double waveInstr = double(blocks) * 4 * iters * (variant ? 4 : 1);
            double cyclesPerCU = ms * 1e-3 * 2.4e9;            // assumes 2.4 GHz
            double perInstrPerCU = cyclesPerCU / (waveInstr / 256.0);
            printf("table %8zu KB  %s  active %2u: %8.3f ms  %6.2f cycles per wave-load per CU  (%.2f per active lane)\n", bytes >> 10,
                   variant == 0 ? "dependent" : (variant == 1 ? "4 indep" : "4 indep, same address in all lanes"), active, ms, perInstrPerCU, perInstrPerCU / active);
        }
        hipFree(t);
    }
    return 0;
}
