// Calibration kernels for the roofline ceilings that bench.py reports for k_trace.  Development aid, not
// part of the library.  Prints one JSON object:
//   valu_wave_instr_per_s   wave-level VALU instructions per second of a pure v_fma_f32 stream at full
//                           occupancy (the ceiling for SQ_INSTS_VALU / time of any kernel)
//   gather16_lane_loads_per_s_{l1,l2,mall}
//                           scattered 16-byte loads per second, all 64 lanes active, 4 independent loads in
//                           flight per lane, from tables of 16 KiB (L1), 1 MiB (L2) and 64 MiB (Infinity Cache)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define FMA1(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c))
__global__ __launch_bounds__(256) void k_valu_fma(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float const m = 1.0000001f, c = 1e-7f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { FMA1(a0); FMA1(a1); FMA1(a2); FMA1(a3); FMA1(a4); FMA1(a5); FMA1(a6); FMA1(a7); }
    }
    float const s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678f) out[0] = s;
}

// The instruction mix of a traversal step: compares writing lane masks, selects, min/max, subtract, multiply.
#define MIX(x, y) asm volatile("v_sub_f32 %0, %0, %2\n v_mul_f32 %1, %1, %3\n v_min_f32 %0, %0, %1\n v_max_f32 %1, %0, %1\n" \
                               "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_and_b32 %1, 0x7fffffff, %1\n v_add_f32 %0, %0, %1" \
                               : "+v"(x), "+v"(y) : "v"(m), "v"(c) : "vcc")
__global__ __launch_bounds__(256) void k_valu_mix(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float const m = 1.0000001f, c = 1e-7f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 2; ++k) { MIX(a0, a1); MIX(a2, a3); MIX(a4, a5); MIX(a6, a7); }
    }
    float const s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void k_gather16(const float4* __restrict__ table, uint32_t mask, int iters, float* out) {
    uint32_t idx = (blockIdx.x * 977u + threadIdx.x * 7919u) & mask;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        uint32_t const i0 = idx, i1 = (idx * 3u + 17u) & mask, i2 = (idx * 5u + 101u) & mask, i3 = (idx * 7u + 1001u) & mask;
        float4 const a = table[i0], b = table[i1], c = table[i2], d = table[i3];
        acc += a.x + b.x + c.x + d.x;
        idx = (idx * 1664525u + 1013904223u) & mask;
    }
    if (acc == 12345.678f) out[0] = acc;
}

static double timeMs(void (*launch)(void*), void* arg) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(arg);                                   // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(arg);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

struct ValuArg { float* out; int iters, blocks; };
struct GatherArg { const float4* table; uint32_t mask; int iters, blocks; float* out; };

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int const cus = prop.multiProcessorCount;
    float* out;
    hipMalloc(&out, 4);
    ValuArg va{out, 4000, cus * 8};
    double const msV = timeMs([](void* p) { auto* a = static_cast<ValuArg*>(p); hipLaunchKernelGGL(k_valu_fma, dim3(a->blocks), dim3(256), 0, 0, a->out, a->iters); }, &va);
    double const valuInstr = double(va.blocks) * 4 * va.iters * 64.0;       // 64 v_fma per loop iteration and wave
    double const msM = timeMs([](void* p) { auto* a = static_cast<ValuArg*>(p); hipLaunchKernelGGL(k_valu_mix, dim3(a->blocks), dim3(256), 0, 0, a->out, a->iters); }, &va);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d,\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    printf(" \"valu_ms\": %.4f, \"valu_wave_instr\": %.0f, \"valu_wave_instr_per_s\": %.4e,\n", msV, valuInstr, valuInstr / (msV * 1e-3));
    printf(" \"valu_mix_ms\": %.4f, \"valu_mix_wave_instr_per_s\": %.4e", msM, valuInstr / (msM * 1e-3));
    const char* names[3] = {"l1", "l2", "mall"};
    size_t const sizes[3] = {size_t(16) << 10, size_t(1) << 20, size_t(64) << 20};
    for (int k = 0; k < 3; ++k) {
        size_t const n = sizes[k] / 16;
        std::vector<float> h(n * 4, 0.0f);
        float4* t;
        hipMalloc(&t, sizes[k]);
        hipMemcpy(t, h.data(), sizes[k], hipMemcpyHostToDevice);
        GatherArg ga{t, uint32_t(n - 1), 2000, cus * 8, out};
        double const ms = timeMs([](void* p) { auto* a = static_cast<GatherArg*>(p); hipLaunchKernelGGL(k_gather16, dim3(a->blocks), dim3(256), 0, 0, a->table, a->mask, a->iters, a->out); }, &ga);
        double const laneLoads = double(ga.blocks) * 256 * ga.iters * 4.0;
        printf(",\n \"gather16_%s_ms\": %.4f, \"gather16_lane_loads_per_s_%s\": %.4e", names[k], ms, names[k], laneLoads / (ms * 1e-3));
        hipFree(t);
    }
    printf("\n}\n");
    return 0;
}
