// Divergent-gather throughput of a CU on gfx950: the ceiling of the L2-gather BVH walk.
// Every lane chases its own pseudo-random chain through a table of NODE-byte records (the walk's node fetch:
// next index depends on the loaded data), WPS waves per SIMD, one workgroup of 256 per slot, every CU busy.
// Reported: wave-level gather instructions per CU per kilo-cycle, lane-records per CU per cycle, and the
// dependent-step latency seen by one wave (cycles per step at 1 wave per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/gather_rate tools/ubench/gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int STEPS = 4096;

// NODE = bytes fetched per step (16: one dwordx4; 32: two, same 32-byte record; 64: four)
// SHARE = log2 of the lanes that share a record (0: every lane its own chain; 6: wave-uniform)
template <int NODE>
__global__ __launch_bounds__(256) void chase(const uint4* __restrict__ tab, uint32_t mask, uint32_t lanes, uint32_t share,
                                             uint32_t* out, unsigned long long* cyc) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = ((blockIdx.x * 256u + threadIdx.x) >> share) * 2654435761u;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (lane < lanes) {
        for (int i = 0; i < STEPS; i++) {
            const uint4* p = tab + (size_t)(idx & mask) * (NODE / 16);
            uint4 a = p[0];
            uint32_t nx = a.x;
            acc += a.y;
            if (NODE >= 32) { uint4 b = p[1]; nx ^= b.x; acc += b.w; }
            if (NODE >= 64) { uint4 c = p[2], d = p[3]; nx ^= c.x ^ d.x; acc += c.w + d.w; }
            idx = nx + acc * 0u;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 0x12345u) out[0] = acc + idx;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (idx == 0xdeadbeefu) out[1] = idx;
}

template <int NODE>
static int run(const char* name, size_t table_bytes, int wps, uint32_t lanes, uint32_t share) {
    int dev = 0; hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, dev));
    const int cus = pr.multiProcessorCount;
    const size_t n = table_bytes / NODE;               // power of two
    std::vector<uint32_t> h(table_bytes / 4);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 16); }
    uint4* d; uint32_t* out; unsigned long long* cyc;
    CHK(hipMalloc(&d, table_bytes)); CHK(hipMalloc(&out, 64)); CHK(hipMalloc(&cyc, 8));
    CHK(hipMemcpy(d, h.data(), table_bytes, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int grid = cus * wps;                        // wps workgroups of 4 waves per CU = wps waves per SIMD
    chase<NODE><<<grid, 256>>>(d, (uint32_t)(n - 1), lanes, share, out, cyc);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    chase<NODE><<<grid, 256>>>(d, (uint32_t)(n - 1), lanes, share, out, cyc);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hc; CHK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
    const double clk = pr.clockRate * 1e3;             // Hz
    const double cu_cycles = ms * 1e-3 * clk;
    const double wave_steps_per_cu = (double)wps * 4 * STEPS;
    const double instr_per_step = NODE / 16;
    printf("%-34s table %6zu KiB  %d w/SIMD  %2u lanes  share %u : %7.1f cyc per wave-step per CU, %5.2f lane-records/cyc/CU, "
           "%6.1f B/cyc/CU useful, %6.1f cyc per gather instr per CU;  wave0 %6.1f cyc/step\n",
           name, table_bytes >> 10, wps, lanes, 1u << share, cu_cycles / wave_steps_per_cu,
           wave_steps_per_cu * lanes / cu_cycles, wave_steps_per_cu * lanes * NODE / cu_cycles,
           cu_cycles / (wave_steps_per_cu * instr_per_step), (double)hc / STEPS * (clk / 1e8));
    hipFree(d); hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, 0));
    printf("%s, %d CUs, %.0f MHz (s_memtime ticks at 100 MHz)\n", pr.name, pr.multiProcessorCount, pr.clockRate / 1e3);
    const size_t sizes[] = {16u << 10, 2u << 20, 32u << 20};
    for (size_t tb : sizes) {
        for (int wps : {1, 5, 8}) {
            run<16>("16-B record (1 x dwordx4)", tb, wps, 64, 0);
            run<32>("32-B record (2 x dwordx4)", tb, wps, 64, 0);
            run<64>("64-B record (4 x dwordx4)", tb, wps, 64, 0);
        }
    }
    // the walk's own shape: 41 of 64 lanes, 5 waves per SIMD, 2 MiB of nodes
    run<32>("32-B, 41 lanes", 2u << 20, 5, 41, 0);
    run<16>("16-B, 41 lanes", 2u << 20, 5, 41, 0);
    run<64>("64-B, 41 lanes", 2u << 20, 5, 41, 0);
    // coherence: groups of lanes on the same record
    for (uint32_t sh : {1u, 2u, 3u, 6u}) run<32>("32-B, shared records", 2u << 20, 5, 64, sh);
    return 0;
}
