// Developer aid (GPU box): how many divergent lane accesses per clock a CU's vector memory path sustains -- the bound the candidate
// generator runs against (profiles/EXPERIMENTS.md, round 5).  Every wave walks a chain of dependent random loads, like the trie walk.
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/ta_rate.hip -o vibrato_amd/lib/ta_rate && vibrato_amd/lib/ta_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// mode 0: one 16-byte load per step; 1: two independent chains (ILP 2); 2: 16-byte load + the 16 bytes behind it (same 64-byte line half the time,
// table entries are 32-byte aligned); 3: 4-byte loads; 4: lanes in pairs share an address; 5: all lanes of a quad share an address;
// 6: one 16-byte load + one scattered 16-byte store per step; 7: 16 lanes share a 64-byte line (lane & 3 picks the 16-byte slot)
template <int kMode>
__global__ void __launch_bounds__(64) chase(const u32x4* __restrict__ tab, u32x4* __restrict__ out, uint32_t mask, uint32_t steps, uint32_t* sink) {
    const uint32_t lane = threadIdx.x, gid = blockIdx.x * 64 + lane;
    uint32_t a = (gid * 2654435761u) & mask, b = (gid * 40503u + 977u) & mask, acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        if (kMode == 4) a = __shfl(a, lane & ~1u);
        if (kMode == 5) a = __shfl(a, lane & ~3u);
        if (kMode == 7) a = (__shfl(a, lane & ~3u) & ~3u) | (lane & 3u);
        if (kMode == 3) {
            const uint32_t v = reinterpret_cast<const uint32_t*>(tab)[a * 4];
            acc += v; a = (v + lane) & mask;
        } else {
            const u32x4 v = tab[kMode == 2 ? (a & ~1u) : a];
            acc += v.y;
            if (kMode == 1) { const u32x4 w = tab[b]; acc += w.z; b = (w.x + lane) & mask; }
            if (kMode == 2) { const u32x4 w = tab[(a & ~1u) + 1]; acc += w.z; }
            if (kMode == 6) out[(v.z + gid) & mask] = v;
            a = (v.x + lane * 7u) & mask;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int kMode>
static int run(const char* what, const u32x4* tab, u32x4* out, uint32_t entries, uint32_t waves, uint32_t steps, uint32_t* sink, double per_step) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(chase<kMode>, dim3(waves), dim3(64), 0, 0, tab, out, entries - 1, steps, sink);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(chase<kMode>, dim3(waves), dim3(64), 0, 0, tab, out, entries - 1, steps, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double lane_acc = (double)waves * 64 * steps * per_step;
    std::printf("  %-58s %8.3f ms  %7.1f G lane accesses/s  %5.2f per clock and CU (256 CUs, 2.4 GHz)\n", what, ms, lane_acc / ms * 1e-6, lane_acc / (ms * 1e-3) / (256 * 2.4e9));
    return 0;
}

int main() {
    uint32_t* sink;
    CK(hipMalloc(&sink, 4));
    for (uint32_t mb : {2u, 32u, 512u}) {
        const uint32_t entries = mb << 16;  // 16-byte entries
        std::vector<uint32_t> h((size_t)entries * 4);
        uint64_t x = 88172645463325252ull;
        for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x >> 16); }
        u32x4 *tab, *out;
        CK(hipMalloc(&tab, (size_t)entries * 16)); CK(hipMalloc(&out, (size_t)entries * 16));
        CK(hipMemcpy(tab, h.data(), (size_t)entries * 16, hipMemcpyHostToDevice));
        for (uint32_t wpc : {32u, 16u}) {  // waves per CU (one launch fills the chip once: 256 CUs)
            const uint32_t waves = 256 * wpc, steps = 2000;
            std::printf("table %u MiB, %u waves per CU, %u dependent steps per wave\n", mb, wpc, steps);
            if (run<0>("one 16-byte load per step", tab, out, entries, waves, steps, sink, 1)) return 1;
            if (run<1>("two independent chains", tab, out, entries, waves, steps, sink, 2)) return 1;
            if (run<2>("16 bytes + the 16 bytes behind them (same line)", tab, out, entries, waves, steps, sink, 2)) return 1;
            if (run<3>("4-byte loads", tab, out, entries, waves, steps, sink, 1)) return 1;
            if (run<4>("lane pairs share an address", tab, out, entries, waves, steps, sink, 1)) return 1;
            if (run<5>("quads share an address", tab, out, entries, waves, steps, sink, 1)) return 1;
            if (run<7>("quads share a 64-byte line, one 16-byte slot per lane", tab, out, entries, waves, steps, sink, 1)) return 1;
            if (run<6>("one load + one scattered 16-byte store per step", tab, out, entries, waves, steps, sink, 2)) return 1;
        }
        CK(hipFree(tab)); CK(hipFree(out));
    }
    return 0;
}
