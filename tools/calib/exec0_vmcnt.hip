// Does a vector memory load issued with EXEC = 0 take part in vmcnt on gfx950?  (lattice_sentence keeps the number of gathers in
// flight static by issuing a fixed number of loads per pass; whether a pass without work may issue them with EXEC = 0 -- no
// register write, nothing to drain -- depends on the answer.)
//
// One wave: a real load that misses to HBM (a fresh line of a 1 GiB buffer per trial), then K loads under EXEC = 0, then
// `s_waitcnt vmcnt(K)`, then the destination register is copied out immediately.  If the masked loads count, vmcnt(K) holds
// until the real load has landed and the copy sees the loaded value in every trial; if they do not, the wait falls through
// (1 <= K outstanding) and the copy sees the stale register.  A second kernel times 1000 masked loads to show what they cost.
//   build + run (on the GPU box): hipcc --offload-arch=gfx950 -O2 -o exec0_vmcnt tools/calib/exec0_vmcnt.hip && ./exec0_vmcnt
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const uint32_t* buf, uint32_t bytes, uint32_t stride, uint32_t trials, uint32_t* stale, uint32_t* good) {
    const uint64_t b = (uint64_t)reinterpret_cast<uintptr_t>(buf);
    u32x4 rs;
    rs.x = (uint32_t)b; rs.y = (uint32_t)(b >> 32) & 0xFFFFu; rs.z = bytes; rs.w = 0x00020000u;
    uint32_t n_stale = 0, n_good = 0;
    for (uint32_t t = 0; t < trials; ++t) {
        const uint32_t off = (t * stride + threadIdx.x * 4u) % (bytes - 256u);
        uint32_t v = 0xDEADBEEFu, d0 = 0, d1 = 0, d2 = 0, d3 = 0, seen;
        asm volatile(
            "s_waitcnt vmcnt(0)\n\t"
            "buffer_load_dword %[v], %[a], %[rs], 0 offen\n\t"
            "s_mov_b64 exec, 0\n\t"
            "buffer_load_dword %[d0], %[a], %[rs], 0 offen\n\t"
            "buffer_load_dword %[d1], %[a], %[rs], 0 offen\n\t"
            "buffer_load_dword %[d2], %[a], %[rs], 0 offen\n\t"
            "buffer_load_dword %[d3], %[a], %[rs], 0 offen\n\t"
            "s_mov_b64 exec, -1\n\t"
            "s_waitcnt vmcnt(4)\n\t"
            "v_mov_b32 %[seen], %[v]\n\t"
            "s_waitcnt vmcnt(0)"
            : [v] "+&v"(v), [d0] "+&v"(d0), [d1] "+&v"(d1), [d2] "+&v"(d2), [d3] "+&v"(d3), [seen] "=&v"(seen)
            : [a] "v"(off), [rs] "s"(rs));
        // buf[i] = i ^ 0x5A5A5A5A: the value the real load delivers is known
        const uint32_t want = (off >> 2) ^ 0x5A5A5A5Au;
        n_stale += seen == 0xDEADBEEFu ? 1u : 0u;
        n_good += seen == want ? 1u : 0u;
        if (d0 | d1 | d2 | d3) n_stale += 1u << 20;  // a masked load wrote its register: cannot happen
    }
    atomicAdd(stale, n_stale);
    atomicAdd(good, n_good);
}

__global__ void masked_cost(const uint32_t* buf, uint32_t bytes, unsigned long long* cycles, int masked) {
    const uint64_t b = (uint64_t)reinterpret_cast<uintptr_t>(buf);
    u32x4 rs;
    rs.x = (uint32_t)b; rs.y = (uint32_t)(b >> 32) & 0xFFFFu; rs.z = bytes; rs.w = 0x00020000u;
    uint32_t d = 0, off = threadIdx.x * 4u;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < 250; ++i) {
        if (masked)
            asm volatile("s_mov_b64 exec, 0\n\t"
                         "buffer_load_dword %[d], %[a], %[rs], 0 offen\n\tbuffer_load_dword %[d], %[a], %[rs], 0 offen\n\t"
                         "buffer_load_dword %[d], %[a], %[rs], 0 offen\n\tbuffer_load_dword %[d], %[a], %[rs], 0 offen\n\t"
                         "s_mov_b64 exec, -1" : [d] "+&v"(d) : [a] "v"(off), [rs] "s"(rs));
        else
            asm volatile("s_mov_b64 exec, 1\n\t"
                         "buffer_load_dword %[d], %[a], %[rs], 0 offen\n\tbuffer_load_dword %[d], %[a], %[rs], 0 offen\n\t"
                         "buffer_load_dword %[d], %[a], %[rs], 0 offen\n\tbuffer_load_dword %[d], %[a], %[rs], 0 offen\n\t"
                         "s_mov_b64 exec, -1" : [d] "+&v"(d) : [a] "v"(off), [rs] "s"(rs));
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(d));
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) *cycles = t1 - t0 + (d == 0x7FFFFFFFu ? 1 : 0);
}

int main() {
    const uint32_t bytes = 1u << 30;
    uint32_t* buf;
    CK(hipMalloc(reinterpret_cast<void**>(&buf), bytes));
    {
        uint32_t* h = static_cast<uint32_t*>(std::malloc(bytes));
        for (uint32_t i = 0; i < bytes / 4; ++i) h[i] = i ^ 0x5A5A5A5Au;
        CK(hipMemcpy(buf, h, bytes, hipMemcpyHostToDevice));
        std::free(h);
    }
    uint32_t* cnt;
    CK(hipMalloc(reinterpret_cast<void**>(&cnt), 16));
    CK(hipMemset(cnt, 0, 16));
    const uint32_t trials = 2000;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, nullptr, buf, bytes, 1000003u * 64u, trials, cnt, cnt + 1);
    CK(hipDeviceSynchronize());
    uint32_t h[2];
    CK(hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost));
    std::printf("exec0_vmcnt: %u lane-trials stale, %u good of %u -> loads under EXEC=0 %s in vmcnt\n", h[0], h[1], trials * 64,
                h[0] == 0 && h[1] == trials * 64 ? "COUNT" : "DO NOT COUNT (or something else is off)");
    unsigned long long* cyc;
    CK(hipMalloc(reinterpret_cast<void**>(&cyc), 16));
    for (int masked = 1; masked >= 0; --masked) {
        hipLaunchKernelGGL(masked_cost, dim3(1), dim3(64), 0, nullptr, buf, bytes, cyc, masked);
        CK(hipDeviceSynchronize());
        unsigned long long c;
        CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        std::printf("  1000 loads with %s: %llu cycles (%.1f per load)\n", masked ? "EXEC=0" : "one lane active (same line)", c, c / 1000.0);
    }
    return 0;
}
