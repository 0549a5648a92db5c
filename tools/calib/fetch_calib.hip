// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU for the access patterns of the tokenizer's kernels
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern"): each kernel moves a
// known number of bytes over a 1 GiB buffer (past the 256 MiB Infinity Cache); tools/profile_round.sh runs it under two
// --pmc passes and tools/summarize_profile.py turns counter / bytes into per-pattern factors.
//   read4 / read8 / read16 : coalesced streaming reads, 4 / 8 / 16 bytes per lane (sweep records: 8, per-char records: 16)
//   gather2                : one random 2-byte read per lane from a 459 MiB window (the connection matrix)
//   gather16               : one random aligned 16-byte read per lane (double-array nodes)
//   write8 / write16 / write24 : coalesced streaming writes (candidate records, per-char records, token records)
// Standalone (hipcc, no torch): tools/calib/build.sh builds it, it is not part of the product library.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <typename T>
__global__ void read_stream(const T* __restrict__ p, size_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = p[i];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= w[k];
    }
    if (acc == 0x12345678u) *sink = acc;
}
template <typename T>
__global__ void write_stream(T* __restrict__ p, size_t n) {
    T v;
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) w[k] = threadIdx.x + k;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
struct Rec24 { uint32_t w[6]; };
template <typename T>
__global__ void gather(const T* __restrict__ p, size_t window, size_t per_thread, uint32_t* sink) {
    uint64_t x = 0x9E3779B97F4A7C15ull * ((size_t)blockIdx.x * blockDim.x + threadIdx.x + 1);
    uint32_t acc = 0;
    for (size_t k = 0; k < per_thread; ++k) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        const T v = p[(x * 0x2545F4914F6CDD1Dull) % window];
        acc ^= *reinterpret_cast<const uint16_t*>(&v);
    }
    if (acc == 0x1234u) *sink = acc;
}

int main() {
    const size_t bytes = 1ull << 30;
    char* buf; uint32_t* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes));
    const dim3 grid(256 * 32), block(256);
    const size_t threads = (size_t)grid.x * block.x, per = 64;
    hipLaunchKernelGGL(read_stream<uint32_t>, grid, block, 0, 0, (const uint32_t*)buf, bytes / 4, sink);
    hipLaunchKernelGGL(read_stream<uint2>, grid, block, 0, 0, (const uint2*)buf, bytes / 8, sink);
    hipLaunchKernelGGL(read_stream<uint4>, grid, block, 0, 0, (const uint4*)buf, bytes / 16, sink);
    hipLaunchKernelGGL(gather<uint16_t>, grid, block, 0, 0, (const uint16_t*)buf, (size_t)480905776 / 2, per, sink);
    hipLaunchKernelGGL(gather<uint4>, grid, block, 0, 0, (const uint4*)buf, bytes / 16, per, sink);
    hipLaunchKernelGGL(write_stream<uint2>, grid, block, 0, 0, (uint2*)buf, bytes / 8);
    hipLaunchKernelGGL(write_stream<uint4>, grid, block, 0, 0, (uint4*)buf, bytes / 16);
    hipLaunchKernelGGL(write_stream<Rec24>, grid, block, 0, 0, (Rec24*)buf, bytes / 24);
    CK(hipDeviceSynchronize());
    // known bytes per kernel, in launch order (gathers: useful bytes; a 2-byte gather pulls a 64-byte line from HBM when it misses)
    std::printf("{\"read4\": %zu, \"read8\": %zu, \"read16\": %zu, \"gather2\": %zu, \"gather16\": %zu, \"write8\": %zu, \"write16\": %zu, \"write24\": %zu}\n",
                bytes, bytes, bytes, threads * per * 2, threads * per * 16, bytes, bytes, bytes / 24 * 24);
    return 0;
}
