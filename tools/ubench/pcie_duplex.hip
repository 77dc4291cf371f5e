// PCIe both ways at once, by mechanism: a kernel that reads page-locked host memory (gather) / writes it (scatter) against the copy
// engines (hipMemcpyAsync), alone and against each other.  Why: the cross-stream batcher (csrc/batcher.cpp) moves every chunk with one
// gather and one scatter KERNEL; the trait-level harness tops out near 30 + 30 GB/s where symaccel_aac_synth_pipelined's engine copies
// reach 44 + 44.     hipcc --offload-arch=gfx950 -O3 tools/ubench/pcie_duplex.hip -o pcie_duplex && ./pcie_duplex
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

template <int PER_LANE>
__global__ __launch_bounds__(256) void copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16, size_t piece16) {
    // a workgroup takes pieces of piece16 x 16 bytes (the batcher: 1024 = 16 KiB), PER_LANE loads in flight per lane
    const size_t pieces = (n16 + piece16 - 1) / piece16;
    for (size_t p = blockIdx.x; p < pieces; p += gridDim.x) {
        const size_t base = p * piece16;
        for (size_t o = 0; o < piece16; o += 256 * PER_LANE) {
            uint4 v[PER_LANE];
#pragma unroll
            for (int k = 0; k < PER_LANE; ++k) {
                const size_t i = base + o + threadIdx.x + 256 * k;
                if (o + threadIdx.x + 256 * k < piece16 && i < n16) v[k] = src[i];
            }
#pragma unroll
            for (int k = 0; k < PER_LANE; ++k) {
                const size_t i = base + o + threadIdx.x + 256 * k;
                if (o + threadIdx.x + 256 * k < piece16 && i < n16) dst[i] = v[k];
            }
        }
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t bytes = (size_t)256 << 20, n16 = bytes / 16;
    void *h_in, *h_out, *d_a, *d_b;
    CK(hipHostMalloc(&h_in, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&h_out, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d_a, bytes));
    CK(hipMalloc(&d_b, bytes));
    std::memset(h_in, 1, bytes);
    std::memset(h_out, 2, bytes);
    CK(hipMemset(d_a, 3, bytes));
    CK(hipMemset(d_b, 4, bytes));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    auto gather = [&](unsigned grid, size_t piece16) { hipLaunchKernelGGL(copy_kernel<4>, dim3(grid), dim3(256), 0, s1, (const uint4 *)h_in, (uint4 *)d_a, n16, piece16); };
    auto scatter = [&](unsigned grid, size_t piece16) { hipLaunchKernelGGL(copy_kernel<4>, dim3(grid), dim3(256), 0, s2, (const uint4 *)d_b, (uint4 *)h_out, n16, piece16); };
    auto time_it = [&](const char *name, int reps, auto &&body) {
        body();
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int r = 0; r < reps; ++r) body();
        CK(hipDeviceSynchronize());
        const double dt = (now() - t0) / reps;
        std::printf("%-64s %8.3f ms  %7.2f GB/s per direction used\n", name, dt * 1e3, bytes / dt / 1e9);
    };
    const int reps = 8;
    for (unsigned grid : {64u, 256u, 1024u, 16384u}) {
        char name[128];
        std::snprintf(name, sizeof name, "kernel gather alone, %u workgroups, 16 KiB pieces", grid);
        time_it(name, reps, [&] { gather(grid, 1024); });
        std::snprintf(name, sizeof name, "kernel scatter alone, %u workgroups", grid);
        time_it(name, reps, [&] { scatter(grid, 1024); });
        std::snprintf(name, sizeof name, "kernel gather || kernel scatter, %u workgroups each", grid);
        time_it(name, reps, [&] { gather(grid, 1024); scatter(grid, 1024); });
    }
    time_it("engine H2D alone (hipMemcpyAsync, 256 MiB)", reps, [&] { CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s1)); });
    time_it("engine D2H alone", reps, [&] { CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s2)); });
    time_it("engine H2D || engine D2H", reps, [&] {
        CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s1));
        CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s2));
    });
    time_it("kernel gather (256 wg) || engine D2H", reps, [&] {
        gather(256, 1024);
        CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s2));
    });
    time_it("engine H2D || kernel scatter (256 wg)", reps, [&] {
        CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s1));
        scatter(256, 1024);
    });
    // engine copies in 2 MiB calls (a submission's plane): what a per-submission hipMemcpyAsync would cost the link
    time_it("engine H2D || engine D2H in 2 MiB calls", reps, [&] {
        for (size_t o = 0; o < bytes; o += (size_t)2 << 20) {
            CK(hipMemcpyAsync((char *)d_a + o, (char *)h_in + o, (size_t)2 << 20, hipMemcpyHostToDevice, s1));
            CK(hipMemcpyAsync((char *)h_out + o, (char *)d_b + o, (size_t)2 << 20, hipMemcpyDeviceToHost, s2));
        }
    });
    {
        const double t0 = now();
        for (int r = 0; r < 1000; ++r) CK(hipMemcpyAsync((char *)d_a, (char *)h_in, 4096, hipMemcpyHostToDevice, s1));
        const double api = (now() - t0) / 1000;
        CK(hipDeviceSynchronize());
        std::printf("hipMemcpyAsync call (4 KiB, one thread, nothing else running): %.2f us of host time\n", api * 1e6);
    }
    return 0;
}
