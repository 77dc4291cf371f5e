// The batcher's device-side pipeline without the batcher: N groups of B bytes each way, per group
//     gather (kernel reading page-locked host memory, stream s_in) -> event -> a stand-in for the synthesis kernel (an HBM copy of the
//     same bytes, stream s_k) -> event -> scatter (kernel writing page-locked host memory, stream s_out)
// enqueued all at once, timed as a whole: what the link gives both ways when the three stages of consecutive groups overlap, against
// the group size, the copy kernels' grid cap, the stand-in's presence, busy host threads (the callers' parse copies: memcpy traffic on
// the same host memory) and the copy engines in place of the copy kernels.  Why: behind the trait the batcher sustains 28 + 28 GB/s
// where tools/ubench/pcie_duplex.hip reaches 46 + 46 with one long kernel per direction (profiles/r06t_copy_timeline.txt).
//     hipcc --offload-arch=gfx950 -O3 tools/ubench/pcie_pipeline.hip -o pcie_pipeline -lpthread && ./pcie_pipeline
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

__global__ __launch_bounds__(256) void copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16, size_t piece16) {
    const size_t pieces = (n16 + piece16 - 1) / piece16;
    for (size_t p = blockIdx.x; p < pieces; p += gridDim.x) {
        const size_t base = p * piece16;
        for (size_t o = 0; o < piece16; o += 1024) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t i = base + o + threadIdx.x + 256 * k;
                if (o + threadIdx.x + 256 * k < piece16 && i < n16) v[k] = src[i];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t i = base + o + threadIdx.x + 256 * k;
                if (o + threadIdx.x + 256 * k < piece16 && i < n16) dst[i] = v[k];
            }
        }
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t total = (size_t)512 << 20;  // each way per measurement
    void *h_in, *h_out, *d_a, *d_b;
    CK(hipHostMalloc(&h_in, total, hipHostMallocDefault));
    CK(hipHostMalloc(&h_out, total, hipHostMallocDefault));
    CK(hipMalloc(&d_a, total));
    CK(hipMalloc(&d_b, total));
    std::memset(h_in, 1, total);
    std::memset(h_out, 2, total);
    hipStream_t s_in, s_k, s_out;
    CK(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(2 * 1024);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

    // host threads copying between two private buffers in 8 KiB pieces (a parser writing packets into slots)
    std::atomic<bool> stop{false};
    std::atomic<int> busy{0};
    std::vector<std::thread> hosts;
    auto start_hosts = [&](int n) {
        stop = false;
        for (int t = 0; t < n; ++t)
            hosts.emplace_back([&] {
                const size_t sz = (size_t)64 << 20;
                char *a = (char *)std::malloc(sz), *b = (char *)std::malloc(sz);
                std::memset(a, 1, sz);
                std::memset(b, 2, sz);
                busy += 1;
                size_t o = 0;
                while (!stop) {
                    std::memcpy(b + o, a + o, 8192);
                    o = (o + 8192) % sz;
                }
                std::free(a);
                std::free(b);
            });
        while (busy < n) std::this_thread::yield();
    };
    auto stop_hosts = [&] {
        stop = true;
        for (auto &t : hosts) t.join();
        hosts.clear();
        busy = 0;
    };

    // mode: 0 = copy kernels, 1 = copy engines (one hipMemcpyAsync per group and direction), 2 = engines in 2 MiB calls,
    //       3 = hipMemcpyBatchAsync of 2 MiB copies (one call per group and direction: a submission's plane per entry)
    auto run = [&](const char *name, size_t group_bytes, unsigned cap, bool synth, int mode) {
        const size_t n_groups = total / group_bytes;
        auto body = [&] {
            for (size_t g = 0; g < n_groups; ++g) {
                const size_t o = g * group_bytes, n16 = group_bytes / 16;
                const size_t pieces = n16 / 1024;
                const unsigned grid = (unsigned)std::min<size_t>(cap, pieces);
                if (mode == 0) hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, s_in, (const uint4 *)((char *)h_in + o), (uint4 *)((char *)d_a + o), n16, (size_t)1024);
                else if (mode == 1) CK(hipMemcpyAsync((char *)d_a + o, (char *)h_in + o, group_bytes, hipMemcpyHostToDevice, s_in));
                else if (mode == 3) {
                    std::vector<void *> dsts, srcs;
                    std::vector<size_t> sizes;
                    for (size_t q = 0; q < group_bytes; q += (size_t)2 << 20) {
                        dsts.push_back((char *)d_a + o + q);
                        srcs.push_back((char *)h_in + o + q);
                        sizes.push_back(std::min<size_t>((size_t)2 << 20, group_bytes - q));
                    }
                    size_t fail = 0;
                    CK(hipMemcpyBatchAsync(dsts.data(), srcs.data(), sizes.data(), dsts.size(), nullptr, nullptr, 0, &fail, s_in));
                } else for (size_t q = 0; q < group_bytes; q += (size_t)2 << 20) CK(hipMemcpyAsync((char *)d_a + o + q, (char *)h_in + o + q, std::min<size_t>((size_t)2 << 20, group_bytes - q), hipMemcpyHostToDevice, s_in));
                CK(hipEventRecord(ev[2 * g], s_in));
                CK(hipStreamWaitEvent(s_k, ev[2 * g], 0));
                if (synth) hipLaunchKernelGGL(copy_kernel, dim3(512), dim3(256), 0, s_k, (const uint4 *)((char *)d_a + o), (uint4 *)((char *)d_b + o), n16, (size_t)1024);
                CK(hipEventRecord(ev[2 * g + 1], s_k));
                CK(hipStreamWaitEvent(s_out, ev[2 * g + 1], 0));
                if (mode == 0) hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, s_out, (const uint4 *)((char *)d_b + o), (uint4 *)((char *)h_out + o), n16, (size_t)1024);
                else if (mode == 1) CK(hipMemcpyAsync((char *)h_out + o, (char *)d_b + o, group_bytes, hipMemcpyDeviceToHost, s_out));
                else if (mode == 3) {
                    std::vector<void *> dsts, srcs;
                    std::vector<size_t> sizes;
                    for (size_t q = 0; q < group_bytes; q += (size_t)2 << 20) {
                        dsts.push_back((char *)h_out + o + q);
                        srcs.push_back((char *)d_b + o + q);
                        sizes.push_back(std::min<size_t>((size_t)2 << 20, group_bytes - q));
                    }
                    size_t fail = 0;
                    CK(hipMemcpyBatchAsync(dsts.data(), srcs.data(), sizes.data(), dsts.size(), nullptr, nullptr, 0, &fail, s_out));
                } else for (size_t q = 0; q < group_bytes; q += (size_t)2 << 20) CK(hipMemcpyAsync((char *)h_out + o + q, (char *)d_b + o + q, std::min<size_t>((size_t)2 << 20, group_bytes - q), hipMemcpyDeviceToHost, s_out));
            }
        };
        body();
        CK(hipDeviceSynchronize());
        double best = 1e9, sum = 0, enq = 0;
        const int reps = 4;
        for (int r = 0; r < reps; ++r) {
            const double t0 = now();
            body();
            enq += now() - t0;
            CK(hipDeviceSynchronize());
            const double dt = now() - t0;
            best = std::min(best, dt);
            sum += dt;
        }
        std::printf("%-86s %4zu groups  %8.3f ms  %6.2f GB/s each way (best %6.2f)  enqueue %.0f us per group\n", name, n_groups, sum / reps * 1e3, total / (sum / reps) / 1e9, total / best / 1e9, enq / reps / n_groups * 1e6);
        std::fflush(stdout);
    };
    char name[160];
    for (int hosts_n : {0, 16}) {
        if (hosts_n) start_hosts(hosts_n);
        std::printf("---- %d busy host threads (memcpy, 8 KiB pieces)\n", hosts_n);
        for (size_t mb : {2, 6, 16, 64}) {
            for (unsigned cap : {64u, 256u}) {
                std::snprintf(name, sizeof name, "copy kernels, groups of %zu MiB, grid cap %u, with the HBM stand-in", mb, cap);
                run(name, mb << 20, cap, true, 0);
            }
            std::snprintf(name, sizeof name, "copy kernels, groups of %zu MiB, grid cap 256, NO kernel between gather and scatter", mb);
            run(name, mb << 20, 256, false, 0);
            std::snprintf(name, sizeof name, "copy ENGINES, groups of %zu MiB (one call per group and direction), with the HBM stand-in", mb);
            run(name, mb << 20, 256, true, 1);
            if (mb > 2) {
                std::snprintf(name, sizeof name, "copy ENGINES in 2 MiB calls, groups of %zu MiB, with the HBM stand-in", mb);
                run(name, mb << 20, 256, true, 2);
                std::snprintf(name, sizeof name, "hipMemcpyBatchAsync of 2 MiB entries, groups of %zu MiB, with the HBM stand-in", mb);
                run(name, mb << 20, 256, true, 3);
            }
        }
        if (hosts_n) stop_hosts();
    }
    return 0;
}
