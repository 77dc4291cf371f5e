// Runtime half of the test-only HIP stand-in (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>

namespace emu {
thread_local Idx t_threadIdx, t_blockIdx;
Idx g_blockDim, g_gridDim;
pthread_barrier_t g_barrier;
pthread_barrier_t g_wave_barrier[32];
uint32_t g_exchange[1024];

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    unsigned nthreads = block.x * block.y * block.z;
    unsigned nblocks = grid.x * grid.y * grid.z;
    if (nthreads == 0 || nblocks == 0) return;
    g_blockDim = Idx{block.x, block.y, block.z};
    g_gridDim = Idx{grid.x, grid.y, grid.z};
    pthread_barrier_init(&g_barrier, nullptr, nthreads);
    const unsigned nwaves = (nthreads + 63) / 64;
    for (unsigned w = 0; w < nwaves; w++) {
        unsigned cnt = (w + 1) * 64 <= nthreads ? 64 : nthreads - w * 64;
        pthread_barrier_init(&g_wave_barrier[w], nullptr, cnt);
    }
    std::vector<std::thread> pool;
    pool.reserve(nthreads);
    for (unsigned t = 0; t < nthreads; t++) {
        pool.emplace_back([=, &body]() {
            t_threadIdx = Idx{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
            for (unsigned b = 0; b < nblocks; b++) {
                t_blockIdx = Idx{b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y)};
                body();
                pthread_barrier_wait(&g_barrier);  // a workgroup's __shared__ storage is reused by the next
            }
        });
    }
    for (auto &th : pool) th.join();
    pthread_barrier_destroy(&g_barrier);
    for (unsigned w = 0; w < nwaves; w++) pthread_barrier_destroy(&g_wave_barrier[w]);
}
}  // namespace emu
