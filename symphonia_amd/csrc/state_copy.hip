// Write-back of the per-chain state (delay lines, overlap, V FIFO, flags) from the launch's scratch copy to the
// caller's in/out buffers: up to three small device-to-device copies in ONE kernel launch.  A plain kernel in the
// launch stream starts a few microseconds after the synthesis kernel ends; the hipMemcpyAsync blits used before
// cost ~50 us of dependency latency per step (profiles/r01_* timelines).
#include <hip/hip_runtime.h>

#include "symaccel_internal.h"

namespace symaccel {

namespace {

struct CopyJob {
    uint32_t *dst[3];
    const uint32_t *src[3];
    size_t words[3];
};

__global__ __launch_bounds__(256) void state_copy_kernel(CopyJob job) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const size_t n = job.words[k];
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
            job.dst[k][i] = job.src[k][i];
    }
}

}  // namespace

// bytes must be multiples of 4 (all state arrays are f32 / i32).
int launch_state_copy(symaccel_ctx *ctx, void *dst0, const void *src0, size_t bytes0, void *dst1, const void *src1,
                      size_t bytes1, void *dst2, const void *src2, size_t bytes2) {
    CopyJob job{{(uint32_t *)dst0, (uint32_t *)dst1, (uint32_t *)dst2},
                {(const uint32_t *)src0, (const uint32_t *)src1, (const uint32_t *)src2},
                {bytes0 / 4, bytes1 / 4, bytes2 / 4}};
    size_t most = job.words[0];
    if (job.words[1] > most) most = job.words[1];
    if (job.words[2] > most) most = job.words[2];
    if (most == 0) return SYMACCEL_OK;
    size_t blocks = (most + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(state_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, job);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
