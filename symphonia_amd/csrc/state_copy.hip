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

// Per-record status arrays (what a reference decoder would have answered for each block / filter): pure functions of
// the descriptors, one thread per record.
__global__ __launch_bounds__(256) void flac_status_kernel(const symaccel_flac_desc *__restrict__ desc, size_t n, unsigned blocksize,
                                                          int8_t *__restrict__ status) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const symaccel_flac_desc d = desc[i];
        int st = SYMACCEL_OK;
        if (d.kind > SYMACCEL_FLAC_LPC || d.order > blocksize) st = SYMACCEL_ERR_DECODE;                  // decoder.rs:431-433, 456-458
        else if (d.kind == SYMACCEL_FLAC_FIXED && d.order > 4) st = SYMACCEL_ERR_DECODE;                  // fixed orders are 0..4
        else if (d.kind == SYMACCEL_FLAC_LPC && (d.order < 1 || d.order > 32)) st = SYMACCEL_ERR_DECODE;  // decoder.rs:361
        else if (d.kind == SYMACCEL_FLAC_LPC && d.shift > 31) st = SYMACCEL_ERR_UNSUPPORTED;              // decoder.rs:506-508
        status[i] = (int8_t)st;
    }
}
__global__ __launch_bounds__(256) void alac_status_kernel(const symaccel_alac_desc *__restrict__ desc, size_t n, int8_t *__restrict__ status) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned mode = desc[i].mode;
        status[i] = (int8_t)((mode > 0 && mode < 15) ? SYMACCEL_ERR_DECODE : SYMACCEL_OK);  // lib.rs:167-169
    }
}
// a wavefront per channel-block's y row (<= 65 values): any value the floor-1 kernel's arithmetic does not cover?
__global__ __launch_bounds__(256) void floor1_status_kernel(const uint32_t *__restrict__ y, size_t count, unsigned n_posts,
                                                            int8_t *__restrict__ status) {
    const unsigned lane = threadIdx.x & 63u;
    for (size_t b = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6); b < count; b += (size_t)gridDim.x * 4u) {
        const uint32_t *row = y + b * n_posts;
        uint32_t worst = lane < n_posts ? row[lane] : 0u;
        if (lane == 0 && n_posts > 64) worst = worst > row[64] ? worst : row[64];
        const bool big = __any(worst > 511u) != 0;  // (up to 511 the kernel is exact: integer render_point above |dy| 511, final_y inside int16)
        if (lane == 0) status[b] = (int8_t)(big ? SYMACCEL_ERR_UNSUPPORTED : SYMACCEL_OK);
    }
}
__global__ __launch_bounds__(256) void tns_status_kernel(const symaccel_aac_tns_filter *__restrict__ f, size_t n, unsigned n_frames,
                                                         int8_t *__restrict__ status) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const symaccel_aac_tns_filter x = f[i];
        const bool ok = x.frame < n_frames && x.start < x.end && x.end <= 1024 && x.order >= 1 && x.order <= 20;
        status[i] = (int8_t)(ok ? SYMACCEL_OK : SYMACCEL_ERR_INVALID_ARG);
    }
}

}  // namespace

static unsigned status_grid(size_t n) {
    const size_t b = (n + 255) / 256;
    return (unsigned)(b < 4096 ? (b ? b : 1) : 4096);
}
int launch_flac_status(symaccel_ctx *ctx, const symaccel_flac_desc *d_desc, size_t n, size_t blocksize, int8_t *d_status) {
    hipLaunchKernelGGL(flac_status_kernel, dim3(status_grid(n)), dim3(256), 0, ctx->stream, d_desc, n, (unsigned)blocksize, d_status);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}
int launch_alac_status(symaccel_ctx *ctx, const symaccel_alac_desc *d_desc, size_t n, int8_t *d_status) {
    hipLaunchKernelGGL(alac_status_kernel, dim3(status_grid(n)), dim3(256), 0, ctx->stream, d_desc, n, d_status);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}
int launch_floor1_status(symaccel_ctx *ctx, const uint32_t *d_y, size_t count, int n_posts, int8_t *d_status) {
    const size_t b = (count + 3) / 4;
    hipLaunchKernelGGL(floor1_status_kernel, dim3((unsigned)(b < 16384 ? b : 16384)), dim3(256), 0, ctx->stream, d_y, count,
                       (unsigned)n_posts, d_status);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}
int launch_tns_status(symaccel_ctx *ctx, const symaccel_aac_tns_filter *d_filters, size_t n, size_t n_frames, int8_t *d_status) {
    hipLaunchKernelGGL(tns_status_kernel, dim3(status_grid(n)), dim3(256), 0, ctx->stream, d_filters, n,
                       (unsigned)(n_frames > 0xffffffffu ? 0xffffffffu : n_frames), d_status);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

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
