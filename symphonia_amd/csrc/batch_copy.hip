// Gather / scatter between page-locked HOST memory and HBM for the cross-stream batcher (csrc/batcher.cpp): ONE launch moves
// every piece of a chunk -- the submissions of many streams sit in separate page-locked slots, and a hipMemcpyAsync per plane and
// submission (five per submission) costs more host time than the bytes cost link time.  hipHostMalloc memory is mapped into the
// device's address space, so the kernel reads (gather) or writes (scatter) the slots directly; the descriptor list itself is read
// from page-locked memory too.  A workgroup takes one piece of at most 16 KiB: four 16-byte loads per lane in flight, then the
// stores -- with a grid of thousands of pieces the link sees enough outstanding reads to run at its rate.
#include <hip/hip_runtime.h>

#include "symaccel_internal.h"

namespace symaccel {

namespace {

__global__ __launch_bounds__(256) void batch_copy_kernel(const BatchCopyDesc *__restrict__ descs) {
    const BatchCopyDesc d = descs[blockIdx.x];
    const unsigned tid = threadIdx.x;
    const uintptr_t s = reinterpret_cast<uintptr_t>(d.src), t = reinterpret_cast<uintptr_t>(d.dst);
    if (((s | t | d.bytes) & 15u) == 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(d.src);
        uint4 *dst = reinterpret_cast<uint4 *>(d.dst);
        const unsigned n = d.bytes / 16u;  // <= 1024
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tid + 256u * k < n) v[k] = src[tid + 256u * k];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tid + 256u * k < n) dst[tid + 256u * k] = v[k];
    } else if (((s | t | d.bytes) & 3u) == 0) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(d.src);
        uint32_t *dst = reinterpret_cast<uint32_t *>(d.dst);
        for (unsigned i = tid; i < d.bytes / 4u; i += 256u) dst[i] = src[i];
    } else {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(d.src);
        uint8_t *dst = reinterpret_cast<uint8_t *>(d.dst);
        for (unsigned i = tid; i < d.bytes; i += 256u) dst[i] = src[i];
    }
}

}  // namespace

int launch_batch_copy(symaccel_ctx *ctx, hipStream_t stream, const BatchCopyDesc *descs, size_t n) {
    if (n == 0) return SYMACCEL_OK;
    if (n > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(batch_copy_kernel, dim3((unsigned)n), dim3(256), 0, stream, descs);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
