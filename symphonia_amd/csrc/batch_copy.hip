// Gather / scatter between page-locked HOST memory and HBM for the cross-stream batcher (csrc/batcher.cpp): ONE launch moves
// every piece of a chunk -- the submissions of many streams sit in separate page-locked slots, and a hipMemcpyAsync per plane and
// submission (five per submission) costs more host time than the bytes cost link time.  hipHostMalloc memory is mapped into the
// device's address space, so the kernel reads (gather) or writes (scatter) the slots directly; the descriptor list itself is read
// from page-locked memory too.  A workgroup takes one piece of at most 16 KiB: four 16-byte loads per lane in flight, then the
// stores -- with a grid of thousands of pieces the link sees enough outstanding reads to run at its rate.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "symaccel_internal.h"

namespace symaccel {

namespace {

// A capped grid walks the piece list: the link needs about a hundred KiB in flight (50 GB/s x 2 us), not the 32 MiB a grid of one
// workgroup per piece keeps resident -- such a grid fills every wave slot of the device with workgroups that wait for PCIe, and the
// scatter and the synthesis kernel of the neighbouring chunk (other streams) queue behind it instead of running beside it.
// (DIR -- 0 = gather, host to device; 1 = scatter -- changes nothing in the code: it names the two directions apart in a kernel trace)
template <bool NT, int DIR>
__global__ __launch_bounds__(256) void batch_copy_kernel(const BatchCopyDesc *__restrict__ descs, unsigned n_pieces) {
  for (unsigned piece = blockIdx.x; piece < n_pieces; piece += gridDim.x) {
    const BatchCopyDesc d = descs[piece];
    const unsigned tid = threadIdx.x;
    const uintptr_t s = reinterpret_cast<uintptr_t>(d.src), t = reinterpret_cast<uintptr_t>(d.dst);
    if (((s | t | d.bytes) & 15u) == 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(d.src);
        uint4 *dst = reinterpret_cast<uint4 *>(d.dst);
        const unsigned n = d.bytes / 16u;  // <= 1024
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tid + 256u * k < n) {
                if constexpr (NT) {
                    const unsigned *p = reinterpret_cast<const unsigned *>(src + tid + 256u * k);
                    v[k] = uint4{__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2), __builtin_nontemporal_load(p + 3)};
                } else {
                    v[k] = src[tid + 256u * k];
                }
            }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tid + 256u * k < n) {
                if constexpr (NT) {
                    unsigned *p = reinterpret_cast<unsigned *>(dst + tid + 256u * k);
                    __builtin_nontemporal_store(v[k].x, p);
                    __builtin_nontemporal_store(v[k].y, p + 1);
                    __builtin_nontemporal_store(v[k].z, p + 2);
                    __builtin_nontemporal_store(v[k].w, p + 3);
                } else {
                    dst[tid + 256u * k] = v[k];
                }
            }
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
}

// The completion flag of a launch: ONE 64-bit word in page-locked host memory, written behind the last scatter of the launch (same
// stream: the scatter's stores to host memory are complete when this kernel starts).  Waiters read the word -- no runtime call on the
// wait path, nothing for sixteen caller threads to contend on.  The value only grows (a launch sequence number per block).
__global__ void batch_flag_kernel(unsigned long long *flag, unsigned long long seq) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __atomic_store_n(flag, seq, __ATOMIC_RELAXED);
}

}  // namespace

int launch_batch_flag(symaccel_ctx *ctx, hipStream_t stream, uint64_t *h_flag, uint64_t seq) {
    hipLaunchKernelGGL(batch_flag_kernel, dim3(1), dim3(1), 0, stream, reinterpret_cast<unsigned long long *>(h_flag), (unsigned long long)seq);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

int launch_batch_copy(symaccel_ctx *ctx, hipStream_t stream, const BatchCopyDesc *descs, size_t n, bool scatter) {
    if (n == 0) return SYMACCEL_OK;
    if (n > 0x7fffffffu) return SYMACCEL_ERR_INVALID_ARG;
    // workgroups per copy launch.  64: the link needs few -- a copy kernel that fills the device keeps the synthesis kernel and the copy of
    // the other direction waiting for compute units (256 -> 64: AAC behind the trait 3.96 -> 4.3-4.45 M packets/s at S = 256, 32 and 96 are
    // both slower; profiles/r06z4_big_groups.jsonl, r06z5_copy_grid.jsonl).  Development knobs: SYMACCEL_BATCH_COPY_WGS (both directions;
    // 0 = one workgroup per piece, as round 5 had it), SYMACCEL_BATCH_COPY_WGS_G / _S (gather / scatter alone)
    static const unsigned caps[2] = {
        [] { const char *e = std::getenv("SYMACCEL_BATCH_COPY_WGS_G"); if (!e) e = std::getenv("SYMACCEL_BATCH_COPY_WGS"); return e ? (unsigned)std::atoi(e) : 64u; }(),
        [] { const char *e = std::getenv("SYMACCEL_BATCH_COPY_WGS_S"); if (!e) e = std::getenv("SYMACCEL_BATCH_COPY_WGS"); return e ? (unsigned)std::atoi(e) : 64u; }()};
    const unsigned cap = caps[scatter ? 1 : 0];
    const unsigned grid = cap ? (unsigned)std::min<size_t>(n, cap) : (unsigned)n;
    static const bool nt = [] {  // development knob: non-temporal accesses
        const char *e = std::getenv("SYMACCEL_BATCH_COPY_NT");
        return e && std::atoi(e) != 0;
    }();
    if (nt && scatter) hipLaunchKernelGGL((batch_copy_kernel<true, 1>), dim3(grid), dim3(256), 0, stream, descs, (unsigned)n);
    else if (nt) hipLaunchKernelGGL((batch_copy_kernel<true, 0>), dim3(grid), dim3(256), 0, stream, descs, (unsigned)n);
    else if (scatter) hipLaunchKernelGGL((batch_copy_kernel<false, 1>), dim3(grid), dim3(256), 0, stream, descs, (unsigned)n);
    else hipLaunchKernelGGL((batch_copy_kernel<false, 0>), dim3(grid), dim3(256), 0, stream, descs, (unsigned)n);
    SYM_GPU(ctx, hipGetLastError());
    return SYMACCEL_OK;
}

}  // namespace symaccel
