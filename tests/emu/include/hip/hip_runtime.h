// TEST-ONLY stand-in for <hip/hip_runtime.h>: lets g++ compile the product's .hip sources
// unchanged and run every workgroup on CPU threads, so kernel index math / host logic can be
// checked against the oracle in the GPU-less container (tests/test_emu_*.py).
//
// It is NOT part of the product: libsymaccel.so is only ever built by hipcc for gfx950
// (symphonia_amd/build.py) and has no CPU path.  Nothing under symphonia_amd/ includes this.
//
// Model: the work-items of a workgroup are fibers of the launching host thread (emu_rt.cpp), workgroups
// run one after another, __syncthreads() is a real barrier, `__shared__` is function-scope static
// storage, and the cross-lane builtins the kernels use are emulated through a per-workgroup exchange buffer.
#pragma once

#include <pthread.h>
#include <sched.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define SYMACCEL_EMULATED_HIP 1

// ---- qualifiers -------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

// ---- vector types -----------------------------------------------------------------------
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) short4 { short x, y, z, w; };
struct alignas(4) uchar4 { unsigned char x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- runtime API subset -----------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct emu_stream *hipStream_t;
typedef struct emu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulated hip error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) {
    *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind,
                                          hipStream_t) {
    for (size_t r = 0; r < height; ++r) memmove((char *)d + r * dpitch, (const char *)s + r * spitch, width);
    return hipSuccess;
}
enum { hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostRegisterDefault = 0 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void *) { return hipSuccess; }

// ---- execution model --------------------------------------------------------------------
namespace emu {
struct Idx { unsigned x, y, z; };
extern thread_local Idx t_threadIdx, t_blockIdx;
extern Idx g_blockDim, g_gridDim;
void wave_barrier();       // rendezvous of the calling work-item's wavefront (the work-items of it that have not returned)
void workgroup_barrier();  //                ... of its workgroup
extern uint32_t g_exchange[1024];
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
}  // namespace emu

#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)
static const int warpSize = 64;

static inline void __syncthreads() { emu::workgroup_barrier(); }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })

// ---- device builtins used by the kernels ------------------------------------------------
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
#define __builtin_amdgcn_fence(order, ...) __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
static inline void __builtin_amdgcn_s_sleep(int) { sched_yield(); }
// A real rendezvous of the wavefront's work-items: the hardware runs them in lock step, the emulation
// runs them one after the other, so every point where the kernel relies on lock step must synchronise.
static inline void __builtin_amdgcn_wave_barrier() { emu::wave_barrier(); }
static inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }
// LDS-DMA load (global_load_lds): lane L's `size` bytes land at the wave-uniform LDS base + offset + L * size.  Done at
// the point of issue here; on the GPU it lands some time before the kernel's vmcnt wait -- a kernel that still reads the
// old bytes after issuing the load is wrong on both.
static inline void __builtin_amdgcn_global_load_lds(const void *g, void *l, unsigned size, int offset, int /*aux*/) {
    const unsigned lane = emu::t_threadIdx.x & 63u;
    std::memcpy(static_cast<char *>(l) + offset + (size_t)lane * size, g, size);
}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_s_waitcnt(int) {}
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  // only used on values every lane already agrees on
static inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }

// dst[lane] = src[(byte_addr / 4) % 64] within the lane's wavefront
static inline int __builtin_amdgcn_ds_bpermute(int byte_addr, int value) {
    unsigned tid = emu::t_threadIdx.x;
    unsigned wave_base = tid & ~63u;
    emu::g_exchange[tid] = (uint32_t)value;
    __builtin_amdgcn_wave_barrier();
    int r = (int)emu::g_exchange[wave_base + (((unsigned)byte_addr >> 2) & 63u)];
    __builtin_amdgcn_wave_barrier();
    return r;
}
// mask of the wavefront's lanes whose predicate is non-zero
static inline unsigned long long __ballot(int pred) {
    unsigned tid = emu::t_threadIdx.x;
    unsigned wave_base = tid & ~63u;
    emu::g_exchange[tid] = pred ? 1u : 0u;
    __builtin_amdgcn_wave_barrier();
    unsigned long long m = 0;
    for (unsigned l = 0; l < 64 && wave_base + l < emu::g_blockDim.x * emu::g_blockDim.y * emu::g_blockDim.z; ++l)
        if (emu::g_exchange[wave_base + l]) m |= 1ull << l;
    __builtin_amdgcn_wave_barrier();
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0ull; }
// every ACTIVE lane's predicate is non-zero (a wavefront that is only partly populated is still "all")
static inline int __all(int pred) { return __ballot(!pred) == 0ull; }
// v_mul_i32_i24: the low 32 bits of the product of the operands' low 24 bits, each taken as a signed number
static inline int __mul24(int a, int b) {
    const int64_t sa = (int64_t)((int32_t)((uint32_t)a << 8) >> 8), sb = (int64_t)((int32_t)((uint32_t)b << 8) >> 8);
    return (int)(uint32_t)(uint64_t)(sa * sb);
}
static inline int __shfl(int v, int src_lane, int width = 64) {
    (void)width;
    return __builtin_amdgcn_ds_bpermute(src_lane * 4, v);
}
static inline float __shfl(float v, int src_lane, int width = 64) {
    return __int_as_float(__shfl(__float_as_int(v), src_lane, width));
}
static inline int __shfl_xor(int v, int mask, int width = 64) {
    return __shfl(v, (int)((emu::t_threadIdx.x & 63u) ^ (unsigned)mask), width);
}
static inline float __shfl_xor(float v, int mask, int width = 64) {
    return __int_as_float(__shfl_xor(__float_as_int(v), mask, width));
}
static inline int __shfl_up(int v, unsigned d, int width = 64) {
    (void)width;
    unsigned lane = emu::t_threadIdx.x & 63u;
    int r = __shfl(v, (int)(lane >= d ? lane - d : lane));
    return r;
}
static inline int __shfl_down(int v, unsigned d, int width = 64) {
    (void)width;
    unsigned lane = emu::t_threadIdx.x & 63u;
    return __shfl(v, (int)(lane + d < 64 ? lane + d : lane));
}
static inline float __shfl_up(float v, unsigned d, int w = 64) { return __int_as_float(__shfl_up(__float_as_int(v), d, w)); }
static inline float __shfl_down(float v, unsigned d, int w = 64) { return __int_as_float(__shfl_down(__float_as_int(v), d, w)); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned __brev(unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i);
    return r;
}
