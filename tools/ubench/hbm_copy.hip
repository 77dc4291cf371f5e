// Micro-benchmark (development tool): practical HBM ceiling for a 1:1 read/write stream on MI355X, the traffic
// shape of the synthesis kernels (every byte read once, every byte written once).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_copy tools/ubench/hbm_copy.hip && /tmp/hbm_copy
#include <hip/hip_runtime.h>

#include <cstdio>

template <typename T>
__global__ void copy_k(const T *__restrict__ in, T *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// each wavefront streams its own contiguous 4 KiB frames (like the per-chain walkers): frame = 256 float4
__global__ void copy_frames(const float4 *__restrict__ in, float4 *__restrict__ out, size_t frames, int per_wave) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (int f = 0; f < per_wave; ++f) {
        const size_t fr = wave * per_wave + f;
        if (fr >= frames) return;
        const float4 *s = in + fr * 256;
        float4 *d = out + fr * 256;
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = s[lane + 64 * q];
#pragma unroll
        for (int q = 0; q < 4; ++q) d[lane + 64 * q] = v[q];
    }
}
// the same, with each wavefront's run of frames starting at a staggered offset (tests channel hot-spotting when all
// wavefronts advance in lockstep through addresses that differ by a large power of two)
__global__ void copy_frames_staggered(const float4 *__restrict__ in, float4 *__restrict__ out, size_t frames, int per_wave,
                                      int stagger) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t start = (wave * per_wave + (wave * 5 % stagger)) % frames;
    for (int f = 0; f < per_wave; ++f) {
        const size_t fr = (start + f) % frames;
        const float4 *s = in + fr * 256;
        float4 *d = out + fr * 256;
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = s[lane + 64 * q];
#pragma unroll
        for (int q = 0; q < 4; ++q) d[lane + 64 * q] = v[q];
    }
}
// frame-major layout: frame f of chain c at (f * chains + c); wavefront (c, seg) walks `per_wave` frames of its chain,
// so wavefronts that run side by side touch neighbouring 4 KiB frames
__global__ void copy_frames_fm(const float4 *__restrict__ in, float4 *__restrict__ out, int chains, int per_wave) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t c = wave % chains, seg = wave / chains;
    for (int f = 0; f < per_wave; ++f) {
        const size_t fr = (seg * per_wave + f) * chains + c;
        const float4 *s = in + fr * 256;
        float4 *d = out + fr * 256;
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = s[lane + 64 * q];
#pragma unroll
        for (int q = 0; q < 4; ++q) d[lane + 64 * q] = v[q];
    }
}
__global__ void read_k(const float4 *__restrict__ in, float *out, size_t n) {
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void write_k(float4 *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <typename F>
float time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const size_t bytes = (size_t)512 << 20;  // 512 MiB in, 512 MiB out (config 2's footprint)
    void *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    const size_t n4 = bytes / 16;
    for (int blocks : {2048, 8192, 65536}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(copy_k<float4>, dim3(blocks), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, n4); }, 10);
        printf("copy float4 grid-stride  blocks=%6d  %.3f ms  %.2f TB/s (read+write)\n", blocks, ms, 2.0 * bytes / ms / 1e9);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(copy_k<float2>, dim3(8192), dim3(256), 0, 0, (const float2 *)a, (float2 *)b, bytes / 8); }, 10);
        printf("copy float2 grid-stride  blocks=  8192  %.3f ms  %.2f TB/s\n", ms, 2.0 * bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(copy_k<float>, dim3(8192), dim3(256), 0, 0, (const float *)a, (float *)b, bytes / 4); }, 10);
        printf("copy float  grid-stride  blocks=  8192  %.3f ms  %.2f TB/s\n", ms, 2.0 * bytes / ms / 1e9);
    }
    const size_t frames = bytes / 4096;
    for (int per_wave : {1, 8, 32, 64}) {
        const size_t waves = (frames + per_wave - 1) / per_wave;
        const unsigned blocks = (unsigned)((waves + 3) / 4);
        float ms = time_ms([&] { hipLaunchKernelGGL(copy_frames, dim3(blocks), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, frames, per_wave); }, 10);
        printf("copy 4 KiB frames, %2d consecutive frames per wavefront (%u blocks)  %.3f ms  %.2f TB/s\n", per_wave, blocks, ms, 2.0 * bytes / ms / 1e9);
    }
    for (int per_wave : {32, 64}) {  // config 2: 128 chains x 1024 frames
        const int chains = 128;
        const unsigned blocks = (unsigned)(frames / per_wave / 4);
        float msf = time_ms([&] { hipLaunchKernelGGL(copy_frames_fm, dim3(blocks), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, chains, per_wave); }, 10);
        printf("copy 4 KiB frames, frame-major [frame][128 chains], %2d frames per wavefront  %.3f ms  %.2f TB/s\n", per_wave, msf, 2.0 * bytes / msf / 1e9);
    }
    for (int stagger : {1, 7, 16, 61}) {
        const int per_wave = 64;
        const unsigned blocks = (unsigned)((frames / per_wave + 3) / 4);
        float msx = time_ms([&] { hipLaunchKernelGGL(copy_frames_staggered, dim3(blocks), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, frames, per_wave, stagger); }, 10);
        printf("copy 4 KiB frames, 64 per wavefront, start staggered mod %2d  %.3f ms  %.2f TB/s\n", stagger, msx, 2.0 * bytes / msx / 1e9);
    }
    float ms = time_ms([&] { hipLaunchKernelGGL(read_k, dim3(8192), dim3(256), 0, 0, (const float4 *)a, (float *)b, n4); }, 10);
    printf("read-only  float4  %.3f ms  %.2f TB/s\n", ms, 1.0 * bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(write_k, dim3(8192), dim3(256), 0, 0, (float4 *)b, n4); }, 10);
    printf("write-only float4  %.3f ms  %.2f TB/s\n", ms, 1.0 * bytes / ms / 1e9);
    ms = time_ms([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 10);
    printf("hipMemcpy D2D      %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e9);
    return 0;
}
