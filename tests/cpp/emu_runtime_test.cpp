// The semantics of tests/emu/emu_rt.cpp (the fiber runtime behind the CPU-emulation build), checked on small kernels:
// barriers order LDS traffic, ended work-items do not count at a barrier, cross-lane builtins see the right lanes, launches from
// several host threads serialise, and 1024-work-item workgroups fit.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <thread>
#include <vector>

static int failures = 0;
#define CHECK(c) do { if (!(c)) { ++failures; std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); } } while (0)

__global__ void k_reverse(const int *in, int *out) {  // through __shared__, one __syncthreads()
    __shared__ int tile[256];
    tile[threadIdx.x] = in[blockIdx.x * 256 + threadIdx.x];
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = tile[255 - threadIdx.x];
}

__global__ void k_early_exit(int *out) {  // wavefronts 2 and 3 return before the barrier the others wait at (twice)
    __shared__ int acc[2];
    if (threadIdx.x >= 128) return;
    if (threadIdx.x < 2) acc[threadIdx.x] = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) acc[threadIdx.x >> 6] = 100 + (int)(threadIdx.x >> 6);
    __syncthreads();
    out[threadIdx.x] = acc[0] + acc[1];
}

__global__ void k_partial_wave(int *out) {  // lanes 40..63 of the wavefront have returned: the wave-local rendezvous still completes
    if (threadIdx.x >= 40) return;
    const int v = __shfl_xor((int)threadIdx.x * 3, 1);
    out[threadIdx.x] = v;
}

__global__ void k_cross_lane(int *out, unsigned long long *ballots) {
    const int lane = (int)threadIdx.x & 63;
    out[threadIdx.x] = __builtin_amdgcn_ds_bpermute(4 * (63 - lane), (int)threadIdx.x);  // reversed within the wavefront
    const unsigned long long b = __ballot(lane % 3 == 0);
    if (lane == 0) ballots[threadIdx.x >> 6] = b;
}

__global__ void k_big(int *out) {  // 1024 work-items, a reduction tree over __shared__
    __shared__ int s[1024];
    s[threadIdx.x] = (int)threadIdx.x;
    __syncthreads();
    for (int d = 512; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = s[0];
}

int main() {
    {
        std::vector<int> in(1024), out(1024, -1);
        for (int i = 0; i < 1024; ++i) in[i] = i * 7;
        const int *pi = in.data();
        int *po = out.data();
        hipLaunchKernelGGL(k_reverse, dim3(4), dim3(256), 0, nullptr, pi, po);
        for (int b = 0; b < 4; ++b)
            for (int i = 0; i < 256; ++i) CHECK(out[b * 256 + i] == in[b * 256 + 255 - i]);
    }
    {
        std::vector<int> out(256, -1);
        int *po = out.data();
        hipLaunchKernelGGL(k_early_exit, dim3(1), dim3(256), 0, nullptr, po);
        for (int i = 0; i < 128; ++i) CHECK(out[i] == 201);
        for (int i = 128; i < 256; ++i) CHECK(out[i] == -1);
    }
    {
        std::vector<int> out(64, -1);
        int *po = out.data();
        hipLaunchKernelGGL(k_partial_wave, dim3(1), dim3(64), 0, nullptr, po);
        for (int i = 0; i < 40; ++i) CHECK(out[i] == (i ^ 1) * 3);
    }
    {
        std::vector<int> out(128, -1);
        std::vector<unsigned long long> ballots(2, 0);
        int *po = out.data();
        unsigned long long *pb = ballots.data();
        hipLaunchKernelGGL(k_cross_lane, dim3(1), dim3(128), 0, nullptr, po, pb);
        for (int i = 0; i < 128; ++i) CHECK(out[i] == (i & ~63) + 63 - (i & 63));
        unsigned long long want = 0;
        for (int l = 0; l < 64; ++l)
            if (l % 3 == 0) want |= 1ull << l;
        CHECK(ballots[0] == want && ballots[1] == want);
    }
    {
        std::vector<int> out(3, -1);
        int *po = out.data();
        hipLaunchKernelGGL(k_big, dim3(3), dim3(1024), 0, nullptr, po);
        for (int b = 0; b < 3; ++b) CHECK(out[b] == 1023 * 1024 / 2);
    }
    {   // launches from several host threads: each sees its own kernel's result (the runtime serialises them)
        std::vector<std::thread> th;
        std::vector<std::vector<int>> outs(6, std::vector<int>(256, -1));
        std::vector<int> in(256);
        for (int i = 0; i < 256; ++i) in[i] = i;
        for (int t = 0; t < 6; ++t)
            th.emplace_back([&, t]() {
                const int *pi = in.data();
                int *po = outs[t].data();
                for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k_reverse, dim3(1), dim3(256), 0, nullptr, pi, po);
            });
        for (auto &x : th) x.join();
        for (int t = 0; t < 6; ++t)
            for (int i = 0; i < 256; ++i) CHECK(outs[t][i] == 255 - i);
    }
    if (failures) {
        std::printf("%d checks failed\n", failures);
        return 1;
    }
    std::printf("all checks passed\n");
    return 0;
}
