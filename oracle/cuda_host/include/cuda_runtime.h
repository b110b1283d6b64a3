// TEST INFRASTRUCTURE ONLY (oracle) -- "CUDA on the host": lets g++ compile the REFERENCE's own
// gaussian-pointcloud-rasterization/cuda_rasterizer/*.cu and rasterize_points.cu, where they lie under
// /root/reference, into oracle/_ref/ (recipe: oracle/build_ref.py).  Nothing here is product code and nothing
// here restates the reference: it only supplies the CUDA runtime / cooperative-groups / CUB names those files use,
// on top of the single-threaded fibre engine the CPU test-suite already owns (tests/hipemu/hip/hip_runtime.h):
// one fibre per GPU thread, blocks one after another in blockIdx order, threads of a block in thread_rank order
// between barriers, barriers count the threads that have not returned (like the hardware's).
//
// Consequences for the reference's racy statements (SURVEY §2.2 defect 3) are spelled out in oracle/build_ref.py.
#pragma once
#include <math.h>      // libstdc++'s wrapper: float overloads of exp / sqrt / ceil in the global namespace, like CUDA's
#include <stdint.h>
#include <stdio.h>
#include <stdexcept>
#include "../../../tests/hipemu/hip/hip_runtime.h"

// CUDA's mixed-signedness integer overloads (the reference calls min(unsigned, int), auxiliary.h:45-55)
static inline unsigned min(unsigned a, int b) { return a < (unsigned)b ? a : (unsigned)b; }
static inline unsigned min(int a, unsigned b) { return (unsigned)a < b ? (unsigned)a : b; }
static inline unsigned max(unsigned a, int b) { return a > (unsigned)b ? a : (unsigned)b; }
static inline unsigned max(int a, unsigned b) { return (unsigned)a > b ? (unsigned)a : b; }

typedef int cudaError_t;
#define cudaSuccess 0
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3 };
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline const char* cudaGetErrorString(cudaError_t) { return "cuda-on-host"; }
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaMemset(void* p, int v, size_t n) { memset(p, v, n); return cudaSuccess; }
static inline void __trap() { fprintf(stderr, "cuda-on-host: __trap()\n"); abort(); }

// kernel<<<grid, block>>>(args...) is not C++: build_ref.py rewrites each launch to CUEMU_LAUNCH((kernel), grid, block)(args...)
namespace cuemu {
template <typename K> struct Launcher {
    K k; dim3 g, b;
    template <typename... A> void operator()(A... args) const { hipemu::launch_now(k, g, b, args...); }
};
template <typename K> static inline Launcher<K> launcher(K k, dim3 g, dim3 b) { return Launcher<K>{k, g, b}; }
}  // namespace cuemu
#define CUEMU_LAUNCH(k, g, b) cuemu::launcher(k, dim3(g), dim3(b))
