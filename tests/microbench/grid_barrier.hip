// Cost of a device-wide barrier inside one persistent kernel on MI355X (256 workgroups, one per CU), and of the
// producer->consumer hand-off through it (a value written before the barrier by one workgroup, read after it by all).
// build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned * counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);   // agent scope by default in HIP for global memory
        while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// variant: relaxed atomics at system-coherent (sc1) level + explicit fences
__device__ __forceinline__ void grid_barrier2(unsigned * counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// variant 3: two-level counters: 8 group counters (blockIdx % 8 ~ XCD), the last arrival of a group bumps the top counter
__device__ __forceinline__ void grid_barrier3(unsigned * counters, unsigned epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned g = blockIdx.x & 7, per = (gridDim.x + 7 - g) / 8;   // workgroups in this group
        unsigned * gc = counters + 32 * (1 + g);
        const unsigned prev = __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned groups = gridDim.x < 8 ? gridDim.x : 8;
        if (prev + 1 == epoch * per) __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * groups) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
// variant 4: flags: every workgroup stores its epoch into its own slot; workgroup 0 polls all slots and publishes the epoch
__device__ __forceinline__ void grid_barrier4(unsigned * flags, unsigned epoch) {
    __syncthreads();
    if (blockIdx.x == 0) {
        if (threadIdx.x < 64) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            bool done;
            do {
                done = true;
                for (unsigned i = 1 + threadIdx.x; i < gridDim.x; i += 64)
                    if (__hip_atomic_load(flags + 32 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) done = false;
                done = __all(done);
            } while (!done);
            if (threadIdx.x == 0) __hip_atomic_store(flags, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    } else if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(flags + 32 + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int V>
__global__ void __launch_bounds__(512) barrier_loop(unsigned * counter, int iters, float * data, float * out, int check) {
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        if (check) {
            // producer: workgroup (it % grid) writes 256 floats; everyone reads them after the barrier
            if ((int) blockIdx.x == it % (int) gridDim.x && threadIdx.x < 256) data[(it & 1) * 256 + threadIdx.x] = (float) (it + 1) + threadIdx.x;
        }
        if (V == 1) grid_barrier(counter, (unsigned) (it + 1) * gridDim.x);
        else if (V == 2) grid_barrier2(counter, (unsigned) (it + 1) * gridDim.x);
        else if (V == 3) grid_barrier3(counter, (unsigned) (it + 1));
        else grid_barrier4(counter, (unsigned) (it + 1));
        if (check) {
            const float v = data[(it & 1) * 256 + (threadIdx.x & 255)];
            if (v != (float) (it + 1) + (threadIdx.x & 255)) acc += 1.f;
        }
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

int main() {
    unsigned * counter; float * data, * out;
    CK(hipMalloc(&counter, 8192)); CK(hipMalloc(&data, 4096)); CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 2; variant <= 4; variant++)
    for (int check = 0; check <= 1; check++)
    for (int grid : { 64, 256 })
    for (int threads : { 512 }) {
        const int iters = 2000;
        std::vector<float> ms;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipMemset(counter, 0, 8192)); CK(hipMemset(data, 0, 4096));
            CK(hipEventRecord(e0));
            if (variant == 2) barrier_loop<2><<<grid, threads>>>(counter, iters, data, out, check);
            else if (variant == 3) barrier_loop<3><<<grid, threads>>>(counter, iters, data, out, check);
            else barrier_loop<4><<<grid, threads>>>(counter, iters, data, out, check);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        std::vector<float> h(256); CK(hipMemcpy(h.data(), out, grid * 4, hipMemcpyDeviceToHost));
        float bad = 0; for (int i = 0; i < grid; i++) bad += h[i];
        printf("variant %d check %d grid %3d threads %3d: %.3f us per barrier (stale reads: %.0f)\n", variant, check, grid, threads, 1e3 * ms[2] / iters, bad);
    }
    return 0;
}
