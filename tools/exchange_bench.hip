// micro-benchmark + protocol prototype: all-gather inside groups of G workgroups through global memory
// (write-through sc1 stores + one flag per member, relaxed sc1 polls, sc1 payload loads, bounded spins).
// 512 workgroups of 256 threads (2 per CU), groups of 8 whose members share blockIdx % 8 (same XCD under
// the usual round-robin placement; speed only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define G 8
#define SPIN_MAX 2000000

__device__ __forceinline__ unsigned ld_flag(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int SLICE_F4>   // float4 per member slice (e.g. 256 -> 4 KB)
__global__ void __launch_bounds__(256) xbench(float* buf, unsigned* flags, unsigned* err, int iters, int ngroups, float* out, int work) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int member = blockIdx.x / ngroups, group = blockIdx.x % ngroups;    // members of a group are ngroups apart
    float* gbuf = buf + (size_t)group * G * SLICE_F4 * 4;
    unsigned* gflag = flags + group * G;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gbuf, 0, G * SLICE_F4 * 16, 0x00020000);
    float acc = 0.f;
    for (int e = 1; e <= iters; ++e) {
        // fake compute
        for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
        // publish my slice (write-through)
        for (int i = tid; i < SLICE_F4; i += blockDim.x) {
            u32x4 v = {__float_as_uint(acc + e), (unsigned)e, (unsigned)member, (unsigned)i};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (member * SLICE_F4 + i) * 16, 0, 16 /*sc1*/);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(gflag + member, (unsigned)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // wait for the other members
        if (tid < G && tid != member) {
            int spins = 0;
            while (ld_flag(gflag + tid) < (unsigned)e) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_MAX) { atomicExch(err, 1u); break; }
            }
        }
        __syncthreads();
        // gather everybody's slice into LDS (sc1 loads bypass this CU's L1)
        for (int i = tid; i < G * SLICE_F4; i += blockDim.x) {
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 16 /*sc1*/);
            if (v.y != (unsigned)e) atomicExch(err + 1, (unsigned)e);          // stale data check
            reinterpret_cast<u32x4*>(lds)[i] = v;
        }
        __syncthreads();
        acc += lds[(tid * 4) % (G * SLICE_F4 * 4)];
        // nobody may overwrite its slice before all have read: second flag round (arrival of readers)
        if (tid == 0) __hip_atomic_store(gflag + G * ngroups * 1 + member + (group * G) - (group * G) , 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the real kernel gets this ordering from its next exchange; here: a second all-to-all flag)
        if (tid == 0) __hip_atomic_store(flags + (size_t)ngroups * G + group * G + member, (unsigned)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < G && tid != member) {
            int spins = 0;
            while (ld_flag(flags + (size_t)ngroups * G + group * G + tid) < (unsigned)e) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_MAX) { atomicExch(err, 2u); break; }
            }
        }
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + tid] = acc;
}

template <int SLICE_F4>
void run(int work) {
    const int ngroups = 64, nwg = ngroups * G, iters = 2000;
    float* buf; unsigned* flags; unsigned* err; float* out;
    hipMalloc(&buf, (size_t)ngroups * G * SLICE_F4 * 16);
    hipMalloc(&flags, (size_t)2 * ngroups * G * 4 + 4096);
    hipMalloc(&err, 64); hipMalloc(&out, nwg * 256 * 4);
    hipMemset(flags, 0, (size_t)2 * ngroups * G * 4 + 4096); hipMemset(err, 0, 64);
    size_t ldsb = (size_t)G * SLICE_F4 * 16;
    hipFuncSetAttribute((const void*)xbench<SLICE_F4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(xbench<SLICE_F4>, dim3(nwg), dim3(256), ldsb, 0, buf, flags, err, iters, ngroups, out, work);
    hipEventRecord(e1, 0);
    hipError_t rc = hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned h[2]; hipMemcpy(h, err, 8, hipMemcpyDeviceToHost);
    printf("slice %5d B x %d members, work=%5d: %.2f us per (all-gather + reader-arrival) round; err=%u stale_epoch=%u rc=%d\n",
           SLICE_F4 * 16, G, work, ms * 1e3 / iters, h[0], h[1], (int)rc);
    hipFree(buf); hipFree(flags); hipFree(err); hipFree(out);
}

int main() {
    run<64>(0);      // 1 KB slices
    run<256>(0);     // 4 KB slices (16 samples x 64 columns)
    run<256>(2000);  // with ~8k cycles of fake compute per round
    run<512>(0);     // 8 KB
    return 0;
}
