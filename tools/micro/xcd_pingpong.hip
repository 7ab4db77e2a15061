// Microbenchmark: what does one hop of the split-role kernel's data-tagged exchange cost, and which stores leave L2 on its memory side?
//
// Two workgroups play ping-pong through two buffers in device memory: A stores a fragment tagged with the round number, B polls
// until it sees that tag and answers into its own buffer, A polls for the answer.  One-way hop = (time of N rounds) / 2N.
// Varied: the store's cache policy (aux bits of buffer_store: 0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0+sc1), the polling load's policy
// (1 sc0, 16 sc1, 17 sc0+sc1), the placement (partner on the same XCD: blockIdx differing by 8; on another XCD: by 1; on the same CU
// is not controllable), the fragment size (16 B: one lane; 1 KiB: b128 from all 64 lanes), and the rollout kernel's sentinel reset
// (every payload store is accompanied by a store of all-ones words into the other parity's slot, as nocf_duo.hip does).
// Every variant is its own kernel instantiation, so `rocprofv3 --pmc WRITE_SIZE` / `--pmc TCC_EA0_WRREQ_sum TCP_TCC_WRITE_REQ_sum`
// (separate passes, counters only) attributes memory-side writes per variant by kernel name:
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/pp tools/micro/xcd_pingpong.hip && /tmp/pp
//     rocprofv3 --pmc TCC_EA0_WRREQ_sum TCP_TCC_WRITE_REQ_sum -d out -o pp --output-format csv -- /tmp/pp 2000
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// ST / LD: aux bits; BIG: 1 KiB fragments; RESET: sentinel store into the other parity's slot with every payload
template <int ST, int LD, bool BIG, bool RESET>
__global__ void __launch_bounds__(64) pp(unsigned* buf, int partner_delta, int rounds, unsigned long long* out, unsigned* xcc_out, int nap) {
    const int bid = blockIdx.x;
    if (bid != 0 && bid != partner_delta) return;
    const int me = bid == 0 ? 0 : 1;
    const int lane = threadIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (lane == 0) xcc_out[me] = xcc & 0xf;
    // buffers: [who writes 2][parity 2][64 lanes][4 words]
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 2 * 2 * 1024, 0x00020000);
    const bool act = BIG || lane == 0;
    const int vb = lane * 16;
    unsigned long long t0 = 0;
    for (int r = 1; r <= rounds + 8; ++r) {
        if (r == 9) t0 = now();
        const int par = r & 1;
        const u32x4 pay = {(unsigned)r, (unsigned)r, (unsigned)r, (unsigned)r};
        const u32x4 sen = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (me == 0) {
            if (act) {
                __builtin_amdgcn_raw_buffer_store_b128(pay, rs, vb, (0 * 2 + par) * 1024, ST);
                if (RESET) __builtin_amdgcn_raw_buffer_store_b128(sen, rs, vb, (0 * 2 + (par ^ 1)) * 1024, ST);
            }
        }
        // poll the partner's buffer (B: A's payload of this round; A: B's answer of this round)
        const int src = me == 0 ? 1 : 0;
        int spins = 0;
        while (true) {
            u32x4 v = {0, 0, 0, 0};
            if (act) v = __builtin_amdgcn_raw_buffer_load_b128(rs, vb, (src * 2 + par) * 1024, LD);
            const bool ok = !act || (v.x == (unsigned)r && v.y == (unsigned)r && v.z == (unsigned)r && v.w == (unsigned)r);
            if (__all(ok)) break;
            if (nap) __builtin_amdgcn_s_sleep(1);
            if (++spins > 4000000) { if (lane == 0) out[2] = 0xdeadULL; return; }
        }
        if (me == 1) {
            if (act) {
                __builtin_amdgcn_raw_buffer_store_b128(pay, rs, vb, (1 * 2 + par) * 1024, ST);
                if (RESET) __builtin_amdgcn_raw_buffer_store_b128(sen, rs, vb, (1 * 2 + (par ^ 1)) * 1024, ST);
            }
        }
    }
    const unsigned long long t1 = now();
    if (lane == 0) out[me] = t1 - t0;
}

// how long does a store take to be acknowledged (store; s_waitcnt vmcnt(0)), and a load behind it?
template <int ST>
__global__ void __launch_bounds__(64) ack(unsigned* buf, int rounds, unsigned long long* out) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 4096, 0x00020000);
    const int lane = threadIdx.x;
    const u32x4 pay = {1u, 2u, 3u, 4u};
    unsigned long long tot = 0;
    for (int r = 0; r < rounds + 8; ++r) {
        if (r == 8) tot = 0;
        const unsigned long long t0 = now();
        __builtin_amdgcn_raw_buffer_store_b128(pay, rs, lane * 16, (r & 3) * 1024, ST);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tot += now() - t0;
    }
    if (lane == 0) out[0] = tot;
}

static const char* pol(int a) { return a == 0 ? "plain" : a == 1 ? "sc0" : a == 2 ? "nt" : a == 16 ? "sc1" : a == 17 ? "sc0+sc1" : a == 3 ? "sc0+nt" : "?"; }

template <int ST, int LD, bool BIG, bool RESET>
static void run(unsigned* buf, unsigned long long* out, unsigned* xo, int rounds) {
    for (int place = 0; place < 2; ++place) {
        const int delta = place == 0 ? 8 : 1;
        for (int nap = 0; nap < 2; ++nap) {
            hipMemset(buf, 0, 4096); hipMemset(out, 0, 32);
            hipLaunchKernelGGL((pp<ST, LD, BIG, RESET>), dim3(16), dim3(64), 0, 0, buf, delta, rounds, out, xo, nap);
            hipDeviceSynchronize();
            unsigned long long h[4]; unsigned x[2];
            hipMemcpy(h, out, 32, hipMemcpyDeviceToHost); hipMemcpy(x, xo, 8, hipMemcpyDeviceToHost);
            printf("store %-8s load %-8s frag %-5s reset %d  partner %-9s (xcc %u/%u) nap %d : hop %7.0f cycles%s\n", pol(ST), pol(LD), BIG ? "1KiB" : "16B", (int)RESET,
                   place == 0 ? "same-XCD" : "cross-XCD", x[0], x[1], nap, (double)h[0] / (2.0 * rounds), h[2] == 0xdeadULL ? "  TIMEOUT" : "");
        }
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned* buf; unsigned long long* out; unsigned* xo;
    hipMalloc(&buf, 8192); hipMalloc(&out, 64); hipMalloc(&xo, 64);
    printf("# xcd_pingpong: %d rounds; hop = one-way producer-store -> consumer-sees-it latency in shader-clock cycles (s_memtime)\n", rounds);
    // the rollout kernel's two forms: plain store / sc1 store, sc1 polling load
    run<0, 16, true, false>(buf, out, xo, rounds);
    run<0, 16, true, true>(buf, out, xo, rounds);
    run<16, 16, true, false>(buf, out, xo, rounds);
    run<16, 16, true, true>(buf, out, xo, rounds);
    run<0, 16, false, false>(buf, out, xo, rounds);
    run<16, 16, false, false>(buf, out, xo, rounds);
    // other policies
    run<1, 16, true, false>(buf, out, xo, rounds);
    run<2, 16, true, false>(buf, out, xo, rounds);
    run<17, 16, true, false>(buf, out, xo, rounds);
    run<0, 1, true, false>(buf, out, xo, rounds);
    run<1, 1, true, false>(buf, out, xo, rounds);
    run<16, 1, true, false>(buf, out, xo, rounds);
    run<0, 17, true, false>(buf, out, xo, rounds);
    run<17, 17, true, false>(buf, out, xo, rounds);
    run<2, 1, true, true>(buf, out, xo, rounds);
    run<1, 1, true, true>(buf, out, xo, rounds);
#define ACK(ST) do { hipMemset(out, 0, 32); hipLaunchKernelGGL((ack<ST>), dim3(1), dim3(64), 0, 0, buf, rounds, out); hipDeviceSynchronize(); \
        unsigned long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); printf("store %-8s 1 KiB + s_waitcnt vmcnt(0): %7.0f cycles\n", pol(ST), (double)h / rounds); } while (0)
    ACK(0); ACK(1); ACK(2); ACK(16); ACK(17);
    return 0;
}
