// Dev microbenchmark: latency of one data-carrying hand-off between two workgroups (ping-pong of a 1-KiB tile whose fp16 sign bits are the
// validity flags, as csrc/latent_persist.hip exchanges activations), by placement and store flavour:
//   same XCD / different XCD   x   plain stores (stay in the XCD's L2) / sc1 stores (write-through)   ; loads always sc1 (L1 bypass).
// 256 workgroups x 160 KB LDS = one per CU; every workgroup reads its XCC id; two chosen ones play, the rest exit.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

__global__ __launch_bounds__(256) void census(unsigned* xcc) { if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id(); }

template <bool SC1_STORE>
__global__ __launch_bounds__(256) void pingpong(char* buf, int a, int b, int iters, unsigned long long* ticks, unsigned* bad) {
    extern __shared__ char lds[];
    if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
    if (threadIdx.x >= 64) return;
    const bool is_a = (int)blockIdx.x == a;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1 << 20, 0x00020000);
    const int lane = threadIdx.x;
    // two tiles: ping (a -> b) at 0, pong (b -> a) at 4096; value = iteration (positive fp16-ish pattern with sign clear), poison = 0xffffffff
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 1; it <= iters; ++it) {
        const unsigned val = (unsigned)it & 0x7fff7fffu;
        const u32x4 v = {val, val, val, val};
        const u32x4 p = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        const int mine = is_a ? 0 : 4096, theirs = is_a ? 4096 : 0;
        const int par = (it & 1) * 8192;              // alternate two slots so a slot can be re-poisoned by its reader... no: by its writer, one round later
        // order (as in latent_persist.hip): poison my tile of the other slot BEFORE this round's data store, so that whoever has
        // seen this round's data also finds the next round's slot poisoned
        auto poison_other = [&]() {
            if (SC1_STORE) __builtin_amdgcn_raw_buffer_store_b128(p, rs, (par ^ 8192) + mine + lane * 16, 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b128(p, rs, (par ^ 8192) + mine + lane * 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        auto send = [&]() {
            if (SC1_STORE) __builtin_amdgcn_raw_buffer_store_b128(v, rs, par + mine + lane * 16, 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b128(v, rs, par + mine + lane * 16, 0, 0);
        };
        if (is_a) { poison_other(); send(); }
        for (unsigned spin = 0;; ++spin) {
            asm volatile("" ::: "memory");
            const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(rs, par + theirs + lane * 16, 0, 16);
            const bool poison = ((g.x | g.y | g.z | g.w) & 0x80008000u) != 0u;
            if (__builtin_amdgcn_ballot_w64(poison) == 0ull) { if (g.x != val) atomicAdd(bad, 1u); break; }
            if (spin > 4000000u) { atomicAdd(bad, 1000000u); return; }
        }
        if (!is_a) { poison_other(); send(); }
    }
    if (is_a && lane == 0) *ticks = __builtin_amdgcn_s_memrealtime() - t0;
}

int main() {
    const int lds = 160 * 1024 - 64;
    hipFuncSetAttribute((const void*)pingpong<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void*)pingpong<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    unsigned* dx; hipMalloc(&dx, 256 * 4);
    hipLaunchKernelGGL(census, dim3(256), dim3(256), 0, 0, dx);
    std::vector<unsigned> x(256); hipMemcpy(x.data(), dx, 1024, hipMemcpyDeviceToHost);
    printf("xcc of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %u", x[i]); printf("\n");
    int hist[16] = {}; for (unsigned v : x) hist[v & 15]++;
    printf("blocks per xcc:"); for (int i = 0; i < 8; ++i) printf(" %d", hist[i]); printf("\n");
    char* buf; hipMalloc(&buf, 1 << 20);
    unsigned long long* dt; hipMalloc(&dt, 8);
    unsigned* dbad; hipMalloc(&dbad, 4);
    const int iters = 2000;
    // the census launch used tiny blocks; placement of the 160-KB-LDS launch is assumed round-robin the same way (blocks b, b+8 share an XCD): verified in-kernel? (speed only)
    for (int rep = 0; rep < 2; ++rep)
        for (int same = 1; same >= 0; --same)
            for (int sc1 = 0; sc1 <= 1; ++sc1) {
                if (!same && !sc1) continue;                       // plain stores are not visible across XCDs: would time out
                const int a = 0, b = same ? 8 : 1;
                hipMemset(buf, 0xff, 1 << 20); hipMemset(dbad, 0, 4); hipMemset(dt, 0, 8);
                if (sc1) hipLaunchKernelGGL(pingpong<true>, dim3(256), dim3(256), lds, 0, buf, a, b, iters, dt, dbad);
                else hipLaunchKernelGGL(pingpong<false>, dim3(256), dim3(256), lds, 0, buf, a, b, iters, dt, dbad);
                hipDeviceSynchronize();
                unsigned long long t; unsigned bad; hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost); hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
                printf("%s XCD (blocks %d,%d: xcc %u,%u)  %s stores: %.3f us per one-way hand-off  (errors %u)\n", same ? "same" : "other", a, b, x[a], x[b],
                       sc1 ? "sc1  " : "plain", t / 100.0 / iters / 2.0, bad);
            }
    return 0;
}
