// What does a grid-wide barrier cost on MI355X (8 XCDs, one L2 each, coherent with each other only through memory)?
// A persistent kernel of one workgroup per CU alternates "produce a small block of data, barrier, consume a block another
// workgroup (on another XCD) produced".  Variants of the data path / barrier:
//   0  plain stores + __threadfence() (agent-scope release: L2 write-back) + relaxed atomic arrive / spin + acquire fence (L2 invalidate)
//   1  write-through stores (relaxed agent-scope atomic stores: sc1) + s_waitcnt + relaxed atomic arrive / spin, consumer loads are
//      relaxed agent-scope atomic loads (sc1: served from memory, never from this XCD's L2) - no cache maintenance instruction at all
//   2  as 1, but the arrive counter is hierarchical: one counter per XCD (workgroup id & 7), the last arriver of an XCD bumps the global one
// Every spin loop has an iteration cap (the kernel then records a failure and all later barriers fall through), so a wrong
// protocol cannot hang the GPU.  Prints us per iteration and whether every consumer saw the right data.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gbp tools/grid_barrier_probe.hip && /tmp/gbp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kWords = 1024;          // 4 KiB per workgroup per iteration
constexpr int kSpinCap = 2000000;

struct Ctl { unsigned count[2]; unsigned xcd[2][8]; unsigned fail; };

template <int MODE>
__device__ __forceinline__ void grid_barrier(Ctl* c, unsigned nwg, unsigned it) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned phase = it & 1;
        const unsigned target = nwg * ((it >> 1) + 1);
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (MODE == 2) {
            const unsigned x = blockIdx.x & 7, per = nwg >> 3;
            const unsigned old = __hip_atomic_fetch_add(&c->xcd[phase][x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == per * ((it >> 1) + 1)) __hip_atomic_fetch_add(&c->count[phase], per, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_fetch_add(&c->count[phase], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int spins = 0;
        while (__hip_atomic_load(&c->count[phase], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > kSpinCap || __hip_atomic_load(&c->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&c->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned* data, Ctl* c, unsigned* bad, int iters, int words) {
    const unsigned nwg = gridDim.x, me = blockIdx.x;
    unsigned errors = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned* mine = data + ((size_t)(it & 1) * nwg + me) * kWords;
        for (int i = threadIdx.x; i < words; i += 256) {
            const unsigned v = (unsigned)it * 2654435761u + me * 40503u + i;
            if (MODE == 0) mine[i] = v;
            else __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        grid_barrier<MODE>(c, nwg, it);
        const unsigned src = (me + 3) % nwg;            // +3: another XCD (workgroup w runs on XCD w & 7)
        const unsigned* theirs = data + ((size_t)(it & 1) * nwg + src) * kWords;
        for (int i = threadIdx.x; i < words; i += 256) {
            const unsigned want = (unsigned)it * 2654435761u + src * 40503u + i;
            const unsigned got = MODE == 0 ? theirs[i] : __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            errors += got != want;
        }
    }
    if (errors) atomicAdd(bad, errors);
}

template <int MODE>
static void run(int nwg, int iters, int words) {
    unsigned *data, *bad; Ctl* c;
    hipMalloc(&data, (size_t)2 * nwg * kWords * 4); hipMalloc(&bad, 4); hipMalloc(&c, sizeof(Ctl));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f; unsigned hbad = 0, hfail = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(bad, 0, 4); hipMemset(c, 0, sizeof(Ctl)); hipMemset(data, 0, (size_t)2 * nwg * kWords * 4);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(256), 0, 0, data, c, bad, iters, words);
        hipEventRecord(b, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
        unsigned hb; Ctl hc; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost);
        hbad += hb; hfail += hc.fail;
    }
    printf("mode %d  %3d workgroups  %4d B per workgroup: %6.2f us per iteration (produce + barrier + consume), wrong words %u, timeouts %u\n",
           MODE, nwg, words * 4, best * 1e3f / iters, hbad, hfail);
    hipFree(data); hipFree(bad); hipFree(c);
}

int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("device has %d CUs\n", cus);
    const int iters = 2000;
    for (int nwg : {64, 128, 256}) {
        if (nwg > cus) continue;          // every workgroup must be resident (one per CU): a larger grid would deadlock the barrier
        for (int words : {64, 1024}) {
            run<0>(nwg, iters, words);
            run<1>(nwg, iters, words);
            run<2>(nwg, iters, words);
        }
    }
    return 0;
}
