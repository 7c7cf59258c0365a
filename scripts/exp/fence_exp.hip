// Experiment: cost of "last arriver finishes the shared row" (release fence + device atomic + acquire fence) inside a
// user-phase-like kernel on gfx950, against the plain kernel.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC.
#include <hip/hip_runtime.h>
#include <stdint.h>

// team of 16 lanes per triplet; rows of 64 floats
template <int MODE>   // 0: plain (3 reads, 3 writes); 1: shared fraction stashes z + fence + atomic + finisher
__global__ __launch_bounds__(256) void k(float4 *U, float4 *I, float4 *Z, int *cnt, const int *u, const int *p, const int *n,
                                         const int *partner, int B) {
    const int t = blockIdx.x * 16 + threadIdx.x / 16, lane = threadIdx.x % 16;
    if (t >= B) return;
    const int uu = u[t], pp = p[t], nn = n[t], pr = partner[t];   // pr >= 0: this triplet's p row is shared with triplet pr
    float4 a = U[(size_t)uu * 16 + lane], b = I[(size_t)pp * 16 + lane], c = I[(size_t)nn * 16 + lane];
    float d = a.x * (b.x - c.x) + a.y * (b.y - c.y) + a.z * (b.z - c.z) + a.w * (b.w - c.w);
    for (int o = 8; o; o >>= 1) d += __shfl_xor(d, o, 16);
    const float cf = 1.f / (1.f + __expf(d)) * 1e-3f;
    float4 gu = make_float4(cf * (b.x - c.x), cf * (b.y - c.y), cf * (b.z - c.z), cf * (b.w - c.w));
    float4 zp = make_float4(cf * a.x, cf * a.y, cf * a.z, cf * a.w);
    U[(size_t)uu * 16 + lane] = make_float4(a.x - gu.x, a.y - gu.y, a.z - gu.z, a.w - gu.w);
    I[(size_t)nn * 16 + lane] = make_float4(c.x + zp.x, c.y + zp.y, c.z + zp.z, c.w + zp.w);
    if (MODE == 0 || pr < 0) {
        I[(size_t)pp * 16 + lane] = make_float4(b.x - zp.x, b.y - zp.y, b.z - zp.z, b.w - zp.w);
        return;
    }
    if (MODE == 3) {
        // agent-scope (write-through, sc1) stores of z, wait for their completion, relaxed device atomic; the finisher reads
        // z with agent-scope loads: no whole-L2 writeback / invalidate
        float *zz = reinterpret_cast<float *>(&Z[(size_t)t * 16 + lane]);
        __hip_atomic_store(zz + 0, zp.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(zz + 1, zp.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(zz + 2, zp.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(zz + 3, zp.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int old = 0;
        const int slot = t < pr ? t : pr;
        if (lane == 0) old = __hip_atomic_fetch_add(&cnt[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __shfl(old, 0, 16);
        if (old == 1) {
            const float *q0 = reinterpret_cast<const float *>(&Z[(size_t)slot * 16 + lane]);
            const float *q1 = reinterpret_cast<const float *>(&Z[(size_t)(t < pr ? pr : t) * 16 + lane]);
            float4 z0, z1;
            z0.x = __hip_atomic_load(q0 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z0.y = __hip_atomic_load(q0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z0.z = __hip_atomic_load(q0 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z0.w = __hip_atomic_load(q0 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z1.x = __hip_atomic_load(q1 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z1.y = __hip_atomic_load(q1 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z1.z = __hip_atomic_load(q1 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z1.w = __hip_atomic_load(q1 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            I[(size_t)pp * 16 + lane] = make_float4(b.x - (z0.x + z1.x), b.y - (z0.y + z1.y), b.z - (z0.z + z1.z), b.w - (z0.w + z1.w));
            if (lane == 0) __hip_atomic_store(&cnt[slot], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    Z[(size_t)t * 16 + lane] = zp;
    if (MODE == 1) {
        __threadfence();
        int old = 0;
        const int slot = t < pr ? t : pr;
        if (lane == 0) old = atomicAdd(&cnt[slot], 1);
        old = __shfl(old, 0, 16);
        if (old == 1) {   // second of the pair: finish the row in fixed order (lower triplet first)
            __threadfence();
            float4 z0 = Z[(size_t)slot * 16 + lane], z1 = Z[(size_t)(t < pr ? pr : t) * 16 + lane];
            I[(size_t)pp * 16 + lane] = make_float4(b.x - (z0.x + z1.x), b.y - (z0.y + z1.y), b.z - (z0.z + z1.z), b.w - (z0.w + z1.w));
            if (lane == 0) cnt[slot] = 0;
        }
    }
}

// the separate second kernel of today's design: one team per pair
__global__ __launch_bounds__(256) void k_item(float4 *I, const float4 *Z, const int *p, const int *pairs, int n_pairs) {
    const int j = blockIdx.x * 16 + threadIdx.x / 16, lane = threadIdx.x % 16;
    if (j >= n_pairs) return;
    const int t0 = pairs[2 * j], t1 = pairs[2 * j + 1];
    const int pp = p[t0];
    float4 b = I[(size_t)pp * 16 + lane], z0 = Z[(size_t)t0 * 16 + lane], z1 = Z[(size_t)t1 * 16 + lane];
    I[(size_t)pp * 16 + lane] = make_float4(b.x - (z0.x + z1.x), b.y - (z0.y + z1.y), b.z - (z0.z + z1.z), b.w - (z0.w + z1.w));
}

extern "C" void run(int mode, float *U, float *I, float *Z, int *cnt, const int *u, const int *p, const int *n,
                    const int *partner, int B, const int *pairs, int n_pairs, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    dim3 g((B + 15) / 16), b(256);
    if (mode == 0) hipLaunchKernelGGL(k<0>, g, b, 0, s, (float4 *)U, (float4 *)I, (float4 *)Z, cnt, u, p, n, partner, B);
    if (mode == 1) hipLaunchKernelGGL(k<1>, g, b, 0, s, (float4 *)U, (float4 *)I, (float4 *)Z, cnt, u, p, n, partner, B);
    if (mode == 3) hipLaunchKernelGGL(k<3>, g, b, 0, s, (float4 *)U, (float4 *)I, (float4 *)Z, cnt, u, p, n, partner, B);
    if (mode == 2) {   // two kernels: stash only, then the pair kernel
        hipLaunchKernelGGL(k<2>, g, b, 0, s, (float4 *)U, (float4 *)I, (float4 *)Z, cnt, u, p, n, partner, B);
        hipLaunchKernelGGL(k_item, dim3((n_pairs + 15) / 16), b, 0, s, (float4 *)I, (const float4 *)Z, p, pairs, n_pairs);
    }
}
