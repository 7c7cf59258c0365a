// Traffic-only stand-in for the folded Adam user phase: per 16-lane team the nine rows the step touches (weights and both
// moments of one user row and two item rows, 256 B each) are read and written back, plus the three 4-byte step stamps.
// Nothing else: the time of this kernel is what the access pattern costs on this memory system.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(256) void adam_traffic(float4 *Wu, float4 *Mu, float4 *Vu, float4 *Wi, float4 *Mi, float4 *Vi,
                                                               int *lastU, int *lastI, const int *u, const int *p, const int *n,
                                                               int B, int stamps, int t) {
    extern __shared__ int occupancy_limiter[];      // dynamic LDS only limits the workgroups per CU
    if (B < 0) occupancy_limiter[threadIdx.x] = 1;
    const int lane = threadIdx.x & 15, team = (blockIdx.x * 256 + threadIdx.x) >> 4;
    if (team >= B) return;
    const int64_t ru = (int64_t)u[team] * 16 + lane, rp = (int64_t)p[team] * 16 + lane, rn = (int64_t)n[team] * 16 + lane;
    int lu = 0, lp = 0, ln = 0;
    if (stamps) { lu = lastU[u[team]]; lp = lastI[p[team]]; ln = lastI[n[team]]; }
    float4 a = Wu[ru], b = Mu[ru], c = Vu[ru], d = Wi[rp], e = Mi[rp], f = Vi[rp], g = Wi[rn], h = Mi[rn], i = Vi[rn];
    const float s = 1.0f + 1e-9f * (float)(lu + lp + ln);
    a.x *= s; b.x *= s; c.x *= s; d.x *= s; e.x *= s; f.x *= s; g.x *= s; h.x *= s; i.x *= s;
    Wu[ru] = a; Mu[ru] = b; Vu[ru] = c; Wi[rp] = d; Mi[rp] = e; Vi[rp] = f; Wi[rn] = g; Mi[rn] = h; Vi[rn] = i;
    if (stamps && lane == 0) { lastU[u[team]] = t; lastI[p[team]] = t; lastI[n[team]] = t; }
}
extern "C" void run(void *Wu, void *Mu, void *Vu, void *Wi, void *Mi, void *Vi, void *lastU, void *lastI, const void *u,
                    const void *p, const void *n, int B, int stamps, int t, int lds_bytes, void *stream) {
    hipLaunchKernelGGL(adam_traffic, dim3((B * 16 + 255) / 256), dim3(256), lds_bytes, (hipStream_t)stream, (float4 *)Wu, (float4 *)Mu,
                       (float4 *)Vu, (float4 *)Wi, (float4 *)Mi, (float4 *)Vi, (int *)lastU, (int *)lastI, (const int *)u,
                       (const int *)p, (const int *)n, B, stamps, t);
}
