// Micro-benchmark (round 4): pure-store ceilings of the MultiSnake 'full' observation layout (T, K, N, 3*S*S) when one
// WORKGROUP owns G consecutive envs, so that each agent's observations of those envs are ONE linear run of G * 7500
// bytes written by all the workgroup's writer waves.  K = 4, S = 25, N = 4096, T = 16.
//   run   G W x4 : workgroup = G envs, W writer waves; for agent a: the W waves sweep the run [G*7500 B] with 16-byte
//                  stores, consecutive waves on consecutive 1 KiB pieces
//   runa  G W x4 : the same, but wave w owns agent (w % K) and the waves of one agent split that run (K streams per WG)
//   dword variants: 4-byte stores (256 B per wave-instruction)
//   fill        : one linear fill of the whole buffer (the plain-store ceiling at this footprint)
//   idle waves  : +S extra waves per workgroup that only spin on the LDS flag of a stepper (occupancy as in the real kernel)
// build: hipcc --offload-arch=gfx950 -O3 -o store_runs store_runs.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float vf4 __attribute__((ext_vector_type(4)));

constexpr int K = 4, C = 625, E = 3 * C; // floats per (agent, env)

// MODE 0: all W waves sweep agent 0's run, then agent 1's, ...   MODE 1: wave w -> agent w % K, W / K waves per agent
template <int MODE, bool X4>
__global__ __launch_bounds__(1024) void runs(float *out, int T, long long N, int G, int W)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= W) return;
    const long long env0 = (long long)blockIdx.x * G;
    if (env0 >= N) return;
    const float v = (float)lane;
    const int run = G * E; // floats in one agent's run
    for (int t = 0; t < T; ++t) {
        if (MODE == 0) {
            for (int a = 0; a < K; ++a) {
                float *b = out + (((long long)t * K + a) * N + env0) * E;
                if (X4) for (int i = wave * 64 + lane; i < run / 4; i += 64 * W) *(vf4 *)(b + 4 * i) = vf4{v, v, v, v};
                else for (int i = wave * 64 + lane; i < run; i += 64 * W) b[i] = v;
            }
        } else {
            const int a = wave % K, w = wave / K, wa = W / K;
            float *b = out + (((long long)t * K + a) * N + env0) * E;
            if (X4) for (int i = w * 64 + lane; i < run / 4; i += 64 * wa) *(vf4 *)(b + 4 * i) = vf4{v, v, v, v};
            else for (int i = w * 64 + lane; i < run; i += 64 * wa) b[i] = v;
        }
    }
}

// the real kernel's time structure: per step the writer waits for a barrier with G "stepper" waves that spend `spin`
// cycles of dependent LDS traffic per step (the transition), then writes while the steppers do the next step
template <bool X4>
__global__ __launch_bounds__(1024) void runs_paced(float *out, int T, long long N, int G, int W, int spin)
{
    extern __shared__ int lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long env0 = (long long)blockIdx.x * G;
    const float v = (float)lane;
    const int run = G * E;
    if (wave >= W) { // stepper stand-in: a chain of dependent LDS reads, then the barrier
        int *mine = lds + (wave - W) * 64;
        mine[lane] = (lane + 1) & 63;
        int idx = lane;
        for (int t = 0; t < T; ++t) {
            for (int i = 0; i < spin; ++i) idx = mine[idx];
            __syncthreads();
        }
        if (idx == 12345) out[0] = 1.0f;
        return;
    }
    for (int t = 0; t < T; ++t) {
        __syncthreads();
        for (int a = 0; a < K; ++a) {
            float *b = out + (((long long)t * K + a) * N + env0) * E;
            if (X4) for (int i = wave * 64 + lane; i < run / 4; i += 64 * W) *(vf4 *)(b + 4 * i) = vf4{v, v, v, v};
            else for (int i = wave * 64 + lane; i < run; i += 64 * W) b[i] = v;
        }
    }
}

__global__ __launch_bounds__(256) void fill(vf4 *out, long long n4)
{
    const vf4 v = vf4{1.f, 2.f, 3.f, 4.f};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) out[i] = v;
}

template <typename F>
static float timeit(F f, int reps = 10)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    const long long N = 4096; const int T = 16;
    const double bytes = 4.0 * E * K * T * N;
    float *buf; hipMalloc(&buf, (size_t)bytes);
    {
        float ms = timeit([&] { hipLaunchKernelGGL(fill, dim3(256 * 8), dim3(256), 0, 0, (vf4 *)buf, (long long)(bytes / 16)); });
        printf("fill linear x4 (2048 WG)           : %7.3f ms  %5.2f TB/s\n", ms, bytes / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(fill, dim3(256 * 32), dim3(256), 0, 0, (vf4 *)buf, (long long)(bytes / 16)); });
        printf("fill linear x4 (8192 WG)           : %7.3f ms  %5.2f TB/s\n", ms, bytes / ms / 1e9);
    }
    for (int G : {1, 2, 4, 8, 16, 32}) {
        for (int W : {1, 2, 4, 8}) {
            if (W > 4 * G) continue;
            dim3 grid((unsigned)(N / G)), block(64 * W);
            float a = timeit([&] { hipLaunchKernelGGL((runs<0, true>), grid, block, 0, 0, buf, T, N, G, W); });
            float b = timeit([&] { hipLaunchKernelGGL((runs<0, false>), grid, block, 0, 0, buf, T, N, G, W); });
            float c = -1, d = -1;
            if (W % K == 0) {
                c = timeit([&] { hipLaunchKernelGGL((runs<1, true>), grid, block, 0, 0, buf, T, N, G, W); });
                d = timeit([&] { hipLaunchKernelGGL((runs<1, false>), grid, block, 0, 0, buf, T, N, G, W); });
            }
            printf("run G=%2d W=%d : x4 %6.3f ms %5.2f TB/s | dword %6.3f ms %5.2f | per-agent-waves x4 %6.3f dword %6.3f\n", G, W, a,
                   bytes / a / 1e9, b, bytes / b / 1e9, c, d);
        }
    }
    // paced by stepper stand-ins: G steppers + W writers per workgroup
    for (int G : {4, 8, 16}) {
        for (int W : {2, 4}) {
            for (int spin : {0, 100, 200, 400}) {
                dim3 grid((unsigned)(N / G)), block(64 * (W + G));
                float a = timeit([&] { hipLaunchKernelGGL((runs_paced<true>), grid, block, (size_t)G * 256, 0, buf, T, N, G, W, spin); });
                printf("paced G=%2d W=%d spin=%3d : x4 %6.3f ms %5.2f TB/s\n", G, W, spin, a, bytes / a / 1e9);
            }
        }
    }
    return 0;
}
