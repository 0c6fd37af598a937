// Micro-benchmark (round 4): why does torch's fill write 1.97 GB at 6.9 TB/s when every persistent-workgroup pattern of
// store_runs.hip tops out at 5.4 - 5.7?  Same bytes (the cfg4 observation stream of 16 steps), different ORDER IN TIME:
//   tiny P      : one block of 256 threads per P bytes, linear, no loop (torch's vectorized fill: P = 4 KB .. 16 KB)
//   runs        : one block per (t, agent, group of G envs) = one 7500 G byte run, blocks in ADDRESS order, W waves per block
//   runs_tmajor : the same blocks ordered (t, group, agent): the four agents' runs of a group back to back in dispatch order
//   persist R   : R resident workgroups per CU loop over the runs in address order (grid-stride) — long-lived waves
// build: hipcc --offload-arch=gfx950 -O3 -o store_window store_window.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float vf4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void tiny(vf4 *out, long long n4, int per_thread)
{
    const vf4 v = vf4{1.f, 2.f, 3.f, 4.f};
    long long base = (long long)blockIdx.x * (256 * per_thread) + threadIdx.x;
    for (int i = 0; i < per_thread; ++i) {
        const long long j = base + (long long)i * 256;
        if (j < n4) out[j] = v;
    }
}

// one block = one run of run4 float4s starting at float4 index run_index * run4 (linear) or permuted (tmajor)
__global__ __launch_bounds__(1024) void runs(vf4 *out, int run4, int K, long long groups, int tmajor)
{
    const vf4 v = vf4{1.f, 2.f, 3.f, 4.f};
    long long b = blockIdx.x;
    if (tmajor) { // dispatch order (t, group, agent) -> address order (t, agent, group)
        const long long per_t = (long long)K * groups, t = b / per_t, r = b - t * per_t, g = r / K, a = r - g * K;
        b = t * per_t + a * groups + g;
    }
    vf4 *p = out + b * run4;
    for (int i = threadIdx.x; i < run4; i += blockDim.x) p[i] = v;
}

__global__ __launch_bounds__(256) void persist(vf4 *out, int run4, long long nruns)
{
    const vf4 v = vf4{1.f, 2.f, 3.f, 4.f};
    for (long long b = blockIdx.x; b < nruns; b += gridDim.x) {
        vf4 *p = out + b * run4;
        for (int i = threadIdx.x; i < run4; i += blockDim.x) p[i] = v;
    }
}

template <typename F>
static float timeit(F f, int reps = 10)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    const long long N = 4096; const int T = 16, K = 4, E = 1875;
    const double bytes = 4.0 * E * K * T * N;
    const long long n4 = (long long)(bytes / 16);
    vf4 *buf; (void)hipMalloc(&buf, (size_t)bytes);
    for (int per : {1, 2, 4, 8, 16}) {
        const long long blocks = (n4 + 256 * per - 1) / (256 * per);
        float ms = timeit([&] { hipLaunchKernelGGL(tiny, dim3((unsigned)blocks), dim3(256), 0, 0, buf, n4, per); });
        printf("tiny   %3d KB per block (%7lld blocks)      : %6.3f ms %5.2f TB/s\n", 4 * per, blocks, ms, bytes / ms / 1e9);
    }
    for (int G : {1, 4, 8, 16}) {
        const int run4 = G * E * 4 / 4 / 4 * 4 / 4; // floats G*E -> float4s (G multiple of 4 keeps it integral; G = 1: truncated)
        const int r4 = G * E / 4;
        const long long groups = N / G, nruns = (long long)T * K * groups;
        for (int W : {1, 4, 16}) {
            float a = timeit([&] { hipLaunchKernelGGL(runs, dim3((unsigned)nruns), dim3(64 * W), 0, 0, buf, r4, K, groups, 0); });
            float b = timeit([&] { hipLaunchKernelGGL(runs, dim3((unsigned)nruns), dim3(64 * W), 0, 0, buf, r4, K, groups, 1); });
            printf("runs   G=%2d (%6.1f KB) W=%2d : address order %6.3f ms %5.2f TB/s | (t, group, agent) order %6.3f ms %5.2f TB/s\n", G,
                   r4 * 16 / 1024.0, W, a, bytes / a / 1e9, b, bytes / b / 1e9);
        }
        for (int R : {1, 2, 4, 8}) {
            float c = timeit([&] { hipLaunchKernelGGL(persist, dim3(256 * R), dim3(256), 0, 0, buf, r4, nruns); });
            printf("persist G=%2d R=%d workgroups per CU, runs in address order: %6.3f ms %5.2f TB/s\n", G, R, c, bytes / c / 1e9);
        }
        (void)run4;
    }
    return 0;
}
