// Micro-benchmark (round 6): pure-store ceilings of the per-env observation run as the clock-grid / lane rollouts write it —
// one WAVE per env, each step an E-float run at out[(t * N + env) * E], 16-byte stores — by STORE FLAVOUR and ORDER:
//   flavour: plain | nt | sc1 | sc0 sc1   (MI355X_MICROARCH.md: plain / nt keep the line in the XCD's L2, sc1 drops it)
//   order  : planes — the three channel planes interleaved, 1 KiB of r, 1 KiB of g, 1 KiB of b, ... (grid_observe today)
//            linear — the run front to back
//   lds    : dynamic LDS per workgroup (bytes) = residency, as launch_grid_rollout sets it
// shapes: cfg5 (8192 x 36 x 36 'default': E = 3888, T = 16), cfg3 'default' (65536 x 9 x 9: E = 243 -> per wave 32 envs)
// build: hipcc --offload-arch=gfx950 -O3 -o store_flavours store_flavours.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float vf4 __attribute__((ext_vector_type(4)));

template <int FL>
__device__ __forceinline__ void st16(float *p, vf4 v)
{
    if (FL == 0) *(vf4 *)p = v;
    else if (FL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
    else if (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

// one wave per env; C cells per plane (C % 4 == 0), 3 planes
template <int FL, bool LINEAR>
__global__ __launch_bounds__(256) void env_runs(float *out, int T, long long N, int C)
{
    extern __shared__ int lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const long long env = (long long)blockIdx.x * wpb + wave;
    if (env >= N) return;
    const int E = 3 * C;
    const vf4 v = vf4{(float)lane, 1.f, 2.f, 3.f};
    for (int t = 0; t < T; ++t) {
        float *o = out + ((long long)t * N + env) * E;
        if (LINEAR) {
            for (int i = 4 * lane; i < E; i += 256) st16<FL>(o + i, v);
        } else {
            for (int c0 = 4 * lane; c0 < ((C + 255) & ~255); c0 += 256)
                if (c0 < C) {
                    st16<FL>(o + c0, v);
                    st16<FL>(o + C + c0, v);
                    st16<FL>(o + 2 * C + c0, v);
                }
        }
    }
    if (lds[0] == 12345) out[0] = 0.f;
}

// one wave per EPW consecutive envs of E floats each (the lane kernels): a run of EPW * E floats per step, linear
template <int FL>
__global__ __launch_bounds__(256) void wave_runs(float *out, int T, long long N, int E, int EPW)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const long long env0 = ((long long)blockIdx.x * wpb + wave) * EPW;
    if (env0 >= N) return;
    const int run = EPW * E;
    const vf4 v = vf4{(float)lane, 1.f, 2.f, 3.f};
    for (int t = 0; t < T; ++t) {
        float *o = out + ((long long)t * N + env0) * E;
        for (int i = 4 * lane; i < run; i += 256) st16<FL>(o + i, v);
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

static const char *FLN[] = {"plain", "nt", "sc1", "sc0 sc1"};

template <int FL>
static void cfg5(float *buf, long long N, int T, int C, double bytes)
{
    for (int lds : {0, 13312 * 4, 65536}) { // all resident / 12 waves per CU / 8 waves per CU (4 per workgroup)
        dim3 grid((unsigned)((N + 3) / 4)), block(256);
        hipFuncSetAttribute((const void *)env_runs<FL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute((const void *)env_runs<FL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        float a = timeit([&] { hipLaunchKernelGGL((env_runs<FL, false>), grid, block, (size_t)lds, 0, buf, T, N, C); });
        float b = timeit([&] { hipLaunchKernelGGL((env_runs<FL, true>), grid, block, (size_t)lds, 0, buf, T, N, C); });
        printf("cfg5 %-8s lds %5d : planes %6.3f ms %5.2f TB/s | linear %6.3f ms %5.2f TB/s\n", FLN[FL], lds, a, bytes / a / 1e9, b,
               bytes / b / 1e9);
    }
}

template <int FL>
static void cfg3(float *buf, long long N, int T, int E, double bytes)
{
    for (int epw : {16, 32, 64}) {
        dim3 grid((unsigned)((N / epw + 3) / 4)), block(256);
        float a = timeit([&] { hipLaunchKernelGGL((wave_runs<FL>), grid, block, 0, 0, buf, T, N, E, epw); });
        printf("cfg3-like E=%d %-8s %2d envs per wave : %6.3f ms %5.2f TB/s\n", E, FLN[FL], epw, a, bytes / a / 1e9);
    }
}

int main()
{
    {
        const long long N = 8192; const int T = 16, C = 1296;
        const double bytes = 4.0 * 3 * C * T * N;
        float *buf; hipMalloc(&buf, (size_t)bytes);
        float ms = timeit([&] { hipLaunchKernelGGL(fill, dim3(256 * 32), dim3(256), 0, 0, (vf4 *)buf, (long long)(bytes / 16)); });
        printf("fill linear x4 (8192 WG), %.2f GB      : %7.3f ms  %5.2f TB/s\n", bytes / 1e9, ms, bytes / ms / 1e9);
        cfg5<0>(buf, N, T, C, bytes);
        cfg5<1>(buf, N, T, C, bytes);
        cfg5<2>(buf, N, T, C, bytes);
        cfg5<3>(buf, N, T, C, bytes);
        hipFree(buf);
    }
    { // the same pure-store launches over 10 allocations held at once: does the FLOOR move with where the buffer lies?
        const long long N = 8192; const int T = 16, C = 1296;
        const double bytes = 4.0 * 3 * C * T * N;
        float *bufs[10];
        for (int i = 0; i < 10; ++i) hipMalloc(&bufs[i], (size_t)bytes);
        dim3 grid((unsigned)((N + 3) / 4)), block(256);
        for (int i = 0; i < 10; ++i) {
            float *buf = bufs[i];
            float a = timeit([&] { hipLaunchKernelGGL((env_runs<0, false>), grid, block, (size_t)53248, 0, buf, T, N, C); });
            float b = timeit([&] { hipLaunchKernelGGL((env_runs<0, true>), grid, block, (size_t)53248, 0, buf, T, N, C); });
            float f = timeit([&] { hipLaunchKernelGGL(fill, dim3(256 * 32), dim3(256), 0, 0, (vf4 *)buf, (long long)(bytes / 16)); });
            printf("cfg5 plain, allocation %d at %p : planes %6.3f ms %5.2f TB/s | linear %6.3f ms %5.2f TB/s | fill %6.3f ms %5.2f TB/s\n", i,
                   (void *)buf, a, bytes / a / 1e9, b, bytes / b / 1e9, f, bytes / f / 1e9);
        }
        for (int i = 0; i < 10; ++i) hipFree(bufs[i]);
    }
    {
        const long long N = 65536; const int T = 16, E = 244; // (243 rounded to whole float4s: the byte model of a 'default' 9 x 9 run)
        const double bytes = 4.0 * E * T * N;
        float *buf; hipMalloc(&buf, (size_t)bytes);
        cfg3<0>(buf, N, T, E, bytes);
        cfg3<1>(buf, N, T, E, bytes);
        cfg3<2>(buf, N, T, E, bytes);
        hipFree(buf);
    }
    return 0;
}
