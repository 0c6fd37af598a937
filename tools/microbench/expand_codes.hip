// Micro-benchmark (round 4, second session): can the cfg4 observation stream (MultiSnake 4 096 x 25 x 25 x 4 `full`, 16 steps:
// 1.97 GB of fp32) be written at the rate of a fill if the TRANSITION and the WRITING are decoupled?
//   steppers  (blocks [0, NS), one wave = one env): per step spend `step_us` (the measured transition: ~10 us) and publish the
//              env's 16-bit class codes (640 u16 per env-step, 128-byte aligned) with write-through (sc1) stores, drain them,
//              then add to the ready counter of (step, group of G envs) — agent-scope atomic.
//   expanders (blocks [NS, NS + T K N/G), one wave each, in ADDRESS order of the observation tensor (T, K, N, 3 S^2)): poll the
//              counter of their (step, group) with sc1 loads, read the G envs' codes with sc1 loads into LDS, and write agent
//              k's view of the G envs — one linear run of G x 7 500 bytes — three table reads and three 4-byte stores per cell.
// Lower block ids are dispatched first, so every stepper is resident before any expander can occupy a slot (the ordering
// rocPRIM's decoupled look-back relies on); steppers never wait for expanders, so the launch cannot deadlock.  Every poll
// loop is bounded all the same and reports a timeout instead of hanging.
// build: hipcc --offload-arch=gfx950 -O3 -o expand_codes expand_codes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int S2 = 625, CODE_STRIDE = 640, E = 1875;

struct Args {
    unsigned short *codes; // [T][N][640]
    float *out;            // [T][K][N][1875]
    int *ready;            // [T][N / G]
    int *timeouts;
    int T, K, N, G, NS;
    int step_cycles; // s_memtime ticks (100 MHz: 10 ns each) a stepper spends per step
    unsigned salt;
};

__device__ __forceinline__ unsigned code_of(unsigned salt, int t, int n, int cell)
{
    unsigned h = salt * 0x9E3779B9u + (unsigned)t * 0x85EBCA6Bu + (unsigned)n * 0xC2B2AE35u + (unsigned)cell * 0x27D4EB2Fu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    unsigned w = 0; // four 3-bit classes 0..6
    for (int k = 0; k < 4; ++k) w |= ((h >> (4 * k)) % 7u) << (3 * k);
    return w;
}

__device__ __forceinline__ float table_value(int c, int cls) { return (float)(c * 8 + cls) * 0.125f; }

extern __shared__ unsigned short lds_codes[];

__global__ __launch_bounds__(64) void combo(Args a)
{
    const int lane = threadIdx.x;
    if ((int)blockIdx.x < a.NS) {
        const int n = blockIdx.x;
        for (int t = 0; t < a.T; ++t) {
            const unsigned long long t0 = wall_clock64();
            while ((long long)(wall_clock64() - t0) < (long long)a.step_cycles) __builtin_amdgcn_s_sleep(8);
            unsigned *dst = (unsigned *)(a.codes + ((size_t)t * a.N + n) * CODE_STRIDE);
            for (int i = 0; i < 5; ++i) {
                const int c0 = 2 * (i * 64 + lane);
                const unsigned w = code_of(a.salt, t, n, c0) | (code_of(a.salt, t, n, c0 + 1) << 16);
                __hip_atomic_store(dst + i * 64 + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // global_store_dword sc1
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(a.ready + (size_t)t * (a.N / a.G) + n / a.G, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    __shared__ float tab[24];
    if (lane < 24) tab[lane] = table_value(lane >> 3, lane & 7);
    const int NG = a.N / a.G;
    int item = blockIdx.x - a.NS;
    const int t = item / (a.K * NG);
    item -= t * a.K * NG;
    const int k = item / NG, g = item - k * NG;
    if (a.NS > 0) {
        const int *flag = a.ready + (size_t)t * NG + g;
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.G) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1 << 20)) { if (lane == 0) atomicAdd(a.timeouts, 1); return; }
        }
    }
    // codes of the G envs -> LDS (sc1 loads: the producer may sit on another XCD)
    const unsigned *src = (const unsigned *)(a.codes + ((size_t)t * a.N + (size_t)g * a.G) * CODE_STRIDE);
    unsigned *l32 = (unsigned *)lds_codes;
    for (int i = lane; i < a.G * (CODE_STRIDE / 2); i += 64)
        l32[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    float *run = a.out + (((size_t)t * a.K + k) * a.N + (size_t)g * a.G) * E;
    const int sh = 3 * k;
    for (int e = 0; e < a.G; ++e) {
        const unsigned short *c = lds_codes + e * CODE_STRIDE;
        float *o = run + (size_t)e * E;
#pragma unroll 2
        for (int cell = lane; cell < S2; cell += 64) {
            const int cls = (c[cell] >> sh) & 7;
            o[cell] = tab[cls];
            o[S2 + cell] = tab[8 + cls];
            o[2 * S2 + cell] = tab[16 + cls];
        }
    }
}

__global__ void check(Args a, unsigned long long *bad)
{
    const size_t total = (size_t)a.T * a.K * a.N * E;
    unsigned long long mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int cell = r % S2; r /= S2;
        const int c = r % 3; r /= 3;
        const int n = r % a.N; r /= a.N;
        const int k = r % a.K; const int t = r / a.K;
        const int cls = (code_of(a.salt, t, n, cell) >> (3 * k)) & 7;
        if (a.out[i] != table_value(c, cls)) ++mine;
    }
    if (mine) atomicAdd(bad, mine);
}

__global__ void fill_codes(Args a)
{
    const size_t total = (size_t)a.T * a.N * CODE_STRIDE;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cell = i % CODE_STRIDE; const size_t r = i / CODE_STRIDE;
        a.codes[i] = (unsigned short)code_of(a.salt, (int)(r / a.N), (int)(r % a.N), cell);
    }
}

int main()
{
    Args a{};
    a.T = 16; a.K = 4; a.N = 4096;
    const double bytes = 4.0 * E * a.K * a.T * a.N;
    (void)hipMalloc(&a.codes, (size_t)a.T * a.N * CODE_STRIDE * 2);
    (void)hipMalloc(&a.out, (size_t)bytes);
    (void)hipMalloc(&a.ready, (size_t)a.T * a.N * 4);
    (void)hipMalloc(&a.timeouts, 4);
    unsigned long long *bad; (void)hipMalloc(&bad, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    unsigned salt = 1;
    for (int steppers : {0, 1}) {
        for (int step_us : {10, 6, 14}) {
            if (!steppers && step_us != 10) continue;
            for (int G : {2, 4, 8}) {
                for (int lds_kb : {4, 6, 8, 12}) {
                    if (G * CODE_STRIDE * 2 > lds_kb * 1024) continue;
                    a.G = G; a.NS = steppers ? a.N : 0; a.step_cycles = step_us * 100;
                    const unsigned blocks = a.NS + a.T * a.K * (a.N / G);
                    float best = 1e9f, sum = 0;
                    const int reps = 6;
                    unsigned long long hbad = 0; int hto = 0;
                    for (int r = 0; r < reps; ++r) {
                        a.salt = ++salt;
                        (void)hipMemsetAsync(a.ready, 0, (size_t)a.T * a.N * 4);
                        (void)hipMemsetAsync(a.timeouts, 0, 4);
                        if (!steppers) hipLaunchKernelGGL(fill_codes, dim3(2048), dim3(256), 0, 0, a);
                        (void)hipEventRecord(e0);
                        hipLaunchKernelGGL(combo, dim3(blocks), dim3(64), lds_kb * 1024 - 128, 0, a);
                        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                        if (r) { sum += ms; if (ms < best) best = ms; }
                        if (r == reps - 1) {
                            (void)hipMemset(bad, 0, 8);
                            hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, a, bad);
                            (void)hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost);
                            (void)hipMemcpy(&hto, a.timeouts, 4, hipMemcpyDeviceToHost);
                        }
                    }
                    printf("steppers=%d step=%2d us G=%d lds=%2d KB (%2d waves/CU max): best %6.3f ms  mean %6.3f ms  %5.2f TB/s   wrong %llu timeouts %d\n",
                           steppers, step_us, G, lds_kb, 160 / lds_kb > 32 ? 32 : 160 / lds_kb, best, sum / (reps - 1), bytes / best / 1e9, hbad, hto);
                    fflush(stdout);
                }
            }
        }
    }
    return 0;
}
