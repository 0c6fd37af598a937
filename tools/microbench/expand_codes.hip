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


// ---- fill form: blocks of 256 threads over PER * 4 KB of the output in plain address order, one aligned float4 per lane and
// iteration; every float's (step, agent, env, plane, cell) comes from its flat index, its class from the env's codes (global,
// L2-resident: each code is read 12 times within a few microseconds), its value from the table.
typedef float vf4 __attribute__((ext_vector_type(4)));

template <bool SC1>
__device__ __forceinline__ unsigned load_code(const unsigned short *p)
{
    if (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

template <bool COMPUTE>
__global__ __launch_bounds__(256) void fillform(Args a, int per)
{
    __shared__ float tab[24];
    const int tid = threadIdx.x;
    if (tid < 24) tab[tid] = table_value(tid >> 3, tid & 7);
    const long long nblk = (long long)gridDim.x - a.NS;
    if ((long long)blockIdx.x < a.NS) { // stepper emulation: 64 of the 256 threads work
        if (tid >= 64) return;
        const int n = blockIdx.x, lane = tid;
        for (int t = 0; t < a.T; ++t) {
            const unsigned long long t0 = wall_clock64();
            while ((long long)(wall_clock64() - t0) < (long long)a.step_cycles) __builtin_amdgcn_s_sleep(8);
            unsigned *dst = (unsigned *)(a.codes + ((size_t)t * a.N + n) * CODE_STRIDE);
            for (int i = 0; i < 5; ++i) {
                const int c0 = 2 * (i * 64 + lane);
                const unsigned w = code_of(a.salt, t, n, c0) | (code_of(a.salt, t, n, c0 + 1) << 16);
                __hip_atomic_store(dst + i * 64 + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(a.ready + (size_t)t * a.N + n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const long long blk = (long long)blockIdx.x - a.NS;
    const long long total4 = (long long)a.T * a.K * a.N * E / 4;
    const long long first4 = blk * 256 * per;
    if (a.NS > 0) { // the envs this block's floats belong to: their codes must be there
        if (tid < 64) {
            const long long q0 = first4 * 4 / E;
            long long last4 = first4 + 256ll * per - 1; if (last4 >= total4) last4 = total4 - 1;
            const long long q1 = (last4 * 4 + 3) / E;
            for (long long q = q0 + tid; q <= q1; q += 64) {
                const long long n = q % a.N, t = q / a.N / a.K;
                int spins = 0;
                while (__hip_atomic_load(a.ready + t * a.N + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    __builtin_amdgcn_s_sleep(16);
                    if (++spins > (1 << 20)) { atomicAdd(a.timeouts, 1); break; }
                }
            }
        }
    }
    __syncthreads();
    (void)nblk;
    for (int i = 0; i < per; ++i) {
        const long long j4 = first4 + (long long)i * 256 + tid;
        if (j4 >= total4) break;
        vf4 v = vf4{1.f, 2.f, 3.f, 4.f};
        if (COMPUTE) {
            const long long j = j4 * 4;
            long long q = j / E;
            int r = (int)(j - q * E);
            float out[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int rr = r + e; long long qq = q;
                if (rr >= E) { rr -= E; ++qq; }
                const int c = rr >= 2 * S2 ? 2 : rr >= S2 ? 1 : 0;
                const int cell = rr - c * S2;
                const long long n = qq % a.N, tk = qq / a.N;
                const int k = (int)(tk % a.K); const long long t = tk / a.K;
                const unsigned code = a.NS > 0 ? load_code<true>(a.codes + (t * a.N + n) * CODE_STRIDE + cell)
                                               : load_code<false>(a.codes + (t * a.N + n) * CODE_STRIDE + cell);
                out[e] = tab[c * 8 + ((code >> (3 * k)) & 7)];
            }
            v = vf4{out[0], out[1], out[2], out[3]};
        }
        ((vf4 *)a.out)[j4] = v;
    }
}

// ---- fill form 2: an item is (step, agent, group of 4 envs) = 4 * 1875 floats = 1875 ALIGNED float4s (30 000 bytes), split
// into PARTS blocks of 256 threads; 32-bit arithmetic per lane, no division: env by three compares, plane and cell carried.
template <bool SC1, int PARTS, bool STAGE = false>
__global__ __launch_bounds__(256) void fillform2(Args a)
{
    __shared__ float tab[24];
    __shared__ unsigned short lcodes[STAGE ? 4 * CODE_STRIDE : 2];
    const int tid = threadIdx.x;
    if (tid < 24) tab[tid] = table_value(tid >> 3, tid & 7);
    if ((int)blockIdx.x < a.NS) { // stepper emulation: 64 of the 256 threads work
        if (tid >= 64) return;
        const int n = blockIdx.x, lane = tid;
        for (int t = 0; t < a.T; ++t) {
            const unsigned long long t0 = wall_clock64();
            while ((long long)(wall_clock64() - t0) < (long long)a.step_cycles) __builtin_amdgcn_s_sleep(8);
            unsigned *dst = (unsigned *)(a.codes + ((size_t)t * a.N + n) * CODE_STRIDE);
            for (int i = 0; i < 5; ++i) {
                const int c0 = 2 * (i * 64 + lane);
                const unsigned w = code_of(a.salt, t, n, c0) | (code_of(a.salt, t, n, c0 + 1) << 16);
                __hip_atomic_store(dst + i * 64 + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(a.ready + (size_t)t * a.N + n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const unsigned blk = blockIdx.x - a.NS;
    const unsigned item = blk / PARTS, part = blk - item * PARTS;
    const unsigned NG = a.N / 4;
    const unsigned tk = item / NG, g4 = item - tk * NG;
    const unsigned t = tk / a.K, k = tk - t * a.K;
    const unsigned env0 = g4 * 4;
    if (a.NS > 0) {
        if (tid < 4) {
            int spins = 0;
            while (__hip_atomic_load(a.ready + (size_t)t * a.N + env0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1 << 20)) { atomicAdd(a.timeouts, 1); break; }
            }
        }
    }
    if (a.NS > 0) __syncthreads();
    const unsigned short *cb = a.codes + ((size_t)t * a.N + env0) * CODE_STRIDE;
    if (STAGE) { // every load of the block up front: 4 envs x 320 dwords = 5 per thread
        const unsigned *src = (const unsigned *)cb;
        unsigned w[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) w[i] = SC1 ? __hip_atomic_load(src + i * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : src[i * 256 + tid];
#pragma unroll
        for (int i = 0; i < 5; ++i) ((unsigned *)lcodes)[i * 256 + tid] = w[i];
    }
    __syncthreads();
    vf4 *ob = (vf4 *)(a.out + ((size_t)tk * a.N + env0) * E);
    const unsigned f0 = part * E / PARTS, f1 = (part + 1) * E / PARTS; // float4 indices of this part (E float4s per item)
    const unsigned sh = 3 * k;
    for (unsigned f = f0 + tid; f < f1; f += 256) {
        const unsigned x = 4 * f;
        unsigned e = (x >= (unsigned)E) + (x >= 2u * E) + (x >= 3u * E);
        unsigned r = x - e * E;
        unsigned c = (r >= (unsigned)S2) + (r >= 2u * S2);
        unsigned cell = r - c * S2;
        float out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned code = STAGE ? lcodes[e * CODE_STRIDE + cell]
                                : SC1 ? __hip_atomic_load(cb + e * CODE_STRIDE + cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : cb[e * CODE_STRIDE + cell];
            out[i] = tab[c * 8 + ((code >> sh) & 7)];
            ++cell;
            if (cell == (unsigned)S2) { cell = 0; ++c; if (c == 3) { c = 0; ++e; } }
        }
        ob[f] = vf4{out[0], out[1], out[2], out[3]};
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
        for (int stage : {1, 0}) for (int parts : {1, 2, 4, 8}) {
            if (stage && parts == 8) continue;
            a.G = 1; a.NS = steppers ? a.N : 0; a.step_cycles = 1000;
            const unsigned blocks = a.NS + (unsigned)(a.T * a.K * (a.N / 4) * parts);
            float best = 1e9f, sum = 0;
            const int reps = 6;
            unsigned long long hbad = 0; int hto = 0;
            for (int r = 0; r < reps; ++r) {
                a.salt = ++salt;
                (void)hipMemsetAsync(a.ready, 0, (size_t)a.T * a.N * 4);
                (void)hipMemsetAsync(a.timeouts, 0, 4);
                if (!steppers) hipLaunchKernelGGL(fill_codes, dim3(2048), dim3(256), 0, 0, a);
                (void)hipEventRecord(e0);
                if (stage && steppers) {
                    if (parts == 1) hipLaunchKernelGGL((fillform2<true, 1, true>), dim3(blocks), dim3(256), 0, 0, a);
                    if (parts == 2) hipLaunchKernelGGL((fillform2<true, 2, true>), dim3(blocks), dim3(256), 0, 0, a);
                    if (parts == 4) hipLaunchKernelGGL((fillform2<true, 4, true>), dim3(blocks), dim3(256), 0, 0, a);
                } else if (stage) {
                    if (parts == 1) hipLaunchKernelGGL((fillform2<false, 1, true>), dim3(blocks), dim3(256), 0, 0, a);
                    if (parts == 2) hipLaunchKernelGGL((fillform2<false, 2, true>), dim3(blocks), dim3(256), 0, 0, a);
                    if (parts == 4) hipLaunchKernelGGL((fillform2<false, 4, true>), dim3(blocks), dim3(256), 0, 0, a);
                } else if (steppers) {
                    if (parts == 1) hipLaunchKernelGGL((fillform2<true, 1>), dim3(blocks), dim3(256), 0, 0, a);
                    if (parts == 2) hipLaunchKernelGGL((fillform2<true, 2>), dim3(blocks), dim3(256), 0, 0, a);
                    if (parts == 4) hipLaunchKernelGGL((fillform2<true, 4>), dim3(blocks), dim3(256), 0, 0, a);
                    if (parts == 8) hipLaunchKernelGGL((fillform2<true, 8>), dim3(blocks), dim3(256), 0, 0, a);
                } else {
                    if (parts == 1) hipLaunchKernelGGL((fillform2<false, 1>), dim3(blocks), dim3(256), 0, 0, a);
                    if (parts == 2) hipLaunchKernelGGL((fillform2<false, 2>), dim3(blocks), dim3(256), 0, 0, a);
                    if (parts == 4) hipLaunchKernelGGL((fillform2<false, 4>), dim3(blocks), dim3(256), 0, 0, a);
                    if (parts == 8) hipLaunchKernelGGL((fillform2<false, 8>), dim3(blocks), dim3(256), 0, 0, a);
                }
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
            printf("fillform2 stage=%d steppers=%d parts=%d (%5.1f KB per block): best %6.3f ms  mean %6.3f ms  %5.2f TB/s   wrong %llu timeouts %d\n",
                   stage, steppers, parts, 30.0 / parts, best, sum / (reps - 1), bytes / best / 1e9, hbad, hto);
            fflush(stdout);
        }
    }
    for (int steppers : {0, 1}) {
        for (int per : {1, 4}) {
            for (int compute : {0, 1}) {
                if (steppers || compute) continue;
                a.G = 1; a.NS = steppers ? a.N : 0; a.step_cycles = 1000;
                const long long total4 = (long long)a.T * a.K * a.N * E / 4;
                const unsigned blocks = a.NS + (unsigned)((total4 + 256ll * per - 1) / (256ll * per));
                float best = 1e9f, sum = 0;
                const int reps = 6;
                unsigned long long hbad = 0; int hto = 0;
                for (int r = 0; r < reps; ++r) {
                    a.salt = ++salt;
                    (void)hipMemsetAsync(a.ready, 0, (size_t)a.T * a.N * 4);
                    (void)hipMemsetAsync(a.timeouts, 0, 4);
                    if (!steppers) hipLaunchKernelGGL(fill_codes, dim3(2048), dim3(256), 0, 0, a);
                    (void)hipEventRecord(e0);
                    if (compute) hipLaunchKernelGGL(fillform<true>, dim3(blocks), dim3(256), 0, 0, a, per);
                    else hipLaunchKernelGGL(fillform<false>, dim3(blocks), dim3(256), 0, 0, a, per);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                    if (r) { sum += ms; if (ms < best) best = ms; }
                    if (r == reps - 1 && compute) {
                        (void)hipMemset(bad, 0, 8);
                        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, a, bad);
                        (void)hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost);
                        (void)hipMemcpy(&hto, a.timeouts, 4, hipMemcpyDeviceToHost);
                    }
                }
                printf("fillform steppers=%d %2d KB per block compute=%d: best %6.3f ms  mean %6.3f ms  %5.2f TB/s   wrong %llu timeouts %d\n",
                       steppers, 4 * per, compute, best, sum / (reps - 1), bytes / best / 1e9, hbad, hto);
                fflush(stdout);
            }
        }
    }
    for (int steppers : {0, 1}) {
        for (int step_us : {10}) {
            if (!steppers && step_us != 10) continue;
            for (int G : {2, 4, 8}) {
                for (int lds_kb : {4, 12}) {
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
