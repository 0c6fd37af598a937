// Micro-benchmark: how fast can one-wave-per-env kernels STREAM observations to HBM on MI355X, by store shape?
// The cfg5 / cfg4 rollouts write 15.5 KB / 30 KB of fp32 observation per env-step; round 1 used one dword per lane
// per instruction (256 B per wave instruction).  Shapes timed here, same bytes, same one-wave-per-env ownership:
//   dword    : lane writes o[plane*C + lane + 64k]                (global_store_dword, 256 B / instruction)
//   dwordx2  : lane writes 2 consecutive floats                   (512 B / instruction)
//   dwordx4  : lane writes 4 consecutive floats                   (1 KB / instruction)
//   dwordx4nt: the same with the non-temporal hint
// and with a fake "compute phase" of V dependent VALU ops between a step's stores (0 / 256 / 1024) to see whether
// stores overlap with the compute of the next step inside ONE wave, at 1 / 2 / 4 waves per workgroup and with a
// register budget limiting the waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o store_bw store_bw.hip ; run: ./store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int W, bool NT, int VALU>
__global__ __launch_bounds__(256) void store_kernel(float *out, int floats_per_env_step, int T, long long N, int skew)
{
    extern __shared__ int dummy_lds[];
    if (skew > 0) { const int g = (int)(((unsigned)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u) >> 29); for (int i = 0; i < g * skew; ++i) __builtin_amdgcn_s_sleep(127); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const long long env = (long long)blockIdx.x * wpb + wave;
    if (env >= N) return;
    float v = (float)lane;
    const int per_instr = 64 * W;
    for (int t = 0; t < T; ++t) {
        float *o = out + ((long long)t * N + env) * floats_per_env_step;
#pragma unroll 1
        for (int i = 0; i < VALU; ++i) asm volatile("v_add_f32 %0, %0, 1.0" : "+v"(v));
        for (int base = 0; base < floats_per_env_step; base += per_instr) {
            const int idx = base + lane * W;
            if (idx < floats_per_env_step) {
                if (W == 1) {
                    if (NT) __builtin_nontemporal_store(v, o + idx); else o[idx] = v;
                } else if (W == 2) {
                    vf2 x = {v, v};
                    if (NT) __builtin_nontemporal_store(x, (vf2 *)(o + idx)); else *(vf2 *)(o + idx) = x;
                } else {
                    vf4 x = {v, v, v, v};
                    if (NT) __builtin_nontemporal_store(x, (vf4 *)(o + idx)); else *(vf4 *)(o + idx) = x;
                }
            }
        }
    }
}

template <int W, bool NT, int VALU>
static void run(const char *name, float *buf, int fpe, int T, long long N, int wpb, int lds_bytes = 0, int skew = 0)
{
    dim3 block(64 * wpb), grid((unsigned)((N + wpb - 1) / wpb));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((store_kernel<W, NT, VALU>), grid, block, lds_bytes, 0, buf, fpe, T, N, skew);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((store_kernel<W, NT, VALU>), grid, block, lds_bytes, 0, buf, fpe, T, N, skew);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    double bytes = 4.0 * fpe * T * N;
    printf("%-28s N=%6lld fpe=%6d T=%3d wpb=%d valu=%4d lds=%6d skew=%d : %8.3f ms  %7.2f TB/s\n", name, N, fpe, T, wpb, VALU, lds_bytes, skew, ms, bytes / ms / 1e9);
}

int main()
{
    const long long N = 8192; const int fpe = 3 * 36 * 36;
    float *buf; hipMalloc(&buf, 4ull * fpe * 64 * N);
    // (1) does a longer launch stream faster (waves drifting out of lockstep)?
    for (int T : {16, 64}) {
        run<4, false, 0>("dwordx4", buf, fpe, T, N, 4);
        run<1, false, 0>("dword", buf, fpe, T, N, 4);
    }
    // (2) waves per CU limited through dynamic LDS per 4-wave block: 160 KB / lds = blocks per CU
    for (int lds : {0, 20480, 26624, 40960, 65536}) {
        run<4, false, 0>("dwordx4 occupancy", buf, fpe, 16, N, 4, lds);
        run<4, false, 0>("dwordx4 occupancy", buf, fpe, 64, N, 4, lds);
    }
    // (3) start-up skew of pseudo-random wave groups (8 groups x skew x ~3.4 us)
    for (int skew : {1, 2, 4, 8}) {
        run<4, false, 0>("dwordx4 skew", buf, fpe, 16, N, 4, 0, skew);
        run<4, false, 0>("dwordx4 skew", buf, fpe, 64, N, 4, 0, skew);
    }
    // (4) with some arithmetic per step (the real kernel: ~300 VALU per step)
    for (int lds : {0, 26624}) {
        run<4, false, 64>("dwordx4 + 64 loop iters", buf, fpe, 16, N, 4, lds);
        run<4, false, 64>("dwordx4 + 64 loop iters", buf, fpe, 64, N, 4, lds);
    }
    return 0;
}
