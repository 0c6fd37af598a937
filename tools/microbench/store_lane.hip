// Micro-benchmark for the one-env-per-LANE rollout (round 3): how fast do the observation stores of a wave that owns
// EPW consecutive 9 x 9 envs ('partial_2': 75 floats = 300 B per env-step, [T][N][75] in HBM) reach HBM, by store shape?
//   coop_x4   : the wave's EPW*300 contiguous bytes of a step as 16-byte stores, lane l -> bytes 16*(64 i + l)  (aligned)
//   coop_x4+4 : the same with the base shifted by 4 bytes (x4 stores that are only dword aligned)
//   coop_dword: the same bytes as dword stores, lane l -> float 64 i + l
//   lane_x4   : each lane stores ITS env's 300 bytes (18 x 16 B + 12 B), lanes 300 bytes apart
// EPW x TC = 64: a chunk is TC steps of EPW envs (EPW = 64: one step; EPW = 8: eight steps, 2400-byte runs).
// build: hipcc --offload-arch=gfx950 -O3 -o store_lane store_lane.hip ; run: ./store_lane
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf3 __attribute__((ext_vector_type(3)));

enum { COOP_X4 = 0, COOP_DWORD = 1, LANE_X4 = 2, COOP_X4_NT = 3 };

template <int MODE, int EPW, int VALU>
__global__ __launch_bounds__(64) void k(float *out, long long N, int T, int shift)
{
    constexpr int TC = 64 / EPW;
    const int lane = threadIdx.x;
    const long long env0 = (long long)blockIdx.x * EPW;
    if (env0 >= N) return;
    float v = (float)lane;
    char *ob = (char *)out + 4 * shift;
    for (int t0 = 0; t0 < T; t0 += TC) {
#pragma unroll 1
        for (int i = 0; i < VALU; i += 8)
            asm volatile("v_add_f32 %0, %0, 1.0\n\tv_add_f32 %0, %0, 1.0\n\tv_add_f32 %0, %0, 1.0\n\tv_add_f32 %0, %0, 1.0\n\t"
                         "v_add_f32 %0, %0, 1.0\n\tv_add_f32 %0, %0, 1.0\n\tv_add_f32 %0, %0, 1.0\n\tv_add_f32 %0, %0, 1.0" : "+v"(v));
        if (MODE == COOP_X4 || MODE == COOP_X4_NT) {
            constexpr int GS = EPW * 75 / 4; // 16-byte groups per step
#pragma unroll
            for (int i = 0; i < 19; ++i) {
                const int j = 64 * i + lane;
                if (j < 1200) {
                    const int s = j / GS, r = j - s * GS;
                    char *q = ob + ((long long)(t0 + s) * N + env0) * 300 + 16 * r;
                    vf4 x = {v, v, v, v};
                    if (MODE == COOP_X4_NT) __builtin_nontemporal_store(x, (vf4 *)q); else *(vf4 *)q = x;
                }
            }
        } else if (MODE == COOP_DWORD) {
            constexpr int FS = EPW * 75;
#pragma unroll 5
            for (int i = 0; i < 75; ++i) {
                const int j = 64 * i + lane;
                const int s = j / FS, r = j - s * FS;
                *(float *)(ob + ((long long)(t0 + s) * N + env0) * 300 + 4 * r) = v;
            }
        } else {
            const int s = lane / EPW, e = lane - s * EPW;
            char *q = ob + ((long long)(t0 + s) * N + env0 + e) * 300;
#pragma unroll
            for (int g = 0; g < 18; ++g) {
                vf4 x = {v, v, v, v};
                *(vf4 *)(q + 16 * g) = x;
            }
            vf3 y = {v, v, v};
            *(vf3 *)(q + 288) = y;
        }
    }
}

template <int MODE, int EPW, int VALU>
static void run(const char *name, float *buf, long long N, int T, int shift = 0)
{
    dim3 block(64), grid((unsigned)(N / EPW));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<MODE, EPW, VALU>), grid, block, 0, 0, buf, N, T, shift);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, EPW, VALU>), grid, block, 0, 0, buf, N, T, shift);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = 300.0 * T * N;
    printf("%-12s EPW=%2d N=%6lld T=%3d valu=%4d shift=%d : %8.3f ms  %6.2f TB/s  %7.3f us/batch-step  %.3g env-steps/s\n", name, EPW, N,
           T, VALU, shift, ms, bytes / ms / 1e9, ms * 1e3 / T, (double)N * T / (ms * 1e-3));
}

int main()
{
    float *buf;
    hipMalloc(&buf, 300ull * 65536 * 128 + 4096);
    for (long long N : {65536ll, 32768ll, 8192ll}) {
        const int T = N >= 32768 ? 64 : 128;
        printf("---- N = %lld\n", N);
        run<COOP_X4, 64, 0>("coop_x4", buf, N, T);
        run<COOP_X4_NT, 64, 0>("coop_x4_nt", buf, N, T);
        run<COOP_X4_NT, 32, 0>("coop_x4_nt", buf, N, T);
        run<COOP_X4_NT, 8, 0>("coop_x4_nt", buf, N, T);
        run<COOP_X4_NT, 32, 700>("coop_x4_nt", buf, N, T);
        run<COOP_X4, 64, 0>("coop_x4+4", buf, N, T, 1);
        run<COOP_X4, 32, 0>("coop_x4", buf, N, T);
        run<COOP_X4, 16, 0>("coop_x4", buf, N, T);
        run<COOP_X4, 8, 0>("coop_x4", buf, N, T);
        run<COOP_X4, 8, 0>("coop_x4+4", buf, N, T, 1);
        run<COOP_X4, 4, 0>("coop_x4", buf, N, T);
        run<COOP_DWORD, 64, 0>("coop_dword", buf, N, T);
        run<COOP_DWORD, 16, 0>("coop_dword", buf, N, T);
        run<COOP_DWORD, 8, 0>("coop_dword", buf, N, T);
        run<LANE_X4, 64, 0>("lane_x4", buf, N, T);
        run<LANE_X4, 8, 0>("lane_x4", buf, N, T);
        // with arithmetic per chunk (the real kernel: ~600 (EPW = 64) ... ~1300 (EPW = 8) instructions per chunk)
        run<COOP_X4, 64, 600>("coop_x4", buf, N, T);
        run<COOP_X4, 32, 700>("coop_x4", buf, N, T);
        run<COOP_X4, 16, 900>("coop_x4", buf, N, T);
        run<COOP_X4, 8, 1300>("coop_x4", buf, N, T);
        run<COOP_X4, 4, 2000>("coop_x4", buf, N, T);
    }
    return 0;
}
