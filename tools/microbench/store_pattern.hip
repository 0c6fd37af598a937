// Micro-benchmark: the MultiSnake 'full' observation layout (T, K, N, 3*S*S) written by one wave per env — which order
// of the 120 dword stores per env-step streams fastest?  K = 4 agents, S = 25 (C = 625 cells, planes of 2500 B, agent
// regions of 7500 B that are N * 7500 B apart), N = 4096 envs, T = 16 steps.
//   rows   : for row k: for agent a: planes R,G,B       (round 1 order: 12 interleaved streams per wave)
//   agents : for agent a: for row k: planes R,G,B       (3 interleaved streams per wave)
//   planes : for agent a: for plane: for row k          (1 sequential stream per wave)
//   x4     : agent-major, lane owns 4 consecutive floats of the agent's 7500-byte region (global_store_dwordx4,
//            region starts only 4-byte aligned)
// build: hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float vf4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int T, long long N)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const long long env = (long long)blockIdx.x * wpb + wave;
    if (env >= N) return;
    const int K = 4, C = 625, E = 3 * C, cpl = 10;
    float v = (float)lane;
    for (int t = 0; t < T; ++t) {
        float *o = out + ((long long)t * K * N + env) * E;   // agent 0
        const long long as = N * E;
        if (MODE == 0) {
            for (int kk = 0; kk < cpl; ++kk) { int c = lane + 64 * kk; if (c < C) for (int a = 0; a < K; ++a) { float *b = o + a * as; b[c] = v; b[C + c] = v; b[2 * C + c] = v; } }
        } else if (MODE == 1) {
            for (int a = 0; a < K; ++a) { float *b = o + a * as; for (int kk = 0; kk < cpl; ++kk) { int c = lane + 64 * kk; if (c < C) { b[c] = v; b[C + c] = v; b[2 * C + c] = v; } } }
        } else if (MODE == 2) {
            for (int a = 0; a < K; ++a) { float *b = o + a * as; for (int i = lane; i < E; i += 64) b[i] = v; }
        } else if (MODE == 3) {
            for (int a = 0; a < K; ++a) { float *b = o + a * as; for (int i = 4 * lane; i < E; i += 256) { if (i + 3 < E) *(vf4 *)(b + i) = vf4{v, v, v, v}; else for (int j = i; j < E; ++j) b[j] = v; } }
        } else if (MODE == 4) {
            // x4 on absolute 16-byte boundaries: head floats up to the boundary and tail floats as dword stores
            for (int a = 0; a < K; ++a) {
                float *b = o + a * as;
                const int h = (int)(((16 - ((size_t)b & 15)) & 15) >> 2);       // floats before the first boundary
                if (lane < h) b[lane] = v;
                const int n4 = (E - h) >> 2;
                for (int i = lane; i < n4; i += 64) *(vf4 *)(b + h + 4 * i) = vf4{v, v, v, v};
                const int t0 = h + 4 * n4;
                if (t0 + lane < E) b[t0 + lane] = v;
            }
        } else {
            // 4 waves of a workgroup = 4 consecutive envs: wave w writes AGENT w's regions of all four envs (30 000 contiguous,
            // 16-byte aligned bytes), x4
            const long long env0 = (long long)blockIdx.x * 4;
            float *b = out + (((long long)t * K + wave) * N + env0) * E;
            for (int i = lane; i < 4 * E / 4; i += 64) *(vf4 *)(b + 4 * i) = vf4{v, v, v, v};
        }
    }
}

template <int MODE>
static void run(const char *name, float *buf, int T, long long N, int wpb)
{
    dim3 block(64 * wpb), grid((unsigned)((N + wpb - 1) / wpb));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, buf, T, N);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, buf, T, N);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-8s T=%d wpb=%d : %7.3f ms  %5.2f TB/s\n", name, T, wpb, ms, 4.0 * 7500 * T * N / ms / 1e9);
}

int main()
{
    const long long N = 4096; const int T = 16;
    float *buf; hipMalloc(&buf, 4ull * 7500 * 64 * N);
    for (int wpb : {1, 4}) {
        run<0>("rows", buf, T, N, wpb);
        run<1>("agents", buf, T, N, wpb);
        run<2>("planes", buf, T, N, wpb);
        run<3>("x4", buf, T, N, wpb);
        run<4>("x4peel", buf, T, N, wpb);
    }
    run<5>("x4by4env", buf, T, N, 4);
    run<5>("x4by4env", buf, 64, N, 4);
    run<4>("x4peel", buf, 64, N, 4);
    run<1>("agents", buf, 64, N, 4);
    return 0;
}
