// checks the wave reductions of wurm_device.hpp (wave_max / min / sum_i32) against a host loop
#include "../../wurm_amd/csrc/wurm_device.hpp"
#include <cstdio>
#include <cstdlib>
using namespace wurm;
__global__ void k(const int *in, int *out) {
    int v = in[blockIdx.x * 64 + threadIdx.x];
    int a = wave_max_i32(v), b = wave_min_i32(v), c = wave_sum_i32(v);
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = a; out[blockIdx.x * 3 + 1] = b; out[blockIdx.x * 3 + 2] = c; }
}
int main() {
    const int B = 4096;
    int *h = (int *)malloc(B * 64 * 4), *ho = (int *)malloc(B * 12), *d, *dout;
    srand(1);
    for (int i = 0; i < B * 64; ++i) h[i] = (rand() % 2000001) - 1000000;
    hipMalloc(&d, B * 64 * 4); hipMalloc(&dout, B * 12);
    hipMemcpy(d, h, B * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, d, dout);
    hipMemcpy(ho, dout, B * 12, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < B; ++b) {
        int mx = h[b * 64], mn = h[b * 64], s = 0;
        for (int i = 0; i < 64; ++i) { int x = h[b * 64 + i]; mx = x > mx ? x : mx; mn = x < mn ? x : mn; s += x; }
        if (ho[b * 3] != mx || ho[b * 3 + 1] != mn || ho[b * 3 + 2] != s) ++bad;
    }
    printf("reductions: %d of %d blocks wrong\n", bad, B);
    return bad != 0;
}
