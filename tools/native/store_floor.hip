// store_floor.hip — what a 4096^2 RGB8 raster (50.3 MB) costs to WRITE on this GPU, nothing computed: the floor under the
// sky tiles of chess and under config 2.  hipcc --offload-arch=gfx950 -O3 store_floor.hip -o store_floor
//   x4      16 B per lane, consecutive lanes consecutive addresses, one store per lane per iteration, grid-stride
//   x3      12 B per lane (4 pixels: the specialised kernel's WIDE store), rows of 256 px per wave like its tiles
//   x3tile  the same, but each wave walks 32 rows of its own 256-px column strip (12 KiB row pitch between its stores)
//   memset  hipMemsetAsync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct u3 { unsigned a, b, c; };

__global__ void __launch_bounds__(256) fill_x4(uint4 *p, size_t n16)
{
    const uint4 v = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// one wave per 256-px segment of a row, rows x segments in order
__global__ void __launch_bounds__(256) fill_x3(unsigned char *p, unsigned w, unsigned h)
{
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned segs = w / 256, nw = (gridDim.x * blockDim.x) >> 6;
    const u3 v = {0, 0, 0};
    for (size_t s = wave; s < (size_t)segs * h; s += nw) {
        const size_t y = s / segs, x0 = (s % segs) * 256;
        *(u3 *)(p + (y * w + x0 + 4 * lane) * 3) = v;
    }
}

// one wave per tile of 256 px x 32 rows
__global__ void __launch_bounds__(256) fill_x3tile(unsigned char *p, unsigned w, unsigned h)
{
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned segs = w / 256, nw = (gridDim.x * blockDim.x) >> 6;
    const u3 v = {0, 0, 0};
    for (size_t t = wave; t < (size_t)segs * (h / 32); t += nw) {
        const size_t ty = t / segs, x0 = (t % segs) * 256;
        for (unsigned r = 0; r < 32; r++) *(u3 *)(p + ((ty * 32 + r) * w + x0 + 4 * lane) * 3) = v;
    }
}

template <class F> static double time_us(F f, int reps = 200)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 20; i++) f();
    std::vector<float> ms;
    for (int k = 0; k < 5; k++) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; i++) f();
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t / reps * 1000.0f);
    }
    std::sort(ms.begin(), ms.end());
    return ms[2];
}

int main()
{
    const unsigned w = 4096, h = 4096; const size_t bytes = (size_t)w * h * 3;
    unsigned char *p; CK(hipMalloc(&p, bytes));
    printf("{\"bytes\": %zu", bytes);
    for (unsigned blocks : {512u, 1024u, 2048u, 4096u, 8192u, 16384u}) {
        double a = time_us([&] { fill_x4<<<blocks, 256>>>((uint4 *)p, bytes / 16); });
        double b = time_us([&] { fill_x3<<<blocks, 256>>>(p, w, h); });
        double c = time_us([&] { fill_x3tile<<<blocks, 256>>>(p, w, h); });
        printf(", \"blocks_%u\": {\"x4_us\": %.2f, \"x3_us\": %.2f, \"x3tile_us\": %.2f}", blocks, a, b, c);
    }
    double m = time_us([&] { CK(hipMemsetAsync(p, 0, bytes, 0)); });
    double e = time_us([&] { fill_x4<<<1, 64>>>((uint4 *)p, 64); });
    printf(", \"memset_us\": %.2f, \"empty_launch_us\": %.2f}\n", m, e);
    return 0;
}
