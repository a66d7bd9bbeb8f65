// sky_cost.hip — what a wavefront of sky costs as a function of its kernel's footprint: the specialised
// kernel's four-wide sky variant restated alone (a strip of 2 tiles of 256 pixels per wavefront, 12 B per lane and tile, 4
// wavefronts per block, 4096 x 4096 RGB8) in kernels that differ in nothing but what they RESERVE: LDS per block, the highest
// VGPR and the highest SGPR they name.  hipcc --offload-arch=gfx950 -O3 sky_cost.hip -o sky_cost
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct u3 { unsigned a, b, c; };

template <int LDS, int VGPR, int SGPR, int WAVES>
__global__ void __launch_bounds__(256, WAVES) sky(unsigned char *p, unsigned w, unsigned n_tx, unsigned tiles, const unsigned *order)
{
    __shared__ unsigned pad[LDS ? LDS / 4 : 1];
    if (LDS && order[0] == 0xffffffffu) pad[threadIdx.x] = 1u;           // (never true: keeps the allocation)
    if (VGPR == 40) asm volatile("" ::: "v39");
    if (VGPR == 64) asm volatile("" ::: "v63");
    if (SGPR == 80) asm volatile("" ::: "s77");
    const unsigned wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const unsigned tile0 = (blockIdx.x * 4u + wv) * tiles;
    if (tile0 >= n_tx) return;
    const unsigned r = order[blockIdx.y];
    const u3 v = {r, 0, 0};
    for (unsigned t = 0; t < tiles; t++) {
        const unsigned x0 = (tile0 + t) * 256u;
        if (x0 >= w) break;
        *(u3 *)(p + ((size_t)r * w + x0 + 4u * lane) * 3) = v;
    }
    if (LDS && order[0] == 0xffffffffu) p[0] = (unsigned char)pad[lane];
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

template <int LDS, int VGPR, int SGPR, int WAVES> static void run(unsigned char *p, const unsigned *order, bool first)
{
    const unsigned w = 4096, h = 4096, n_tx = 16;
    printf("%s\n {\"lds\": %d, \"vgpr\": %d, \"sgpr\": %d, \"waves\": %d", first ? "" : ",", LDS, VGPR, SGPR, WAVES);
    for (unsigned tiles : {1u, 2u, 4u}) {
        const unsigned gx = (n_tx + 4 * tiles - 1) / (4 * tiles);
        const double us = time_us([&] { sky<LDS, VGPR, SGPR, WAVES><<<dim3(gx, h), 256>>>(p, w, n_tx, tiles, order); });
        printf(", \"tiles_%u_us\": %.2f", tiles, us);
    }
    printf("}");
}

int main()
{
    const unsigned w = 4096, h = 4096; const size_t bytes = (size_t)w * h * 3;
    unsigned char *p; CK(hipMalloc(&p, bytes));
    std::vector<unsigned> ord(h);
    for (unsigned i = 0; i < h; i++) ord[i] = i;
    unsigned *d_ord; CK(hipMalloc(&d_ord, h * 4)); CK(hipMemcpy(d_ord, ord.data(), h * 4, hipMemcpyHostToDevice));
    printf("{\"bytes\": %zu, \"kernels\": [", bytes);
    run<0, 12, 16, 8>(p, d_ord, true);
    run<4096, 12, 16, 8>(p, d_ord, false);
    run<0, 40, 16, 8>(p, d_ord, false);
    run<0, 12, 80, 8>(p, d_ord, false);
    run<4096, 40, 80, 8>(p, d_ord, false);
    run<0, 64, 80, 8>(p, d_ord, false);
    run<16384, 12, 16, 8>(p, d_ord, false);
    run<4096, 40, 80, 6>(p, d_ord, false);
    printf("\n]}\n");
    return 0;
}
