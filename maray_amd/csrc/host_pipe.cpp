// host_pipe.cpp — see host_pipe.hpp (product code).
#include <cstdlib>
#include "host_pipe.hpp"

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace maray {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            throw Error{MARAY_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)};           \
    } while (0)

bool host_range_is_pinned(const void *p, size_t n)
{
    if (!p || !n) return false;
    const void *ends[2] = {p, (const char *)p + n - 1};
    for (const void *q : ends) {
        hipPointerAttribute_t a;
        memset(&a, 0, sizeof a);
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }    // older runtimes: an error for pageable memory
        if (a.type != hipMemoryTypeHost) return false;                                                  // newer ones: hipMemoryTypeUnregistered
    }
    return true;
}

void *host_alloc_pinned(size_t bytes)
{
    void *p = nullptr;
    // portable: every device of the process may DMA into it (a raster shared by the workers of a multi-GPU render)
    HIP_TRY(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable));
    return p;
}

void host_free_pinned(void *p)
{
    if (p) (void)hipHostFree(p);
}

void host_register(void *p, size_t bytes) { HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable)); }
void host_unregister(void *p) { HIP_TRY(hipHostUnregister(p)); }

HostPipe::~HostPipe()
{
    if (!compute && !copy) return;
    (void)hipSetDevice(device);
    for (int s = 0; s < SLOTS; s++) {
        if (kernels_done[s]) (void)hipEventDestroy(kernels_done[s]);
        if (copy_done[s]) (void)hipEventDestroy(copy_done[s]);
        (void)hipFree(d8[s]);
        (void)hipFree(d64[s]);
        if (ring[s]) (void)hipHostFree(ring[s]);
    }
    if (compute) (void)hipStreamDestroy(compute);
    if (copy) (void)hipStreamDestroy(copy);
}

void HostPipe::init(int dev)
{
    device = dev;
    HIP_TRY(hipSetDevice(dev));
    HIP_TRY(hipStreamCreateWithFlags(&compute, hipStreamNonBlocking));
    // (the copy stream is created by the first run(): a stream costs a context several milliseconds, and one that only
    // ever renders into device buffers never copies)
    for (int s = 0; s < SLOTS; s++) {
        HIP_TRY(hipEventCreateWithFlags(&kernels_done[s], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&copy_done[s], hipEventDisableTiming));
    }
}

void HostPipe::reserve(uint32_t w, uint32_t tile_rows, int n_slots)
{
    HIP_TRY(hipSetDevice(device));
    if (!copy) HIP_TRY(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    const size_t bytes = (size_t)w * tile_rows * 3;
    for (int s = 0; s < std::min(n_slots, (int)SLOTS); s++) {
        if (d8_cap[s] >= bytes) continue;
        if (d8[s]) HIP_TRY(hipFree(d8[s]));
        d8[s] = nullptr; d8_cap[s] = 0;
        HIP_TRY(hipMalloc((void **)&d8[s], bytes));
        d8_cap[s] = bytes;
    }
}

void HostPipe::trim()
{
    (void)hipSetDevice(device);
    for (int s = 0; s < SLOTS; s++) {
        if (d64[s]) { (void)hipFree(d64[s]); d64[s] = nullptr; d64_cap[s] = 0; }
        if (ring[s]) { (void)hipHostFree(ring[s]); ring[s] = nullptr; ring_cap[s] = 0; }
        if (d8_cap[s] > ((size_t)64 << 20)) { (void)hipFree(d8[s]); d8[s] = nullptr; d8_cap[s] = 0; }      // (an unusually large tile)
    }
}

namespace {
std::mutex g_pool_mutex;
// never destroyed: at process exit the HIP runtime may be gone before a static's destructor runs
std::multimap<int, std::unique_ptr<HostPipe>> &g_pool = *new std::multimap<int, std::unique_ptr<HostPipe>>();
}   // namespace

std::unique_ptr<HostPipe> host_pipe_acquire(int device)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        auto it = g_pool.find(device);
        if (it != g_pool.end()) { std::unique_ptr<HostPipe> p = std::move(it->second); g_pool.erase(it); return p; }
    }
    std::unique_ptr<HostPipe> p(new HostPipe());
    p->init(device);
    return p;
}

void host_pipe_release(std::unique_ptr<HostPipe> p) noexcept
{
    if (!p) return;
    p->trim();
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        if (g_pool.count(p->device_index()) < 4) { const int d = p->device_index(); g_pool.emplace(d, std::move(p)); return; }
    }
    p.reset();          // (outside the lock: streams are destroyed)
}

void host_pipe_prewarm(int device, uint32_t w, uint32_t tile_rows, int n_slots) noexcept
{
    try {
        std::unique_ptr<HostPipe> p = host_pipe_acquire(device);
        try { p->reserve(w, tile_rows, n_slots); } catch (const Error &) {}
        host_pipe_release(std::move(p));
    } catch (...) {}
}

std::vector<RowTile> cut_row_tiles(uint32_t w, uint32_t y0, uint32_t y1, bool want8, bool want64)
{
    // ~24 MiB per tile: the DMA engine reaches its rate (57 GB/s measured) on copies of this size and every copy has a fixed
    // start-up cost, yet the first copy still starts early and only the first tile's kernels are exposed (chess @4096^2 into a
    // pinned raster, ms per frame at 4 / 8 / 16 / 24 / 48 MiB: specialised kernels 1.17 / 1.04 / 0.99 / 0.96 / 0.96,
    // interpreter 4.29 / 2.95 / 2.47 / 2.35 / 2.64).  A multiple of 32 rows: the specialised kernels evaluate their guards
    // per group of 32 rows of a launch.
    const uint64_t row_bytes = (uint64_t)w * ((want8 ? 3 : 0) + (want64 ? 24 : 0));
    uint64_t tile_mib = 24;
    if (const char *e_ = getenv("MARAY_TILE_MIB")) if (atoi(e_) > 0) tile_mib = (uint64_t)atoi(e_);      // measurement knob
    uint64_t rows = row_bytes ? (tile_mib << 20) / row_bytes : (y1 - y0);
    rows = rows >= 32 ? rows / 32 * 32 : std::max<uint64_t>(8, rows / 8 * 8);
    std::vector<RowTile> t;
    for (uint64_t y = y0; y < y1; y += rows) t.push_back(RowTile{(uint32_t)y, (uint32_t)std::min<uint64_t>(y1, y + rows)});
    return t;
}

void HostPipe::run(uint32_t w, const std::vector<RowTile> &tiles, uint32_t row0, uint8_t *rgb8, double *rgb64,
                   const LaunchFn &launch, const DoneFn &done)
{
    if (tiles.empty() || !w || (!rgb8 && !rgb64)) return;
    HIP_TRY(hipSetDevice(device));
    if (!copy) HIP_TRY(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    size_t max_px = 0;
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (const RowTile &t : tiles) {
        max_px = std::max(max_px, (size_t)(t.y1 - t.y0) * w);
        lo = std::min(lo, t.y0); hi = std::max(hi, t.y1);
    }
    if (lo < row0) throw Error{MARAY_E_ARG, "tile above the first row of the raster"};
    // is the part of each raster that these tiles touch pinned?  (the caller may have registered just its slice)
    const size_t span_off = (size_t)(lo - row0) * w * 3, span_n = (size_t)(hi - lo) * w * 3;
    const bool direct8 = rgb8 && host_range_is_pinned(rgb8 + span_off, span_n);
    const bool direct64 = rgb64 && host_range_is_pinned(rgb64 + span_off, span_n * 8);
    const size_t ring_need = ((rgb8 && !direct8) ? max_px * 3 : 0) + ((rgb64 && !direct64) ? max_px * 24 : 0);
    const int n_slots = (int)std::min<size_t>(SLOTS, tiles.size());
    for (int s = 0; s < n_slots; s++) {
        if (rgb8 && d8_cap[s] < max_px * 3) {
            if (d8[s]) HIP_TRY(hipFree(d8[s]));
            d8[s] = nullptr; d8_cap[s] = 0;
            HIP_TRY(hipMalloc((void **)&d8[s], max_px * 3));
            d8_cap[s] = max_px * 3;
        }
        if (rgb64 && d64_cap[s] < max_px * 3) {
            if (d64[s]) HIP_TRY(hipFree(d64[s]));
            d64[s] = nullptr; d64_cap[s] = 0;
            HIP_TRY(hipMalloc((void **)&d64[s], max_px * 24));
            d64_cap[s] = max_px * 3;
        }
        if (ring_need > ring_cap[s]) {
            if (ring[s]) HIP_TRY(hipHostFree(ring[s]));
            ring[s] = nullptr; ring_cap[s] = 0;
            HIP_TRY(hipHostMalloc(&ring[s], ring_need, hipHostMallocDefault));
            ring_cap[s] = ring_need;
        }
    }

    size_t drained = 0;         // tiles whose rows are in the caller's raster
    auto drain_one = [&]() {
        const size_t k = drained;
        const int s = (int)(k % SLOTS);
        const RowTile &t = tiles[k];
        HIP_TRY(hipEventSynchronize(copy_done[s]));
        const size_t px = (size_t)(t.y1 - t.y0) * w, off = (size_t)(t.y0 - row0) * w * 3;
        if (rgb8 && !direct8) memcpy(rgb8 + off, ring[s], px * 3);
        if (rgb64 && !direct64) memcpy(rgb64 + off, (const char *)ring[s] + ((rgb8 && !direct8) ? max_px * 3 : 0), px * 24);
        drained++;
        if (done) done(t.y0, t.y1);
    };
    try {
        for (size_t k = 0; k < tiles.size(); k++) {
            // slot k % SLOTS was last used by tile k - SLOTS: the host keeps at most SLOTS - 1 tiles in flight
            while (drained + (SLOTS - 1) <= k) drain_one();
            const int s = (int)(k % SLOTS);
            const RowTile &t = tiles[k];
            const size_t px = (size_t)(t.y1 - t.y0) * w, off = (size_t)(t.y0 - row0) * w * 3;
            launch(RowBlocks::range(t.y0, t.y1), rgb8 ? d8[s] : nullptr, rgb64 ? d64[s] : nullptr, compute);
            HIP_TRY(hipEventRecord(kernels_done[s], compute));
            HIP_TRY(hipStreamWaitEvent(copy, kernels_done[s], 0));
            if (rgb8) HIP_TRY(hipMemcpyAsync(direct8 ? (void *)(rgb8 + off) : ring[s], d8[s], px * 3, hipMemcpyDeviceToHost, copy));
            if (rgb64)
                HIP_TRY(hipMemcpyAsync(direct64 ? (void *)(rgb64 + off) : (void *)((char *)ring[s] + ((rgb8 && !direct8) ? max_px * 3 : 0)),
                                       d64[s], px * 24, hipMemcpyDeviceToHost, copy));
            HIP_TRY(hipEventRecord(copy_done[s], copy));
        }
        while (drained < tiles.size()) drain_one();
    } catch (...) {
        // nothing may still be writing the caller's raster or the ring when this call returns
        (void)hipStreamSynchronize(compute);
        if (copy) (void)hipStreamSynchronize(copy);
        throw;
    }
}

}   // namespace maray
