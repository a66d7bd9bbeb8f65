// host_pipe.hpp — device -> host leg of the render entry points that take host rasters (product code).
//
// Replaces the collector of the reference's renderers (src/render.rs:54-83: a thread that receives
// finished rows over a channel and copies them pixel by pixel into the caller's RgbImage): here the
// rows of a tile leave the device by DMA, tile k's copy running under tile k+1's kernels.
//
//   compute stream:  [ROW + PIXEL kernels, tile k] -> event -> [tile k+1] ...
//   copy stream   :            wait event -> [D2H tile k] -> event -> ...
//   host thread   :  stays two tiles ahead of the copy engine; joins at the end (or per tile when the
//                    caller wants progress callbacks)
//
// Destination in pinned memory (maray_host_alloc, or registered by the caller): the DMA writes the
// caller's raster directly -- PCIe rate.  Pageable destination: the DMA lands in a pinned ring of
// the context and the calling thread copies tile k out of it while tile k+1 is in flight.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

#include "backend.hpp"

namespace maray {

class HostPipe {
public:
    // launch(rb, d8, d64, stream): enqueue the kernels that render the rows of `rb` into device buffers
    using LaunchFn = std::function<void(const RowBlocks &, unsigned char *, double *, hipStream_t)>;
    using DoneFn = std::function<void(uint32_t, uint32_t)>;       // rows [y0, y1) are in the host raster

    HostPipe() = default;
    HostPipe(const HostPipe &) = delete;
    HostPipe &operator=(const HostPipe &) = delete;
    ~HostPipe();
    void init(int device);
    hipStream_t compute_stream() const { return compute; }
    int device_index() const { return device; }
    // what the first run() of a w-pixel-wide image with tiles of `tile_rows` rows would create: the copy stream and the
    // device slots (RGB8).  Done ahead of time by host_pipe_prewarm, on another thread than the one that lowers the scene.
    void reserve(uint32_t w, uint32_t tile_rows, int n_slots);
    void trim();        // frees the f64 slots and the pinned ring (8x and 1x an RGB8 slot): what an idle pipe in the pool does not keep

    // Renders `tiles` (row ranges of a w-pixel-wide image) in order.  rgb8 / rgb64 point at the
    // first byte of image row `row0` of the caller's rasters (either may be null).
    void run(uint32_t w, const std::vector<RowTile> &tiles, uint32_t row0, uint8_t *rgb8, double *rgb64,
             const LaunchFn &launch, const DoneFn &done);

private:
    static constexpr int SLOTS = 3;
    int device = 0;
    hipStream_t compute = nullptr, copy = nullptr;
    hipEvent_t kernels_done[SLOTS] = {}, copy_done[SLOTS] = {};
    unsigned char *d8[SLOTS] = {};
    double *d64[SLOTS] = {};
    size_t d8_cap[SLOTS] = {}, d64_cap[SLOTS] = {};
    void *ring[SLOTS] = {};             // pinned staging, pageable destinations only
    size_t ring_cap[SLOTS] = {};
};

// Pipes outlive contexts: a pipe's streams, events and device slots do not depend on the program, and creating them is a
// third of what a context costs (3.6 ms of streams and events at creation, a copy stream and two or three hipMalloc of
// 24 MiB on the first frame).  A context takes its pipe from a per-device pool and gives it back when it is freed;
// host_pipe_prewarm fills the pool from another thread while the caller is still lowering its scene (gen.cpp).
std::unique_ptr<HostPipe> host_pipe_acquire(int device);             // throws Error
void host_pipe_release(std::unique_ptr<HostPipe> p) noexcept;         // (at most 4 idle pipes per device are kept)
void host_pipe_prewarm(int device, uint32_t w, uint32_t tile_rows, int n_slots) noexcept;      // errors are left for the context to find

}   // namespace maray
