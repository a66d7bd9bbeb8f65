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

}   // namespace maray
