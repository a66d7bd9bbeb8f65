// backend.hpp — device evaluator interface behind the tape-level ABI (product code).
#pragma once

#include <cstdint>
#include <string>

#include "expr.hpp"
#include "maray_hip.h"

namespace maray {

struct Backend {
    virtual ~Backend() {}
    // rows [y0,y1) into device buffers, enqueued on `stream`, no synchronisation
    virtual void render_device(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, void *d8, void *d64, void *stream) = 0;
    // rows [y0,y1) into host buffers (blocking)
    virtual void render_host(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint8_t *rgb8, double *rgb64) = 0;
    // average ms per launch of the pixel kernel, HIP events on the launch stream
    virtual float time_rows(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, void *d8, void *d64, int reps) = 0;
    virtual const char *kernel_name() const = 0;
};

void set_last_error(const std::string &m);   // thread-local message behind maray_last_error()
int hip_device_count();
Backend *make_tape_backend(int device, const maray_program &prog, const maray_texture *tex, uint32_t n_tex, bool lds_variant);
Backend *make_jit_backend(int device, const maray_program &prog, const maray_texture *tex, uint32_t n_tex);

}   // namespace maray

struct maray_ctx {
    maray::Backend *backend = nullptr;
    uint32_t n_tex = 0;
};
