// jit_backend.cpp — hiprtc-specialised evaluator (product code).  Placeholder
// until the specialising back-end lands: creation fails loudly.
#include "backend.hpp"

namespace maray {

Backend *make_jit_backend(int, const maray_program &, const maray_texture *, uint32_t)
{
    throw Error{MARAY_E_ARG, "MARAY_BACKEND_JIT is not built yet"};
}

}   // namespace maray
