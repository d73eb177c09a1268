// Per-(device, stream, kind) scratch buffers for kernels that pass partial results through HBM (the weight gradient's
// partial tiles, the split-K convolution's partial sums). Launches on one stream are ordered, so consecutive layers
// reuse the buffer; it only grows (after draining the stream).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace dasws {
enum Kind { WGRAD = 0, CONV_SPLITK = 1, BN_FOLD = 2, SPLITK_CNT = 3, N_KINDS };   // SPLITK_CNT: zero-filled when (re)allocated (arrival
                                                                                  // counters that their kernels leave at zero)
// nullptr on allocation failure. `min_bytes`: size of the first allocation (avoids regrowth layer by layer).
float* get(Kind kind, hipStream_t s, size_t bytes, size_t min_bytes);
}  // namespace dasws
