// Dense f32-MFMA path for wide hidden layers (placeholder: not yet enabled).
#include <hip/hip_runtime.h>
#include "kan_launch.h"

namespace pdekan {

bool mfma_eligible(int, int, const LayerView&) { return false; }
int mfma_weight_chunks(int64_t, const LayerView&) { return 1; }
void launch_fwd_mfma(const LayerView&, const float*, const float*, float*, int64_t, hipStream_t) {}
void launch_bwd_input_mfma(const LayerView&, const float*, const float*, const float*, float*, int64_t, hipStream_t) {}
void launch_bwd_weight_mfma(const LayerView&, const float*, const float*, float*, int, int64_t, hipStream_t) {}

}  // namespace pdekan
