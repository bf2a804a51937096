// Fused persistent FISTA kernel -- placeholder until the bf16 path lands.
#include "fc_fused.h"

namespace vtc {
bool fused_shape_supported(int64_t, int64_t, int64_t, int) { return false; }
size_t fused_workspace_bytes(int64_t, int64_t, int64_t, int) { return 256; }
int run_fused(const float*, const float*, const float*, float*, int64_t,
              int64_t, int64_t, float, float, int, int, int, int, void*,
              size_t, int*, hipStream_t) {
  set_error("fused FISTA kernel not built");
  return VTC_ERR_UNSUPPORTED;
}
}  // namespace vtc
