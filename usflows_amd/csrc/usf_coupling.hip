// usf_coupling_additive_f32 -- placeholder until the fused kernel lands (next commit).
#include "usf_common.h"

namespace usf {
int coupling_max_width() { return 0; }
int coupling_dispatch(const usf_coupling_desc*, hipStream_t) {
  set_error("usf_coupling_additive_f32: fused coupling kernel not built; use usf_linear_f32 chain");
  return -38;
}
}  // namespace usf
