// binning_bucket.hip -- tile-bucket binning back-end (GSR_BINNING_TILE_BUCKET).  Not built yet in this round:
// the global radix back-end is the default and the only one enabled.
#include "gsr_common.h"

namespace gsr {
int bucket_binning(const GeomState &, const int *, int, int, int, size_t, BinningState &, uint2 *, hipStream_t, int) {
  set_error("GSR_BINNING_TILE_BUCKET is not available in this build");
  return GSR_EINVAL;
}
}  // namespace gsr
