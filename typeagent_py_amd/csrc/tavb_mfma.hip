// placeholder until the MFMA batched kernel lands (next commit)
#include "tavb_internal.h"
namespace tavb {
hipError_t launch_mfma_scan(const MfmaParams&, hipStream_t) { return hipErrorNotSupported; }
int mfma_query_tile() { return 128; }
int mfma_pick_splits(int64_t, int, int) { return 1; }
bool mfma_supported(int, int) { return false; }
}  // namespace tavb
