// placeholder until the MFMA forward lands
#include "sfa_common.hpp"
#include "sfa_internal.hpp"
namespace sfa {
bool fwd_mfma_supported(int, int) { return false; }
int fwd_mfma(const sfa_tensor*, const sfa_tensor*, const sfa_tensor*, const sfa_tensor*, float*, const float*,
             const Problem&, hipStream_t) {
    set_error("fwd_mfma not built");
    return SFA_ERR_UNSUPPORTED;
}
}  // namespace sfa
