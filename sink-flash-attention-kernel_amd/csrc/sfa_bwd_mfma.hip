// placeholder until the MFMA backward lands
#include "sfa_common.hpp"
#include "sfa_internal.hpp"
namespace sfa {
bool bwd_mfma_supported(int, int) { return false; }
size_t bwd_mfma_workspace_bytes(const Problem&, int) { return 0; }
int bwd_mfma(const sfa_tensor*, const sfa_tensor*, const sfa_tensor*, const sfa_tensor*, const float*, const float*,
             const sfa_tensor*, const sfa_tensor*, const sfa_tensor*, void*, const Problem&, hipStream_t) {
    set_error("bwd_mfma not built");
    return SFA_ERR_UNSUPPORTED;
}
}  // namespace sfa
