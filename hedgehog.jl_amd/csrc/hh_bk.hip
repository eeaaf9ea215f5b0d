// Broadie–Kaya exact Heston sampler — placeholder until the kernel lands.
#include "hh_kernels.h"
namespace hh {
int launch_bk(const hh_model&, const hh_config&, const DevicePtrs&, hipStream_t) {
  return (int)hipErrorNotSupported;
}
}  // namespace hh
