// rrt_resident.hip -- register-resident RRT grow kernel (placeholder until the kernel lands).
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"

namespace oxhip {
bool resident_supported(uint32_t, uint32_t) { return false; }
void launch_rrt_resident(const DevParams&, hipStream_t) {}
}  // namespace oxhip
