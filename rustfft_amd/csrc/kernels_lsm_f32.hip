// Complex<float> instances of the LDS stage machine (lsm.h): run-time programmed MixedRadix / Rader trees in one workgroup
#include "lsm_launch.h"
namespace mi355 {
void register_lsm_f32(std::vector<KernelEntry>& reg) {
    reg.push_back(make_lsm<float>(32, "lsm"));
}
}  // namespace mi355
