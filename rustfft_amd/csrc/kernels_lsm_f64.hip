// Complex<double> instances of the LDS stage machine (lsm.h)
#include "lsm_launch.h"
namespace mi355 {
void register_lsm_f64(std::vector<KernelEntry>& reg) {
    reg.push_back(make_lsm<double>(64, "lsm"));
}
}  // namespace mi355
