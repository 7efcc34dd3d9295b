// K1 (batched contiguous) instantiations, Complex<double>.
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_k1_f64(std::vector<KernelEntry>& reg) {
    MI_K1_LIST(double, 64);
}
}  // namespace mi355
