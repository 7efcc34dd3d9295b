// K1 (batched contiguous) instantiations, Complex<float>.
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_k1_f32(std::vector<KernelEntry>& reg) {
    MI_K1_LIST(float, 32);
}
}  // namespace mi355
