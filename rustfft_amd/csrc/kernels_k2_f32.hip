// K2 (column-tile passes of the large-N decomposition) instantiations, Complex<float>.
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_k2_f32(std::vector<KernelEntry>& reg) {
    MI_K2(float, 32, 64, false, 64, 8, 8, 8);
    MI_K2(float, 32, 64, false, 128, 8, 16, 8);
    MI_K2(float, 32, 32, false, 256, 16, 16, 16);
    MI_K2(float, 32, 16, false, 512, 32, 16, 8, 4);
    MI_K2(float, 32, 16, true, 1024, 32, 16, 16, 4);
}
}  // namespace mi355
