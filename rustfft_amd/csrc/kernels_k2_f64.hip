// K2 (column-tile passes of the large-N decomposition) instantiations, Complex<double>.
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_k2_f64(std::vector<KernelEntry>& reg) {
    MI_K2(double, 64, 32, false, 64, 8, 8, 8);
    MI_K2(double, 64, 32, false, 128, 8, 16, 8);
    MI_K2(double, 64, 16, false, 256, 16, 16, 16);
    MI_K2(double, 64, 8, false, 512, 32, 16, 8, 4);
    // (16-column split tile for 512 rows: measured 5 % slower in f64, not instantiated)
    MI_K2(double, 64, 8, true, 1024, 32, 16, 16, 4);
}
}  // namespace mi355
