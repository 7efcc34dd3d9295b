// K1 (batched contiguous) instantiations, Complex<float>.
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_k1_f32(std::vector<KernelEntry>& reg) {
    MI_K1_LIST(float, 32);
    // ablation probes of the 1024-point kernel (MI355FFT_VARIANT=5..7, wrong results by design): measured 5.36 TB/s for the
    // load/store skeleton against 5.1 - 5.2 TB/s for the full kernel.  Tuning history (no gain, removed): F = 2 / 8 rows per
    // workgroup, radix-8 schedules, 128-thread 4096 kernel, non-temporal loads/stores (tools/membench shows +11 % for an
    // in-place copy, the real kernels lose 1 - 3 %).
    MI_K1ABL(5, 12, float, 32, 4, false, 1024, 64, 16, 16, 4);  // loads + stores only
    MI_K1ABL(6, 8, float, 32, 4, false, 1024, 64, 16, 16, 4);   // arithmetic without the exchanges
    MI_K1ABL(7, 4, float, 32, 4, false, 1024, 64, 16, 16, 4);   // exchanges without the arithmetic
}
}  // namespace mi355
