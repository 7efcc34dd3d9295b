// Non-power-of-two instantiations, Complex<float>: native mixed radix for BASELINE config 3 (N = 1200),
// Rader for config 4 (N = 1009, inner 1008 = 16 x 9 x 7) and the Bluestein bodies that cover every other length.
#include "launch.h"
#include "kernel_lists.h"
namespace mi355 {
void register_np2_f32(std::vector<KernelEntry>& reg) {
    MI_K1(float, 32, 2, false, 1200, 120, 10, 10, 12);
    MI_RADER(float, 32, 1, 1008, 144, 16, 9, 7);  // one row per workgroup: more independent workgroups per CU (+11 % over two rows)
    MI_RADERV(1, float, 32, 2, 1008, 144, 16, 9, 7);
    MI_RADERV(2, float, 32, 2, 1008, 63, 16, 9, 7);  // tuning: one wave per row, up to 21 values per thread (slower)
    MI_BS_LIST(float, 32);
    MI_BS_LIST3_F32(float, 32);
    reg.push_back(make_pointwise<float>(32));
    reg.push_back(make_dyn_k1<float>(32));
    reg.push_back(make_dyn_rader<float>(32));
}
}  // namespace mi355
