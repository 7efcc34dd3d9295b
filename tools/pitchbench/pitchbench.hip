// Does the row pitch of a column tile matter to HBM?  A column-tile pass reads R row segments of SEG bytes at a fixed pitch per
// workgroup -- a power of two for power-of-two transforms (8 KiB at 2^20, 16 KiB at 2^22).  If the channel / bank selection of
// the memory system uses low address bits, all rows of a tile queue on few channels.  This probe reads (and separately writes)
// tiles of R x SEG bytes at pitch = 2^k and at 2^k + skew and prints GB/s: workgroups of THREADS lanes, lanes walk across the
// segment first, each thread keeps R * SEG / (8 * THREADS) loads of 8 bytes in flight (the FFT tiles' pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2 __attribute__((ext_vector_type(2)));
template <int THREADS, int PER, bool WRITE>
__global__ __launch_bounds__(THREADS) void tile_rw(v2* buf, size_t pitch_e, unsigned tiles_per_row, unsigned seg_e, size_t row_block_e, float* sink) {
    // tile index -> (block of R rows, column segment)
    const unsigned t = blockIdx.x, colseg = t % tiles_per_row, blk = t / tiles_per_row;
    v2* base = buf + (size_t)blk * row_block_e + (size_t)colseg * seg_e;
    const unsigned lane_col = threadIdx.x % seg_e, r0 = threadIdx.x / seg_e, rstep = THREADS / seg_e;
    v2 acc = {0.f, 0.f};
    v2 v[PER];
    if (!WRITE) {
#pragma unroll
        for (int k = 0; k < PER; ++k) v[k] = base[(size_t)(r0 + k * rstep) * pitch_e + lane_col];
#pragma unroll
        for (int k = 0; k < PER; ++k) acc += v[k];
        if (acc.x == 12345.678f) sink[0] = acc.y;
    } else {
#pragma unroll
        for (int k = 0; k < PER; ++k) base[(size_t)(r0 + k * rstep) * pitch_e + lane_col] = v2{(float)k, (float)threadIdx.x};
    }
}
int main() {
    const size_t total = (size_t)6 << 30;  // buffer (the skewed pitches need slack)
    v2* buf;
    float* sink;
    CHECK(hipMalloc(&buf, total));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 0, total));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-6s %-8s %-10s %-8s %10s\n", "op", "rows", "pitch_B", "seg_B", "GB/s");
    for (int write = 0; write < 2; ++write)
        for (int rows : {1024, 2048})
            for (size_t pitch : {(size_t)8192, (size_t)8192 + 128, (size_t)8192 + 256, (size_t)8192 + 512, (size_t)8192 + 1024, (size_t)16384, (size_t)16384 + 128, (size_t)16384 + 256, (size_t)16384 + 512,
                                 (size_t)32768, (size_t)32768 + 256, (size_t)65536, (size_t)65536 + 256, (size_t)(1 << 19), (size_t)(1 << 19) + 256}) {
                const unsigned seg_b = 128, seg_e = seg_b / 8;
                const size_t pitch_e = pitch / 8;
                const unsigned tiles_per_row = (unsigned)((pitch < 16384 ? 8192 : pitch < 32768 ? 16384 : pitch < 65536 ? 32768 : pitch < (1 << 19) ? 65536 : (1 << 19)) / seg_b);  // the segments of the un-skewed row
                const size_t row_block_e = (size_t)rows * pitch_e;
                const size_t payload = (size_t)2 << 30;  // bytes moved
                const unsigned nblk = (unsigned)(payload / ((size_t)rows * tiles_per_row * seg_b));
                if ((size_t)nblk * row_block_e * 8 > total) continue;
                const unsigned grid = nblk * tiles_per_row;
                auto launch = [&]() {
                    if (rows == 1024) {
                        if (write) hipLaunchKernelGGL((tile_rw<512, 32, true>), dim3(grid), dim3(512), 0, 0, buf, pitch_e, tiles_per_row, seg_e, row_block_e, sink);
                        else hipLaunchKernelGGL((tile_rw<512, 32, false>), dim3(grid), dim3(512), 0, 0, buf, pitch_e, tiles_per_row, seg_e, row_block_e, sink);
                    } else {
                        if (write) hipLaunchKernelGGL((tile_rw<1024, 32, true>), dim3(grid), dim3(1024), 0, 0, buf, pitch_e, tiles_per_row, seg_e, row_block_e, sink);
                        else hipLaunchKernelGGL((tile_rw<1024, 32, false>), dim3(grid), dim3(1024), 0, 0, buf, pitch_e, tiles_per_row, seg_e, row_block_e, sink);
                    }
                };
                launch();
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                for (int i = 0; i < 5; ++i) launch();
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf("%-6s %-8d %-10zu %-8u %10.1f\n", write ? "write" : "read", rows, pitch, seg_b, (double)grid * rows * seg_b * 5 / (ms * 1e-3) / 1e9);
            }
    return 0;
}
