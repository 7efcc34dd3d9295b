// C++17 host-side mirror of RustFFT's planner / trait surface over the C ABI (include/mi355fft.h).
// Same names, argument meaning and error behaviour as the reference:
//   FftPlanner<T>::new / plan_fft / plan_fft_forward / plan_fft_inverse        src/plan.rs:72-126
//   Fft<T>::process / process_with_scratch / process_outofplace_with_scratch /
//          process_immutable_with_scratch / get_*_scratch_len / len / fft_direction   src/lib.rs:140-278
// A reference panic (src/common.rs:13-104) becomes `mi355::FftPanic` carrying the same message.
// Header-only; link with -lmi355fft.  There is no CPU fallback: FftPlanner's constructor throws when no
// gfx950 device is visible (the `Err(())` of FftPlannerAvx::new, src/avx/avx_planner.rs:113-164).
#pragma once
#include <complex>
#include <cstddef>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mi355fft.h"

namespace mi355 {

enum class FftDirection { Forward = MI355FFT_FORWARD, Inverse = MI355FFT_INVERSE };

struct FftPanic : std::runtime_error {
    int status;
    FftPanic(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

namespace detail {
template <class T> struct precision_of;
template <> struct precision_of<float> { static constexpr int value = 32; };
template <> struct precision_of<double> { static constexpr int value = 64; };
inline void check(int rc) {
    if (rc != MI355FFT_OK) {
        const char* m = mi355fft_last_error();
        throw FftPanic(rc, (m && *m) ? m : mi355fft_strerror(rc));
    }
}
}  // namespace detail

// `dyn Fft<T>`: immutable after construction, shareable between threads (Send + Sync).
template <class T> class Fft {
   public:
    using Complex = std::complex<T>;  // same layout as num_complex::Complex<T>
    Fft(std::size_t len, FftDirection direction) {
        detail::check(mi355fft_plan_create(len, (int)direction, detail::precision_of<T>::value, &plan_));
    }
    // the host planner in charge (mi355fft_plan_create_ex): Recipe family, its own compute_twiddle, finished tables
    Fft(std::size_t len, FftDirection direction, const mi355fft_plan_options& options) {
        detail::check(mi355fft_plan_create_ex(len, (int)direction, detail::precision_of<T>::value, &options, &plan_));
    }
    ~Fft() { mi355fft_plan_destroy(plan_); }
    Fft(const Fft&) = delete;
    Fft& operator=(const Fft&) = delete;

    std::size_t len() const { return mi355fft_plan_len(plan_); }
    FftDirection fft_direction() const { return (FftDirection)mi355fft_plan_direction(plan_); }
    std::size_t get_inplace_scratch_len() const { return mi355fft_scratch_len(plan_, MI355FFT_SCRATCH_INPLACE); }
    std::size_t get_outofplace_scratch_len() const { return mi355fft_scratch_len(plan_, MI355FFT_SCRATCH_OUTOFPLACE); }
    std::size_t get_immutable_scratch_len() const { return mi355fft_scratch_len(plan_, MI355FFT_SCRATCH_IMMUTABLE); }

    // host slices (pointer + element count), exactly the trait's four methods
    void process(Complex* buffer, std::size_t n) const {
        std::vector<Complex> scratch(get_inplace_scratch_len());
        process_with_scratch(buffer, n, scratch.data(), scratch.size());
    }
    void process_with_scratch(Complex* buffer, std::size_t n, Complex* scratch, std::size_t scratch_len) const {
        detail::check(mi355fft_process_inplace_host(plan_, buffer, n, scratch, scratch_len));
    }
    void process_outofplace_with_scratch(Complex* input, std::size_t n_in, Complex* output, std::size_t n_out, Complex* scratch,
                                         std::size_t scratch_len) const {
        detail::check(mi355fft_process_outofplace_host(plan_, input, n_in, output, n_out, scratch, scratch_len));
    }
    void process_immutable_with_scratch(const Complex* input, std::size_t n_in, Complex* output, std::size_t n_out, Complex* scratch,
                                        std::size_t scratch_len) const {
        detail::check(mi355fft_process_immutable_host(plan_, input, n_in, output, n_out, scratch, scratch_len));
    }
    // HBM-resident buffers, asynchronous on `stream` (a hipStream_t)
    void process_device(void* buffer, std::size_t batch, void* stream = nullptr) const {
        detail::check(mi355fft_process_inplace_dev(plan_, buffer, batch, stream));
    }
    void process_outofplace_device(void* input, void* output, std::size_t batch, void* stream = nullptr) const {
        detail::check(mi355fft_process_outofplace_dev(plan_, input, output, batch, stream));
    }
    void process_immutable_device(const void* input, void* output, std::size_t batch, void* stream = nullptr) const {
        detail::check(mi355fft_process_immutable_dev(plan_, input, output, batch, stream));
    }
    // waits for `stream` and throws if a fused launch of this plan on it gave up a dependency wait: the verdict the asynchronous device
    // calls above cannot return themselves (mi355fft_plan_synchronize; src/lib.rs:184: an Fft is never silently wrong)
    void synchronize(void* stream = nullptr) const { detail::check(mi355fft_plan_synchronize(plan_, stream)); }
    // what the plan took from options.recipe (MI355FFT_RECIPE_STATUS_*)
    int recipe_status() const { return mi355fft_plan_recipe_status(plan_); }
    std::string describe() const {
        char buf[1024];
        detail::check(mi355fft_plan_describe(plan_, buf, sizeof buf));
        return buf;
    }

   private:
    mi355fft_plan* plan_ = nullptr;
};

// `FftPlanner<T>` with the instance cache of src/fft_cache.rs:5-39 (one Arc per (len, direction)).
template <class T> class FftPlanner {
   public:
    explicit FftPlanner(int device = 0) { detail::check(mi355fft_init(device)); }
    std::shared_ptr<const Fft<T>> plan_fft(std::size_t len, FftDirection direction) {
        auto key = std::make_pair(len, (int)direction);
        auto it = cache_.find(key);
        if (it != cache_.end()) return it->second;
        auto fft = std::make_shared<const Fft<T>>(len, direction);
        cache_[key] = fft;
        return fft;
    }
    // not cached: the options belong to one call (src/plan.rs:134-188 Recipe -> algorithm; twiddles.rs:6-23 -> twiddle_fn)
    std::shared_ptr<const Fft<T>> plan_fft_with(std::size_t len, FftDirection direction, mi355fft_plan_options options) {
        options.struct_size = sizeof(mi355fft_plan_options);
        return std::make_shared<const Fft<T>>(len, direction, options);
    }
    std::shared_ptr<const Fft<T>> plan_fft_forward(std::size_t len) { return plan_fft(len, FftDirection::Forward); }
    std::shared_ptr<const Fft<T>> plan_fft_inverse(std::size_t len) { return plan_fft(len, FftDirection::Inverse); }

   private:
    std::map<std::pair<std::size_t, int>, std::shared_ptr<const Fft<T>>> cache_;
};

}  // namespace mi355
