// ORACLE — TEST INFRASTRUCTURE ONLY (see rustfft_scalar.hpp header).
// extern "C" surface so tests/, smoke() and bench.py's cpu_baseline leg can drive the CPU
// restatement through ctypes.  Handles own a shared_ptr<const Fft<T>> for T = f32 or f64.
#include "rustfft_scalar.hpp"

#include <chrono>
#include <thread>

using namespace rustfft_oracle;

namespace {
struct Handle {
    int prec;  // 32 or 64
    FftPtr<float> f32;
    FftPtr<double> f64;
};
thread_local std::string g_last_error;
FftPlannerScalar<float> g_planner32;
FftPlannerScalar<double> g_planner64;

template <class F> int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (const FftPanic& e) {
        g_last_error = e.what();
        return 1;
    } catch (const std::exception& e) {
        g_last_error = std::string("exception: ") + e.what();
        return 2;
    }
}
template <class F> Handle* make(int prec, F&& f) {
    Handle* h = new Handle();
    h->prec = prec;
    int rc = guarded([&] { f(h); });
    if (rc != 0) {
        delete h;
        return nullptr;
    }
    return h;
}
Direction dir_of(int inverse) { return inverse ? Direction::Inverse : Direction::Forward; }
}  // namespace

extern "C" {

const char* rfo_last_error() { return g_last_error.c_str(); }
void rfo_free(void* h) { delete (Handle*)h; }

// --- constructors ---------------------------------------------------------------------------
void* rfo_plan(int prec, size_t len, int inverse) {  // FftPlannerScalar::plan_fft (plan.rs:289-295)
    return make(prec, [&](Handle* h) {
        if (prec == 32)
            h->f32 = g_planner32.plan_fft(len, dir_of(inverse));
        else
            h->f64 = g_planner64.plan_fft(len, dir_of(inverse));
    });
}
void* rfo_new_dft(int prec, size_t len, int inverse) {
    return make(prec, [&](Handle* h) {
        if (prec == 32)
            h->f32 = std::make_shared<Dft<float>>(len, dir_of(inverse));
        else
            h->f64 = std::make_shared<Dft<double>>(len, dir_of(inverse));
    });
}
void* rfo_new_butterfly(int prec, size_t len, int inverse) {
    return make(prec, [&](Handle* h) {
        if (prec == 32)
            h->f32 = std::make_shared<Butterfly<float>>(len, dir_of(inverse));
        else
            h->f64 = std::make_shared<Butterfly<double>>(len, dir_of(inverse));
    });
}
void* rfo_new_radix4(int prec, size_t len, int inverse) {  // Radix4::new (radix4.rs:42-66)
    return make(prec, [&](Handle* h) {
        if (prec == 32)
            h->f32 = Radix4<float>::make(len, dir_of(inverse));
        else
            h->f64 = Radix4<double>::make(len, dir_of(inverse));
    });
}
void* rfo_new_radix4_with_base(unsigned k, void* base) {  // Radix4::new_with_base (radix4.rs:69-119)
    Handle* b = (Handle*)base;
    return make(b->prec, [&](Handle* h) {
        if (b->prec == 32)
            h->f32 = std::make_shared<Radix4<float>>(k, b->f32);
        else
            h->f64 = std::make_shared<Radix4<double>>(k, b->f64);
    });
}
void* rfo_new_radixn(const unsigned char* factors, size_t nfactors, void* base) {  // RadixN::new (radixn.rs:54-155)
    Handle* b = (Handle*)base;
    std::vector<uint8_t> fs(factors, factors + nfactors);
    return make(b->prec, [&](Handle* h) {
        if (b->prec == 32)
            h->f32 = std::make_shared<RadixN<float>>(fs, b->f32);
        else
            h->f64 = std::make_shared<RadixN<double>>(fs, b->f64);
    });
}
void* rfo_new_mixed_radix(void* width, void* height, int small) {
    Handle *w = (Handle*)width, *hh = (Handle*)height;
    return make(w->prec, [&](Handle* h) {
        if (w->prec == 32)
            h->f32 = std::make_shared<MixedRadix<float>>(w->f32, hh->f32, small != 0);
        else
            h->f64 = std::make_shared<MixedRadix<double>>(w->f64, hh->f64, small != 0);
    });
}
void* rfo_new_good_thomas(void* width, void* height, int small) {
    Handle *w = (Handle*)width, *hh = (Handle*)height;
    return make(w->prec, [&](Handle* h) {
        if (w->prec == 32)
            h->f32 = std::make_shared<GoodThomas<float>>(w->f32, hh->f32, small != 0);
        else
            h->f64 = std::make_shared<GoodThomas<double>>(w->f64, hh->f64, small != 0);
    });
}
void* rfo_new_raders(void* inner) {
    Handle* i = (Handle*)inner;
    return make(i->prec, [&](Handle* h) {
        if (i->prec == 32)
            h->f32 = std::make_shared<RadersAlgorithm<float>>(i->f32);
        else
            h->f64 = std::make_shared<RadersAlgorithm<double>>(i->f64);
    });
}
void* rfo_new_bluesteins(size_t len, void* inner) {
    Handle* i = (Handle*)inner;
    return make(i->prec, [&](Handle* h) {
        if (i->prec == 32)
            h->f32 = std::make_shared<BluesteinsAlgorithm<float>>(len, i->f32);
        else
            h->f64 = std::make_shared<BluesteinsAlgorithm<double>>(len, i->f64);
    });
}

// --- trait surface ----------------------------------------------------------------------------
size_t rfo_len(void* hv) {
    Handle* h = (Handle*)hv;
    return h->prec == 32 ? h->f32->len() : h->f64->len();
}
int rfo_direction(void* hv) {
    Handle* h = (Handle*)hv;
    return (int)(h->prec == 32 ? h->f32->fft_direction() : h->f64->fft_direction());
}
const char* rfo_name(void* hv) {
    Handle* h = (Handle*)hv;
    return h->prec == 32 ? h->f32->name() : h->f64->name();
}
size_t rfo_scratch_len(void* hv, int mode) {  // 0 inplace, 1 outofplace, 2 immutable
    Handle* h = (Handle*)hv;
    if (h->prec == 32)
        return mode == 0 ? h->f32->get_inplace_scratch_len()
                         : (mode == 1 ? h->f32->get_outofplace_scratch_len() : h->f32->get_immutable_scratch_len());
    return mode == 0 ? h->f64->get_inplace_scratch_len()
                     : (mode == 1 ? h->f64->get_outofplace_scratch_len() : h->f64->get_immutable_scratch_len());
}
int rfo_process(void* hv, void* buf, size_t n) {
    Handle* h = (Handle*)hv;
    return guarded([&] {
        if (h->prec == 32)
            h->f32->process((cx<float>*)buf, n);
        else
            h->f64->process((cx<double>*)buf, n);
    });
}
int rfo_process_with_scratch(void* hv, void* buf, size_t n, void* scratch, size_t ns) {
    Handle* h = (Handle*)hv;
    return guarded([&] {
        if (h->prec == 32)
            h->f32->process_with_scratch((cx<float>*)buf, n, (cx<float>*)scratch, ns);
        else
            h->f64->process_with_scratch((cx<double>*)buf, n, (cx<double>*)scratch, ns);
    });
}
int rfo_process_outofplace_with_scratch(void* hv, void* in, size_t n_in, void* out, size_t n_out, void* scratch, size_t ns) {
    Handle* h = (Handle*)hv;
    return guarded([&] {
        if (h->prec == 32)
            h->f32->process_outofplace_with_scratch((cx<float>*)in, n_in, (cx<float>*)out, n_out, (cx<float>*)scratch, ns);
        else
            h->f64->process_outofplace_with_scratch((cx<double>*)in, n_in, (cx<double>*)out, n_out, (cx<double>*)scratch, ns);
    });
}
int rfo_process_immutable_with_scratch(void* hv, const void* in, size_t n_in, void* out, size_t n_out, void* scratch, size_t ns) {
    Handle* h = (Handle*)hv;
    return guarded([&] {
        if (h->prec == 32)
            h->f32->process_immutable_with_scratch((const cx<float>*)in, n_in, (cx<float>*)out, n_out, (cx<float>*)scratch, ns);
        else
            h->f64->process_immutable_with_scratch((const cx<double>*)in, n_in, (cx<double>*)out, n_out, (cx<double>*)scratch, ns);
    });
}

// cpu_baseline helper: `threads` caller threads share one Fft instance, each owning a contiguous
// slice of the batch (examples/concurrency.rs:9-30); every thread runs process_with_scratch `reps`
// times on its slice with its own scratch (benches/bench_rustfft.rs:43-54).  Returns seconds.
double rfo_time_batch(void* hv, void* buf, size_t batch, int reps, int threads) {
    Handle* h = (Handle*)hv;
    const size_t n = rfo_len(hv);
    const size_t esz = h->prec == 32 ? sizeof(cx<float>) : sizeof(cx<double>);
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; ++t) {
        size_t lo = batch * t / threads, hi = batch * (t + 1) / threads;
        pool.emplace_back([=] {
            std::vector<char> scratch(rfo_scratch_len(hv, 0) * esz);
            char* p = (char*)buf + lo * n * esz;
            for (int r = 0; r < reps; ++r) rfo_process_with_scratch(hv, p, (hi - lo) * n, scratch.data(), scratch.size() / esz);
        });
    }
    for (auto& th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

// --- planner introspection --------------------------------------------------------------------
// Writes the recipe string for `len` (FftPlannerScalar::design_fft_for_len, plan.rs:312-323).
int rfo_recipe(size_t len, char* out, size_t cap) {
    return guarded([&] {
        FftPlannerScalar<double> p;
        std::string s = p.design_fft_for_len(len)->str();
        if (cap == 0) return;
        size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
        std::memcpy(out, s.data(), n);
        out[n] = 0;
    });
}
// Arc::ptr_eq analogue for the cache tests (plan.rs:832-870): same underlying instance?
int rfo_same_instance(void* a, void* b) {
    Handle *x = (Handle*)a, *y = (Handle*)b;
    if (x->prec != y->prec) return 0;
    return x->prec == 32 ? (x->f32.get() == y->f32.get()) : (x->f64.get() == y->f64.get());
}

// --- math_utils -------------------------------------------------------------------------------
unsigned long long rfo_modular_exponent(unsigned long long b, unsigned long long e, unsigned long long m) {
    return modular_exponent(b, e, m);
}
unsigned long long rfo_primitive_root(unsigned long long prime) {
    uint64_t r = 0;
    return primitive_root(prime, &r) ? r : 0;
}
size_t rfo_distinct_prime_factors(unsigned long long n, unsigned long long* out, size_t cap) {
    auto v = distinct_prime_factors(n);
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return v.size();
}
// out = [n, power_two, power_three, total_factor_count, distinct_factor_count, n_other, (value,count)...]
size_t rfo_prime_factors(size_t n, unsigned long long* out, size_t cap) {
    PrimeFactors f = PrimeFactors::compute(n);
    std::vector<unsigned long long> v = {f.n, f.power_two, f.power_three, f.total_factor_count, f.distinct_factor_count,
                                         f.other_factors.size()};
    for (auto& o : f.other_factors) {
        v.push_back(o.value);
        v.push_back(o.count);
    }
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return v.size();
}
int rfo_partition_factors(size_t n, unsigned long long* left, unsigned long long* right) {
    return guarded([&] {
        auto p = PrimeFactors::compute(n).partition_factors();
        *left = p.first.get_product();
        *right = p.second.get_product();
    });
}
size_t rfo_reverse_bits(size_t value, size_t d, unsigned rev_digits) { return reverse_bits(value, d, rev_digits); }
void rfo_compute_twiddle(int prec, size_t index, size_t fft_len, int inverse, double* re, double* im) {
    if (prec == 32) {
        auto t = compute_twiddle<float>(index, fft_len, dir_of(inverse));
        *re = t.re;
        *im = t.im;
    } else {
        auto t = compute_twiddle<double>(index, fft_len, dir_of(inverse));
        *re = t.re;
        *im = t.im;
    }
}
}  // extern "C"
