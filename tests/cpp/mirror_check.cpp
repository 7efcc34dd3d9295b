// Exercises the C++17 host mirror (rustfft_amd/host/mi355fft.hpp) the way a RustFFT caller exercises FftPlanner / Fft
// (src/test_utils.rs:70-209 restated): plan, len / direction, the four API entry points on a batch of three with dirty
// scratch, the panic messages of src/common.rs, against the O(n^2) definition (src/algorithm/dft.rs:55-70) in double.
// Exit codes: 0 = all checks passed on a gfx950 device; 3 = no device, FftPanic raised loudly (what the CPU-only test
// expects: there is no fallback); 1 = a check failed.
#include <cmath>
#include <cstdio>
#include <random>

#include "../../rustfft_amd/host/mi355fft.hpp"

template <class T> static std::vector<std::complex<double>> dft(const std::vector<std::complex<T>>& x, size_t n, size_t row, bool inverse) {
    std::vector<std::complex<double>> out(n);
    const double pi = 3.14159265358979323846264338327950288;
    for (size_t k = 0; k < n; ++k) {
        std::complex<double> acc(0, 0);
        for (size_t j = 0; j < n; ++j) {
            const double a = (inverse ? 2.0 : -2.0) * pi * (double)((j * k) % n) / (double)n;
            acc += std::complex<double>(x[row * n + j].real(), x[row * n + j].imag()) * std::complex<double>(std::cos(a), std::sin(a));
        }
        out[k] = acc;
    }
    return out;
}

template <class T> static int check_length(mi355::FftPlanner<T>& planner, size_t n, mi355::FftDirection dir) {
    using C = std::complex<T>;
    auto fft = planner.plan_fft(n, dir);
    if (fft->len() != n || fft->fft_direction() != dir) return std::printf("len/direction mismatch at %zu\n", n), 1;
    if (planner.plan_fft(n, dir).get() != fft.get()) return std::printf("planner cache miss at %zu\n", n), 1;
    const size_t batch = 3;
    std::mt19937 rng(n);
    std::uniform_real_distribution<double> u(0.0, 10.0);  // tests/accuracy.rs:84-95
    std::vector<C> x(n * batch);
    for (auto& v : x) v = C((T)u(rng), (T)u(rng));
    std::vector<std::vector<std::complex<double>>> want;
    for (size_t r = 0; r < batch; ++r) want.push_back(dft(x, n, r, dir == mi355::FftDirection::Inverse));
    auto mean_err = [&](const std::vector<C>& got) {
        double s = 0;
        for (size_t r = 0; r < batch; ++r)
            for (size_t k = 0; k < n; ++k) s += std::abs(std::complex<double>(got[r * n + k].real(), got[r * n + k].imag()) - want[r][k]);
        return s / (double)(n * batch);
    };
    const C dirty((T)100, (T)100);
    std::vector<C> a = x;
    fft->process(a.data(), a.size());
    std::vector<C> b = x, sb(fft->get_inplace_scratch_len(), dirty);
    fft->process_with_scratch(b.data(), b.size(), sb.data(), sb.size());
    std::vector<C> cin = x, c(n * batch), sc(fft->get_outofplace_scratch_len(), dirty);
    fft->process_outofplace_with_scratch(cin.data(), cin.size(), c.data(), c.size(), sc.data(), sc.size());
    std::vector<C> din = x, d(n * batch), sd(fft->get_immutable_scratch_len(), dirty);
    fft->process_immutable_with_scratch(din.data(), din.size(), d.data(), d.size(), sd.data(), sd.size());
    if (din != x) return std::printf("immutable input modified at %zu\n", n), 1;
    const double tol = 0.1;  // tests/accuracy.rs:30-37
    for (const auto* got : {&a, &b, &c, &d})
        if (!(mean_err(*got) < tol)) return std::printf("n=%zu: mean error %g\n", n, mean_err(*got)), 1;
    // panics keep the reference's text (src/common.rs:13-104)
    try {
        std::vector<C> bad(n + 1);
        if (n > 1) {
            fft->process(bad.data(), bad.size());
            return std::printf("n=%zu: expected a panic for a non-multiple buffer\n", n), 1;
        }
    } catch (const mi355::FftPanic& e) {
        if (std::string(e.what()).find("multiple of FFT length") == std::string::npos) return std::printf("unexpected panic text: %s\n", e.what()), 1;
    }
    std::printf("ok n=%zu dir=%d %s: %s\n", n, (int)dir, sizeof(T) == 4 ? "f32" : "f64", fft->describe().c_str());
    return 0;
}

// the host planner in charge (mi355fft_plan_create_ex): its own compute_twiddle (src/twiddles.rs:6-23 restated) and the
// Recipe family it designed; the result must equal the default plan's bit for bit (same formula) / within tolerance
static void host_twiddle(void* ctx, size_t index, size_t fft_len, double* re, double* im) {
    ++*(size_t*)ctx;
    const double angle = -2.0 * 3.14159265358979323846264338327950288 / (double)fft_len * (double)index;
    *re = std::cos(angle);
    *im = std::sin(angle);
}
static int check_host_planner() {
    using C = std::complex<float>;
    mi355::FftPlanner<float> planner;
    for (size_t n : {1024, 1200, 65536}) {
        size_t calls = 0;
        mi355fft_plan_options o{};
        o.twiddle_fn = host_twiddle;
        o.twiddle_ctx = &calls;
        auto hosted = planner.plan_fft_with(n, mi355::FftDirection::Forward, o);
        auto plain = planner.plan_fft_forward(n);
        std::vector<C> a(2 * n), b;
        for (size_t i = 0; i < a.size(); ++i) a[i] = C((float)(i % 17), (float)(i % 5));
        b = a;
        hosted->process(a.data(), a.size());
        plain->process(b.data(), b.size());
        if (!calls || a != b) return std::printf("host twiddle plan differs at %zu (calls %zu)\n", n, calls), 1;
    }
    mi355fft_plan_options o{};
    o.algorithm = MI355FFT_ALGO_BLUESTEIN;
    if (planner.plan_fft_with(1024, mi355::FftDirection::Forward, o)->describe().find("bluestein") == std::string::npos)
        return std::printf("ALGO_BLUESTEIN ignored\n"), 1;
    o.algorithm = MI355FFT_ALGO_RADER;
    try {
        planner.plan_fft_with(1000, mi355::FftDirection::Forward, o);
        return std::printf("ALGO_RADER accepted a composite length\n"), 1;
    } catch (const mi355::FftPanic& e) {
        if (e.status != MI355FFT_ERR_UNSUPPORTED) return std::printf("unexpected status %d\n", e.status), 1;
    }
    std::printf("ok host planner options\n");
    return 0;
}

int main() {
    try {
        mi355::FftPlanner<float> pf;
        mi355::FftPlanner<double> pd;
        int bad = 0;
        for (size_t n : {1, 2, 7, 16, 100, 127, 1009, 1024, 1200, 4099})
            for (auto dir : {mi355::FftDirection::Forward, mi355::FftDirection::Inverse}) {
                bad += check_length<float>(pf, n, dir);
                bad += check_length<double>(pd, n, dir);
            }
        bad += check_host_planner();
        return bad ? 1 : 0;
    } catch (const mi355::FftPanic& e) {
        std::printf("FftPanic(%d): %s\n", e.status, e.what());
        return e.status == MI355FFT_ERR_NO_DEVICE ? 3 : 1;
    }
}
