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
        return bad ? 1 : 0;
    } catch (const mi355::FftPanic& e) {
        std::printf("FftPanic(%d): %s\n", e.status, e.what());
        return e.status == MI355FFT_ERR_NO_DEVICE ? 3 : 1;
    }
}
