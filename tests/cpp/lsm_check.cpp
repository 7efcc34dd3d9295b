// TEST INFRASTRUCTURE: drives the LDS stage machine (rustfft_amd/csrc/lsm.h + lsm_plan.h) on the CPU -- the planner, the program
// and the very kernel body the GPU runs, every thread of a workgroup emulated phase by phase -- against a naive f64 DFT.
//   lsm_check <first> <last> [f32|f64] [reverse]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "lsm_plan.h"

using namespace mi355;
typedef std::complex<double> cd;

template <class T> struct HostExec {
    int nt;
    bool reverse;
    std::vector<cx<T>> regs;
    std::vector<unsigned> wd;
    HostExec(int n, bool rev) : nt(n), reverse(rev), regs((size_t)n * kLsmEmax, cx<T>{0, 0}), wd((size_t)n * 4 * kLsmItems, 0u) {}
    template <class Fn> void for_threads(Fn&& fn) {
        if (reverse)
            for (int t = nt - 1; t >= 0; --t) fn(t, regs.data() + (size_t)t * kLsmEmax);
        else
            for (int t = 0; t < nt; ++t) fn(t, regs.data() + (size_t)t * kLsmEmax);
    }
    unsigned* words(int tid) { return wd.data() + (size_t)tid * 4 * kLsmItems; }
    void barrier() {}
};

static void naive_dft(std::vector<cd>& a) {
    const size_t n = a.size();
    std::vector<cd> w(n), o(n);
    for (size_t i = 0; i < n; ++i) w[i] = std::polar(1.0, -2.0 * M_PI * (double)i / (double)n);
    for (size_t k = 0; k < n; ++k) {
        cd s = 0;
        for (size_t j = 0; j < n; ++j) s += a[j] * w[(j * k) % n];
        o[k] = s;
    }
    a = o;
}

template <class T> static int run(int n, bool reverse, bool verbose) {
    lsm::Hooks hooks;
    hooks.tw = [](size_t i, size_t len) { return std::polar(1.0, -2.0 * M_PI * (double)i / (double)len); };
    hooks.dft = naive_dft;
    lsm::Program pr;
    if (!lsm::build_program(n, (int)sizeof(cx<T>), hooks, pr)) {
        if (verbose) printf("n=%d: no program\n", n);
        return 2;
    }
    std::vector<cx<T>> ltab(pr.ltab.size()), gtab(pr.gtab.size() + 1);
    for (size_t i = 0; i < pr.ltab.size(); ++i) ltab[i] = cx<T>{(T)pr.ltab[i].real(), (T)pr.ltab[i].imag()};
    for (size_t i = 0; i < pr.gtab.size(); ++i) gtab[i] = cx<T>{(T)pr.gtab[i].real(), (T)pr.gtab[i].imag()};
    const int batch = pr.f * 2 + (pr.f > 1 ? 1 : 0);  // two full workgroups and a ragged one
    std::mt19937 rng(n * 7 + 1);
    std::uniform_real_distribution<double> U(0, 10);
    std::vector<cx<T>> x((size_t)batch * n), y((size_t)batch * n);
    for (auto& e : x) e = cx<T>{(T)U(rng), (T)U(rng)};
    double worst = 0;
    for (int dir = 0; dir < 2; ++dir) {
        LsmParams<T> p{};
        p.in = x.data();
        p.out = y.data();
        p.stages = pr.stages.data();
        p.desc = pr.desc.data();
        p.ltab = ltab.data();
        p.gtab = gtab.data();
        p.ldperm = pr.ldperm.data();
        p.stperm = pr.stperm.data();
        p.batch = batch;
        p.nstages = (int)pr.stages.size();
        p.ltab_n = (int)ltab.size();
        p.n = n;
        p.f = pr.f;
        p.tab_off = pr.tab_off;
        p.nt = pr.nt;
        p.sgn = dir ? (T)-1 : (T)1;
        const long long grid = (batch + pr.f - 1) / pr.f;
        std::vector<cx<T>> lds(pr.lds_elems + 8);
        for (long long b = 0; b < grid; ++b) {
            for (auto& e : lds) e = cx<T>{(T)NAN, (T)NAN};  // a slot nobody wrote must not be read
            HostExec<T> ex(pr.nt, reverse);
            if (pr.nt == 64)
                lsm_body<T, 64>(ex, p, b, lds.data());
            else if (pr.nt == 128)
                lsm_body<T, 128>(ex, p, b, lds.data());
            else if (pr.nt == 256)
                lsm_body<T, 256>(ex, p, b, lds.data());
            else if (pr.nt == 512)
                lsm_body<T, 512>(ex, p, b, lds.data());
            else
                lsm_body<T, 1024>(ex, p, b, lds.data());
        }
        for (int r = 0; r < batch; ++r) {
            std::vector<cd> a(n);
            for (int i = 0; i < n; ++i) a[i] = cd(x[(size_t)r * n + i].re, dir ? -x[(size_t)r * n + i].im : x[(size_t)r * n + i].im);
            naive_dft(a);
            double num = 0, den = 0;
            for (int i = 0; i < n; ++i) {
                const cd got(y[(size_t)r * n + i].re, dir ? -y[(size_t)r * n + i].im : y[(size_t)r * n + i].im);
                num += std::norm(got - a[i]);
                den += std::norm(a[i]);
            }
            const double rel = std::sqrt(num / den);
            if (!(rel <= worst)) worst = rel;
        }
    }
    const double tol = sizeof(T) == 4 ? 2e-5 : 1e-13;
    const bool ok = worst < tol;
    if (verbose || !ok)
        printf("n=%d %s nt=%d f=%d stages=%zu lds=%zu ltab=%zu gtab=%zu rel=%.3e %s  %s\n", n, sizeof(T) == 4 ? "f32" : "f64", pr.nt, pr.f, pr.stages.size(), pr.lds_elems * sizeof(cx<T>),
               pr.ltab.size(), pr.gtab.size(), worst, ok ? "ok" : "FAIL", pr.desc_str.c_str());
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    const int first = argc > 1 ? atoi(argv[1]) : 74, last = argc > 2 ? atoi(argv[2]) : first;
    const bool f64 = argc > 3 && !strcmp(argv[3], "f64"), reverse = argc > 4 && !strcmp(argv[4], "reverse");
    int planned = 0, bad = 0;
    for (int n = first; n <= last; ++n) {
        const int rc = f64 ? run<double>(n, reverse, last - first < 40) : run<float>(n, reverse, last - first < 40);
        if (rc == 0) ++planned;
        if (rc == 1) ++bad;
    }
    printf("[%d, %d] %s: %d lengths planned and correct, %d FAILED\n", first, last, f64 ? "f64" : "f32", planned, bad);
    return bad ? 1 : 0;
}
