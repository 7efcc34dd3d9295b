// Host side of the engine: plan construction (kernel selection + device tables) and execution.
//
// Plays the role the algorithm constructors play in the reference (twiddle precomputation in
// Radix4::new_with_base, src/algorithm/radix4.rs:69-119; MixedRadix::new, mixed_radix.rs:53-126) and the
// role of FftPlannerScalar::design_fft_for_len (src/plan.rs:312-323) for the GPU: the decisions differ
// from the CPU recipes (they encode LDS capacity and HBM coalescing, not CPU caches) but the transform
// they realise is the same unnormalised DFT.
#include "plan.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <sstream>

#include "backend.h"
#include "kernels_params.h"
#include "lsm_plan.h"

namespace mi355 {

// ---- why a call failed (mi355fft_last_error carries it) ----------------------------------------------------------------------
static thread_local std::string t_detail;
const std::string& exec_detail() { return t_detail; }
static int fail_detail(int rc, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
static int fail_detail(int rc, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    t_detail = buf;
    return rc;
}
// a device allocation failed: how much was asked for, what the device has left, and what the runtime said
static int fail_alloc(const char* what, size_t bytes) {
    size_t fr = 0, tot = 0;
    const std::string why = backend::last_error();
    const bool have = backend::mem_info(&fr, &tot) == 0;
    return fail_detail(MI355FFT_ERR_OUT_OF_MEMORY, "%s: device allocation of %zu bytes failed (%s); device memory free %zu of %zu bytes%s", what, bytes,
                       why.c_str(), fr, tot, have ? "" : " (hipMemGetInfo failed too)");
}

constexpr int KIND_LSM = 22;  // lsm_launch.h
void register_lsm_f32(std::vector<KernelEntry>&);
void register_lsm_f64(std::vector<KernelEntry>&);
static std::once_flag g_reg_once;
std::vector<KernelEntry>& registry() {
    static std::vector<KernelEntry> r;
    return r;
}
void ensure_registry() {
    std::call_once(g_reg_once, [] {
        auto& r = registry();
        register_k1_f32(r);
        register_k1_f64(r);
        register_lsm_f32(r);
        register_lsm_f64(r);
        register_k2_f32(r);
        register_k2_f64(r);
        register_k2f_f32(r);
        register_k2f_f64(r);
        register_np2_f32(r);
        register_bs_f32(r);
#if defined(MI355_MINIMAL)
        register_bs57_f32(r);
        register_np2_f64(r);
        register_bs57_f64(r);  // `make tuning-min`: power-of-two kernels + the f32 Rader / Bluestein unit (kernel experiments that need a one-minute build)
#if defined(MI355_MINIMAL_RADER)  // the emulator's tuning build: also the compiled Rader bodies (a variant is chosen per prime that HAS a default body)
        register_rader_f32_0(r);
        register_rader_f32_1(r);
        register_rader_f32_2(r);
        register_rader_f32_3(r);
        register_rader_f32_ns0(r);
        register_rader_f32_ns1(r);
        register_rader_f64_0(r);
        register_rader_f64_1(r);
        register_rader_f64_2(r);
        register_rader_f64_3(r);
#endif
        return;
#endif
#if !defined(MI355_MINIMAL)
        register_np2_f64(r);
        register_bs57_f32(r);
        register_bs57_f64(r);
#endif
        register_k2g_f32_0(r);
        register_k2g_f32_1(r);
        register_k2g_f32_2(r);
        register_k2g_f32_3(r);
        register_k2g_f32_4(r);
        register_k2g_f32_5(r);
        register_k2g_f32_6(r);
        register_k2g_f32_7(r);
        register_k2g_f32_ns0(r);
        register_k2g_f32_ns1(r);
        register_k2g_f32_ns2(r);
        register_k2g_f32_ns3(r);
        register_k2g_f64_0(r);
        register_k2g_f64_1(r);
        register_k2g_f64_2(r);
        register_k2g_f64_3(r);
        register_k2g_f64_4(r);
        register_k2g_f64_5(r);
        register_k2g_f64_6(r);
        register_k2g_f64_7(r);
        register_k2gr_f32_0(r);
        register_k2gr_f32_1(r);
        register_k2gr_f32_2(r);
        register_k2gr_f32_3(r);
        register_k2gr_f64_0(r);
        register_k2gr_f64_1(r);
        register_k2gr_f64_2(r);
        register_k2gr_f64_3(r);
        register_k2r_f32(r);
        register_k2r_f64(r);
        register_smooth_f32_0(r);
        register_smooth_f32_1(r);
        register_smooth_f32_2(r);
        register_smooth_f32_3(r);
        register_smooth_f32_4(r);
        register_smooth_f32_5(r);
        register_smooth_f32_6(r);
        register_smooth_f32_7(r);
        register_smooth_f64_0(r);
        register_smooth_f64_1(r);
        register_smooth_f64_2(r);
        register_smooth_f64_3(r);
        register_smooth_f64_4(r);
        register_smooth_f64_5(r);
        register_smooth_f64_6(r);
        register_smooth_f64_7(r);
        register_smooth2_f32_0(r);
        register_smooth2_f32_1(r);
        register_smooth2_f32_2(r);
        register_smooth2_f32_3(r);
        register_smooth2_f64_0(r);
        register_smooth2_f64_1(r);
        register_smooth2_f64_2(r);
        register_smooth2_f64_3(r);
        register_smooth4_f32_0(r);
        register_smooth4_f32_1(r);
        register_smooth4_f32_ns0(r);
        register_smooth4_f32_ns1(r);
        register_smooth4_f32_ns2(r);
        register_smooth4_f32_ns3(r);
        register_smooth4_f32_ns4(r);
        register_smooth4_f32_ns5(r);
        register_smooth4_f32_ns6(r);
        register_smooth4_f32_ns7(r);
        register_smooth4_f32_ns8(r);
        register_smooth4_f32_ns9(r);
        register_smooth4_f64_0(r);
        register_smooth4_f64_1(r);
        register_smooth4_f64_2(r);
        register_smooth4_f64_3(r);
        register_smooth4_f64_4(r);
        register_smooth4_f64_5(r);
        register_smooth4_f64_6(r);
        register_smooth4_f64_7(r);
        register_smooth5_f32_ns0(r);
        register_smooth5_f32_ns1(r);
        register_smooth5_f32_ns2(r);
        register_smooth5_f32_ns3(r);
        register_smooth5_f32_ns4(r);
        register_smooth5_f32_ns5(r);
        register_smooth5_f32_ns6(r);
        register_smooth5_f32_ns7(r);
        register_smooth5_f64_0(r);
        register_smooth5_f64_1(r);
        register_smooth5_f64_2(r);
        register_smooth5_f64_3(r);
        register_smooth5_f64_4(r);
        register_smooth5_f64_5(r);
        register_smooth5_f64_6(r);
        register_smooth5_f64_7(r);
        register_smooth3_f32_0(r);
        register_smooth3_f32_1(r);
        register_smooth3_f32_2(r);
        register_smooth3_f32_3(r);
        register_smooth3_f32_4(r);
        register_smooth3_f32_5(r);
        register_smooth3_f32_6(r);
        register_smooth3_f32_7(r);
        register_smooth3_f32_8(r);
        register_smooth3_f32_9(r);
        register_smooth3_f32_10(r);
        register_smooth3_f32_11(r);
        register_smooth3_f32_12(r);
        register_smooth3_f32_13(r);
        register_smooth3_f64_0(r);
        register_smooth3_f64_1(r);
        register_smooth3_f64_2(r);
        register_smooth3_f64_3(r);
        register_smooth3_f64_4(r);
        register_smooth3_f64_5(r);
        register_smooth3_f64_6(r);
        register_smooth3_f64_7(r);
        register_smooth_f32_ns0(r);
        register_smooth2_f32_ns0(r);
        register_smooth2_f32_ns1(r);
        register_smooth2_f32_ns2(r);
        register_smooth3_f32_ns0(r);
        register_smooth3_f32_ns1(r);
        register_smooth3_f32_ns2(r);
        register_smooth3_f32_ns3(r);
        register_smooth3_f32_ns4(r);
        register_smooth3_f32_ns5(r);
        register_smooth3_f32_ns6(r);
        register_rader_f32_0(r);
        register_rader_f32_1(r);
        register_rader_f32_2(r);
        register_rader_f32_3(r);
        register_rader_f32_ns0(r);
        register_rader_f32_ns1(r);
        register_rader_f64_0(r);
        register_rader_f64_1(r);
        register_rader_f64_2(r);
        register_rader_f64_3(r);
    });
}

// Tuning knobs (alternative tilings, ablation probes, measurement switches) exist only in builds made with
// -DMI355_TUNING (`make tuning` -> lib/libmi355fft_tuning.so).  The shipped library reads NO environment variable:
// nothing outside the C ABI can change what a plan computes.
#if defined(MI355_TUNING) || defined(MI355_EMU)
static int env_int(const char* name) {
    const char* s = getenv(name);
    return s ? atoi(s) : 0;
}
#else
static int env_int(const char*) { return 0; }
#endif
// default tiling = variant 0; MI355FFT_VARIANT=v prefers a variant-v instantiation where one exists (tuning aid)
static const KernelEntry* find_kernel(int kind, int prec, size_t n) {
    const int want = env_int("MI355FFT_VARIANT");
    const KernelEntry* fallback = nullptr;
    for (auto& e : registry())
        if (e.kind == kind && e.prec == prec && (size_t)e.n == n) {
            if (e.variant == want) return &e;
            if (e.variant == 0) fallback = &e;
        }
    return fallback;
}

// src/twiddles.rs:6-23 — forward twiddle, f64 angle, rounded to T by the caller.  While a plan is being built with
// mi355fft_plan_options::twiddle_fn set, every entry comes from the host planner's own compute_twiddle instead.
static thread_local mi355fft_twiddle_fn t_twiddle_fn = nullptr;
static thread_local void* t_twiddle_ctx = nullptr;
static inline void twiddle_f64(size_t index, size_t fft_len, double* re, double* im) {
    if (t_twiddle_fn) {
        t_twiddle_fn(t_twiddle_ctx, index, fft_len, re, im);
        return;
    }
    const double constant = -2.0 * 3.14159265358979323846264338327950288 / (double)fft_len;
    const double angle = constant * (double)index;
    *re = std::cos(angle);
    *im = std::sin(angle);
}
template <class T> static void push_tw(std::vector<T>& v, size_t index, size_t fft_len) {
    double re, im;
    twiddle_f64(index, fft_len, &re, &im);
    v.push_back((T)re);
    v.push_back((T)im);
}

// sub-pass table of a workgroup schedule: pass p >= 1, layout [k-1][r], value w_{s_p R_p}^{r k}
template <class T> static std::vector<T> build_subpass_twiddles(const KernelEntry& k) {
    std::vector<T> t;
    size_t s = k.radix[0];
    for (int p = 1; p < k.np; ++p) {
        const size_t R = k.radix[p];
        for (size_t kk = 1; kk < R; ++kk)
            for (size_t r = 0; r < s; ++r) push_tw<T>(t, r * kk, s * R);
        s *= R;
    }
    return t;
}

Plan::~Plan() {
    // asynchronous launches may still be reading the tables / workspaces: drain the device (not the cached stream
    // handles -- the caller may have destroyed those streams already)
    DeviceGuard dev(device);
    backend::sync_device();
    for (void* p : device_allocs) backend::dfree(p);
    for (auto& kv : slots) {
        backend::dfree(kv.second->ws.ptr);
        PipeState& pp = kv.second->pipe;
        backend::dfree(pp.ring.ptr);
        backend::dfree(pp.ctrl);
        backend::host_word_free((void*)pp.err_host);
        for (void* s : pp.side) backend::stream_destroy(s);
        for (void* e : pp.ev) backend::event_destroy(e);
        if (pp.ev_fork) backend::event_destroy(pp.ev_fork);
    }
    for (auto* pool : {&host_pool, &host_busy})  // host-slice staging contexts
        for (auto& c : *pool) {
            backend::dfree(c->in.ptr);
            backend::dfree(c->out.ptr);
            backend::stream_destroy(c->stream_a);
            backend::stream_destroy(c->stream_b);
        }
}

template <class T> static void* upload(Plan& plan, const std::vector<T>& host, int* rc) {
    void* d = backend::dmalloc(host.size() * sizeof(T));
    if (!d) {
        *rc = MI355FFT_ERR_OUT_OF_MEMORY;
        return nullptr;
    }
    plan.device_allocs.push_back(d);
    if (!host.empty()) {
        if (backend::h2d(d, host.data(), host.size() * sizeof(T), nullptr) || backend::sync(nullptr)) {
            *rc = MI355FFT_ERR_HIP;
            return nullptr;
        }
    }
    return d;
}

// ---- host-side double precision DFT used only to precompute the Rader / Bluestein spectra -------------------
typedef std::complex<double> cd;
static void host_dft(std::vector<cd>& a) {
    const size_t n = a.size();
    if (n <= 1) return;
    std::vector<cd> w(n);
    for (size_t i = 0; i < n; ++i) {
        double re, im;
        twiddle_f64(i, n, &re, &im);
        w[i] = cd(re, im);
    }
    if ((n & (n - 1)) == 0) {  // iterative radix-2, twiddles from the exact table
        for (size_t i = 1, j = 0; i < n; ++i) {
            size_t bit = n >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) std::swap(a[i], a[j]);
        }
        for (size_t len = 2; len <= n; len <<= 1) {
            const size_t step = n / len;
            for (size_t i = 0; i < n; i += len)
                for (size_t k = 0; k < len / 2; ++k) {
                    cd u = a[i + k], v = a[i + k + len / 2] * w[k * step];
                    a[i + k] = u + v;
                    a[i + k + len / 2] = u - v;
                }
        }
        return;
    }
    // any other length: decimation in time by the smallest prime factor, O(n * sum of prime factors); the roots always
    // come from the exact table of the full length
    struct Rec {
        static std::vector<cd> go(const std::vector<cd>& x, const std::vector<cd>& w) {
            const size_t n = x.size(), N = w.size(), scale = N / n;
            if (n == 1) return x;
            size_t p = n;
            for (size_t d = 2; d * d <= n; ++d)
                if (n % d == 0) {
                    p = d;
                    break;
                }
            std::vector<cd> out(n);
            if (p == n) {  // prime: the definition
                for (size_t k = 0; k < n; ++k) {
                    cd acc(0, 0);
                    size_t idx = 0;
                    for (size_t j = 0; j < n; ++j) {
                        acc += x[j] * w[idx * scale];
                        idx += k;
                        if (idx >= n) idx -= n;
                    }
                    out[k] = acc;
                }
                return out;
            }
            const size_t m = n / p;
            std::vector<std::vector<cd>> y(p);
            std::vector<cd> sub(m);
            for (size_t r = 0; r < p; ++r) {
                for (size_t j = 0; j < m; ++j) sub[j] = x[r + p * j];
                y[r] = go(sub, w);
            }
            for (size_t q = 0; q < p; ++q)
                for (size_t k = 0; k < m; ++k) {
                    const size_t kk = k + m * q;
                    cd acc = y[0][k];
                    for (size_t r = 1; r < p; ++r) acc += y[r][k] * w[((r * kk) % n) * scale];
                    out[kk] = acc;
                }
            return out;
        }
    };
    a = Rec::go(a, w);
}
template <class T> static std::vector<T> to_interleaved(const std::vector<cd>& v) {
    std::vector<T> o;
    o.reserve(v.size() * 2);
    for (auto& c : v) {
        o.push_back((T)c.real());
        o.push_back((T)c.imag());
    }
    return o;
}
// src/twiddles.rs:25-57 — chirp w[i] = twiddle(i^2 mod 2n, 2n), forward
static std::vector<cd> bluestein_chirp(size_t n) {
    std::vector<cd> w(n);
    for (size_t i = 0; i < n; ++i) {
        unsigned __int128 sq = (unsigned __int128)i * i;
        size_t e = (size_t)(sq % (unsigned __int128)(2 * n));
        double re, im;
        twiddle_f64(e, 2 * n, &re, &im);
        w[i] = cd(re, im);
    }
    return w;
}
static uint64_t modpow(uint64_t b, uint64_t e, uint64_t m) {
    unsigned __int128 r = 1, bb = b % m;
    while (e) {
        if (e & 1) r = r * bb % m;
        bb = bb * bb % m;
        e >>= 1;
    }
    return (uint64_t)r;
}
// smallest primitive root of the prime p (same definition as src/math_utils.rs:3-20)
static uint64_t primitive_root(uint64_t p) {
    std::vector<uint64_t> fs;
    uint64_t m = p - 1;
    for (uint64_t d = 2; d * d <= m; ++d)
        if (m % d == 0) {
            fs.push_back(d);
            while (m % d == 0) m /= d;
        }
    if (m > 1) fs.push_back(m);
    for (uint64_t g = 2; g < p; ++g) {
        bool ok = true;
        for (uint64_t f : fs)
            if (modpow(g, (p - 1) / f, p) == 1) {
                ok = false;
                break;
            }
        if (ok) return g;
    }
    return 0;
}

// Run-time schedule for a 13-smooth length (dyn_engine.h): greedy largest-radix factorisation over the compiled
// butterfly set, threads per sequence from the 16-value register budget, sequences per workgroup from LDS.
static bool build_dyn_sched(size_t n, size_t esz, size_t extra_lds_elems_per_seq, DynSched& s) {
    static const int allowed_full[] = {16, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};
    static const int allowed_light[] = {12, 10, 9, 8, 6, 5, 4, 3, 2};  // dyn_engine.h LIGHT set (lengths 2^a 3^b 5^c)
    static const int allowed_heavy[] = {31, 29, 23, 19, 17, 16, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};  // HEAVY set: 32 values per thread
    if (n < 2 || n > 16384) return false;
    std::vector<int> radices;
    int set = 1;  // 1 light, 0 full, 2 heavy
    {
        size_t t = n;
        for (int q : {2, 3, 5})
            while (t % q == 0) t /= q;
        if (t != 1) {
            set = 0;
            for (int q : {17, 19, 23, 29, 31})
                if (n % q == 0) set = 2;
        }
    }
    const int* allowed = set == 1 ? allowed_light : set == 2 ? allowed_heavy : allowed_full;
    const int n_allowed = set == 1 ? 9 : set == 2 ? 18 : 13;
    const int emax = set == 1 ? 12 : set == 2 ? 32 : 16;
    const int max_threads = set == 2 ? 256 : 512;
    size_t rem = n;
    while (rem > 1) {
        int pick = 0;
        for (int i = 0; i < n_allowed; ++i)
            if (rem % allowed[i] == 0) {
                pick = allowed[i];
                break;
            }
        if (!pick) return false;  // a prime factor above 31 (13 for the other sets)
        radices.push_back(pick);
        rem /= pick;
    }
    if ((int)radices.size() > kDynMaxPass) return false;
    std::sort(radices.begin(), radices.end(), std::greater<int>());
    s = DynSched{};
    s.n = (int)n;
    s.light = set;
    s.np = (int)radices.size();
    int tpf = 1, stride = 1, off = 0;
    for (int p = 0; p < s.np; ++p) {
        const int R = radices[p], nb = (int)n / R, per_thread = emax / R;
        tpf = std::max(tpf, (nb + per_thread - 1) / per_thread);
        s.radix[p] = R;
        s.nb[p] = nb;
        s.stride[p] = stride;
        s.rcp_stride[p] = stride > 1 ? (unsigned)((((unsigned long long)1 << 32) + stride - 1) / stride) : 0;
        s.tw_off[p] = off;
        if (p >= 1) off += (R - 1) * stride;
        stride *= R;
    }
    if (tpf > max_threads) return false;  // the kernels are compiled for at most 512 (HEAVY: 256) threads per workgroup
    s.tpf = tpf;
    s.rcp_tpf = tpf > 1 ? (unsigned)((((unsigned long long)1 << 32) + tpf - 1) / tpf) : 0;
    for (int p = 0; p < s.np; ++p) s.bpt[p] = (s.nb[p] + tpf - 1) / tpf;
    s.pitch = (dyn_phys((int)n - 1) + 1) | 1;
    const size_t per_seq = ((size_t)s.pitch + extra_lds_elems_per_seq) * esz;
    int f = std::max(1, 256 / tpf);
    while (f > 1 && (size_t)f * per_seq > 64 * 1024) --f;
    while (f > 1 && f * tpf > max_threads) --f;
    if ((size_t)f * per_seq > 150 * 1024 || f * tpf > max_threads) return false;
    s.f = f;
    return true;
}
template <class T> static std::vector<T> build_dyn_twiddles(const DynSched& s) {
    std::vector<T> t;
    for (int p = 1; p < s.np; ++p)
        for (int kk = 1; kk < s.radix[p]; ++kk)
            for (int r = 0; r < s.stride[p]; ++r) push_tw<T>(t, (size_t)r * kk, (size_t)s.stride[p] * s.radix[p]);
    return t;
}
static const KernelEntry* find_kind(int kind, int prec) {
    for (auto& e : registry())
        if (e.kind == kind && e.prec == prec) return &e;
    return nullptr;
}
static bool is_prime_sz(size_t n) {
    if (n < 2) return false;
    for (size_t d = 2; d * d <= n; ++d)
        if (n % d == 0) return false;
    return true;
}

// choose macro radices r_1..r_P (each with a FIRST and a LATER kernel) whose product is n:
// fewest passes, then the most balanced split, larger radices first.
// `fused_resplit` (may be null): set when the split was taken ONLY because a default fused kernel exists for it -- build_plan then keeps the
// balanced split as the plan's unfused form (Plan::unfused_alt); allow_fused_resplit = false builds that form.
static bool choose_macro_radices(int prec, size_t n, std::vector<size_t>& out, bool allow_fused_resplit = true, bool* fused_resplit = nullptr) {
    std::vector<size_t> avail;
    for (auto& e : registry())
        if (e.kind == KIND_K2_FIRST && e.prec == prec && e.variant == 0 && find_kernel(KIND_K2_LATER, prec, e.n) &&
            (env_int("MI355FFT_MAXR") == 0 || e.n <= env_int("MI355FFT_MAXR")))
            avail.push_back(e.n);
    std::sort(avail.begin(), avail.end(), std::greater<size_t>());
    avail.erase(std::unique(avail.begin(), avail.end()), avail.end());
    std::vector<size_t> best, cur;
    size_t best_max = 0;
    for (int P = 2; P <= 5 && best.empty(); ++P) {
        // enumerate non-increasing sequences of length P
        struct Rec {
            static void go(const std::vector<size_t>& av, size_t start, size_t rem, int left, std::vector<size_t>& cur,
                           std::vector<size_t>& best, size_t& best_max) {
                if (left == 0) {
                    if (rem == 1 && (best.empty() || cur.front() < best_max)) {
                        best = cur;
                        best_max = cur.front();
                    }
                    return;
                }
                for (size_t i = start; i < av.size(); ++i) {
                    if (rem % av[i]) continue;
                    cur.push_back(av[i]);
                    go(av, i, rem / av[i], left - 1, cur, best, best_max);
                    cur.pop_back();
                }
            }
        };
        Rec::go(avail, 0, n, P, cur, best, best_max);
    }
    if (best.empty()) return false;
    out = best;
    if (const int r0 = env_int("MI355FFT_R0"))  // tuning: a two-pass plan with this first tile height
        if (n % (size_t)r0 == 0 && find_kernel(KIND_K2_FIRST, prec, (size_t)r0) && find_kernel(KIND_K2_LATER, prec, n / (size_t)r0)) {
            out = {(size_t)r0, n / (size_t)r0};
            return true;
        }
    if (env_int("MI355FFT_ORDER") == 1) std::reverse(out.begin(), out.end());  // tuning: smallest radix first
    // a two-pass plan whose balanced split has no default fused kernel while ANOTHER split (or pass order) of the same length has one
    // takes that one (Complex<f32>: 2^17 as 256 x 512, 2^18 as 256 x 1024 -- measured, kernels_k2f_f32.hip)
    if (out.size() == 2 && env_int("MI355FFT_ORDER") == 0 && allow_fused_resplit) {
        auto named = [&](int kind, const char* name) -> const KernelEntry* {
            for (auto& e : registry())
                if (e.kind == kind && e.prec == prec && e.variant == 0 && !strcmp(e.name, name)) return &e;
            return nullptr;
        };
        auto fused_auto = [&](size_t first, size_t later) {
            const KernelEntry *kf = find_kernel(KIND_K2_FIRST, prec, first), *kl = find_kernel(KIND_K2_LATER, prec, later);
            if (!kf || !kl) return false;
            for (auto& e : registry())
                if (e.kind == KIND_K2_FUSED && e.prec == prec && e.aux == 1 && e.variant == 0 && !strcmp(e.part[0], kf->name) && !strcmp(e.part[1], kl->name)) return true;
            return false;
        };
        if (!fused_auto(out[0], out[1]))
            for (auto& e : registry()) {
                if (e.kind != KIND_K2_FUSED || e.prec != prec || e.aux != 1 || e.variant != 0 || (size_t)e.n != n) continue;
                const KernelEntry *kf = named(KIND_K2_FIRST, e.part[0]), *kl = named(KIND_K2_LATER, e.part[1]);
                if (kf && kl && (size_t)kf->n * (size_t)kl->n == n && fused_auto((size_t)kf->n, (size_t)kl->n)) {
                    out = {(size_t)kf->n, (size_t)kl->n};
                    if (fused_resplit) *fused_resplit = true;
                    break;
                }
            }
    }
    return true;
}

// radices for a composite length that is not a power of two: tile heights with a compiled general pass kernel
// (k2g_body), fewest passes, then the split that wastes the fewest columns, then the most balanced one.
// Column waste: pass p runs ceil(M_p / F_p) tiles of F_p columns over M_p = n / R_p columns, and the short tile heights are WIDE
// (up to 128 columns, so that a workgroup has enough threads): the balanced split of a length just above 4096 -- 4225 = 65 x 65 --
// is one 128-column tile per transform with 63 columns masked, 1.97x the work and traffic, where 169 x 25 wastes 1.40x.  185 of the
// 13-smooth lengths below 20000 lose more than 12 % that way (profiles/r4/general_split_waste.json); the cost below is the mean
// padded-over-real column ratio of the passes, compared in steps of 10 % (smaller differences: the balanced split).
static bool choose_general_radices(int prec, size_t n, std::vector<size_t>& out) {
    std::vector<size_t> avail;
    std::vector<int> width;
    for (auto& e : registry())
        if ((e.kind == KIND_K2G_FIRST || (e.kind == KIND_K2R_FIRST && env_int("MI355FFT_NO_K2R") == 0)) && e.prec == prec) avail.push_back(e.n);
    std::sort(avail.begin(), avail.end(), std::greater<size_t>());
    for (size_t r : avail) {
        const KernelEntry* k = find_kernel(KIND_K2G_FIRST, prec, r);
        if (!k) k = find_kernel(KIND_K2R_FIRST, prec, r);
        width.push_back(k ? k->f : 1);
    }
    std::vector<size_t> best, cur;
    std::vector<size_t> idx;
    long best_cost = 0;
#if defined(MI355_SPLIT_BALANCED)  // A/B builds: the round-3 rule (most balanced split)
    const bool waste_aware = false;
#else
    // Complex<f32> only: measured on 48 / 10 changed lengths (profiles/r4/ab_split_waste_*.jsonl) f32 gains wherever the waste drops
    // by 0.15 or more (+12 ... +41 %; median of all changed lengths +3.9 %), f64 -- narrower tiles, less to gain, and the
    // alternative splits bring radix-13 tiles -- loses on 4 of 10
    const bool waste_aware = prec == 32 && env_int("MI355FFT_SPLIT_BALANCED") == 0;
#endif
    std::function<void(size_t, size_t, int)> go = [&](size_t start, size_t rem, int left) {
        if (left == 0) {
            if (rem != 1) return;
            double c = 0;
            for (size_t p = 0; p < cur.size(); ++p) {
                const size_t M = n / cur[p], f = (size_t)width[idx[p]];
                c += (double)(((M + f - 1) / f) * f) / (double)M;
            }
            const long cost = waste_aware ? std::lround(c / (double)cur.size() * 10.0) : 0;  // 10 % steps
            if (best.empty() || cost < best_cost || (cost == best_cost && cur.front() < best.front())) {
                best = cur;
                best_cost = cost;
            }
            return;
        }
        for (size_t i = start; i < avail.size(); ++i) {
            if (rem % avail[i]) continue;
            cur.push_back(avail[i]);
            idx.push_back(i);
            go(i, rem / avail[i], left - 1);
            cur.pop_back();
            idx.pop_back();
        }
    };
    for (int P = 2; P <= 4 && best.empty(); ++P) go(0, n, P);
    if (best.empty()) return false;
    out = best;
    return true;
}

// two-level table for w_Q^e, e < Q:  e = (e >> h) << h | (e & mask)  (the inter-pass twiddles of pass p > 0, Q = S R)
template <class T> static int build_lohi(Plan& plan, PassDesc& pd, size_t Q) {
    int rc = MI355FFT_OK, bits = 0;
    while (((size_t)1 << bits) < Q) ++bits;
    const int h = (bits + 1) / 2;
    std::vector<T> lo, hi;
    for (size_t e = 0; e < ((size_t)1 << h); ++e) push_tw<T>(lo, e, Q);
    for (size_t q = 0; q <= ((Q - 1) >> h); ++q) push_tw<T>(hi, q << h, Q);
    pd.hshift = h;
    pd.lmask = (int)(((size_t)1 << h) - 1);
    pd.d_tlo = upload<T>(plan, lo, &rc);
    if (rc) return rc;
    pd.d_thi = upload<T>(plan, hi, &rc);
    return rc;
}

// Passes of one length-N transform over the general column-tile kernels; kinds[p] names the kernel family of pass p
// (plain first / later, or one of the fused Bluestein passes).
template <class T> static int rader_tables(Plan& plan, PassDesc& pd, bool inverse_map_in, size_t prime);
template <class T>
static int append_general_passes(Plan& plan, size_t N, const std::vector<size_t>& radices, const std::vector<int>& kinds, void* tab_first,
                                 void* tab_last) {
    int rc = MI355FFT_OK;
    size_t s = 1;
    for (size_t p = 0; p < radices.size(); ++p) {
        const size_t R = radices[p];
        const KernelEntry* k = find_kernel(kinds[p], plan.prec, R);
        // a prime tile height: Rader inside the tile.  Only PLAIN pass kinds can be prime tiles: the gather / multiply / scatter
        // passes of the fused sequences exist for smooth heights only, but the plain passes in between them may be prime (the fused
        // Rader takes its radices from choose_general_radices(p - 1): 41959 runs gather<42> -> k2rlater<37> -> rmul<27> | ...)
        const bool prime_tile = !k && (kinds[p] == KIND_K2G_FIRST || kinds[p] == KIND_K2G_LATER);
        if (prime_tile) k = find_kernel(kinds[p] == KIND_K2G_FIRST ? KIND_K2R_FIRST : KIND_K2R_LATER, plan.prec, R);
        if (!k) return MI355FFT_ERR_UNSUPPORTED;
        if (k->prepare()) return MI355FFT_ERR_HIP;
        PassDesc pd{};
        pd.k = k;
        pd.m = (long long)(N / R);
        pd.s = (long long)s;
        pd.row_n = (long long)N;
        pd.d_aux1 = (p == 0) ? tab_first : (p + 1 == radices.size()) ? tab_last : nullptr;
        if (prime_tile && (rc = rader_tables<T>(plan, pd, true, R))) return rc;  // d[R - 1], the inverse gather map, g^-(j+1)
        pd.d_tw = upload<T>(plan, build_subpass_twiddles<T>(*k), &rc);
        if (rc) return rc;
        if (p > 0 && (rc = build_lohi<T>(plan, pd, s * R))) return rc;
        plan.passes.push_back(pd);
        s *= R;
    }
    return MI355FFT_OK;
}

// Inner length of the fused multi-kernel Bluestein: the smallest product of 2, 3 or 4 tile heights that have the fused
// pass kernels (tools/gen_k2g_kernels.py FUSED) reaching 2n - 1; fewest passes first, most balanced split among equals.
static bool choose_fused_radices(int prec, size_t need, std::vector<size_t>& out) {
    std::vector<size_t> av;
    for (auto& e : registry())
        if (e.kind == KIND_K2G_FIRST_CHIRP && e.prec == prec && find_kernel(KIND_K2G_LAST_MUL, prec, e.n) &&
            find_kernel(KIND_K2G_LAST_CHIRP, prec, e.n) && find_kernel(KIND_K2G_FIRST, prec, e.n) && find_kernel(KIND_K2G_LATER, prec, e.n))
            av.push_back(e.n);
    if (av.empty()) return false;
    std::sort(av.begin(), av.end(), std::greater<size_t>());
    for (int P = 2; P <= 4; ++P) {
        size_t top = 1;
        for (int i = 0; i < P; ++i) top *= av.front();
        if (top < need) continue;
        std::vector<size_t> best, cur;
        size_t best_prod = 0;
        struct Rec {
            static void go(const std::vector<size_t>& av, size_t start, size_t prod, int left, size_t need, std::vector<size_t>& cur,
                           std::vector<size_t>& best, size_t& best_prod) {
                if (left == 0) {
                    if (prod >= need && (best.empty() || prod < best_prod || (prod == best_prod && cur.front() < best.front()))) {
                        best = cur;
                        best_prod = prod;
                    }
                    return;
                }
                for (size_t i = start; i < av.size(); ++i) {
                    cur.push_back(av[i]);
                    go(av, i, prod * av[i], left - 1, need, cur, best, best_prod);
                    cur.pop_back();
                }
            }
        };
        Rec::go(av, 0, 1, P, need, cur, best, best_prod);
        if (!best.empty()) {
            out = best;
            return true;
        }
    }
    return false;
}

// host table handed over by the caller's planner (interleaved Complex<T>, in the PLAN's direction) -> forward-direction
// complex doubles: the device runs the inverse as conj(FFT(conj x)), and every table of the reference's inverse
// algorithm objects is the conjugate of the forward one
template <class T> static std::vector<cd> from_host_table(const void* p, size_t n, bool inverse) {
    const T* t = (const T*)p;
    std::vector<cd> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = cd((double)t[2 * i], inverse ? -(double)t[2 * i + 1] : (double)t[2 * i + 1]);
    return v;
}

// Rader tables (raders_algorithm.rs:65-124): d[j] = FFT_{p-1}(twiddle(g^-j mod p, p)) / (p - 1), g^(j+1), g^-(j+1)
template <class T> static int rader_tables(Plan& plan, PassDesc& pd, bool inverse_map_in, size_t prime) {
    const uint64_t pp = prime ? prime : plan.len, g = primitive_root(pp), ginv = modpow(g, pp - 2, pp);
    std::vector<cd> d(pp - 1);
    std::vector<int> pin(pp - 1), pout(pp - 1);
    uint64_t ti = 1, a = 1, b = 1;
    for (size_t j = 0; j + 1 < pp; ++j) {
        double re, im;
        twiddle_f64(ti, pp, &re, &im);
        d[j] = cd(re, im) / (double)(pp - 1);
        ti = ti * ginv % pp;
        a = a * g % pp;
        b = b * ginv % pp;
        pin[j] = (int)a;
        pout[j] = (int)b;
    }
    if (plan.opt_rader && pp == plan.len)  // the host planner's table belongs to the plan's own length
        d = from_host_table<T>(plan.opt_rader, pp - 1, plan.direction == MI355FFT_INVERSE);
    else
        host_dft(d);
    if (inverse_map_in) {  // bodies that scatter on load want the inverse map t -> j with g^(j+1) = t
        std::vector<int> inv(pp, 0);
        for (size_t j = 0; j + 1 < pp; ++j) inv[(size_t)pin[j]] = (int)j;
        pin.swap(inv);
    }
    int rc = MI355FFT_OK;
    pd.d_aux1 = upload<T>(plan, to_interleaved<T>(d), &rc);
    if (rc) return rc;
    pd.d_perm_in = upload<int>(plan, pin, &rc);
    if (rc) return rc;
    pd.d_perm_out = upload<int>(plan, pout, &rc);
    return rc;
}

// Bluestein tables (bluesteins_algorithm.rs:63-98): chirp[n], and FFT_M of the mirrored conjugate chirp scaled by 1 / M
template <class T> static int bluestein_tables(Plan& plan, size_t M, void** d_chirp, void** d_bf) {
    const size_t n = plan.len;
    const bool inverse = plan.direction == MI355FFT_INVERSE;
    if (plan.opt_bs_mul && plan.opt_bs_inner != M) return MI355FFT_ERR_INVALID_ARG;  // sized for another inner length
    std::vector<cd> chirp = plan.opt_bs_tw ? from_host_table<T>(plan.opt_bs_tw, n, inverse) : bluestein_chirp(n);
    std::vector<cd> bvec;
    if (plan.opt_bs_mul) {
        bvec = from_host_table<T>(plan.opt_bs_mul, M, inverse);
    } else {
        bvec.assign(M, cd(0, 0));
        bvec[0] = std::conj(chirp[0]) / (double)M;
        for (size_t i = 1; i < n; ++i) {
            bvec[i] = std::conj(chirp[i]) / (double)M;
            bvec[M - i] = bvec[i];
        }
        host_dft(bvec);
    }
    int rc = MI355FFT_OK;
    *d_chirp = upload<T>(plan, to_interleaved<T>(chirp), &rc);
    if (rc) return rc;
    *d_bf = upload<T>(plan, to_interleaved<T>(bvec), &rc);
    return rc;
}

// Which Bluestein form serves length n, and over which inner length M (shared by the planner and
// mi355fft_bluestein_inner_len).  form: 1 = one kernel, 2 = two whole-row kernels, 3 = fused column-tile passes,
// 4 = separate element-wise kernels around an inner plan; 0 = none.
struct BluesteinChoice {
    int form = 0;
    size_t M = 0;
    const KernelEntry* k1 = nullptr;
    const KernelEntry* k2 = nullptr;
    std::vector<size_t> radices;
};
static bool choose_fused_radices(int prec, size_t need, std::vector<size_t>& out);
// `prefer`: an inner length the host planner names (its recipe's inner_fft.len(), or the one its finished multiplier table is
// sized for).  Taken when a kernel form exists for exactly that length, in the same order of forms; else the GPU planner's own.
static BluesteinChoice choose_bluestein(int prec, size_t n, size_t prefer = 0, bool* preferred = nullptr) {
    BluesteinChoice c;
    if (preferred) *preferred = false;
    if (n < 2) return c;
    if (prefer >= 2 * n - 1 && prefer < ((size_t)1 << 31)) {
        if (const KernelEntry* k = find_kernel(KIND_BLUESTEIN, prec, prefer)) {
            c.form = 1, c.M = prefer, c.k1 = k;
        } else if (find_kernel(KIND_BS2_FIRST, prec, prefer) && find_kernel(KIND_BS2_SECOND, prec, prefer)) {
            c.form = 2, c.M = prefer, c.k1 = find_kernel(KIND_BS2_FIRST, prec, prefer), c.k2 = find_kernel(KIND_BS2_SECOND, prec, prefer);
        } else if (choose_fused_radices(prec, prefer, c.radices)) {
            size_t prod = 1;
            for (size_t r : c.radices) prod *= r;
            if (prod == prefer) c.form = 3, c.M = prefer;
        }
        if (c.form) {
            if (preferred) *preferred = true;
            return c;
        }
        c = BluesteinChoice();
    }
    for (auto& e : registry())
        if (e.kind == KIND_BLUESTEIN && e.prec == prec && e.variant == 0 && (size_t)e.n >= 2 * n - 1 && (!c.k1 || e.n < c.k1->n)) c.k1 = &e;
    if (c.k1 && env_int("MI355FFT_VARIANT"))  // tuning: an alternative body for the same inner length
        for (auto& e : registry())
            if (e.kind == KIND_BLUESTEIN && e.prec == prec && e.n == c.k1->n && e.variant == env_int("MI355FFT_VARIANT")) c.k1 = &e;
    if (c.k1) {
        c.form = 1;
        c.M = c.k1->n;
        return c;
    }
    if (env_int("MI355FFT_BLUESTEIN_UNFUSED") == 0) {
        for (auto& e : registry())
            if (e.kind == KIND_BS2_FIRST && e.prec == prec && e.variant == 0 && (size_t)e.n >= 2 * n - 1 && (!c.k1 || e.n < c.k1->n) &&
                find_kernel(KIND_BS2_SECOND, prec, e.n))
                c.k1 = &e;
        if (c.k1) {
            c.form = 2;
            c.M = c.k1->n;
            c.k2 = find_kernel(KIND_BS2_SECOND, prec, c.k1->n);
            return c;
        }
        if (2 * n - 1 < ((size_t)1 << 31) && choose_fused_radices(prec, 2 * n - 1, c.radices)) {
            c.form = 3;
            c.M = 1;
            for (size_t r : c.radices) c.M *= r;
            return c;
        }
    }
    size_t M = 1;
    while (M < 2 * n - 1) M <<= 1;
    // a 7-smooth M between 2n - 1 and that power of two pads less (every pass of the pipeline runs over M, not n):
    // take the smallest one the general passes cover in no more kernels than the power of two needs
    if (M < ((size_t)1 << 31) && env_int("MI355FFT_BLUESTEIN_POW2") == 0) {
        std::vector<size_t> r0, r1, cand;
        const size_t p0 = choose_macro_radices(prec, M, r0) ? r0.size() : 4;
        for (size_t a = 1; a < M; a *= 2)
            for (size_t b = a; b < M; b *= 3)
                for (size_t cc = b; cc < M; cc *= 5)
                    for (size_t d = cc; d < M; d *= 7)
                        if (d >= 2 * n - 1 && d > 4096) cand.push_back(d);
        std::sort(cand.begin(), cand.end());
        for (size_t cnd : cand)
            if (cnd * 20 <= M * 17 && choose_general_radices(prec, cnd, r1) && r1.size() <= p0) {  // >= 15 % smaller: the general passes run ~10 % below the power-of-two tiles
                M = cnd;
                break;
            }
    }
    if (M < ((size_t)1 << 31)) {
        c.form = 4;
        c.M = M;
    }
    return c;
}
size_t bluestein_inner_len(size_t len, int prec) {
    ensure_registry();
    return choose_bluestein(prec, len).M;
}

template <class T> static int build_plan_t(Plan& plan) {
    const size_t n = plan.len;
    if (n <= 1) {
        plan.kind = PLAN_TRIVIAL;
        return MI355FFT_OK;
    }
    int rc = MI355FFT_OK;
    const int algo = plan.algorithm;
    const bool direct_ok = (algo == MI355FFT_ALGO_AUTO || algo == MI355FFT_ALGO_MIXED_RADIX);
    const bool rader_ok = (algo == MI355FFT_ALGO_AUTO || algo == MI355FFT_ALGO_RADER);
    const bool bluestein_ok = (algo == MI355FFT_ALGO_AUTO || algo == MI355FFT_ALGO_BLUESTEIN);
    if (algo == MI355FFT_ALGO_RADER && (!is_prime_sz(n) || n < 5)) return MI355FFT_ERR_UNSUPPORTED;  // raders_algorithm.rs:68
    // Lengths with prime factors outside the compiled radix set that fit one workgroup: the LDS stage machine (lsm.h) runs the tree
    // the reference plans for them -- MixedRadix over the smooth part and one Rader per large prime factor, Rader over a recursively
    // planned inner length (src/plan.rs:412-425, 474-506, 636-665) -- as ONE kernel from a run-time program.  A host planner's
    // MixedRadix (composite) / Rader (prime) request gets it too.
    // Returns 1 when the plan was made, 0 when this length is not taken, an error otherwise.  `always`: whatever the program costs.
    auto try_lsm = [&](bool always, int cap = 1 << 20) -> int {
        if (algo == MI355FFT_ALGO_BLUESTEIN || n > 16384 || plan.opt_rader || env_int("MI355FFT_NO_LSM") != 0) return 0;  // (a host planner's finished Rader table is in the reference's form: the compiled / run-time Rader bodies take it)
        lsm::Hooks hooks;
        hooks.tw = [](size_t i, size_t len) {
            double re, im;
            twiddle_f64(i, len, &re, &im);
            return cd(re, im);
        };
        hooks.dft = [](std::vector<cd>& v) { host_dft(v); };
        lsm::Program best;
        const KernelEntry* lk = nullptr;
        for (auto& e : registry())
            if (e.kind == KIND_LSM && e.prec == plan.prec) lk = &e;
        bool have = lk && lsm::build_program((int)n, (int)(2 * sizeof(T)), hooks, best, 64 * 1024, 160 * 1024, env_int("MI355FFT_LSM_NT"), env_int("MI355FFT_LSM_F"));
        // a host planner that names a family gets that family: Rader = a tree whose ROOT is Rader's algorithm (a prime above the radix set),
        // MixedRadix = a composite (a prime is no MixedRadix: src/plan.rs:412-425)
        if (have && algo == MI355FFT_ALGO_RADER && best.root_kind != lsm::RADER) have = false;
        if (have && algo == MI355FFT_ALGO_MIXED_RADIX && best.root_kind == lsm::RADER) have = false;
        // AUTO takes the program where it beat the Bluestein plan of the length in an on-device A/B over 400 random lengths per precision
        // (profiles/r6/lsm_calib_*.jsonl, tools/r6_lsm_calib_report.py), by program length -- the one-kernel Bluestein it competes with up to
        // 4096 runs 1.7 - 2.6 TB/s, the split-exchange and two-kernel forms above it 0.5 - 1.2:
        //   Complex<f32>: up to 1800 (one-kernel Bluestein at 2.0 - 2.8 TB/s) <= 7 stages (medians x1.54, x1.69, x1.32, x1.07 at 4 .. 7; 8: x0.82);
        //                 1800 .. 4096 (Bluestein 1.4 - 1.8) <= 9 (x1.38, x1.14, x1.05 at 7 .. 9; 10: x0.92); <= 9 up to 8192 (x1.5 .. x2.5; 10: x0.96);
        //                 <= 13 above (x1.3 .. x2.9)
        //   Complex<f64>: up to 1800 <= 7 (x1.30, x1.56, x1.28, x1.03); 1800 .. 4096 <= 8 (x1.25, x1.07 at 7, 8; 9: x0.96); <= 9 up to 8192
        //                 (x1.15 .. x1.39); <= 13 above (x2.0 .. x2.5)
        const int calibrated = sizeof(T) == 4 ? (n <= 1800 ? 7 : n <= 4096 ? 9 : n <= 8192 ? 9 : 13) : (n <= 1800 ? 7 : n <= 4096 ? 8 : n <= 8192 ? 9 : 13);
        const int max_stages = env_int("MI355FFT_LSM_MAX_STAGES") ? env_int("MI355FFT_LSM_MAX_STAGES") : (calibrated < cap ? calibrated : cap);
        if (have && ((int)best.stages.size() <= max_stages || always)) {
            if (lk->prepare()) return -MI355FFT_ERR_HIP;
            plan.kind = PLAN_SINGLE;
            PassDesc pd{};
            pd.k = lk;
            std::vector<T> lt = to_interleaved<T>(best.ltab), gt = to_interleaved<T>(best.gtab);
            if (gt.empty()) gt.assign(2, (T)0);
            std::vector<unsigned short> perms(best.ldperm);
            perms.insert(perms.end(), best.stperm.begin(), best.stperm.end());
            pd.lsm.d_stages = upload<LsmStage>(plan, best.stages, &rc);
            if (rc) return -rc;
            pd.lsm.d_desc = upload<unsigned>(plan, best.desc, &rc);
            if (rc) return -rc;
            pd.lsm.d_ltab = upload<T>(plan, lt, &rc);
            if (rc) return -rc;
            pd.lsm.d_gtab = upload<T>(plan, gt, &rc);
            if (rc) return -rc;
            pd.lsm.d_ldperm = upload<unsigned short>(plan, perms, &rc);
            if (rc) return -rc;
            pd.lsm.d_stperm = (unsigned short*)pd.lsm.d_ldperm + best.ldperm.size();
            pd.lsm.nstages = (int)best.stages.size();
            pd.lsm.ltab_n = (int)best.ltab.size();
            pd.lsm.n = (int)n;
            pd.lsm.f = best.f;
            pd.lsm.tab_off = best.tab_off;
            pd.lsm.nt = best.nt;
            pd.lsm.lds_bytes = (int)(best.lds_elems * 2 * sizeof(T));
            pd.lsm.desc = best.desc_str;
            plan.passes.push_back(pd);
            return 1;
        }
        return 0;
    };
    std::vector<size_t> radices;
    if (direct_ok) {
        if (const KernelEntry* k = find_kernel(KIND_K1, plan.prec, n)) {
            if (k->prepare()) return MI355FFT_ERR_HIP;
            plan.kind = PLAN_SINGLE;
            PassDesc pd{};
            pd.k = k;
            pd.d_tw = upload<T>(plan, build_subpass_twiddles<T>(*k), &rc);
            if (rc) return rc;
            plan.passes.push_back(pd);
            return MI355FFT_OK;
        }
        // power-of-two column-tile passes over the given heights; `probe` only answers whether every pass has a kernel whose
        // tile width divides the column count and the stride
        auto macro_passes = [&](const std::vector<size_t>& rs, bool probe) -> int {
            size_t s = 1;
            for (size_t p = 0; p < rs.size(); ++p) {
                const size_t R = rs[p];
                const KernelEntry* k = find_kernel(p == 0 ? KIND_K2_FIRST : KIND_K2_LATER, plan.prec, R);
                if (!k) return MI355FFT_ERR_UNSUPPORTED;
                const size_t M = n / R;
                if (M % k->f != 0 || (p > 0 && s % k->f != 0)) return MI355FFT_ERR_UNSUPPORTED;
                if (!probe) {
                    if (k->prepare()) return MI355FFT_ERR_HIP;
                    PassDesc pd{};
                    pd.k = k;
                    pd.m = (long long)M;
                    pd.s = (long long)s;
                    pd.d_tw = upload<T>(plan, build_subpass_twiddles<T>(*k), &rc);
                    if (rc) return rc;
                    if (p > 0 && (rc = build_lohi<T>(plan, pd, s * R))) return rc;
                    plan.passes.push_back(pd);
                }
                s *= R;
            }
            return MI355FFT_OK;
        };
        auto general_passes = [&](const std::vector<size_t>& rs) -> int {
            plan.kind = PLAN_MACRO;
            std::vector<int> kinds(rs.size(), KIND_K2G_LATER);
            kinds[0] = KIND_K2G_FIRST;
            rc = append_general_passes<T>(plan, n, rs, kinds, nullptr, nullptr);
            if (rc) return rc;
            for (auto& pd : plan.passes) pd.row_n = 0;
            return MI355FFT_OK;
        };
        // the host planner's six-step split (mi355fft_plan_options.recipe, a MixedRadix / GoodThomas root: mixed_radix.rs:53-158):
        // its leaves become the pass heights when every one of them has a compiled column tile
        if (plan.recipe_split.size() >= 2 && plan.recipe_split.size() <= 5 && n < ((size_t)1 << 31)) {
            const std::vector<size_t>& rs = plan.recipe_split;
            if (macro_passes(rs, true) == MI355FFT_OK) {
                plan.kind = PLAN_MACRO;
                plan.recipe_status = MI355FFT_RECIPE_STATUS_SPLIT;
                return macro_passes(rs, false);
            }
            bool have = rs.size() <= 4;
            for (size_t p = 0; p < rs.size() && have; ++p)
                have = find_kernel(p == 0 ? KIND_K2G_FIRST : KIND_K2G_LATER, plan.prec, rs[p]) ||
                       find_kernel(p == 0 ? KIND_K2R_FIRST : KIND_K2R_LATER, plan.prec, rs[p]);
            if (have) {
                plan.recipe_status = MI355FFT_RECIPE_STATUS_SPLIT;
                return general_passes(rs);
            }
        }
        // the large-N passes address one transform with 32-bit element offsets
        if (n < ((size_t)1 << 31) && choose_macro_radices(plan.prec, n, radices, !plan.no_fused_resplit, &plan.fused_resplit)) {
            plan.kind = PLAN_MACRO;
            return macro_passes(radices, false);
        }
        // composite lengths above one workgroup: two to four general passes over 13-smooth tile heights (k2g_body) and prime tile
        // heights 37 .. 631 (k2r_body: Rader inside the tile) -- the reference's MixedRadix over Rader inner FFTs for lengths such
        // as 101 x 103 (src/plan.rs:474-506).  At or below 4096 (37 x 41: the one-kernel Bluestein moves each row once) only a host
        // planner's MixedRadix recipe takes the prime tiles.
        // at or below 4096 the stage machine is the one-kernel form of the reference's MixedRadix over Rader: a host planner's MixedRadix request
        // gets it whatever it costs, AUTO where its program is short (measured against the one-kernel Bluestein: profiles/r6/lsm_vs_bluestein_*.jsonl)
        bool rader_body = false;  // (a compiled Rader body -- the primes <= 4096 with 31-smooth p - 1 -- is faster than the stage machine's Rader)
        for (auto& e0 : registry()) rader_body = rader_body || (e0.kind == KIND_RADER && e0.prec == plan.prec && (size_t)e0.aux == n && e0.variant == 0);
        if (!rader_body) {
            // Above 4096 the competitor in front of Bluestein is a two- or three-pass plan over general / prime tile heights (1.2 - 1.9 TB/s: every
            // element crosses HBM four or six times): the stage machine goes first where its program is short enough to beat THAT -- measured with
            // a build that always prefers it against the shipped order, 160 random lengths with a prime factor 37 .. 631 per precision
            // (profiles/r6/lsm_vs_tiles_*.jsonl): Complex<f32> x1.50 / x1.83 at 5 / 7 stages up to 8192 (every one of 20 lengths wins), x1.07 .. x1.15 at
            // 5 .. 7 above (21 wins of 27); Complex<f64> x1.07 / x1.32 up to 8192, x1.08 .. x1.19 at 5 .. 7 above
#if defined(MI355_LSM_FIRST)  // (that measurement build)
            const int cap = 1 << 20;
#else
            const int cap = n <= 4096 ? (1 << 20) : n <= 8192 ? 7 : 0;  // (above 8192 the gain is x1.07 .. x1.15 in the median with losers -- 9990: 1.35 against 1.68 TB/s -- the passes stay first)
#endif
            const int r = try_lsm(algo == MI355FFT_ALGO_MIXED_RADIX && n <= 4096, cap);
            if (r < 0) return -r;
            if (r > 0) return MI355FFT_OK;
        }
        bool general = n < ((size_t)1 << 31) && choose_general_radices(plan.prec, n, radices);
        if (general && n <= 4096) {
            bool prime_tile = false;
            for (size_t r : radices) prime_tile = prime_tile || (r > 31 && is_prime_sz(r));
            general = prime_tile && (algo == MI355FFT_ALGO_MIXED_RADIX || env_int("MI355FFT_K2R_SMALL") == 1);
        }
        if (general) return general_passes(radices);
    }
    // prime length with a compiled Rader body (raders_algorithm.rs:65-124 precomputation, in f64)
    if (rader_ok) {
        for (auto& e0 : registry()) {
            if (e0.kind != KIND_RADER || e0.prec != plan.prec || (size_t)e0.aux != n || e0.variant != 0) continue;
            const KernelEntry* chosen = &e0;
            for (auto& ev : registry())  // tuning: MI355FFT_VARIANT selects an alternative tiling of the same prime
                if (ev.kind == KIND_RADER && ev.prec == plan.prec && ev.aux == e0.aux && ev.variant == env_int("MI355FFT_VARIANT")) chosen = &ev;
            const KernelEntry& e = *chosen;
            if (e.prepare()) return MI355FFT_ERR_HIP;
            plan.kind = PLAN_RADER;
            PassDesc pd{};
            pd.k = &e;
            pd.d_tw = upload<T>(plan, build_subpass_twiddles<T>(e), &rc);
            if (rc) return rc;
            {  // bodies with the register hand-over (kernels.h rader_body MODE 5) run the second inner transform with the sub-passes in
               // reverse order; the other forms ignore this table
                KernelEntry rev = e;
                for (int i = 0; i < rev.np; ++i) rev.radix[i] = e.radix[rev.np - 1 - i];
                pd.d_tw2 = upload<T>(plan, build_subpass_twiddles<T>(rev), &rc);
                if (rc) return rc;
            }
            if ((rc = rader_tables<T>(plan, pd, e.split, 0))) return rc;  // e.split: MODE >= 1 bodies scatter on load
            plan.passes.push_back(pd);
            return MI355FFT_OK;
        }
    }
    // primes beyond one workgroup whose p - 1 factors into general tile heights: multi-kernel Rader (raders_algorithm.rs:126-283; the
    // reference's planner takes Rader for exactly these primes, src/plan.rs:636-665).  Two inner transforms of length p - 1, each two
    // to four column-tile passes; the g^j gather, the spectrum multiply with the x[0] / X[0] step and the g^-j scatter ride on the
    // first load / last stores (k2g_body FUSE 4, 5, 6).  Traffic: 4 P (p - 1) elements against 4 P M, M >= 2p - 1, for the fused
    // Bluestein that served these primes before.  AUTO takes it above 8192 (below, the one-kernel Bluestein through the split
    // exchange moves each row once: measured choice, profiles/r3/rader_large_ab.jsonl); a host planner's MI355FFT_ALGO_RADER
    // gets it for every such prime.
    if (rader_ok && n > 4096 && n < ((size_t)1 << 31) && is_prime_sz(n) && (algo == MI355FFT_ALGO_RADER || n > 8192 || env_int("MI355FFT_RADER_LARGE") == 1) &&
        env_int("MI355FFT_RADER_LARGE") != 2) {
        std::vector<size_t> radices;
        if (choose_general_radices(plan.prec, n - 1, radices) && find_kernel(KIND_K2G_FIRST_GATHER, plan.prec, radices.front()) &&
            find_kernel(KIND_K2G_LAST_RMUL, plan.prec, radices.back()) && find_kernel(KIND_K2G_LAST_SCATTER, plan.prec, radices.back())) {
            PassDesc tabs{};
            if ((rc = rader_tables<T>(plan, tabs, false, 0))) return rc;
            const size_t P = radices.size();
            std::vector<int> k1(P, KIND_K2G_LATER), k2(P, KIND_K2G_LATER);
            k1[0] = KIND_K2G_FIRST_GATHER;
            k1[P - 1] = KIND_K2G_LAST_RMUL;
            k2[0] = KIND_K2G_FIRST;
            k2[P - 1] = KIND_K2G_LAST_SCATTER;
            if ((rc = append_general_passes<T>(plan, n - 1, radices, k1, nullptr, tabs.d_aux1))) return rc;
            plan.passes[0].d_perm_in = tabs.d_perm_in;
            if ((rc = append_general_passes<T>(plan, n - 1, radices, k2, nullptr, nullptr))) return rc;
            plan.passes.back().d_perm_in = tabs.d_perm_out;
            plan.kind = PLAN_RADER_FUSED;
            return MI355FFT_OK;
        }
    }
    // 13-smooth lengths that fit one workgroup: the run-time scheduled mixed-radix kernel (the RadixN analogue)
    if (direct_ok && env_int("MI355FFT_NO_DYN") == 0) {
        DynSched ds;
        const KernelEntry* dk = find_kind(KIND_DYN_K1, plan.prec);
        // measured (profiles/r2/pr_vs_bs_*.jsonl, heavy_vs_bluestein_above_4096.txt): the HEAVY set (prime radices 17 .. 31, 32
        // values per thread) runs at 0.5 - 1.0 TB/s -- behind the one-kernel Bluestein wherever that one exists (n <= 8192:
        // 1.1 - 1.3 TB/s up to 4096, 0.8 - 1.3 through the split exchange above), ahead of the multi-kernel forms.
        // Only the HEAVY set is planned this way: every 13-smooth length has a compiled schedule (<= 4096) or runs in
        // two to four column-tile passes (above), both faster than this kernel (1.0 - 1.8 TB/s).
        // (round 5: the HEAVY set has compiled whole-row schedules up to 16384 now -- kernels_smooth5_* -- and this kernel's LDS layout ends at
        // 8192, so no plan of the full build reaches it any more (tools/kernel_reachability.py); it still serves builds without the generated
        // units, e.g. `make tuning-min`.)
        const bool heavy_loses = algo == MI355FFT_ALGO_AUTO && n <= 8192;
        if (dk && build_dyn_sched(n, 2 * sizeof(T), 0, ds) && ds.light == 2 && !heavy_loses) {
            if (dk->prepare()) return MI355FFT_ERR_HIP;
            plan.kind = PLAN_SINGLE;
            PassDesc pd{};
            pd.k = dk;
            pd.dyn = ds;
            pd.d_tw = upload<T>(plan, build_dyn_twiddles<T>(ds), &rc);
            if (rc) return rc;
            plan.passes.push_back(pd);
            return MI355FFT_OK;
        }
    }
    {  // everything that reaches this point would take Bluestein (or the run-time scheduled Rader below): the stage machine first
        const int r = try_lsm(algo != MI355FFT_ALGO_AUTO);
        if (r < 0) return -r;
        if (r > 0) return MI355FFT_OK;
    }
    // primes whose p - 1 is 13-smooth and that have no compiled body: run-time scheduled Rader.  Measured slower than the
    // one-workgroup Bluestein on MI355X, so AUTO does not pick it; a host planner asks for it with MI355FFT_ALGO_RADER.
    if (algo == MI355FFT_ALGO_RADER) {
        DynSched ds;
        const KernelEntry* rk = find_kind(KIND_DYN_RADER, plan.prec);
        if (rk && build_dyn_sched(n - 1, 2 * sizeof(T), n, ds)) {
            if (rk->prepare()) return MI355FFT_ERR_HIP;
            plan.kind = PLAN_RADER;
            PassDesc pd{};
            pd.k = rk;
            pd.dyn = ds;
            pd.d_tw = upload<T>(plan, build_dyn_twiddles<T>(ds), &rc);
            if (rc) return rc;
            if ((rc = rader_tables<T>(plan, pd, false, 0))) return rc;
            plan.passes.push_back(pd);
            return MI355FFT_OK;
        }
        return MI355FFT_ERR_UNSUPPORTED;
    }
    if (!bluestein_ok) return MI355FFT_ERR_UNSUPPORTED;
    bool bs_preferred = false;
    const BluesteinChoice bc = choose_bluestein(plan.prec, n, plan.opt_bs_mul ? plan.opt_bs_inner : plan.recipe_bs_inner, &bs_preferred);
    if (bs_preferred && plan.recipe_bs_inner == bc.M) plan.recipe_status = MI355FFT_RECIPE_STATUS_SPLIT;
    if (bc.form == 1) {
        // any length that fits one workgroup: Bluestein over the smallest compiled M >= 2n - 1
        if (bc.k1->prepare()) return MI355FFT_ERR_HIP;
        plan.kind = PLAN_BLUESTEIN;
        PassDesc pd{};
        pd.k = bc.k1;
        pd.d_tw = upload<T>(plan, build_subpass_twiddles<T>(*bc.k1), &rc);
        if (rc) return rc;
        KernelEntry rev = *bc.k1;  // the second transform runs the sub-passes in reverse order (kernels.h bluestein_body)
        for (int i = 0; i < rev.np; ++i) rev.radix[i] = bc.k1->radix[rev.np - 1 - i];
        pd.d_tw2 = upload<T>(plan, build_subpass_twiddles<T>(rev), &rc);
        if (rc) return rc;
        if ((rc = bluestein_tables<T>(plan, bc.M, &pd.d_aux1, &pd.d_aux2))) return rc;
        plan.passes.push_back(pd);
        return MI355FFT_OK;
    }
    if (bc.form == 2) {
        // 4096 < n with 2n - 1 <= 32768 (f64: 16384): two-kernel Bluestein, each kernel one whole-row transform of the
        // padded length through the split exchange (k1bs_body)
        if (bc.k1->prepare() || bc.k2->prepare()) return MI355FFT_ERR_HIP;
        PassDesc pd{};
        pd.k = bc.k1;
        pd.row_n = (long long)bc.M;
        pd.d_tw = upload<T>(plan, build_subpass_twiddles<T>(*bc.k1), &rc);
        if (rc) return rc;
        if ((rc = bluestein_tables<T>(plan, bc.M, &pd.d_aux1, &pd.d_aux2))) return rc;
        plan.passes.push_back(pd);
        pd.k = bc.k2;
        plan.passes.push_back(pd);
        plan.kind = PLAN_BLUESTEIN_2K;
        return MI355FFT_OK;
    }
    if (bc.form == 3) {
        // multi-kernel Bluestein (bluesteins_algorithm.rs:58-136) with the inner FFT_M realised by the general column-tile
        // passes and the three element-wise stages fused into their first load / last store
        void *d_chirp = nullptr, *d_bf = nullptr;
        if ((rc = bluestein_tables<T>(plan, bc.M, &d_chirp, &d_bf))) return rc;
        const size_t P = bc.radices.size();
        std::vector<int> k1(P, KIND_K2G_LATER), k2(P, KIND_K2G_LATER);
        k1[0] = KIND_K2G_FIRST_CHIRP;
        k1[P - 1] = KIND_K2G_LAST_MUL;
        k2[0] = KIND_K2G_FIRST;
        k2[P - 1] = KIND_K2G_LAST_CHIRP;
        rc = append_general_passes<T>(plan, bc.M, bc.radices, k1, d_chirp, d_bf);
        if (rc) return rc;
        rc = append_general_passes<T>(plan, bc.M, bc.radices, k2, nullptr, d_chirp);
        if (rc) return rc;
        plan.kind = PLAN_BLUESTEIN_FUSED;
        return MI355FFT_OK;
    }
    if (bc.form == 4) {
        // fallback (and MI355FFT_BLUESTEIN_UNFUSED=1 in tuning builds): separate chirp / multiply kernels around an inner plan
        const KernelEntry* pw = nullptr;
        for (auto& e : registry())
            if (e.kind == KIND_POINTWISE && e.prec == plan.prec) pw = &e;
        if (pw) {
            plan.inner.reset(new Plan());
            plan.inner->len = bc.M;
            plan.inner->direction = MI355FFT_FORWARD;
            plan.inner->prec = plan.prec;
            plan.inner->device = plan.device;
            int irc = build_plan_t<T>(*plan.inner);
            if (irc) return irc;
            plan.kind = PLAN_BLUESTEIN_LARGE;
            PassDesc pd{};
            pd.k = pw;
            if ((rc = bluestein_tables<T>(plan, bc.M, &pd.d_aux1, &pd.d_aux2))) return rc;
            plan.passes.push_back(pd);
            return MI355FFT_OK;
        }
    }
    return MI355FFT_ERR_UNSUPPORTED;
}

// The host planner's Recipe tree (mi355fft_plan_options.recipe; `enum Recipe`, src/plan.rs:134-188).  Checked the way the
// reference's constructors assert, then reduced to what the GPU planner can use: the top-level family, the six-step split of a
// MixedRadix / GoodThomas root, the inner length of a Bluesteins root.
int apply_recipe(Plan& plan, const char** why) {
    const std::vector<mi355fft_recipe_node>& r = plan.recipe;
    const int n = (int)r.size();
    auto bad = [&](const char* m) {
        *why = m;
        return MI355FFT_ERR_INVALID_ARG;
    };
    if (n == 0) return MI355FFT_OK;
    auto child_ok = [&](int i, int c) { return c > i && c < n; };  // children follow their parent: no cycles, no sharing upward
    for (int i = 0; i < n; ++i) {
        const mi355fft_recipe_node& e = r[i];
        switch (e.kind) {
        case MI355FFT_RECIPE_DFT:
        case MI355FFT_RECIPE_BUTTERFLY:
            if (e.left != -1 || e.right != -1) return bad("recipe: a Dft / Butterfly node has no children");
            break;
        case MI355FFT_RECIPE_MIXED_RADIX:
        case MI355FFT_RECIPE_GOOD_THOMAS:
        case MI355FFT_RECIPE_MIXED_RADIX_SMALL:
        case MI355FFT_RECIPE_GOOD_THOMAS_SMALL:
            if (!child_ok(i, e.left) || !child_ok(i, e.right)) return bad("recipe: MixedRadix / GoodThomas needs left_fft and right_fft after the node");
            if (r[e.left].len == 0 || r[e.right].len == 0 || r[e.left].len > e.len || (e.len / r[e.left].len) != r[e.right].len ||
                e.len % r[e.left].len != 0)
                return bad("recipe: left_fft.len() * right_fft.len() != len (mixed_radix.rs:53-62)");
            break;
        case MI355FFT_RECIPE_RADERS:
            if (!child_ok(i, e.left) || e.right != -1) return bad("recipe: RadersAlgorithm needs inner_fft (left) only");
            if (e.len < 3 || r[e.left].len != e.len - 1) return bad("recipe: RadersAlgorithm inner_fft.len() != len - 1 (raders_algorithm.rs:68-78)");
            break;
        case MI355FFT_RECIPE_BLUESTEINS:
            if (!child_ok(i, e.left) || e.right != -1) return bad("recipe: BluesteinsAlgorithm needs inner_fft (left) only");
            if (e.len == 0 || e.len > ((size_t)-1) / 2 || r[e.left].len < 2 * e.len - 1)
                return bad("recipe: BluesteinsAlgorithm inner_fft.len() < 2 len - 1 (bluesteins_algorithm.rs:55-61)");
            break;
        case MI355FFT_RECIPE_RADIXN:
        case MI355FFT_RECIPE_RADIX4:
            if (!child_ok(i, e.left) || e.right != -1) return bad("recipe: RadixN / Radix4 needs base_fft (left) only");
            if (r[e.left].len == 0 || e.len % r[e.left].len != 0) return bad("recipe: RadixN / Radix4 len is not a multiple of base_fft.len()");
            break;
        default:
            return bad("recipe: unknown node kind");
        }
    }
    if (r[0].len != plan.len) return bad("recipe: the root's len differs from the plan's len");
    int family = MI355FFT_ALGO_MIXED_RADIX;
    if (r[0].kind == MI355FFT_RECIPE_DFT) family = MI355FFT_ALGO_AUTO;  // the planner's Dft(len) is its "no algorithm" answer (plan.rs:313-314)
    if (r[0].kind == MI355FFT_RECIPE_RADERS) family = MI355FFT_ALGO_RADER;
    if (r[0].kind == MI355FFT_RECIPE_BLUESTEINS) family = MI355FFT_ALGO_BLUESTEIN;
    if (plan.algorithm == MI355FFT_ALGO_AUTO)
        plan.algorithm = family;
    else if (family != MI355FFT_ALGO_AUTO && plan.algorithm != family)
        return bad("recipe: `algorithm` names another family than the recipe's root");
    plan.recipe_status = MI355FFT_RECIPE_STATUS_FAMILY;
    auto is_split = [](int k) {
        return k == MI355FFT_RECIPE_MIXED_RADIX || k == MI355FFT_RECIPE_GOOD_THOMAS || k == MI355FFT_RECIPE_MIXED_RADIX_SMALL ||
               k == MI355FFT_RECIPE_GOOD_THOMAS_SMALL;
    };
    plan.recipe_split.clear();
    if (is_split(r[0].kind)) {
        // leaves of the split sub-tree, right (height: transformed first, mixed_radix.rs:128-158) before left (width)
        std::vector<int> stack{0};
        while (!stack.empty()) {
            const int i = stack.back();
            stack.pop_back();
            if (is_split(r[i].kind)) {
                stack.push_back(r[i].left);  // popped second
                stack.push_back(r[i].right);
            } else {
                plan.recipe_split.push_back(r[i].len);
            }
            if (plan.recipe_split.size() > 8) break;
        }
    }
    plan.recipe_bs_inner = r[0].kind == MI355FFT_RECIPE_BLUESTEINS ? r[r[0].left].len : 0;
    return MI355FFT_OK;
}

int build_plan(Plan& plan) {
    ensure_registry();
    plan.device = backend::current_device();
    plan.dbg = env_int("MI355FFT_DBG");  // measurement knobs (tuning builds only; 0 in the shipped library)
    if (env_int("MI355FFT_PIPE")) plan.pipe_mode = env_int("MI355FFT_PIPE");
    if (env_int("MI355FFT_PIPE_MIB")) plan.pipe_slot_bytes = (size_t)env_int("MI355FFT_PIPE_MIB") << 20;
    if (env_int("MI355FFT_PIPE_SLOTS")) plan.pipe_slots = env_int("MI355FFT_PIPE_SLOTS");
    if (env_int("MI355FFT_FUSE")) {  // tuning: 4 = fused launch without the dependency protocol (timing probe), 5 = with it, 7 = + tickets, 8 = off
        plan.fuse_on = env_int("MI355FFT_FUSE") != 8;
        plan.fuse_mode = env_int("MI355FFT_FUSE") & 3;
        plan.fuse_mode |= env_int("MI355FFT_FUSE_PROBE") & 12;  // 4: no release fence, 8: no acquire fence (wrong by design)
    }
    plan.fuse_lag = env_int("MI355FFT_FUSE_LAG");
    plan.fuse_slots = env_int("MI355FFT_FUSE_SLOTS");
    struct TwiddleScope {  // the host planner's compute_twiddle, for this thread, for the duration of the build
        TwiddleScope(mi355fft_twiddle_fn f, void* c) {
            t_twiddle_fn = f;
            t_twiddle_ctx = c;
        }
        ~TwiddleScope() {
            t_twiddle_fn = nullptr;
            t_twiddle_ctx = nullptr;
        }
    } scope(plan.tw_fn, plan.tw_ctx);
    const int rc = plan.prec == 32 ? build_plan_t<float>(plan) : build_plan_t<double>(plan);
    if (rc) return rc;
    // a fused two-pass kernel for exactly this pair of column-tile passes, if one is compiled
    // (three-pass plans: the same kernels fuse passes 0 and 1 over units, execute_fused; the third pass stays a launch of its own)
    const bool two = plan.passes.size() == 2, three = plan.passes.size() == 3;
    if (plan.kind == PLAN_MACRO && (two || three) && plan.passes[0].k->kind == KIND_K2_FIRST && plan.passes[1].k->kind == KIND_K2_LATER)
        for (auto& e : registry())
            if (e.kind == KIND_K2_FUSED && e.prec == plan.prec && !strcmp(e.part[0], plan.passes[0].k->name) && !strcmp(e.part[1], plan.passes[1].k->name) &&
                (e.variant == 0 || (env_int("MI355FFT_FUSE_RING") && e.variant == 10 + env_int("MI355FFT_FUSE_RING") - 100))) {
                if (three && (plan.passes[2].k->n % e.f != 0 || ((long long)plan.passes[0].k->n * e.f) % e.f2 != 0)) continue;
                if (e.prepare()) return MI355FFT_ERR_HIP;
                plan.fused = &e;
                {   // what the chip holds of this kernel: the device's compute units (32 per XCD partition in CPX mode, 256 in SPX) x the
                    // runtime's occupancy for its registers, LDS and block size -- the in-flight window lag and ring are derived from
                    const int cus = backend::cu_count(), per = e.blocks_per_cu ? e.blocks_per_cu() : 0;
                    plan.fuse_resident = (cus > 0 && per > 0) ? cus * per : 256 * (e.threads >= 1024 ? 1 : 2);
                }
                if (two && e.aux == 1 && env_int("MI355FFT_FUSE") == 0) plan.fuse_on = true;  // the measured default for this length
                // three passes: measured for the 256 x 256 pairs (2^23 / 2^24: Complex<f32> +5 % / +3 %, Complex<f64> +21 %,
                // profiles/r4/ab_fused3_*.jsonl); larger lengths have units beyond what the ring holds in the cache
                if (three && e.aux == 1 && plan.len <= ((size_t)1 << 24) && env_int("MI355FFT_FUSE") == 0) plan.fuse_on = true;
                plan.fuse_default = (two && e.aux == 1) || (three && e.aux == 1 && plan.len <= ((size_t)1 << 24));
                if (e.variant != 0) break;  // tuning: the requested ring-access variant wins over the default
            }
    // a split taken only for its fused kernel (Complex<f32> 2^17 as 256 x 512, 2^18 as 256 x 1024, 2^21 as 1024 x 2048) is slower than the
    // balanced one whenever the fused launch cannot run -- a batch with fewer transforms than the ring has slots (the common interactive case),
    // mi355fft_plan_set_fused(plan, 0): 6.33 against 5.92 ms at 2^18.  Such a plan keeps BOTH pass sets: the balanced split as a second plan
    // object (tables only: a few hundred KiB) that execute() takes when it is not going to fuse.
    if (plan.fused_resplit && !plan.no_fused_resplit && plan.fused && plan.fuse_default && plan.kind == PLAN_MACRO && two) {
        std::unique_ptr<Plan> alt(new Plan());
        alt->len = plan.len;
        alt->direction = plan.direction;
        alt->prec = plan.prec;
        alt->algorithm = plan.algorithm;
        alt->tw_fn = plan.tw_fn;
        alt->tw_ctx = plan.tw_ctx;
        alt->no_fused_resplit = true;
        if (int arc = build_plan(*alt)) return arc;
        alt->tw_fn = nullptr;
        alt->tw_ctx = nullptr;
        alt->fuse_on = false;
        alt->fuse_default = false;
        if (alt->kind == PLAN_MACRO && alt->passes.size() == 2) plan.unfused_alt = std::move(alt);
    }
    return MI355FFT_OK;
}

std::string Plan::describe() const {
    if (unfused_alt && !(fuse_on && fused)) return unfused_alt->describe();  // what execute() runs when it does not fuse
    std::ostringstream s;
    if (kind == PLAN_TRIVIAL) s << "trivial(len=" << len << ")";
    if (kind == PLAN_BLUESTEIN_LARGE) {
        s << "bluestein_large(M=" << inner->len << ": " << inner->describe() << ")";
        return s.str();
    }
    if (kind == PLAN_BLUESTEIN_FUSED || kind == PLAN_RADER_FUSED) {
        s << (kind == PLAN_RADER_FUSED ? "rader_large(p-1=" : "bluestein_large(M=") << passes[0].row_n << " fused: ";
        for (size_t i = 0; i < passes.size(); ++i) s << (i == 0 ? "" : (i == passes.size() / 2 ? " | " : " -> ")) << passes[i].k->name;
        s << ")";
        return s.str();
    }
    if (fuse_on && fused) s << "fused{";
    for (size_t i = 0; i < passes.size(); ++i) {
        if (i == 2 && fuse_on && fused) s << "}";  // three passes: the first two in one launch
        s << (i ? (fuse_on && fused && i == 1 ? " | " : " -> ") : "") << passes[i].k->name;
        if (passes[i].k->kind == KIND_LSM)
            s << "<" << passes[i].lsm.desc << ">x" << passes[i].lsm.nt << "t" << passes[i].lsm.nstages << "s" << "F" << passes[i].lsm.f;
        if (passes[i].k->kind == KIND_DYN_K1 || passes[i].k->kind == KIND_DYN_RADER) {
            const DynSched& d = passes[i].dyn;
            s << "<" << d.n << ", " << d.tpf;
            for (int q = 0; q < d.np; ++q) s << ", " << d.radix[q];
            s << ">xF" << d.f;
        }
    }
    if (fuse_on && fused && passes.size() == 2) s << "}";
    return s.str();
}

// ---- workspace -----------------------------------------------------------------------------------------
StreamSlot& Plan::slot_for(void* stream) {
    std::lock_guard<std::mutex> g(ws_mutex);
    std::unique_ptr<StreamSlot>& s = slots[stream];
    if (!s) s.reset(new StreamSlot());
    return *s;  // slots are never erased while the plan lives, so the reference stays valid without ws_mutex
}
// Caller holds slot.launch_mutex for its whole enqueue sequence, so nobody else can be between "fetched the pointer"
// and "launched with it" on this stream: growth only has to wait for work already enqueued on `stream`.
void* Plan::workspace_in(StreamSlot& slot, size_t bytes, void* stream) {
    Workspace& w = slot.ws;
    if (w.bytes < bytes) {
        if (w.ptr) {
            backend::sync(stream);
            backend::dfree(w.ptr);
        }
        w.ptr = backend::dmalloc(bytes);
        w.bytes = w.ptr ? bytes : 0;
        w.placed = false;
    }
    return w.ptr;
}
size_t Plan::workspace_bytes() {
    std::vector<StreamSlot*> all;
    {
        std::lock_guard<std::mutex> g(ws_mutex);
        for (auto& kv : slots) all.push_back(kv.second.get());
    }
    size_t total = 0;
    for (StreamSlot* s : all) {  // ws.bytes changes under the slot's launch lock (workspace_in)
        std::lock_guard<std::mutex> g(s->launch_mutex);
        total += s->ws.bytes + s->pipe.ring.bytes;
    }
    {  // staging buffers of the host-slice path: idle contexts and the ones lent to running calls
        std::lock_guard<std::mutex> g(host_pool_mutex);
        for (auto* pool : {&host_pool, &host_busy})
            for (auto& c : *pool) total += c->in.bytes + c->out.bytes;
    }
    return total + (inner ? inner->workspace_bytes() : 0) + (unfused_alt ? unfused_alt->workspace_bytes() : 0);
}
// Releases every cached workspace (the map of slots stays).  Per slot: take the launch lock FIRST (no caller can enqueue a
// pass that uses the workspace from here on), then drain the device (the stream the slot was used on may be gone; launches
// already enqueued must finish before their workspace is freed), then free.
size_t Plan::trim_workspaces() {
    DeviceGuard dev(device);
    size_t freed = (inner ? inner->trim_workspaces() : 0) + (unfused_alt ? unfused_alt->trim_workspaces() : 0);
    std::vector<StreamSlot*> all;
    {
        std::lock_guard<std::mutex> g(ws_mutex);
        for (auto& kv : slots) all.push_back(kv.second.get());
    }
    for (StreamSlot* s : all) {
        std::lock_guard<std::mutex> g(s->launch_mutex);
        if (!s->ws.ptr && !s->pipe.ring.ptr) continue;
        backend::sync_device();
        freed += s->ws.bytes + s->pipe.ring.bytes;
        backend::dfree(s->ws.ptr);
        backend::dfree(s->pipe.ring.ptr);
        s->ws = Workspace{};
        s->pipe.ring = Workspace{};
    }
    {  // idle staging contexts of the host-slice path (their last call has synchronised its streams before returning them)
        std::lock_guard<std::mutex> g(host_pool_mutex);
        for (auto& c : host_pool) {
            freed += c->in.bytes + c->out.bytes;
            backend::dfree(c->in.ptr);
            backend::dfree(c->out.ptr);
            c->in = Workspace{};
            c->out = Workspace{};
        }
    }
    return freed;
}

// ---- execution -------------------------------------------------------------------------------------------
// A grid above the HIP limit cannot be reached with buffers that fit 288 GB (the smallest workgroup moves 4 KiB), but a
// truncated launch would transform a subset of the rows silently: checked before every launch.
static const long long kMaxGrid = 0x7fffffffLL;
// Parameter block of column-tile pass `pi` (power-of-two tiles: general = false, tile width f).
template <class T>
static int fill_k2_params(const Plan& plan, size_t pi, const void* in, void* out, size_t batch, bool general, int f, const void* xin, void* xout, K2Params<T>& p) {
    const PassDesc& pd = plan.passes[pi];
    const bool inverse = plan.direction == MI355FFT_INVERSE;
    p.in = (const cx<T>*)in;
    p.out = (cx<T>*)out;
    p.tw = (const cx<T>*)pd.d_tw;
    p.tlo = (const cx<T>*)pd.d_tlo;
    p.thi = (const cx<T>*)pd.d_thi;
    p.hshift = pd.hshift;
    p.lmask = pd.lmask;
    p.n = pd.row_n ? pd.row_n : (long long)plan.len;
    p.m = pd.m;
    p.s = pd.s;
    p.batch = (long long)batch;
    p.tab = (const cx<T>*)pd.d_aux1;  // fused Bluestein / Rader passes only
    p.perm = (const int*)pd.d_perm_in;  // fused Rader: g^(j+1) on the gather pass, g^-(j+1) on the scatter pass; prime tiles: both maps
    p.perm2 = (const int*)pd.d_perm_out;
    p.xin = (const cx<T>*)xin;
    p.xout = (cx<T>*)xout;
    p.sgn_x = inverse ? (T)-1 : (T)1;
    p.n_io = (long long)plan.len;
    p.n_valid = (unsigned)plan.len;
    p.tiles_per_fft = general ? (pd.m + f - 1) / f : pd.m / f;
    p.sgn_in = (inverse && pi == 0) ? (T)-1 : (T)1;
    p.sgn_out = (inverse && pi + 1 == plan.passes.size()) ? (T)-1 : (T)1;
    p.dbg = plan.dbg;
    if (!general) {
        while ((1LL << p.tiles_shift) < p.tiles_per_fft) ++p.tiles_shift;
        while ((1LL << p.s_shift) < pd.s) ++p.s_shift;
        if ((1LL << p.tiles_shift) != p.tiles_per_fft || (1LL << p.s_shift) != pd.s) return MI355FFT_ERR_UNSUPPORTED;
    }
    if (!general && !(plan.dbg & 2)) {
        // XCD id at address bits 9..11 of the row segment: tile-index bit k is address bit log2(segment bytes) + k
        int w = 0;
        while ((1LL << (w + 1)) <= (long long)f * (long long)(2 * sizeof(T))) ++w;
        int xp = w < 9 ? 9 - w : 0, xq = 3;
        if (env_int("MI355FFT_XP")) xp = env_int("MI355FFT_XP") - 1;  // tuning builds only
        while (xq > 0 && p.tiles_per_fft % (8LL << xq) != 0) --xq;
        if (p.tiles_per_fft % 8 != 0) xq = 0;
        if (xp > xq) xp = xq;
        p.xp = xp;
        p.xq = xq;
    }
    return MI355FFT_OK;
}
static int grid_too_large(size_t pi, const KernelEntry& k, long long grid, size_t batch) {
    return fail_detail(MI355FFT_ERR_INVALID_ARG, "pass %zu (%s): %zu rows need a grid of %lld workgroups, above the HIP limit of %lld -- split the call", pi, k.name, batch, grid,
                       kMaxGrid);
}
template <class T>
static int launch_pass(const Plan& plan, size_t pi, const void* in, void* out, size_t batch, void* stream, Tracer* tr, const void* xin = nullptr,
                       void* xout = nullptr) {
    const PassDesc& pd = plan.passes[pi];
    const KernelEntry& k = *pd.k;
    const bool inverse = plan.direction == MI355FFT_INVERSE;
    long long grid;
    if (tr) tr->before((int)pi, stream);
    if (k.kind == KIND_K1) {
        K1Params<T> p{};
        p.in = (const cx<T>*)in;
        p.out = (cx<T>*)out;
        p.tw = (const cx<T>*)pd.d_tw;
        p.batch = (long long)batch;
        p.sgn = inverse ? (T)-1 : (T)1;
        grid = (long long)((batch + k.f - 1) / k.f);
        if (grid > kMaxGrid) return grid_too_large(pi, k, grid, batch);
        k.launch(&p, grid, stream);
    } else if (k.kind == KIND_LSM) {
        LsmParams<T> p{};
        p.in = (const cx<T>*)in;
        p.out = (cx<T>*)out;
        p.stages = (const LsmStage*)pd.lsm.d_stages;
        p.desc = (const unsigned*)pd.lsm.d_desc;
        p.ltab = (const cx<T>*)pd.lsm.d_ltab;
        p.gtab = (const cx<T>*)pd.lsm.d_gtab;
        p.ldperm = (const unsigned short*)pd.lsm.d_ldperm;
        p.stperm = (const unsigned short*)pd.lsm.d_stperm;
        p.batch = (long long)batch;
        p.nstages = pd.lsm.nstages;
        p.ltab_n = pd.lsm.ltab_n;
        p.n = pd.lsm.n;
        p.f = pd.lsm.f;
        p.tab_off = pd.lsm.tab_off;
        p.nt = pd.lsm.nt;
        p.lds_bytes = pd.lsm.lds_bytes;
        p.sgn = inverse ? (T)-1 : (T)1;
        grid = (long long)((batch + pd.lsm.f - 1) / pd.lsm.f);
        if (grid > kMaxGrid) return grid_too_large(pi, k, grid, batch);
        k.launch(&p, grid, stream);
    } else if (k.kind == KIND_DYN_K1) {
        DynK1Params<T> p{};
        p.in = (const cx<T>*)in;
        p.out = (cx<T>*)out;
        p.tw = (const cx<T>*)pd.d_tw;
        p.batch = (long long)batch;
        p.sgn = inverse ? (T)-1 : (T)1;
        p.s = pd.dyn;
        grid = (long long)((batch + pd.dyn.f - 1) / pd.dyn.f);
        if (grid > kMaxGrid) return grid_too_large(pi, k, grid, batch);
        k.launch(&p, grid, stream);
    } else if (k.kind == KIND_DYN_RADER) {
        DynRaderParams<T> p{};
        p.in = (const cx<T>*)in;
        p.out = (cx<T>*)out;
        p.tw = (const cx<T>*)pd.d_tw;
        p.d = (const cx<T>*)pd.d_aux1;
        p.perm_in = (const int*)pd.d_perm_in;
        p.perm_out = (const int*)pd.d_perm_out;
        p.batch = (long long)batch;
        p.sgn = inverse ? (T)-1 : (T)1;
        p.s = pd.dyn;
        grid = (long long)((batch + pd.dyn.f - 1) / pd.dyn.f);
        if (grid > kMaxGrid) return grid_too_large(pi, k, grid, batch);
        k.launch(&p, grid, stream);
    } else if (k.kind == KIND_BLUESTEIN || k.kind == KIND_BS2_FIRST || k.kind == KIND_BS2_SECOND) {
        BluesteinParams<T> p{};
        p.in = (const cx<T>*)in;
        p.out = (cx<T>*)out;
        p.tw = (const cx<T>*)pd.d_tw;
        p.tw2 = (const cx<T>*)pd.d_tw2;
        p.chirp = (const cx<T>*)pd.d_aux1;
        p.bf = (const cx<T>*)pd.d_aux2;
        p.batch = (long long)batch;
        p.n = (int)plan.len;
        p.sgn = inverse ? (T)-1 : (T)1;
        grid = (long long)((batch + k.f - 1) / k.f);
        if (grid > kMaxGrid) return grid_too_large(pi, k, grid, batch);
        k.launch(&p, grid, stream);
    } else if (k.kind == KIND_RADER) {
        RaderParams<T> p{};
        p.in = (const cx<T>*)in;
        p.out = (cx<T>*)out;
        p.tw = (const cx<T>*)pd.d_tw;
        p.d = (const cx<T>*)pd.d_aux1;
        p.perm_in = (const int*)pd.d_perm_in;
        p.perm_out = (const int*)pd.d_perm_out;
        p.batch = (long long)batch;
        p.p = (int)plan.len;
        p.sgn = inverse ? (T)-1 : (T)1;
        p.tw2 = (const cx<T>*)pd.d_tw2;
        grid = (long long)((batch + k.f - 1) / k.f);
        if (grid > kMaxGrid) return grid_too_large(pi, k, grid, batch);
        k.launch(&p, grid, stream);
    } else {
        K2Params<T> p{};
        const bool general = (k.kind == KIND_K2G_FIRST || k.kind == KIND_K2G_LATER || k.kind >= KIND_K2G_FIRST_CHIRP);  // incl. the prime tiles
        if (int rc = fill_k2_params<T>(plan, pi, in, out, batch, general, k.f, xin, xout, p))
            return fail_detail(rc, "pass %zu (%s): the column-tile parameters do not fit this kernel (m = %lld, s = %lld, tile width %d)", pi, k.name, pd.m, pd.s, k.f);
        grid = (long long)batch * p.tiles_per_fft;
        if (grid > kMaxGrid) return grid_too_large(pi, k, grid, batch);
        if ((k.kind == KIND_K2G_FIRST || k.kind == KIND_K2G_LATER) && !(plan.dbg & 2)) {
            p.xq = 3;  // k2g_body: XCD-aware order over the global workgroup index (the placement itself is a compile-time constant there)
            p.xfull = (int)(grid >> 6);
        }
        k.launch(&p, grid, stream);
    }
    if (tr) tr->after((int)pi, stream);
    if (backend::check_launch())
        return fail_detail(MI355FFT_ERR_HIP, "launch of pass %zu (%s: grid %lld x %d threads, %zu bytes of LDS, %zu rows of length %zu) failed: %s", pi, k.name, grid,
                           k.threads, k.lds_bytes, batch, plan.len, backend::last_error().c_str());
    return MI355FFT_OK;
}


// ---- chunk pipeline of the multi-pass plans ------------------------------------------------------------------------------
// A P-pass plan moves every element 2 P times between the CUs and memory.  With one full-size workspace all of it is HBM traffic:
// pass p + 1 starts after pass p has written the whole batch (8 GiB at config 2), long after the 256 MiB Infinity Cache has lost
// it.  Here the batch runs in CHUNKS of a few transforms through a small ring of intermediate buffers (2 slots of <= 64 MiB per
// intermediate): pass p + 1 of chunk c reads what pass p wrote microseconds earlier, and the next chunk overwrites the slot
// before its lines need to go anywhere -- the intermediates live in the Infinity Cache, HBM sees one read and one write of the
// caller's buffer per TRANSFORM instead of per pass (tools/mallbench: a tile copy through a 64 MiB ring moves 4 GiB in 2.4 ms
// against 3.1 ms through a 4 GiB workspace).  Mode 2 puts every pass but the last on a side stream, ordered by events, so
// pass p of chunk c + 1 overlaps pass p + 1 of chunk c: HBM reads, ring traffic and HBM writes are in flight together and no
// launch boundary drains the chip.  The caller's buffers are touched by the first pass (reads chunk c of `in`) and the last
// (writes chunk c of `out`) only, so all three API modes are the same code and `in` is never clobbered.  The last pass runs
// on the caller's stream and depends on everything else, so the call stays asynchronous and stream-ordered for the caller
// (and capturable: the side streams fork from and join the caller's stream through events).
#if defined(MI355_TUNING) || defined(MI355_EMU)  // measured slower than one launch per pass AND than the fused launch (profiles/r4/ab_pipe_*): kept as the experiment it was
template <class T> static int execute_pipelined(Plan& plan, const void* in, void* out, size_t batch, void* stream) {
    const size_t esz = 2 * sizeof(T), n = plan.len, P = plan.passes.size();
    const size_t slot_target = plan.pipe_slot_bytes ? plan.pipe_slot_bytes : ((size_t)64 << 20);
    size_t C = std::max<size_t>(1, slot_target / (n * esz));
    if (C > batch) C = batch;
    const size_t NS = plan.pipe_slots > 0 ? (size_t)plan.pipe_slots : 2, slot_bytes = C * n * esz;
    const bool overlap = plan.pipe_mode >= 2;
    StreamSlot& slot = plan.slot_for(stream);
    std::lock_guard<std::mutex> launch_lock(slot.launch_mutex);
    PipeState& pp = slot.pipe;
    const size_t need = (P - 1) * NS * slot_bytes;
    if (pp.ring.bytes < need) {
        if (pp.ring.ptr) {
            backend::sync(stream);  // earlier calls' last passes (which depend on everything the side streams did)
            backend::dfree(pp.ring.ptr);
        }
        pp.ring.ptr = backend::dmalloc(need);
        pp.ring.bytes = pp.ring.ptr ? need : 0;
        if (!pp.ring.ptr) return MI355FFT_ERR_OUT_OF_MEMORY;
    }
    if (overlap) {
        while (pp.side.size() < P - 1) {
            void* s = backend::stream_create();
            if (!s) return MI355FFT_ERR_HIP;
            pp.side.push_back(s);
        }
        while (pp.ev.size() < P * NS) {
            void* e = backend::event_create_notiming();
            if (!e) return MI355FFT_ERR_HIP;
            pp.ev.push_back(e);
        }
        if (!pp.ev_fork && !(pp.ev_fork = backend::event_create_notiming())) return MI355FFT_ERR_HIP;
        backend::event_record(pp.ev_fork, stream);  // the side streams start behind whatever the caller has enqueued so far
        for (size_t p = 0; p + 1 < P; ++p) backend::stream_wait_event(pp.side[p], pp.ev_fork);
    }
    char* ring = (char*)pp.ring.ptr;
    size_t ci = 0;
    for (size_t c0 = 0; c0 < batch; c0 += C, ++ci) {
        const size_t rows = std::min(C, batch - c0), sl = ci % NS;
        for (size_t p = 0; p < P; ++p) {
            void* sp = (overlap && p + 1 < P) ? pp.side[p] : stream;
            const char* src = p == 0 ? (const char*)in + c0 * n * esz : ring + ((p - 1) * NS + sl) * slot_bytes;
            char* dst = p + 1 == P ? (char*)out + c0 * n * esz : ring + (p * NS + sl) * slot_bytes;
            if (overlap) {
                if (p > 0) backend::stream_wait_event(sp, pp.ev[(p - 1) * NS + sl]);                 // my input is written
                if (p + 1 < P && ci >= NS) backend::stream_wait_event(sp, pp.ev[(p + 1) * NS + sl]);  // my output slot has been read
            }
            int rc = launch_pass<T>(plan, p, src, dst, rows, sp, nullptr, nullptr, nullptr);
            if (rc) return rc;
            if (overlap) backend::event_record(pp.ev[p * NS + sl], sp);
        }
    }
    return MI355FFT_OK;
}


#endif

// ---- fused two-pass launch ---------------------------------------------------------------------------------------------------
// Both passes of a two-pass power-of-two plan in ONE launch (launch.h k2f_kernel): pass 2 of transform g - lag runs beside
// pass 1 of transform g, the intermediate goes through a ring of `ns` transform-sized slots that stays in the Infinity Cache
// (HBM sees one read and one write per transform), and there is no launch boundary at which the chip drains.  lag and ns follow
// from how many transforms are in flight: F = ceil(workgroups the chip holds / tiles per step); lag = 2 F steps (the first-pass
// tiles of a transform have long retired when its second-pass tiles come up), ns = 2 lag slots, at most 128 MiB (a slot is not
// rewritten while a resident second-pass tile can still read it -- the counters enforce it; the slack only keeps anybody from waiting).
// Three-pass plans (N = R0 R1 R2, kernels_params.h): the same kernel fuses passes 0 and 1 over UNITS of F0 N / R2 elements (U = R2 / F0 per
// transform, a ring slot holds one unit in compact form); `out` is then the buffer the third pass reads -- never the caller's input: a
// unit's second-pass tiles write into a transform whose other units are still being read.
template <class T> static int execute_fused(Plan& plan, const void* in, void* out, size_t batch, void* stream, bool have_lock) {
    const KernelEntry& k = *plan.fused;
    const size_t esz = 2 * sizeof(T), n = plan.len;
    const bool three = plan.passes.size() == 3;
    K2FusedParams<T> fp{};
    if (int rc = fill_k2_params<T>(plan, 0, in, nullptr, batch, false, k.f, nullptr, nullptr, fp.pass[0])) return rc;
    if (int rc = fill_k2_params<T>(plan, 1, nullptr, out, batch, false, k.f2, nullptr, nullptr, fp.pass[1])) return rc;
    int t0 = (int)fp.pass[0].tiles_per_fft, t1 = (int)fp.pass[1].tiles_per_fft, units = 1;
    if (three) {
        const long long R0 = plan.passes[0].k->n, R2 = plan.passes[2].k->n;
        if (R2 % k.f != 0 || (R0 * k.f) % k.f2 != 0) return MI355FFT_ERR_UNSUPPORTED;
        units = (int)(R2 / k.f);
        t0 /= units;                        // = R1
        t1 = (int)(R0 * k.f / k.f2);        // second-pass tiles over the unit's F0 R0 columns
        fp.pass[1].m = R0 * k.f;            // the ring slot as an (R1) x (F0 R0) matrix
        K2Params<T>* ps[2] = {&fp.pass[0], &fp.pass[1]};
        const int ts[2] = {t0, t1};
        for (int q = 0; q < 2; ++q) {  // the XCD-aware order over the UNIT's tiles (fill_k2_params sized it for the transform's)
            K2Params<T>& pq = *ps[q];
            pq.tiles_per_fft = ts[q];
            if (pq.xq > 0 || pq.xp > 0) {
                int xq = 3;
                while (xq > 0 && ts[q] % (8 << xq) != 0) --xq;
                if (ts[q] % 8 != 0) xq = 0;
                pq.xq = xq;
                if (pq.xp > xq) pq.xp = xq;
            }
        }
    }
    const size_t steps = batch * (size_t)units, slot_elems = n / (size_t)units;
    const int resident = plan.fuse_resident > 0 ? plan.fuse_resident : 256 * (k.threads >= 1024 ? 1 : 2);  // workgroups the chip holds (build_plan)
    const int inflight = (resident + t0 + t1 - 1) / (t0 + t1);
    // measured (profiles/r4/ab_fused_lag_2p*.jsonl): a lag of one in-flight window + 1 leaves second-pass tiles waiting (2^20: 11.3 ms
    // per pair; lag 3: 14.5), two windows do not (11.0), and a ring beyond 128 MiB falls out of the cache's sweet spot (lag 12 / 24
    // slots = 192 MiB: 12.2)
    int lag = plan.fuse_lag > 0 ? plan.fuse_lag : 2 * inflight;
    int ns = plan.fuse_slots > 0 ? plan.fuse_slots : 2 * lag;
    while (plan.fuse_slots <= 0 && ns > lag + 1 && (size_t)ns * slot_elems * esz > ((size_t)128 << 20)) --ns;  // ring <= 128 MiB
    if (lag < 1) lag = 1;         // a second-pass tile waits for first-pass tiles of an EARLIER step (lower indices) only
    if (ns <= lag) ns = lag + 1;  // a first-pass tile of step s waits for second-pass tiles of step s - ns + lag: an earlier step as well
    if (steps < (size_t)ns) return MI355FFT_ERR_UNSUPPORTED;  // fewer steps than ring slots: nothing to overlap
    StreamSlot& slot = plan.slot_for(stream);
    std::unique_lock<std::mutex> launch_lock;
    if (!have_lock) launch_lock = std::unique_lock<std::mutex>(slot.launch_mutex);
    PipeState& pp = slot.pipe;
    const size_t need = (size_t)ns * slot_elems * esz, cbytes = (size_t)k2f_ctrl_words(ns) * sizeof(unsigned);
    if (pp.ring.bytes < need || pp.ctrl_bytes < cbytes || !pp.err_host) {
        backend::sync(stream);
        if (pp.ring.bytes < need) {
            backend::dfree(pp.ring.ptr);
            pp.ring.ptr = backend::dmalloc(need);
            pp.ring.bytes = pp.ring.ptr ? need : 0;
            if (!pp.ring.ptr) return fail_alloc("ring of the fused two-pass launch", need);
        }
        if (pp.ctrl_bytes < cbytes) {
            backend::dfree(pp.ctrl);
            pp.ctrl = backend::dmalloc(cbytes);
            pp.ctrl_bytes = pp.ctrl ? cbytes : 0;
            if (!pp.ctrl) return fail_alloc("control block of the fused two-pass launch", cbytes);
        }
        if (!pp.err_host) {
            pp.err_host = (volatile unsigned*)backend::host_word_alloc(&pp.err_dev);
            if (!pp.err_host) return fail_detail(MI355FFT_ERR_HIP, "pinned error word of the fused two-pass launch: %s", backend::last_error().c_str());
        }
    }
    fp.pass[0].out = (cx<T>*)pp.ring.ptr;
    fp.pass[1].in = (const cx<T>*)pp.ring.ptr;
    fp.ctrl = (unsigned*)pp.ctrl;
    fp.err = (unsigned*)pp.err_dev;
    fp.tiles[0] = t0;
    fp.tiles[1] = t1;
    fp.lag = lag;
    fp.ns = ns;
    fp.batch = (long long)steps;
    fp.units = units;
    fp.slot_elems = (long long)slot_elems;
    fp.mode = plan.fuse_mode;
    fp.spin_limit = plan.fuse_spin_limit < 0 ? 0 : plan.fuse_spin_limit;  // x s_sleep(8) ~ 0.5 us each: about a second by default
    const long long grid = k2f_grid((long long)steps, t0, t1, lag);
    if (grid > kMaxGrid) return grid_too_large(0, k, grid, batch);
    // ticket and dependency counters start at zero for every launch; the error word is NOT part of the block (PipeState::err_host)
    if (backend::memset_async(pp.ctrl, 0, cbytes, stream)) return fail_detail(MI355FFT_ERR_HIP, "zeroing the fused launch's control block: %s", backend::last_error().c_str());
    k.launch(&fp, grid, stream);
    if (backend::check_launch())
        return fail_detail(MI355FFT_ERR_HIP, "fused launch (%s: grid %lld x %d threads, %zu bytes of LDS, %zu steps, lag %d, %d ring slots) failed: %s", k.name, grid, k.threads,
                           k.lds_bytes, steps, lag, ns, backend::last_error().c_str());
    // test hooks: what a tile whose wait gave up leaves behind, on EVERY fused launch (a limit of 0 only produces the give-ups that really
    // occur, which at some sizes is one launch in ten) -- mi355fft_plan_set_fused_wait_limit(plan, -1); the emulator: MI355FFT_FUSED_GIVEUP
    if (plan.fuse_spin_limit < 0) *pp.err_host = 1u;
#if defined(MI355_EMU)
    if (env_int("MI355FFT_FUSED_GIVEUP")) *pp.err_host = 1u;
#endif
    return MI355FFT_OK;
}

// mode: 0 in-place (in == out), 1 out-of-place (input may be clobbered), 2 immutable input
template <class T> static int execute_t(Plan& plan, const void* in, void* out, size_t batch, void* stream, int mode, Tracer* tr, int flags) {
    const bool may_fuse = plan.fuse_on && plan.fused && tr == nullptr && !(flags & EXEC_NO_FUSE);
    const size_t esz = 2 * sizeof(T);
    const size_t n = plan.len;
    if (batch == 0 || n == 0) return MI355FFT_OK;
    if (plan.kind == PLAN_TRIVIAL) {  // len 1: the DFT is the identity (reference plans Dft(1), src/plan.rs:313-314)
        if (in != out && backend::d2d(out, in, batch * n * esz, stream)) return MI355FFT_ERR_HIP;
        return MI355FFT_OK;
    }
    if (plan.kind == PLAN_BLUESTEIN_2K) {
        const size_t M = (size_t)plan.passes[0].row_n;
        size_t chunk = std::max<size_t>(1, ((size_t)1 << 32) / (M * esz));  // <= 4 GiB of padded rows at a time
        if (chunk > batch) chunk = batch;
        StreamSlot& slot = plan.slot_for(stream);
        std::lock_guard<std::mutex> launch_lock(slot.launch_mutex);
        char* ws = (char*)plan.workspace_in(slot, chunk * M * esz, stream);
        if (!ws) return fail_alloc("workspace of the padded rows", chunk * M * esz);
        for (size_t c0 = 0; c0 < batch; c0 += chunk) {
            const size_t rows = std::min(chunk, batch - c0);
            int rc = launch_pass<T>(plan, 0, (const char*)in + c0 * n * esz, ws, rows, stream, c0 == 0 ? tr : nullptr);
            if (rc) return rc;
            rc = launch_pass<T>(plan, 1, ws, (char*)out + c0 * n * esz, rows, stream, c0 == 0 ? tr : nullptr);
            if (rc) return rc;
        }
        return MI355FFT_OK;
    }
    if (plan.kind == PLAN_BLUESTEIN_FUSED || plan.kind == PLAN_RADER_FUSED) {
        // two transforms of length M, P passes each: caller rows -> A [-> B ...] -> (in place) | -> other [...] -> caller rows
        const size_t M = (size_t)plan.passes[0].row_n, P = plan.passes.size() / 2;
        size_t chunk = std::max<size_t>(1, ((size_t)1 << 31) / (M * esz));  // <= 2 x 2 GiB of padded rows at a time
        if (chunk > batch) chunk = batch;
        StreamSlot& slot = plan.slot_for(stream);
        std::lock_guard<std::mutex> launch_lock(slot.launch_mutex);
        char* ws = (char*)plan.workspace_in(slot, 2 * chunk * M * esz, stream);
        if (!ws) return fail_alloc("workspace of the two inner transforms", 2 * chunk * M * esz);
        char* bufs[2] = {ws, ws + chunk * M * esz};
        for (size_t c0 = 0; c0 < batch; c0 += chunk) {
            const size_t rows = std::min(chunk, batch - c0);
            const char* src = (const char*)in + c0 * n * esz;
            int cur = 1;  // the first pass writes bufs[0]
            for (size_t pi = 0; pi < 2 * P; ++pi) {
                const bool last1 = (pi == P - 1), last2 = (pi == 2 * P - 1);
                char* dst = last2 ? (char*)out + c0 * n * esz : last1 ? (char*)src : bufs[1 - cur];
                int rc = launch_pass<T>(plan, pi, src, dst, rows, stream, c0 == 0 ? tr : nullptr, (const char*)in + c0 * n * esz, (char*)out + c0 * n * esz);
                if (rc) return rc;
                if (!last1 && !last2) cur = 1 - cur;
                src = dst;
            }
        }
        return MI355FFT_OK;
    }
    if (plan.kind == PLAN_BLUESTEIN_LARGE) {
        const size_t M = plan.inner->len;
        const PassDesc& pd = plan.passes[0];
        const bool inverse = plan.direction == MI355FFT_INVERSE;
        size_t chunk = std::max<size_t>(1, ((size_t)1 << 31) / (M * esz));  // <= 2 GiB of padded rows at a time
        if (chunk > batch) chunk = batch;
        StreamSlot& slot = plan.slot_for(stream);
        std::lock_guard<std::mutex> launch_lock(slot.launch_mutex);
        char* ws = (char*)plan.workspace_in(slot, chunk * M * esz, stream);
        if (!ws) return fail_alloc("workspace of the padded rows", chunk * M * esz);
        for (size_t c0 = 0; c0 < batch; c0 += chunk) {
            const size_t rows = std::min(chunk, batch - c0);
            PointwiseParams<T> pp{};
            pp.rows = (long long)rows;
            pp.n = (long long)n;
            pp.m = (long long)M;
            pp.sgn = inverse ? (T)-1 : (T)1;
            pp.in = (const cx<T>*)((const char*)in + c0 * n * esz);
            pp.out = (cx<T>*)ws;
            pp.tab = (const cx<T>*)pd.d_aux1;
            pp.stage = 0;
            pd.k->launch(&pp, (long long)(rows * M), stream);
            int rc = execute_t<T>(*plan.inner, ws, ws, rows, stream, 0, nullptr, flags);
            if (rc) return rc;
            pp.in = (const cx<T>*)ws;
            pp.tab = (const cx<T>*)pd.d_aux2;
            pp.stage = 1;
            pd.k->launch(&pp, (long long)(rows * M), stream);
            rc = execute_t<T>(*plan.inner, ws, ws, rows, stream, 0, nullptr, flags);
            if (rc) return rc;
            pp.out = (cx<T>*)((char*)out + c0 * n * esz);
            pp.tab = (const cx<T>*)pd.d_aux1;
            pp.stage = 2;
            pd.k->launch(&pp, (long long)(rows * n), stream);
            if (backend::check_launch()) return fail_detail(MI355FFT_ERR_HIP, "pointwise pass of the large Bluestein plan (inner length %zu) failed: %s", M, backend::last_error().c_str());
        }
        return MI355FFT_OK;
    }
    const size_t P = plan.passes.size();
    if (P == 1) return launch_pass<T>(plan, 0, in, out, batch, stream, tr);
    if (may_fuse && P == 2) {
        const int rcf = execute_fused<T>(plan, in, out, batch, stream, false);
        if (rcf != MI355FFT_ERR_UNSUPPORTED) return rcf;  // a batch too small to pipeline runs as two launches
    }
    // ... of the balanced split when this plan's own split exists for its fused kernel only (build_plan); the profiling hook times the plan's
    // own named kernels
    if (plan.unfused_alt && tr == nullptr && P == 2) return execute_t<T>(*plan.unfused_alt, in, out, batch, stream, mode, nullptr, flags | EXEC_NO_FUSE);
#if defined(MI355_TUNING) || defined(MI355_EMU)
    if (plan.pipe_mode > 0 && tr == nullptr && plan.kind == PLAN_MACRO) return execute_pipelined<T>(plan, in, out, batch, stream);
#endif

    // Buffer rotation.  Every pass but the last is out-of-place; the last one may run in place.
    //   in-place : buf -> ws -> buf -> ws ... -> buf
    //   oop      : in -> out -> in -> out ... -> out   (clobbers `in`, which the trait allows)
    //   immutable: in -> {out, ws alternating, ending ...-> out -> out}
    size_t chunk = plan.chunk_batch ? plan.chunk_batch : batch;
    if (chunk > batch) chunk = batch;
    const bool need_ws = (mode == 0) || (mode == 2 && P >= 3);
    char* ws = nullptr;
    // the slot lock spans the workspace lookup AND every pass launch below (see StreamSlot): calls that use no
    // workspace touch only the caller's own buffers and need no lock
    std::unique_lock<std::mutex> launch_lock;
    if (need_ws) {
        StreamSlot& slot = plan.slot_for(stream);
        launch_lock = std::unique_lock<std::mutex>(slot.launch_mutex);
        ws = (char*)plan.workspace_in(slot, chunk * n * esz, stream);
        if (!ws) return fail_alloc("workspace of a multi-pass plan", chunk * n * esz);
        // Workspace placement (OPT-IN: mi355fft_plan_set_workspace_placement).  Identical kernels on identical data run 3 - 4 % apart
        // depending on WHICH device allocation the workspace is (measured in one process: eight plans of 2^20 x 1024, each with its own
        // 8 GiB workspace, fall into two groups, 5.61 / 5.52 TB/s and 5.42 / 5.32, independent of the workspace's offset inside its
        // allocation and of its distance to the caller's buffer: profiles/r3/ab_ws_offset_probe*.jsonl).  With the option set, a large
        // workspace is chosen by measurement once per (plan, stream, size): up to three allocations, the first pass of this very call
        // timed into each, the fastest kept.  That first call BLOCKS the host, allocates up to 3x the workspace for its duration and
        // cannot be captured into a graph -- which is why it is not the default behind an API documented as asynchronous.
        if (plan.place_workspace && !slot.ws.placed && mode == 0 && chunk * n * esz >= ((size_t)256 << 20) && tr == nullptr && !(plan.dbg & 4)) {
            slot.ws.placed = true;
            const size_t bytes = chunk * n * esz, cb0 = std::min(chunk, batch);
            void* cand[3] = {slot.ws.ptr, backend::dmalloc(bytes), backend::dmalloc(bytes)};
            if (!cand[1] || !cand[2]) backend::check_launch();  // a failed candidate allocation must not leave a sticky error for the next launch check
            float best_ms = 0;
            int best = 0;
            void *e0 = backend::event_create(), *e1 = backend::event_create();
            for (int c = 0; c < 3 && e0 && e1; ++c) {
                if (!cand[c]) continue;
                float ms = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {  // the first repetition also warms the candidate's pages
                    backend::event_record(e0, stream);
                    if (launch_pass<T>(plan, 0, in, cand[c], cb0, stream, nullptr)) break;
                    backend::event_record(e1, stream);
                    const float t = backend::event_elapsed_ms(e0, e1);
                    if (rep > 0 && t < ms) ms = t;
                }
                if (c == 0 || ms < best_ms) {
                    best_ms = ms;
                    best = c;
                }
            }
            if (e0) backend::event_destroy(e0);
            if (e1) backend::event_destroy(e1);
            backend::sync(stream);
            for (int c = 0; c < 3; ++c)
                if (c != best && cand[c]) backend::dfree(cand[c]);
            slot.ws.ptr = cand[best];
            ws = (char*)slot.ws.ptr;
        }
    } else {
        chunk = batch;
    }
    for (size_t c0 = 0; c0 < batch; c0 += chunk) {
        const size_t cb = std::min(chunk, batch - c0);
        const char* cin = (const char*)in + c0 * n * esz;
        char* cout = (char*)out + c0 * n * esz;
        char *A, *B;
        if (mode == 0) {
            A = ws;
            B = cout;
        } else if (mode == 1) {
            A = cout;
            B = (char*)cin;
        } else {
            A = (P % 2 == 0) ? cout : ws;
            B = (P % 2 == 0) ? ws : cout;
        }
        const char* src = cin;
        size_t p0 = 0;
        if (may_fuse && P == 3 && plan.kind == PLAN_MACRO) {
            // passes 0 and 1 in one launch, cin -> A (never back into the caller's input: see execute_fused); the third pass reads A
            const int rcf = execute_fused<T>(plan, cin, A, cb, stream, need_ws);
            if (rcf != MI355FFT_ERR_UNSUPPORTED) {
                if (rcf) return rcf;
                src = A;
                p0 = 2;
            }
        }
        for (size_t p = p0; p < P; ++p) {
            char* dstp = (p + 1 == P) ? cout : ((p % 2 == 0) ? A : B);
            int rc = launch_pass<T>(plan, p, src, dstp, cb, stream, tr);
            if (rc) return rc;
            src = dstp;
        }
    }
    return MI355FFT_OK;
}

unsigned fused_check(Plan& plan, void* stream, bool clear, bool all_streams) {
    unsigned word = 0;
    std::lock_guard<std::mutex> g(plan.ws_mutex);  // the map; err_host itself is written once (under the slot's launch lock) and only ever read here
    for (auto& kv : plan.slots) {
        if (!all_streams && kv.first != stream) continue;
        volatile unsigned* w = kv.second->pipe.err_host;
        if (!w) continue;
        word |= *w;
        if (clear && *w) *w = 0;
    }
    if (plan.inner) word |= fused_check(*plan.inner, stream, clear, all_streams);
    return word;
}

int execute(Plan& plan, const void* in, void* out, size_t batch, void* stream, int mode, Tracer* tr, int flags) {
    t_detail.clear();
    // A fused launch enqueued earlier on this plan and stream gave up a dependency wait (launch.h k2f_wait): what it wrote is invalid, and the
    // caller of an asynchronous entry point can only learn it here -- this call fails INSTEAD of running (src/lib.rs:184: an Fft is never silently wrong)
    if ((plan.fused || plan.inner) && !(flags & EXEC_NO_STICKY_CHECK) && fused_check(plan, stream, true))
        return fail_detail(MI355FFT_ERR_HIP,
                           "an earlier fused two-pass launch of this plan on this stream gave up waiting for a dependency (device oversubscribed?): the results of that "
                           "call are INVALID; this call was not run.  Re-run both, or mi355fft_plan_set_fused(plan, 0) for one launch per pass");
    DeviceGuard dev(plan.device);  // launches and workspace allocations go to the device that holds the tables
    return plan.prec == 32 ? execute_t<float>(plan, in, out, batch, stream, mode, tr, flags)
                           : execute_t<double>(plan, in, out, batch, stream, mode, tr, flags);
}

}  // namespace mi355
